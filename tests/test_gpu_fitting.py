"""End-to-end: model_galaxy_mcmc on the reference's example field (BASELINE
config 1 shape: 64 walkers x 50 iterations) through the batched GPU posterior."""
import os
import shutil

import numpy as np
import pytest

import helpers
import synth_field

pytestmark = pytest.mark.gpu


def test_model_galaxy_mcmc_example(tmp_path):
    from psfmc_amd import model_galaxy_mcmc, load_database, fits_io, MultiComponentModel
    src = os.path.join(helpers.GOLDEN, 'example')
    for name in os.listdir(src):
        if os.path.isfile(os.path.join(src, name)):       # a stray __pycache__ is not a fixture
            shutil.copy(os.path.join(src, name), tmp_path)
    model_file = str(tmp_path / 'model_example.py')
    out = str(tmp_path / 'out_example')
    np.random.seed(42)                                   # prior draws use the global numpy RNG
    model, db = model_galaxy_mcmc(model_file, output_name=out, iterations=50, burn=30, chains=64,
                                  random_state=7, quiet=True)
    # trace database
    assert os.path.exists(out + '_db.fits')
    db2 = load_database(out + '_db.fits')
    assert len(db2) == 64 * 50 and db2.meta['MCITER'] == 50 and db2.meta['MCBURN'] == 30
    assert db2.meta['MCCHAINS'] == 64 and 0.0 < db2.meta['MCACCEPT'] < 1.0
    assert db2.colnames[:3] == ['0_Sky_adu', '1_PointSource_mag', '1_PointSource_xy']
    # every stored log-probability is reproducible from its stored parameters
    theta = db2.param_matrix(model.param_names)
    again = model.log_posterior_batch(theta[::37])
    assert helpers.rel_err(again, db2['lnprobability'][::37]) <= 1e-12
    # the chain moved towards higher probability during 80 iterations
    lnp = db2['lnprobability'].reshape(64, 50)
    assert np.median(lnp[:, -1]) > np.median(lnp[:, 0]) - 1.0
    assert model.accumulated_samples == 64 * 50
    # posterior images: written, finite, and consistent with each other
    imgs = {}
    for kind in ('raw_model', 'convolved_model', 'composite_ivm', 'residual', 'point_source_subtracted'):
        data, hdr = fits_io.read_image(out + '_' + kind + '.fits', with_header=True)
        assert data.shape == (128, 128) and np.isfinite(data).all()
        assert hdr['OBJECT'] == kind and hdr['MCITER'] == 50
        imgs[kind] = data
    sci = model.config.obs_data.astype(np.float64)
    assert np.allclose(imgs['residual'], sci - imgs['convolved_model'], atol=1e-9)
    assert np.all(imgs['composite_ivm'] > 0)
    # weighted-mean images equal the mean of the per-sample images of the kept rows
    # (all walkers kept here unless some are lost): check against a direct recomputation
    from psfmc_amd.database import filter_lowp_walkers
    kept = filter_lowp_walkers(db2, percentile=10)
    if len(kept) == len(db2):
        direct = model.sample_images(theta[:256], ('convolved_model',))['convolved_model']
        model.reset_images()
        model.accumulate_images({'convolved_model': direct})
        assert np.allclose(model.posterior_images['convolved_model'], direct.mean(axis=0), rtol=1e-12)
    # a second call finds the database and skips sampling
    model2, db3 = model_galaxy_mcmc(model_file, output_name=out, iterations=50, burn=30, chains=64,
                                    quiet=True)
    assert np.array_equal(db3['lnprobability'], db2['lnprobability'])
    model.close()
    model2.close()


@pytest.mark.parametrize('backend', ['fused', 'hipfft'])
def test_device_accumulation_matches_reference_running_mean(tmp_path, backend):
    """psfmc_accumulate_images vs the reference's running mean (models.py:74-97)
    restated with numpy on the per-sample images."""
    case = helpers.load_case('edge')
    model = helpers.build_model('edge', case, tmp_path, backend=backend, max_walkers=16)
    ok = np.isfinite(case['lnprob'])
    theta = case['params'][ok][:40]
    imgs = model.sample_images(theta)
    want = {k: np.ones(imgs[k].shape[1:]) for k in imgs}
    with np.errstate(all='ignore'):
        want['composite_ivm'] = 1 / want['composite_ivm']
        for i in range(len(theta)):
            n = i + 1
            for k in imgs:
                step = 1 / imgs[k][i] if k == 'composite_ivm' else imgs[k][i]
                want[k] = (want[k] * (n - 1) + step) / n
        want['composite_ivm'] = 1 / want['composite_ivm']
    model.reset_images()
    model.accumulate_samples(theta[:13])             # slices larger than max_walkers too
    model.accumulate_samples(theta[13:])
    assert model.accumulated_samples == 40
    got = model.collect_posterior_images()
    for k in want:
        fin = np.isfinite(want[k])
        assert np.array_equal(np.isfinite(got[k]), fin), k
        # The fused back end adds samples up as raw / raw^2 / PS-only raw and convolves the sums once
        # (linear: DESIGN section 8), the running mean above is over per-sample convolutions.  On this
        # fixture (1e4-count point sources) a per-sample packed transform leaks 3.5e-12 of the peak from
        # its variance channel into the model channel and the weight map is conditioned to a few 1e-6
        # either way (both measured against the fp64 oracle: the summed form is the closer one).
        peak = np.abs(want[k][fin]).max()
        if k == 'composite_ivm':
            assert np.allclose(got[k][fin], want[k][fin], rtol=1e-5, atol=0), k
        else:
            assert np.abs(got[k][fin] - want[k][fin]).max() <= 2e-11 * peak, k
    # host-side accumulation of blobs keeps working and merges with device sums
    model.reset_images()
    model.accumulate_samples(theta[:20])
    model.accumulate_images({k: v[20:] for k, v in imgs.items()})
    both = model.collect_posterior_images()
    for k in want:
        fin = np.isfinite(want[k])
        assert np.allclose(both[k][fin], want[k][fin], rtol=1e-11, atol=1e-13 * np.abs(want[k][fin]).max()), k
    model.close()


def test_device_sampler_reproduces_host_sampler(tmp_path):
    """Walkers resident on the GPU (psfmc_stretch_run) vs the host loop around the
    batched posterior: same RandomState, same draw order -> the same chain."""
    from psfmc_amd.sampler import EnsembleSampler, DeviceEnsembleSampler
    case = helpers.load_case('synth128x2')
    model = helpers.build_model('synth128x2', case, tmp_path, max_walkers=64)
    np.random.seed(2)
    p0 = model.init_params_from_priors(40)
    host = EnsembleSampler(40, model.num_params, batch_lnpostfn=model.log_posterior_batch)
    dev = DeviceEnsembleSampler(40, model, block=7)
    for s in (host, dev):
        s.random_state = np.random.RandomState(11).get_state()
    out_h = list(host.sample(p0, iterations=25))
    out_d = list(dev.sample(p0, iterations=25))
    assert np.array_equal(dev.chain, host.chain)
    assert np.array_equal(dev.naccepted, host.naccepted)
    assert helpers.rel_err(dev.lnprobability, host.lnprobability) <= 1e-13
    assert np.array_equal(out_d[-1][0], out_h[-1][0]) and len(out_d) == 25
    assert 0.05 < dev.acceptance_fraction.mean() < 0.9
    # half an ensemble of 20 walkers is below the library's `speculate` bound: the run above took ONE pipeline
    # pass per iteration (both candidate proposals of every second-half walker, psfmc_hip.hip stretch_run_impl);
    # two half-steps after each other give the same chain
    assert model.engine.get_option('speculated_runs') > 0
    before = model.engine.get_option('speculated_runs')
    model.engine.set_option('speculate', 0)
    plain = DeviceEnsembleSampler(40, model, block=7)
    plain.random_state = np.random.RandomState(11).get_state()
    list(plain.sample(p0, iterations=25))
    assert model.engine.get_option('speculated_runs') == before
    assert np.array_equal(plain.chain, dev.chain) and np.array_equal(plain.lnprobability, dev.lnprobability)
    assert np.array_equal(plain.naccepted, dev.naccepted)
    model.engine.set_option('speculate', -1)
    # continuing a run (lnprob0 given) and thinning
    more_h = list(host.sample(out_h[-1][0], lnprob0=out_h[-1][1], iterations=6, thin=2))
    more_d = list(dev.sample(out_d[-1][0], lnprob0=out_d[-1][1], iterations=6, thin=2))
    assert dev.chain.shape == (40, 28, model.num_params) and np.array_equal(dev.chain, host.chain)
    # resuming from ANY yielded (pos, lnprob, rstate) continues the same chain, although the
    # device sampler draws the next block's random numbers ahead of time
    for stop in (3, 7, 10):                       # inside a block, at a block edge, in the next
        pos, lnp, rstate = out_d[stop - 1]
        again = DeviceEnsembleSampler(40, model, block=7)
        list(again.sample(pos, lnprob0=lnp, rstate0=rstate, iterations=25 - stop))
        assert np.array_equal(again.chain, host.chain[:, stop:25]), stop
        rstate_h = out_h[stop - 1][2]             # (the host loop yields its arrays by reference)
        assert all(np.array_equal(a, b) for a, b in zip(rstate, rstate_h))
    # replaying one captured iteration as a hipGraph (option "graph") gives the same chain
    model.engine.set_option('graph', 1)
    launches = model.engine.get_option('graph_launches')
    gdev = DeviceEnsembleSampler(40, model, block=9)
    gdev.random_state = np.random.RandomState(11).get_state()
    list(gdev.sample(p0, iterations=25))
    assert model.engine.get_option('graph_launches') > launches
    assert np.array_equal(gdev.chain, host.chain[:, :25])
    model.engine.set_option('graph', 0)
    # accumulation inside the device loop == accumulating every iteration's positions
    model.reset_images()
    acc = DeviceEnsembleSampler(40, model, block=4, accumulate=True)
    acc.random_state = np.random.RandomState(3).get_state()
    steps = list(acc.sample(p0, iterations=5))
    got = {k: v.copy() for k, v in model.collect_posterior_images().items()}
    assert model.accumulated_samples == 200
    model.reset_images()
    for pos, _, _ in steps:
        model.accumulate_samples(pos)
    want = model.collect_posterior_images()
    for k in want:      # (device-derived vs host-derived Sersic constants: 1e-14 apart)
        assert np.allclose(got[k], want[k], rtol=1e-11, atol=1e-12 * np.abs(want[k]).max()), k
    # ... and the same sums when the iteration, accumulation included, is replayed as a hipGraph
    model.engine.set_option('graph', 1)
    model.reset_images()
    launches = model.engine.get_option('graph_launches')
    gacc = DeviceEnsembleSampler(40, model, block=5, accumulate=True)
    gacc.random_state = np.random.RandomState(3).get_state()
    list(gacc.sample(p0, iterations=5))
    assert model.engine.get_option('graph_launches') > launches and model.accumulated_samples == 200
    replay = model.collect_posterior_images()
    for k in got:
        assert np.array_equal(replay[k], got[k], equal_nan=True), k
    model.engine.set_option('graph', 0)
    model.close()


@pytest.mark.parametrize('name', ['edge', 'synth256'])
def test_linear_accumulation_against_the_oracle(tmp_path, name):
    """The fused back end's posterior images -- samples added up as raw / raw^2 / PS-only raw,
    convolved once when asked for -- against the running mean of the fp64 oracle's per-sample images
    (models.py:74-97: the weight map averaged as a variance), two PSFs and a mask included (`edge`);
    and against round 1's way (every sample through the full pipeline, set_option
    'linear_accumulation' 0)."""
    import psfmc_oracle as orc
    case = helpers.load_case(name)
    model = helpers.build_model(name, case, tmp_path, backend='fused', max_walkers=16)
    ok = np.isfinite(case['lnprob'])
    theta = case['params'][ok][:24]
    field = helpers.oracle_field(case)
    layout = helpers.LAYOUT[name]
    sums = {}
    with np.errstate(all='ignore'):
        for t in theta:
            comps, psf = helpers.comps_from_theta(layout, t, name == 'edge')
            _, imgs = orc.evaluate(field, comps, psf, raw_dtype=np.float64, want_ps_sub=True)
            for k, v in imgs.items():
                sums[k] = sums.get(k, 0.0) + (1.0 / v if k == 'composite_ivm' else v)
        want = {k: (len(theta) / v if k == 'composite_ivm' else v / len(theta)) for k, v in sums.items()}
    got = {}
    for linear in (1, 0):
        model.engine.set_option('linear_accumulation', linear)
        assert model.engine.get_option('linear_accumulation') == linear
        model.reset_images()
        model.accumulate_samples(theta[:5])              # a slice, a flush in between, the rest
        if linear:
            model.collect_posterior_images()
        model.accumulate_samples(theta[5:])
        assert model.accumulated_samples == len(theta)
        got[linear] = {k: v.copy() for k, v in model.collect_posterior_images().items()}
    for k in want:
        fin = np.isfinite(want[k])
        peak = np.abs(want[k][fin]).max()
        for linear in (1, 0):
            assert np.array_equal(np.isfinite(got[linear][k]), fin), (k, linear)
        err = np.abs(got[1][k][fin] - want[k][fin]).max() / peak
        old = np.abs(got[0][k][fin] - want[k][fin]).max() / peak
        if k == 'composite_ivm':
            assert err <= 2e-5 and old <= 2e-5, (k, err, old)      # the variance channel's conditioning
        else:
            assert err <= 1e-13 and old <= 1e-11, (k, err, old)     # observed 2e-15 / 3.5e-12 on `edge`
    model.close()


def test_samplers_agree_on_a_general_side_field():
    """A 150 x 150 field (factors 3 and 5: the mixed-radix shapes with idle lanes and spare
    stage-2 slots) through both samplers: the device-resident chain equals the host loop's, and
    the raw-vector path agrees with the hipFFT back end on the same walkers."""
    from test_gpu_fullsize import make_model
    from psfmc_amd.sampler import EnsembleSampler, DeviceEnsembleSampler
    model, fld = make_model(150, 1, 'auto', max_walkers=32)
    assert model._backend == 'fused'
    other, _ = make_model(150, 1, 'hipfft', max_walkers=32)
    p0 = synth_field.draw_walkers(150, 1, 24, seed=4, near_truth=fld['truth'])
    assert helpers.rel_err(model.log_posterior_batch(p0), other.log_posterior_batch(p0)) <= 1e-11
    host = EnsembleSampler(24, model.num_params, batch_lnpostfn=model.log_posterior_batch)
    dev = DeviceEnsembleSampler(24, model, block=5)
    for s in (host, dev):
        s.random_state = np.random.RandomState(8).get_state()
    list(host.sample(p0, iterations=12))
    list(dev.sample(p0, iterations=12))
    assert np.array_equal(dev.chain, host.chain) and np.array_equal(dev.naccepted, host.naccepted)
    assert dev.naccepted.sum() > 0
    model.close()
    other.close()


@pytest.mark.parametrize('n_side', [1152, 1000])
def test_samplers_agree_above_1024(n_side):
    """A side above 1024 (1152: built; 1000: embedded in 1152 with a wrap-around margin): the one-row-per-wave row
    kernels with their lane-pair layout and the three-stage column kernel under BOTH samplers -- the device-resident
    chain equals the host loop's bit for bit, the posterior images accumulated on the device equal the mean of the
    chain's own sample images, and the log-posteriors agree with the hipFFT back end."""
    from test_gpu_fullsize import make_model
    from psfmc_amd.sampler import EnsembleSampler, DeviceEnsembleSampler
    model, fld = make_model(n_side, 1, 'auto', max_walkers=24)
    assert model._backend == 'fused' and model.engine.get_option('rows3') == 3
    other, _ = make_model(n_side, 1, 'hipfft', max_walkers=24)
    p0 = synth_field.draw_walkers(n_side, 1, 22, seed=5, near_truth=fld['truth'])
    assert helpers.rel_err(model.log_posterior_batch(p0), other.log_posterior_batch(p0)) <= 1e-11
    host = EnsembleSampler(22, model.num_params, batch_lnpostfn=model.log_posterior_batch)
    dev = DeviceEnsembleSampler(22, model, block=2, accumulate=True)
    for s in (host, dev):
        s.random_state = np.random.RandomState(3).get_state()
    list(host.sample(p0, iterations=4))
    list(dev.sample(p0, iterations=4))
    assert np.array_equal(dev.chain, host.chain) and np.array_equal(dev.naccepted, host.naccepted)
    assert dev.naccepted.sum() > 0
    post = model.collect_posterior_images()
    flat = dev.chain.transpose(1, 0, 2).reshape(-1, model.num_params)          # every retained sample
    imgs = model.sample_images(flat)
    for kind in ('convolved_model', 'raw_model'):
        want = imgs[kind].mean(axis=0)
        assert np.abs(post[kind] - want).max() <= 1e-11 * np.abs(want).max(), (n_side, kind)
    model.close()
    other.close()
