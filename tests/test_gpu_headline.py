"""GPU tests in the regimes bench.py times: the headline batch (256^2, W = 4096 and the
ragged W = 4085, default options: two passes in flight, library-chosen pass size) and the
single-GPU shares of BASELINE configs 3 and 4 (512^2 x 1024 walkers, 1024^2 x 256 walkers).

The pass-splitting arithmetic (psfmc_hip.hip `pass_size` / `run_pipeline`) is where round
1's one memory bug lived, so known vectors sit at the FIRST and LAST slot of every internal
pass: `synth256` golden vectors against the reference's own log-posteriors (<= 1e-6, the
reference's float32 raw-model floor) and the fp64 oracle (<= 1e-9); 64 random walkers of the
batch against the oracle; and per-walker results bitwise independent of where in the batch a
vector sits."""
import numpy as np
import pytest

import helpers
import psfmc_oracle as orc
import synth_field

pytestmark = pytest.mark.gpu

REF_TOL = 1e-6
ORACLE_TOL = 1e-9


def pass_slots(eng, n_w):
    size = eng.pass_size(n_w)
    starts = list(range(0, n_w, size))
    ends = [min(s + size, n_w) - 1 for s in starts]
    return size, sorted(set(starts + ends))


@pytest.mark.parametrize('n_w', [4096, 4085])
def test_headline_batch_with_golden_vectors_at_every_pass_boundary(tmp_path, n_w):
    case = helpers.load_case('synth256')
    model = helpers.build_model('synth256', case, tmp_path, backend='fused', max_walkers=4096)
    eng = model.engine
    assert eng.get_option('streams') == 2                 # the default the bench times
    size, slots = pass_slots(eng, n_w)
    assert size < n_w and len(slots) >= 2 * (n_w // size)
    half = n_w // 2
    theta = np.vstack([synth_field.draw_walkers(256, 1, half, seed=31),
                       synth_field.draw_walkers(256, 1, n_w - half, seed=32,
                                                near_truth=case['params'][-1])])
    which = np.arange(len(slots)) % len(case['params'])
    theta[slots] = case['params'][which]
    got = model.log_posterior_batch(theta)
    assert got.shape == (n_w,) and not np.isnan(got).any()
    # (1) the reference's own log-posteriors at the pass boundaries
    assert helpers.rel_err(got[slots], case['lnprob'][which]) <= REF_TOL
    # (2) the fp64 oracle there
    want = case['loglike_f64'][which] + case['lnprior'][which]
    want = np.where(np.isfinite(case['lnprior'][which]), want, -np.inf)
    assert helpers.rel_err(got[slots], want) <= ORACLE_TOL
    # (3) 64 random walkers of the batch against the oracle
    field = helpers.oracle_field(case)
    layout = helpers.LAYOUT['synth256']
    pick = np.random.RandomState(5).choice(n_w, 64, replace=False)
    prior = model.log_priors_batch(theta[pick])
    ref = np.array([helpers.oracle_loglike(field, layout, t) if np.isfinite(p) else -np.inf
                    for t, p in zip(theta[pick], prior)]) + np.where(np.isfinite(prior), prior, 0.0)
    assert helpers.rel_err(got[pick], ref) <= ORACLE_TOL
    # (4) position independence, bitwise: the boundary vectors alone, and the batch reversed
    assert np.array_equal(model.log_posterior_batch(theta[slots]), got[slots])
    assert np.array_equal(model.log_posterior_batch(theta[::-1])[::-1], got)
    # (5) the likelihood-only entry point over the same batch agrees with the raw-vector one
    ll = model.log_likelihood_batch(theta[pick])
    fin = np.isfinite(prior)
    assert helpers.rel_err(ll[fin] + prior[fin], got[pick][fin]) <= 1e-12
    model.close()


@pytest.mark.parametrize('name,n_w', [('synth512x2', 1024), ('synth1024x4', 256)])
def test_config3_and_config4_share_at_full_batch(tmp_path, name, n_w):
    """BASELINE config 3 (512^2, 1 PS + 2 Sersic, 1024 walkers) and config 4's per-GPU share
    (1024^2, 1 PS + 4 Sersic, 2048 / 8 = 256 walkers): the whole batch in one call with default
    options.  Vectors of the `light` golden fixtures (the REFERENCE's own log-posteriors at these sizes,
    tests/golden/make_golden.py) sit at the first and last slot of every internal pass -- against the
    reference (<= 1e-6, its float32 raw-model floor) and the fp64 oracle evaluated next to it (<= 1e-9);
    32 random walkers of the batch against the oracle; pass-boundary slots bitwise equal to their
    evaluation in a small batch."""
    n_side, n_sersic = helpers.LIGHT[name]
    case, fld = helpers.load_light_case(name)
    # (the model file make_golden.py gave the reference: FITS images + tools/synth_field.py model_file_text)
    model = helpers.build_model(name, case, tmp_path, backend='fused', max_walkers=n_w)
    eng = model.engine
    size, slots = pass_slots(eng, n_w)
    assert size < n_w
    half = n_w // 2
    theta = np.vstack([synth_field.draw_walkers(n_side, n_sersic, half, seed=41),
                       synth_field.draw_walkers(n_side, n_sersic, n_w - half, seed=42,
                                                near_truth=fld['truth'])])
    n_gold = len(case['params'])
    which = np.arange(len(slots)) % n_gold
    theta[slots] = case['params'][which]
    assert len(slots) >= n_gold or name == 'synth512x2'          # (every vector of the fixture is in the batch)
    spare = [i for i in range(n_gold) if i not in set(which)]
    free = [i for i in range(1, n_w - 1) if i not in set(slots)][:len(spare)]
    theta[free] = case['params'][spare]                          # ... the rest of them anywhere inside a pass
    got = model.log_posterior_batch(theta)
    assert got.shape == (n_w,) and not np.isnan(got).any()
    where, gold = np.array(list(slots) + free), np.concatenate([which, np.array(spare, dtype=int)])
    # (1) the reference's own log-posteriors
    assert (case['lnprob'][gold] == -np.inf).sum() >= 2
    assert helpers.rel_err(got[where], case['lnprob'][gold]) <= REF_TOL
    # (2) the fp64 oracle evaluated next to the reference
    want = np.where(np.isfinite(case['lnprob'][gold]), case['loglike_f64'][gold] + case['lnprior'][gold], -np.inf)
    assert helpers.rel_err(got[where], want) <= ORACLE_TOL
    # (3) 32 random walkers of the batch against the oracle
    field = helpers.oracle_field(case)
    layout = helpers.synth_layout(n_sersic)
    pick = np.random.RandomState(6).choice(n_w, 32, replace=False)
    prior = model.log_priors_batch(theta[pick])
    ref = np.array([helpers.oracle_loglike(field, layout, t) if np.isfinite(p) else -np.inf
                    for t, p in zip(theta[pick], prior)]) + np.where(np.isfinite(prior), prior, 0.0)
    ref = np.where(np.isfinite(ref), ref, -np.inf)
    assert helpers.rel_err(got[pick], ref) <= ORACLE_TOL
    # (4) position independence, bitwise
    small = model.log_posterior_batch(theta[slots[:16]])
    assert np.array_equal(small, got[slots[:16]])
    assert np.array_equal(model.log_posterior_batch(theta[slots[-3:]]), got[slots[-3:]])
    model.close()


def test_config5_share_eight_fields_of_256_walkers():
    """BASELINE config 5's per-GPU share at full size: 8 independent 256^2 fields, one context each,
    256 walkers per field (`bench.py --fields 8 --walkers 256`): two walkers of every field against
    the oracle, results independent of the order the fields are evaluated in."""
    from test_gpu_fullsize import make_model
    models = [make_model(256, 1, 'fused', max_walkers=256, seed=s) for s in range(8)]
    thetas = [synth_field.draw_walkers(256, 1, 256, seed=70 + i, near_truth=fld['truth'])
              for i, (_, fld) in enumerate(models)]
    outs = [m.log_posterior_batch(t) for (m, _), t in zip(models, thetas)]
    layout = helpers.synth_layout(1)
    for (model, fld), theta, got in zip(models, thetas, outs):
        assert np.isfinite(got).all()
        field = orc.make_field(fld['sci'], fld['ivm'], [fld['psf']], [fld['psf_ivm']], mag_zp=fld['mag_zp'])
        pick = [0, 255]
        prior = model.log_priors_batch(theta[pick])
        for i, p in zip(pick, prior):
            want = helpers.oracle_loglike(field, layout, theta[i]) + p
            assert abs(got[i] - want) <= 1e-10 * abs(want), (i, got[i], want)
    for k in reversed(range(8)):                         # interleaved contexts do not disturb each other
        assert np.array_equal(models[k][0].log_posterior_batch(thetas[k]), outs[k])
    for model, _ in models:
        model.close()


def test_field_set_shares_batches_between_fields():
    """`FieldSet` / psfmc_ctx_create_fields: 8 independent 256^2 fields x 256 walkers in ONE context
    and one batch (BASELINE config 5's per-GPU share as it is meant to run): every field's
    log-posteriors equal those of its own one-field context bit for bit, uneven and empty segments
    included; entry points that need a single field refuse."""
    from test_gpu_fullsize import make_model
    from psfmc_amd import FieldSet, engine
    models = [make_model(256, 1, 'fused', max_walkers=256, seed=s) for s in range(8)]
    thetas = [synth_field.draw_walkers(256, 1, 256, seed=90 + i, near_truth=fld['truth'])
              for i, (_, fld) in enumerate(models)]
    alone = [m.log_posterior_batch(t) for (m, _), t in zip(models, thetas)]
    fresh = [make_model(256, 1, 'fused', max_walkers=1, seed=s)[0] for s in range(8)]
    fs = FieldSet(fresh, max_walkers=2048)
    assert fs.context.n_fields == 8 and fs.num_params == models[0][0].num_params
    got = fs.log_posterior_batch(thetas)
    for f in range(8):
        assert np.array_equal(got[f], alone[f]), f
    # uneven shares, a field left out, a single walker
    part = [thetas[0][:5], None, thetas[2][:1], thetas[3][:0], thetas[4][7:100], thetas[5], thetas[6][:33], thetas[7][250:]]
    got = fs.log_posterior_batch(part)
    for f, t in enumerate(part):
        if t is None or len(t) == 0:
            assert len(got[f]) == 0
        else:
            lo = {4: 7, 7: 250}.get(f, 0)
            assert np.array_equal(got[f], alone[f][lo:lo + len(t)]), f
    with pytest.raises(ValueError):
        fs.log_posterior_batch([np.vstack([t, t]) for t in thetas])          # 4096 > max_walkers
    with pytest.raises(engine.NativeError):                                   # a one-field entry point
        fs.context._check(fs.context._lib.psfmc_accumulate_images(fs.context._ctx, 1, None))
    fs.close()
    for m, _ in models:
        m.close()


@pytest.mark.parametrize('side,n_sersic', [(1024, 2), (200, 1), (64, 1), (784, 1), (1152, 1)])
def test_field_set_other_shapes(side, n_sersic):
    """Two fields in one context at nx = 1024 (the inverse row kernel's own multi-field instantiation),
    a general shape and the smallest one; round 4: a side whose inverse row kernel is the three-stage one beside
    a two-stage forward kernel (784) and a side above 1024 (both three-stage): bit-identical to the fields' own
    contexts."""
    from test_gpu_fullsize import make_model
    from psfmc_amd import FieldSet
    own = [make_model(side, n_sersic, 'fused', max_walkers=8, seed=s) for s in (3, 4)]
    thetas = [synth_field.draw_walkers(side, n_sersic, 7, seed=20 + i, near_truth=fld['truth'])
              for i, (_, fld) in enumerate(own)]
    alone = [m.log_posterior_batch(t) for (m, _), t in zip(own, thetas)]
    fs = FieldSet([make_model(side, n_sersic, 'fused', max_walkers=1, seed=s)[0] for s in (3, 4)], max_walkers=16)
    got = fs.log_posterior_batch(thetas)
    assert np.isfinite(got[0]).all() and not np.array_equal(got[0], got[1])
    for f in range(2):
        assert np.array_equal(got[f], alone[f]), f
    fs.close()
    for m, _ in own:
        m.close()


def test_field_set_from_model_files_with_two_psfs_each(tmp_path):
    """`FieldSet` built from model FILES: the `edge` fixture (two PSFs chosen by a free psf_index, a
    mask, bad pixels, out-of-support vectors) as field 0 AND field 1 next to each other -- a walker's
    kernel spectrum is (field x 2 + PSF) -- against the reference's own log-posteriors and, bit for bit,
    the one-field context."""
    from psfmc_amd import FieldSet
    case = helpers.load_case('edge')
    files = []
    for k in range(2):
        d = tmp_path / ('f%d' % k)
        d.mkdir()
        files.append(helpers.write_case_files('edge', case, d))
    alone = helpers.build_model('edge', case, tmp_path, backend='fused', max_walkers=64)
    want = alone.log_posterior_batch(case['params'])
    assert helpers.rel_err(want, case['lnprob']) <= REF_TOL
    fs = FieldSet(files, max_walkers=128)
    assert fs.context.n_psf == 2 and fs.context.n_fields == 2
    half = len(case['params']) // 2
    got = fs.log_posterior_batch([case['params'], case['params'][::-1]])
    assert np.array_equal(got[0], want) and np.array_equal(got[1], want[::-1])
    got = fs.log_posterior_batch([case['params'][:half], case['params'][half:]])
    assert np.array_equal(np.concatenate(got), want)
    assert helpers.rel_err(np.concatenate(got), case['lnprob']) <= REF_TOL
    fs.close()
    alone.close()


def test_device_group_splits_walkers_over_devices(tmp_path):
    """psfmc_group_* (one process, several devices): with the one GPU of the test box listed
    twice the walkers are split over two contexts; results equal the single context's bit for
    bit, for the raw-vector and the derived-row entry points, ragged and tiny batches included."""
    case = helpers.load_case('synth256')
    model = helpers.build_model('synth256', case, tmp_path, backend='fused', max_walkers=128)
    single = model.log_posterior_batch(case['params'])
    assert helpers.rel_err(single, case['lnprob']) <= REF_TOL
    grp = model.device_group([0, 0], max_walkers=128)
    for n in (65, 64, 3, 1):
        assert np.array_equal(grp.logpost_theta(case['params'][:n]), single[:n]), n
    fin = np.isfinite(case['lnprior'])
    rows = model.derived_rows(case['params'][fin])
    assert np.array_equal(grp.loglike(rows), model.engine.loglike(rows))
    skip = np.zeros(len(rows), dtype=bool)
    skip[::3] = True
    got = grp.loglike(rows, skip)
    assert np.all(got[skip] == -np.inf) and np.array_equal(got[~skip], model.engine.loglike(rows)[~skip])
    with pytest.raises(Exception):
        grp.logpost_theta(np.tile(case['params'], (3, 1)))            # 195 > max_walkers
    grp.close()
    model.close()


@pytest.mark.parametrize('name', ['synth256', 'example'])
def test_single_precision_storage_option(tmp_path, name):
    """storage='f32' (complex64 half-spectra between the kernels, fp64 arithmetic): the
    log-posterior stays within the reference's own float32 class -- a few 1e-7 relative, far
    inside BASELINE's 1e-5 -- non-finite cases are unchanged, results stay bitwise independent
    of the batch; shapes it is not built for are refused."""
    case = helpers.load_case(name)
    full = helpers.build_model(name, case, tmp_path, backend='fused', max_walkers=128)
    (tmp_path / 'f32').mkdir()
    from psfmc_amd import MultiComponentModel
    half = MultiComponentModel(helpers.write_case_files(name, case, tmp_path / 'f32'), backend='fused',
                               max_walkers=128, storage='f32')
    assert half.engine.get_option('storage_f32') == 1 and full.engine.get_option('storage_f32') == 0
    got64 = full.log_posterior_batch(case['params'])
    got32 = half.log_posterior_batch(case['params'])
    assert helpers.rel_err(got32, case['lnprob']) <= 5e-6                     # vs the reference
    err = helpers.rel_err(got32, got64)
    assert 0 < err <= 5e-6                                                    # it IS a different rounding
    fin = np.isfinite(got64)
    near = fin & (np.abs(got64) < 1e6)           # walkers near the mode (the prior draws sit at -5e6)
    assert np.abs(got32[near] - got64[near]).max() <= 0.05                    # observed 1e-4 ... 1e-2
    perm = np.random.RandomState(3).permutation(len(got32))
    assert np.array_equal(half.log_posterior_batch(case['params'][perm]), got32[perm])
    assert np.array_equal(half.log_posterior_batch(case['params'][5:9]), got32[5:9])
    imgs = half.sample_images(case['params'][:1], ('convolved_model',))
    ref = full.sample_images(case['params'][:1], ('convolved_model',))
    scale = np.abs(ref['convolved_model']).max()
    assert np.abs(imgs['convolved_model'] - ref['convolved_model']).max() <= 2e-6 * scale
    full.close()
    half.close()


def test_single_precision_storage_is_refused_for_general_sides():
    from test_gpu_fullsize import make_model
    from psfmc_amd import engine
    model, _ = make_model(200, 1, 'fused', max_walkers=8)
    with pytest.raises(engine.NativeError):
        model.engine.set_option('storage_f32', 1)
    model.close()


@pytest.mark.gpu
def test_rasteriser_forms_by_row_length():
    """The fused rasteriser has two forms of (rho^2)^p per Sersic pixel (psfmc_device.h pow_tabs_side): per-walker
    power tables for transforms of more than 256 pixels per row, log2 + exp2 per pixel up to 256 (there the
    tables' L2 traffic costs more than their instructions save).  Both forms are held against the oracle by
    tests/test_gpu_random.py (106 shapes on either side of 256) and tests/test_gpu_fullsize.py; here: the rule
    as the context reports it, and that the table form does not depend on the batch -- small batches form the
    table entries in the row waves, large ones read k_pow_tables' output, subsets and permutations of a batch
    return the same bits (also across the size at which the library switches between the two)."""
    import synth_field
    from test_gpu_fullsize import make_model
    for side, want in ((128, 0.0), (256, 0.0), (264, 1.0), (512, 1.0)):
        m, fld = make_model(side, 2, 'fused', max_walkers=256)
        assert m.engine.get_option('pow_tabs') == want, side
        if side in (264, 512):
            theta = np.vstack([fld['truth'][None, :],
                               synth_field.draw_walkers(side, 2, 199, seed=3, near_truth=fld['truth'])])
            got = m.log_posterior_batch(theta)                       # 400 (walker, component) pairs: k_pow_tables
            assert np.isfinite(got).sum() > 100
            assert np.array_equal(m.log_posterior_batch(theta[5:9]), got[5:9])        # 8 pairs: in the row waves
            # the boundary: batches of up to 8192 (row wave, component) pairs form their entries in the row waves
            last = 8192 // (2 * int(m.engine.get_option('partials_per_walker')))      # walkers of the last such batch
            assert 4 < last < 150
            assert np.array_equal(m.log_posterior_batch(theta[40:40 + last]), got[40:40 + last])
            assert np.array_equal(m.log_posterior_batch(theta[40:41 + last]), got[40:41 + last])
            perm = np.random.RandomState(2).permutation(len(got))
            assert np.array_equal(m.log_posterior_batch(theta[perm]), got[perm])
        m.close()


@pytest.mark.gpu
@pytest.mark.parametrize('side', [288, 512])
def test_power_table_rasteriser_extreme_indices(side):
    """The power-table form over the Sersic indices a prior can reach (n = 0.1 ... 15: exponents p = 1 / (2n) from
    5 down to 1/30), a centre on a pixel corner and one a hair off a pixel centre (rho^2 down to 1e-14), against the
    fp64 oracle: log-likelihood and raw model."""
    from test_gpu_fullsize import make_model
    model, fld = make_model(side, 1, 'fused', max_walkers=16)
    assert model.engine.get_option('pow_tabs') == 1
    field = orc.make_field(fld['sci'], fld['ivm'], [fld['psf']], [fld['psf_ivm']], mag_zp=fld['mag_zp'])
    layout = helpers.synth_layout(1)
    base = fld['truth'].copy()                       # [ps mag, x, y | angle, index, mag, reff, reff_b, x, y]
    thetas = []
    for n_index in (0.1, 0.3, 0.5, 1.0, 2.5, 6.0, 15.0):
        t = base.copy()
        t[4] = n_index
        thetas.append(t)
    for cx, cy in ((side / 2 + 0.5, side / 2 + 0.5), (side / 2 + 1e-7, side / 2 - 3e-8)):
        t = base.copy()
        t[8], t[9] = cx, cy
        thetas.append(t)
    thetas = np.array(thetas)
    got = model.log_likelihood_batch(thetas)
    imgs = model.sample_images(thetas, ('raw_model',))['raw_model']
    for i, t in enumerate(thetas):
        want = helpers.oracle_loglike(field, layout, t)
        # the last case puts a pixel 1e-7 px from the centre: the reference's centroid term makes it 1e9 times its
        # neighbours, and ANY fp64 transform of that image carries eps x 1e9 into the rest (the oracle's numpy FFT and
        # the kernels differ by 1.6e-7 in the log-likelihood there): the raw model is the check, BASELINE's 1e-5 the bound
        tol = 1e-5 if i == len(thetas) - 1 else 2e-10
        assert np.isfinite(want) and abs(got[i] - want) <= tol * abs(want), (side, i, got[i], want)
        comps, psf_index = helpers.comps_from_theta(layout, t)
        _, ref = orc.evaluate(field, comps, psf_index, raw_dtype=np.float64)
        raw = ref['raw_model']
        assert np.all(np.isfinite(raw)) and np.all(np.isfinite(imgs[i]))
        # per pixel, relative: every pixel's Sersic value to a few 1e-14 whatever the image's peak
        big = np.abs(raw) > 1e-280                     # (n = 0.1 underflows to 0 a few r_eff out)
        assert np.max(np.abs(imgs[i][big] - raw[big]) / np.abs(raw[big])) <= 1e-11, (side, i)
        assert np.all(np.abs(imgs[i][~big]) <= 1e-279), (side, i)
    model.close()
