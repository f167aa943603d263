"""GPU tests at the BASELINE.json full sizes (512^2, 1024^2, many fields): the
oracle on a few walkers plus size-independent properties the domain offers
(analytic sky-only / zero-flux likelihoods, flux conservation of the PSF
convolution, batch-order invariance, back-end agreement)."""
import numpy as np
import pytest

import helpers
import psfmc_oracle as orc
import synth_field

pytestmark = pytest.mark.gpu


def make_model(n_side, n_sersic, backend, max_walkers, seed=0):
    from psfmc_amd import MultiComponentModel
    from psfmc_amd.ModelComponents import Configuration, PointSource, Sersic
    from psfmc_amd.distributions import Uniform, WeibullMinimum
    fld = synth_field.make_field(n_side, n_sersic, seed=seed)
    c = np.array((n_side / 2 + 0.5,) * 2)
    comps = [Configuration(fld['sci'], fld['ivm'], fld['psf'], fld['psf_ivm'], mag_zeropoint=fld['mag_zp']),
             PointSource(xy=Uniform(loc=c - 8, scale=16 * np.ones(2)), mag=Uniform(loc=18.0, scale=2.0))]
    for _ in range(n_sersic):
        comps.append(Sersic(xy=Uniform(loc=c - 8, scale=16 * np.ones(2)), mag=Uniform(loc=19.0, scale=5.0),
                            reff=Uniform(loc=2.0, scale=n_side / 16.0),
                            reff_b=Uniform(loc=2.0, scale=n_side / 16.0),
                            index=WeibullMinimum(c=1.5, scale=4), angle=Uniform(loc=0, scale=180),
                            angle_degrees=True))
    return MultiComponentModel(comps, backend=backend, max_walkers=max_walkers), fld


def analytic_no_model(model, sky, psf_sum, psf_var_sum):
    """loglike of a constant-sky model: conv = sky * sum(psf), var = sky^2 * sum(psf_var)
    (circular convolution of a constant); obs_var as the setup stores it (float32 values)."""
    sci = model.config.obs_data.astype(np.float64)
    d = model.config.obs_var.astype(np.float64) + sky * sky * psf_var_sum
    return -0.5 * np.sum((sci - sky * psf_sum) ** 2 / d + np.log(2 * np.pi * d))


@pytest.mark.parametrize('n_side,n_sersic,n_walk', [(512, 2, 96), (1024, 4, 24)])
def test_full_size_against_oracle_and_properties(n_side, n_sersic, n_walk):
    model, fld = make_model(n_side, n_sersic, 'fused', max_walkers=n_walk)
    ref_model, _ = make_model(n_side, n_sersic, 'hipfft', max_walkers=n_walk)
    theta = np.vstack([fld['truth'][None, :],
                       synth_field.draw_walkers(n_side, n_sersic, n_walk // 2 - 1, seed=5),
                       synth_field.draw_walkers(n_side, n_sersic, n_walk // 2, seed=6,
                                                near_truth=fld['truth'])])
    got = model.log_posterior_batch(theta)
    assert np.isfinite(got).all()
    # (1) the two back ends (independent arithmetic: hand-written FFT + fast math vs
    #     hipFFT + OCML literal formula) agree
    other = ref_model.log_posterior_batch(theta)
    assert helpers.rel_err(got, other) <= 1e-11
    # (2) the oracle on a few walkers (0.1-0.5 s each at these sizes)
    field = orc.make_field(fld['sci'], fld['ivm'], [fld['psf']], [fld['psf_ivm']], mag_zp=fld['mag_zp'])
    layout = helpers.synth_layout(n_sersic)
    prior = model.log_priors_batch(theta)
    for i in (0, 1, n_walk - 1):
        want = helpers.oracle_loglike(field, layout, theta[i]) + prior[i]
        assert abs(got[i] - want) <= 1e-10 * abs(want), (i, got[i], want)
    # (3) order / batch-composition invariance, bitwise
    perm = np.random.RandomState(1).permutation(n_walk)
    assert np.array_equal(model.log_posterior_batch(theta[perm]), got[perm])
    assert np.array_equal(model.log_posterior_batch(theta[3:7]), got[3:7])
    # (4) flux conservation: sum(conv) = sum(raw) * sum(psf) (unit-sum PSF, circular convolution)
    imgs = model.sample_images(theta[:2], ('raw_model', 'convolved_model'))
    raw_sum = imgs['raw_model'].sum(axis=(1, 2))
    conv_sum = imgs['convolved_model'].sum(axis=(1, 2))
    psf_sum = float(np.sum(model.config.psf_selector.psf_data[0], dtype=np.float64))   # float32-normalised
    assert np.allclose(conv_sum, raw_sum * psf_sum, rtol=1e-12)
    model.close()
    ref_model.close()


@pytest.mark.parametrize('backend', ['fused', 'hipfft'])
def test_sky_only_model_is_analytic(backend):
    """Sky-only and zero-flux models have closed-form likelihoods at any size."""
    from psfmc_amd import MultiComponentModel
    from psfmc_amd.ModelComponents import Configuration, Sky, PointSource
    from psfmc_amd.distributions import Uniform
    n_side = 512
    fld = synth_field.make_field(n_side, 1, seed=3)
    cfg = Configuration(fld['sci'], fld['ivm'], fld['psf'], fld['psf_ivm'], mag_zeropoint=25.0)
    model = MultiComponentModel([cfg, Sky(adu=Uniform(loc=-1, scale=2)),
                                 PointSource(xy=(200.3, 180.7), mag=Uniform(loc=20, scale=200))],
                                backend=backend, max_walkers=8)
    pvar_sum = float(np.sum(model.config.psf_selector.psf_var[0], dtype=np.float64))
    psf_sum = float(np.sum(model.config.psf_selector.psf_data[0], dtype=np.float64))
    skies = np.array([0.0, 0.01, -0.02, 0.5])
    theta = np.stack([skies, np.full(4, 200.0)], axis=1)        # mag 200: flux 1e-70, negligible
    got = model.log_likelihood_batch(theta)
    want = np.array([analytic_no_model(model, s, psf_sum, pvar_sum) for s in skies])
    assert helpers.rel_err(got, want) <= 1e-12
    model.close()


def test_many_fields_each_with_its_own_context():
    """BASELINE config 5 in miniature: independent fields, one context each,
    walkers stay with their field."""
    models = [make_model(256, 1, 'fused', max_walkers=32, seed=s) for s in range(4)]
    outs = []
    for model, fld in models:
        theta = synth_field.draw_walkers(256, 1, 32, seed=9, near_truth=fld['truth'])
        outs.append(model.log_posterior_batch(theta))
    # same walkers, different data -> different answers; re-evaluation is reproducible
    assert len({round(float(o[0]), 3) for o in outs}) == 4
    for (model, fld), o in zip(models, outs):
        theta = synth_field.draw_walkers(256, 1, 32, seed=9, near_truth=fld['truth'])
        assert np.array_equal(model.log_posterior_batch(theta), o)
        model.close()


def test_auto_backend_keeps_unbuilt_sides_on_the_fused_kernels():
    """backend='auto': a side outside the built list (170 = 2 * 5 * 17) runs on the fused kernels embedded in
    the next built side (round 2: the hipFFT back end), like a built one (140); so does 1000 with its 64-pixel
    PSF since round 4 (embedded in 1152); only a side too large to embed (1986 + 63 > 2048) still goes to hipFFT
    (not run here: its oracle alone takes a minute).  All against the oracle."""
    import psfmc_oracle as orc
    from psfmc_amd import engine
    assert engine.fused_supports(1000, 1000, (64, 64)) and not engine.fused_supports(1986, 1986, (64, 64))
    for side, want_backend in ((170, 'fused'), (140, 'fused'), (1000, 'fused')):
        model, fld = make_model(side, 1, 'auto', max_walkers=8)
        assert model._backend == want_backend
        theta = synth_field.draw_walkers(side, 1, 4, seed=3, near_truth=fld['truth'])
        got = model.log_posterior_batch(theta)
        field = orc.make_field(fld['sci'], fld['ivm'], [fld['psf']], [fld['psf_ivm']], mag_zp=fld['mag_zp'])
        prior = model.log_priors_batch(theta)
        for t, p, g in zip(theta, prior, got):
            want = helpers.oracle_loglike(field, helpers.synth_layout(1), t) + p
            assert abs(g - want) <= 1e-10 * abs(want)
        model.close()


def test_rectangular_and_small_sizes_fused():
    """64 x 128 and 128 x 64: every FFT shape combination the goldens do not hit."""
    from psfmc_amd import MultiComponentModel
    from psfmc_amd.ModelComponents import Configuration, Sky, PointSource, Sersic
    rng = np.random.RandomState(8)
    for ny, nx in ((128, 64), (64, 64), (256, 128), (64, 512)):
        sci = (rng.normal(size=(ny, nx)) * 0.05).astype(np.float32)
        ivm = np.full((ny, nx), 400.0, dtype=np.float32)
        psf = synth_field.moffat_psf(32, fwhm=2.4).astype(np.float32) * 100
        pivm = (1.0 / (0.01 + np.abs(psf) / 30)).astype(np.float32)
        vals = {}
        for backend in ('fused', 'hipfft'):
            cfg = Configuration(sci, ivm, psf, pivm, mag_zeropoint=24.0)
            model = MultiComponentModel(
                [cfg, Sky(adu=0.01), PointSource(xy=(nx / 2 - 3.3, ny / 2 + 2.6), mag=18.0),
                 Sersic(xy=(nx / 2 + 1.2, ny / 2 - 0.7), mag=17.0, reff=6.0, reff_b=3.5, index=2.2,
                        angle=0.7)], backend=backend, max_walkers=4)
            vals[backend] = model.log_likelihood_batch(np.zeros((3, 0)))
            imgs = model.sample_images(np.zeros((1, 0)), ('convolved_model',))
            vals[backend + '_img'] = imgs['convolved_model'][0]
            model.close()
        assert helpers.rel_err(vals['fused'], vals['hipfft']) <= 1e-12, (ny, nx)
        assert np.abs(vals['fused_img'] - vals['hipfft_img']).max() <= 1e-12 * np.abs(vals['hipfft_img']).max()
        field = orc.make_field(sci, ivm, [psf], [pivm], mag_zp=24.0)
        comps = [dict(type='sky', adu=0.01), dict(type='ps', xy=(nx / 2 - 3.3, ny / 2 + 2.6), mag=18.0),
                 dict(type='sersic', xy=(nx / 2 + 1.2, ny / 2 - 0.7), mag=17.0, reff=6.0, reff_b=3.5,
                      index=2.2, angle=0.7, angle_degrees=False)]
        want = orc.log_likelihood(field, comps, raw_dtype=np.float64)
        assert abs(vals['fused'][0] - want) <= 1e-11 * abs(want), (ny, nx)


@pytest.mark.parametrize('backend', ['fused', 'hipfft'])
def test_degenerate_component_sets(backend):
    """Models with no point source / no Sersic / nothing but sky, W = 1, all-skipped batches."""
    from psfmc_amd import MultiComponentModel
    from psfmc_amd.ModelComponents import Configuration, Sky, PointSource, Sersic
    from psfmc_amd.distributions import Uniform
    fld = synth_field.make_field(128, 1, seed=2)
    field = orc.make_field(fld['sci'], fld['ivm'], [fld['psf']], [fld['psf_ivm']], mag_zp=25.0)
    sets = {
        'sky': ([Sky(adu=0.02)], [dict(type='sky', adu=0.02)]),
        'sersic': ([Sersic(xy=(60.2, 66.9), mag=19.5, reff=7.0, reff_b=4.0, index=1.3, angle=20.0,
                           angle_degrees=True)],
                   [dict(type='sersic', xy=(60.2, 66.9), mag=19.5, reff=7.0, reff_b=4.0, index=1.3,
                         angle=20.0, angle_degrees=True)]),
        'ps3': ([PointSource(xy=(10.5 + 30 * k, 100.25 - 20 * k), mag=18.0 + k) for k in range(3)],
                [dict(type='ps', xy=(10.5 + 30 * k, 100.25 - 20 * k), mag=18.0 + k) for k in range(3)]),
    }
    for name, (comps, ocomps) in sets.items():
        cfg = Configuration(fld['sci'], fld['ivm'], fld['psf'], fld['psf_ivm'], mag_zeropoint=25.0)
        model = MultiComponentModel([cfg] + comps, backend=backend, max_walkers=4)
        got = model.log_likelihood_batch(np.zeros((1, 0)))
        want = orc.log_likelihood(field, ocomps, raw_dtype=np.float64)
        assert got.shape == (1,) and abs(got[0] - want) <= 1e-11 * abs(want), name
        model.close()
    # a free-parameter model where every walker is outside the prior support
    cfg = Configuration(fld['sci'], fld['ivm'], fld['psf'], fld['psf_ivm'], mag_zeropoint=25.0)
    model = MultiComponentModel([cfg, Sky(adu=Uniform(loc=0, scale=1))], backend=backend, max_walkers=4)
    assert np.all(model.log_posterior_batch(np.array([[2.0], [-1.0], [5.0]])) == -np.inf)
    assert np.isfinite(model.log_posterior_batch(np.array([[0.5]]))).all()
    model.close()


def test_unbuilt_even_sizes_on_both_back_ends():
    """An even side that is not among the built ones (90): the hipFFT back end transforms it as it is, the
    fused one embeds it in the next built side -- both against the oracle; a side too large to embed is
    refused with a clear message; odd sizes are rejected by the setup like in the reference."""
    from psfmc_amd import MultiComponentModel, engine
    from psfmc_amd.ModelComponents import Configuration, Sky, PointSource, Sersic
    rng = np.random.RandomState(12)
    ny, nx = 100, 90
    sci = (rng.normal(size=(ny, nx)) * 0.05).astype(np.float32)
    ivm = np.full((ny, nx), 400.0, dtype=np.float32)
    psf = synth_field.moffat_psf(21, fwhm=2.4).astype(np.float32) * 100
    pivm = (1.0 / (0.01 + np.abs(psf) / 30)).astype(np.float32)
    comps = [dict(type='sky', adu=0.01), dict(type='ps', xy=(40.3, 52.6), mag=18.0),
             dict(type='sersic', xy=(47.2, 49.3), mag=17.0, reff=6.0, reff_b=3.5, index=2.2, angle=0.7,
                  angle_degrees=False)]

    def build(backend):
        cfg = Configuration(sci, ivm, psf, pivm, mag_zeropoint=24.0)
        return MultiComponentModel([cfg, Sky(adu=0.01), PointSource(xy=(40.3, 52.6), mag=18.0),
                                    Sersic(xy=(47.2, 49.3), mag=17.0, reff=6.0, reff_b=3.5, index=2.2,
                                           angle=0.7)], backend=backend, max_walkers=4)
    model = build('hipfft')
    got = model.log_likelihood_batch(np.zeros((2, 0)))
    field = orc.make_field(sci, ivm, [psf], [pivm], mag_zp=24.0)
    want = orc.log_likelihood(field, comps, raw_dtype=np.float64)
    assert abs(got[0] - want) <= 1e-11 * abs(want)
    model.close()
    model = build('fused')
    got = model.log_likelihood_batch(np.zeros((2, 0)))
    assert abs(got[0] - want) <= 1e-11 * abs(want)
    model.close()
    big = np.zeros((64, 2040), dtype=np.float32)                 # 2040 + 21 - 1 > 2048, the largest built side
    with pytest.raises(engine.NativeError) as err:
        cfg = Configuration(big, big + 400.0, psf, pivm, mag_zeropoint=24.0)
        MultiComponentModel([cfg, Sky(adu=0.01)], backend='fused', max_walkers=4).log_likelihood_batch(np.zeros((1, 0)))
    assert 'exceeds the largest built side' in str(err.value)
    with pytest.raises(ValueError):
        Configuration(sci[:99], ivm[:99], psf, pivm)
