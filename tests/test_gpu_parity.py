"""GPU parity tests (the parity gate): the HIP path, called through the C ABI,
against (a) the committed reference outputs and (b) the oracle on the same
inputs.  Run on an MI355X with `pytest -m gpu`.

Tolerances
  * vs reference log-posteriors: BASELINE.json asks for <= 1e-5 relative.  The
    reference accumulates the raw model in float32 (SURVEY.md note D); the GPU
    works in fp64 throughout, so the floor is ~1e-7.  Asserted: 1e-6.
  * vs the fp64 oracle: asserted 1e-9 relative (observed ~1e-12).
"""
import numpy as np
import pytest

import helpers
import psfmc_oracle as orc

pytestmark = pytest.mark.gpu

CASES = ['example', 'synth256', 'synth128x2', 'edge']
BACKENDS = ['hipfft', 'fused']
REF_TOL = 1e-6
ORACLE_TOL = 1e-9


@pytest.fixture(scope='module')
def models(tmp_path_factory):
    made = {}

    def get(name, backend):
        key = (name, backend)
        if key not in made:
            case = helpers.load_case(name)
            d = tmp_path_factory.mktemp(name + backend)
            made[key] = (case, helpers.build_model(name, case, d, backend=backend,
                                                   max_walkers=256))
        return made[key]
    yield get
    for _, m in made.values():
        m.close()


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('name', CASES)
def test_log_posterior_matches_reference(models, name, backend):
    case, model = models(name, backend)
    got = model.log_posterior_batch(case['params'])
    ok = helpers.well_conditioned(name, len(got))
    assert helpers.rel_err(got[ok], case['lnprob'][ok]) <= REF_TOL
    assert helpers.rel_err(got[~ok], case['lnprob'][~ok]) <= 1e-5
    # the static single-vector entry point of the reference API
    lp, blobs = type(model).log_posterior(case['params'][0], model=model)
    assert blobs == {} and (lp == got[0])


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('name', CASES)
def test_log_likelihood_matches_fp64_oracle(models, name, backend):
    case, model = models(name, backend)
    fin = np.isfinite(case['lnprior'])
    ll = model.log_likelihood_batch(case['params'][fin])
    ref = case['loglike_f64'][fin]
    ok = helpers.well_conditioned(name, len(case['params']))[fin]
    assert helpers.rel_err(ll[ok], ref[ok]) <= ORACLE_TOL
    assert helpers.rel_err(ll[~ok], ref[~ok]) <= 1e-5


@pytest.mark.parametrize('backend', BACKENDS)
def test_images_match_reference(models, backend):
    for name, rows in (('example', [1]), ('edge', [0, 1])):
        case, model = models(name, backend)
        imgs = model.sample_images(case['params'][rows])
        for j, r in enumerate(rows):
            for kind in imgs:
                ref = case['img%d_%s' % (r, kind)].astype(np.float64)
                got = imgs[kind][j]
                fin = np.isfinite(ref)
                assert np.array_equal(np.isnan(got), np.isnan(ref)), (name, kind)
                scale = np.abs(ref[fin]).max()
                # float32 raw model in the reference: 6e-8 relative rounding
                assert np.abs(got[fin] - ref[fin]).max() <= 3e-7 * scale, (name, kind)


@pytest.mark.parametrize('backend', BACKENDS)
def test_images_match_fp64_oracle(models, backend):
    case, model = models('edge', backend)
    field = helpers.oracle_field(case)
    theta = case['params'][1]
    comps, psf = helpers.comps_from_theta(helpers.LAYOUT['edge'], theta, True)
    _, ref = orc.evaluate(field, comps, psf, raw_dtype=np.float64, want_ps_sub=True)
    imgs = model.sample_images(theta)
    for kind, want in ref.items():
        got = imgs[kind][0]
        fin = np.isfinite(want)
        assert np.abs(got[fin] - want[fin]).max() <= 1e-12 * np.abs(want[fin]).max(), kind


@pytest.mark.parametrize('backend', BACKENDS)
def test_device_psf_spectra_match_rfft2(models, backend):
    """F0: the on-device replacement of pad_and_rfft_image / pre_fft_psf."""
    for name in ('example', 'edge'):
        case, model = models(name, backend)
        field = helpers.oracle_field(case)
        pspec, vspec = model.engine.spectra()
        for k in range(len(field.psf_spec)):
            assert np.abs(pspec[k] - field.psf_spec[k]).max() <= 1e-14
            scale = np.abs(field.var_spec[k]).max()
            assert np.abs(vspec[k] - field.var_spec[k]).max() <= 1e-14 * scale


@pytest.mark.parametrize('backend', BACKENDS)
def test_skip_chunking_and_order_invariance(models, backend):
    case, model = models('synth128x2', backend)
    theta = np.tile(case['params'], (3, 1))[:90]
    base = model.log_posterior_batch(theta)
    # per-walker results do not depend on batch composition or order (bitwise)
    perm = np.random.RandomState(0).permutation(len(theta))
    assert np.array_equal(model.log_posterior_batch(theta[perm]), base[perm])
    assert np.array_equal(model.log_posterior_batch(theta[:7]), base[:7])
    # internal chunking changes nothing
    model.engine.set_option('chunk_walkers', 16)
    assert np.array_equal(model.log_posterior_batch(theta), base)
    # pass sizes that are not multiples of the library's rounding (4 / 8) must stay within
    # the T buffers (regression: 95-walker passes over 4085 walkers ran one walker past them),
    # with one or two passes in flight and with single-pass batches split over both streams
    for chunk, streams, min_split in ((5, 1, 0), (7, 2, 0), (13, 2, 1), (89, 2, 1), (90, 2, 1)):
        model.engine.set_option('chunk_walkers', chunk)
        model.engine.set_option('streams', streams)
        if min_split:
            model.engine.set_option('min_split', min_split)
        assert np.array_equal(model.log_posterior_batch(theta), base), (chunk, streams, min_split)
    model.engine.set_option('min_split', 1 << 30)
    model.engine.set_option('streams', 2)
    model.engine.set_option('chunk_walkers', 256)
    # walkers outside the prior support are skipped and come back -inf
    bad = theta.copy()
    bad[::5, 0] = 99.0                       # PointSource mag outside U(18, 20)
    out = model.log_posterior_batch(bad)
    assert np.all(out[::5] == -np.inf)
    keep = np.ones(len(bad), dtype=bool)
    keep[::5] = False
    assert np.array_equal(out[keep], base[keep])
    # empty batch
    assert model.log_posterior_batch(np.zeros((0, theta.shape[1]))).shape == (0,)


@pytest.mark.parametrize('backend', BACKENDS)
def test_pool_adapter_matches_batch(models, backend):
    from psfmc_amd import BatchLogPosterior
    case, model = models('example', backend)
    blp = BatchLogPosterior(model)
    plist = [p for p in case['params'][:10]]
    res = blp.as_pool().map(None, plist)
    assert [r[0] for r in res] == list(model.log_posterior_batch(case['params'][:10]))
    assert all(r[1] == {} for r in res)
    assert blp.as_lnpostfn()(plist[1])[0] == res[1][0]


def test_native_errors_are_reported(models):
    from psfmc_amd import engine
    case, model = models('example', BACKENDS[0])
    with pytest.raises(ValueError):
        model.engine.loglike(np.zeros((2, 3)))
    with pytest.raises(ValueError):
        model.engine.loglike(np.zeros((1000, model.engine.row_len)))
    with pytest.raises(engine.NativeError):
        model.engine.set_option('no_such_option', 1)


@pytest.mark.parametrize('backend', BACKENDS)
def test_caller_rows_with_a_wild_psf_index_are_clamped(models, backend):
    """The last column of a caller row indexes the kernel spectra on the device; the public
    row entry points round and clamp it to [0, n_psf) instead of reading out of bounds
    (the raw-vector path never gets there: an index outside the prior's support is -inf)."""
    case, model = models('edge', backend)                    # two PSFs
    fin = np.flatnonzero(np.isfinite(case['lnprior']))[:6]
    rows = model.derived_rows(case['params'][fin])
    base = {k: model.engine.loglike(np.column_stack([rows[:, :-1], np.full(len(rows), float(k))]))
            for k in (0, 1)}
    assert not np.array_equal(base[0], base[1])
    for wild, want in ((7.0, 1), (1e300, 1), (-3.0, 0), (np.nan, 0), (0.4, 0), (0.6, 1), (np.inf, 1)):
        r = rows.copy()
        r[:, -1] = wild
        assert np.array_equal(model.engine.loglike(r), base[want]), wild
        imgs = model.engine.images(r[:1], ('convolved_model',))
        assert np.isfinite(imgs['convolved_model']).all()


def test_debug_sweep_reports_plausible_rates():
    """psfmc_debug_sweep (bench.py's `sweep_ceiling`): the three plain sweeps over 64 MiB finish, and
    at rates an MI355X can have (between 1 and 30 TB/s: the buffer may sit in the Infinity Cache)."""
    from psfmc_amd import engine
    nbytes = 64 << 20
    for mode, passes in (('write', 1), ('read_write', 2), ('read', 1)):
        us = engine.debug_sweep(mode, nbytes, reps=5)
        assert 1e3 < passes * nbytes / us / 1e3 < 3e4, (mode, us)
    with pytest.raises(engine.NativeError):
        engine.debug_sweep('read', 1024)                     # below the 1 MiB minimum


def test_device_math():
    """The rasteriser's hand-written fp64 log2 / exp2 / reciprocals vs numpy."""
    from psfmc_amd import engine
    rng = np.random.RandomState(4)
    x = np.concatenate([10.0 ** rng.uniform(-12, 8, 20000), 1.0 + rng.normal(size=4000) * 1e-3,
                        [1.0, 2.0, 0.5, 0.70710678118654746, 0.70710678118654757, 1e-300, 1e300]])
    got = engine.debug_math('log2', x)
    ref = np.log2(x)
    assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)) <= 4e-16
    close = np.abs(x - 1.0) < 0.2                 # relative accuracy near the zero of log
    nz = close & (ref != 0)
    assert np.max(np.abs(got[nz] - ref[nz]) / np.abs(ref[nz])) <= 1e-15
    assert engine.debug_math('log2', np.array([1.0]))[0] == 0.0
    # the rasteriser's table-driven log2: ABSOLUTE accuracy is its contract (it feeds 2^(p log2 x)):
    # one ulp of the result's magnitude (observed 2.2e-16 scaled, the same as the table-free log2)
    edges = 2.0 ** rng.randint(-30, 30, 512) * (1.0 + rng.randint(0, 256, 512) / 256.0)     # table cell borders
    xt = np.concatenate([x, edges, np.nextafter(edges, 0), np.nextafter(edges, np.inf)])
    got = engine.debug_math('log2_tab', xt)
    ref = np.log2(xt.astype(np.longdouble)).astype(np.float64)
    assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)) <= 2.3e-16

    y = np.concatenate([rng.uniform(-60, 60, 20000), rng.uniform(-1075, -1000, 200),
                        [0.0, 1.0, -1.0, 0.5, -0.5, 1023.0, -1074.0, -1100.0, -5000.0, -np.inf]])
    got = engine.debug_math('exp2', y)
    ref = np.exp2(y)
    normal = ref > 1e-300
    assert np.max(np.abs(got[normal] - ref[normal]) / ref[normal]) <= 6e-16
    assert np.all(np.abs(got[~normal] - ref[~normal]) <= 5e-324 + 1e-15 * ref[~normal])
    assert engine.debug_math('exp2', np.array([2000.0, np.inf])).tolist() == [np.inf, np.inf]
    # documented: the min/max clamp turns a NaN argument into 2^-1100 = 0
    assert engine.debug_math('exp2', np.array([np.nan]))[0] == 0.0
    fin = np.abs(y) < 1000
    assert np.array_equal(engine.debug_math('exp2_noclamp', y[fin]), got[fin])
    assert np.isnan(engine.debug_math('exp2_noclamp', np.array([np.nan]))[0])
    # the brightness factor's variant (lower clamp only): the same bits up to overflow, inf beyond
    assert np.array_equal(engine.debug_math('exp2_floor', y), got)
    assert engine.debug_math('exp2_floor', np.array([1500.0, np.nan])).tolist() == [np.inf, 0.0]

    # (rho^2)^p through the per-walker power tables (what the fused rasteriser evaluates per Sersic pixel):
    # relative accuracy of a few ulp for every Sersic index the reference's priors reach, over the whole
    # range of squared radii an image can hold; mantissa-cell borders included
    xp = np.concatenate([10.0 ** rng.uniform(-30, 12, 20000), edges, np.nextafter(edges, 0),
                         np.nextafter(edges, np.inf), [1.0, 2.0 ** -128, 2.0 ** 126]])
    for n_index, bound in ((0.05, 4e-14), (0.3, 1e-15), (0.5, 1e-15), (1.0, 1e-15), (2.5, 1e-15), (4.0, 1e-15),
                           (8.0, 1e-15), (60.0, 1e-15)):
        pw = 0.5 / n_index
        got = engine.debug_pow_tab(xp, pw)
        ref = (xp.astype(np.longdouble) ** np.longdouble(pw)).astype(np.float64)
        ok = np.isfinite(ref) & (ref > 1e-300)
        assert ok.sum() > 0.9 * xp.size
        assert np.max(np.abs(got[ok] - ref[ok]) / ref[ok]) <= bound, n_index
    # outside the exponent table (|log2 x| > 128: no pixel of any image) the exponent is clamped, never read
    # out of range; zero gives a finite value (the pixel is NaN through the centroid term)
    assert np.all(np.isfinite(engine.debug_pow_tab(np.array([0.0, 1e-300, 1e300]), 0.125)))

    z = 10.0 ** rng.uniform(-200, 200, 20000)
    assert np.max(np.abs(engine.debug_math('rcp', z) * z - 1.0)) <= 4e-16
    assert np.isnan(engine.debug_math('rcp', np.array([0.0]))[0] * 0.0)
    assert np.max(np.abs(engine.debug_math('rcp1', z) * z - 1.0)) <= 2e-15
    assert np.isnan(engine.debug_math('rcp1', np.array([0.0]))[0])


@pytest.mark.parametrize('name', CASES)
def test_device_derivation_and_priors_match_host(models, name):
    """Raw-vector path: rows (flux, kappa, Sigma_e, ellipse matrix), log-priors and
    skip flags computed on the device vs scipy on the host (= the reference's calls)."""
    case, model = models(name, 'fused')
    theta = case['params']
    rows, lnprior, skip = model.engine.debug_theta_rows(theta)
    want_prior = model.log_priors_batch(theta)
    assert np.array_equal(skip, ~np.isfinite(want_prior))
    assert helpers.rel_err(np.where(skip, -np.inf, lnprior), np.where(skip, -np.inf, want_prior)) <= 1e-13
    ok = ~skip
    want_rows = model.derived_rows(theta[ok])
    err = np.abs(rows[ok] - want_rows) / np.maximum(np.abs(want_rows), 1e-300)
    assert err.max() <= 5e-14, err.max()
    # and the full posterior through both routes
    dev = model.log_posterior_batch(theta)
    host = model.log_posterior_batch_host(theta)
    cond = helpers.well_conditioned(name, len(theta))
    assert helpers.rel_err(dev[cond], host[cond]) <= 1e-11
    assert helpers.rel_err(dev[~cond], host[~cond]) <= 1e-5


def test_device_kappa_over_the_index_range(tmp_path):
    """kappa = gammaincinv(2n, 1/2) on the device vs scipy, n from 0.05 to 60."""
    from scipy.special import gammaincinv, gamma
    from psfmc_amd import MultiComponentModel
    from psfmc_amd.ModelComponents import Configuration, Sersic
    from psfmc_amd.distributions import Uniform
    case = helpers.load_case('synth128x2')
    cfg = Configuration(case['sci'], case['ivm'], case['psfs'][0], case['psf_ivms'][0], mag_zeropoint=25.0)
    model = MultiComponentModel([cfg, Sersic(xy=(64.2, 63.1), mag=20.0, reff=8.0, reff_b=5.0,
                                             index=Uniform(loc=0.01, scale=100.0), angle=0.3)],
                                max_walkers=512)
    n = np.concatenate([np.linspace(0.05, 1.0, 96), np.linspace(1.0, 12.0, 256), np.linspace(12, 60, 100)])
    rows, _, skip = model.engine.debug_theta_rows(n[:, None])
    assert not skip.any()
    kap = gammaincinv(2 * n, 0.5)
    assert np.max(np.abs(rows[:, 7] - kap) / kap) <= 2e-14
    flux = 10 ** (-0.4 * (20.0 - 25.0))
    sbe = flux / (np.pi * 8.0 * 5.0 * 2 * n * np.exp(kap + np.log(kap) * -2 * n) * gamma(2 * n))
    assert np.max(np.abs(rows[:, 9] - sbe) / sbe) <= 2e-12
    model.close()


def test_host_only_prior_families_are_combined(tmp_path):
    """A prior family the library does not evaluate (here a Gamma distribution) stays
    on the host and is added per walker."""
    from psfmc_amd import MultiComponentModel
    from psfmc_amd.ModelComponents import Configuration, Sersic, Sky
    from psfmc_amd.distributions import Gamma, Normal
    case = helpers.load_case('synth128x2')
    cfg = Configuration(case['sci'], case['ivm'], case['psfs'][0], case['psf_ivms'][0], mag_zeropoint=25.0)
    model = MultiComponentModel([cfg, Sky(adu=Normal(loc=0, scale=0.02)),
                                 Sersic(xy=(64.2, 63.1), mag=20.0, reff=8.0, reff_b=5.0,
                                        index=Gamma(2.0, scale=1.5), angle=0.3)], max_walkers=8)
    model.engine                                     # context + layout are created on first use
    assert len(model._host_priors) == 1
    theta = np.array([[0.01, 2.0], [-0.02, 0.7], [0.0, -1.0]])
    dev = model.log_posterior_batch(theta)
    host = model.log_posterior_batch_host(theta)
    assert dev[2] == -np.inf and helpers.rel_err(dev, host) <= 1e-12
    model.close()
