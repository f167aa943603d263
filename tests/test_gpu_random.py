"""Randomised GPU-vs-oracle parity: random image shapes (square and rectangular),
PSF shapes (odd and even), component sets, shift methods, angle units, bad pixels,
masks, several PSFs -- every draw evaluated by both GPU back ends and by the fp64
oracle on the same parameters."""
import numpy as np
import pytest

import helpers
import psfmc_oracle as orc
import synth_field

pytestmark = pytest.mark.gpu


def random_case(seed, shape=None):
    rng = np.random.RandomState(seed)
    ny, nx = rng.choice([64, 128, 256]), rng.choice([64, 128, 256])
    if shape is not None:
        ny, nx = shape
    n_psf = rng.choice([1, 1, 2, 3])
    py, px = rng.choice([9, 16, 21, 32, 33]), rng.choice([9, 16, 21, 32, 33])
    psfs, pivms = [], []
    for k in range(n_psf):
        yy, xx = np.mgrid[0:py, 0:px].astype(float)
        fw = 1.6 + 0.4 * k + rng.uniform(0, 0.5)
        core = (1 + ((xx - px // 2 - 0.2 * k) ** 2 + (yy - py // 2 + 0.1) ** 2) / fw ** 2) ** -2.5 * 300
        var = 0.01 + np.abs(core) / rng.uniform(20, 60)
        img = core + rng.normal(size=core.shape) * np.sqrt(var)
        iv = 1.0 / var
        if rng.rand() < 0.5:
            iv[rng.randint(py), rng.randint(px)] = 0.0
        psfs.append(img.astype(np.float32))
        pivms.append(iv.astype(np.float32))
    sci = (rng.normal(size=(ny, nx)) * 0.05).astype(np.float32)
    ivm = (1.0 / rng.uniform(0.02, 0.08, size=(ny, nx)) ** 2).astype(np.float32)
    for _ in range(rng.randint(0, 6)):
        ivm[rng.randint(ny), rng.randint(nx)] = rng.choice([0.0, -1.0, np.nan])
    for _ in range(rng.randint(0, 3)):
        sci[rng.randint(ny), rng.randint(nx)] = np.nan
    mask = None
    if rng.rand() < 0.5:
        mask = np.zeros((ny, nx), dtype=np.uint8)
        y0, x0 = rng.randint(ny - 8), rng.randint(nx - 8)
        mask[y0:y0 + rng.randint(1, 8), x0:x0 + rng.randint(1, 8)] = 1
    zp = rng.uniform(22, 27)
    comps = []
    if rng.rand() < 0.6:
        comps.append(dict(type='sky', adu=rng.normal() * 0.02))
    for _ in range(rng.randint(0, 3)):
        edge = rng.rand() < 0.3
        xy = (rng.uniform(-3, nx + 3), rng.uniform(-3, ny + 3)) if edge else \
            (rng.uniform(8, nx - 8), rng.uniform(8, ny - 8))
        if rng.rand() < 0.2:
            xy = (np.floor(xy[0]) + rng.choice([0.0, 0.5]), np.floor(xy[1]) + rng.choice([0.0, 0.5]))
        comps.append(dict(type='ps', xy=xy, mag=rng.uniform(16, 22), method=rng.choice(['lanczos3', 'bilinear'])))
    for _ in range(rng.randint(0, 4)):
        reff = rng.uniform(1.5, min(ny, nx) / 6)
        deg = bool(rng.rand() < 0.5)
        comps.append(dict(type='sersic', xy=(rng.uniform(10, nx - 10), rng.uniform(10, ny - 10)),
                          mag=rng.uniform(15, 23), reff=reff, reff_b=reff * rng.uniform(0.2, 1.0),
                          index=rng.choice([0.5, 1.0, 4.0, rng.uniform(0.3, 8.0)]),
                          angle=rng.uniform(0, 180) if deg else rng.uniform(-3.2, 3.2), angle_degrees=deg))
    if not comps:
        comps.append(dict(type='sky', adu=0.01))
    psf_index = rng.randint(n_psf)
    return dict(sci=sci, ivm=ivm, psfs=psfs, pivms=pivms, mask=mask, zp=zp, comps=comps, psf_index=psf_index)


def build(case, backend):
    from psfmc_amd import MultiComponentModel
    from psfmc_amd.ModelComponents import Configuration, Sky, PointSource, Sersic
    cfg = Configuration(case['sci'], case['ivm'], case['psfs'] if len(case['psfs']) > 1 else case['psfs'][0],
                        case['pivms'] if len(case['pivms']) > 1 else case['pivms'][0],
                        mask_file=case['mask'], mag_zeropoint=case['zp'])
    objs = [cfg]
    for c in case['comps']:
        if c['type'] == 'sky':
            objs.append(Sky(adu=c['adu']))
        elif c['type'] == 'ps':
            objs.append(PointSource(xy=c['xy'], mag=c['mag'], shift_method=c['method']))
        else:
            objs.append(Sersic(xy=c['xy'], mag=c['mag'], reff=c['reff'], reff_b=c['reff_b'], index=c['index'],
                               angle=c['angle'], angle_degrees=c['angle_degrees']))
    return MultiComponentModel(objs, backend=backend, max_walkers=4)


@pytest.mark.parametrize('seed', range(24))
def test_random_model_matches_oracle(seed):
    case = random_case(seed)
    field = orc.make_field(case['sci'], case['ivm'], case['psfs'], case['pivms'], mask=case['mask'],
                           mag_zp=case['zp'])
    want, imgs = orc.evaluate(field, case['comps'], case['psf_index'], raw_dtype=np.float64,
                              want_ps_sub=True)
    want = want if np.isfinite(want) else -np.inf
    n_free = 1 if len(case['psfs']) > 1 else 0           # only psf_index is free
    theta = np.full((2, n_free), float(case['psf_index']))
    for backend in ('fused', 'hipfft'):
        model = build(case, backend)
        got = model.log_likelihood_batch(theta)
        assert got[0] == got[1]
        if np.isfinite(want):
            assert abs(got[0] - want) <= 2e-10 * abs(want), (seed, backend, got[0], want)
        else:
            assert got[0] == -np.inf
        if seed % 4 == 0 and np.isfinite(want):
            dev = model.sample_images(theta[:1])
            for kind, ref in imgs.items():
                fin = np.isfinite(ref)
                assert np.array_equal(np.isfinite(dev[kind][0]), fin), (seed, backend, kind)
                scale = max(np.abs(ref[fin]).max(), 1e-300)
                # the weight map divides by (model variance + obs_var): the packed-FFT
                # variance channel is good to ~1e-11 of obs_var, not of its own tiny values
                tol = 1e-9 if kind == 'composite_ivm' else 1e-11
                assert np.abs(dev[kind][0][fin] - ref[fin]).max() <= tol * scale, (seed, backend, kind)
        model.close()


# every side the fused kernels are built for beyond the powers of two (psfmc_fft.h FftShape),
# each once as the row length and once as the column length, in rectangular pairs that also mix
# in power-of-two sides
GENERAL_SIDES = [96, 100, 120, 144, 150, 160, 180, 192, 200, 240, 250, 288, 300, 320, 360, 384, 400, 480,
                 500, 576, 600, 640, 720, 768, 800, 900, 960]


def general_shapes():
    from psfmc_amd import engine
    n = len(GENERAL_SIDES)
    shapes = [(GENERAL_SIDES[i], GENERAL_SIDES[(5 * i + 3) % n]) for i in range(n)]
    shapes += [(200, 200), (300, 300), (500, 500), (256, 200), (200, 256), (96, 512), (1024, 120), (160, 64), (150, 96),
               # a power-of-two nx whose unguarded row kernels do not divide ny: the guarded variant
               (150, 64), (100, 128), (150, 256), (250, 512), (500, 1024), (96, 1024)]
    # sides with a factor 7 (radix-7 codelet): every one of them once, paired with a side of another kind
    sevens = [84, 98, 112, 126, 140, 168, 196, 210, 224, 252, 280, 294, 336, 350, 392, 420, 448, 504, 560, 630,
              672, 700, 784, 840, 896]
    partners = [84, 256, 100, 126, 64, 150, 196, 128, 96, 252, 140, 120, 64, 350, 96, 210, 128, 84, 160, 98, 144,
                112, 168, 64, 224]
    shapes += list(zip(sevens, partners)) + [(140, 140), (64, 448), (200, 294)]
    # ... and once as the ROW length (round-2 advice: k_rows_fwd / k_rows_inv / k_raster_sums / k_pack_field of
    # these NX had only ever run for the few sides that happened to be a partner)
    shapes += [(p, s) for s, p in zip(sevens, partners) if (p, s) not in shapes]
    # sides with a factor 11 or 13 (the generic prime-radix codelet)
    primes = [88, 104, 110, 130, 132, 156, 176, 208, 220, 260, 264, 286, 308, 312, 330, 352, 364, 390, 416, 440, 484,
              520, 528, 572, 616, 624, 650, 660, 676, 704, 728, 780, 832]
    mates = [88, 64, 100, 130, 96, 128, 176, 84, 110, 64, 120, 104, 96, 156, 64, 88, 100, 130, 64, 132, 96,
             104, 64, 110, 88, 96, 64, 84, 100, 64, 104, 96, 64]
    shapes += list(zip(primes, mates)) + [(128, 286), (64, 676)]
    shapes += [(m, s) for s, m in zip(primes, mates) if (m, s) not in shapes]
    shapes += [(420, 420), (560, 560), (308, 308), (832, 832), (512, 100)]          # squares of the larger seven / prime sides
    # round 4: sides above 1024 (three-stage row and column kernels only), as the row and as the column length,
    # against two-stage, three-stage and power-of-two partners, and squares
    shapes += [(1152, 64), (96, 1152), (1280, 100), (128, 1280), (1536, 480), (250, 1536), (2048, 64), (84, 2048),
               (1152, 1152), (1280, 1536), (2048, 1152), (1536, 2048)]
    # round 4, second column survey: the seven sides that moved to the three-stage column kernel, against row lengths of
    # every layout (row groups of 8 -- where a side with 8 not dividing L falls back to the two-stage kernel --, 4, 2, 1)
    shapes += [(300, 128), (336, 128), (288, 64), (630, 128), (360, 1024), (280, 512), (350, 256), (300, 1152), (336, 2048)]
    assert all(engine.fused_supports(ny, nx) for ny, nx in shapes)
    # the claim above, enforced: every built side runs as the column length AND as the row length
    assert {ny for ny, _ in shapes} >= set(engine.FUSED_SIDES), sorted(set(engine.FUSED_SIDES) - {ny for ny, _ in shapes})
    assert {nx for _, nx in shapes} >= set(engine.FUSED_SIDES), sorted(set(engine.FUSED_SIDES) - {nx for _, nx in shapes})
    # a side with a prime factor > 13 (or factors the shapes cannot split into P, T <= 32) is not built ...
    assert not engine.fused_supports(170, 170) and not engine.fused_supports(256, 90) and not engine.fused_supports(490, 64)
    # ... but given the PSF's shape it is embedded in the next built side (test_embedded_sides_match_oracle)
    assert engine.fused_supports(170, 170, (33, 33)) and engine.fused_supports(490, 64, (16, 9))
    assert not engine.fused_supports(171, 170, (9, 9))
    # every even side up to 2048 - PSF side + 1 runs on the hand-written kernels
    assert all(engine.fused_supports(n, n, (64, 64)) for n in range(64, 1986, 2))
    assert not engine.fused_supports(1986, 1986, (64, 64)) and engine.fused_supports(2048, 2048, (64, 64))
    return shapes


@pytest.mark.parametrize('shape', general_shapes(), ids=lambda s: '%dx%d' % s)
def test_general_sides_match_oracle(shape):
    """Sides with factors 3, 5, 7, 11 and 13 (real cut-outs are rarely 2^k) on the fused kernels: likelihood
    and all five images against the fp64 oracle, and the two back ends against each other."""
    seed = 1000 + shape[0] * 7 + shape[1]
    case = random_case(seed, shape)
    field = orc.make_field(case['sci'], case['ivm'], case['psfs'], case['pivms'], mask=case['mask'],
                           mag_zp=case['zp'])
    want, imgs = orc.evaluate(field, case['comps'], case['psf_index'], raw_dtype=np.float64,
                              want_ps_sub=True)
    want = want if np.isfinite(want) else -np.inf
    n_free = 1 if len(case['psfs']) > 1 else 0
    theta = np.full((3, n_free), float(case['psf_index']))
    model = build(case, 'fused')
    assert model._backend == 'fused'
    got = model.log_likelihood_batch(theta)
    assert got[0] == got[1] == got[2]
    if np.isfinite(want):
        assert abs(got[0] - want) <= 2e-10 * abs(want), (shape, got[0], want)
        dev = model.sample_images(theta[:1])
        for kind, ref in imgs.items():
            fin = np.isfinite(ref)
            assert np.array_equal(np.isfinite(dev[kind][0]), fin), (shape, kind)
            scale = max(np.abs(ref[fin]).max(), 1e-300)
            # (the variance channel: see test_random_model_matches_oracle; 1.8e-9 observed at
            # 900 x 600.  Its rounding error is eps x the norm of raw^2 whatever transforms it --
            # 224 x 96 and 200 x 294 draw a 6e3-count peak and BOTH back ends, i.e. plain rfft2 too,
            # sit at 2.0e-8 from the oracle -- so the bound grows with the squared peak)
            peak = np.nanmax(np.abs(imgs['raw_model']))
            tol = 5e-9 * max(1.0, (peak / 2e3) ** 2) if kind == 'composite_ivm' else 1e-11
            assert np.abs(dev[kind][0][fin] - ref[fin]).max() <= tol * scale, (shape, kind)
            if kind == 'composite_ivm' and peak > 2e3 and np.finfo(np.longdouble).nmant >= 63:
                # the evidence for that bound (tests/test_oracle_precision.py): against the weight map
                # computed with 80-bit transforms the GPU is no farther off than a few times the fp64
                # oracle itself
                exact = helpers.longdouble_weight_map(field, imgs['raw_model'], case['psf_index'])
                e_orc = float(np.abs(ref[fin].astype(np.longdouble) - exact[fin]).max())
                e_gpu = float(np.abs(dev[kind][0][fin].astype(np.longdouble) - exact[fin]).max())
                assert e_gpu <= 4.0 * e_orc + 1e-11 * scale, (shape, e_gpu / scale, e_orc / scale)
        if max(shape) > 1024:
            # sides above 1024: the posterior-image sums too (k_raster_sums with one row per wave, the forward row
            # kernel's from-image form and the inverse kernel's image outputs of the three-stage family)
            model.accumulate_samples(theta[:2])
            post = model.collect_posterior_images()
            for kind, ref in imgs.items():
                fin = np.isfinite(ref)
                scale = max(np.abs(ref[fin]).max(), 1e-300)
                peak = np.nanmax(np.abs(imgs['raw_model']))
                tol = 5e-9 * max(1.0, (peak / 2e3) ** 2) if kind == 'composite_ivm' else 1e-11
                assert np.abs(post[kind][fin] - ref[fin]).max() <= tol * scale, (shape, kind, 'posterior')
    else:
        assert got[0] == -np.inf
    # the device-computed PSF spectra of this shape against numpy
    psf_spec, var_spec = model.engine.spectra()
    for k in range(len(case['psfs'])):
        assert np.abs(psf_spec[k] - field.psf_spec[k]).max() <= 1e-13 * np.abs(field.psf_spec[k]).max()
        assert np.abs(var_spec[k] - field.var_spec[k]).max() <= 1e-13 * np.abs(field.var_spec[k]).max()
    model.close()


@pytest.mark.parametrize('n_side', [96, 100, 120, 150, 180, 200, 250, 300, 384, 500, 640, 900])
def test_general_sides_with_distinct_walkers(n_side):
    """A batch of DISTINCT walkers (prior draws and near-truth) on square general-side fields:
    the fused kernels against the hipFFT back end (independent arithmetic), the oracle on two
    walkers, and bitwise independence of batch order and composition -- the column kernel's
    kx-major work order, idle lanes and spare slots must not leak between walkers."""
    from test_gpu_fullsize import make_model
    n_sersic = 2 if n_side <= 300 else 1
    n_w = 24 if n_side <= 500 else 10
    model, fld = make_model(n_side, n_sersic, 'fused', max_walkers=n_w)
    ref, _ = make_model(n_side, n_sersic, 'hipfft', max_walkers=n_w)
    theta = np.vstack([synth_field.draw_walkers(n_side, n_sersic, n_w // 2, seed=n_side),
                       synth_field.draw_walkers(n_side, n_sersic, n_w - n_w // 2, seed=n_side + 1,
                                                near_truth=fld['truth'])])
    got = model.log_posterior_batch(theta)
    assert np.isfinite(got).all()
    assert helpers.rel_err(got, ref.log_posterior_batch(theta)) <= 1e-11
    field = orc.make_field(fld['sci'], fld['ivm'], [fld['psf']], [fld['psf_ivm']], mag_zp=fld['mag_zp'])
    prior = model.log_priors_batch(theta[[0, n_w - 1]])
    for i, p in zip((0, n_w - 1), prior):
        want = helpers.oracle_loglike(field, helpers.synth_layout(n_sersic), theta[i]) + p
        assert abs(got[i] - want) <= 1e-10 * abs(want), (n_side, i)
    perm = np.random.RandomState(n_side).permutation(n_w)
    assert np.array_equal(model.log_posterior_batch(theta[perm]), got[perm])
    assert np.array_equal(model.log_posterior_batch(theta[2:5]), got[2:5])
    assert np.array_equal(model.log_posterior_batch(theta[-1:]), got[-1:])
    model.close()
    ref.close()


# even sides the transforms are NOT built for (prime factors above 13, or no P x T split with P, T <= 32):
# embedded in the next built side >= side + PSF side - 1 (psfmc_device.h WrapDesc), each axis on its own
EMBEDDED_SHAPES = [(170, 170), (256, 90), (490, 64), (64, 490), (74, 74), (134, 256), (200, 134), (166, 226),
                   (238, 340), (68, 1000), (958, 70), (290, 292), (990, 82), (94, 102), (502, 514), (686, 98),
                   (642, 70), (70, 642),        # (642: the smallest sides that fit -- 650, 660, 676 -- have slow kernels)
                   (1000, 1000), (1100, 1024), (66, 1030), (1300, 70), (1984, 64)]   # round 4: embedded above 1024


@pytest.mark.parametrize('shape', EMBEDDED_SHAPES, ids=lambda s: '%dx%d' % s)
def test_embedded_sides_match_oracle(shape):
    """Image sides outside the built list on the fused kernels (round-2 review: `backend='auto'` left them
    to hipFFT): the log-likelihood and all five images against the fp64 oracle at the SAME tolerances as
    the built sides -- the circular convolution of the image's own size, not of a padded one
    (psfMC/utils.py:25-32) -- and against the hipFFT back end, which transforms at the image's size."""
    from psfmc_amd import engine
    seed = 3000 + shape[0] * 7 + shape[1]
    case = random_case(seed, shape)
    psf_shape = case['psfs'][0].shape
    assert engine.fused_supports(shape[0], shape[1], psf_shape) and not engine.fused_supports(*shape)
    field = orc.make_field(case['sci'], case['ivm'], case['psfs'], case['pivms'], mask=case['mask'],
                           mag_zp=case['zp'])
    want, imgs = orc.evaluate(field, case['comps'], case['psf_index'], raw_dtype=np.float64, want_ps_sub=True)
    want = want if np.isfinite(want) else -np.inf
    n_free = 1 if len(case['psfs']) > 1 else 0
    theta = np.full((3, n_free), float(case['psf_index']))
    model = build(case, 'auto')
    assert model._backend == 'fused'
    got = model.log_likelihood_batch(theta)
    assert got[0] == got[1] == got[2]
    # the transform shape: built sides with room for the image and the wrap-around margin -- the smallest
    # such side or a larger one whose kernels are cheaper (psfmc_hip.hip choose_embedding)
    for axis, key in ((0, 'transform_ny'), (1, 'transform_nx')):
        side = int(model.engine.get_option(key))
        assert side in engine.FUSED_SIDES
        assert side == shape[axis] or side >= engine.embedding_side(shape[axis], psf_shape[axis])
    ref = build(case, 'hipfft')
    if np.isfinite(want):
        assert abs(got[0] - want) <= 2e-10 * abs(want), (shape, got[0], want)
        assert abs(got[0] - ref.log_likelihood_batch(theta)[0]) <= 2e-10 * abs(want)
        dev = model.sample_images(theta[:1])
        for kind, img in imgs.items():
            assert dev[kind][0].shape == tuple(shape)
            fin = np.isfinite(img)
            assert np.array_equal(np.isfinite(dev[kind][0]), fin), (shape, kind)
            scale = max(np.abs(img[fin]).max(), 1e-300)
            peak = np.nanmax(np.abs(imgs['raw_model']))
            tol = 5e-9 * max(1.0, (peak / 2e3) ** 2) if kind == 'composite_ivm' else 1e-11
            assert np.abs(dev[kind][0][fin] - img[fin]).max() <= tol * scale, (shape, kind)
        # posterior-image sums (the linear-sum route rasterises with the wrapped coordinates too)
        model.accumulate_samples(theta[:2])
        post = model.collect_posterior_images()
        for kind, img in imgs.items():
            fin = np.isfinite(img)
            scale = max(np.abs(img[fin]).max(), 1e-300)
            peak = np.nanmax(np.abs(imgs['raw_model']))
            tol = 5e-9 * max(1.0, (peak / 2e3) ** 2) if kind == 'composite_ivm' else 1e-11
            assert np.abs(post[kind][fin] - img[fin]).max() <= tol * scale, (shape, kind, 'posterior')
    else:
        assert got[0] == -np.inf
    model.close()
    ref.close()


def test_embedded_side_with_distinct_walkers_and_priors():
    """170 x 170 (2 x 5 x 17) through the raw-vector path: priors, early-out, a batch of distinct walkers,
    the device sampler -- against the hipFFT back end and the oracle."""
    from test_gpu_fullsize import make_model
    n_side, n_sersic, n_w = 170, 1, 24
    model, fld = make_model(n_side, n_sersic, 'auto', max_walkers=n_w)
    assert model._backend == 'fused'
    ref, _ = make_model(n_side, n_sersic, 'hipfft', max_walkers=n_w)
    theta = np.vstack([synth_field.draw_walkers(n_side, n_sersic, n_w // 2, seed=n_side),
                       synth_field.draw_walkers(n_side, n_sersic, n_w - n_w // 2, seed=n_side + 1,
                                                near_truth=fld['truth'])])
    got = model.log_posterior_batch(theta)
    assert np.isfinite(got).all()
    assert helpers.rel_err(got, ref.log_posterior_batch(theta)) <= 1e-11
    field = orc.make_field(fld['sci'], fld['ivm'], [fld['psf']], [fld['psf_ivm']], mag_zp=fld['mag_zp'])
    prior = model.log_priors_batch(theta[[0, n_w - 1]])
    for i, p in zip((0, n_w - 1), prior):
        want = helpers.oracle_loglike(field, helpers.synth_layout(n_sersic), theta[i]) + p
        assert abs(got[i] - want) <= 1e-10 * abs(want), i
    perm = np.random.RandomState(n_side).permutation(n_w)
    assert np.array_equal(model.log_posterior_batch(theta[perm]), got[perm])
    from psfmc_amd import DeviceEnsembleSampler
    samp = DeviceEnsembleSampler(n_w, model, block=3)
    samp.random_state = np.random.RandomState(5).get_state()
    for res in samp.sample(theta, iterations=4):
        pass
    assert np.isfinite(res[1]).all() and samp.naccepted.sum() > 0
    assert np.array_equal(model.log_posterior_batch(res[0]), res[1])
    with pytest.raises(Exception):
        model.engine.spectra()
    model.close()
    ref.close()
