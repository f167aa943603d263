"""
Generate the committed golden vectors under tests/golden/ from the REAL
reference (mmechtley/psfMC at /root/reference), and pin oracle/psfmc_oracle.py
against it in the same process.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_golden.py

Interpreter: conda python3.9 (numpy 1.26 / scipy 1.7 / astropy 4.3 / numexpr
2.7) -- the only one here that can import the reference's hot-path modules.
Harness-only shims (reference files untouched):
  * `np.asscalar` re-added (removed in numpy 1.23; psfMC/distributions.py:136)
  * a bare `psfMC` package object so psfMC/__init__.py (emcee, corner) is skipped

Outputs (.npz, data only): parameter matrices, reference log-posteriors and
log-priors, a few per-stage images, and the input arrays of every field.
"""
from __future__ import division, print_function

import gzip
import os
import shutil
import sys
import tempfile
import types
import warnings

import numpy as np

warnings.filterwarnings('ignore')
np.asscalar = lambda a: np.asarray(a).item()
_pkg = types.ModuleType('psfMC')
_pkg.__path__ = ['/root/reference/psfMC']
sys.modules['psfMC'] = _pkg

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

from astropy.io import fits                                   # noqa: E402
from psfMC.models import MultiComponentModel                  # noqa: E402
from psfMC.ModelComponents import Sky, PointSource, Sersic    # noqa: E402
import psfmc_oracle as orc                                    # noqa: E402
import synth_field                                            # noqa: E402

IMG_KEYS = ('raw_model', 'convolved_model', 'residual', 'composite_ivm',
            'point_source_subtracted')


def comps_from_reference(model):
    """Read the reference components' *current* values into oracle dicts."""
    comps, psf_index = [], 0
    for c in model.components:
        if isinstance(c, Sky):
            comps.append(dict(type='sky', adu=float(c.adu)))
        elif isinstance(c, PointSource):
            comps.append(dict(type='ps', xy=np.array(c.xy, dtype=float),
                              mag=float(c.mag), method=c.shift_method))
        elif isinstance(c, Sersic):
            comps.append(dict(type='sersic', xy=np.array(c.xy, dtype=float),
                              mag=float(c.mag), reff=float(c.reff),
                              reff_b=float(c.reff_b), index=float(c.index),
                              angle=float(c.angle),
                              angle_degrees=bool(c.angle_degrees)))
        else:                      # PSFSelector
            psf_index = c.psf_index
    return comps, psf_index


def oracle_field_from_reference(model, sci, ivm, psfs, psf_ivms, mask):
    fld = orc.make_field(sci, ivm, psfs, psf_ivms, mask=mask,
                         mag_zp=model.config.mag_zeropoint)
    cfg = model.config
    assert np.array_equal(fld.bad_px, cfg.bad_px)
    assert np.array_equal(fld.obs_var, cfg.obs_var)
    for a, b in zip(fld.psf_spec, cfg.psf_selector.psf_list):
        assert np.allclose(a, b, rtol=0, atol=1e-15), np.abs(a - b).max()
    for a, b in zip(fld.var_spec, cfg.psf_selector.var_list):
        assert np.allclose(a, b, rtol=1e-13, atol=1e-22), np.abs(a - b).max()
    return fld


def field_check(arrays):
    """What a `light` fixture keeps of its input arrays (the tests regenerate them from the seed and compare):
    float64 sums and a few pixels of every array."""
    vals = []
    for a in [arrays['sci'], arrays['ivm']] + list(arrays['psfs']) + list(arrays['psf_ivms']):
        a = np.asarray(a, dtype=np.float64)
        vals += [a.sum(), np.abs(a).sum(), (a * a).sum(), a[0, 0], a[a.shape[0] // 2, a.shape[1] // 3], a[-1, -1]]
    return np.array(vals)


def run_case(name, model_path, vectors, arrays, full_image_rows=(0,),
             mask=None, light=False):
    """Evaluate every vector with the reference; check the oracle; save.
    light: keep only the vectors and the reference's scalars (the field is regenerated from its seed by
    tools/synth_field.py on the test side and compared through `field_check`)."""
    model = MultiComponentModel(model_path)
    names = sum([c.stochastic_names() for c in model.components], [])
    fld = oracle_field_from_reference(model, arrays['sci'], arrays['ivm'],
                                      arrays['psfs'], arrays['psf_ivms'], mask)
    n = len(vectors)
    lnprob = np.empty(n)
    lnprior = np.empty(n)
    loglike = np.empty(n)
    sums = np.full((n, 5), np.nan)
    derived = []
    worst = 0.0
    worst64 = 0.0
    images = {}
    for i, vec in enumerate(vectors):
        lp, blobs = MultiComponentModel.log_posterior(np.array(vec), model=model)
        model.param_values = np.array(vec)
        lnprior[i] = model.log_priors()
        lnprob[i] = lp
        comps, psf_index = comps_from_reference(model)
        if not np.isfinite(lnprior[i]):
            loglike[i] = np.nan
            derived.append(None)
            assert lp == -np.inf and blobs == {}
            continue
        derived.append(orc.derived_row(fld, comps, psf_index))
        ll32, img32 = orc.evaluate(fld, comps, psf_index, raw_dtype=None,
                                   want_ps_sub=True)
        ll64, _ = orc.evaluate(fld, comps, psf_index, raw_dtype=np.float64)
        loglike[i] = ll64 if np.isfinite(ll64) else -np.inf
        o32 = (ll32 + lnprior[i]) if np.isfinite(ll32) else -np.inf
        o64 = (ll64 + lnprior[i]) if np.isfinite(ll64) else -np.inf
        if np.isfinite(lp):
            worst = max(worst, abs(o32 - lp) / abs(lp))
            worst64 = max(worst64, abs(o64 - lp) / abs(lp))
            for k in IMG_KEYS:
                ref = np.asarray(blobs[k], dtype=np.float64)
                got = np.asarray(img32[k], dtype=np.float64)
                assert np.array_equal(np.isnan(ref), np.isnan(got)), (name, i, k)
                fin = np.isfinite(ref)
                assert np.array_equal(ref[~fin & ~np.isnan(ref)],
                                      got[~fin & ~np.isnan(ref)]), (name, i, k)
                scale = np.abs(ref[fin]).max()
                assert np.abs(got[fin] - ref[fin]).max() <= 1e-11 * scale, \
                    (name, i, k)
        else:
            assert o32 == -np.inf, (name, i, o32, lp)
        if blobs:
            sums[i] = [np.nansum(np.asarray(blobs[k], dtype=np.float64))
                       for k in IMG_KEYS]
        if i in full_image_rows:
            for k in IMG_KEYS:
                images['img%d_%s' % (i, k)] = np.asarray(blobs[k])
    assert worst <= 1e-11, (name, worst)
    print('%-10s %3d vectors  finite=%3d  oracle(f32 raw) vs ref: %.2e   '
          'oracle(f64) vs ref: %.2e' % (name, n, np.isfinite(lnprob).sum(),
                                         worst, worst64))
    width = max(len(d) for d in derived if d is not None)
    dmat = np.full((n, width), np.nan)
    for i, d in enumerate(derived):
        if d is not None:
            dmat[i] = d
    out = dict(params=np.asarray(vectors, dtype=np.float64), lnprob=lnprob,
               lnprior=lnprior, loglike_f64=loglike, image_sums=sums,
               derived=dmat, param_names=np.array(names),
               mag_zp=np.float64(model.config.mag_zeropoint),
               oracle_vs_ref_rel=np.float64(worst),
               oracle64_vs_ref_rel=np.float64(worst64))
    if light:
        out['field_check'] = field_check(arrays)
    else:
        out.update(sci=arrays['sci'], ivm=arrays['ivm'], psfs=np.asarray(arrays['psfs']),
                   psf_ivms=np.asarray(arrays['psf_ivms']))
    if mask is not None:
        out['mask'] = mask
    out.update(images)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    return model


def prior_draws(model, n, seed):
    np.random.seed(seed)
    return model.init_params_from_priors(n)


def native(a):
    return np.ascontiguousarray(a, dtype=a.dtype.newbyteorder('='))


# --------------------------------------------------------------------------
def case_example():
    ex = os.path.join(HERE, 'example')
    arrays = dict(
        sci=native(fits.getdata(os.path.join(ex, 'sci_J0005-0006.fits'))),
        ivm=native(fits.getdata(os.path.join(ex, 'ivm_J0005-0006.fits'))),
        psfs=[native(fits.getdata(os.path.join(ex, 'sci_psf.fits')))],
        psf_ivms=[native(fits.getdata(os.path.join(ex, 'ivm_psf.fits')))])
    path = os.path.join(ex, 'model_example.py')
    model = MultiComponentModel(path)
    median = np.concatenate([np.ravel(c.set_stochastic_values('median'))
                             for c in model.components])
    hand = np.array([0.001, 20.9, 64.3, 64.8, 35, 2.5, 22.5, 6, 4, 65.1, 63.7,
                     100, 1.2, 24.5, 4, 3, 46.2, 85.1])
    bad_axis = hand.copy()
    bad_axis[8] = 7.0                          # reff_b > reff -> -inf
    on_pixel = hand.copy()
    on_pixel[9:11] = (65.0, 64.0)              # Sersic centre on a pixel -> NaN
    draws = prior_draws(model, 40, seed=11)
    rng = np.random.RandomState(5)
    near = hand + rng.normal(size=(20, hand.size)) * 0.02
    vectors = np.vstack([median, hand, bad_axis, on_pixel, draws, near])
    # the survey's known answers must reproduce from this file
    lp, _ = MultiComponentModel.log_posterior(median.copy(), model=model)
    assert abs(lp - (-65129.74748302071)) < 1e-6, lp
    lp, _ = MultiComponentModel.log_posterior(hand.copy(), model=model)
    assert abs(lp - (-68376.94489354931)) < 1e-6, lp
    run_case('example', path, vectors, arrays, full_image_rows=(1,))


def _write_fits(path, arr):
    fits.PrimaryHDU(np.asarray(arr)).writeto(path, overwrite=True)


def case_synth(name, n_side, n_sersic, n_prior, n_near, tmp, light=False):
    fld = synth_field.make_field(n_side, n_sersic, seed=0)
    d = os.path.join(tmp, name)
    os.makedirs(d)
    _write_fits(os.path.join(d, 'sci.fits'), fld['sci'])
    _write_fits(os.path.join(d, 'ivm.fits'), fld['ivm'])
    _write_fits(os.path.join(d, 'psf.fits'), fld['psf'])
    _write_fits(os.path.join(d, 'psf_ivm.fits'), fld['psf_ivm'])
    path = os.path.join(d, 'model.py')
    with open(path, 'w') as f:
        f.write(synth_field.model_file_text(n_side, n_sersic))
    vec = np.vstack([
        fld['truth'][None, :],
        synth_field.draw_walkers(n_side, n_sersic, n_prior, seed=1),
        synth_field.draw_walkers(n_side, n_sersic, n_near, seed=2,
                                 near_truth=fld['truth'])])
    if light:
        # (the large configurations: BASELINE configs 3 and 4) one vector with reff_b > reff of the first Sersic
        # component (Sersic.py:41-45: -inf) and one with the first component's centre on a pixel (0/0 -> NaN -> -inf)
        extra = np.repeat(fld['truth'][None, :], 2, axis=0)
        extra[0, 7] = extra[0, 6] + 0.5                  # order per Sersic: angle, index, mag, reff, reff_b, x, y
        extra[1, 8:10] = np.rint(extra[1, 8:10])
        vec = np.vstack([vec, extra])
    arrays = dict(sci=fld['sci'], ivm=fld['ivm'], psfs=[fld['psf']],
                  psf_ivms=[fld['psf_ivm']])
    run_case(name, path, vec, arrays, full_image_rows=(), light=light)


def case_edge(tmp):
    """64 x 128 field, two 32 x 32 PSFs (psf_index free), FITS mask, bad
    pixels, bilinear + lanczos3 point sources, border clipping."""
    rng = np.random.RandomState(77)
    ny, nx = 64, 128
    base = synth_field.moffat_psf(32, fwhm=2.2, beta=2.5) * 500
    psfs, ivms = [], []
    for k in range(2):
        wid = synth_field.moffat_psf(32, fwhm=2.2 + 0.3 * k, beta=2.5) * 500
        var = 0.02 ** 2 + np.abs(wid) / 40.0
        p = wid + rng.normal(size=wid.shape) * np.sqrt(var)
        iv = 1.0 / var
        if k == 1:
            p[3, 4] = np.nan                   # bad PSF pixel (utils.py:113-116)
            iv[10, 20] = 0.0
        psfs.append(p.astype(np.float32))
        ivms.append(iv.astype(np.float32))
    del base
    sci = (rng.normal(size=(ny, nx)) * 0.05).astype(np.float32)
    yy, xx = np.mgrid[0:ny, 0:nx]
    sci += (30 * np.exp(-((xx - 70.3) ** 2 + (yy - 30.8) ** 2) / 18.0)
            ).astype(np.float32)
    ivm = np.full((ny, nx), 400.0, dtype=np.float32)
    ivm[5, 7] = 0.0                            # zero weight
    ivm[40, 100] = -1.0                        # negative weight
    sci[20, 50] = np.nan                       # non-finite data
    mask = np.zeros((ny, nx), dtype=np.int16)
    mask[50:60, 10:30] = 1
    d = os.path.join(tmp, 'edge')
    os.makedirs(d)
    _write_fits(os.path.join(d, 'sci.fits'), sci)
    _write_fits(os.path.join(d, 'ivm.fits'), ivm)
    for k in range(2):
        _write_fits(os.path.join(d, 'psf%d.fits' % k), psfs[k])
        _write_fits(os.path.join(d, 'psfivm%d.fits' % k), ivms[k])
    _write_fits(os.path.join(d, 'mask.fits'), mask)
    text = '\n'.join([
        'from numpy import array',
        "Configuration(obs_file='sci.fits', obsivm_file='ivm.fits',",
        "              psf_files=['psf0.fits', 'psf1.fits'],",
        "              psfivm_files=['psfivm0.fits', 'psfivm1.fits'],",
        "              mask_file='mask.fits', mag_zeropoint=24.0)",
        'Sky(adu=Normal(loc=0, scale=0.05))',
        "PointSource(xy=Uniform(loc=array((-2.0, -2.0)), scale=array((132.0, 68.0))),",
        "            mag=Uniform(loc=17.0, scale=5.0), shift_method='bilinear')",
        'PointSource(xy=Uniform(loc=array((-2.0, -2.0)), scale=array((132.0, 68.0))),',
        '            mag=Uniform(loc=17.0, scale=5.0))',
        'Sersic(xy=Uniform(loc=array((40.0, 10.0)), scale=array((50.0, 40.0))),',
        '       mag=Uniform(loc=16.0, scale=6.0), reff=Uniform(loc=1.0, scale=20.0),',
        '       reff_b=Uniform(loc=1.0, scale=20.0), index=Uniform(loc=0.4, scale=7.6),',
        '       angle=Uniform(loc=-4, scale=8))',
        ''])
    path = os.path.join(d, 'model.py')
    with open(path, 'w') as f:
        f.write(text)
    # order: sky adu | PS1 mag,x,y | PS2 mag,x,y | Sersic angle,index,mag,reff,
    # reff_b,x,y | psf_index
    base = np.array([0.01, 19.0, 30.2, 20.7, 18.5, 90.4, 40.6,
                     0.7, 2.2, 18.0, 8.0, 5.0, 70.3, 30.8, 0.0])

    def v(**kw):
        idx = dict(sky=0, m1=1, x1=2, y1=3, m2=4, x2=5, y2=6, ang=7, n=8, ms=9,
                   re=10, rb=11, xs=12, ys=13, psf=14)
        out = base.copy()
        for k, val in kw.items():
            out[idx[k]] = val
        return out
    vectors = [
        base,
        v(psf=1.0), v(psf=0.4), v(psf=0.6), v(psf=1.49),
        v(psf=1.7),                            # rint -> 2: outside support
        v(sky=-0.03),
        v(x1=0.2, y1=0.3), v(x1=-0.4, y1=-1.2),       # bilinear at/over the edge
        v(x1=127.3, y1=63.4), v(x1=128.9, y1=64.7),
        v(x2=1.3, y2=2.1), v(x2=-1.5, y2=-1.0),       # lanczos3 window clipped
        v(x2=126.2, y2=62.7), v(x2=129.5, y2=65.5),
        v(x2=64.5, y2=31.5), v(x2=64.0, y2=31.0),     # half-pixel / on-pixel
        v(x2=2.5, y2=2.5), v(x2=3.5, y2=60.5),        # round-half-even cases
        v(n=0.5), v(n=1.0), v(n=4.0), v(n=7.9), v(n=0.41),
        v(re=8.0, rb=8.0), v(re=5.0, rb=5.000001),    # reff_b > reff -> -inf
        v(xs=70.0, ys=31.0),                          # centre on a pixel: NaN
        v(xs=70.0, ys=30.999999),
        v(ang=-3.9), v(ang=3.9), v(re=20.9, rb=1.05),
        v(ms=16.1), v(ms=21.9),
    ]
    rng2 = np.random.RandomState(3)
    model = MultiComponentModel(path)
    vectors = np.vstack([np.array(vectors), prior_draws(model, 24, seed=21)])
    arrays = dict(sci=sci, ivm=ivm, psfs=psfs, psf_ivms=ivms)
    del rng2
    run_case('edge', path, vectors, arrays, full_image_rows=(0, 1),
             mask=mask)
    with open(os.path.join(HERE, 'edge_model.py'), 'w') as f:
        f.write(text)


def case_galfit():
    """The reference's own Sersic fixtures (tests/gfsim_n*.fits.gz, GALFIT
    renderings) + what the reference renders for the same parameters
    (tests/test_components.py:62-75)."""
    out = {}
    coords = orc.array_coords((128, 128))
    for n in ('0.5', '1.0', '3.1', '4.0', '6.5'):
        path = '/root/reference/tests/gfsim_n%s.fits.gz' % n
        with gzip.open(path) as f:
            hdul = fits.open(f)
            img = np.array(hdul[0].data, dtype=np.float32)
            hdr = hdul[0].header
            pars = {}
            for key in ('1_XC', '1_YC', '1_MAG', '1_RE', '1_N', '1_AR', '1_PA'):
                pars[key] = float(str(hdr[key]).split('+/-')[0])
            pars['MAGZPT'] = float(hdr['MAGZPT'])
        ser = Sersic(xy=(pars['1_XC'] - 1, pars['1_YC'] - 1), mag=pars['1_MAG'],
                     index=pars['1_N'], reff=pars['1_RE'],
                     reff_b=pars['1_RE'] * pars['1_AR'], angle=pars['1_PA'],
                     angle_degrees=True)
        ref = np.zeros((128, 128))
        ser.add_to_array(ref, mag_zp=pars['MAGZPT'], coords=coords)
        mine = np.zeros((128, 128))
        orc.add_sersic(mine, (pars['1_XC'] - 1, pars['1_YC'] - 1),
                       pars['1_MAG'], pars['1_RE'], pars['1_RE'] * pars['1_AR'],
                       pars['1_N'], pars['1_PA'], True, pars['MAGZPT'], coords)
        assert np.abs(mine - ref).max() <= 1e-12 * np.abs(ref).max()
        tag = n.replace('.', 'p')
        out['galfit_' + tag] = img
        out['psfmc_' + tag] = ref
        out['pars_' + tag] = np.array([pars[k] for k in
                                       ('1_XC', '1_YC', '1_MAG', '1_RE', '1_N',
                                        '1_AR', '1_PA', 'MAGZPT')])
    np.savez_compressed(os.path.join(HERE, 'galfit.npz'), **out)
    print('galfit     5 fixtures converted')


def main():
    # `make_golden.py synth512x2 synth1024x4` regenerates only the named light fixtures
    only = set(sys.argv[1:]) or None
    tmp = tempfile.mkdtemp(prefix='psfmc_golden_')
    try:
        if only is not None:
            if 'synth512x2' in only:
                case_synth('synth512x2', 512, 2, n_prior=14, n_near=11, tmp=tmp, light=True)
            if 'synth1024x4' in only:
                case_synth('synth1024x4', 1024, 4, n_prior=4, n_near=5, tmp=tmp, light=True)
            return
        case_example()
        case_synth('synth256', 256, 1, n_prior=40, n_near=24, tmp=tmp)
        case_synth('synth128x2', 128, 2, n_prior=24, n_near=8, tmp=tmp)
        if only is None or 'synth512x2' in only:
            case_synth('synth512x2', 512, 2, n_prior=14, n_near=11, tmp=tmp, light=True)
        if only is None or 'synth1024x4' in only:
            case_synth('synth1024x4', 1024, 4, n_prior=4, n_near=5, tmp=tmp, light=True)
        case_edge(tmp)
        case_galfit()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
