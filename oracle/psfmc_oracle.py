"""
CPU oracle for the psfMC per-sample log-posterior -- TEST INFRASTRUCTURE ONLY.

This module is a plain numpy/scipy restatement of the reference algorithm for the
hot path named in BASELINE.json (`MultiComponentModel.log_posterior`,
/root/reference/psfMC/models.py:193-243).  It is the *checker*: only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it.
The shipped package (`psfmc_amd/`) never imports it and has no CPU fallback.

Parity status: PINNED.  `tests/golden/make_golden.py` imports the real reference
(conda python3.9) next to this file, in one process, and
asserts agreement <=1e-11 relative on log-posteriors and per-stage images before
it writes the committed `tests/golden/*.npz` vectors; `tests/test_oracle_golden.py`
re-checks the oracle against those vectors on every run.

Every function cites the reference file:line it follows.  One walker per call,
like the reference (emcee maps `lnpostfn` over walkers one at a time,
psfMC/fitting.py:55-58).  Only numpy + scipy.special are used (the same
third-party arithmetic the reference itself calls: numpy pocketfft,
scipy.special.gammaincinv / gamma).

Dtype note (SURVEY.md section 8(a) note D): the reference accumulates the raw model in
the dtype of `obs_var` (float32 for float32 FITS input, models.py:249) and
squares it in that dtype (models.py:277).  `raw_dtype=None` reproduces that;
`raw_dtype=np.float64` is the all-fp64 pipeline the HIP kernels implement (the
two differ by <=6e-8 relative in the log-posterior).
"""
from __future__ import division

from math import fsum

import numpy as np
from scipy.special import gamma as _gamma
from scipy.special import gammaincinv as _gammaincinv

__all__ = ['Field', 'make_field', 'array_coords', 'mag_to_flux', 'sersic_kappa',
           'sersic_sb_eff', 'add_sky', 'add_point_source', 'add_sersic',
           'raw_model', 'convolve', 'evaluate', 'log_likelihood',
           'derived_row', 'DERIVED_SKY', 'DERIVED_PS', 'DERIVED_SERSIC']


# --------------------------------------------------------------------------
# setup half: psfMC/utils.py:9-22, 45-79, 106-157 ; Configuration.py:41-52 ;
# PSFSelector.py:32-43
# --------------------------------------------------------------------------
def _pad_and_rfft(img, newshape):
    """utils.py:9-22 -- centre-pad at offset pad//2, then rfft2 (fp64)."""
    pad = np.asarray(newshape) - np.asarray(img.shape)
    if np.any(pad < 0):
        raise NotImplementedError('PSF larger than observation')
    canvas = np.zeros(newshape, dtype=np.float64)
    canvas[pad[0] // 2:pad[0] // 2 + img.shape[0],
           pad[1] // 2:pad[1] // 2 + img.shape[1]] = img
    return np.fft.rfft2(canvas)


def preprocess_psf(psf_data, psf_ivm):
    """utils.py:106-123 (+ norm_psf :45-51).  Works in the input dtype, like
    the reference does on the arrays astropy hands it."""
    psf_data = np.array(psf_data, dtype=np.asarray(psf_data).dtype.newbyteorder('='))
    psf_ivm = np.array(psf_ivm, dtype=np.asarray(psf_ivm).dtype.newbyteorder('='))
    bad = ~np.isfinite(psf_data) | ~np.isfinite(psf_ivm) | (psf_ivm <= 0)
    psf_data[bad] = 0
    psf_ivm[bad] = 0
    total = fsum(psf_data.flat)
    psf_data = psf_data / total
    psf_ivm = psf_ivm * total ** 2
    with np.errstate(divide='ignore'):
        psf_var = np.where(psf_ivm <= 0, 0, 1 / psf_ivm)
    return psf_data, psf_var


class Field(object):
    """Shared per-field arrays (SURVEY.md row Cfg): what Configuration +
    PSFSelector hold after setup."""

    def __init__(self, sci, obs_var, bad_px, mag_zp, psf_list, var_list,
                 psf_spec, var_spec):
        self.sci = sci
        self.obs_var = obs_var
        self.bad_px = bad_px
        self.mag_zp = mag_zp
        self.psf_list = psf_list      # normalised real-space PSFs
        self.var_list = var_list      # real-space PSF variance maps
        self.psf_spec = psf_spec      # rfft2 of centre-padded PSFs
        self.var_spec = var_spec
        self.shape = sci.shape
        self.coords = array_coords(sci.shape)


def make_field(sci, ivm, psfs, psf_ivms, mask=None, mag_zp=0.0):
    """preprocess_obs (utils.py:54-79) + PSFSelector.__init__ (PSFSelector.py:32-43).

    `sci`/`ivm` 2-D arrays as read from FITS (dtype preserved, like the
    reference); `psfs`/`psf_ivms` lists of 2-D arrays; `mask` optional array,
    nonzero = exclude (utils.py:87-90)."""
    sci = np.asarray(sci)
    ivm = np.asarray(ivm)
    sci = sci.astype(sci.dtype.newbyteorder('='))
    ivm = ivm.astype(ivm.dtype.newbyteorder('='))
    bad = ~np.isfinite(sci) | ~np.isfinite(ivm) | (ivm <= 0)
    with np.errstate(divide='ignore', invalid='ignore'):
        obs_var = np.where(bad, np.inf, 1 / ivm).astype(ivm.dtype)
    if mask is not None:
        bad = bad | np.asarray(mask).astype(bool)
    pairs = [preprocess_psf(p, v) for p, v in zip(psfs, psf_ivms)]
    psf_list = [p for p, _ in pairs]
    var_list = [v for _, v in pairs]
    if len(psf_list) > 1:                      # utils.py:136-157
        mismatch = np.var(psf_list, axis=0)
        var_list = [v + mismatch for v in var_list]
    psf_spec = [_pad_and_rfft(p, sci.shape) for p in psf_list]
    var_spec = [_pad_and_rfft(v, sci.shape) for v in var_list]
    return Field(sci, obs_var, bad, float(mag_zp), psf_list, var_list,
                 psf_spec, var_spec)


def array_coords(shape):
    """utils.py:35-42 -- (S,2) float64, col 0 = x, col 1 = y."""
    idx = np.arange(int(np.prod(shape)))
    return np.transpose([idx % shape[1], idx // shape[1]]).astype('float64')


def mag_to_flux(mag, mag_zp):
    """utils.py:160-164."""
    return 10 ** (-0.4 * (mag - mag_zp))


# --------------------------------------------------------------------------
# rasteriser: Sky.py:14-16, PointSource.py:24-97, Sersic.py:47-153
# --------------------------------------------------------------------------
def add_sky(arr, adu):
    """Sky.py:14-16."""
    arr += adu
    return arr


def _sinc(x):
    with np.errstate(invalid='ignore', divide='ignore'):
        return np.where(x != 0, np.sin(np.pi * x) / (np.pi * x), 1.0)


def _lanczos(x, a):
    """PointSource.py:84-97."""
    return np.where(np.abs(x) < a, _sinc(x) * _sinc(x / a), 0)


def minimal_slice(xy, radius, shape):
    """PointSource.py:60-81 -- clip (yx order), numpy round-half-even."""
    radius = np.array(radius)
    shape = np.array(shape)
    clipped = np.clip(np.asarray(xy, dtype=np.float64)[::-1],
                      radius - 0.5, shape - (radius + 0.5))
    lo = np.round(clipped - radius).astype(int)
    hi = np.round(clipped + radius).astype(int)
    return slice(lo[0], hi[0] + 1), slice(lo[1], hi[1] + 1)


def add_point_source(arr, xy, mag, mag_zp, coords, method='lanczos3'):
    """PointSource.py:24-57."""
    xy = np.asarray(xy, dtype=np.float64)
    grid = coords.reshape(arr.shape + (2,))
    if method == 'bilinear':
        window = minimal_slice(xy, 0.5, arr.shape)
        kern = np.prod(1 - np.abs(grid[window] - xy), axis=-1)
    elif method == 'lanczos3':
        window = minimal_slice(xy, 3, arr.shape)
        kern = np.prod(_lanczos(grid[window] - xy, 3), axis=-1)
    else:
        raise ValueError('Unknown shift method: {}'.format(method))
    arr[window] += kern * mag_to_flux(mag, mag_zp)
    return arr


def sersic_kappa(index):
    """Sersic.py:47-53."""
    return _gammaincinv(2 * index, 0.5)


def sersic_sb_eff(flux_tot, index, reff, reff_b, kappa):
    """Sersic.py:55-71."""
    return flux_tot / (np.pi * reff * reff_b * 2 * index *
                       np.exp(kappa + np.log(kappa) * -2 * index) *
                       _gamma(2 * index))


def sersic_xform(reff, reff_b, angle, angle_degrees):
    """Sersic.py:80-91 -- inverse scale * inverse rotation."""
    theta = np.deg2rad(angle) if angle_degrees else angle
    theta = theta + 0.5 * np.pi
    s, c = np.sin(theta), np.cos(theta)
    return np.asarray(((c / reff, s / reff), (-s / reff_b, c / reff_b)))


def add_sersic(arr, xy, mag, reff, reff_b, index, angle, angle_degrees,
               mag_zp, coords):
    """Sersic.py:98-134 (+ coordinate_sq_radii :73-96, _normed_grad :136-153)."""
    kappa = sersic_kappa(index)
    sbeff = sersic_sb_eff(mag_to_flux(mag, mag_zp), index, reff, reff_b, kappa)
    offs = (coords - np.asarray(xy, dtype=np.float64)).T
    xform = sersic_xform(reff, reff_b, angle, angle_degrees)
    with np.errstate(all='ignore'):
        sq_radii = np.sum(np.dot(xform, offs) ** 2, axis=0)
        sq_delta_r = sq_radii / np.sum(offs ** 2, axis=0)
        sq_radii = sq_radii.reshape(arr.shape)
        sq_delta_r = sq_delta_r.reshape(arr.shape)
        radius_pow = 0.5 / index
        sb = np.exp(-kappa * np.expm1(np.log(sq_radii) * radius_pow))
        grad = -kappa * 2 * radius_pow * np.exp(
            np.log(sq_radii) * (radius_pow - 0.5))
        cent = sq_delta_r / 12 * grad
        arr += sbeff * sb * (1 + grad * cent)
    return arr


def raw_model(field, comps, raw_dtype=None, only=None):
    """models.py:245-253.  `comps` = list of dicts in model-file order:
    {'type':'sky','adu':..} | {'type':'ps','xy':(x,y),'mag':..,'method':..} |
    {'type':'sersic','xy':..,'mag':..,'reff':..,'reff_b':..,'index':..,
     'angle':..,'angle_degrees':bool}.  `only` restricts to one type
    (models.py:302-304 uses PointSource only)."""
    dtype = field.obs_var.dtype if raw_dtype is None else raw_dtype
    arr = np.zeros(field.shape, dtype=dtype)
    for c in comps:
        if only is not None and c['type'] != only:
            continue
        if c['type'] == 'sky':
            add_sky(arr, c['adu'])
        elif c['type'] == 'ps':
            add_point_source(arr, c['xy'], c['mag'], field.mag_zp,
                             field.coords, c.get('method', 'lanczos3'))
        elif c['type'] == 'sersic':
            add_sersic(arr, c['xy'], c['mag'], c['reff'], c['reff_b'],
                       c['index'], c['angle'], c.get('angle_degrees', False),
                       field.mag_zp, field.coords)
        else:
            raise ValueError(c['type'])
    return arr


# --------------------------------------------------------------------------
# convolution + likelihood: utils.py:25-32, models.py:213-243
# --------------------------------------------------------------------------
def convolve(img, kernel_spec):
    """utils.py:25-32.  fp64 transform regardless of input dtype (numpy 1.x
    behaviour the reference was written against; numpy>=2 would otherwise
    transform float32 input in complex64)."""
    img = np.asarray(img, dtype=np.float64)
    return np.fft.ifftshift(np.fft.irfft2(np.fft.rfft2(img) * kernel_spec))


def evaluate(field, comps, psf_index=0, raw_dtype=None, want_ps_sub=False):
    """models.py:213-241: returns (log_likelihood, images dict).  The
    log-likelihood is NaN/inf-preserving; the caller maps non-finite to -inf
    (models.py:240-241)."""
    psf_index = int(np.rint(psf_index))        # distributions.py:131-132
    raw = raw_model(field, comps, raw_dtype)
    with np.errstate(all='ignore'):
        conv = convolve(raw, field.psf_spec[psf_index])
        resid = field.sci - conv
        model_var = convolve(raw ** 2, field.var_spec[psf_index])
        ivm = 1 / (model_var + field.obs_var)
        images = {'raw_model': raw, 'convolved_model': conv,
                  'residual': resid, 'composite_ivm': ivm}
        if want_ps_sub:                         # models.py:296-306
            ps = raw_model(field, comps, raw_dtype, only='ps')
            images['point_source_subtracted'] = \
                field.sci - convolve(ps, field.psf_spec[psf_index])
        good = ~field.bad_px
        ivm_flat = ivm[good]
        resid_flat = resid[good]
        loglike = -0.5 * np.sum(resid_flat ** 2 * ivm_flat
                                - np.log(0.5 / np.pi * ivm_flat))
    return loglike, images


def log_likelihood(field, comps, psf_index=0, raw_dtype=None):
    """models.py:233-241 with the NaN guard applied."""
    ll, _ = evaluate(field, comps, psf_index, raw_dtype)
    return ll if np.isfinite(ll) else -np.inf


# --------------------------------------------------------------------------
# Derived-scalar rows: the layout the C-ABI takes (include/psfmc_hip.h).  Kept
# here as an *independent* restatement so tests can cross-check the host
# packing code of the product against it.
# --------------------------------------------------------------------------
DERIVED_SKY = 1
DERIVED_PS = 4
DERIVED_SERSIC = 9


def derived_row(field, comps, psf_index=0):
    """[sky_adu | per PS: flux,x0,y0,method | per Sersic: x0,y0,m00,m01,m10,
    m11,kappa,p,sb_eff | psf_index] -- Sky first, then point sources, then
    Sersics (each group in model-file order)."""
    sky = sum(c['adu'] for c in comps if c['type'] == 'sky')
    row = [float(sky)]
    for c in comps:
        if c['type'] == 'ps':
            meth = {'lanczos3': 0.0, 'bilinear': 1.0}[c.get('method', 'lanczos3')]
            row += [mag_to_flux(c['mag'], field.mag_zp), c['xy'][0], c['xy'][1], meth]
    for c in comps:
        if c['type'] == 'sersic':
            kappa = sersic_kappa(c['index'])
            m = sersic_xform(c['reff'], c['reff_b'], c['angle'],
                             c.get('angle_degrees', False))
            row += [c['xy'][0], c['xy'][1], m[0, 0], m[0, 1], m[1, 0], m[1, 1],
                    kappa, 0.5 / c['index'],
                    sersic_sb_eff(mag_to_flux(c['mag'], field.mag_zp),
                                  c['index'], c['reff'], c['reff_b'], kappa)]
    row.append(float(np.rint(psf_index)))
    return np.asarray(row, dtype=np.float64)
