"""
Synthetic field generator for the BASELINE.json configs.

Produces *inputs only* (a noisy observed image, its inverse-variance map, a
Moffat PSF and its inverse-variance map, float32 like HST drizzle products) and
the text of a psfMC model file describing the matching priors.  It is a data
generator, not part of the evaluated path: the truth image is rendered with a
few lines of plain numpy (no sub-pixel corrections) because any image of the
right shape will do.  `numpy.random.RandomState(seed)` (MT19937) keeps the
arrays identical across numpy versions.

Runs under both the system python3.10 and the conda python3.9 used by
tests/golden/make_golden.py.
"""
from __future__ import division

import numpy as np
from scipy.special import gamma, gammaincinv

MAG_ZP = 25.0
PSF_SIDE = 64


def moffat_psf(side=PSF_SIDE, fwhm=2.5, beta=3.0):
    yy, xx = np.mgrid[0:side, 0:side].astype(np.float64)
    alpha = fwhm / (2.0 * np.sqrt(2.0 ** (1.0 / beta) - 1.0))
    rr = ((xx - side // 2) ** 2 + (yy - side // 2) ** 2) / alpha ** 2
    return (1.0 + rr) ** (-beta)


def _sersic(shape, x0, y0, flux, reff, reff_b, n, angle_deg):
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]].astype(np.float64)
    th = np.deg2rad(angle_deg) + 0.5 * np.pi
    dx, dy = xx - x0, yy - y0
    u = (np.cos(th) * dx + np.sin(th) * dy) / reff
    v = (-np.sin(th) * dx + np.cos(th) * dy) / reff_b
    rho = np.sqrt(u * u + v * v)
    kap = gammaincinv(2 * n, 0.5)
    sbe = flux / (np.pi * reff * reff_b * 2 * n * np.exp(kap) * kap ** (-2 * n)
                  * gamma(2 * n))
    return sbe * np.exp(-kap * (rho ** (1.0 / n) - 1.0))


def _cconv(img, kern):
    """circular convolution with a small kernel whose origin is its centre"""
    pad = np.zeros_like(img)
    ky, kx = kern.shape
    pad[:ky, :kx] = kern
    pad = np.roll(pad, (-(ky // 2), -(kx // 2)), axis=(0, 1))
    return np.fft.irfft2(np.fft.rfft2(img) * np.fft.rfft2(pad), s=img.shape)


def truth_params(n_side, n_sersic, rng):
    """Truth parameter vector in emcee packing order (PointSource: mag, x, y;
    each Sersic: angle, index, mag, reff, reff_b, x, y)."""
    c = n_side / 2 + 0.5
    vec = [18.5, c + rng.uniform(-0.5, 0.5), c + rng.uniform(-0.5, 0.5)]
    for k in range(n_sersic):
        reff = min(9.0 + 3 * k, 1.5 + n_side / 16.0)   # stay inside the prior
        reff_b = min(6.0 + 2 * k, reff - 1.0)
        vec += [40.0 + 30.0 * k, 2.5, 20.5 + k, reff, reff_b,
                c + rng.uniform(-8, 8), c + rng.uniform(-8, 8)]
    return np.asarray(vec, dtype=np.float64)


def make_field(n_side=256, n_sersic=1, seed=0):
    """Returns dict(sci, ivm, psf, psf_ivm [float32], truth [float64 vector],
    mag_zp)."""
    rng = np.random.RandomState(seed)
    psf_true = moffat_psf()
    psf_cnt = psf_true * 1000.0
    psf_var = 0.01 ** 2 + np.abs(psf_cnt) / 50.0
    psf_obs = psf_cnt + rng.normal(size=psf_cnt.shape) * np.sqrt(psf_var)
    psf_ivm = 1.0 / psf_var

    truth = truth_params(n_side, n_sersic, rng)
    shape = (n_side, n_side)
    model = np.zeros(shape)
    flux = 10 ** (-0.4 * (truth[0] - MAG_ZP))
    x0, y0 = truth[1], truth[2]
    ix, iy = int(np.floor(x0)), int(np.floor(y0))
    fx, fy = x0 - ix, y0 - iy
    for (jy, wy) in ((iy, 1 - fy), (iy + 1, fy)):
        for (jx, wx) in ((ix, 1 - fx), (ix + 1, fx)):
            model[jy, jx] += flux * wx * wy
    for k in range(n_sersic):
        ang, n, mag, re, rb, sx, sy = truth[3 + 7 * k:10 + 7 * k]
        model += _sersic(shape, sx, sy, 10 ** (-0.4 * (mag - MAG_ZP)), re, rb,
                         n, ang)
    pnorm = psf_true / psf_true.sum()
    pvar_n = psf_var / psf_cnt.sum() ** 2
    conv = _cconv(model, pnorm)
    obs_var = 0.02 ** 2
    tot_var = np.maximum(_cconv(model ** 2, pvar_n), 0.0) + obs_var
    sci = conv + rng.normal(size=shape) * np.sqrt(tot_var)
    ivm = np.full(shape, 1.0 / obs_var)
    f32 = np.float32
    return dict(sci=sci.astype(f32), ivm=ivm.astype(f32),
                psf=psf_obs.astype(f32), psf_ivm=psf_ivm.astype(f32),
                truth=truth, mag_zp=MAG_ZP, n_side=n_side, n_sersic=n_sersic)


def model_file_text(n_side, n_sersic, sci='sci.fits', ivm='ivm.fits',
                    psf='psf.fits', psf_ivm='psf_ivm.fits', extra_config=''):
    """psfMC model-file DSL text for the synthetic field (priors of SURVEY.md
    section 8(d))."""
    lines = [
        'from numpy import array',
        "Configuration(obs_file='{}', obsivm_file='{}', psf_files='{}',".format(
            sci, ivm, psf),
        "              psfivm_files='{}', mag_zeropoint={!r}{})".format(
            psf_ivm, MAG_ZP, extra_config),
        'c = array(({0!r}, {0!r}))'.format(n_side / 2 + 0.5),
        'ms = array((8.0, 8.0))',
        'PointSource(xy=Uniform(loc=c - ms, scale=2 * ms),',
        '            mag=Uniform(loc=18.0, scale=2.0))',
    ]
    for _ in range(n_sersic):
        lines += [
            'Sersic(xy=Uniform(loc=c - ms, scale=2 * ms),',
            '       mag=Uniform(loc=19.0, scale=5.0),',
            '       reff=Uniform(loc=2.0, scale={!r}),'.format(n_side / 16.0),
            '       reff_b=Uniform(loc=2.0, scale={!r}),'.format(n_side / 16.0),
            '       index=WeibullMinimum(c=1.5, scale=4),',
            '       angle=Uniform(loc=0, scale=180), angle_degrees=True)',
        ]
    return '\n'.join(lines) + '\n'


def draw_walkers(n_side, n_sersic, n_walkers, seed=1, near_truth=None,
                 jitter=1e-2):
    """Walker parameter vectors: prior draws with the reject-until-valid rule
    (reff_b <= reff, psfMC/models.py:117-129), or truth + Gaussian jitter."""
    rng = np.random.RandomState(seed)
    c = n_side / 2 + 0.5
    dim = 3 + 7 * n_sersic
    out = np.empty((n_walkers, dim))
    if near_truth is not None:
        scale = np.abs(near_truth) * 0 + jitter
        out[:] = near_truth + rng.normal(size=out.shape) * scale
        for k in range(n_sersic):        # keep reff_b <= reff
            re = out[:, 3 + 7 * k + 3]
            rb = out[:, 3 + 7 * k + 4]
            out[:, 3 + 7 * k + 4] = np.minimum(rb, re - 1e-3)
        return out
    out[:, 0] = rng.uniform(18.0, 20.0, n_walkers)
    out[:, 1:3] = rng.uniform(c - 8, c + 8, (n_walkers, 2))
    for k in range(n_sersic):
        o = 3 + 7 * k
        out[:, o + 0] = rng.uniform(0, 180, n_walkers)
        out[:, o + 1] = 4.0 * rng.weibull(1.5, n_walkers)
        out[:, o + 2] = rng.uniform(19.0, 24.0, n_walkers)
        ra = rng.uniform(2.0, 2.0 + n_side / 16.0, n_walkers)
        rb = rng.uniform(2.0, 2.0 + n_side / 16.0, n_walkers)
        out[:, o + 3] = np.maximum(ra, rb)
        out[:, o + 4] = np.minimum(ra, rb)
        out[:, o + 5:o + 7] = rng.uniform(c - 8, c + 8, (n_walkers, 2))
    return out
