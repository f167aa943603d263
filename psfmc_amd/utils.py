"""
One-time field setup (host glue): FITS -> the shared arrays the GPU context is
built from.  Restates the setup half of the reference's psfMC/utils.py with the
minimal FITS reader of this package; the Fourier transform of the PSFs (the
reference's `pad_and_rfft_image` / `pre_fft_psf`, utils.py:9-22, :126-133) is
NOT here -- it runs on the device inside `psfmc_ctx_create`.
"""
from math import fsum
from warnings import warn

import numpy as np

from . import fits_io


def _as_image(src, with_header=False):
    """File name or array -> native-endian ndarray (dtype preserved)."""
    if isinstance(src, str):
        return fits_io.read_image(src, with_header=with_header)
    arr = np.asarray(src)
    arr = arr.astype(arr.dtype.newbyteorder('='))
    return (arr, {}) if with_header else arr


def mag_to_flux(mag, mag_zp):
    """Total flux for a magnitude (utils.py:160-164)."""
    return 10 ** (-0.4 * (mag - mag_zp))


def norm_psf(psf_data, psf_ivm):
    """Unit-sum PSF and rescaled weight map; exact summation (utils.py:45-51)."""
    total = fsum(psf_data.flat)
    return psf_data / total, psf_ivm * total ** 2


def preprocess_obs(obs_data, obs_ivm, mask_file=None):
    """Observed image, variance map (+inf at bad pixels) and bad-pixel mask
    (utils.py:54-79).  The mask file only extends `bad_px`; variances are left
    alone."""
    obs_data, obs_hdr = _as_image(obs_data, with_header=True)
    obs_ivm = _as_image(obs_ivm)
    if obs_data.shape != obs_ivm.shape:
        raise ValueError('observation and weight map shapes differ: {} vs {}'
                         .format(obs_data.shape, obs_ivm.shape))
    bad_px = ~np.isfinite(obs_data) | ~np.isfinite(obs_ivm) | (obs_ivm <= 0)
    with np.errstate(divide='ignore', invalid='ignore'):
        obs_var = np.where(bad_px, np.inf, 1 / obs_ivm).astype(obs_ivm.dtype)
    if mask_file is not None:
        exclude = mask_from_file(mask_file, obs_hdr, obs_data.shape)
        if exclude is not None:
            bad_px = bad_px | exclude
    return obs_hdr, obs_data, obs_var, bad_px


def mask_from_file(mask_file, obs_hdr, shape):
    """FITS mask (nonzero = exclude), array, or ds9 region file in image
    coordinates (utils.py:82-103).  Region files: the fitting region is the union of
    the plain shapes minus the `-`-prefixed ones, applied in file order; everything
    outside is excluded (the reference does `~filter.mask(shape)` with pyregion).
    Only `circle`, `box` (unrotated) and `ellipse` (unrotated) in `image` coordinates
    are understood; anything else is ignored with a warning, like the reference
    without pyregion.  PARITY UNPINNED: pyregion is not available to compare with
    (pixel (ix, iy), 0-based, has ds9 image coordinates (ix + 1, iy + 1))."""
    if not isinstance(mask_file, str):
        return np.asarray(mask_file).astype(bool)
    try:
        return fits_io.read_image(mask_file).astype(bool)
    except (IOError, OSError, KeyError, ValueError):
        pass
    try:
        return ~region_filter(mask_file, shape)
    except (IOError, OSError, ValueError, UnicodeDecodeError) as err:
        warn('{} is neither a FITS mask nor a supported ds9 region file ({}); it will be '
             'ignored.'.format(mask_file, err))
    return None


def region_filter(region_file, shape):
    """Boolean image, True inside the fitting region of a ds9 region file (the reference
    hands the file to pyregion, utils.py:92-95; parity unpinned -- pyregion is absent here).
    Shapes: circle, ellipse and box (with position angle), annulus, polygon; `-shape` excludes;
    image / physical coordinates only."""
    import re
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]].astype(np.float64)
    inside = np.zeros(shape, dtype=bool)
    system, n_shapes = 'image', 0
    with open(region_file) as f:
        for raw in f:
            line = raw.split('#')[0].strip()
            if not line or line.startswith('global'):
                continue
            if line.lower() in ('image', 'physical', 'fk5', 'icrs', 'galactic', 'j2000', 'fk4'):
                system = line.lower()
                continue
            m = re.match(r'^(?:(\w+)\s*;\s*)?([+-]?)\s*(circle|box|ellipse|annulus|polygon)\s*\(([^)]*)\)',
                         line, re.I)
            if not m:
                raise ValueError('unsupported region line: ' + line)
            if (m.group(1) or system).lower() not in ('image', 'physical'):
                raise ValueError('only image coordinates are supported, got ' + (m.group(1) or system))
            args = [float(v.strip().rstrip('"\'')) for v in m.group(4).split(',')]
            kind = m.group(3).lower()
            if kind == 'polygon':
                if len(args) < 6 or len(args) % 2:
                    raise ValueError('polygon needs at least three x,y pairs: ' + line)
                # even-odd rule on pixel centres (ds9 coordinates are 1-based)
                px, py = np.array(args[0::2]) - 1.0, np.array(args[1::2]) - 1.0
                sel = np.zeros(shape, dtype=bool)
                for k in range(len(px)):
                    x1, y1, x2, y2 = px[k], py[k], px[(k + 1) % len(px)], py[(k + 1) % len(px)]
                    if y1 == y2:
                        continue
                    crosses = ((y1 > yy) != (y2 > yy)) & (xx < x1 + (yy - y1) * (x2 - x1) / (y2 - y1))
                    sel ^= crosses
            else:
                dx, dy = xx - (args[0] - 1.0), yy - (args[1] - 1.0)
                if kind == 'circle':
                    sel = dx * dx + dy * dy <= args[2] ** 2
                elif kind == 'annulus':
                    r2 = dx * dx + dy * dy
                    sel = (r2 >= args[2] ** 2) & (r2 <= args[3] ** 2)
                else:
                    # box / ellipse: optional position angle, degrees counter-clockwise from +x
                    ang = np.deg2rad(args[4]) if len(args) > 4 else 0.0
                    u = dx * np.cos(ang) + dy * np.sin(ang)
                    v = -dx * np.sin(ang) + dy * np.cos(ang)
                    if kind == 'box':
                        sel = (np.abs(u) <= args[2] / 2) & (np.abs(v) <= args[3] / 2)
                    else:
                        sel = (u / args[2]) ** 2 + (v / args[3]) ** 2 <= 1.0
            inside = (inside & ~sel) if m.group(2) == '-' else (inside | sel)
            n_shapes += 1
    if n_shapes == 0:
        raise ValueError('no shapes found')
    return inside


def preprocess_psf(psf_data, psf_ivm):
    """Normalised PSF and its variance map; zero-weight pixels are zeroed in
    both (utils.py:106-123)."""
    psf_data = np.array(_as_image(psf_data))
    psf_ivm = np.array(_as_image(psf_ivm))
    bad = ~np.isfinite(psf_data) | ~np.isfinite(psf_ivm) | (psf_ivm <= 0)
    psf_data[bad] = 0
    psf_ivm[bad] = 0
    psf_data, psf_ivm = norm_psf(psf_data, psf_ivm)
    with np.errstate(divide='ignore'):
        psf_var = np.where(psf_ivm <= 0, 0, 1 / psf_ivm)
    return psf_data, psf_var


def calculate_psf_variability(psf_data, psf_vars):
    """Adds the pixel-wise variance between PSFs (breathing / mismatch) to each
    PSF's variance map when several PSFs are given (utils.py:136-157)."""
    if len(psf_data) == 1:
        return list(psf_data), list(psf_vars)
    mismatch = np.var(psf_data, axis=0)
    return list(psf_data), [v + mismatch for v in psf_vars]


def print_progress(sample, max_samples, stage='Burning'):
    nxt = 100 * (sample + 1) // max_samples
    if nxt - 100 * sample // max_samples > 0:
        print('{}: {:d}%'.format(stage, nxt))
