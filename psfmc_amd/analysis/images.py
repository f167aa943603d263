"""Posterior model images (reference: psfMC/analysis/images.py:17-144)."""
from collections import OrderedDict
from warnings import warn

import numpy as np

from .. import fits_io
from ..database import filter_lowp_walkers, annotate_metadata
from ..utils import print_progress

default_filetypes = ('raw_model', 'convolved_model', 'composite_ivm', 'residual',
                     'point_source_subtracted')


def posterior_stats(model, database):
    """FITS abbreviation -> 'mean +/- std' of every parameter (images.py:120-130)."""
    stats = OrderedDict()
    for name, abbr in zip(model.param_names, model.param_fits_abbrs):
        col = np.asarray(database[name], dtype=np.float64)
        mean, std = col.mean(axis=0), col.std(axis=0)
        if np.ndim(mean) == 0:
            stats[abbr] = '{:0.4g} +/- {:0.4g}'.format(mean, std)
        else:
            stats[abbr] = '({}) +/- ({})'.format(','.join('{:0.4g}'.format(v) for v in mean),
                                                 ','.join('{:0.4g}'.format(v) for v in std))
    return stats


def posterior_psf_filename(model, database):
    """The PSF file recorded as PSFIMG.  With several PSFs the reference takes the
    `PSF_Index` of row `argmax(database['walker'])` -- the first row of the highest-numbered
    walker, not the MAP row its comment announces (images.py:133-139); the same row is
    used here so that the header matches the reference's for the same database."""
    sel = model.config.psf_selector
    if len(sel.filenames) > 1 and 'PSF_Index' in database.colnames:
        row = int(np.argmax(database['walker']))
        idx = int(np.rint(np.ravel(database['PSF_Index'][row])[0]))
        return sel.filenames[min(max(idx, 0), len(sel.filenames) - 1)]
    return sel.filenames[0]


def save_posterior_images(model, database, output_name='out_{}', mode='weighted',
                          filetypes=default_filetypes, bad_px_value=0,
                          walker_min_percentile=10, batch=256):
    """Write the posterior images: per-pixel mean over all retained samples
    ('weighted') or the maximum a posteriori sample ('maximum' / 'MAP').  When the
    model has not accumulated exactly these samples during sampling, the images are
    recomputed from the database rows in GPU batches."""
    if '{}' not in output_name:
        output_name += '_{}'
    database = filter_lowp_walkers(database, percentile=walker_min_percentile)
    header = OrderedDict(model.obs_header)
    for key, (val, _) in annotate_metadata(database.meta).items():
        header[key] = val
    header.update(posterior_stats(model, database))
    header['PSFIMG'] = str(posterior_psf_filename(model, database))[:60]

    unknown = [f for f in filetypes if f not in default_filetypes]
    if unknown:
        warn('Unknown filetypes requested: {}'.format(unknown))
        filetypes = [f for f in filetypes if f in default_filetypes]
    names = model.param_names
    out = {}
    if mode in ('maximum', 'MAP'):
        best = int(np.argmax(database['lnprobability']))
        theta = database.param_matrix(names)[best:best + 1]
        imgs = model.sample_images(theta, filetypes)
        out = {k: imgs[k][0] for k in filetypes}
    elif mode == 'weighted':
        total = len(database)
        if total != model.accumulated_samples:
            model.reset_images()
            theta = database.param_matrix(names)
            batch = max(batch, getattr(model, '_max_walkers', batch))
            for lo in range(0, total, batch):
                print_progress(lo, total, 'Creating posterior images')
                model.accumulate_samples(theta[lo:lo + batch])
        post = model.collect_posterior_images()
        out = {k: np.array(post[k]) for k in filetypes}
    else:
        warn('Unknown posterior output mode ({}).'.format(mode))
        return
    for kind in filetypes:
        img = out[kind]
        img[~np.isfinite(img)] = bad_px_value
        header['OBJECT'] = kind
        fits_io.write_image(output_name.format(kind) + '.fits', img, header=header)
