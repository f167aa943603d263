"""
Trace database: the sampled chains as a FITS binary table with run metadata in
the header (reference: psfMC/database.py, which uses astropy.table; astropy is
not available next to the ROCm stack, so a small table type and the FITS
reader/writer of `fits_io` are used).  Column order and names follow
`save_database` (:6-46): one column per stochastic (vector-valued ones such as
`xy` as 2-wide columns), then `lnprobability`, `walker`, `sample`.
"""
from collections import OrderedDict

import numpy as np

from . import fits_io

_COMMENTS = {'MCITER': 'number of retained samples',
             'MCBURN': 'number of burn-in (discarded) samples',
             'MCCHAINS': 'number of walkers run',
             'MCWALKRS': 'number of walkers run',
             'MCCONVRG': 'Has MCMC sampler converged?',
             'MCACCEPT': 'Acceptance fraction (avg of all walkers)',
             'MAPWLKR': 'Walker index of maximum posterior model',
             'MAPSAMP': 'Sample index of maximum posterior model',
             'PSFIMG': 'PSF image of maximum posterior model'}


class Table(object):
    """Minimal column table: `t[name]` -> column, `t[mask]` -> rows, `t.meta`."""

    def __init__(self, columns, meta=None):
        self.columns = OrderedDict((k, np.asarray(v)) for k, v in columns.items())
        self.meta = OrderedDict(meta or {})

    @property
    def colnames(self):
        return list(self.columns)

    def __len__(self):
        return len(next(iter(self.columns.values()))) if self.columns else 0

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.columns[key]
        return Table(OrderedDict((k, v[key]) for k, v in self.columns.items()), self.meta)

    def param_matrix(self, names):
        """[nrows, P] parameter vectors in the order of `names`."""
        cols = [self.columns[n].reshape(len(self), -1) for n in names]
        return np.concatenate(cols, axis=1).astype(np.float64)


def annotate_metadata(meta):
    """key -> (value, FITS comment) (database.py:90-109)."""
    out = OrderedDict()
    for key, val in meta.items():
        if isinstance(val, tuple):
            out[key] = val
        else:
            out[key] = (val, _COMMENTS.get(key, 'psfMC model parameter'))
    return out


def save_database(sampler, model, db_name, meta_dict=None, sample_index='reference'):
    """Write chain + lnprobability + walker/sample indices; returns the table as
    re-loaded from disk (database.py:6-46).

    sample_index: 'reference' (default) writes the `sample` column exactly as the
    reference does -- `repeat(arange(n_iter), n_walkers)` (database.py:27), which for row
    (walker w, iteration i) holds (w * n_iter + i) // n_walkers, not i; its readers
    (MAPSAMP in the header, analysis/images.py:52-66) then see the same file.
    'iteration' writes i, the iteration of the row."""
    chain = sampler.chain
    n_w, n_it, _ = chain.shape
    flat = chain.reshape(n_w * n_it, chain.shape[2])
    cols = OrderedDict()
    pos = 0
    for name, width in zip(model.param_names, model.param_lens):
        cols[name] = flat[:, pos:pos + width]      # [n, 1] for scalars, like np.split (database.py:24)
        pos += width
    cols['lnprobability'] = np.asarray(sampler.lnprobability).reshape(-1)
    cols['walker'] = np.repeat(np.arange(n_w, dtype=np.int64), n_it)
    if sample_index == 'reference':
        cols['sample'] = np.repeat(np.arange(n_it, dtype=np.int64), n_w)
    elif sample_index == 'iteration':
        cols['sample'] = np.tile(np.arange(n_it, dtype=np.int64), n_w)
    else:
        raise ValueError("sample_index must be 'reference' or 'iteration'")
    meta = OrderedDict(meta_dict or {})
    best = int(np.argmax(cols['lnprobability']))
    meta['MAPWLKR'] = int(cols['walker'][best])
    meta['MAPSAMP'] = int(cols['sample'][best])
    fits_io.write_table(db_name, cols, annotate_metadata(meta))
    return load_database(db_name)


def load_database(db_name):
    cols, hdr = fits_io.read_table(db_name)
    structural = ('XTENSION', 'BITPIX', 'NAXIS', 'NAXIS1', 'NAXIS2', 'PCOUNT', 'GCOUNT', 'TFIELDS')
    meta = OrderedDict((k, v) for k, v in hdr.items()
                       if k not in structural and not k.startswith(('TTYPE', 'TFORM', 'TDIM')))
    return Table(cols, meta)


def get_sampler_state(database):
    """Last position and log-probability of every walker, to resume sampling
    (the reference's version, database.py:59-83, is unused and indexes the wrong
    column; this one does what its docstring says)."""
    names = [n for n in database.colnames if n not in ('walker', 'sample', 'lnprobability')]
    walkers = np.unique(database['walker'])
    last = np.array([np.flatnonzero(database['walker'] == w)[-1] for w in walkers])
    return database.param_matrix(names)[last], database['lnprobability'][last]


def filter_lowp_walkers(database, percentile=10):
    """Drop walkers ALL of whose samples lie below the given percentile of
    lnprobability ("lost" walkers; database.py:112-126)."""
    cut = np.percentile(database['lnprobability'], percentile)
    ok = np.unique(database['walker'][database['lnprobability'] > cut])
    return database[np.isin(database['walker'], ok)]
