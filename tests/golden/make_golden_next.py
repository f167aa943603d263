"""
Golden vectors for the rows SURVEY.md section 8(f) marks "next" (posterior-image
accumulation, trace database, walker filter, convergence statistics, posterior
header statistics), produced by RUNNING THE REFERENCE'S OWN FUNCTIONS
(mmechtley/psfMC at /root/reference) on small inputs:

    psfMC/models.py:74-97              MultiComponentModel.accumulate_images
    psfMC/database.py:6-56             save_database / load_database (astropy.table FITS)
    psfMC/database.py:112-126          filter_lowp_walkers
    psfMC/analysis/statistics.py:46-89 potential_scale_reduction / num_effective_samples
    psfMC/analysis/images.py:104-144   _add_stats_to_header

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_golden_next.py

Harness-only shims (reference files untouched), on top of make_golden.py's two:
  * `np.alen` re-added (removed in numpy 1.23; astropy 4.3's table code calls it)
  * a stub `emcee.autocorr.AutocorrError` (psfMC/analysis/statistics.py:4 imports it;
    emcee itself is absent, so `check_convergence_autocorr` stays unpinned)
  * `ModelView`: the reference model plus `param_names` / `param_lens` /
    `param_fits_abbrs` as plain list concatenations -- the reference computes them with
    `np.sum` over ragged lists (models.py:139-162), which numpy >= 1.24 refuses
    (SURVEY.md note E); the concatenation is what that `np.sum` returned.
Inputs: the committed `edge.npz` field (64 x 128, two PSFs, FITS mask) and its parameter
vectors; a synthetic chain drawn here with a fixed seed.

Outputs: tests/golden/next.npz (data only) and tests/golden/next_db.fits -- the trace
database exactly as the reference's `save_database` (astropy) wrote it, which pins this
package's own BINTABLE reader.  The script also checks, here where astropy is available,
that astropy reads a database written by this package's writer back identically.
"""
from __future__ import division, print_function

import os
import shutil
import sys
import tempfile
import types
import warnings
from collections import OrderedDict

import numpy as np

warnings.filterwarnings('ignore')
np.asscalar = lambda a: np.asarray(a).item()
np.alen = lambda a: len(a)
_pkg = types.ModuleType('psfMC')
_pkg.__path__ = ['/root/reference/psfMC']
sys.modules['psfMC'] = _pkg
_pa = types.ModuleType('psfMC.analysis')
_pa.__path__ = ['/root/reference/psfMC/analysis']
sys.modules['psfMC.analysis'] = _pa
_em, _ac = types.ModuleType('emcee'), types.ModuleType('emcee.autocorr')


class AutocorrError(Exception):
    pass


_ac.AutocorrError = AutocorrError
_em.autocorr = _ac
sys.modules['emcee'], sys.modules['emcee.autocorr'] = _em, _ac

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

from astropy.io import fits                                   # noqa: E402
from astropy.table import Table                               # noqa: E402
from psfMC.models import MultiComponentModel                  # noqa: E402
import psfMC.database as rdb                                  # noqa: E402
import psfMC.analysis.statistics as rstat                     # noqa: E402
import psfMC.analysis.images as rimg                          # noqa: E402

IMG_KEYS = ('raw_model', 'convolved_model', 'residual', 'composite_ivm',
            'point_source_subtracted')


class ModelView(object):
    """The reference model with the three ragged-`np.sum` properties replaced by the
    list concatenation they stood for."""

    def __init__(self, model):
        self._m = model

    def __getattr__(self, name):
        return getattr(self._m, name)

    @property
    def param_names(self):
        return [n for c in self._m.components for n in c.stochastic_names()]

    @property
    def param_fits_abbrs(self):
        return [n for c in self._m.components for n in c.stochastic_names(name_attr='fitsname')]

    @property
    def param_lens(self):
        return [n for c in self._m.components for n in c.stochastic_lens()]


class FakeSampler(object):
    def __init__(self, chain, lnprob):
        self.chain, self.lnprobability = chain, lnprob


def build_reference_model(case, tmp):
    def w(name, arr):
        fits.PrimaryHDU(np.asarray(arr)).writeto(os.path.join(tmp, name), overwrite=True)
    w('sci.fits', case['sci'])
    w('ivm.fits', case['ivm'])
    w('mask.fits', case['mask'])
    for k in range(len(case['psfs'])):
        w('psf%d.fits' % k, case['psfs'][k])
        w('psfivm%d.fits' % k, case['psf_ivms'][k])
    shutil.copy(os.path.join(HERE, 'edge_model.py'), os.path.join(tmp, 'model.py'))
    return MultiComponentModel(os.path.join(tmp, 'model.py'))


def main():
    case = dict(np.load(os.path.join(HERE, 'edge.npz'), allow_pickle=False))
    tmp = tempfile.mkdtemp(prefix='psfmc_golden_next_')
    out = {}
    try:
        model = build_reference_model(case, tmp)
        view = ModelView(model)

        # ---- (f2) accumulate_images: the running mean over per-sample blobs, in two calls
        # like two sampler iterations (fitting.py:83).  (An empty blob dict -- a walker whose
        # prior is -inf -- raises KeyError in the reference, models.py:88; emcee never hands
        # one over after the start because such proposals are never accepted.)
        fin = np.flatnonzero(np.isfinite(case['lnprob']))[:12]
        order = list(fin)
        blobs = []
        for i in order:
            lp, b = MultiComponentModel.log_posterior(case['params'][i].copy(), model=model)
            assert lp == case['lnprob'][i] and len(b) == 5
            blobs.append(b)
        model.reset_images()
        model.accumulate_images(blobs[:6])
        model.accumulate_images(blobs[6:])
        out['acc_rows'] = np.array(order, dtype=np.int64)
        out['acc_split'] = np.int64(6)
        out['acc_count'] = np.int64(model.accumulated_samples)
        for k in IMG_KEYS:
            out['acc_' + k] = np.array(model.posterior_images[k], dtype=np.float64)

        # ---- (f4) a synthetic chain -> the reference's own database file
        rng = np.random.RandomState(20261004)
        n_w, n_it, dim = 8, 30, int(model.num_params)
        centre = case['params'][fin[0]]
        chain = centre + np.cumsum(rng.normal(size=(n_w, n_it, dim)) * 0.01, axis=1)
        chain[:, :, -1] = rng.randint(0, 2, size=(n_w, n_it))    # PSF_Index samples
        lnprob = -0.5 * np.sum(((chain - centre) / 0.05) ** 2, axis=2)
        lnprob[5] -= 1e4                                          # a lost walker
        lnprob[2, :20] -= 1e4                                     # one that recovers
        sampler = FakeSampler(chain, lnprob)
        meta = OrderedDict([('MCITER', n_it), ('MCBURN', 7), ('MCCHAINS', n_w),
                            ('MCCONVRG', False), ('MCACCEPT', 0.3125)])
        db_path = os.path.join(HERE, 'next_db.fits')
        db = rdb.save_database(sampler, view, db_path, meta_dict=meta.copy())
        out['chain'], out['lnprob'] = chain, lnprob
        out['db_colnames'] = np.array(db.colnames)
        out['db_walker'] = np.asarray(db['walker'], dtype=np.int64)
        out['db_sample'] = np.asarray(db['sample'], dtype=np.int64)
        out['db_lnprobability'] = np.asarray(db['lnprobability'], dtype=np.float64)
        out['db_xy'] = np.asarray(db['3_Sersic_xy'], dtype=np.float64)
        out['db_mapwlkr'] = np.int64(db.meta['MAPWLKR'])
        out['db_mapsamp'] = np.int64(db.meta['MAPSAMP'])
        meta_keys = [k for k in db.meta if k in ('MCITER', 'MCBURN', 'MCCHAINS', 'MCCONVRG',
                                                 'MCACCEPT', 'MAPWLKR', 'MAPSAMP')]
        out['db_meta_keys'] = np.array(meta_keys)
        out['db_meta_vals'] = np.array([repr(db.meta[k]) for k in meta_keys])

        # ---- filter_lowp_walkers at several percentiles
        for pct in (10, 30, 60):
            kept = rdb.filter_lowp_walkers(db, percentile=pct)
            out['filter%d_walkers' % pct] = np.unique(np.asarray(kept['walker'], dtype=np.int64))
            out['filter%d_rows' % pct] = np.int64(len(kept))

        # ---- Gelman-Rubin statistics on the walkers' traces of three parameters
        stats = []
        for col in (0, 1, 9):
            traces = [chain[w, :, col] for w in range(n_w)]
            stats.append([rstat.potential_scale_reduction(traces),
                          rstat.num_effective_samples(traces)])
        const = [np.full(n_it, 3.0) for _ in range(3)]                # zero within-variance
        stats.append([rstat.potential_scale_reduction(const), rstat.num_effective_samples(const)])
        out['stat_cols'] = np.array([0, 1, 9, -1], dtype=np.int64)
        out['stat_values'] = np.array(stats, dtype=np.float64)

        # ---- posterior statistics in the image header (on the 10 % filtered database)
        filtered = rdb.filter_lowp_walkers(db, percentile=10)
        header = fits.Header()
        rimg._add_stats_to_header(header, view, filtered)
        cards = [(c.keyword, c.value) for c in header.cards if c.keyword]
        out['hdr_keys'] = np.array([k for k, _ in cards])
        out['hdr_vals'] = np.array([str(v) for _, v in cards])
        out['hdr_types'] = np.array([type(v).__name__ for _, v in cards])

        # ---- this package's writer, read back by astropy (checked here, where astropy exists)
        sys.path.insert(0, ROOT)
        from psfmc_amd import fits_io
        cols = OrderedDict((n, np.asarray(db[n])) for n in db.colnames)
        mine = os.path.join(tmp, 'mine_db.fits')
        fits_io.write_table(mine, cols, rdb.annotate_metadata(OrderedDict(
            (k, db.meta[k]) for k in meta_keys)))
        back = Table.read(mine, format='fits')
        assert back.colnames == db.colnames
        for n in db.colnames:
            assert np.array_equal(np.asarray(back[n]), np.asarray(db[n])), n
        for k in meta_keys:
            assert back.meta[k] == db.meta[k], k
        print('astropy reads the package-written table identically (%d columns)' % len(db.colnames))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    np.savez_compressed(os.path.join(HERE, 'next.npz'), **out)
    print('next.npz: %d accumulated samples, database %d rows x %d columns, MAPWLKR %d MAPSAMP %d'
          % (out['acc_count'], len(out['db_walker']), len(out['db_colnames']),
             out['db_mapwlkr'], out['db_mapsamp']))
    print('header cards:', list(zip(out['hdr_keys'], out['hdr_vals']))[:40])


if __name__ == '__main__':
    main()
