"""The rows SURVEY.md section 8(f) marks "next", pinned to the reference's OWN functions:
`tests/golden/next.npz` and `tests/golden/next_db.fits` were produced by running
psfMC's `accumulate_images`, `save_database` (astropy.table), `filter_lowp_walkers`,
`potential_scale_reduction`, `num_effective_samples` and `_add_stats_to_header`
(tests/golden/make_golden_next.py).  CPU tests cover the host code; the `-m gpu` test
compares the device-resident image sums with the reference's running mean."""
import os
from collections import OrderedDict

import numpy as np
import pytest

import helpers
import psfmc_oracle as orc
from psfmc_amd import fits_io
from psfmc_amd.analysis import images as pimg
from psfmc_amd.analysis.statistics import potential_scale_reduction, num_effective_samples
from psfmc_amd.database import save_database, load_database, filter_lowp_walkers, annotate_metadata

IMG_KEYS = ('raw_model', 'convolved_model', 'residual', 'composite_ivm', 'point_source_subtracted')


class FakeSampler(object):
    def __init__(self, chain, lnprob):
        self.chain, self.lnprobability = chain, lnprob


@pytest.fixture(scope='module')
def nxt():
    return dict(np.load(os.path.join(helpers.GOLDEN, 'next.npz'), allow_pickle=False))


@pytest.fixture(scope='module')
def edge_model(tmp_path_factory):
    case = helpers.load_case('edge')
    model = helpers.build_model('edge', case, tmp_path_factory.mktemp('edge_next'), max_walkers=16)
    yield case, model
    model.close()


def test_reader_loads_the_reference_written_database(nxt):
    """next_db.fits was written by the reference's save_database through astropy.table."""
    db = load_database(os.path.join(helpers.GOLDEN, 'next_db.fits'))
    assert db.colnames == [str(s) for s in nxt['db_colnames']]
    assert np.array_equal(db['walker'], nxt['db_walker']) and db['walker'].dtype == np.int64
    assert np.array_equal(db['sample'], nxt['db_sample'])
    assert np.array_equal(db['lnprobability'], nxt['db_lnprobability'])
    assert np.array_equal(db['3_Sersic_xy'], nxt['db_xy'])
    assert db['0_Sky_adu'].shape == (240, 1)              # scalars are [n, 1] columns (TDIM '(1)')
    n_w, n_it, dim = nxt['chain'].shape
    names = [n for n in db.colnames if n not in ('lnprobability', 'walker', 'sample')]
    assert np.array_equal(db.param_matrix(names), nxt['chain'].reshape(n_w * n_it, dim))
    for key, val in zip(nxt['db_meta_keys'], nxt['db_meta_vals']):
        assert repr(db.meta[str(key)]) == str(val), key


def test_save_database_writes_what_the_reference_writes(nxt, edge_model, tmp_path):
    _, model = edge_model
    sampler = FakeSampler(nxt['chain'], nxt['lnprob'])
    meta = OrderedDict([('MCITER', 30), ('MCBURN', 7), ('MCCHAINS', 8), ('MCCONVRG', False),
                        ('MCACCEPT', 0.3125)])
    mine = save_database(sampler, model, str(tmp_path / 'mine_db.fits'), meta_dict=meta)
    ref = load_database(os.path.join(helpers.GOLDEN, 'next_db.fits'))
    assert mine.colnames == ref.colnames
    for name in ref.colnames:
        assert mine[name].shape == ref[name].shape and mine[name].dtype == ref[name].dtype, name
        assert np.array_equal(mine[name], ref[name]), name
    assert mine.meta['MAPWLKR'] == nxt['db_mapwlkr'] and mine.meta['MAPSAMP'] == nxt['db_mapsamp']
    for key in nxt['db_meta_keys']:
        assert mine.meta[str(key)] == ref.meta[str(key)], key
    # the column description cards of the two files agree (names, formats, per-row shapes)
    _, h_mine = fits_io.read_table(str(tmp_path / 'mine_db.fits'))
    _, h_ref = fits_io.read_table(os.path.join(helpers.GOLDEN, 'next_db.fits'))
    for key in h_ref:
        if key.startswith(('TTYPE', 'TFORM', 'TDIM', 'NAXIS', 'TFIELDS')):
            assert str(h_mine.get(key)).strip() == str(h_ref[key]).strip(), key


def test_filter_lowp_walkers_matches_reference(nxt):
    db = load_database(os.path.join(helpers.GOLDEN, 'next_db.fits'))
    for pct in (10, 30, 60):
        kept = filter_lowp_walkers(db, percentile=pct)
        assert np.array_equal(np.unique(kept['walker']), nxt['filter%d_walkers' % pct])
        assert len(kept) == int(nxt['filter%d_rows' % pct])


def test_gelman_rubin_statistics_match_reference(nxt):
    chain = nxt['chain']
    for col, (psrf, neff) in zip(nxt['stat_cols'], nxt['stat_values']):
        traces = ([chain[w, :, col] for w in range(chain.shape[0])] if col >= 0
                  else [np.full(chain.shape[1], 3.0) for _ in range(3)])
        assert abs(potential_scale_reduction(traces) - psrf) <= 1e-13 * abs(psrf)
        assert abs(num_effective_samples(traces) - neff) <= 1e-13 * abs(neff)


def test_posterior_header_statistics_match_reference(nxt, edge_model):
    """Every card `_add_stats_to_header` adds (images.py:104-144): sampler metadata,
    'mean +/- std' of every parameter, PSFIMG."""
    _, model = edge_model
    db = filter_lowp_walkers(load_database(os.path.join(helpers.GOLDEN, 'next_db.fits')), percentile=10)
    header = OrderedDict()
    for key, (val, _) in annotate_metadata(db.meta).items():
        header[key] = val
    header.update(pimg.posterior_stats(model, db))
    header['PSFIMG'] = os.path.basename(str(pimg.posterior_psf_filename(model, db)))
    header = OrderedDict((k.upper()[:8], v) for k, v in header.items())    # as fits_io writes them
    for key, val, typ in zip(nxt['hdr_keys'], nxt['hdr_vals'], nxt['hdr_types']):
        assert str(key) in header, key
        assert str(header[str(key)]) == str(val), (key, header[str(key)], val)


def test_host_running_mean_matches_reference(nxt, edge_model):
    """`accumulate_images` (the emcee-blob route) on oracle-rendered sample images against the
    reference's posterior images of the same vectors (oracle images equal the reference's
    blobs to <= 1e-11, tests/golden/make_golden.py)."""
    case, model = edge_model
    field = helpers.oracle_field(case)
    blobs = []
    for i in nxt['acc_rows']:
        comps, psf = helpers.comps_from_theta(helpers.LAYOUT['edge'], case['params'][i], True)
        _, imgs = orc.evaluate(field, comps, int(np.rint(psf)), raw_dtype=None, want_ps_sub=True)
        blobs.append({k: np.asarray(imgs[k], dtype=np.float64) for k in IMG_KEYS})
    model.reset_images()
    split = int(nxt['acc_split'])
    model.accumulate_images(blobs[:split])
    model.accumulate_images(blobs[split:])
    assert model.accumulated_samples == int(nxt['acc_count'])
    for k in IMG_KEYS:
        ref, got = nxt['acc_' + k], model.posterior_images[k]
        assert np.array_equal(np.isfinite(got), np.isfinite(ref)), k
        fin = np.isfinite(ref)
        assert np.abs(got[fin] - ref[fin]).max() <= 1e-10 * np.abs(ref[fin]).max(), k
    model.reset_images()


@pytest.mark.gpu
@pytest.mark.parametrize('backend', ['fused', 'hipfft'])
def test_device_accumulation_matches_reference_posterior_images(nxt, tmp_path, backend):
    """The device-resident sums (psfmc_accumulate_images) of the same 12 vectors against the
    posterior images the REFERENCE's accumulate_images produced (models.py:74-97), in two
    calls like two sampler iterations.  Tolerance: the reference's float32 raw model."""
    case = helpers.load_case('edge')
    model = helpers.build_model('edge', case, tmp_path, backend=backend, max_walkers=16)
    theta = case['params'][nxt['acc_rows']]
    split = int(nxt['acc_split'])
    model.reset_images()
    model.accumulate_samples(theta[:split])
    model.accumulate_samples(theta[split:])
    post = model.collect_posterior_images()
    assert model.accumulated_samples == int(nxt['acc_count'])
    for k in IMG_KEYS:
        ref, got = nxt['acc_' + k], post[k]
        fin = np.isfinite(ref)
        assert np.array_equal(np.isfinite(got), fin), k
        assert np.abs(got[fin] - ref[fin]).max() <= 3e-7 * np.abs(ref[fin]).max(), k
    model.close()
