"""GPU tests in the regimes bench.py times: the headline batch (256^2, W = 4096 and the
ragged W = 4085, default options: two passes in flight, library-chosen pass size) and the
single-GPU shares of BASELINE configs 3 and 4 (512^2 x 1024 walkers, 1024^2 x 256 walkers).

The pass-splitting arithmetic (psfmc_hip.hip `pass_size` / `run_pipeline`) is where round
1's one memory bug lived, so known vectors sit at the FIRST and LAST slot of every internal
pass: `synth256` golden vectors against the reference's own log-posteriors (<= 1e-6, the
reference's float32 raw-model floor) and the fp64 oracle (<= 1e-9); 64 random walkers of the
batch against the oracle; and per-walker results bitwise independent of where in the batch a
vector sits."""
import numpy as np
import pytest

import helpers
import psfmc_oracle as orc
import synth_field

pytestmark = pytest.mark.gpu

REF_TOL = 1e-6
ORACLE_TOL = 1e-9


def pass_slots(eng, n_w):
    size = eng.pass_size(n_w)
    starts = list(range(0, n_w, size))
    ends = [min(s + size, n_w) - 1 for s in starts]
    return size, sorted(set(starts + ends))


@pytest.mark.parametrize('n_w', [4096, 4085])
def test_headline_batch_with_golden_vectors_at_every_pass_boundary(tmp_path, n_w):
    case = helpers.load_case('synth256')
    model = helpers.build_model('synth256', case, tmp_path, backend='fused', max_walkers=4096)
    eng = model.engine
    assert eng.get_option('streams') == 2                 # the default the bench times
    size, slots = pass_slots(eng, n_w)
    assert size < n_w and len(slots) >= 2 * (n_w // size)
    half = n_w // 2
    theta = np.vstack([synth_field.draw_walkers(256, 1, half, seed=31),
                       synth_field.draw_walkers(256, 1, n_w - half, seed=32,
                                                near_truth=case['params'][-1])])
    which = np.arange(len(slots)) % len(case['params'])
    theta[slots] = case['params'][which]
    got = model.log_posterior_batch(theta)
    assert got.shape == (n_w,) and not np.isnan(got).any()
    # (1) the reference's own log-posteriors at the pass boundaries
    assert helpers.rel_err(got[slots], case['lnprob'][which]) <= REF_TOL
    # (2) the fp64 oracle there
    want = case['loglike_f64'][which] + case['lnprior'][which]
    want = np.where(np.isfinite(case['lnprior'][which]), want, -np.inf)
    assert helpers.rel_err(got[slots], want) <= ORACLE_TOL
    # (3) 64 random walkers of the batch against the oracle
    field = helpers.oracle_field(case)
    layout = helpers.LAYOUT['synth256']
    pick = np.random.RandomState(5).choice(n_w, 64, replace=False)
    prior = model.log_priors_batch(theta[pick])
    ref = np.array([helpers.oracle_loglike(field, layout, t) if np.isfinite(p) else -np.inf
                    for t, p in zip(theta[pick], prior)]) + np.where(np.isfinite(prior), prior, 0.0)
    assert helpers.rel_err(got[pick], ref) <= ORACLE_TOL
    # (4) position independence, bitwise: the boundary vectors alone, and the batch reversed
    assert np.array_equal(model.log_posterior_batch(theta[slots]), got[slots])
    assert np.array_equal(model.log_posterior_batch(theta[::-1])[::-1], got)
    # (5) the likelihood-only entry point over the same batch agrees with the raw-vector one
    ll = model.log_likelihood_batch(theta[pick])
    fin = np.isfinite(prior)
    assert helpers.rel_err(ll[fin] + prior[fin], got[pick][fin]) <= 1e-12
    model.close()


@pytest.mark.parametrize('n_side,n_sersic,n_w', [(512, 2, 1024), (1024, 4, 256)])
def test_config3_and_config4_share_at_full_batch(n_side, n_sersic, n_w):
    """BASELINE config 3 (512^2, 1 PS + 2 Sersic, 1024 walkers) and config 4's per-GPU share
    (1024^2, 1 PS + 4 Sersic, 2048 / 8 = 256 walkers): the whole batch in one call with
    default options; first / middle / last walker against the oracle, pass-boundary slots
    bitwise equal to their evaluation in a small batch."""
    from test_gpu_fullsize import make_model
    model, fld = make_model(n_side, n_sersic, 'fused', max_walkers=n_w)
    eng = model.engine
    size, slots = pass_slots(eng, n_w)
    half = n_w // 2
    theta = np.vstack([synth_field.draw_walkers(n_side, n_sersic, half, seed=41),
                       synth_field.draw_walkers(n_side, n_sersic, n_w - half, seed=42,
                                                near_truth=fld['truth'])])
    theta[slots[-1]] = fld['truth']
    got = model.log_posterior_batch(theta)
    assert np.isfinite(got).all()
    field = orc.make_field(fld['sci'], fld['ivm'], [fld['psf']], [fld['psf_ivm']], mag_zp=fld['mag_zp'])
    layout = helpers.synth_layout(n_sersic)
    sample = [0, n_w // 2 + 1, n_w - 1]
    prior = model.log_priors_batch(theta[sample])
    for i, p in zip(sample, prior):
        want = helpers.oracle_loglike(field, layout, theta[i]) + p
        assert abs(got[i] - want) <= 1e-10 * abs(want), (i, got[i], want)
    small = model.log_posterior_batch(theta[slots[:16]])
    assert np.array_equal(small, got[slots[:16]])
    assert np.array_equal(model.log_posterior_batch(theta[slots[-3:]]), got[slots[-3:]])
    model.close()


def test_device_group_splits_walkers_over_devices(tmp_path):
    """psfmc_group_* (one process, several devices): with the one GPU of the test box listed
    twice the walkers are split over two contexts; results equal the single context's bit for
    bit, for the raw-vector and the derived-row entry points, ragged and tiny batches included."""
    case = helpers.load_case('synth256')
    model = helpers.build_model('synth256', case, tmp_path, backend='fused', max_walkers=128)
    single = model.log_posterior_batch(case['params'])
    assert helpers.rel_err(single, case['lnprob']) <= REF_TOL
    grp = model.device_group([0, 0], max_walkers=128)
    for n in (65, 64, 3, 1):
        assert np.array_equal(grp.logpost_theta(case['params'][:n]), single[:n]), n
    fin = np.isfinite(case['lnprior'])
    rows = model.derived_rows(case['params'][fin])
    assert np.array_equal(grp.loglike(rows), model.engine.loglike(rows))
    skip = np.zeros(len(rows), dtype=bool)
    skip[::3] = True
    got = grp.loglike(rows, skip)
    assert np.all(got[skip] == -np.inf) and np.array_equal(got[~skip], model.engine.loglike(rows)[~skip])
    with pytest.raises(Exception):
        grp.logpost_theta(np.tile(case['params'], (3, 1)))            # 195 > max_walkers
    grp.close()
    model.close()
