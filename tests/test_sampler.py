"""EnsembleSampler stand-in (CPU): API surface the reference touches and
statistical behaviour on analytic targets."""
import numpy as np
import pytest

from psfmc_amd.sampler import EnsembleSampler, AutocorrError, integrated_time
from psfmc_amd.analysis import check_convergence_autocorr, potential_scale_reduction


def gauss_batch(cov):
    icov = np.linalg.inv(cov)
    return lambda x: -0.5 * np.einsum('ij,jk,ik->i', x, icov, x)


def test_recovers_gaussian_moments():
    cov = np.array([[2.0, 0.6, 0.0], [0.6, 1.0, -0.3], [0.0, -0.3, 0.5]])
    sampler = EnsembleSampler(40, 3, batch_lnpostfn=gauss_batch(cov))
    sampler.random_state = np.random.RandomState(12).get_state()
    p0 = np.random.RandomState(1).normal(size=(40, 3))
    pos = None
    for pos, lnp, state in sampler.sample(p0, iterations=300):
        pass
    sampler.reset()
    for pos, lnp, state in sampler.sample(pos, iterations=1500):
        pass
    flat = sampler.flatchain
    assert sampler.chain.shape == (40, 1500, 3) and sampler.lnprobability.shape == (40, 1500)
    assert np.allclose(flat.mean(axis=0), 0, atol=0.12)
    assert np.allclose(np.cov(flat.T), cov, atol=0.25)
    acc = sampler.acceptance_fraction
    assert acc.shape == (40,) and 0.3 < acc.mean() < 0.85
    tau = sampler.get_autocorr_time(c=1)
    assert tau.shape == (3,) and np.all(tau > 1) and np.all(tau < 200)
    assert check_convergence_autocorr(sampler)
    halves = [sampler.chain[:20, :, 0].ravel(), sampler.chain[20:, :, 0].ravel()]
    assert abs(potential_scale_reduction(halves) - 1) < 0.05
    # lnprobability stores the value of the stored position
    assert np.allclose(sampler.lnprobability[:, -1], gauss_batch(cov)(sampler.chain[:, -1]))


def test_per_walker_function_pool_and_blobs():
    calls = []

    def lnpost(x, scale, model=None):
        assert model == 'm'
        return -0.5 * np.sum((x / scale) ** 2), {'tag': float(x[0])}

    class Pool(object):
        def map(self, fn, items):
            items = list(items)
            calls.append(len(items))
            return [fn(i) for i in items]

    s = EnsembleSampler(8, 2, lnpost, args=[2.0], kwargs={'model': 'm'}, pool=Pool())
    p0 = np.random.RandomState(3).normal(size=(8, 2))
    out = list(s.sample(p0, iterations=3))
    assert calls == [8] + [4] * 6                    # start + two half-ensembles per iteration
    assert len(out[0]) == 4 and len(out[0][3]) == 8   # (pos, lnprob, rstate, blobs)
    for walker in range(8):                           # blobs follow the walkers
        assert out[-1][3][walker]['tag'] == out[-1][0][walker, 0]
    assert len(s.blobs) == 3
    s.clear_blobs()
    assert s.blobs == []
    s.reset()
    assert s.chain.shape == (8, 0, 2) and s.iterations == 0


def test_seeded_runs_are_reproducible_and_nan_is_rejected():
    f = gauss_batch(np.eye(2))
    runs = []
    for _ in range(2):
        s = EnsembleSampler(10, 2, batch_lnpostfn=f)
        s.random_state = np.random.RandomState(5).get_state()
        s.run_mcmc(np.random.RandomState(0).normal(size=(10, 2)), 20)
        runs.append(s.chain.copy())
    assert np.array_equal(runs[0], runs[1])
    with pytest.raises(ValueError):
        EnsembleSampler(5, 2, batch_lnpostfn=f)               # odd
    with pytest.raises(ValueError):
        EnsembleSampler(2, 2, batch_lnpostfn=f)               # fewer than 2*dim
    s = EnsembleSampler(4, 2, batch_lnpostfn=lambda x: np.full(len(x), np.nan))
    with pytest.raises(ValueError):
        next(s.sample(np.zeros((4, 2)) + np.arange(4)[:, None]))
    s = EnsembleSampler(4, 2, batch_lnpostfn=f)
    with pytest.raises(ValueError):
        next(s.sample(np.full((4, 2), np.inf)))
    # -inf log-probabilities are fine (walkers outside the prior support)
    s = EnsembleSampler(4, 1, batch_lnpostfn=lambda x: np.where(x[:, 0] > 0, -x[:, 0], -np.inf),
                        live_dangerously=True)
    s.run_mcmc(np.array([[1.0], [2.0], [0.5], [-1.0]]), 30)
    assert np.all(s.chain[:3] > 0)


def test_autocorr_time_of_ar1_process():
    rng = np.random.RandomState(2)
    rho, n = 0.9, 40000
    x = np.zeros(n)
    for i in range(1, n):
        x[i] = rho * x[i - 1] + rng.normal()
    tau = integrated_time(x[:, None], c=5)
    assert abs(tau[0] - (1 + rho) / (1 - rho)) < 4          # 19
    with pytest.raises(AutocorrError):
        integrated_time(x[:50, None])
