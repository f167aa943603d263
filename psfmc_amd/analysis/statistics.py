"""Convergence diagnostics (reference: psfMC/analysis/statistics.py)."""
from warnings import warn

import numpy as np

from ..sampler import AutocorrError


def _chain_stats(traces):
    samples = np.column_stack(traces)            # samples in rows, chains in columns
    n, m = samples.shape
    means = samples.mean(axis=0)
    between = n / (m - 1) * np.sum((means - means.mean()) ** 2)
    within = np.mean(np.sum((samples - means) ** 2, axis=0) / (n - 1))
    pooled = (n - 1) / n * within + between / n
    return n, m, between, within, pooled


def potential_scale_reduction(traces):
    """Gelman-Rubin R-hat of two or more traces (statistics.py:46-65)."""
    n, m, _, within, pooled = _chain_stats(traces)
    if within == 0:
        return 1.0
    return np.sqrt((m + 1) / m * pooled / within + (1 - n) / (m * n))


def num_effective_samples(traces):
    """Gelman's n_eff, capped at the number of samples (statistics.py:68-89)."""
    n, m, between, _, pooled = _chain_stats(traces)
    if between == 0 or pooled > between:
        return n * m
    return n * m * pooled / between


def check_convergence_autocorr(sampler, min_chain_to_tau_ratio=10, verbose=0):
    """True when every parameter's integrated autocorrelation time (quick c=1
    window) is shorter than chain length / ratio (statistics.py:134-155)."""
    try:
        acorr = sampler.get_autocorr_time(c=1)
    except AutocorrError:
        warn('unable to estimate the autocorrelation time, assuming chain is not converged')
        return False
    if verbose > 0:
        print('Autocorrelation times: {}'.format(acorr))
    return bool(np.all(sampler.chain.shape[1] > min_chain_to_tau_ratio * acorr))
