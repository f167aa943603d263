"""
`BatchLogPosterior`: the batched evaluator and its adapters for emcee-style
samplers.

emcee evaluates a list of parameter vectors through `pool.map(lnpostfn, list)`
when a `pool` object is given (psfMC/fitting.py:56-58 builds the sampler with
`lnpostfn=model.log_posterior`).  `as_pool()` returns an object whose `map`
ignores the function, stacks the vectors and evaluates them in ONE GPU batch,
so an unmodified EnsembleSampler becomes batched; `as_lnpostfn()` is the
per-walker callable for samplers without a pool hook.
"""
import numpy as np


class BatchLogPosterior(object):
    def __init__(self, model, blobs=False):
        self.model = model
        self.blobs = blobs
        self.n_calls = 0
        self.n_evals = 0

    def __call__(self, theta):
        """[W, P] (or [P]) float64 -> [W] log-posteriors (never NaN)."""
        theta = np.asarray(theta, dtype=np.float64)
        single = theta.ndim == 1
        out = self.model.log_posterior_batch(theta)
        self.n_calls += 1
        self.n_evals += out.shape[0]
        return out[0] if single else out

    def map(self, func, iterable):
        """pool.map replacement: `func` is ignored (it is the per-walker
        log_posterior); returns [(lnprob, blobs), ...] like it would."""
        vectors = [np.asarray(v, dtype=np.float64) for v in iterable]
        if not vectors:
            return []
        theta = np.stack(vectors)
        lnp = self(theta)
        if self.blobs:
            imgs = self.model.sample_images(theta)
            finite = np.isfinite(self.model.log_priors_batch(theta))
            return [(float(lnp[i]),
                     {k: v[i] for k, v in imgs.items()} if finite[i] else {})
                    for i in range(len(vectors))]
        return [(float(v), {}) for v in lnp]

    def as_pool(self):
        return self

    def as_lnpostfn(self):
        def lnpostfn(param_values, **kwargs):
            return float(self(param_values)), {}
        return lnpostfn
