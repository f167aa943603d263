"""
Affine-invariant ensemble sampler (Goodman & Weare 2010 stretch move) with the
interface of emcee 2.2.1's `EnsembleSampler`, which the reference drives
(psfMC/fitting.py:56-86, psfMC/database.py:18-19, analysis/statistics.py:145).

emcee is not installable here (no network) and is not vendored by the
reference, so this is written from the published algorithm and emcee's
documented API; SURVEY.md Appendix A lists the behaviour it reproduces:
two half-ensemble proposals per iteration, `z ~ g(z) ∝ 1/sqrt(z)` on [1/a, a],
acceptance `ln q = (dim-1) ln z + lnp(new) - lnp(old) > ln U`, a
`numpy.random.RandomState` owned by the sampler, `(lnprob, blob)` results,
`ValueError` on NaN log-probabilities or non-finite parameters, the `pool.map`
hook, `chain [nwalkers, iterations, dim]`, `lnprobability`, `acceptance_fraction`,
`reset`, `clear_blobs`, `get_autocorr_time`.  PARITY UNPINNED against emcee
itself (no copy available); its statistical behaviour is tested on Gaussian
targets (tests/test_sampler.py).

What is new: every half-step evaluates its walkers in ONE call --
`batch_lnpostfn([n, dim]) -> [n]` (e.g. `model.log_posterior_batch`, one GPU
batch) or `pool.map` (e.g. `BatchLogPosterior.as_pool()`), instead of one Python
call per walker.
"""
import threading

import numpy as np

__all__ = ['EnsembleSampler', 'DeviceEnsembleSampler', 'FieldSetSampler', 'AutocorrError', 'integrated_time']


class AutocorrError(Exception):
    """The chain is too short to estimate the autocorrelation time."""


def autocorr_function(x, axis=0):
    """Normalised autocorrelation function along `axis` (FFT based)."""
    x = np.atleast_1d(x)
    n = x.shape[axis]
    idx = [slice(None)] * x.ndim
    f = np.fft.fft(x - np.mean(x, axis=axis, keepdims=True), n=2 * n, axis=axis)
    idx[axis] = slice(0, n)
    acf = np.fft.ifft(f * np.conjugate(f), axis=axis)[tuple(idx)].real
    idx[axis] = slice(0, 1)
    return acf / acf[tuple(idx)]


def integrated_time(x, low=10, high=None, step=1, c=10, axis=0):
    """Integrated autocorrelation time with the self-consistent window of
    emcee 2.x: the smallest window M in [low, high) with M > c * tau(M)."""
    size = 0.5 * x.shape[axis]
    if int(c * low) >= size:
        raise AutocorrError('The chain is too short')
    f = autocorr_function(x, axis=axis)
    if high is None:
        high = int(size / c)
    idx = [slice(None)] * f.ndim
    for m in np.arange(low, high, step).astype(int):
        idx[axis] = slice(1, m)
        tau = 1 + 2 * np.sum(f[tuple(idx)], axis=axis)
        if np.all(tau > 1.0) and m > c * tau.max():
            return tau
        if c * tau.max() >= size:           # emcee 2.2.1 gives up here (recalled; parity unpinned)
            break
    raise AutocorrError('The chain is too short to reliably estimate the '
                        'autocorrelation time')


class _Wrapped(object):
    """lnpostfn with its extra arguments bound (a fresh kwargs dict per call:
    the reference's log_posterior pops 'model' from it, models.py:205)."""

    def __init__(self, f, args, kwargs):
        self.f, self.args, self.kwargs = f, list(args or []), dict(kwargs or {})

    def __call__(self, x):
        return self.f(x, *self.args, **self.kwargs)


class EnsembleSampler(object):
    def __init__(self, nwalkers, dim, lnpostfn=None, a=2.0, args=None, kwargs=None,
                 threads=1, pool=None, live_dangerously=False, batch_lnpostfn=None):
        if nwalkers % 2:
            raise ValueError('The number of walkers must be even.')
        if nwalkers < 2 * dim and not live_dangerously:
            raise ValueError('The number of walkers needs to be more than twice the '
                             'dimension of your parameter space.')
        if lnpostfn is None and batch_lnpostfn is None:
            raise ValueError('need lnpostfn or batch_lnpostfn')
        self.k, self.dim, self.a = int(nwalkers), int(dim), float(a)
        self.lnprobfn = _Wrapped(lnpostfn, args, kwargs) if lnpostfn is not None else None
        self.batch_lnpostfn = batch_lnpostfn
        self.pool = pool
        self.threads = threads
        self._random = np.random.mtrand.RandomState()
        self._last_run_mcmc_result = None
        self.reset()

    # -- state ----------------------------------------------------------------
    def reset(self):
        self.iterations = 0
        self.naccepted = np.zeros(self.k)
        self._chain = np.empty((self.k, 0, self.dim))
        self._lnprob = np.empty((self.k, 0))
        self._blobs = []
        self._last_run_mcmc_result = None

    clear_chain = reset

    def clear_blobs(self):
        self._blobs = []

    @property
    def random_state(self):
        return self._random.get_state()

    @random_state.setter
    def random_state(self, state):
        try:
            self._random.set_state(state)
        except (TypeError, ValueError):
            pass

    @property
    def chain(self):
        return self._chain

    @property
    def flatchain(self):
        s = self._chain.shape
        return self._chain.reshape(s[0] * s[1], s[2])

    @property
    def lnprobability(self):
        return self._lnprob

    @property
    def flatlnprobability(self):
        return self._lnprob.flatten()

    @property
    def blobs(self):
        return self._blobs

    @property
    def acceptance_fraction(self):
        return self.naccepted / max(self.iterations, 1)

    @property
    def acor(self):
        return self.get_autocorr_time()

    def get_autocorr_time(self, low=10, high=None, step=1, c=10):
        return integrated_time(np.mean(self._chain, axis=0), axis=0, low=low, high=high,
                               step=step, c=c)

    # -- evaluation -------------------------------------------------------------
    def _get_lnprob(self, pos):
        p = np.asarray(pos, dtype=np.float64)
        if np.any(np.isinf(p)):
            raise ValueError('At least one parameter value was infinite.')
        if np.any(np.isnan(p)):
            raise ValueError('At least one parameter value was NaN.')
        if self.batch_lnpostfn is not None:
            lnprob, blob = np.asarray(self.batch_lnpostfn(p), dtype=np.float64), None
        else:
            mapper = self.pool.map if self.pool is not None else map
            results = list(mapper(self.lnprobfn, [p[i] for i in range(len(p))]))
            try:
                lnprob = np.array([float(r[0]) for r in results])
                blob = [r[1] for r in results]
            except (IndexError, TypeError):
                lnprob = np.array([float(r) for r in results])
                blob = None
        if np.any(np.isnan(lnprob)):
            raise ValueError('lnprob returned NaN.')
        return lnprob, blob

    def _propose_stretch(self, p0, p1, lnprob0):
        s, c = np.atleast_2d(p0), np.atleast_2d(p1)
        ns, nc = len(s), len(c)
        zz = ((self.a - 1.0) * self._random.rand(ns) + 1) ** 2.0 / self.a
        rint = self._random.randint(nc, size=(ns,))
        q = c[rint] - zz[:, np.newaxis] * (c[rint] - s)
        newlnprob, blob = self._get_lnprob(q)
        lnpdiff = (self.dim - 1.0) * np.log(zz) + newlnprob - lnprob0
        accept = lnpdiff > np.log(self._random.rand(len(lnpdiff)))
        return q, newlnprob, accept, blob

    # -- sampling ----------------------------------------------------------------
    def sample(self, p0, lnprob0=None, rstate0=None, blobs0=None, iterations=1, thin=1,
               storechain=True):
        """Generator advancing the ensemble; yields (pos, lnprob, rstate[, blobs])
        after every iteration."""
        if rstate0 is not None:
            self.random_state = rstate0
        p = np.array(p0, dtype=np.float64)
        if p.shape != (self.k, self.dim):
            raise ValueError('p0 must have shape ({}, {})'.format(self.k, self.dim))
        halfk = self.k // 2
        lnprob, blobs = lnprob0, blobs0
        if lnprob is None:
            lnprob, blobs = self._get_lnprob(p)
        lnprob = np.array(lnprob, dtype=np.float64)
        if np.any(np.isnan(lnprob)):
            raise ValueError('The initial lnprob was NaN.')
        i0 = self._chain.shape[1]
        if storechain:
            n_keep = int(iterations // thin)
            self._chain = np.concatenate((self._chain, np.zeros((self.k, n_keep, self.dim))), axis=1)
            self._lnprob = np.concatenate((self._lnprob, np.zeros((self.k, n_keep))), axis=1)
        first, second = slice(halfk), slice(halfk, self.k)
        for i in range(int(iterations)):
            self.iterations += 1
            for s0, s1 in ((first, second), (second, first)):
                q, newlnp, acc, blob = self._propose_stretch(p[s0], p[s1], lnprob[s0])
                if np.any(acc):
                    lnprob[s0][acc] = newlnp[acc]
                    p[s0][acc] = q[acc]
                    self.naccepted[s0][acc] += 1
                    if blob is not None and blobs is not None:
                        full = np.arange(self.k)[s0][acc]
                        for j, src in zip(full, np.arange(len(acc))[acc]):
                            blobs[j] = blob[src]
            if storechain and i % thin == 0:
                ind = i0 + int(i // thin)
                self._chain[:, ind, :] = p
                self._lnprob[:, ind] = lnprob
                if blobs is not None:
                    self._blobs.append(list(blobs))
            if blobs is not None:
                yield p, lnprob, self.random_state, blobs
            else:
                yield p, lnprob, self.random_state

    def run_mcmc(self, pos0, N, rstate0=None, lnprob0=None, **kwargs):
        if pos0 is None:
            if self._last_run_mcmc_result is None:
                raise ValueError('Cannot have pos0=None if run_mcmc has never been called.')
            pos0, lnprob0, rstate0 = self._last_run_mcmc_result[:3]
        results = None
        for results in self.sample(pos0, lnprob0, rstate0, iterations=N, **kwargs):
            pass
        self._last_run_mcmc_result = results[:3] if results is not None else None
        return results


def _run_async(fn, *args, **kwargs):
    """Start fn(*args, **kwargs) on a thread; the returned callable joins and gives its result
    (or raises what it raised)."""
    box = {}

    def work():
        try:
            box['value'] = fn(*args, **kwargs)
        except BaseException as exc:            # re-raised by the caller
            box['error'] = exc

    th = threading.Thread(target=work)
    th.start()

    def join():
        th.join()
        if 'error' in box:
            raise box['error']
        return box['value']
    return join


class DeviceEnsembleSampler(EnsembleSampler):
    """The same sampler with the walkers resident on the GPU: proposal, log-posterior
    (priors included), acceptance and chain storage all run on the device
    (`psfmc_stretch_run`); the host only draws the random numbers, from the same
    `RandomState` in the same order as the host loop, so both samplers produce the
    same chain.  `block` iterations are enqueued per library call (nothing is copied
    back in between); `sample()` still yields once per iteration, each time with the
    generator state that follows that iteration's draws, while `iterations`, `naccepted`
    and the stored chain advance a whole block at a time.

    group: a torch.distributed process group (True = the default group) or a
    `parallel.RankGroup`: the walkers of every half-step's proposals are then sharded over the
    ranks (one process per GPU) -- each rank evaluates its contiguous block, the blocks'
    log-posteriors are all-gathered (the one collective of the path) and every rank applies the
    same accept / move, so all ranks hold the same chain, equal to the single-GPU chain bit for
    bit.  Every rank must construct the sampler with the same `random_state` and call `sample`
    with the same arguments.

    Needs a `MultiComponentModel` whose priors all belong to the families the library
    evaluates (uniform, normal, weibull_min, discrete uniform); otherwise use
    `EnsembleSampler(batch_lnpostfn=model.log_posterior_batch)`.
    """

    def __init__(self, nwalkers, model, a=2.0, live_dangerously=False, block=64, accumulate=False,
                 group=None):
        from .parallel import RankGroup, ShardedLogPosterior
        ranks = group if isinstance(group, RankGroup) else None
        if ranks is None and group is not None:
            ranks = RankGroup(None if group is True else group, 'cuda:%d' % model._device)
        self.ranks = ranks if ranks is not None and not ranks.single else None
        evaluate = (ShardedLogPosterior(model, group=self.ranks) if self.ranks is not None
                    else model.log_posterior_batch)
        super(DeviceEnsembleSampler, self).__init__(nwalkers, model.num_params, a=a,
                                                    batch_lnpostfn=evaluate,
                                                    live_dangerously=live_dangerously)
        self.model = model
        self.block = int(block)
        self.accumulate = bool(accumulate)
        if nwalkers > model._max_walkers:
            raise ValueError('model was built for at most {} walkers'.format(model._max_walkers))
        eng = model.engine                # context + layout
        if not hasattr(type(eng), 'stretch_run'):
            # (a model of a FieldSet: its context is shared by the set's fields)
            eng.stretch_run()             # raises NotImplementedError naming the FieldSet
        if model._host_priors:
            raise ValueError('priors {} are evaluated on the host: the device sampler cannot be '
                             'used'.format([p.name for p, _ in model._host_priors]))

    def _run_block_sharded(self, pos, lnprob, z, lz, partner, log_u, nacc, store=True, accumulate=False):
        """One block of iterations with every half-step's proposals sharded over the ranks
        (include/psfmc_hip.h psfmc_stretch_open / _half_eval / _half_accept / _close)."""
        rg, eng = self.ranks, self.model.engine
        torch = rg.torch
        n_iter, half = int(np.shape(z)[0]), self.k // 2
        lo, hi = rg.block(half)
        a_lo, a_hi = rg.block(self.k)
        with torch.cuda.device(rg.device), rg.on_stream():
            stream = rg.stream_ptr()
            send = torch.zeros(rg.slot(half), dtype=torch.float64, device=rg.device)
            eng.stretch_open(pos, lnprob, z, lz, partner, log_u, nacc, store=store)
            for it in range(n_iter):
                for h in range(2):
                    eng.stretch_half_eval(it, h, lo, hi - lo, send.data_ptr(), stream)
                    full = rg.all_gather_blocks(send, half).contiguous()
                    eng.stretch_half_accept(it, h, full.data_ptr(), stream)
                if accumulate:
                    eng.stretch_accumulate(a_lo, a_hi - a_lo, stream)
            torch.cuda.current_stream(rg.device).synchronize()
            return eng.stretch_close(nacc, stream)

    def _state_snapshotter(self):
        """A cheap `get_state()` for the per-iteration snapshots: `RandomState.get_state` costs
        ~50 us (it would double the host time per iteration); the MT19937 key and position are
        copied straight from the bit generator's state block (1 us), the Gaussian cache -- which
        `rand` / `randint` never touch -- is taken from one full call.  Falls back to `get_state`
        if the bit generator does not expose its state the expected way."""
        rs = self._random
        full = rs.get_state()
        try:
            import ctypes
            bg = rs._bit_generator
            if type(bg).__name__ != 'MT19937':
                return rs.get_state
            block = (ctypes.c_uint32 * 625).from_address(bg.ctypes.state_address)
            view = np.frombuffer(block, dtype=np.uint32)
            if not (np.array_equal(view[:624], full[1]) and int(view[624]) == full[2]):
                return rs.get_state
        except Exception:
            return rs.get_state
        tail = tuple(full[3:])

        def snap():
            a = view.copy()
            return ('MT19937', a[:624], int(a[624])) + tail
        return snap

    def _draw(self, n_iter):
        """Random numbers of n_iter iterations in emcee's order (per half-step:
        rand(Ns) -> z, randint(Nc, Ns) -> partner, rand(Ns) -> ln u)."""
        half = self.k // 2
        z = np.empty((n_iter, 2, half))
        partner = np.empty((n_iter, 2, half), dtype=np.int32)
        log_u = np.empty((n_iter, 2, half))
        states = []                     # generator state after each iteration's draws
        snap = self._state_snapshotter()
        for it in range(n_iter):
            for h in range(2):
                z[it, h] = ((self.a - 1.0) * self._random.rand(half) + 1) ** 2.0 / self.a
                partner[it, h] = self._random.randint(half, size=(half,))
                log_u[it, h] = np.log(self._random.rand(half))
            states.append(snap())
        return (z, (self.dim - 1.0) * np.log(z), partner, log_u), states

    def sample(self, p0, lnprob0=None, rstate0=None, blobs0=None, iterations=1, thin=1,
               storechain=True):
        if rstate0 is not None:
            self.random_state = rstate0
        p = np.array(p0, dtype=np.float64)
        if p.shape != (self.k, self.dim):
            raise ValueError('p0 must have shape ({}, {})'.format(self.k, self.dim))
        if np.any(~np.isfinite(p)):
            raise ValueError('At least one parameter value was infinite or NaN.')
        lnprob = None if lnprob0 is None else np.array(lnprob0, dtype=np.float64)
        if lnprob is None:
            lnprob, _ = self._get_lnprob(p)
        if np.any(np.isnan(lnprob)):
            raise ValueError('The initial lnprob was NaN.')
        i0 = self._chain.shape[1]
        if storechain:
            n_keep = int(iterations // thin)
            self._chain = np.concatenate((self._chain, np.zeros((self.k, n_keep, self.dim))), axis=1)
            self._lnprob = np.concatenate((self._lnprob, np.zeros((self.k, n_keep))), axis=1)
        nacc = self.naccepted.astype(np.int64)
        done = 0
        # The next block's random numbers are drawn while the GPU works on the current one
        # (the library call releases the GIL).  The draws stay in emcee's order because the
        # stream is sequential: block k+1 is drawn right after block k, only earlier in time.
        draws, states = self._draw(min(self.block, iterations)) if iterations > 0 else (None, None)
        while done < iterations:
            n = min(self.block, iterations - done)
            n_next = min(self.block, iterations - done - n)
            block_states = states
            if self.ranks is not None:
                # collectives are issued from this thread, in the same order on every rank
                result = self._run_block_sharded(p, lnprob, *draws, nacc, store=True,
                                                 accumulate=self.accumulate)
                draws, states = self._draw(n_next) if n_next > 0 else (None, None)
            else:
                job = _run_async(self.model.engine.stretch_run, p, lnprob, *draws, nacc, store=True,
                                 accumulate=self.accumulate)
                try:
                    draws, states = self._draw(n_next) if n_next > 0 else (None, None)
                finally:
                    result = job()
            p, lnprob, chain, lnchain = result
            if self.accumulate:                 # sharded: this rank summed its block of the walkers
                share = n * self.k if self.ranks is None else n * (lambda b: b[1] - b[0])(self.ranks.block(self.k))
                self.model._device_samples += share
                self.model.accumulated_samples += share
            # per-block bookkeeping in whole-array operations: at a few hundred microseconds
            # per iteration on the GPU, per-iteration numpy calls here were 10 % of the run
            if storechain:
                # kept iterations of this block: (done + i) % thin == 0 -- an arithmetic
                # progression, copied with slices (index arrays made this 1.6 ms per block)
                first = (-done) % thin
                if first < n:
                    count = (n - first + thin - 1) // thin
                    dst = i0 + (done + first) // thin
                    self._chain[:, dst:dst + count, :] = chain[:, first::thin, :]
                    self._lnprob[:, dst:dst + count] = lnchain[:, first::thin]
            # counters advance with the block (the device reports acceptances per block), so
            # `acceptance_fraction` is consistent whenever the consumer looks; the yielded
            # generator state is the one right after iteration j's draws, so that
            # (pos, lnprob, rstate) of any yield resumes the same chain
            self.naccepted = nacc.astype(np.float64)
            self.iterations += n
            for j in range(n):
                yield chain[:, j, :].copy(), lnchain[:, j].copy(), block_states[j]
            done += n


class _FieldChain(EnsembleSampler):
    """One field's share of a `FieldSetSampler`: the emcee-style state (chain, lnprobability,
    acceptance counters, random state) of that field's ensemble -- what `save_database`,
    `check_convergence_autocorr` and the reference's own post-processing read."""

    def __init__(self, nwalkers, model, a=2.0):
        super(_FieldChain, self).__init__(nwalkers, model.num_params, a=a,
                                          batch_lnpostfn=model.log_posterior_batch)
        self.model = model

    _state_snapshotter = DeviceEnsembleSampler._state_snapshotter
    _draw = DeviceEnsembleSampler._draw


class FieldSetSampler(object):
    """The device-resident stretch-move sampler for every field of a `models.FieldSet` at once
    (`psfmc_stretch_run_fields`): each field is its own ensemble of `nwalkers` walkers with its own
    `RandomState` -- exactly the `DeviceEnsembleSampler` of that field alone -- but the half-step
    proposals of all fields are evaluated as ONE batch, so many small ensembles sample at the rate of one
    large one (BASELINE config 5 as a fit).  `fields[f]` is field f's emcee-style sampler object (`.chain`,
    `.lnprobability`, `.acceptance_fraction`, `.random_state`, ...).  A field's chain equals the chain its
    own one-field `DeviceEnsembleSampler` produces from the same start and random state, bit for bit.
    The reference fits one field per process (psfMC/fitting.py:13-113)."""

    def __init__(self, nwalkers, fieldset, a=2.0, block=64, accumulate=False):
        if nwalkers % 2:
            raise ValueError('The number of walkers must be even.')
        if nwalkers * len(fieldset.models) > fieldset.max_walkers:
            raise ValueError('the FieldSet was built for at most {} walkers in all'.format(fieldset.max_walkers))
        self.fieldset = fieldset
        self.k, self.dim = int(nwalkers), fieldset.num_params
        self.block = int(block)
        self.accumulate = bool(accumulate)
        self.fields = [_FieldChain(nwalkers, m, a=a) for m in fieldset.models]

    def reset(self):
        for f in self.fields:
            f.reset()

    def clear_blobs(self):
        pass

    def sample(self, p0, lnprob0=None, iterations=1, thin=1, storechain=True):
        """p0: [F, W, P] (or a list of [W, P]) start positions; lnprob0 likewise or None.  Yields, once per
        iteration, the list over fields of (pos, lnprob, rstate)."""
        ctx = self.fieldset.context
        n_f = len(self.fields)
        p = np.array([np.asarray(q, dtype=np.float64) for q in p0])
        if p.shape != (n_f, self.k, self.dim):
            raise ValueError('p0 must have shape ({}, {}, {})'.format(n_f, self.k, self.dim))
        if np.any(~np.isfinite(p)):
            raise ValueError('At least one parameter value was infinite or NaN.')
        lnprob = None if lnprob0 is None else np.array([np.asarray(q, dtype=np.float64) for q in lnprob0])
        if lnprob is None:
            lnprob = np.array(self.fieldset.log_posterior_batch(list(p)))
        if np.any(np.isnan(lnprob)):
            raise ValueError('The initial lnprob was NaN.')
        i0 = [f._chain.shape[1] for f in self.fields]
        if storechain:
            n_keep = int(iterations // thin)
            for f in self.fields:
                f._chain = np.concatenate((f._chain, np.zeros((self.k, n_keep, self.dim))), axis=1)
                f._lnprob = np.concatenate((f._lnprob, np.zeros((self.k, n_keep))), axis=1)
        nacc = np.ascontiguousarray([f.naccepted.astype(np.int64) for f in self.fields])
        done = 0

        def draw(n):
            per = [f._draw(n) for f in self.fields]            # every field from its own generator
            arrays = [np.ascontiguousarray([d[0][j] for d in per]) for j in range(4)]
            return arrays, [d[1] for d in per]

        draws, states = draw(min(self.block, iterations)) if iterations > 0 else (None, None)
        while done < iterations:
            n = min(self.block, iterations - done)
            n_next = min(self.block, iterations - done - n)
            block_states = states
            job = _run_async(ctx.stretch_run, p, lnprob, *draws, nacc, store=True, accumulate=self.accumulate)
            try:
                draws, states = draw(n_next) if n_next > 0 else (None, None)
            finally:
                p, lnprob, chain, lnchain = job()
            first = (-done) % thin
            for i, f in enumerate(self.fields):
                if self.accumulate:
                    f.model._device_samples += n * self.k
                    f.model.accumulated_samples += n * self.k
                if storechain and first < n:
                    count = (n - first + thin - 1) // thin
                    dst = i0[i] + (done + first) // thin
                    f._chain[:, dst:dst + count, :] = chain[i][:, first::thin, :]
                    f._lnprob[:, dst:dst + count] = lnchain[i][:, first::thin]
                f.naccepted = nacc[i].astype(np.float64)
                f.iterations += n
            for j in range(n):
                yield [(chain[i][:, j, :].copy(), lnchain[i][:, j].copy(), block_states[i][j])
                       for i in range(n_f)]
            done += n
