"""
Walker sharding across the GPUs of one node: one process per GPU, launched by
`python -m torch.distributed.run`, ranks evaluate contiguous walker blocks and
exchange only the per-walker log-posteriors with ONE all-gather (RCCL over xGMI
with the `nccl` backend; `gloo` on CPU for tests and for rehearsals on a box with
fewer GPUs than ranks).  The shared field arrays are replicated per GPU at context
creation, so there is no other data-path collective.  The
reference has no counterpart: psfMC/fitting.py:55 notes that parallel evaluation of
the walkers had to be given up.

  shard_bounds          contiguous block of a rank
  RankGroup             this rank's view of the process group + the device-side all-gather
  ShardedLogPosterior   [W, P] vectors -> [W] log-posteriors, every rank taking its block
                        (host sampler / `pool.map` route)
The device-resident sampler shards each half-step's proposals the same way
(`sampler.DeviceEnsembleSampler(group=...)`).
"""
import numpy as np


def shard_bounds(n_walkers, world_size, rank):
    """Contiguous block [lo, hi) of rank `rank`; blocks differ by at most one."""
    base, extra = divmod(int(n_walkers), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class RankGroup(object):
    """This rank's view of a torch.distributed process group.

    group   a torch.distributed group (None: the default group; no initialised process
            group at all -> a world of one)
    device  torch device of this rank's GPU (None: CPU tensors, for CPU-only tests)
    With the `gloo` backend the gathered tensors are staged through the host (tests,
    rehearsals); with `nccl` (RCCL) they stay on the device.
    shortcut  False: a world of ONE rank still goes through the collectives (the single-GPU
              rehearsal of the RCCL path: tests/test_gpu_multirank.py); default True."""

    def __init__(self, group=None, device=None, shortcut=True):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = group
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        self.device = torch.device(device) if device is not None else torch.device('cpu')
        backend = dist.get_backend(group) if self.active else 'none'
        self.backend = backend
        self.host_staged = self.active and (backend != 'nccl' or self.device.type == 'cpu')
        # a lone rank skips the collectives unless the caller wants them exercised
        self.single = self.world == 1 and (shortcut or not self.active)
        self._index = {}
        self._stream = None

    def _on_device(self):
        """Context manager: this rank's GPU is torch's CURRENT device.  RCCL places the tensors of
        its object collectives (`broadcast_object_list`) and its communicator on the current
        device; a rank that never called `torch.cuda.set_device` would otherwise put them on
        cuda:0 like every other rank (duplicate-GPU error or a hang at the first broadcast)."""
        import contextlib
        if self.device.type != 'cuda':
            return contextlib.nullcontext()
        return self.torch.cuda.device(self.device)

    def on_stream(self):
        """Context manager: torch work and library launches of the sharded paths share ONE real
        (non-NULL) HIP stream.  The library treats a NULL stream argument as "the context's own
        stream", which is not ordered with torch's default stream."""
        import contextlib
        if self.device.type != 'cuda':
            return contextlib.nullcontext()
        if self._stream is None:
            self._stream = self.torch.cuda.Stream(self.device)
            self._stream.wait_stream(self.torch.cuda.current_stream(self.device))
        return self.torch.cuda.stream(self._stream)

    def stream_ptr(self):
        """hipStream_t (int) of the stream `on_stream` selects; call inside `on_stream()`."""
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def block(self, n):
        return shard_bounds(n, self.world, self.rank)

    def slot(self, n):
        return -(-int(n) // self.world)

    def _gather_index(self, n):
        """Positions of the n results inside the [world, slot] gathered buffer."""
        if n not in self._index:
            slot = self.slot(n)
            idx = np.concatenate([r * slot + np.arange(b - a) for r in range(self.world)
                                  for a, b in [shard_bounds(n, self.world, r)]])
            self._index[n] = self.torch.from_numpy(idx.astype(np.int64)).to(self.device)
        return self._index[n]

    def all_gather_blocks(self, send, n):
        """send: [slot(n)] float64 tensor on `self.device` whose first (hi - lo) entries are
        this rank's block -> [n] tensor with every rank's block in place.  One all-gather."""
        torch, dist = self.torch, self.dist
        if self.single:
            return send[:n]
        slot = self.slot(n)
        with self._on_device():
            if self.host_staged:
                recv = torch.empty(slot * self.world, dtype=torch.float64)
                dist.all_gather_into_tensor(recv, send.detach().cpu().contiguous(), group=self.group)
                recv = recv.to(self.device)
            else:
                recv = torch.empty(slot * self.world, dtype=torch.float64, device=self.device)
                dist.all_gather_into_tensor(recv, send, group=self.group)
            return recv.index_select(0, self._gather_index(n))

    def all_reduce_sum_host(self, array):
        """Element-wise sum over ranks of a host float64 array (posterior-image sums: a
        one-off exchange at the end of sampling)."""
        if self.single:
            return array
        torch, dist = self.torch, self.dist
        t = torch.from_numpy(np.ascontiguousarray(array, dtype=np.float64))
        if self.host_staged:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            return t.numpy()
        with self._on_device():
            d = t.to(self.device)
            dist.all_reduce(d, op=dist.ReduceOp.SUM, group=self.group)
            return d.cpu().numpy()

    def broadcast_object(self, obj, src=0):
        if self.single:
            return obj
        box = [obj]
        with self._on_device():
            self.dist.broadcast_object_list(box, src=src, group=self.group)
        return box[0]


class ShardedLogPosterior(object):
    """Evaluate `[W, P]` parameter vectors with every rank taking its block.

    evaluate  callable([w, P] float64 ndarray) -> [w] float64 (a rank-local batched
              evaluator), or a `MultiComponentModel`: then the block is evaluated with the
              walkers resident on the model's GPU (`psfmc_eval_theta_device`) and the blocks
              are gathered as device tensors -- no host copy between evaluation and exchange.
    Every rank must call with the same `theta`; every rank gets the full [W] result.
    With world_size 1 (or no process group) it is a plain call.
    """

    def __init__(self, evaluate, group=None, device=None, shortcut=True):
        self.model = evaluate if hasattr(evaluate, 'log_posterior_batch') else None
        self.evaluate = self.model.log_posterior_batch if self.model is not None else evaluate
        if self.model is not None and device is None:
            device = 'cuda:%d' % self.model._device
        self._group_arg, self._device_arg, self._shortcut = group, device, shortcut
        self._rg = group if isinstance(group, RankGroup) else None

    @property
    def ranks(self):
        if self._rg is None:
            self._rg = RankGroup(self._group_arg, self._device_arg, shortcut=self._shortcut)
        return self._rg

    def evaluate_device(self, theta_dev):
        """The same with the walkers RESIDENT on this rank's GPU: `theta_dev` a contiguous [W, P] float64
        torch tensor on the model's device (the same vectors on every rank) -> [W] float64 tensor on
        that device with every rank's block in place.  Nothing crosses the host: the rank evaluates its
        contiguous block (`psfmc_eval_theta_device`) and the blocks are exchanged by the one all-gather
        (`RankGroup.all_gather_blocks`).  Needs a `MultiComponentModel` whose priors all have a device
        form.  The result is ordered on the stream of `RankGroup.on_stream()`: synchronise the device (or
        that stream) before reading it elsewhere."""
        if self.model is None:
            raise ValueError('evaluate_device needs a MultiComponentModel')
        eng = self.model.engine                                     # creates context + layout
        if self.model._host_priors:
            raise ValueError('evaluate_device needs a model whose priors all have a device form')
        rg = self.ranks
        torch = rg.torch
        n_w = int(theta_dev.shape[0])
        lo, hi = rg.block(n_w)
        with rg.on_stream():
            stream = rg.stream_ptr()
            send = torch.full((max(rg.slot(n_w), 1),), float('nan'), dtype=torch.float64, device=rg.device)
            for a in range(lo, hi, eng.max_walkers):                # larger blocks go through in slices
                b = min(a + eng.max_walkers, hi)
                eng.logpost_theta_device(b - a, theta_dev[a:b].data_ptr(), 0, send[a - lo:].data_ptr(), stream)
            return rg.all_gather_blocks(send, n_w)

    def __call__(self, theta):
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        rg = self.ranks
        if rg.single:
            return self.evaluate(theta)
        torch = rg.torch
        n_w = theta.shape[0]
        lo, hi = rg.block(n_w)
        eng = self.model.engine if self.model is not None else None      # creates context + layout
        with rg.on_stream():
            send = torch.full((rg.slot(n_w),), float('nan'), dtype=torch.float64, device=rg.device)
            if eng is not None and not self.model._host_priors and hi > lo:
                stream = rg.stream_ptr()
                keep = []
                for a in range(lo, hi, eng.max_walkers):            # larger blocks go through in slices
                    b = min(a + eng.max_walkers, hi)
                    th = torch.from_numpy(theta[a:b]).to(rg.device)
                    eng.logpost_theta_device(b - a, th.data_ptr(), 0, send[a - lo:].data_ptr(), stream)
                    keep.append(th)                                 # alive until the gather is enqueued
            elif hi > lo:
                send[:hi - lo] = torch.from_numpy(np.ascontiguousarray(self.evaluate(theta[lo:hi]))).to(rg.device)
            return rg.all_gather_blocks(send, n_w).cpu().numpy()
