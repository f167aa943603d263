"""
Walker sharding across the GPUs of one node: one process per GPU, launched by
`python -m torch.distributed.run`, ranks evaluate contiguous walker blocks and
exchange only the per-walker log-posteriors with ONE all-gather (RCCL over xGMI
with the `nccl` backend; `gloo` on CPU for tests).  The shared field arrays are
replicated per GPU at context creation, so there is no other data-path
collective.
"""
import numpy as np


def shard_bounds(n_walkers, world_size, rank):
    """Contiguous block [lo, hi) of rank `rank`; blocks differ by at most one."""
    base, extra = divmod(int(n_walkers), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class ShardedLogPosterior(object):
    """Evaluate `[W, P]` parameter vectors with every rank taking its block.

    evaluate  callable([w, P] float64 ndarray) -> [w] float64 (the rank-local
              batched evaluator, e.g. `model.log_posterior_batch`)
    Every rank must call with the same `theta`; every rank gets the full [W]
    result.  With world_size 1 (or no process group) it is a plain call.
    """

    def __init__(self, evaluate, group=None, device=None):
        self.evaluate = evaluate
        self.group = group
        self.device = device

    def __call__(self, theta):
        import torch
        import torch.distributed as dist
        theta = np.asarray(theta, dtype=np.float64)
        if not (dist.is_available() and dist.is_initialized()):
            return self.evaluate(theta)
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        if world == 1:
            return self.evaluate(theta)
        n_w = theta.shape[0]
        lo, hi = shard_bounds(n_w, world, rank)
        mine = self.evaluate(theta[lo:hi]) if hi > lo else np.zeros(0)
        # equal-sized slots so a single all_gather_into_tensor does the exchange
        slot = -(-n_w // world)
        dev = self.device if self.device is not None else 'cpu'
        send = torch.full((slot,), float('nan'), dtype=torch.float64, device=dev)
        send[:hi - lo] = torch.from_numpy(np.ascontiguousarray(mine)).to(dev)
        recv = torch.empty(slot * world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(recv, send, group=self.group)
        recv = recv.cpu().numpy().reshape(world, slot)
        out = np.empty(n_w)
        for r in range(world):
            a, b = shard_bounds(n_w, world, r)
            out[a:b] = recv[r, :b - a]
        return out
