"""
Multi-GPU data parallelism over the pupil dimension (one process per GPU, RCCL over xGMI).

Rays never interact inside the trace; the only coupling is the per-field spot statistics.
So every rank traces its own contiguous slice of the pupil grid and the job needs exactly
two tiny exchanges per step:

  #1 forward : sum the [F, 10] fp64 spot moments          (all_reduce_sum, differentiable)
  #2 backward: sum the parameter gradients of the leaves (all_reduce_grads)

Both messages are < 2 KB, i.e. latency-bound: one collective each, launched on the compute
stream, no host synchronisation in between.  The loss is REPLICATED (every rank evaluates the
same closed form on the same summed moments), so the backward of #1 is the identity.
"""
from __future__ import annotations

import os
from typing import Iterable, Tuple

import torch
import torch.distributed as dist


def init_group(device, backend: str = "nccl", force: bool = False):
    """One process per GPU: join the job's process group and return it (None for a single process).

    RANK / WORLD_SIZE / MASTER_* come from the launcher (torch.distributed.run).  `backend` "nccl" is RCCL
    on ROCm.  `force=True` builds the group even for WORLD_SIZE=1 (a 1-rank RCCL communicator on this GPU):
    every collective of the sharded path then really executes, which is how the path is exercised on a
    one-GPU box."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and not force:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        # a port of our own choosing only works when nobody else has to find it: with several ranks each would
        # bind a different one and the rendezvous would hang until its timeout
        if world > 1:
            raise RuntimeError("WORLD_SIZE > 1 but MASTER_PORT is unset: start the ranks with torch.distributed.run "
                               "(or dist.spawn_local_ranks), which hands every rank the same rendezvous port")
        os.environ["MASTER_PORT"] = str(free_port())
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(backend)
    return dist.group.WORLD


def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_local_ranks(script: str, argv, n_ranks: int, timeout=None) -> int:
    """Start `n_ranks` ranks of `script argv...` on this node with torch.distributed.run (one process per GPU,
    rendezvous on 127.0.0.1) as a CHILD process and return its exit code; stdout / stderr are inherited, so rank
    0's report goes straight through.  Must be called before the calling process has touched the GPU: a process
    that has initialised HIP must never be replaced or forked into GPU work on this platform, which is why the
    ranks are a fresh process tree and the caller only waits for them."""
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    env["MASTER_ADDR"] = "127.0.0.1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it here
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_ranks)}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    return subprocess.run(cmd, env=env, timeout=timeout).returncode


def ranks_seen(group, device) -> int:
    """Sum of a one over the ranks of `group`, computed by the collective itself on `device`."""
    one = torch.ones(1, dtype=torch.float64, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM, group=group)
    return int(round(one.item()))


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) slice of n pupil points owned by `rank` (sizes differ by <= 1)."""
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


# How the two < 2 KB exchanges are done.  "allreduce": one RCCL all-reduce (default).  "allgather": the shape
# SURVEY 8e prefers on a fully connected xGMI node -- every rank sends its partial vector to every peer in one hop
# (all-gather), then sums the N vectors locally in rank order: latency-optimal for tiny messages, and the result is
# bitwise identical on every rank and from run to run by construction, whatever the collective's internal algorithm.
_collective = os.environ.get("TORCHOPTICS_AMD_COLLECTIVE", "allreduce")


def set_collective(name: str) -> None:
    global _collective
    if name not in ("allreduce", "allgather"):
        raise ValueError("collective must be 'allreduce' or 'allgather'")
    _collective = name


def get_collective() -> str:
    return _collective


def _sum_over_ranks(t: torch.Tensor, group, inplace: bool = False) -> torch.Tensor:
    """Sum of `t` over the ranks of `group`: a new tensor, or `t` itself overwritten when `inplace`."""
    if _collective == "allgather":
        world = dist.get_world_size(group)
        buf = t.new_empty((world,) + tuple(t.shape))
        dist.all_gather(list(buf.unbind(0)), t.contiguous(), group=group)      # views of one buffer, rank order
        return torch.sum(buf, dim=0, out=t) if inplace else buf.sum(dim=0)
    out = t if inplace else t.clone()
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    return out


class _AllReduceSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, group):
        return _sum_over_ranks(t, group)

    @staticmethod
    def backward(ctx, g):
        # replicated loss: d(loss_r)/d(local moments) = d(loss_r)/d(summed moments)
        return g, None


def all_reduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    """Differentiable sum over the ranks of `group` (collective #1)."""
    return _AllReduceSum.apply(t, group)


def all_reduce_grads(params: Iterable[torch.Tensor], group=None) -> None:
    """Sum the .grad of the (replicated) leaves over the ranks in ONE collective (#2).

    The gradients are packed into a single fp64 buffer so the sum is done once, in fp64, and
    every rank ends with bitwise identical values.
    """
    params = [p for p in params if p.grad is not None]
    if not params:
        return
    grads = [p.grad for p in params]
    # three launches + one collective, whatever the number of leaves: widen into one fp64 buffer (a single multi-tensor
    # copy), sum over the ranks in place, narrow back into the .grad tensors (a single multi-tensor copy)
    flat = torch.empty(sum(g.numel() for g in grads), dtype=torch.float64, device=grads[0].device)
    views, off = [], 0
    for g in grads:
        n = g.numel()
        views.append(flat[off:off + n].view(g.shape))
        off += n
    torch._foreach_copy_(views, grads)
    _sum_over_ranks(flat, group, inplace=True)
    torch._foreach_copy_(grads, views)
