#!/usr/bin/env python3
"""Child process of tests/test_gpu_rccl.py (and a stand-alone check): ONE rank, backend "nccl" (= RCCL on
ROCm), one pupil-sharded optimisation-style step through the HIP kernels with BOTH collectives of the data-parallel
path really executed on device tensors --

    #1 compute_rms2d(group=...)      fp64 [F,10] moments all-reduce (dist.all_reduce_sum)
    #2 dist.all_reduce_grads(...)    packed fp64 leaf-gradient all-reduce

-- next to the same step without a group.  With one rank a sum over the ranks is the identity, so loss and leaf
gradients must be BITWISE equal.  Prints one JSON line.  Started as a fresh process: the process group is
initialised before anything else touches the GPU."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    from torchoptics_amd import dist as tl_dist
    torch.cuda.set_device(0)
    group = tl_dist.init_group("cuda:0", "nccl", force=True)
    assert group is not None and dist.get_backend(group) == "nccl"
    n_seen = tl_dist.ranks_seen(group, "cuda:0")

    import yaml_free_lenses as L
    import torchoptics_amd as ta
    from torchoptics_amd import ray_tracing as rt

    def step(group):
        lens, specs, leaves = L.build("cooke", "cuda:0")
        tr = ta.RayTracer(mode="circular", n_rays=(64, 64), rel_fields=(0., 0.707, 1.), wavelengths=("C", "d", "F"),
                          default_device="cuda:0")
        start, stop = tl_dist.shard_range(64 * 64, 0, 1)
        xy = rt.circle_index_range(64, 64, start, stop, "cuda:0")
        x, y, cx, cy, ok, back = tr.trace_rays(specs, lens, xy=xy)
        loss = rt.compute_rms2d(x, y, ok, group=group, n_per_field=64 * 64 * 3)
        loss.backward()
        params = [leaves[k] for k in ("c", "t", "nd", "v")]
        if group is not None:
            tl_dist.all_reduce_grads(params, group)
        return loss.detach(), [p.grad.clone() for p in params]

    l0, g0 = step(None)
    l1, g1 = step(group)
    tl_dist.set_collective("allgather")           # the one-hop shape of the same two exchanges
    l2, g2 = step(group)
    tl_dist.set_collective("allreduce")
    torch.cuda.synchronize()
    dist.barrier(group)
    out = dict(backend=dist.get_backend(group), world=dist.get_world_size(group), n_ranks_seen=n_seen,
               loss_plain=float(l0), loss_dist=float(l1), loss_bitwise_equal=bool(torch.equal(l0, l1)),
               grads_bitwise_equal=bool(all(torch.equal(a, b) for a, b in zip(g0, g1))),
               allgather_bitwise_equal=bool(torch.equal(l0, l2) and all(torch.equal(a, b) for a, b in zip(g0, g2))),
               grad_norms=[float(g.norm()) for g in g1])
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
