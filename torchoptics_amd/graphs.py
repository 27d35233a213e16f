"""
Whole-step HIP graphs: record one optimisation / loss step (host chain, both trace kernels, autograd, optimiser) once
and replay it -- what makes small workloads GPU-bound instead of launch-bound.

Two rules come with capturing a step that calls `.backward()` on ROCm, both learnt the hard way (round-1 segfault in
hipStreamEndCapture):

* every autograd node the captured backward touches must have been created ON THE CAPTURE STREAM.  The engine runs a
  leaf's AccumulateGrad node on the stream that node was created on; one left over from an eager step on the default
  stream makes the engine synchronise the capturing stream with the default stream, which ends the capture with a
  segfault.  So: use leaves that have not been through an eager backward (`fresh_leaves`), keep no autograd graph
  alive across steps (re-build `Lens(...)` from the bare leaves inside the step), and warm up on the capture stream;
* nothing inside the step may copy from the host (`torch.tensor(...)`, `.to(device)` of a CPU tensor, `.item()`):
  the package's own constants are cached device tensors for that reason.
"""
import torch

__all__ = ["capture_step", "fresh_leaves"]


def fresh_leaves(*tensors):
    """Detached clones with requires_grad=True: leaves whose AccumulateGrad nodes do not exist yet."""
    out = tuple(t.detach().clone().requires_grad_(True) for t in tensors)
    return out[0] if len(out) == 1 else out


def capture_step(step, device="cuda", warmup=3):
    """Warm `step()` up on a side stream, record it into a HIP graph on that stream, return (graph, result).

    `step` must be self-contained (zero / drop the gradients it produces, build its autograd graph from bare leaves,
    call backward, optionally the optimiser) and return the tensor(s) to read after each `graph.replay()`; the
    returned `result` is the static output of the recorded step.  Update inputs in place (`leaf.copy_(new)`) between
    replays."""
    device = torch.device(device)
    cap = torch.cuda.Stream(device)
    cap.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(cap):
        for _ in range(max(1, warmup)):
            step()
    torch.cuda.current_stream(device).wait_stream(cap)
    torch.cuda.synchronize(device)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=cap):
        result = step()
    torch.cuda.synchronize(device)
    return graph, result
