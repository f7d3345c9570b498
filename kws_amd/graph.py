"""A training (or inference) step captured once in a HIP graph and replayed.

The operator behind ``FastGRNNCUDA`` makes no host-side decision that depends on data, allocates through torch's
allocator only and launches on the current stream, so forward + autograd backward of a whole model capture as they
are (``torch.cuda.CUDAGraph`` is a hipGraph on ROCm).  A replay costs the host about ten microseconds instead of the
130-270 us of the eager step, which takes the host out of the step time on a slow or busy core.
Shapes, flags and tensor addresses are frozen at capture: feed new batches by copying into the captured input
tensors (``x.copy_(batch)``), read results from the captured outputs / ``.grad`` tensors.
"""
import torch


class GraphedStep:
    """``GraphedStep(fn)`` runs ``fn()`` a few times on a side stream (plan and workspace caches, lazy library
    state), captures one call, and replays it on every ``__call__``.  ``fn`` typically resets ``.grad`` to None and
    runs forward + backward; the gradient tensors allocated during capture stay alive and are overwritten in place
    by each replay.  ``outputs`` holds whatever ``fn`` returned during capture (static tensors)."""

    def __init__(self, fn, warmup=3):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs a GPU (HIP graph capture)")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = fn()

    def __call__(self):
        self.graph.replay()
        return self.outputs
