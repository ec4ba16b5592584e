"""hipGraph replay of a whole eval-mode forward.

A PSMNet forward is ~110 launches of 10-700 us; replaying them from one captured graph takes the
Python/ctypes launch path (3-5 ms of host time per forward) off the critical path, which matters
when eight ranks share one host.  Everything on the eval path is capture-safe: the library
launches on torch's current stream, never allocates or synchronises, and all scratch and
output tensors come from torch's caching allocator (graph-private pool during capture).
"""
import torch


class GraphedForward(object):
    """``g = GraphedForward(model, left, right)``; ``scales, disps = g(left2, right2)``.

    Inputs of the captured shapes are copied into static buffers and the graph is replayed;
    the returned tensors are the graph's static outputs (overwritten by the next call --
    ``clone()`` what must outlive it).  Eval mode / no-grad only."""

    def __init__(self, model, *example_inputs, warmup=3):
        if model.training:
            raise ValueError("GraphedForward captures an eval-mode forward: call model.eval() first")
        if not all(t.is_cuda for t in example_inputs):
            raise RuntimeError("GraphedForward needs CUDA (HIP) tensors: there is no CPU fallback")
        self.model = model
        self.static_in = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(warmup):           # weight packing, BN folding, kernel attributes
                model(*self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = model(*self.static_in)

    def replay(self):
        self.graph.replay()
        return self.static_out

    def __call__(self, *inputs):
        if len(inputs) != len(self.static_in):
            raise ValueError("expected %d inputs" % len(self.static_in))
        for dst, src in zip(self.static_in, inputs):
            if src.shape != dst.shape or src.dtype != dst.dtype:
                raise ValueError("GraphedForward was captured for %s %s, got %s %s"
                                 % (tuple(dst.shape), dst.dtype, tuple(src.shape), src.dtype))
            dst.copy_(src)
        return self.replay()
