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


class GraphedTrainStep(object):
    """One supervised training step -- train-mode forward, pyramid loss, backward, optimizer step
    (``train.train_step`` without its host-side metrics) -- captured once and replayed.

    A PSMNet step is ~3,000 kernel launches, most of them stock torch / MIOpen kernels of ~5 us for
    the train-mode 2-D towers; on an idle host the replay takes about as long as eager launches, and
    what the graph buys is independence from the host when several ranks share it.
    ``step = GraphedTrainStep(model, optim, lossfun, batch)``;
    ``loss, disps = step(next_batch)`` (static tensors).
    The optimizer must be built with ``capturable=True``; build it with ``fused=True`` as well: the
    capturable foreach Adam issues ~4 tiny kernels per parameter tensor (518 divisions and 145
    counter increments per PSMNet step, 3 ms of a 27 ms step at 256x512 -- DESIGN.md section 9).

    Several ranks (``torch.distributed`` initialised, or ``world=N``): the forward and the backward
    are the captured graph -- the ~3,000 launches that would otherwise go through eight Python
    interpreters sharing one host -- with every gradient a view of one flat buffer
    (``sharding.FlatGradients``); after the replay come, eagerly, ONE all-reduce of that buffer
    over RCCL (+ one division) and the fused optimizer step: three or four launches per step outside
    the graph.  (Capturing the collective itself would save those at the price of tying the graph
    to the communicator's state; with 20.9 MB per step and a step of >= 100 ms it buys nothing.)
    A rank whose batch has no ground-truth pixel contributes zero gradients and takes part in the
    collective all the same; the number of ranks that had ground truth rides in the same buffer
    and the optimizer step is skipped on every rank together when it is zero (as
    ``train.train_step`` does).

    An eager backward through the same model before the capture leaves gradient tensors and
    autograd buffers that were allocated OUTSIDE the capture's private pool, on the default
    stream; re-used inside the capture (``.grad`` accumulation, a freed block handed back by the
    caching allocator with a pending cross-stream event) they invalidate it, and the HIP runtime
    aborted in ``capture_end`` instead of raising.  The constructor therefore drops every
    gradient, collects garbage (dead autograd graphs release their buffers) and synchronises
    the device before the side-stream warm-up and again before the capture, so that the capture
    starts from allocations of its own (tests/test_models_gpu.py:
    ``test_graphed_train_step_after_an_eager_step_on_the_same_model``)."""

    def __init__(self, model, optim, lossfun, example_batch, warmup=3, world=None, flat_gradients=None):
        """``flat_gradients``: force (True) the multi-rank form -- forward + backward captured, flat
        gradient exchange and optimizer step outside -- also in a single process (tests); default:
        exactly when there are several ranks."""
        import torch.distributed as dist
        if world is None:
            world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.world = int(world)
        self.split = bool(flat_gradients) if flat_gradients is not None else self.world > 1
        if not example_batch.is_cuda or example_batch.shape[1] < 7:
            raise ValueError("GraphedTrainStep needs a CUDA (HIP) batch (B, >=7, H, W): imL | imR | dispL")
        if not all(g.get("capturable", False) for g in optim.param_groups):
            raise ValueError("GraphedTrainStep: build the optimizer with capturable=True")
        self.model, self.optim, self.lossfun = model, optim, lossfun
        lossfun.capturable = True
        model.train()
        self.batch = example_batch[:, :7].clone()
        self.flatgrads = None
        self._quiesce()
        if self.split:
            from . import sharding
            self.flatgrads = sharding.FlatGradients(model.parameters(), n_extra=1)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        self._quiesce()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.disps = self._step()

    def _quiesce(self):
        """No gradient, no dead autograd graph, no kernel in flight: the state a capture (or the
        warm-up on the side stream) must start from."""
        import gc
        if self.flatgrads is None:
            self.optim.zero_grad(set_to_none=True)
            for p in self.model.parameters():
                p.grad = None
        else:
            self.flatgrads.attach()
        gc.collect()
        torch.cuda.synchronize()

    def _step(self):
        from . import costvolume as cv
        with cv.amax_scope(self.batch.device):      # fp16 convolution modes: one arena of maxima per step
            return self._step_body()

    def _step_body(self):
        b = self.batch
        if self.flatgrads is not None:
            # several ranks: only forward + backward are captured; gradients accumulate into the
            # views of the flat buffer, whose last float counts "this rank had ground truth"
            self.flatgrads.zero()
            scales, disps = self.model(b[:, :3], b[:, 3:6])
            loss = self.lossfun({"disp_gt": b[:, 6:7], "disps": disps, "scale_disps": scales,
                                 "flag_smooth": True})
            loss.backward()
            self.flatgrads.extra.copy_((b[:, 6:7] > 0).any().to(self.flatgrads.flat.dtype).reshape(1))
            return loss.detach(), [d.detach() for d in disps]
        self.optim.zero_grad(set_to_none=True)
        scales, disps = self.model(b[:, :3], b[:, 3:6])
        loss = self.lossfun({"disp_gt": b[:, 6:7], "disps": disps, "scale_disps": scales,
                             "flag_smooth": True})
        loss.backward()
        self.optim.step()
        return loss.detach(), [d.detach() for d in disps]

    def _exchange_and_update(self):
        """The part of a multi-rank step that stays outside the graph."""
        n_gt = self.flatgrads.allreduce(self.world)
        if float(n_gt) > 0:                      # (a host read: one scalar per step)
            self.optim.step()

    def __call__(self, batch):
        if batch.shape[0] != self.batch.shape[0] or batch.shape[2:] != self.batch.shape[2:]:
            raise ValueError("GraphedTrainStep was captured for %s, got %s"
                             % (tuple(self.batch.shape), tuple(batch.shape)))
        self.batch.copy_(batch[:, :7])
        self.graph.replay()
        if self.flatgrads is not None:
            self._exchange_and_update()
        return self.loss, self.disps
