"""Timing of a PSMNet training step (BASELINE config #5; DSM_CONV_PRECISION=f16x2|f16|bf16x3|fp32; one MI355X): train-mode forward,
smooth-L1 on the three heads, backward, SGD step.  Prints ms/step and the per-kernel breakdown
of the HIP launches (forward, bwd-data and bwd-weight kernels)."""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
import torch.nn.functional as F
from dsmnet_amd import costvolume as cv
from dsmnet_amd.models import model_create_by_name

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 512)
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
import os
# several ranks (torch.distributed.run): RCCL, or -- DSM_BENCH_REHEARSE=1, ranks sharing one GPU --
# gloo: a rehearsal of the multi-rank control path (flat gradient all-reduce around the captured
# forward + backward), not a measurement
from dsmnet_amd import sharding
rehearse = os.environ.get("DSM_BENCH_REHEARSE") == "1"
rank, world, local_rank = sharding.init_from_env("gloo" if rehearse else None)
if world > 1 and not rehearse:
    torch.cuda.set_device(local_rank)
say = print if rank == 0 else (lambda *a, **k: None)
torch.manual_seed(0)
m = model_create_by_name("psmnet", 192).cuda().train()
for i in (1, 2, 3):
    with torch.no_grad():
        getattr(m, "classif%d" % i)[2].weight.mul_(1e-3)
for pat in [q for q in os.environ.get("DSM_BENCH_FREEZE", "").split(",") if q]:   # attribution runs
    for n, q in m.named_parameters():
        if pat in n:
            q.requires_grad_(False)
torch.manual_seed(100 + rank)                    # every rank its own pairs
left = torch.rand(B, 3, H, W, device="cuda")
right = torch.roll(left, -6, dims=3)
target = torch.full((B, H, W), 6.0, device="cuda")
opt = torch.optim.SGD([q for q in m.parameters() if q.requires_grad], lr=1e-4)


def step():
    with cv.amax_scope(left.device):
        opt.zero_grad(set_to_none=True)
        _, preds = m(left, right)
        loss = sum(w * F.smooth_l1_loss(p, target) for w, p in zip((0.5, 0.7, 1.0), preds))
        loss.backward()
        opt.step()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
timer = cv.LaunchTimer()
cv.set_timer(timer)
N = 5
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(N):
    loss = step()
b.record()
torch.cuda.synchronize()
cv.set_timer(None)
ms = a.elapsed_time(b) / N
say("PSMNet training step %dx%d batch %d D=192, conv_precision %s: %.1f ms/step (%.2f pairs/s), loss %.4f"
      % (H, W, B, cv.get_option("conv_precision"), ms, B * 1e3 / ms, loss.item()))
tot = 0.0
for k, v in sorted(timer.summary().items(), key=lambda kv: -kv[1]["ms"]):
    n = v["launches"]
    tot += v["ms"] / N
    rate = v["work"] / (v["ms"] * 1e-3)
    say("   %-48s x%-4d %8.1f us avg %8.2f ms/step %8.2f %s" % (
        k, n // N, v["ms"] / n * 1e3, v["ms"] / N, rate / 1e12, "TFLOP/s" if ("mfma" in k or "wgrad" in k) else "TB/s"))
# the same step replayed from a hipGraph (dsmnet_amd/graphs.py).  The eager steps above ran on
# the default stream: drop every reference to their autograd graphs first (a stale
# AccumulateGrad node tied to another stream invalidates the capture).
import gc
final_loss = loss.item()
del loss
opt.zero_grad(set_to_none=True)
gc.collect()
from dsmnet_amd import train
from dsmnet_amd.graphs import GraphedTrainStep
batch = torch.cat([left, right, target.unsqueeze(1)], 1)
lossfun = train.losses("supervised", 1, 0)
lossfun.Weight_Adjust_levels(0)
# fused=True: one multi-tensor kernel per update instead of ~4 tiny kernels per parameter tensor
# (capturable foreach Adam: 518 divisions + 145 counter increments of ~5 us each, 3 ms of the step)
gopt = torch.optim.Adam([q for q in m.parameters() if q.requires_grad], lr=1e-4, capturable=True,
                        fused=os.environ.get("DSM_BENCH_ADAM", "fused") == "fused")
gstep = GraphedTrainStep(m, gopt, lossfun, batch)
gstep(batch); torch.cuda.synchronize()
a.record()
for _ in range(N):
    gstep(batch)
b.record(); torch.cuda.synchronize()
ms_g = a.elapsed_time(b) / N
if world > 1:
    import torch.distributed as dist
    dist.barrier()
    t = torch.tensor([ms_g], dtype=torch.float64, device="cpu" if rehearse else "cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ms_g = float(t.item())
say("   hipGraph replay of the whole step (supervised pyramid loss, Adam)%s: %.1f ms/step" % (
    "" if world == 1 else " on %d ranks%s: forward + backward captured, flat gradient all-reduce + fused Adam outside"
    % (world, " SHARING ONE GPU over gloo (rehearsal, not a measurement)" if rehearse else ""), ms_g))
say("   HIP kernels %.1f ms/step, everything else (stock torch: towers, BN, ReLU, adds, optimizer) %.1f ms/step"
      % (tot, ms - tot))
