"""Same-process A/B of the PSMNet eval forward with host-side options toggled (interleaved rounds):
    s3 on/off, fuse_volume on/off, the S3 kernel's two tilings, and optional per-stage breakdown from the LaunchTimer."""
import sys
import time

sys.path.insert(0, ".")
import torch

from dsmnet_amd import calibrate, costvolume as cv
from dsmnet_amd.models import model_create_by_name

H, W = 384, 1280
torch.manual_seed(0)
dev = "cuda"
m = model_create_by_name("psmnet", 192).to(dev)
g = torch.Generator().manual_seed(1)
l, r = torch.rand(1, 3, H, W, generator=g).to(dev), torch.rand(1, 3, H, W, generator=g).to(dev)
calibrate.calibrate_batchnorm(m, l, r)
calibrate.calibrate_psmnet_heads(m, l, r)
m.eval()
# (s3, fuse_volume, s3_tiling, overlap_heads); s3in stays off
configs = {"s3+fuse": (True, True, 0, False), "s3": (True, False, 0, False), "r01": (False, False, 0, False),
           "tile8x32": (True, True, 1, False), "tile4x32": (True, True, 2, False),
           "overlap": (True, True, 0, True), "nofirst3": (True, True, 0, False)}
GRAPH = "--graph" in sys.argv                      # time hipGraph replays instead of eager launches
sys.argv = [a for a in sys.argv if a != "--graph"]
if len(sys.argv) > 1:
    configs = {k: v for k, v in configs.items() if k in sys.argv[1:]}
res = {k: [] for k in configs}
stages = {}
graphs = {}
from dsmnet_amd.graphs import GraphedForward
with torch.no_grad():
    for rnd in range(6):
        for name, (s3, fuse, tiling, overlap) in configs.items():
            cv.set_option("s3", s3), cv.set_option("fuse_volume", fuse), cv.set_option("s3_tiling", tiling)
            cv.set_option("overlap_heads", overlap)
            cv.set_option("first3", name != "nofirst3")
            if GRAPH:
                if name not in graphs:
                    graphs[name] = GraphedForward(m, l, r)
                run = graphs[name].replay
            else:
                run = lambda: m(l, r)
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                run()
            torch.cuda.synchronize()
            if rnd:
                res[name].append((time.perf_counter() - t0) / 10 * 1e3)
    for name, (s3, fuse, tiling, overlap) in configs.items():
        cv.set_option("s3", s3), cv.set_option("fuse_volume", fuse), cv.set_option("s3_tiling", tiling)
        cv.set_option("overlap_heads", False)
        cv.set_option("first3", name != "nofirst3")
        t = cv.LaunchTimer()
        cv.set_timer(t)
        for _ in range(5):
            m(l, r)
        torch.cuda.synchronize()
        cv.set_timer(None)
        stages[name] = {k: (v["ms"] / 5, v["launches"] / 5) for k, v in t.summary().items()}
for name, t in res.items():
    t = sorted(t)
    print("%-10s %s forward: median %.3f ms  min %.3f ms" % (name, "replayed" if GRAPH else "eager", t[len(t) // 2], t[0]))
names = sorted({k for s in stages.values() for k in s})
for k in names:
    print("%-62s" % k + "  ".join("%s %6.3f ms (%2d)" % (c, stages[c].get(k, (0, 0))[0], stages[c].get(k, (0, 0))[1]) for c in stages))
