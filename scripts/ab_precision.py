"""Per-layer timing of the PSMNet forward's convolution shapes (384x1280, D = 192) in every
precision mode, in one process: `python scripts/ab_precision.py [--modes bf16x3,f16x2,f16]`.
HIP events around REP back-to-back launches of one layer (inputs resident, L2-warm weights)."""
import argparse
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from dsmnet_amd import costvolume as cv, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--modes", default="bf16x3,f16x2,f16")
ap.add_argument("--rep", type=int, default=30)
ap.add_argument("--only", default="")
ap.add_argument("--res", action="store_true", help="with a skip tensor of the output's shape (residual add in the epilogue)")
ap.add_argument("--graph", action="store_true", help="REP launches captured into one hipGraph and replayed: GPU time per launch (small kernels are host-bound when launched eagerly)")
ap.add_argument("--flags", default="0,%d" % _lib.DSM_CONV_NO_NSPLIT, help="conv_flags variants, e.g. 0,4,16 (16 = 4-row tiles)")
args = ap.parse_args()

# name, kind, cin, cout, stride, transposed, dil, shape (B, [D,] H, W)
LAYERS = [
    ("2d 32->32 192x640 x8", "2d", 32, 32, 1, False, 1, (2, 192, 640)),
    ("2d 64->64 96x320 x31", "2d", 64, 64, 1, False, 1, (2, 96, 320)),
    ("bb 64->64->64 96x320 x15", "bb", 64, 64, 1, False, 1, (2, 96, 320)),
    ("bb 32->32->32 192x640 x3", "bb", 32, 32, 1, False, 1, (2, 192, 640)),
    ("2d 128->128 96x320 x7", "2d", 128, 128, 1, False, 1, (2, 96, 320)),
    ("2d 128->128 dil2 x6", "2d", 128, 128, 1, False, 2, (2, 96, 320)),
    ("2d 320->128 lastconv", "2d", 320, 128, 1, False, 1, (2, 96, 320)),
    ("3d 64->32 48x96x320", "3d", 64, 32, 1, False, 1, (1, 48, 96, 320)),
    ("3d 32->32 48x96x320 x6", "3d", 32, 32, 1, False, 1, (1, 48, 96, 320)),
    ("3d 32->64 s2 x3", "3d", 32, 64, 2, False, 1, (1, 48, 96, 320)),
    ("3d 64->64 24x48x160 x3", "3d", 64, 64, 1, False, 1, (1, 24, 48, 160)),
    ("3d 64->64 s2 x3", "3d", 64, 64, 2, False, 1, (1, 24, 48, 160)),
    ("3d 64->64 12x24x80 x3", "3d", 64, 64, 1, False, 1, (1, 12, 24, 80)),
    ("deconv 64->64 12x24x80 x3", "3d", 64, 64, 2, True, 1, (1, 12, 24, 80)),
    ("deconv 64->32 24x48x160 x3", "3d", 64, 32, 2, True, 1, (1, 24, 48, 160)),
]


def bench(fn, rep):
    for _ in range(3):
        fn()
    if args.graph:
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(rep):
                fn()
        g.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(5):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / (5 * rep) * 1e3
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(rep):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / rep * 1e3


variants = [(m, int(f), " f=%s" % f if int(f) else "") for m in args.modes.split(",") for f in args.flags.split(",")]
print("%-30s" % "layer" + "".join("%16s" % (m + t) for m, f, t in variants))
for name, kind, cin, cout, stride, tr, dil, shape in LAYERS:
    if args.only and args.only not in name:
        continue
    torch.manual_seed(0)
    x = torch.randn(shape[0], cin, *shape[1:], device="cuda").relu()
    if kind == "bb":
        if "f16" not in args.modes:
            continue
        x = x.contiguous(memory_format=torch.channels_last)
        w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        packed, packed2 = cv.pack_conv2d_weight(w), cv.pack_conv2d_weight(w.flip(0))
        run = lambda res=None: cv.basicblock2d(x, packed, None, None, packed2, None, None)
    elif kind == "2d":
        x = x.contiguous(memory_format=torch.channels_last)
        w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        packed = cv.pack_conv2d_weight(w)
        run = lambda res=None: cv.conv2d_block(x, packed, cout, residual=res, relu=1, dilation=dil)
    else:
        x = x.contiguous(memory_format=torch.channels_last_3d)
        w = torch.randn(*((cin, cout) if tr else (cout, cin)), 3, 3, 3, device="cuda") * 0.05
        packed = cv.pack_conv3d_weight(w, tr)
        run = lambda res=None: cv.conv3d_block(x, packed, cout, residual=res, stride=stride, transposed=tr, relu=1)
    row = "%-30s" % name
    for mode, flags, _ in variants:
        if kind == "bb" and mode not in ("f16x2", "f16"):
            row += "%16s" % "-"
            continue
        o1, o2 = cv.set_option("conv_precision", mode), cv.set_option("conv_flags", flags)
        with cv.amax_scope(x.device):
            if cv.needs_amax():
                cv.absmax(x)
            fn = run
            if args.res:
                skip = torch.randn_like(run())
                fn = lambda: run(skip)
            row += "%16.1f" % bench(fn, args.rep)
        cv.set_option("conv_precision", o1), cv.set_option("conv_flags", o2)
    print(row, flush=True)
