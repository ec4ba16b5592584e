"""Same-process A/B of plan choices (dsm_conv3d_args.flags: tile height, persistent grid size) on one
convolution shape.   python3 scripts/ab_conv2d_flags.py [cin cout H W B [D]]   (D > 1: a 3-D layer)"""
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch

from dsmnet_amd import _lib, costvolume as cv

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
cin, cout, H, W, B = (int(v) for v in argv[:5]) if len(argv) > 4 else (64, 64, 96, 320, 2)
D = int(argv[5]) if len(argv) > 5 else 1
_lib.load()
torch.manual_seed(0)
if D > 1:
    x = torch.randn(B, cin, D, H, W, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    packed = cv.pack_conv3d_weight(torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05, False)
    run = lambda: cv.conv3d_block(x, packed, cout, relu=1)
else:
    x = torch.randn(B, cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    packed = cv.pack_conv2d_weight(torch.randn(cout, cin, 3, 3, device="cuda") * 0.05)
    run = lambda: cv.conv2d_block(x, packed, cout, relu=1)
variants = {"default": 0, "no N-split": _lib.DSM_CONV_NO_NSPLIT}
for tm in (() if "--nsplit" in sys.argv else (1, 2, 4)):
    for blocks in (0, 512, 768):
        variants["TM=%d,blocks=%d" % (tm, blocks)] = (tm << _lib.DSM_CONV_TM_SHIFT) | (blocks << _lib.DSM_CONV_BLOCKS_SHIFT)
times = {k: [] for k in variants}
ref = None
for rnd in range(7):
    for name, flags in variants.items():
        cv.set_option("conv_flags", flags)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y = run()
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            times[name].append(e0.elapsed_time(e1) / 10 * 1e3)
        elif ref is None:
            ref = y.clone()
        else:
            assert float((y - ref).abs().max()) == 0.0, name
cv.set_option("conv_flags", 0)
for name, t in times.items():
    t = sorted(t)
    print("%-22s median %7.1f us  min %7.1f us" % (name, t[len(t) // 2], t[0]))
