"""Run each hot-path kernel alone, N times, at the PSMNet 384x1280 shapes -- the target of
the rocprofv3 --pmc passes (profiles/) and of quick A/B timing.

    python3 scripts/bench_kernels.py [kernel ...] [--iters N]
kernels: s3conv32 s3conv64in s3conv64virt volume_s3 volume volume_ncdhw conv32 conv64in conv_l1 conv_s2 deconv6 deconv5 cout1 softargmin corr
"""
import argparse
import sys

sys.path.insert(0, ".")
import torch

from dsmnet_amd import costvolume as cv

ap = argparse.ArgumentParser()
ap.add_argument("kernels", nargs="*")
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
dev = "cuda"
torch.manual_seed(0)
CL = torch.channels_last_3d


L0, L1, L2 = (48, 96, 320), (24, 48, 160), (12, 24, 80)


def vol(c, d, h, w):
    return torch.randn(1, c, d, h, w, device=dev).contiguous(memory_format=CL)


def conv_case(cin, cout, dims, stride=1, transposed=False, relu=1, res=False):
    x = vol(cin, *dims)
    wshape = (cin, cout, 3, 3, 3) if transposed else (cout, cin, 3, 3, 3)
    w = torch.randn(*wshape, device=dev) * (2.0 / (27 * cout)) ** 0.5
    packed = cv.pack_conv3d_weight(w, transposed)
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
    r = None
    if res:
        r = vol(cout, *cv.conv3d_out_size(dims, stride, transposed))
    flops = 54.0 * cin * cout * (dims[0] * dims[1] * dims[2] if transposed else
                                 (lambda o: o[0] * o[1] * o[2])(cv.conv3d_out_size(dims, stride, False)))
    return (lambda: cv.conv3d_block(x, packed, cout, sc, sh, r, stride, transposed, relu)), flops, "FLOP"


def s3_case(cin, dims, virtual=False, out="both", res=True):
    w = torch.randn(32, cin, 3, 3, 3, device=dev) * (2.0 / (27 * 32)) ** 0.5
    packed = cv.pack_conv3d_s3_weight(w)
    sc, sh = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.1
    if virtual:
        a, b = torch.randn(1, cin // 2, *dims[1:], device=dev), torch.randn(1, cin // 2, *dims[1:], device=dev)
        xs = cv.concat_volume_s3(a, b, dims[0], True, materialise=False)
    else:
        xs = cv.s3_from_tensor(vol(cin, *dims))
    r = vol(32, *dims) if res else None
    flops = 54.0 * cin * 32 * dims[0] * dims[1] * dims[2]
    return (lambda: cv.conv3d_s3_block(xs, packed, sc, sh, r, relu=1, out=out)), flops, "FLOP"


fl, fr = torch.randn(1, 32, 96, 320, device=dev), torch.randn(1, 32, 96, 320, device=dev)
cost = torch.randn(1, 1, 48, 96, 320, device=dev) * 2
cl, cr = torch.randn(1, 128, 96, 320, device=dev), torch.randn(1, 128, 96, 320, device=dev)
CASES = {
    "s3conv32": lambda: s3_case(32, L0),
    "s3conv32_f32out": lambda: s3_case(32, L0, out="f32", res=False),
    "s3conv64in": lambda: s3_case(64, L0, out="s3", res=False),
    "s3conv64virt": lambda: s3_case(64, L0, virtual=True, out="s3", res=False),
    "volume_s3": lambda: ((lambda: cv.concat_volume_s3(fl, fr, 48, True)), 385351680.0, "B"),
    "volume": lambda: ((lambda: cv.concat_volume(fl, fr, 48, True, True)), 385351680.0, "B"),
    "volume_ncdhw": lambda: ((lambda: cv.concat_volume(fl, fr, 48, True, False)), 385351680.0, "B"),
    "conv32": lambda: conv_case(32, 32, L0),
    "conv64in": lambda: conv_case(64, 32, L0),
    "conv_l1": lambda: conv_case(64, 64, L1, res=True),
    "conv_l2": lambda: conv_case(64, 64, L2),
    "conv_s2": lambda: conv_case(32, 64, L0, stride=2),
    "conv_s2b": lambda: conv_case(64, 64, L1, stride=2),
    "deconv6": lambda: conv_case(64, 32, L1, stride=2, transposed=True, relu=0, res=True),
    "deconv5": lambda: conv_case(64, 64, L2, stride=2, transposed=True, res=True),
    "cout1": lambda: ((lambda f: (f[0], 4.0 * (32 * 48 * 96 * 320 + 48 * 96 * 320), "B"))(conv_case(32, 1, L0, relu=0))),
    "softargmin": lambda: ((lambda: cv.soft_argmin(cost, (192, 384, 1280))), 4.0 * (48 * 96 * 320 + 384 * 1280), "B"),
    "corr": lambda: ((lambda: cv.corr1d(cl, cr, 41)), 4.0 * (2 * 128 * 96 * 320 + 41 * 96 * 320), "B"),
}
for name in (args.kernels or list(CASES)):
    fn, work, unit = CASES[name]()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / args.iters * 1e3
    rate = work / (us * 1e-6)
    print("%-13s %9.1f us   %8.2f %s" % (name, us, rate / 1e12, "TFLOP/s" if unit == "FLOP" else "TB/s"),
          flush=True)
