"""Same-process A/B of two builds of the library on the transposed 3-D convolution or (--cout 1) the heads' 32 -> 1 convolution (interleaved rounds;
MI355X devices differ by up to 10 % on MFMA-dense kernels, so builds are only ever ranked inside one
run).    python3 scripts/ab_deconv.py libA.so libB.so [--cout 32|64] [--out f32|both] [--res]"""
import argparse
import ctypes
import sys

sys.path.insert(0, ".")
import torch

from dsmnet_amd import _lib, costvolume as cv

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--cout", type=int, default=32)
ap.add_argument("--out", default="both")
ap.add_argument("--res", action="store_true")
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--conv", action="store_true", help="a stride-1 3x3x3 convolution cout -> cout at 24x48x160 instead")
args = ap.parse_args()
_lib.load()
dev = "cuda"
torch.manual_seed(0)
CL = torch.channels_last_3d
head = args.cout == 1                        # --cout 1: the classifier heads' 32 -> 1 convolution instead
plain = args.conv
cin = 32 if head else (args.cout if plain else 64)
din = (48, 96, 320) if head else ((24, 48, 160) if (args.cout == 32 or plain) else (12, 24, 80))
dout = din if (head or plain) else tuple(2 * d for d in din)
x = torch.randn(1, cin, *din, device=dev).contiguous(memory_format=CL)
if head:
    w = torch.randn(1, cin, 3, 3, 3, device=dev) * 0.05
elif plain:
    w = torch.randn(args.cout, cin, 3, 3, 3, device=dev) * 0.05
else:
    w = torch.randn(cin, args.cout, 3, 3, 3, device=dev) * 0.05
packed = cv.pack_conv3d_weight(w, not (head or plain))
res = torch.randn(1, args.cout, *dout, device=dev).contiguous(memory_format=CL) if args.res else None
y = torch.empty(1, args.cout, *dout, device=dev).contiguous(memory_format=CL)
ys3 = torch.empty(y.numel() * 6, device=dev, dtype=torch.uint8)
a = _lib.Conv3dArgs()
a.x, a.w_packed = x.data_ptr(), packed.data_ptr()
a.residual = None if res is None else res.data_ptr()
a.y = y.data_ptr()
a.y_s3 = ys3.data_ptr() if args.out == "both" else None
a.B, a.Cin, a.Cout = 1, cin, args.cout
a.Di, a.Hi, a.Wi = din
a.Do, a.Ho, a.Wo = dout
a.Dr, a.Hr, a.Wr = dout
a.stride, a.transposed, a.relu = (1, 0, 0) if head else ((1, 0, 1) if plain else (2, 1, 1))
if head or (plain and args.cout != 32):
    a.y_s3 = None
fns = []
for path in args.libs:
    lib = ctypes.CDLL(path)
    f = lib.dsm_conv3d_fwd
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.POINTER(_lib.Conv3dArgs), ctypes.c_void_p]
    fns.append(f)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
outs, times = [], [[] for _ in fns]
for rnd in range(args.rounds + 1):
    for i, f in enumerate(fns):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            rc = f(ctypes.byref(a), stream)
            assert rc == 0, rc
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            times[i].append(e0.elapsed_time(e1) / 5 * 1e3)
        else:
            outs.append(y.clone())
flops = 54.0 * cin * max(args.cout, 1) * din[0] * din[1] * din[2]
for path, t in zip(args.libs, times):
    t = sorted(t)
    print("%-40s median %7.1f us  min %7.1f us  %6.1f TF/s" % (path, t[len(t) // 2], t[0], flops / t[len(t) // 2] / 1e6))
if len(outs) > 1:
    print("max |A - B| =", float((outs[0] - outs[1]).abs().max()))
