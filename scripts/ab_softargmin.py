"""Same-process A/B of two builds of the library on the PSMNet head's fused upsample + soft-argmin
(interleaved rounds).   python3 scripts/ab_softargmin.py libA.so libB.so"""
import ctypes
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch

from dsmnet_amd import _lib

_lib.load()
torch.manual_seed(0)
cost = torch.randn(1, 48, 96, 320, device="cuda") * 2.0
disp = [torch.empty(1, 384, 1280, device="cuda") for _ in sys.argv[1:]]
fns = []
for path in sys.argv[1:]:
    f = ctypes.CDLL(path).dsm_soft_argmin_fwd
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 10 + [ctypes.c_void_p]
    fns.append(f)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
times = [[] for _ in fns]
for rnd in range(7):
    for i, f in enumerate(fns):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            rc = f(cost.data_ptr(), disp[i].data_ptr(), None, 1, 48, 96, 320, 192, 384, 1280, 0, 0, 0, stream)
            assert rc == 0, rc
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            times[i].append(e0.elapsed_time(e1) / 10 * 1e3)
for path, t in zip(sys.argv[1:], times):
    t = sorted(t)
    print("%-40s median %7.1f us  min %7.1f us" % (path, t[len(t) // 2], t[0]))
if len(disp) > 1:
    print("max |A - B| = %.3e px" % float((disp[0] - disp[1]).abs().max()))
