"""Diagnostic: phase shares of the bf16x3 kernel on a small 2-D layer (stamped build:
python dsmnet_amd/csrc/build.py --stamps)."""
import ctypes, os, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from dsmnet_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libdsmnet_hip_stamps.so")
from dsmnet_amd import costvolume as cv
lib = _lib.load()
lib.dsm_debug_read_stamps.restype = ctypes.c_int
lib.dsm_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * (8 * 1024))()
NAMES = ["barrier", "-", "-", "chunk setup", "multiply", "epilogue", "-", "first fragment reads"]
for cin, cout, hw in ((64, 64, (96, 320)), (128, 128, (96, 320)), (32, 32, (192, 640))):
    x = torch.randn(2, cin, *hw, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    packed = cv.pack_conv2d_weight(w)
    f = lambda: cv.conv2d_block(x, packed, cout, None, None, None, 1, 1, 3, 1)
    f(); torch.cuda.synchronize(); lib.dsm_debug_read_stamps(buf, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f(); b.record(); torch.cuda.synchronize()
    lib.dsm_debug_read_stamps(buf, 1)
    tot = [0] * 8; nb = 0
    for i in range(1024):
        row = [buf[i * 8 + j] for j in range(8)]
        if sum(row):
            nb += 1
            for j in range(8): tot[j] += row[j]
    s = float(sum(tot))
    print("conv2d %d->%d %s: %.1f us, %d workgroups, stamped cycles/WG %.0f" % (cin, cout, hw, a.elapsed_time(b) * 1e3, nb, s / nb))
    for j, n in enumerate(NAMES):
        if tot[j]: print("   %-22s %6.2f %%  %9.0f" % (n, 100 * tot[j] / s, tot[j] / nb))
