"""Diagnostic: where does a conv3d_mfma workgroup spend its cycles?  Uses the stamped build
(python dsmnet_amd/csrc/build.py --stamps).  Prints per-phase cycle shares (wave 0 of each WG)."""
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
CL = torch.channels_last_3d
NAMES = ["barrier1", "pf-wait+commit", "barrier2", "prefetch issue", "multiply", "epilogue", "acc reset", "ring prologue"]
CASES = ((32, 1, (48, 96, 320)),) if "cout1" in sys.argv else ((32, 32, (48, 96, 320)), (64, 32, (48, 96, 320)), (64, 64, (24, 48, 160)))
for cin, cout, dims in CASES:
    x = torch.randn(1, cin, *dims, device="cuda").contiguous(memory_format=CL)
    w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05
    packed = cv.pack_conv3d_weight(w, False)
    sc, sh = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
    cv.conv3d_block(x, packed, cout, sc, sh, None, 1, False, 1)
    torch.cuda.synchronize()
    lib.dsm_debug_read_stamps(buf, 1)
    cv.conv3d_block(x, packed, cout, sc, sh, None, 1, False, 1)
    torch.cuda.synchronize()
    lib.dsm_debug_read_stamps(buf, 1)
    tot = [0] * 8
    nb = 0
    for b in range(1024):
        row = [buf[b * 8 + i] for i in range(8)]
        if sum(row):
            nb += 1
            for i in range(8):
                tot[i] += row[i]
    s = float(sum(tot))
    print("conv %d->%d %s: %d workgroups, mean cycles/WG %.0f" % (cin, cout, dims, nb, s / nb))
    for i, n in enumerate(NAMES):
        print("   %-16s %6.2f %%   %9.0f cycles/WG" % (n, 100 * tot[i] / s, tot[i] / nb))
