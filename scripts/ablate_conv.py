"""Diagnostic: time the dominant convolution with a diagnostic build of the library
(DSM_ABLATE=N python dsmnet_amd/csrc/build.py --stamps; cp the .so to the path given).  Outputs
of an ablated build are wrong on purpose; only the time matters."""
import os, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from dsmnet_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from dsmnet_amd import costvolume as cv
CL = torch.channels_last_3d
for cin, cout, dims in ((32, 32, (48, 96, 320)), (64, 32, (48, 96, 320))):
    x = torch.randn(1, cin, *dims, device="cuda").contiguous(memory_format=CL)
    w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05
    packed = cv.pack_conv3d_weight(w, False)
    f = lambda: cv.conv3d_block(x, packed, cout, None, None, None, 1, False, 1)
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f()
    b.record(); torch.cuda.synchronize()
    print("%s %d->%d: %.1f us" % (os.path.basename(sys.argv[1]), cin, cout, a.elapsed_time(b) * 100))

for cin, cout, hw in ((64, 64, (96, 320)), (128, 128, (96, 320)), (32, 32, (192, 640))):
    x = torch.randn(2, cin, *hw, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    packed = cv.pack_conv2d_weight(w)
    sc, sh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
    f = lambda: cv.conv2d_block(x, packed, cout, sc, sh, None, 1, 1, 3, 1)
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): f()
    b.record(); torch.cuda.synchronize()
    print("%s 2-D %d->%d: %.1f us" % (os.path.basename(sys.argv[1]), cin, cout, a.elapsed_time(b) * 20))
