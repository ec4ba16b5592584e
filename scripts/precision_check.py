"""Error of the convolution precisions against a float64 reference, per kernel variant and
input magnitude, every precision in one process: `python scripts/precision_check.py`."""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from dsmnet_amd import costvolume as cv
for cin, cout, dims, xs in ((32, 32, (24, 48, 160), 3.0), (32, 32, (8, 16, 40), 1e-8), (32, 64, (8, 16, 40), 1e-8),
                            (64, 64, (2, 4, 10), 1.0), (64, 64, (2, 4, 10), 1e-9), (64, 32, (8, 16, 40), 1e-6),
                            (32, 32, (8, 16, 40), 1e4), (64, 32, (8, 16, 40), 1e7)):
    torch.manual_seed(0)
    x = torch.randn(1, cin, *dims, device="cuda") * xs
    x = x * (torch.rand_like(x) > 0.3)                       # ReLU-like zeros
    w = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05
    ref = torch.nn.functional.conv3d(x.double().cpu(), w.double().cpu(), padding=1)
    for mode in ("fp32", "bf16x3", "f16x2", "f16"):
        old = cv.set_option("conv_precision", mode)
        y = cv.conv3d_block(x.contiguous(memory_format=torch.channels_last_3d), cv.pack_conv3d_weight(w, False),
                            cout, None, None, None, 1, False, 0)
        cv.set_option("conv_precision", old)
        err = (y.double().cpu() - ref).abs()
        print("%-7s %3d->%-3d %-14s x~%.0e  max rel err %.2e  rms rel %.2e" % (
            mode, cin, cout, dims, xs, err.max().item() / ref.abs().max().item(),
            err.pow(2).mean().sqrt().item() / ref.pow(2).mean().sqrt().item()))
