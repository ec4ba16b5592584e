"""Error of the two convolution precisions against a float64 reference on the dominant layer
shape (32 -> 32, 3x3x3): run once per mode, `DSM_CONV_PRECISION=fp32|bf16x3 python scripts/precision_check.py`."""
import os, sys
sys.path.insert(0, ".")
import torch
from dsmnet_amd import costvolume as cv
torch.manual_seed(0)
x = torch.randn(1, 32, 24, 48, 160, device="cuda") * 3
w = torch.randn(32, 32, 3, 3, 3, device="cuda") * 0.05
y = cv.conv3d_block(x.contiguous(memory_format=torch.channels_last_3d), cv.pack_conv3d_weight(w, False),
                    32, None, None, None, 1, False, 0)
ref = torch.nn.functional.conv3d(x.double().cpu(), w.double().cpu(), padding=1)
err = (y.double().cpu() - ref).abs()
print("%-7s max|err| %.3e  rms err %.3e  (|ref| max %.3f rms %.3f)  -> max rel %.2e" % (
    os.environ.get("DSM_CONV_PRECISION", "bf16x3"), err.max().item(), err.pow(2).mean().sqrt().item(),
    ref.abs().max().item(), ref.pow(2).mean().sqrt().item(), err.max().item() / ref.abs().max().item()))
