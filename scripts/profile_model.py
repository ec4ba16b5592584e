"""One model's eval forward, N times, for `rocprofv3 --kernel-trace --stats -- python3 scripts/profile_model.py <net> [H W] [N]`."""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from dsmnet_amd.models import model_create_by_name
net = sys.argv[1]
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (384, 1280)
N = int(sys.argv[4]) if len(sys.argv) > 4 else 10
torch.manual_seed(0)
m = model_create_by_name(net, 192).cuda().eval()
L, R = torch.randn(1, 3, H, W, device="cuda"), torch.randn(1, 3, H, W, device="cuda")
with torch.no_grad():
    for _ in range(N):
        m(L, R)
torch.cuda.synchronize()
