"""Timing of the other BASELINE configs (parity cases, not the headline): DispNetC / iResNet
correlation at 384x1280 and whole GCNet at 256x512, D=192, on one MI355X."""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from dsmnet_amd import costvolume as cv
from dsmnet_amd.models import model_create_by_name


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


torch.manual_seed(0)
with torch.no_grad():
    fl, fr = torch.randn(1, 128, 96, 320, device="cuda"), torch.randn(1, 128, 96, 320, device="cuda")
    ms = t(lambda: cv.corr1d(fl, fr, 41), 50)
    print("config #2  corr1d D=41 (1,128,96,320): %.1f us  (36.5 MB -> %.2f TB/s, L3-resident)" % (ms * 1e3, 36.5e6 / (ms * 1e-3) / 1e12))
    ms = t(lambda: cv.corr1d(fl, fr, 81), 50)
    print("iResNet    corr1d D=81 (1,128,96,320): %.1f us" % (ms * 1e3))
    cl, cr = torch.randn(1, 64, 192, 640, device="cuda"), torch.randn(1, 64, 192, 640, device="cuda")
    ms = t(lambda: cv.corr1d(cl, cr, 41, 2, 3), 50)
    print("iResNet  r_corr k3 s2 D=41 (1,64,192,640): %.1f us" % (ms * 1e3))
    L, R = torch.randn(1, 3, 384, 1280, device="cuda"), torch.randn(1, 3, 384, 1280, device="cuda")
    m = model_create_by_name("dispnetcorr", 192).cuda().eval()
    print("DispNetC forward 384x1280: %.2f ms" % t(lambda: m(L, R), 5))
    m = model_create_by_name("iresnet", 192).cuda().eval()
    print("iResNet forward 384x1280: %.2f ms" % t(lambda: m(L, R), 5))
    del m
    g = model_create_by_name("gcnet", 192).cuda().eval()
    L2, R2 = torch.randn(1, 3, 256, 512, device="cuda"), torch.randn(1, 3, 256, 512, device="cuda")
    timer = cv.LaunchTimer(); 
    g(L2, R2); torch.cuda.synchronize()
    cv.set_timer(timer)
    ms = t(lambda: g(L2, R2), 5)
    cv.set_timer(None)
    print("config #3  GCNet forward 256x512 D=192: %.2f ms (882.6 GF trunk)" % ms)
    for k, v in sorted(timer.summary().items(), key=lambda kv: -kv[1]["ms"]):
        n = v["launches"]
        rate = v["work"] / (v["ms"] * 1e-3)
        print("   %-46s x%-3d %8.1f us avg  %8.2f %s" % (k, n // 6, v["ms"] / n * 1e3,
              rate / 1e12, "TFLOP/s" if "mfma" in k else "TB/s"))
