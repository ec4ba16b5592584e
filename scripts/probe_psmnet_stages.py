import sys, time; sys.path.insert(0,'.')
import torch
from dsmnet_amd.models import model_create_by_name
from dsmnet_amd import costvolume as cv
m = model_create_by_name("psmnet",192).cuda().eval()
# calibrated-ish BN not needed for timing
L = torch.randn(1,3,384,1280,device='cuda'); R = torch.randn(1,3,384,1280,device='cuda')
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    s=torch.cuda.Event(True); e=torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
with torch.no_grad():
    print("full fwd ms", t(lambda: m(L,R)))
    print("features ms", t(lambda: m.features(L,R)))
    fl,fr = m.features(L,R)
    print("volume NDHWC ms", t(lambda: cv.concat_volume(fl,fr,48,True,True), 20))
    print("volume NCDHW ms", t(lambda: cv.concat_volume(fl,fr,48,True,False), 20))
    cost = cv.concat_volume(fl,fr,48,True,True)
    print("dres0 ms", t(lambda: m.dres0(cost)))
    c0 = m.dres0(cost)
    print("dres1 ms", t(lambda: m.dres1(c0, residual=c0)))
    print("hourglass ms", t(lambda: m.dres2(c0,None,None,skip=c0)))
    print("classif ms", t(lambda: m.classif1(c0)))
    c1 = m.classif1(c0)
    print("softargmin ms", t(lambda: cv.soft_argmin(c1,(192,384,1280)), 20))
    print("conv 32->32 ms", t(lambda: m.dres0[2](c0), 10), " => TF/s", 81.5e9/ (t(lambda: m.dres0[2](c0), 10)*1e-3)/1e12)
