"""A/B: network_sampling with a supplied x0 vs the in-kernel draw (same kernel, same T)."""
import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from conftest import load_case
from bsdf_diffusion_sampling_amd.sampler import FlowSampler
dev=torch.device('cuda')
t=lambda a: torch.from_numpy(np.ascontiguousarray(a,dtype=np.float32)).to(dev)
g,fw=load_case("aniso_miro_7_rgb_disk")
N=1<<20
wi=t(np.tile(g["wi"],(N//2048,1))); x0=t(np.tile(g["x0"],(N//2048,1)))
s=FlowSampler(fw,precision="split3")
def tm(fn):
    for _ in range(2): fn()
    s.set_profiling(True)
    for _ in range(10): fn()
    n,ms=s.profile_read(); return ms/n*1e3
xr,_=s.network_sampling(wi,None,T=1)   # x after 1 step; a draw-like x0
x0b=s.flow_samples_only(wi, torch.zeros_like(x0), T=1)
gen=torch.Generator(device=dev).manual_seed(1)
wi_r=(torch.rand(N,2,device=dev,generator=gen)-0.5)*1.2
for rep in range(3):
    print("rng        ", f"{tm(lambda: s.network_sampling(wi,None,T=8)):.0f} us")
    print("x0 golden  ", f"{tm(lambda: s.network_sampling(wi,x0,T=8)):.0f} us")
    print("x0 zeros   ", f"{tm(lambda: s.network_sampling(wi,torch.zeros_like(x0),T=8)):.0f} us")
    print("x0 = xr    ", f"{tm(lambda: s.network_sampling(wi,xr,T=8)):.0f} us")
    print("rng wi rand", f"{tm(lambda: s.network_sampling(wi_r,None,T=8)):.0f} us")
    print("x0 wi rand ", f"{tm(lambda: s.network_sampling(wi_r,x0,T=8)):.0f} us")
