import sys, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from conftest import load_case
from bsdf_diffusion_sampling_amd.sampler import FlowSampler
dev=torch.device('cuda')
t=lambda a: torch.from_numpy(np.ascontiguousarray(a,dtype=np.float32)).to(dev)
g,fw=load_case("aniso_miro_7_rgb_disk"); N=1<<20
wi=t(np.tile(g["wi"],(N//2048,1))); x0=t(np.tile(g["x0"],(N//2048,1)))
s=FlowSampler(fw); res={}
for T in (1,2,3,4):
    for _ in range(3): s.network_sampling(wi,x0,T=T)
    s.set_profiling(True)
    for _ in range(10): s.network_sampling(wi,x0,T=T)
    n,ms=s.profile_read(); res[T]=ms/n
b,a=np.polyfit(list(res),list(res.values()),1)
print(f"fixed {a*1e3:.1f} us  per-step {b*1e3:.1f} us", res)
