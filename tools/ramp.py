"""Kernel time vs launches since idle: shows the clock ramp of a cold GPU."""
import sys, numpy as np, torch, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from conftest import load_case
from bsdf_diffusion_sampling_amd.sampler import FlowSampler
dev=torch.device('cuda')
t=lambda a: torch.from_numpy(np.ascontiguousarray(a,dtype=np.float32)).to(dev)
g,fw=load_case("aniso_miro_7_rgb_disk")
N=1<<20
wi=t(np.tile(g["wi"],(N//2048,1)))
s=FlowSampler(fw,precision="split3")
torch.cuda.synchronize(); time.sleep(float(sys.argv[1]) if len(sys.argv)>1 else 2.0)
out=[]
t0=time.time()
for blk in range(40):
    s.set_profiling(True)
    for _ in range(25): s.network_sampling(wi,None,T=8)
    n,ms=s.profile_read(); out.append((time.time()-t0, ms/n*1e3))
print(" ".join(f"{a*1e3:.0f}ms:{b:.0f}" for a,b in out))
