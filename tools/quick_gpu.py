import sys, numpy as np, torch, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from conftest import load_case, GOLDEN_CASES
from oracle import bsdf_oracle as O
from bsdf_diffusion_sampling_amd.sampler import FlowSampler
dev=torch.device('cuda')
t=lambda a: torch.from_numpy(np.ascontiguousarray(a,dtype=np.float32)).to(dev)
rel=lambda a,b: np.abs(a-b)/np.maximum(np.abs(b),1e-30)
for stem in GOLDEN_CASES:
    g,fw=load_case(stem); T=int(g["meta_T"]); orc=O.Oracle(fw)
    xo,po=orc.network_sampling(g["wi"],g["x0"],T); _,acc=orc.flow(g["x0"],g["wi"],T,False)
    ok=(np.abs(acc)>1e-3)&(np.abs(acc)<1e3); ok=ok&(np.abs(po)>1e-6*np.percentile(np.abs(po[ok]),99))
    for prec in ("f32","split3","f16"):
        s=FlowSampler(fw,precision=prec)
        x,p=s.network_sampling(t(g["wi"]),t(g["x0"]),T=T); x=x.cpu().numpy(); p=p.cpu().numpy()
        r=rel(p,po)[ok]
        print(f"{stem:40s} {prec:7s} x_err {np.abs(x-xo).max():.2e} pdf med {np.median(r):.2e} p99 {np.percentile(r,99):.2e} max {r.max():.2e} nan {np.isnan(p).sum()}", flush=True)
# perf
for stem,T in (("aniso_miro_7_rgb_disk",8),("aniso_miro_7_rgb_disk",4),("aniso_miro_7_rgb_spherical",8),("aniso_miro_7_rgb_spherical_complex",8)):
    g,fw=load_case(stem)
    N=1<<20
    wi=t(np.tile(g["wi"],(N//2048,1))); x0=t(np.tile(g["x0"],(N//2048,1)))
    for prec in ("f32","split3","f16"):
        s=FlowSampler(fw,precision=prec)
        for _ in range(2): s.network_sampling(wi,x0,T=T)
        torch.cuda.synchronize(); t0=time.time()
        for _ in range(5): s.network_sampling(wi,x0,T=T)
        torch.cuda.synchronize(); dt=(time.time()-t0)/5
        fl=s.flops_per_query(T)*N/dt
        print(f"{stem} T={T} {prec}: {dt*1e3:.3f} ms  {N/dt/1e6:.1f} Msamples/s  {fl/1e12:.1f} TFLOP/s ({fl/2.5e15*100:.2f}% fp16 peak)", flush=True)
