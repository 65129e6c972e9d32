"""Per-step / per-query cost split: time the kernel at several T and fit t = a + b*T."""
import sys, numpy as np, torch, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from conftest import load_case
from bsdf_diffusion_sampling_amd.sampler import FlowSampler
dev=torch.device('cuda')
t=lambda a: torch.from_numpy(np.ascontiguousarray(a,dtype=np.float32)).to(dev)
stems = sys.argv[1:] or ["aniso_miro_7_rgb_disk","aniso_miro_7_rgb_spherical","aniso_miro_7_rgb_spherical_complex"]
for stem in stems:
    g,fw=load_case(stem)
    N=1<<20
    wi=t(np.tile(g["wi"],(N//2048,1))); x0=t(np.tile(g["x0"],(N//2048,1)))
    for prec in ("split3","f16"):
        s=FlowSampler(fw,precision=prec); s.set_profiling(True)
        t0=time.time()
        while time.time()-t0<0.15: s.network_sampling(wi,x0,T=8); torch.cuda.synchronize()  # leave the idle clocks (tools/ramp.py)
        res={}
        for T in (1,2,4,8,16,32):
            for _ in range(2): s.network_sampling(wi,x0,T=T)
            s.set_profiling(True)
            for _ in range(5): s.network_sampling(wi,x0,T=T)
            n,ms=s.profile_read(); res[T]=ms/n
        Ts=np.array(list(res)); ts=np.array([res[k] for k in res])
        b,a=np.polyfit(Ts,ts,1)
        ntile=N/16
        print(f"{stem} {prec}: "+" ".join(f"T{k}={v*1e3:.0f}us" for k,v in res.items())+f" | per-step {b*1e3:.1f} us ({b*1e-3*2.15e9*1024/ntile:.0f} cyc/tile-step @2.15GHz), fixed {a*1e3:.1f} us ({a*1e-3*2.15e9*1024/ntile:.0f} cyc/tile)", flush=True)
    s=FlowSampler(fw,precision="split3")
    for name,fn in (("sample+rng",lambda: s.network_sampling(wi,None,T=8)),("pdf",lambda: s.network_pdf(x0,wi,T=8)),("samples_only",lambda: s.flow_samples_only(wi,x0,T=8))):
        for _ in range(2): fn()
        s.set_profiling(True)
        for _ in range(5): fn()
        n,ms=s.profile_read(); print(f"   {name}: {ms/n*1e3:.0f} us")
