import sys, time, torch
sys.path.insert(0,'.')
import bench
from bsdf_diffusion_sampling_amd import weights as W, _lib
from bsdf_diffusion_sampling_amd.sampler import FlowSampler
dev=torch.device('cuda'); n=1<<20
for dom,T,var in (("disk",4,0),("disk",8,0),("spherical",8,0),("spherical",8,1)):
    s=FlowSampler(W.load(W.shipped_path("aniso_miro_7_rgb",dom)))
    wi=bench.make_wi(dom,n,1,dev); wl=bench.make_wi(dom,n,2,dev)
    wo,p=s.plugin_sample(wi,None,T=T,variant=var,seed=3,offset=7); pl=s.plugin_pdf(wi,wl,T=T,variant=var)
    wo2,p2,pl2=s.plugin_sample_pdf(wi,wl,None,T=T,variant=var,seed=3,offset=7)
    print(dom,T,var,"identical:",torch.equal(wo,wo2),torch.equal(p,p2),torch.equal(pl,pl2))
    t0=time.time()
    while time.time()-t0<0.2: s.plugin_sample(wi,None,T=T,variant=var,seed=3); torch.cuda.synchronize()
    def tm(fn):
        for _ in range(3): fn()
        s.set_profiling(True)
        for _ in range(20): fn()
        k,ms=s.profile_read(); return ms/20*1e3
    a=tm(lambda:(s.plugin_sample(wi,None,T=T,variant=var,seed=3),s.plugin_pdf(wi,wl,T=T,variant=var)))
    b=tm(lambda:s.plugin_sample_pdf(wi,wl,None,T=T,variant=var,seed=3))
    print(f"   separate {a:.1f} us  fused {b:.1f} us  ({(1-b/a)*100:.1f} % less)")
