"""Cost of the full plugin sample() (flow kernel + ground-truth eval + weight/firefly logic) vs its kernels."""
import sys, time, torch, numpy as np
sys.path.insert(0,'.')
import bench
from bsdf_diffusion_sampling_amd.plugin_base import SurfaceInteraction
dev=torch.device('cuda'); n=1<<20
for mod in ("brdf_measured_disk","brdf_measured_spherical"):
    M=__import__("bsdf_diffusion_sampling_amd."+mod,fromlist=["MyBSDF"]).MyBSDF
    plug=M({"filename":"chm_orange_rgb","measured_dir":"tests/golden"})
    bare=M({"filename":"chm_orange_rgb","measured":False})
    wi=bench.make_wi("disk",n,1,dev); si=SurfaceInteraction(wi)
    def tm(fn,reps=20):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps*1e3
    bs,_=bare.sample(None,si,seed=1)
    print(mod, "sample() sampler only %.3f ms | with ground truth (weight+firefly) %.3f ms | eval() %.3f ms | pdf() %.3f ms | eval_pdf() %.3f ms"%(
        tm(lambda: bare.sample(None,si,seed=1)), tm(lambda: plug.sample(None,si,seed=1)), tm(lambda: plug.eval(None,si,bs.wo)), tm(lambda: plug.pdf(None,si,bs.wo)), tm(lambda: plug.eval_pdf(None,si,bs.wo))))
