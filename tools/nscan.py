"""Per-query kernel time vs batch size (plugin-level pdf launches; BSDFD_CHUNK_LOG2 overrides the tile-chunk size in a tools build: tools/ab_build.sh tune "-include tools/tuning_knobs.h", BSDFD_LIB_PATH=build_ab/lib_tune.so)."""
import sys, os, numpy as np, torch, time
sys.path.insert(0,'.')
from bsdf_diffusion_sampling_amd import weights as W, _lib
from bsdf_diffusion_sampling_amd.sampler import FlowSampler
import bench
dev=torch.device('cuda')
sizes=[int(x) for x in sys.argv[1:]] or [12,14,16,17,18,19,20,22,24]
for dom,T in (("disk",4),("disk",8),("spherical",8)):
    fw=W.load(W.shipped_path("aniso_miro_7_rgb",dom)); s=FlowSampler(fw)
    wi=bench.make_wi(dom,1<<24,1,dev); wo=torch.empty_like(wi); p=torch.empty(1<<24,device=dev); p2=torch.empty_like(p)
    s.plugin_sample(wi,None,T=T,seed=1,offset=0,out=(wo,p))
    t0=time.time()
    while time.time()-t0<0.3: s.plugin_pdf(wi[:1<<20],wo[:1<<20],T=T,out=p2[:1<<20]); torch.cuda.synchronize()
    line=[]
    for lg in sizes:
        n=1<<lg
        reps=max(5,min(200,(1<<26)//n))
        for _ in range(3): s.plugin_pdf(wi[:n],wo[:n],T=T,out=p2[:n])
        s.set_profiling(True)
        for _ in range(reps): s.plugin_pdf(wi[:n],wo[:n],T=T,out=p2[:n])
        k,ms=s.profile_read(); tp=ms/k
        line.append(f"2^{lg}:{tp*1e3:.1f}us({tp*1e6/n:.3f})")
    print(f"{dom} T={T} cl={os.environ.get('BSDFD_CHUNK_LOG2','auto')}: "+" ".join(line),flush=True)
