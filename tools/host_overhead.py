"""Host-side cost of one call through the Python wrapper vs the kernel time (small wavefronts)."""
import sys, time, torch
sys.path.insert(0,'.')
import bench
from bsdf_diffusion_sampling_amd import weights as W
from bsdf_diffusion_sampling_amd.sampler import FlowSampler
dev=torch.device('cuda')
for binding, n in [(b, n) for b in ("ctypes", "torch") for n in (4096, 65536)]:
    s=FlowSampler(W.load(W.shipped_path("aniso_miro_7_rgb","disk")), binding=binding)
    wi=bench.make_wi("disk",n,1,dev); wo=torch.empty_like(wi); p=torch.empty(n,device=dev); p2=torch.empty(n,device=dev)
    for _ in range(200): s.plugin_sample(wi,None,T=4,seed=1,offset=0,out=(wo,p)); s.plugin_pdf(wi,wo,T=4,out=p2)
    torch.cuda.synchronize(); t0=time.perf_counter()
    K=2000
    for _ in range(K): s.plugin_sample(wi,None,T=4,seed=1,offset=0,out=(wo,p)); s.plugin_pdf(wi,wo,T=4,out=p2)
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    g=torch.cuda.CUDAGraph(); st=torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        with torch.cuda.graph(g,stream=st):
            for _ in range(10): s.plugin_sample(wi,None,T=4,seed=1,offset=0,out=(wo,p)); s.plugin_pdf(wi,wo,T=4,out=p2)
    torch.cuda.synchronize()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); t3=time.perf_counter()
    for _ in range(K//10): g.replay()
    torch.cuda.synchronize(); t4=time.perf_counter()
    print(f"binding={binding} N={n}: host issue {1e6*(t1-t0)/K/2:.1f} us/call, end-to-end {1e6*(t2-t0)/K/2:.1f} us/call, hipGraph replay {1e6*(t4-t3)/K/2:.1f} us/call")
