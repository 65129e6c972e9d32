import sys, torch, time
sys.path.insert(0,'.')
from bsdf_diffusion_sampling_amd.encoding import positional_encoding_1
for lg in (20,22,24,25):
    n=1<<lg
    x=(torch.rand(n,2,device='cuda')*2-1)
    out=torch.empty(n,22,device='cuda')
    for _ in range(20): positional_encoding_1(x,5,out=out)
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    reps=20
    e0.record()
    for _ in range(reps): positional_encoding_1(x,5,out=out)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/reps
    print(f"N=2^{lg}: {ms*1e3:.1f} us  {n*96/ms/1e6:.0f} GB/s algorithmic (8 B read + 88 B written per row)")
