"""Where the time of a mixed-material step goes (config 4 per-GPU share: 16 Mi queries, 52 materials)."""
import sys, time, torch
sys.path.insert(0,'.')
import bench
from bsdf_diffusion_sampling_amd.materials import MaterialTable
from bsdf_diffusion_sampling_amd.sharding import bucket_by_material
dev=torch.device('cuda'); n=1<<24
tab=MaterialTable.all_measured()
g=torch.Generator(device=dev).manual_seed(0)
mid=torch.randint(0,len(tab),(n,),device=dev,generator=g)
wi=bench.make_wi("disk",n,1,dev)
def tm(fn,reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps*1e3
print("argsort+bincount   %.2f ms"%tm(lambda: bucket_by_material(mid,len(tab))))
perm,counts=bucket_by_material(mid,len(tab))
print("counts.cpu()       %.2f ms"%tm(lambda: counts.cpu().tolist()))
print("gather wi[perm]    %.2f ms"%tm(lambda: wi[perm].contiguous()))
wo=torch.empty_like(wi); pdf=torch.empty(n,device=dev)
def scat():
    a=torch.empty_like(wo); a[perm]=wo; b=torch.empty_like(pdf); b[perm]=pdf
print("scatter wo,pdf     %.2f ms"%tm(scat))
print("sample() total     %.2f ms"%tm(lambda: tab.sample(mid,wi,seed=1)))
wo,_=tab.sample(mid,wi,seed=1)
print("pdf() total        %.2f ms"%tm(lambda: tab.pdf(mid,wi,wo)))
