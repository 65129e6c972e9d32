import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bsdf_diffusion_sampling_amd.sharding import bucket_by_material
ids=torch.randint(0,52,(1<<24,),device='cuda')
for _ in range(5): bucket_by_material(ids,52)
torch.cuda.synchronize()
