# upper bound of what cheaper gather / scatter kernels could buy mixed_16Mi: the pipeline with the two kernels REMOVED (results wrong)
import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from bsdf_diffusion_sampling_amd.materials import MaterialTable, WavefrontPipeline
dev = torch.device("cuda", 0)
tab = MaterialTable.all_measured()
n = 1 << 24
ids = torch.randint(0, len(tab), (n,), generator=torch.Generator().manual_seed(1)).to(dev)
wi = bench.make_wi("spherical", n, 1234, dev)
def run(pipe, reps=12):
    for s in tab.samplers: s.set_profiling(True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); w = None
    for k in range(reps): w = pipe.push(ids, wi, seed=k, ready=False)
    w.result(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps * 1e3
    km = sum(s.profile_read()[1] for s in tab.samplers) / reps
    for s in tab.samplers: s.set_profiling(False)
    return dt, km
full = WavefrontPipeline(tab, direct=False)
bare = WavefrontPipeline(tab, direct=False)
class T2:  # the table with gather / scatter turned into no-ops
    def __init__(self, t): self.t = t
    def __getattr__(self, k): return getattr(self.t, k)
    def gather(self, plan, wi): return wi
    def scatter(self, plan, *ts): return ts
bare.tab = T2(tab)
for rnd in range(4):
    a = run(full); b = run(bare)
    print(json.dumps({"round": rnd, "full_ms": a[0], "full_kernels_ms": a[1], "no_gather_scatter_ms": b[0], "no_gs_kernels_ms": b[1], "ratio": b[0] / a[0]}), flush=True)
