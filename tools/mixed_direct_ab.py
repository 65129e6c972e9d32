#!/usr/bin/env python3
"""config 4 (material-tagged wavefronts): the flow kernels reading / writing lane order through the bucket permutation
(bsdfd_opts.row_index, WavefrontPipeline(direct=True)) against round 5's form (gather of wi and scatter of the results as kernels
of their own on side streams), by wavefront size — alternating, wall time per wavefront of sample() + pdf() incl. the bucketing,
and the flow kernels' own summed time (HIP events).   python tools/mixed_direct_ab.py [--sizes 18,20,22,24] [--rounds 3]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from bsdf_diffusion_sampling_amd.materials import MaterialTable, WavefrontPipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="18,20,22,24")
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    tab = MaterialTable.all_measured()
    out = []
    for lg in (int(x) for x in a.sizes.split(",")):
        n = 1 << lg
        ids = torch.randint(0, len(tab), (n,), generator=torch.Generator().manual_seed(1)).to(dev)
        wi = bench.make_wi("spherical", n, 1234, dev)
        pipes = {"direct": WavefrontPipeline(tab, direct=True), "gather": WavefrontPipeline(tab, direct=False)}
        reps = max(12, min(40, (1 << 26) // n))   # enough wavefronts per round for the three-stream form to reach its steady state
        res = {k: {"wall_ms": [], "kernel_ms": []} for k in pipes}
        for rnd in range(a.rounds + 1):
            for name, pipe in pipes.items():
                for s in tab.samplers:
                    s.set_profiling(True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                w = None
                for k in range(reps):
                    w = pipe.push(ids, wi, seed=k, ready=False)
                w.result()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / reps * 1e3
                km = sum(s.profile_read()[1] for s in tab.samplers) / reps
                for s in tab.samplers:
                    s.set_profiling(False)
                if rnd:   # round 0 warms up
                    res[name]["wall_ms"].append(dt)
                    res[name]["kernel_ms"].append(km)
        med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
        row = {"lanes": n, "materials": len(tab), "wavefronts_per_round": reps}
        for name in pipes:
            row[name] = {"wall_ms": med(res[name]["wall_ms"]), "flow_kernel_ms": med(res[name]["kernel_ms"]),
                         "Msamples_per_s": n / med(res[name]["wall_ms"]) / 1e3}
        row["direct_over_gather_wall"] = row["direct"]["wall_ms"] / row["gather"]["wall_ms"]
        out.append(row)
        print(json.dumps(row), flush=True)
        del pipes, ids, wi
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
