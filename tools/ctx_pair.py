#!/usr/bin/env python3
"""What the per-query context buys per sample() + pdf() pair of ONE wavefront, in both call orders, against the pair without it:
kernel time (HIP events on the launch stream), the three forms interleaved in rounds so that clock drift cancels.
    python tools/ctx_pair.py [N]     ->  one line per (domain, T)"""
import sys
sys.path.insert(0, '.')
import torch
import bench
from bsdf_diffusion_sampling_amd import weights as W
from bsdf_diffusion_sampling_amd.sampler import FlowSampler

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dev = torch.device("cuda")
ORDERS, ROUNDS, PAIRS = ("sample_then_pdf", "pdf_then_sample", "no_context"), 6, 8
for dom, T in (("disk", 4), ("disk", 8), ("spherical", 8)):
    s = FlowSampler(W.load(W.shipped_path("aniso_miro_7_rgb", dom)))
    wi = bench.make_wi(dom, n, 1234, dev)
    wo = torch.empty((n, 3), device=dev); wo2 = torch.empty((n, 3), device=dev)
    p = torch.empty(n, device=dev); p2 = torch.empty(n, device=dev)
    ctx = s.new_context(n)
    s.plugin_sample(wi, None, T=T, seed=1, out=(wo, p))

    def pair(order, k):
        if order == "sample_then_pdf":
            s.plugin_sample(wi, None, T=T, seed=k, out=(wo2, p), ctx_out=ctx); s.plugin_pdf(wi, wo, T=T, out=p2, ctx_in=ctx)
        elif order == "pdf_then_sample":
            s.plugin_pdf(wi, wo, T=T, out=p2, ctx_out=ctx); s.plugin_sample(wi, None, T=T, seed=k, out=(wo2, p), ctx_in=ctx)
        else:
            s.plugin_sample(wi, None, T=T, seed=k, out=(wo2, p)); s.plugin_pdf(wi, wo, T=T, out=p2)
    for k in range(60):
        pair(ORDERS[k % 3], k)
    torch.cuda.synchronize()
    s.set_profiling(True)
    tot, seen = dict.fromkeys(ORDERS, 0.0), 0.0
    for r in range(ROUNDS):
        for o in ORDERS:
            for k in range(PAIRS):
                pair(o, 100 + k)
            _, ms = s.profile_read()
            tot[o] += ms - seen
            seen = ms
    mhz = s.profile_clock_mhz()
    s.set_profiling(False)
    v = {o: tot[o] / (ROUNDS * PAIRS) for o in ORDERS}
    print(f"{dom} T={T} N={n}: pair without context {v['no_context']*1e3:7.1f} us | sample fills, pdf reads {v['sample_then_pdf']*1e3:7.1f} "
          f"({(v['sample_then_pdf']/v['no_context']-1)*100:+.2f} %) | pdf fills, sample reads {v['pdf_then_sample']*1e3:7.1f} "
          f"({(v['pdf_then_sample']/v['no_context']-1)*100:+.2f} %) | context {s.context_floats(n)*4/2**20:.0f} MiB @{mhz:.0f} MHz", flush=True)
    s.close()
