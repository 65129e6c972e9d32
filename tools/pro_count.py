#!/usr/bin/env python3
"""Executed instructions of the per-tile prologue + epilogue: launches the plugin calls at T = 1 and T = 2 (one launch each, in a fixed
order) so that a `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU_TRANS` pass of this script
gives, per case, loop = count(T=2) - count(T=1) and fixed = count(T=1) - loop.   python tools/pro_count.py [disk|spherical] [N]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bsdf_diffusion_sampling_amd import weights as W
from bsdf_diffusion_sampling_amd.sampler import FlowSampler
dom = sys.argv[1] if len(sys.argv) > 1 else "disk"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
dev = torch.device("cuda")
s = FlowSampler(W.load(W.shipped_path("aniso_miro_7_rgb", dom)))
wi = bench.make_wi(dom, n, 1234, dev)
wo = torch.empty((n, 3), device=dev); p = torch.empty(n, device=dev)
ctx = s.new_context(n)
x0 = torch.zeros((n, 2), device=dev) + 0.1
s.plugin_sample(wi, None, T=8, seed=3, out=(wo, p), ctx_out=ctx)   # launch 0: fills wo and the context
torch.cuda.synchronize()
CASES = ["sample", "sample_ctx_read", "sample_x0", "pdf", "pdf_ctx_read"]
for T in (1, 2):
    s.plugin_sample(wi, None, T=T, seed=3, out=(torch.empty_like(wo), p))
    s.plugin_sample(wi, None, T=T, seed=3, out=(torch.empty_like(wo), p), ctx_in=ctx)
    s.plugin_sample(wi, x0, T=T, seed=3, out=(torch.empty_like(wo), p))
    s.plugin_pdf(wi, wo, T=T, out=p)
    s.plugin_pdf(wi, wo, T=T, out=p, ctx_in=ctx)
torch.cuda.synchronize()
print("launch order: 1 warm launch, then for T in (1, 2):", CASES)
