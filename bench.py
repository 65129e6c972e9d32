#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the neural-BSDF hot path, sample() + pdf().

A "pass" is one trip of a wavefront of shading queries through the hot path: ``MyBSDF.sample`` (tensor core:
warp + T Euler steps + Jacobian + guards, in-kernel RNG) followed by ``MyBSDF.pdf`` on the produced directions.
The default workload is BASELINE.json configs[1]: single measured BSDF (aniso_miro_7_rgb), disk-domain net,
1 Mi queries per wavefront, 8 denoise steps, per GPU (weak scaling: every rank owns its own wavefronts).  A "step"
is ``passes_per_step`` consecutive wavefronts (chosen once, at setup, so that the K timed steps last >= 0.5 s
whatever K is — a 20 ms timed region moves by percents with one clock hiccup); value = queries through
sample()+pdf() per second, whole job.  Inputs are resident in HBM before the timed region.  There is no data-path
collective; with N > 1 ranks the final (wo, pdf) shards are concatenated on rank 0 with one RCCL gather inside the
timed region, and the line also carries the rate with no gather and with a gather every step.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python bench.py --gpus 8 --steps 50 --warmup 5                      # starts its own 8 ranks (child processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 50 --warmup 5     # or under an external launcher
    python bench.py --workload mixed_16Mi                               # configs[3] per-GPU share

Prints ONE JSON line on rank 0: the contract's fields plus ``roofline`` (fused flow kernel vs the dense fp16-MFMA
peak, HIP events on the launch stream; ``issue_bound``: shader cycles per tile-step vs the instruction-issue model
of the loop), ``secondary`` (the other configs, timed in the same run), ``encoding_pass`` (HBM-bound),
``cpu_baseline`` (the torch-eager port of the reference's CPU path on this box's host cores, bounded sample).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP16_MFMA_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
MIN_TIMED_S = 0.5

WORKLOADS = {
    # name: (kind, material, domain, per-GPU queries per wavefront, T)
    "disk_1Mi_T8": ("single", "aniso_miro_7_rgb", "disk", 1 << 20, 8),             # BASELINE.json configs[1]
    "disk_1Mi_T4": ("single", "aniso_miro_7_rgb", "disk", 1 << 20, 4),             # plugin default T
    "spherical_16Mi_T8": ("single", "aniso_miro_7_rgb", "spherical", 1 << 24, 8),  # configs[2]
    "mixed_16Mi": ("mixed", "27 disk (T=4) + 25 spherical (T=8) measured materials", "mixed", 1 << 24, 0),  # configs[3] share
    "teacher_64x6_4Mi_T128": ("teacher", "aniso_miro_7_rgb", "spherical", 1 << 22, 128),
    "complex64_1Mi_T8": ("single", "aniso_miro_7_rgb", "spherical:complex", 1 << 20, 8),  # the 64-wide x 6 net, split3, with Jacobian
}
SECONDARY = ("disk_1Mi_T4", "spherical_16Mi_T8", "mixed_16Mi", "teacher_64x6_4Mi_T128", "complex64_1Mi_T8")
# the per-query context pays while it round-trips through the 256 MiB Infinity Cache and stops paying once it streams through
# HBM (plugin_base.NeuralBSDFCore applies the same gate): wavefronts whose record is larger run without it
CONTEXT_MAX_BYTES = 192 << 20
USE_CONTEXT = os.environ.get("BSDFD_BENCH_CONTEXT", "0") != "0"  # --context on: sample() hands the per-query context to pdf()


def make_wi(domain, n, seed, device):
    """disk — uniform on the disk of radius 0.95; spherical — theta_i ~ U(0,1.5),
    phi_i ~ U(-pi,pi); both handed over as unit vectors [N,3] (the warps are inside the timed region)."""
    import numpy as np
    import torch
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n, 2, generator=g)
    if domain == "disk":
        r, a = 0.95 * torch.sqrt(u[:, 0]), 2 * np.pi * u[:, 1]
        x, y = r * torch.cos(a), r * torch.sin(a)
        wi = torch.stack([x, y, torch.sqrt(torch.clamp(1 - x * x - y * y, min=0))], 1)
    else:
        th, ph = 1.5 * u[:, 0], (2 * u[:, 1] - 1) * np.pi
        wi = torch.stack([torch.sin(th) * torch.cos(ph), torch.sin(th) * torch.sin(ph), torch.cos(th)], 1)
    return wi.float().contiguous().to(device)


# ------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only)
# ------------------------------------------------------------------------------------------------------
def cpu_baseline(material, domain, T, budget_n=262144):
    """The reference's CPU PyTorch path, restated op-for-op (oracle/torch_eager_port.py — validated bit-identical
    to the reference and within ~15 % of its wall time, tests/golden/cpu_timing.json).  Two figures: `value` with
    the thread count that maximises throughput (torch eager with hundreds of threads is fork/join-bound), and
    `all_cores`: the contract's form (all host cores, 1 warm-up, median of 5) on a bounded sample."""
    import numpy as np
    import torch
    from bsdf_diffusion_sampling_amd import weights as W
    from oracle import torch_eager_port as P

    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    fw = W.load(W.shipped_path(material, domain))
    base, net = P.BaseNet(fw), P.VelocityNet(fw)

    def make_cond(n):
        g = torch.Generator().manual_seed(1234)
        u = torch.rand(n, 2, generator=g)
        if domain == "disk":
            r, a = 0.95 * torch.sqrt(u[:, 0]), 2 * np.pi * u[:, 1]
            return torch.stack([r * torch.cos(a), r * torch.sin(a)], 1).float()
        return torch.stack([1.5 * u[:, 0], (2 * u[:, 1] - 1) * np.pi], 1).float()

    def one_pass(cond):
        t0 = time.perf_counter()
        x, _ = P.network_sampling(base, net, cond, T)
        t1 = time.perf_counter()
        P.network_pdf(base, net, x, cond, T)
        return t1 - t0, time.perf_counter() - t1

    def measure(n, reps):
        cond = make_cond(n)
        one_pass(cond)  # warm-up
        ts, tp = [], []
        for _ in range(reps):
            a_, b_ = one_pass(cond)
            ts.append(a_)
            tp.append(b_)
        return float(np.median(ts)), float(np.median(tp))

    torch.manual_seed(1234)
    calib = make_cond(16384)
    best_thr, best_t, t_by_thr = 1, float("inf"), {}
    thr = 4
    while thr <= min(ncpu, 64):  # (beyond 64 threads eager only gets slower; a 256-thread calibration pass alone takes minutes)
        torch.set_num_threads(thr)
        one_pass(calib)
        t = sum(one_pass(calib))
        t_by_thr[thr] = t
        if t < best_t:
            best_thr, best_t = thr, t
        elif t > 1.5 * best_t:
            break
        thr *= 2
    torch.set_num_threads(best_thr)
    t_s, t_p = measure(budget_n, 3)
    out = {"value": budget_n / (t_s + t_p) / 1e6, "unit": "Msamples/s", "cores": best_thr, "kind": "port",
           "sample": f"{budget_n} queries x (sample()+pdf()), {domain} T={T}, torch {torch.__version__} eager fp32 with "
                     f"autograd (2 backward/step), 1 warm-up + median of 3; threads chosen by calibration "
                     f"({best_thr} of {ncpu} host CPUs)",
           "sample_Msps": budget_n / t_s / 1e6, "pdf_Msps": budget_n / t_p / 1e6}
    out["all_cores"] = cpu_baseline_all_cores(material, domain, T, ncpu)
    return out


ALL_CORES_BUDGET_S = 30.0
_ALL_CORES_CHILD = """
import sys, time, numpy as np
sys.path.insert(0, {root!r})
import torch
from bsdf_diffusion_sampling_amd import weights as W
from oracle import torch_eager_port as P
torch.set_num_threads({ncpu})
fw = W.load(W.shipped_path({material!r}, {domain!r}))
base, net = P.BaseNet(fw), P.VelocityNet(fw)
print("READY", flush=True)   # the parent's budget starts here: `import torch` alone takes 1-2 minutes on a cold box
def cond_of(n):
    g = torch.Generator().manual_seed(1234)
    u = torch.rand(n, 2, generator=g)
    if {domain!r} == "disk":
        r, a = 0.95 * torch.sqrt(u[:, 0]), 2 * np.pi * u[:, 1]
        return torch.stack([r * torch.cos(a), r * torch.sin(a)], 1).float()
    return torch.stack([1.5 * u[:, 0], (2 * u[:, 1] - 1) * np.pi], 1).float()
def one(cond):
    t0 = time.perf_counter()
    x, _ = P.network_sampling(base, net, cond, {T})
    t1 = time.perf_counter()
    P.network_pdf(base, net, x, cond, {T})
    return t1 - t0, time.perf_counter() - t1
# sizing: a 64-query probe prices the per-op fork/join cost of ALL threads on this host; the timed sample is the largest
# power of two whose 1 warm-up + 3 timed passes are predicted to fit the budget (torch eager at hundreds of threads is
# overhead-bound: a pass costs about the same from 64 to a few thousand queries)
t_probe = sum(one(cond_of(64)))
print("PROBE", 64, t_probe, flush=True)
n = {n_max}
left = {budget} * 0.8 - t_probe
while n > 64 and 4 * t_probe * max(1.0, n / 4096.0) > left:
    n //= 2
cond = cond_of(n)
for i in range(4):  # 1 warm-up + 3 timed
    a, b = one(cond)
    print("PASS", n, i, a, b, flush=True)
"""


def _cgroup_cpu_quota():
    """CPUs the container may actually use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unknown."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(1, int(math.ceil(int(q) / int(per))))
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else max(1, int(math.ceil(q / per)))
    except Exception:
        return None


def cpu_baseline_all_cores(material, domain, T, ncpu, n=16384):
    """The contract's form of the CPU baseline (torch.set_num_threads(ALL host cores), 1 warm-up, then timed
    passes), in a CHILD process with a hard time budget.  On the 256-thread GPU box torch eager with all threads is
    fork/join-bound (round 2: ONE pass of 16 Ki queries took ~2 minutes), so the child first prices a 64-query pass and
    sizes the timed sample to the budget; whatever finished is reported: median of the timed passes, else the warm-up
    pass, else the probe — `value` is a number whenever a single pass of 64 queries finishes inside the budget."""
    import numpy as np
    quota = _cgroup_cpu_quota()
    ladder, thr = [], ncpu if quota is None else max(1, min(ncpu, quota))
    while thr >= 1 and len(ladder) < 3:      # all CPUs first; if that is pathological on this host, half, then a quarter
        ladder.append(thr)
        thr //= 2
    per_try = ALL_CORES_BUDGET_S / len(ladder)
    res, t0 = None, time.perf_counter()
    tried = []

    def run_child(code, budget_s, import_allowance_s=240.0):
        """(stdout, timed_out): the child gets `budget_s` seconds AFTER it printed READY (imports and weight loading done; round 4's
        driver run spent all of its 3 x 10 s inside `import torch` on a cold 256-CPU box and reported no number)."""
        import selectors
        pr = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        sel = selectors.DefaultSelector()
        sel.register(pr.stdout, selectors.EVENT_READ)
        buf, start, ready_at = b"", time.perf_counter(), None
        timed_out = False
        while True:
            now = time.perf_counter()
            deadline = (ready_at + budget_s) if ready_at is not None else (start + import_allowance_s)
            if now >= deadline:
                timed_out = True
                break
            if sel.select(timeout=min(0.25, deadline - now)):
                chunk = os.read(pr.stdout.fileno(), 65536)
                if not chunk:
                    break
                buf += chunk
                if ready_at is None and b"READY" in buf:
                    ready_at = time.perf_counter()
            elif pr.poll() is not None:
                break
        if pr.poll() is None:
            pr.kill()
        pr.wait()
        sel.close()
        return buf.decode(errors="replace"), timed_out

    for thr in ladder:
        code = _ALL_CORES_CHILD.format(root=ROOT, ncpu=thr, material=material, domain=domain, n_max=n, T=T, budget=per_try)
        txt, timed_out = run_child(code, per_try)
        probe = [l.split() for l in txt.splitlines() if l.startswith("PROBE")]
        passes = [l.split() for l in txt.splitlines() if l.startswith("PASS")]
        tried.append({"threads": thr, "timed_out": timed_out, "passes_finished": len(passes)})
        res = {"unit": "Msamples/s", "cores": thr, "host_cpus": ncpu, "cgroup_cpu_quota": quota, "budget_s": ALL_CORES_BUDGET_S,
               "timed_out": timed_out, "timed_passes_finished": max(len(passes) - 1, 0), "attempts": tried}
        if len(passes) >= 2:
            n_used = int(passes[0][1])
            ts = [float(p[3]) + float(p[4]) for p in passes[1:]]
            res.update(value=n_used / float(np.median(ts)) / 1e6, basis=f"median of {len(ts)} timed passes after 1 warm-up", queries=n_used)
        elif len(passes) == 1:
            n_used = int(passes[0][1])
            res.update(value=n_used / (float(passes[0][3]) + float(passes[0][4])) / 1e6, basis="the warm-up pass only (budget)", queries=n_used)
        elif probe:
            res.update(value=64 / float(probe[0][2]) / 1e6, basis="the 64-query sizing probe only (budget)", queries=64)
        else:
            res.update(value=None, value_upper_bound=64 / max(time.perf_counter() - t0, 1e-3) / 1e6, queries=0,
                       basis="not even a 64-query pass finished inside the budget")
            continue   # torch eager with this many threads does not finish ONE tiny pass on this host: try half
        break
    if thr != ladder[0] and res.get("value") is not None:
        res["basis"] += (f"; {ladder[0]} threads (every CPU the process may run on) did not finish a 64-query pass in "
                         f"{per_try:.0f} s on this host — the largest thread count of the ladder {ladder} that does is reported")
    res["sample"] = (f"{res['queries']} queries x (sample()+pdf()), {domain} T={T}, torch.set_num_threads({res['cores']}) of {ncpu} host CPUs "
                     f"(SURVEY §8(d)); {res['basis']}; child processes, {ALL_CORES_BUDGET_S:.0f} s budget in total")
    return res


# ------------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------------
class SingleMaterial:
    """One measured BSDF, plugin-level sample() + pdf() (configs[1], configs[2])."""

    def __init__(self, name, device, rank, precision):
        import torch
        from bsdf_diffusion_sampling_amd import _lib
        from bsdf_diffusion_sampling_amd import weights as W
        from bsdf_diffusion_sampling_amd.sampler import FlowSampler
        _, self.material, self.domain, self.n_local, self.T = WORKLOADS[name]
        self.domain, _, kind = self.domain.partition(":")   # "spherical:complex" = the 64-wide x 6 checkpoint of that material
        self.name, self.rank = name, rank
        self.smp = FlowSampler(W.load(W.shipped_path(self.material, self.domain, kind or None)), precision=precision)
        self.samplers = [self.smp]
        self.variant = _lib.PLUGIN_MEASURED
        n = self.n_local
        self.wi = make_wi(self.domain, n, 1234 + rank, device)
        # double-buffered outputs so that the gather of wavefront k may overlap the compute of k+1
        self.wo = [torch.empty((n, 3), dtype=torch.float32, device=device) for _ in range(2)]
        self.pdf_s = [torch.empty((n,), dtype=torch.float32, device=device) for _ in range(2)]
        self.pdf_p = [torch.empty((n,), dtype=torch.float32, device=device) for _ in range(2)]
        self.flops_per_pass = 2 * self.smp.flops_per_query(self.T) * n   # sample launch + pdf launch
        self.launches_per_pass = 2
        self.query_launches_per_pass = 2 * n
        self.precision = self.smp.precision
        self.last = 0
        # per-query context (include/bsdfd.h, bsdfd_context_bytes): sample() writes what depends on wi alone, pdf() of
        # the same wavefront reads it instead of recomputing the prologue; bit-identical results (tests/test_gpu_parity.py)
        self.ctx = self.smp.new_context(n) if USE_CONTEXT and self.smp.context_floats(n) * 4 <= CONTEXT_MAX_BYTES else None

    def run_pass(self, k):
        b = k & 1
        self.smp.plugin_sample(self.wi, None, T=self.T, variant=self.variant, seed=1000 + k, offset=self.rank * self.n_local,
                               out=(self.wo[b], self.pdf_s[b]), ctx_out=self.ctx)
        self.smp.plugin_pdf(self.wi, self.wo[b], T=self.T, variant=self.variant, out=self.pdf_p[b], ctx_in=self.ctx)
        self.last = b

    def result(self):
        from bsdf_diffusion_sampling_amd.sharding import pack_result
        return pack_result(self.wo[self.last], self.pdf_s[self.last])

    def check(self):
        import torch
        w, p = self.wo[self.last], self.pdf_p[self.last]
        assert torch.isfinite(w).all() and torch.isfinite(p).all()
        assert torch.allclose((w * w).sum(1), torch.ones_like(w[:, 0]), atol=1e-4)

    def loop_probe(self, T):
        self.smp.plugin_sample(self.wi, None, T=T, variant=self.variant, seed=77, out=(self.wo[0], self.pdf_s[0]))

    def config(self):
        return {"material": self.material, "domain": self.domain, "euler_steps": self.T, "tile_queries": self.smp.tile,
                "api": "plugin-level sample()+pdf() (warp + guards fused), in-kernel Philox RNG",
                "per_query_context": self.ctx is not None}


class MixedMaterials:
    """configs[3] per-GPU share: the 52 measured materials (27 disk nets at the plugin's T=4, 25 spherical at T=8),
    16 Mi queries with a uniformly random material id: one bucketing (native stable counting sort) per wavefront,
    shared by its sample() and pdf() calls, ONE segmented launch per kernel signature (2 sample + 2 pdf launches)."""

    def __init__(self, name, device, rank, precision):
        import torch
        from bsdf_diffusion_sampling_amd.materials import MaterialTable
        self.name, self.rank = name, rank
        self.n_local = WORKLOADS[name][3]
        self.tab = MaterialTable.all_measured(precision)
        self.samplers = self.tab.samplers
        g = torch.Generator().manual_seed(1 + rank)
        self.ids = torch.randint(0, len(self.tab), (self.n_local,), generator=g).to(device)
        self.wi = make_wi("spherical", self.n_local, 1234 + rank, device)  # upper-hemisphere unit vectors serve both domains
        counts = torch.bincount(self.ids, minlength=len(self.tab)).cpu().tolist()
        self.flops_per_pass = sum(2 * c * self.tab.samplers[m].flops_per_query(self.tab.T[m]) for m, c in enumerate(counts))
        self.launches_per_pass = 4
        self.query_launches_per_pass = 2 * self.n_local
        self.precision = self.tab.samplers[0].precision
        # per-query contexts of the wavefront's runs (MaterialTable.sample(ctx=)) — under the same size gate as everywhere: at 16 Mi
        # queries the records (2.3 GB) stream through HBM, which measures neutral (profiles/r04_ab/mixed_ctx.txt: fewer cycles, lower clock)
        ctx_bytes = sum(c * self.tab.samplers[m].context_floats(1 << 20) * 4 // (1 << 20) for m, c in enumerate(counts))
        self.ctx = {} if USE_CONTEXT and ctx_bytes <= CONTEXT_MAX_BYTES else None
        # the wavefronts of a step are independent: bucket + gather of wavefront k+1 and the scatter of wavefront k-1 run on
        # side streams under the flow kernels of wavefront k (materials.WavefrontPipeline); every wavefront still does all
        # five stages inside the timed region ($BSDFD_BENCH_MIXED_SERIAL=1: the stages one after the other on one stream)
        from bsdf_diffusion_sampling_amd.materials import WavefrontPipeline
        # $BSDFD_BENCH_MIXED_GATHER=1: round 4/5's form — gather of wi and scatter of the results as kernels of their own on side
        # streams; default (round 6): the flow kernels read and write lane order through the bucket permutation (bsdfd_opts.row_index)
        #   (what the pipeline itself picks up to WavefrontPipeline.DIRECT_MAX_LANES = 8 Mi lanes; at this workload's 16 Mi the
        #    gather form is 5 % faster and moves 0.7x the HBM-side bytes: profiles/r06_ab/mixed_*); $BSDFD_BENCH_MIXED_DIRECT=1 forces it
        force = True if os.environ.get("BSDFD_BENCH_MIXED_DIRECT") else (False if os.environ.get("BSDFD_BENCH_MIXED_GATHER") else None)
        self.direct = force if force is not None else self.n_local <= WavefrontPipeline.DIRECT_MAX_LANES
        self.pipe = None if os.environ.get("BSDFD_BENCH_MIXED_SERIAL") else WavefrontPipeline(self.tab, direct=force)
        self.wave = None
        self._out = None

    def run_pass(self, k):
        # one bucketing and ONE gather of the inputs per wavefront; sample() and pdf() run on the bucket-ordered arrays;
        # one scatter of the three results back to the callers' lane order
        tab = self.tab
        if self.pipe is not None:
            # (ids / wi are resident, fixed inputs: ready=False — nothing to order the side stream behind)
            self.wave = self.pipe.push(self.ids, self.wi, seed=1000 + k, offset=self.rank * self.n_local, ctx=self.ctx, ready=False)
            self._out = None
            return
        plan = tab.bucket(self.ids)
        if self.direct:
            wo, pdf = tab.sample(plan, self.wi, seed=1000 + k, offset=self.rank * self.n_local, ctx=self.ctx, direct=True)
            self._out = (wo, pdf, tab.pdf(plan, self.wi, wo, ctx=self.ctx, direct=True))
            return
        wi_b = tab.gather(plan, self.wi)
        wo_b, pdf_b = tab.sample(plan, wi_b, seed=1000 + k, offset=self.rank * self.n_local, bucketed=True, ctx=self.ctx)
        p_b = tab.pdf(plan, wi_b, wo_b, bucketed=True, ctx=self.ctx)
        self._out = tab.scatter(plan, wo_b, pdf_b, p_b)

    @property
    def out(self):
        if self._out is None and self.wave is not None:
            self._out = self.wave.result()
        return self._out

    def result(self):
        from bsdf_diffusion_sampling_amd.sharding import pack_result
        return pack_result(self.out[0], self.out[1])

    def check(self):
        import torch
        wo, pdf, p = self.out
        assert torch.isfinite(wo).all() and torch.isfinite(p).all() and torch.isfinite(pdf).all()
        assert torch.allclose((wo * wo).sum(1), torch.ones_like(wo[:, 0]), atol=1e-4)

    def config(self):
        return {"materials": len(self.tab), "domain": "27 disk + 25 spherical", "euler_steps": "4 (disk) / 8 (spherical)",
                "api": ("MaterialTable: bucket-by-material (native counting sort), segmented plugin sample()/pdf() launches that read wi "
                        "and write (wo, pdf, pdf) in lane order THROUGH the bucket permutation (bsdfd_opts.row_index) — all inside the step"
                        if self.direct else
                        "MaterialTable: bucket-by-material (native counting sort), one gather of wi, segmented plugin "
                        "sample()/pdf() launches on the bucket-ordered arrays, one scatter of (wo, pdf, pdf) back to lane order "
                        "— all inside the step"), "per_query_context": self.ctx is not None,
                "row_index": self.direct, "pipelined": self.pipe is not None,
                "pipelining": ("bucketing of wavefront k+1 on a side stream under the flow kernels of wavefront k" if self.direct else
                               "bucket + gather of wavefront k+1 and scatter of wavefront k-1 on side streams under the flow "
                               "kernels of wavefront k") + " (materials.WavefrontPipeline)" if self.pipe is not None else "none"}


class Teacher:
    """the reflow teacher sampler (64-wide x 6 net, T = 128, no Jacobian, fp16) — the reference's only
    tiny-cuda-nn call site (learning_repo_cleanup/spherical_domain_sampling.py:147-166)."""

    def __init__(self, name, device, rank, precision):
        import numpy as np
        import torch
        from bsdf_diffusion_sampling_amd import weights as W
        from bsdf_diffusion_sampling_amd.sampler import FlowSampler
        _, material, domain, self.n_local, self.T = WORKLOADS[name]
        self.name, self.rank = name, rank
        fw = W.load(W.shipped_path(material, domain, "complex"))
        self.smp = FlowSampler(fw, precision="f16")
        self.samplers = [self.smp]
        g = torch.Generator().manual_seed(2 + rank)
        n = self.n_local
        u = torch.rand(n, 2, generator=g)
        self.cond = torch.stack([1.5 * u[:, 0], (2 * u[:, 1] - 1) * np.pi], 1).float().to(device)
        self.x0 = torch.stack([0.7 + 0.3 * torch.randn(n, generator=g), (2 * torch.rand(n, generator=g) - 1) * np.pi], 1).float().to(device)
        w = fw.width
        self.flops_per_pass = n * self.T * 2 * (fw.in_dim * w + (fw.n_hidden - 1) * w * w + 2 * w)  # forward only
        self.launches_per_pass = 1
        self.query_launches_per_pass = n
        self.precision = "f16"
        self.x = None

    def run_pass(self, k):
        self.x = self.smp.flow_samples_only(self.cond, self.x0, T=self.T)

    def loop_probe(self, T):
        self.smp.flow_samples_only(self.cond, self.x0, T=T)

    def result(self):
        return self.x

    def check(self):
        import torch
        assert torch.isfinite(self.x).all()

    def config(self):
        return {"net": "64 x 6 (brdf_diffusion_network_complex)", "euler_steps": self.T, "api": "bsdfd_flow_samples_only, fp16, no Jacobian",
                "tile_queries": self.smp.tile_samples_only}


def make_workload(name, device, rank, precision):
    kind = WORKLOADS[name][0]
    return {"single": SingleMaterial, "mixed": MixedMaterials, "teacher": Teacher}[kind](name, device, rank, precision)


# ------------------------------------------------------------------------------------------------------
# measurement helpers
# ------------------------------------------------------------------------------------------------------
def profiling(wl, on):
    for s in wl.samplers:
        s.set_profiling(on)


def profile_read(wl):
    n_tot, ms_tot = 0, 0.0
    for s in wl.samplers:
        n, ms = s.profile_read()
        n_tot += n
        ms_tot += ms
    return n_tot, ms_tot


def profile_read_by_op(wl):
    """{kind: (launches, total kernel ms)} of the profiled launches, by kind of launch (bsdfd_profile_read_op)."""
    out = {}
    for kind in ("sample", "pdf", "samples_only", "sample_pdf"):
        n_tot, ms_tot = 0, 0.0
        for s in wl.samplers:
            n, ms = s.profile_read_op(kind)
            n_tot += n
            ms_tot += ms
        if n_tot:
            out[kind] = (n_tot, ms_tot)
    return out


def profile_clock_mhz(wl):
    """Shader clock (MHz) the chip sustained under the profiled launches themselves (bsdfd_profile_clock_mhz: the waves' own
    shader-cycle over wall-clock counters), launch-time-weighted over the handles of the workload."""
    num, den = 0.0, 0.0
    for s in wl.samplers:
        n, ms = s.profile_read()
        mhz = s.profile_clock_mhz()
        if n and mhz > 0:
            num += mhz * ms
            den += ms
    return num / den if den > 0 else None


def settle(wl, ms):
    """Setup, untimed: leave the idle power state (tools/ramp.py: the first ~30 ms after idle the same kernel takes
    645 us instead of 545 us); results are discarded."""
    import torch
    t0 = time.perf_counter()
    k = 0
    while (time.perf_counter() - t0) * 1e3 < ms:
        wl.run_pass(k)
        torch.cuda.synchronize()
        k += 1


def pass_seconds(wl, reps=3):
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(reps):
        wl.run_pass(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def kernel_source_sha256():
    """Fingerprint of the flow kernels' source (`_lib.kernel_source_sha256`): the committed profile summaries (HBM traffic from
    the PMC passes, the instruction-issue model from the ISA) record it, and bench.py refuses them (null) when the kernels have
    changed since."""
    from bsdf_diffusion_sampling_amd import _lib
    return _lib.kernel_source_sha256()


def profile_lookup(fname, workload):
    """(entry or None, provenance dict) of profiles/<fname> for `workload`; stale or missing -> (None, why)."""
    path = os.path.join(ROOT, "profiles", fname)
    try:
        doc = json.load(open(path))
    except Exception as exc:
        return None, {"file": f"profiles/{fname}", "status": f"unreadable: {exc!r}"}
    meta = doc.get("_meta", {})
    prov = {"file": f"profiles/{fname}", "git": meta.get("git"), "kernel_source_sha256": meta.get("kernel_source_sha256")}
    if meta.get("kernel_source_sha256") != kernel_source_sha256():
        prov["status"] = "stale: the flow kernels' source (csrc/bsdfd.hip, flow32.hip, flow_dev.h) changed since this profile was taken; figure withheld"
        return None, prov
    prov["status"] = "current"
    return doc.get(workload), prov


def tile_key(workload, tile_q):
    """Key of a workload's entry in the looked-up profiles: the 16-query-tile kernels of the nets whose default is the 32-query
    family are filed as `<workload>@tile16` (runs with BSDFD_TILE=16)."""
    return workload + "@tile16" if tile_q == 16 and workload in ("disk_1Mi_T8", "disk_1Mi_T4", "spherical_16Mi_T8", "teacher_64x6_4Mi_T128") else workload


def isa_model(workload):
    """Instruction-issue model of the Euler-step loop of the dominant kernel: profiles/isa_mix_latest.json, written by
    tools/isa_mix.py from the assembly of the shipped build (MFMA + VALU issue cycles per 16-query tile and step)."""
    return profile_lookup("isa_mix_latest.json", workload)


def run_secondary(name, device, precision):
    """One secondary workload, >= ~0.3 s of timed passes, kernel time from HIP events on the launch stream."""
    import torch
    wl = make_workload(name, device, 0, precision)
    settle(wl, 60.0)
    t_pass = pass_seconds(wl, 2)
    reps = max(2, int(math.ceil(0.3 / max(t_pass, 1e-6))))
    torch.cuda.synchronize()
    profiling(wl, True)
    t0 = time.perf_counter()
    for k in range(reps):
        wl.run_pass(10 + k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_launch, kern_ms = profile_read(wl)
    mhz = profile_clock_mhz(wl)  # the clock of the timed launches themselves (in-kernel shader-cycle stamps / HIP-event time)
    profiling(wl, False)
    wl.check()
    out = {"workload": name, "value": wl.n_local * reps / dt / 1e6, "unit": "Msamples/s", "passes": reps,
           "ms_per_pass": dt / reps * 1e3, "kernel_ms_per_pass": kern_ms / reps, "launches_per_pass": n_launch / reps,
           "avg_launch_ms": kern_ms / max(n_launch, 1), "shader_clock_mhz": mhz,
           "shader_clock_basis": "the timed launches' own shader-cycle / wall-clock counters, every wave (bsdfd_profile_clock_mhz)",
           "algorithmic_TFLOPs_wall": wl.flops_per_pass * reps / dt / 1e12,
           "frac": wl.flops_per_pass * reps / (kern_ms * 1e-3) / 1e12 / PEAK_FP16_MFMA_TFLOPS,
           "frac_basis": "algorithmic flop / summed flow-kernel time (HIP events) / 2500 TFLOP/s",
           "precision": wl.precision, "config": wl.config()}
    if hasattr(wl, "loop_probe"):
        try:
            out["issue_bound"] = secondary_issue_bound(name, wl)
        except Exception as exc:   # a side figure never breaks the line
            out["issue_bound"] = {"error": repr(exc)}
    try:
        out["board"] = board_energy(wl, seconds=1.0)
        out["board"].pop("launches", None)
    except Exception as exc:
        out["board"] = {"error": repr(exc)}
    out["joule_per_Mquery"], out["socket_power_w"] = out["board"].get("joule_per_Mquery"), out["board"].get("socket_power_w")
    del wl
    torch.cuda.empty_cache()
    return out


def board_energy(wl, seconds=1.5):
    """Energy of a workload, outside the timed region (bsdf_diffusion_sampling_amd/power.py): its passes — the same sample() + pdf()
    launches `value` is made of — run back to back for `seconds` while the board's socket power and shader clock are polled (sysfs
    hwmon every 20 ms; rocm-smi where that is absent).  joule_per_Mquery = median W x elapsed s / (queries / 1e6): the flow kernels
    sit at the board's power limit, so THIS is what an optimisation has to lower — time follows it."""
    import torch
    from bsdf_diffusion_sampling_amd.power import energy_probe
    props = torch.cuda.get_device_properties(torch.cuda.current_device())
    res = energy_probe(wl.run_pass, wl.n_local, seconds=seconds, sync=torch.cuda.synchronize, pci_bus_id=getattr(props, "pci_bus_id", None))
    res["launches"] = (res["calls"] + 4) * wl.launches_per_pass
    res["basis"] = (f"passes of the workload back to back for {seconds} s behind the timed region; socket power = median of the polled "
                    "samples; MI355X board limit 1 400 W, boost clock 2 400 MHz")
    return res


def secondary_issue_bound(name, wl):
    """Euler-step cost of a secondary workload's kernel in shader cycles per (tile x step) — the same launch at T and at 2T, each
    converted at the clock its own launches ran at; the per-query prologue cancels — next to the instruction-issue model of that
    kernel's loop (tools/isa_mix.py on the shipped build, profiles/isa_mix_latest.json)."""
    import torch
    # (T, 2T) as for the judged workload; the 128-step teacher (44 ms per launch) uses (T/2, T)
    t_lo, t_hi = (wl.T // 2, wl.T) if wl.T >= 64 else (wl.T, 2 * wl.T)
    cyc = {}
    for TT in (t_lo, t_hi):
        for _ in range(3):
            wl.loop_probe(TT)
        torch.cuda.synchronize()
        profiling(wl, True)
        for _ in range(6):
            wl.loop_probe(TT)
        n, ms = profile_read(wl)
        mhz = profile_clock_mhz(wl)
        profiling(wl, False)
        cyc[TT] = (ms / max(n, 1), mhz)
    n_simd = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count * 4
    tile_q = wl.smp.tile_samples_only if isinstance(wl, Teacher) else wl.smp.tile
    tiles = wl.n_local / tile_q
    meas = (cyc[t_hi][0] * cyc[t_hi][1] - cyc[t_lo][0] * cyc[t_lo][1]) / (t_hi - t_lo) * 1e-3 * 1e6 * n_simd / tiles
    ib = {"tile_queries": tile_q, "measured_loop_cycles_per_tile_step": meas,
          "loop_basis": f"(launch at T={t_hi} minus at T={t_lo}) / {t_hi - t_lo}: {cyc[t_hi][0]:.4f} ms @ {cyc[t_hi][1]:.0f} MHz, "
                        f"{cyc[t_lo][0]:.4f} ms @ {cyc[t_lo][1]:.0f} MHz"}
    mdl, prov = isa_model(tile_key(name, tile_q))
    ib["model_source"] = prov
    if mdl and mdl.get("tile_queries", 16) == tile_q:
        ib.update({"model_issue_cycles_per_tile_step": mdl["issue_cycles_total"], "model_mfma_cycles": mdl["issue_cycles_mfma"],
                   "model_valu_cycles": mdl["issue_cycles_valu"], "n_mfma": mdl["n_mfma"], "n_valu": mdl["n_valu"], "n_trans": mdl.get("n_trans"),
                   "frac_of_issue_bound": mdl["issue_cycles_total"] / meas})
    return ib


# ------------------------------------------------------------------------------------------------------
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_children(a):
    """`python bench.py --gpus N` without a launcher environment: start the N ranks as CHILD processes of
    torch.distributed.run.  This parent never touches the GPU (no torch.cuda call, no HIP call: the library is only
    COMPILED here, so the ranks cannot race on the build), never exec()s, and exits with the children's code."""
    from bsdf_diffusion_sampling_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build(verbose=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["BSDFD_BENCH_CHILD"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def worker(a):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before the HIP runtime initialises (RCCL needs dmabuf IPC)
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    # `multi`: run the N > 1 control flow (process group, barriers, the three gather modes).  Always for N > 1; at N = 1 only
    # when BSDFD_BENCH_FORCE_PG=1 — a one-rank process group over the REAL backend, which is how the RCCL branch of this file
    # (communicator set-up, device-side gather, all-reduce of the step size) is exercised on a 1-GPU box
    # (tests/test_gpu_bench.py::test_bench_rccl_branch_with_one_rank)
    multi = world > 1 or os.environ.get("BSDFD_BENCH_FORCE_PG") == "1"
    # stdout carries the ONE JSON line and nothing else: native libraries write there too (RCCL prints a version banner at
    # communicator set-up on this image), so fd 1 is pointed at stderr for the rest of the process and the line goes to the
    # saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # one process per GPU; BSDFD_BENCH_BACKEND=gloo is a TEST hook (several ranks sharing the one GPU of a 1-GPU box,
    # gather staged through host memory) that exercises the N>1 control flow without RCCL
    backend = os.environ.get("BSDFD_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if n_dev < 1:
        raise SystemExit("bench.py: no GPU visible (torch.cuda.device_count() == 0); there is no CPU path")
    if world > 1 and backend == "nccl" and n_dev < local_world:
        # one process per GPU is the contract: never let two RCCL ranks share a device silently
        raise SystemExit(f"bench.py: --gpus {a.gpus} needs {local_world} GPUs on this node but only {n_dev} are visible "
                         f"(rank {rank}); refusing to oversubscribe under the RCCL backend")
    dev_index = local_rank % n_dev  # (only the gloo test hook ever wraps: several ranks sharing the one GPU of a 1-GPU box)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if multi and world == 1:  # forced one-rank group without a launcher: supply the rendezvous ourselves
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if multi:
        import datetime
        tmo = datetime.timedelta(seconds=float(os.environ.get("BSDFD_BENCH_PG_TIMEOUT_S", "180")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
    stage = device if backend == "nccl" else torch.device("cpu")

    from bsdf_diffusion_sampling_amd import _lib
    # clean checkout under an external launcher: rank 0 compiles (to a temporary name, renamed into place), every
    # rank passes the same barrier whether or not a build was needed
    if rank == 0 and not os.path.exists(_lib.LIB_PATH):
        _lib.build(verbose=True)
    if multi:
        dist.barrier()

    wl = make_workload(a.workload, device, rank, a.precision)
    n_local, n_total = wl.n_local, wl.n_local * world

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- setup: clock settle, then size a step so that the timed region lasts >= MIN_TIMED_S ----
    settle(wl, a.settle_ms)
    t_pass = pass_seconds(wl)
    R = a.passes_per_step or max(1, int(math.ceil(MIN_TIMED_S / (max(a.steps, 1) * max(t_pass, 1e-6)))))
    if multi:
        t = torch.tensor([R], dtype=torch.int64, device=stage)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        R = int(t.item())

    # root's gather buffers (world x [n_local, 4] floats: 2 GiB for mixed_16Mi at 8 ranks) are allocated on FIRST USE — a run with
    # --gather none never pays for them
    gather_buf = []
    comm = torch.cuda.Stream(device) if multi else None

    def gather_dst():
        if rank != 0:
            return None
        if not gather_buf:
            gather_buf.extend(torch.empty((n_local, 4), dtype=torch.float32, device=stage) for _ in range(world))
        return gather_buf

    def gather_now():
        dist.gather(wl.result().to(stage), gather_dst(), dst=0)

    def region(mode, first_pass):
        """K steps of R passes; mode: 'final' (one gather of the last wavefront's results inside the region), 'none',
        'every' (every step's last wavefront gathered on a side stream, overlapped with the next step's compute)."""
        pending = None
        fence()
        t0 = time.perf_counter()
        for s in range(a.steps):
            for r in range(R):
                wl.run_pass(first_pass + s * R + r)
            if mode == "every":
                res = wl.result()  # packed on the compute stream: the double buffers may be overwritten afterwards
                ev = torch.cuda.Event()
                ev.record()
                with torch.cuda.stream(comm):
                    comm.wait_event(ev)
                    dist.gather(res.to(stage), gather_dst(), dst=0)
                    res.record_stream(comm)
                pending = True
        if mode == "final":
            gather_now()
        if pending:
            torch.cuda.current_stream().wait_stream(comm)
        fence()
        dt = time.perf_counter() - t0
        if multi:
            tt = torch.tensor([dt], dtype=torch.float64, device=stage)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    for k in range(a.warmup * R):
        wl.run_pass(k)
    if multi:
        gather_now()  # also initialises the communicator outside the timed region
    judged_mode = a.gather if multi else "none"
    profiling(wl, True)
    dt = region(judged_mode, a.warmup * R)
    n_launch, kern_ms = profile_read(wl)
    kern_mhz = profile_clock_mhz(wl)   # the clock of the judged launches themselves (in-kernel cycle stamps / HIP-event time)
    by_op = profile_read_by_op(wl)     # the same launches by kind (sample / pdf / ...)
    profiling(wl, False)
    wl.check()
    extra_regions = {}
    if multi:
        for mode in ("none", "final", "every"):
            if mode != judged_mode:
                extra_regions[mode] = region(mode, (a.warmup + a.steps) * R)

    props = torch.cuda.get_device_properties(dev_index)
    ranks_info = [{"rank": rank, "local_rank": local_rank, "device": dev_index, "name": torch.cuda.get_device_name(dev_index),
                   "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None)}]
    if multi:
        gathered = [None] * world
        dist.all_gather_object(gathered, ranks_info[0])
        ranks_info = gathered

    if rank == 0:
        queries_timed = n_total * R * a.steps
        out = {
            "metric": "Msamples/sec (sample()+pdf())",
            "value": queries_timed / dt / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "fp16-MFMA split3 (hi+lo operands, fp32 accumulate) + fp32 VALU" if wl.precision == "split3" else wl.precision,
            "data": "synthetic",
            "config": dict(wl.config(), workload=a.workload, queries_per_wavefront_per_gpu=n_local, passes_per_step=R,
                           queries_per_step=n_total * R, timed_region_s=dt, min_timed_region_s=MIN_TIMED_S,
                           precision=wl.precision, settle_ms=a.settle_ms,
                           parallelism=f"query-sharded x{world}, no data-path collective" +
                                       ("" if world == 1 else {"final": "; one RCCL gather-to-root of the final (wo,pdf) shards inside the timed region",
                                                               "every": "; RCCL gather-to-root every step, overlapped on a side stream",
                                                               "none": "; results left device-resident"}[judged_mode]),
                           ranks=ranks_info, backend=backend if multi else None,
                           rccl_ranks=world if (multi and backend == "nccl") else 0,
                           distinct_devices=len({(r_["device"], r_.get("uuid")) for r_ in ranks_info})),
        }
        if multi:
            rates = {judged_mode: queries_timed / dt / 1e6}
            rates.update({m: queries_timed / t / 1e6 for m, t in extra_regions.items()})
            out["multi_gpu"] = {"Msamples_per_s_no_gather": rates["none"], "Msamples_per_s_final_gather": rates["final"],
                                "Msamples_per_s_gather_every_step_overlapped": rates["every"], "judged": judged_mode,
                                "gather_bytes_per_rank": n_local * 16, "note": "each figure is its own timed region of the same K steps"}
        # ---- roofline of the dominant kernel: HIP events on the launch stream over the judged region ----
        avg_ms = kern_ms / max(n_launch, 1)
        flops_launch = wl.flops_per_pass / wl.launches_per_pass
        achieved = wl.flops_per_pass * R * a.steps / (kern_ms * 1e-3) / 1e12
        pmc_entry, pmc_prov = profile_lookup("pmc_latest.json", tile_key(a.workload, getattr(getattr(wl, "smp", None), "tile", 32)))
        if pmc_entry is None and pmc_prov.get("status") == "current":   # (no PMC pass of this tiling on file: the bytes do not depend on it)
            pmc_entry, pmc_prov = profile_lookup("pmc_latest.json", a.workload)
        ctx_on = getattr(wl, "ctx", None) is not None
        # (the committed PMC passes are those of the DEFAULT command, i.e. without the per-query context: withheld under --context on)
        traffic = None if ctx_on else (pmc_entry or {}).get("hbm_bytes_per_launch")
        algo_bytes = 28 * n_local
        roof = {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP16_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP16_MFMA_TFLOPS, "traffic": traffic,
                "traffic_source": dict(pmc_prov, ref=f"{pmc_prov.get('file')}@{(pmc_prov.get('kernel_source_sha256') or 'unknown')[:12]}",
                                       measured_in_this_run=False,
                                       how="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (tools/profile.sh), "
                                           "condensed by tools/summarize_profile.py; looked up, not re-measured in this run"),
                "algorithmic_bytes_per_launch": algo_bytes,
                "traffic_over_algorithmic": (traffic / algo_bytes) if traffic else None,
                "traffic_note": ("--context on: a sample launch also WRITES and a pdf launch also READS the per-query context (144 B/query "
                                 "for the 32-wide nets, 5.1x the 28 algorithmic bytes; profiles/r04_ab/: 166 MB per launch); the committed "
                                 "PMC passes are those of the default command and are withheld here") if ctx_on else
                                "wi / wo / pdf rows read and written once (28 B per query); the per-query context is off by default",
                "kernel": ("flow_kernel32 (csrc/flow32.hip, 32-query tiles)" if getattr(getattr(wl, "smp", None), "tile", 16) == 32
                           else "flow_kernel (csrc/bsdfd.hip, 16-query tiles)"), "launches": n_launch, "avg_launch_ms": avg_ms, "shader_clock_mhz": kern_mhz,
                "algorithmic_flop_per_launch": flops_launch, "queries_per_launch": n_local,
                "kernel_Msamples_per_s": wl.query_launches_per_pass * R * a.steps / (kern_ms * 1e-3) / 1e6,
                "kernel_Msamples_basis": "queries x flow-kernel launches each goes through (sample and pdf count separately) "
                                         "/ summed kernel time"}
        if isinstance(wl, SingleMaterial):
            # the two launch kinds of the timed region itself (the library keeps the event totals per kind of launch)
            split = {k: ms / n for k, (n, ms) in by_op.items()}
            roof["flow_launches_after_timed_region"] = 0   # (tools/summarize_profile.py slices the per-launch trace with it)
            roof.update({"algorithmic_flop_per_query": wl.smp.flops_per_query(wl.T), "sample_launch_ms": split.get("sample"),
                         "pdf_launch_ms": split.get("pdf"),
                         "sample_pdf_split_basis": "the timed region's own launches by kind (bsdfd_profile_read_op)",
                         "sample_Msamples_per_s": n_local / (split["sample"] * 1e-3) / 1e6 if split.get("sample") else None,
                         "pdf_Msamples_per_s": n_local / (split["pdf"] * 1e-3) / 1e6 if split.get("pdf") else None})
            # the per-query context in both call orders (outside the timed region): sample() fills / pdf() reads — the bench's
            # pass — and pdf() fills / sample() reads — the order of Mitsuba's path integrator (eval_pdf for the emitter
            # sample first, rendering/brdf_measured_disk.py:126, then sample, :59) — next to the pair without a context
            pair_ctx = wl.ctx if wl.ctx is not None else (wl.smp.new_context(n_local) if wl.smp.context_floats(n_local) * 4 <= CONTEXT_MAX_BYTES else None)
            if pair_ctx is not None:
                def pair(order, k):
                    kw_s = dict(T=wl.T, variant=wl.variant, seed=300 + k, out=(wl.wo[1], wl.pdf_s[1]))
                    kw_p = dict(T=wl.T, variant=wl.variant, out=wl.pdf_p[1])
                    if order == "sample_then_pdf":
                        wl.smp.plugin_sample(wl.wi, None, ctx_out=pair_ctx, **kw_s)
                        wl.smp.plugin_pdf(wl.wi, wl.wo[0], ctx_in=pair_ctx, **kw_p)
                    elif order == "pdf_then_sample":
                        wl.smp.plugin_pdf(wl.wi, wl.wo[0], ctx_out=pair_ctx, **kw_p)
                        wl.smp.plugin_sample(wl.wi, None, ctx_in=pair_ctx, **kw_s)
                    else:
                        wl.smp.plugin_sample(wl.wi, None, **kw_s)
                        wl.smp.plugin_pdf(wl.wi, wl.wo[0], **kw_p)
                # the three forms INTERLEAVED in rounds of 8 pairs (the chip's clock drifts for tens of ms after the host-side
                # pause behind the timed region: one form after the other measures the drift, not the forms)
                ORDERS, ROUNDS, PAIRS = ("sample_then_pdf", "pdf_then_sample", "no_context"), 4, 8
                for k in range(24):
                    pair(ORDERS[k % 3], k)
                torch.cuda.synchronize()
                profiling(wl, True)
                totals, seen = dict.fromkeys(ORDERS, 0.0), 0.0
                for rnd in range(ROUNDS):
                    for order in ORDERS:
                        for k in range(PAIRS):
                            pair(order, 24 + rnd * PAIRS + k)
                        _, ms = profile_read(wl)
                        totals[order] += ms - seen
                        seen = ms
                profiling(wl, False)
                roof["flow_launches_after_timed_region"] += 2 * (24 + 3 * ROUNDS * PAIRS)
                roof["context_pair_ms"] = dict({o: totals[o] / (ROUNDS * PAIRS) for o in ORDERS},
                                               basis=f"kernel time of one sample() + pdf() pair on the same wavefront; the three forms "
                                                     f"interleaved, {ROUNDS} rounds x {PAIRS} pairs each, outside the timed region")
                pair_ctx = None   # (frees the 144 MiB record before the secondary workloads allocate)
            # issue-bound view: measured SIMD cycles per (16-query tile x Euler step) vs the instruction-issue model
            try:
                probe_mhz = _lib.shader_clock_mhz()
                mhz = kern_mhz or probe_mhz
                n_simd = torch.cuda.get_device_properties(dev_index).multi_processor_count * 4
                tile_q = wl.smp.tile
                tiles = n_local / tile_q
                meas = avg_ms * 1e-3 * mhz * 1e6 * n_simd / (tiles * wl.T)
                # loop-only: the same sample launch at T and 2T (outside the timed region) — the per-query prologue cancels
                loop = {}
                for TT in (wl.T, 2 * wl.T):
                    for k in range(6):
                        wl.smp.plugin_sample(wl.wi, None, T=TT, variant=wl.variant, seed=5 + k, out=(wl.wo[0], wl.pdf_s[0]))
                    torch.cuda.synchronize()
                    profiling(wl, True)
                    for k in range(12):
                        wl.smp.plugin_sample(wl.wi, None, T=TT, variant=wl.variant, seed=50 + k, out=(wl.wo[0], wl.pdf_s[0]))
                    _, ms_tt = profile_read(wl)
                    loop[TT] = (ms_tt / 12, profile_clock_mhz(wl))
                    profiling(wl, False)
                roof["flow_launches_after_timed_region"] += 36
                # (each launch length is converted at the clock ITS launches ran at: the two may differ by a few per cent)
                c_long = loop[2 * wl.T][0] * (loop[2 * wl.T][1] or mhz)
                c_short = loop[wl.T][0] * (loop[wl.T][1] or mhz)
                loop_mhz = loop[2 * wl.T][1] or mhz
                meas_loop = (c_long - c_short) / wl.T * 1e-3 * 1e6 * n_simd / tiles
                ib = {"shader_clock_mhz": mhz, "simds": n_simd, "tile_queries": tile_q,
                      "shader_clock_basis": "the judged launches' own shader-cycle / wall-clock counters, every wave "
                                            "(bsdfd_profile_clock_mhz); `probe_clock_mhz` is the stand-alone probe kernel (csrc/clock.hip), "
                                            "a similar instruction mix but not the kernel itself",
                      "probe_clock_mhz": probe_mhz,
                      "measured_simd_cycles_per_tile_step": meas,
                      "measured_loop_cycles_per_tile_step": meas_loop,
                      "loop_basis": f"(sample launch at T={2 * wl.T} minus at T={wl.T}) / {wl.T}: {loop[2 * wl.T][0]:.4f} ms, {loop[wl.T][0]:.4f} ms "
                                    f"at {loop_mhz:.0f} and {(loop[wl.T][1] or mhz):.0f} MHz — the per-query prologue and epilogue cancel",
                      "note": "measured_simd_cycles = avg launch time x clock x SIMDs / (tiles x T) carries the per-query prologue's "
                              "share; measured_loop_cycles is the Euler step alone and is what the model describes.  Model = sum over "
                              "the loop's instructions of what each costs the SIMD (tools/isa_mix.py on the shipped build): in this "
                              "VALU-heavy mix MFMA time and VALU time add — a 16x16x32 MFMA 16.4 cycles, a 32x32x16 32.1, a transcendental "
                              "8.1 (tools/ubench/RESULTS.md; rounds 2-4 priced it at 11, round 5 measured ~7.5-8 in the kernels) — model/measured_loop ~ 1 means the step runs "
                              "at the hardware's issue rate for this instruction mix.  A tile is `tile_queries` queries (one wave64)."}
                mdl, mdl_prov = isa_model(tile_key(a.workload, tile_q))
                ib["model_source"] = mdl_prov
                if mdl and mdl.get("tile_queries", 16) != tile_q:
                    ib["model_source"] = dict(mdl_prov, status=f"model is of the {mdl.get('tile_queries', 16)}-query-tile kernel, the run used {tile_q}")
                    mdl = None
                if mdl:
                    ib.update({"model_issue_cycles_per_tile_step": mdl["issue_cycles_total"], "model_mfma_cycles": mdl["issue_cycles_mfma"],
                               "model_valu_cycles": mdl["issue_cycles_valu"], "n_mfma": mdl["n_mfma"], "n_valu": mdl["n_valu"],
                               "frac_of_issue_bound": mdl["issue_cycles_total"] / meas_loop,
                               "frac_of_issue_bound_basis": "model / measured_loop_cycles_per_tile_step"})
                roof["issue_bound"] = ib
            except Exception as exc:
                roof["issue_bound"] = {"error": repr(exc)}
        # the board's own view while the judged workload's passes run back to back (outside the timed region, whatever the
        # workload): the flow kernels are power-limited — socket power at the limit, shader clock below boost
        if rank == 0 and world == 1:
            try:
                roof["board"] = board_energy(wl)
                roof["flow_launches_after_timed_region"] = roof.get("flow_launches_after_timed_region", 0) + roof["board"].pop("launches", 0)
            except Exception as exc:
                roof["board"] = {"error": repr(exc)}
            roof["joule_per_Mquery"] = roof["board"].get("joule_per_Mquery")
            roof["socket_power_w"] = roof["board"].get("socket_power_w")
        out["roofline"] = roof
        if world == 1 and not a.no_secondary:
            sec = {}
            for name in SECONDARY:
                if name == a.workload:
                    continue
                try:
                    sec[name] = run_secondary(name, device, a.precision)
                except Exception as exc:  # a side figure never breaks the judged line
                    sec[name] = {"error": repr(exc)}
            out["secondary"] = sec
            # the un-fused "encoding pass" BASELINE.json asks an HBM rate for (fused, it never touches HBM):
            # positional_encoding_1 of 16 Mi conditioning rows, 8 B read + 88 B written per row
            try:
                from bsdf_diffusion_sampling_amd.encoding import positional_encoding_1
                n_enc = 1 << 24
                xe = torch.rand((n_enc, 2), device=device) * 2 - 1
                oe = torch.empty((n_enc, 22), device=device)
                for _ in range(5):
                    positional_encoding_1(xe, 5, out=oe)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    positional_encoding_1(xe, 5, out=oe)
                e1.record()
                torch.cuda.synchronize()
                enc_ms = e0.elapsed_time(e1) / 20
                out["encoding_pass"] = {"bound": "hbm", "rows": n_enc, "bytes_per_row": 96, "avg_launch_ms": enc_ms,
                                        "achieved": n_enc * 96 / (enc_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                        "frac": n_enc * 96 / (enc_ms * 1e-3) / 1e9 / 8000.0,
                                        "note": "stand-alone positional_encoding_1 (csrc/encoding.hip); inside the flow "
                                                "kernel the encoding is fused and costs no HBM traffic"}
                del xe, oe
            except Exception as exc:
                out["encoding_pass"] = {"error": repr(exc)}
        if world == 1 and not a.no_cpu_baseline and isinstance(wl, SingleMaterial):
            out["cpu_baseline"] = cpu_baseline(wl.material, wl.domain, wl.T)
        elif world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline("aniso_miro_7_rgb", "disk", 4)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if multi:
        dist.barrier()  # orderly teardown: rank 0 is still printing / timing its side figures
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="disk_1Mi_T8", choices=sorted(WORKLOADS))
    ap.add_argument("--precision", default="default", choices=["default", "f32", "split3", "f16"])
    ap.add_argument("--gather", default="final", choices=["final", "every", "none"],
                    help="N>1: which timed region is the judged `value` — 'final': one RCCL gather of the last wavefront's "
                         "(wo,pdf) shards to rank 0 inside the region (the path has no data-path collective; the final "
                         "concatenation is the only exchange); 'every': every step, overlapped on a side stream; the other "
                         "modes are timed too and reported under `multi_gpu`")
    ap.add_argument("--passes-per-step", type=int, default=0,
                    help="wavefronts per step (0 = sized at setup so that the timed region lasts >= 0.5 s)")
    ap.add_argument("--context", default=None, choices=["on", "off"],
                    help="per-query context hand-over from sample() to pdf() of the same wavefront (default off: it buys 1-2 %% of a pair "
                         "for 5x the HBM traffic; roofline.context_pair_ms reports both forms either way)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="setup: keep the GPU busy with the hot path for this long before the W warm-up steps, so "
                         "that the timed region does not start on an idle-clocked chip")
    a = ap.parse_args()
    if a.context is not None:  # also reaches the self-launched ranks (children inherit the environment)
        global USE_CONTEXT
        USE_CONTEXT = a.context == "on"
        os.environ["BSDFD_BENCH_CONTEXT"] = "1" if USE_CONTEXT else "0"
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "RANK" not in os.environ:
        return launch_children(a)
    return worker(a)


if __name__ == "__main__":
    sys.exit(main())
