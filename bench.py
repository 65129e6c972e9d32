#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the neural-BSDF hot path, sample() + pdf().

A "step" is one pass of the hot path over one batch of synthetic shading queries:
``MyBSDF.sample`` (tensor core: warp + T Euler steps + Jacobian + guards, in-kernel RNG)
followed by ``MyBSDF.pdf`` on the produced directions — BASELINE.json configs[1]:
single measured BSDF (aniso_miro_7_rgb), disk-domain net, 1 Mi queries, 8 denoise steps,
per GPU (weak scaling: every rank gets its own 1 Mi-query sub-batch).  Inputs are resident
in HBM before the timed region.  There is no data-path collective; with N > 1 ranks the final
(wo, pdf) shards are concatenated on rank 0 with one RCCL gather inside the timed region
(`--gather every` gathers every step's shard on a side stream, overlapped with compute).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 50 --warmup 5

Prints ONE JSON line on rank 0 (see the task contract): ``value`` = whole-job
Msamples/s (queries through sample()+pdf() per second, all ranks), plus
``roofline`` (fused flow kernel vs the dense fp16-MFMA peak; HIP events on the launch
stream) and ``cpu_baseline`` (the torch-eager port of the reference's CPU path, timed on
this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_FP16_MFMA_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"

WORKLOADS = {
    # name: (material, domain, per-GPU queries, T)
    "disk_1Mi_T8": ("aniso_miro_7_rgb", "disk", 1 << 20, 8),        # BASELINE.json configs[1]
    "disk_1Mi_T4": ("aniso_miro_7_rgb", "disk", 1 << 20, 4),        # plugin default T
    "spherical_16Mi_T8": ("aniso_miro_7_rgb", "spherical", 1 << 24, 8),  # configs[2]
}


def make_wi(domain, n, seed, device):
    """disk — uniform on the disk of radius 0.95; spherical — theta_i ~ U(0,1.5),
    phi_i ~ U(-pi,pi); both handed over as unit vectors [N,3] (the warps are inside the timed region)."""
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


def cpu_baseline(material, domain, T, budget_n=262144, reps=3):
    """The reference's CPU PyTorch path, restated op-for-op (oracle/torch_eager_port.py — validated
    bit-identical to the reference and within ~15 % of its wall time, tests/golden/cpu_timing.json)."""
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

    # torch eager on many-core hosts is NOT fastest with all cores (256 threads were 100x slower
    # than 32 on the GPU box: tiny per-op work, OpenMP fork/join dominates), so pick the thread
    # count that maximises throughput on a small calibration batch, growing until it stops helping.
    torch.manual_seed(1234)
    calib = make_cond(16384)
    best_thr, best_t = 1, float("inf")
    thr = 4
    while thr <= ncpu:
        torch.set_num_threads(thr)
        one_pass(calib)
        t = sum(one_pass(calib))
        if t < best_t:
            best_thr, best_t = thr, t
        elif t > 1.5 * best_t:
            break
        thr *= 2
    cores = best_thr
    torch.set_num_threads(cores)
    cond = make_cond(budget_n)
    one_pass(cond)  # warm-up
    ts, tp = [], []
    for _ in range(reps):
        a, b = one_pass(cond)
        ts.append(a)
        tp.append(b)
    t_s, t_p = float(np.median(ts)), float(np.median(tp))
    return {"value": budget_n / (t_s + t_p) / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"{budget_n} queries x (sample()+pdf()), {domain} T={T}, torch {torch.__version__} eager fp32 "
                      f"with autograd (2 backward/step), 1 warm-up + median of {reps}; threads chosen by calibration "
                      f"({cores} of {ncpu} host CPUs)",
            "sample_Msps": budget_n / t_s / 1e6, "pdf_Msps": budget_n / t_p / 1e6}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="disk_1Mi_T8", choices=sorted(WORKLOADS))
    ap.add_argument("--precision", default="default", choices=["default", "f32", "split3", "f16"])
    ap.add_argument("--gather", default="final", choices=["final", "every", "none"],
                    help="N>1: RCCL gather of the (wo,pdf) shards to rank 0 — 'final': the last step's results "
                         "once, inside the timed region (the path has no data-path collective; the final "
                         "concatenation is the only exchange); 'every': every step, overlapped on a side stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="setup: keep the GPU busy with the hot path for this long before the W warm-up steps, so "
                         "that the timed region does not start on an idle-clocked chip (tools/ramp.py: the first "
                         "~30 ms after idle the same kernel takes 645 us instead of 545 us)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    import torch.distributed as dist
    # one process per GPU; BSDFD_BENCH_BACKEND=gloo is a TEST hook (several ranks sharing one GPU on a
    # 1-GPU box, gather staged through host memory) to exercise the N>1 control flow without RCCL
    backend = os.environ.get("BSDFD_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    from bsdf_diffusion_sampling_amd import _lib
    from bsdf_diffusion_sampling_amd import weights as W
    from bsdf_diffusion_sampling_amd.sampler import FlowSampler
    from bsdf_diffusion_sampling_amd.sharding import pack_result

    if not os.path.exists(_lib.LIB_PATH):  # clean checkout: compile the HIP library (rank 0), others wait
        if rank == 0:
            _lib.build(verbose=True)
        if world > 1:
            dist.barrier()

    material, domain, n_local, T = WORKLOADS[a.workload]
    n_total = n_local * world
    lo = rank * n_local
    fw = W.load(W.shipped_path(material, domain))
    smp = FlowSampler(fw, precision=a.precision)
    wi = make_wi(domain, n_local, 1234 + rank, device)
    # double-buffered outputs so the gather of step k overlaps the compute of step k+1
    wo = [torch.empty((n_local, 3), dtype=torch.float32, device=device) for _ in range(2)]
    pdf_s = [torch.empty((n_local,), dtype=torch.float32, device=device) for _ in range(2)]
    pdf_p = [torch.empty((n_local,), dtype=torch.float32, device=device) for _ in range(2)]
    do_gather = world > 1 and a.gather == "every"
    final_gather = world > 1 and a.gather == "final"
    comm = torch.cuda.Stream(device) if do_gather else None
    stage = device if backend == "nccl" else torch.device("cpu")
    gather_out = ([torch.empty((n_local, 4), dtype=torch.float32, device=stage) for _ in range(world)]
                  if ((do_gather or final_gather) and rank == 0) else None)
    done_ev = [torch.cuda.Event(), torch.cuda.Event()]
    free_ev = [None, None]
    variant = _lib.PLUGIN_MEASURED

    def step(k):
        b = k & 1
        if free_ev[b] is not None:
            torch.cuda.current_stream().wait_event(free_ev[b])  # buffer b still being gathered
        smp.plugin_sample(wi, None, T=T, variant=variant, seed=1000 + k, offset=lo, out=(wo[b], pdf_s[b]))
        smp.plugin_pdf(wi, wo[b], T=T, variant=variant, out=pdf_p[b])
        if do_gather:
            done_ev[b].record()
            with torch.cuda.stream(comm):
                comm.wait_event(done_ev[b])
                dist.gather(pack_result(wo[b], pdf_s[b]).to(stage), gather_out, dst=0)
                free_ev[b] = torch.cuda.Event()
                free_ev[b].record()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def concat_final(k_last):
        b = k_last & 1
        dist.gather(pack_result(wo[b], pdf_s[b]).to(stage), gather_out, dst=0)

    # setup, untimed: leave the idle power state (see --settle-ms); results are discarded
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < a.settle_ms:
        smp.plugin_sample(wi, None, T=T, variant=variant, seed=1, offset=lo, out=(wo[0], pdf_s[0]))
        smp.plugin_pdf(wi, wo[0], T=T, variant=variant, out=pdf_p[0])
        torch.cuda.synchronize()
    for k in range(a.warmup):
        step(k)
    if final_gather:
        concat_final(max(a.warmup - 1, 0))  # also initialises the RCCL communicator outside the timed region
    fence()
    smp.set_profiling(True)
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(a.warmup + k)
    if final_gather:
        concat_final(a.warmup + a.steps - 1)
    fence()
    dt = time.perf_counter() - t0
    n_launch, kern_ms = smp.profile_read()
    # informational split of the two launch kinds (outside the timed region)
    smp.set_profiling(True)
    for k in range(5):
        smp.plugin_sample(wi, None, T=T, variant=variant, seed=77 + k, offset=lo, out=(wo[0], pdf_s[0]))
    _, ms_sample = smp.profile_read()
    smp.set_profiling(True)
    for k in range(5):
        smp.plugin_pdf(wi, wo[0], T=T, variant=variant, out=pdf_p[0])
    _, ms_pdf = smp.profile_read()
    ms_sample, ms_pdf = ms_sample / 5, ms_pdf / 5
    smp.set_profiling(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=stage)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # sanity on the produced data (outside the timed region): finite, unit vectors
    w = wo[(a.warmup + a.steps - 1) & 1]
    assert torch.isfinite(w).all() and torch.isfinite(pdf_p[(a.warmup + a.steps - 1) & 1]).all()
    assert torch.allclose((w * w).sum(1), torch.ones_like(w[:, 0]), atol=1e-4)

    if rank == 0:
        flops_launch = smp.flops_per_query(T) * n_local
        avg_ms = kern_ms / max(n_launch, 1)
        achieved = flops_launch / (avg_ms * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(a.workload, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Msamples/sec (sample()+pdf())",
            "value": n_total * a.steps / dt / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "fp16-MFMA split3 (hi+lo operands, fp32 accumulate) + fp32 VALU" if smp.precision == "split3"
                     else smp.precision,
            "data": "synthetic",
            "config": {"workload": a.workload, "material": material, "domain": domain,
                       "queries_per_gpu": n_local, "global_queries": n_total, "euler_steps": T,
                       "api": "plugin-level sample()+pdf() (warp + guards fused), in-kernel Philox RNG",
                       "parallelism": f"query-sharded x{world}" + (", RCCL gather-to-root every step (overlapped)" if do_gather else
                                                                   ", RCCL gather-to-root of the final results" if final_gather else ""),
                       "precision": smp.precision, "settle_ms": a.settle_ms},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP16_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP16_MFMA_TFLOPS, "traffic": traffic,
                         "kernel": "flow_kernel", "launches": n_launch, "avg_launch_ms": avg_ms,
                         "algorithmic_flop_per_query": smp.flops_per_query(T), "queries_per_launch": n_local,
                         "kernel_Msamples_per_s": n_local / (avg_ms * 1e-3) / 1e6,
                         "sample_launch_ms": ms_sample, "pdf_launch_ms": ms_pdf,
                         "sample_Msamples_per_s": n_local / (ms_sample * 1e-3) / 1e6,
                         "pdf_Msamples_per_s": n_local / (ms_pdf * 1e-3) / 1e6},
        }
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
        except Exception as exc:  # never let the side figure break the judged line
            out["encoding_pass"] = {"error": repr(exc)}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(material, domain, T)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()  # orderly teardown: rank 0 is still printing / timing its side figures
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
