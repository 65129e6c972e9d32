#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (written by tools/profile.sh on the GPU box) into
the committed, judged summaries under profiles/: <name>_kernel_stats.csv (rocprofv3 --kernel-trace
--stats), <name>_pmc.json (per-launch counter means of the flow kernel) and pmc_latest.json
(HBM bytes per launch, read by bench.py for roofline.traffic)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, name = sys.argv[1], sys.argv[2]   # tools/summarize_profile.py <prof tag> <output name> [workload]
    workload = sys.argv[3] if len(sys.argv) > 3 else "disk_1Mi_T8"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, f"{name}_kernel_stats.csv"))
    pmc = {}
    meta = {}
    for f in sorted(glob.glob(os.path.join(src, "pmc_*", "pmc_counter_collection.csv"))):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "flow_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
                meta = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
                                          "SGPR_Count", "Scratch_Size") if k in r}
        # the counters of the TIMED REGION's launches only, as for the trace below: behind it bench.py issues launches of other
        # kinds (the context pair in both orders, the T / 2T pair of the issue-bound entry) — the pass's own JSON line says how many
        n_timed = n_after = None
        try:
            line = json.loads([l for l in open(os.path.dirname(f) + ".log") if l.startswith("{")][-1])
            n_timed = int(line["roofline"]["launches"])
            n_after = int(line["roofline"].get("flow_launches_after_timed_region", 0))
        except Exception:
            pass
        for k, v in agg.items():
            v = [x for _, x in sorted(v)]
            sel = v[len(v) - n_after - n_timed:len(v) - n_after] if n_timed and len(v) >= n_timed + n_after else v
            pmc[k] = {"launches": len(sel), "mean_per_launch": sum(sel) / len(sel),
                      "of": "the timed region's launches" if sel is not v else "every flow-kernel launch of the pass"}
    out = {"tag": tag, "workload": workload, "kernel": meta, "counters": pmc}
    # kernel duration from the trace
    for r in csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))):
        if "flow_kernel" in r["Name"]:
            out["kernel_trace"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                                   "max_ns": float(r["MaxNs"]), "percentage": float(r["Percentage"])}
    # the stats average covers EVERY launch of the command (clock-settle phase, warm-up, timed region, per-kind
    # split); the per-launch trace lets us average exactly the timed region's launches as bench.py's events do
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
    # the traced command's own JSON line says how many wavefronts a step was sized to (>= 0.5 s timed region)
    try:
        line = [l for l in open(os.path.join(src, "trace.log")) if l.startswith("{")][-1]
        bench = json.loads(line)
        steps = bench["steps"] * bench["config"].get("passes_per_step", 1)
        out["bench_line"] = {k: bench[k] for k in ("value", "ms_per_step", "steps") if k in bench}
        out["bench_line"]["roofline"] = {k: bench["roofline"].get(k) for k in ("avg_launch_ms", "frac", "launches")}
        out["bench_line"]["passes_per_step"] = bench["config"].get("passes_per_step", 1)
        n_timed = int(bench["roofline"].get("launches", 2 * steps))
        n_after = int(bench["roofline"].get("flow_launches_after_timed_region", 0))
    except Exception:
        n_timed, n_after = 2 * steps, 10
    tr = glob.glob(os.path.join(src, "trace", "trace_kernel_trace.csv"))
    if tr and "kernel_trace" in out:
        rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(tr[0]))
                if "flow_kernel" in r["Kernel_Name"]]
        rows.sort()
        # bench.py: ..., the timed region's launches, then the per-kind split and the T / 2T pair of the issue-bound entry
        timed = rows[-(n_timed + n_after):len(rows) - n_after]
        if timed:
            out["kernel_trace"]["timed_region_launches"] = len(timed)
            out["kernel_trace"]["timed_region_avg_ns"] = sum(e - b for b, e in timed) / len(timed)
    # shader clock three ways in ONE run (the GRBM pass): GRBM_GUI_ACTIVE / 8 XCDs / the dispatch's own duration, the flow
    # kernel's in-kernel figure (bsdfd_profile_clock_mhz) and the stand-alone probe, both from the JSON line that run printed
    try:
        f = glob.glob(os.path.join(src, "pmc_GRBM*", "pmc_counter_collection.csv"))[0]
        mhz = [float(r["Counter_Value"]) / 8.0 / max(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), 1) * 1e3
               for r in csv.DictReader(open(f)) if "flow_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
        clk = {"grbm_gui_active_mhz": sum(mhz) / len(mhz), "launches": len(mhz),
               "basis": "GRBM_GUI_ACTIVE / 8 XCDs / (End_Timestamp - Start_Timestamp) per flow-kernel dispatch of the GRBM pass"}
        logf = glob.glob(os.path.join(src, "pmc_GRBM*.log"))[0]
        line = json.loads([l for l in open(logf) if l.startswith("{")][-1])
        ib = line["roofline"].get("issue_bound", {})
        clk["in_kernel_mhz_same_run"] = ib.get("shader_clock_mhz") or line["roofline"].get("shader_clock_mhz")
        clk["probe_mhz_same_run"] = ib.get("probe_clock_mhz")
        if clk["in_kernel_mhz_same_run"]:
            clk["in_kernel_over_grbm"] = clk["in_kernel_mhz_same_run"] / clk["grbm_gui_active_mhz"]
        if clk["probe_mhz_same_run"]:
            clk["probe_over_grbm"] = clk["probe_mhz_same_run"] / clk["grbm_gui_active_mhz"]
        out["shader_clock"] = clk
    except Exception as exc:  # (a trace-only directory has no GRBM pass)
        out["shader_clock"] = {"error": repr(exc)}
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        # rocprofv3 reports KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B: double it
        # (MI355X_MICROARCH.md §HBM; calibrated on this kernel's known 12/24 B-per-query reads)
        fetch = pmc["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2
        write = pmc["WRITE_SIZE"]["mean_per_launch"] * 1024
        out["hbm_bytes_per_launch"] = fetch + write
        out["hbm_fetch_bytes_corrected"] = fetch
        out["hbm_write_bytes"] = write
    json.dump(out, open(os.path.join(dst, f"{name}_pmc.json"), "w"), indent=1)
    latest = {}
    lp = os.path.join(dst, "pmc_latest.json")
    if os.path.exists(lp):
        latest = json.load(open(lp))
    if "hbm_bytes_per_launch" in out:
        import subprocess
        sys.path.insert(0, ROOT)
        from bsdf_diffusion_sampling_amd import _lib
        sha = _lib.kernel_source_sha256()
        git = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        meta = latest.get("_meta", {})
        if meta.get("kernel_source_sha256") != sha:   # entries of an older kernel do not survive next to new ones
            latest = {}
        latest["_meta"] = {"kernel_source_sha256": sha, "git": git, "tool": "tools/profile.sh + tools/summarize_profile.py"}
        latest[workload] = {"hbm_bytes_per_launch": out["hbm_bytes_per_launch"], "source": f"{name}_pmc.json",
                            "hbm_fetch_bytes_corrected": out["hbm_fetch_bytes_corrected"], "hbm_write_bytes": out["hbm_write_bytes"]}
        json.dump(latest, open(lp, "w"), indent=1)
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    main()
