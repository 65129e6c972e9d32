#!/usr/bin/env python3
"""Where a wave spends its time inside a flow kernel launch, phase by phase (tools only: the product source is not touched).

    python tools/phase_clock.py build            # patches a COPY of csrc/bsdfd.hip with s_memtime stamps -> build_ab/lib_phaseclk.so
    python tools/phase_clock.py run [N]          # on the GPU box: sample / pdf / pdf with context, disk and spherical, T = 4 and 8

Every wave keeps six accumulators of shader cycles (s_memtime) and adds them to words 2..7 of its clock slot at the end:
  gap    : previous tile's end (or kernel start) -> this tile's indices are known (tile-loop overhead, LDS image copy of the workgroup)
  inputs : -> the query's inputs have arrived and cart_to_spher is done (global-load latency)
  wi-pro : -> encoding, conditioning term, base net done (or the context has arrived)
  draw   : -> initial state and its density are known
  steps  : -> the T Euler steps
  epi    : -> the results are stored
A stamp is an asm statement that names the value the phase ends with as an operand, so the compiler can move neither the phase's
work behind it nor the next phase's work in front of it.  With 3 waves per SIMD a wave's elapsed cycles include its neighbours'
issue slots: the SHARES are meaningful, and the sum over waves / waves per SIMD approximates SIMD time."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "bsdf_diffusion_sampling_amd", "csrc", "bsdfd.hip")
TMP = os.path.join(ROOT, "bsdf_diffusion_sampling_amd", "csrc", "bsdfd_phaseclk_tmp.hip")
LIB = os.path.join(ROOT, "build_ab", "lib_phaseclk.so")
PHASES = ["gap", "inputs", "wi-pro", "draw", "steps", "epi"]


def patch(s):
    def rep(old, new):
        nonlocal s
        assert s.count(old) == 1, (s.count(old), old[:60])
        s = s.replace(old, new)
    rep("template <int DOMAIN, int NM, int PREC, bool JAC, int NH, bool FUSED>\n__global__",
        '#define PH_STAMP(idx, dep) do { if (p.clk) { unsigned long long t_; '
        'asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "+v"(dep) : : "memory"); '
        'ph_acc[idx] += t_ - ph_last; ph_last = t_; } } while (0)\n'
        "template <int DOMAIN, int NM, int PREC, bool JAC, int NH, bool FUSED>\n__global__")
    rep("    if (p.clk) { clk_c0 = __builtin_readcyclecounter(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }",
        "    if (p.clk) { clk_c0 = __builtin_readcyclecounter(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }\n"
        "    unsigned long long ph_acc[6] = {0, 0, 0, 0, 0, 0}, ph_last = clk_c0;")
    rep("        const long long qi = valid ? qi_raw : q_end - 1;\n",
        "        const long long qi = valid ? qi_raw : q_end - 1;\n"
        "        { float ph_d = (float)(int)qi; PH_STAMP(0, ph_d); }\n")
    rep("        f32x4 cacc[NM];\n        f32x4 bo;\n",
        "        PH_STAMP(1, y0); PH_STAMP(1, xs0); PH_STAMP(1, wi_z);\n        f32x4 cacc[NM];\n        f32x4 bo;\n")
    rep("        }  // !have_ctx\n", "        }  // !have_ctx\n        PH_STAMP(2, bo); PH_STAMP(2, cacc[0]);\n")
    rep("        if (op == OP_SAMPLE) p0 = base_pdf(x0, x1);\n", "        if (op == OP_SAMPLE) p0 = base_pdf(x0, x1);\n        PH_STAMP(3, x0); PH_STAMP(3, p0);\n")
    rep("            x0 += cstep * v[0];\n            x1 += cstep * v[1];\n        }\n",
        "            x0 += cstep * v[0];\n            x1 += cstep * v[1];\n        }\n        PH_STAMP(4, x0); PH_STAMP(4, acc);\n")
    rep("            if (writer) out_pdf[qi] = pdf_sa;\n        }\n        }\n",
        "            if (writer) out_pdf[qi] = pdf_sa;\n        }\n        }\n        { float ph_d = 0.f; PH_STAMP(5, ph_d); }\n")
    rep("            atomicAdd(slot + 1, dr);\n", "            atomicAdd(slot + 1, dr);\n            for (int i = 0; i < 6; ++i) atomicAdd(slot + 2 + i, ph_acc[i]);\n")
    rep("int bsdfd_profile_clock_mhz(bsdfd_handle h, double* mhz) {",
        "int bsdfd_tools_phase_read(bsdfd_handle h, double* out8) {\n"
        "    if (bsdfd_profile_read(h, nullptr, nullptr) != BSDFD_OK) return BSDFD_EHIP;\n"
        "    std::lock_guard<std::mutex> lock(h->prof_mu);\n"
        "    std::vector<unsigned long long> st((size_t)CLK_SLOTS * 8);\n"
        "    HIP_TRY(hipMemcpy(st.data(), h->d_clk, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));\n"
        "    for (int k = 0; k < 8; ++k) { out8[k] = 0.0; for (int i = 0; i < CLK_SLOTS; ++i) out8[k] += (double)st[(size_t)i * 8 + k]; }\n"
        "    return BSDFD_OK;\n}\n\n"
        "int bsdfd_profile_clock_mhz(bsdfd_handle h, double* mhz) {")
    return s


def build():
    open(TMP, "w").write(patch(open(SRC).read()))
    out = os.path.join(ROOT, "build_ab")
    os.makedirs(out, exist_ok=True)
    try:
        subprocess.run(["bash", os.path.join(ROOT, "tools", "ab_build.sh")], check=True)   # the side translation units
        common = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed", "-I", os.path.join(ROOT, "include")]
        subprocess.run(["hipcc", *common, "-c", TMP, "-o", os.path.join(out, "bsdfd_phaseclk.o")], check=True)
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(out, "bsdfd_phaseclk.o")] +
                       [os.path.join(out, f"{t}.o") for t in ("wavefront", "encoding", "measured", "bucket", "clock")] + ["-o", LIB], check=True)
    finally:
        os.remove(TMP)
    print("built", LIB)


def run(n):
    os.environ["BSDFD_LIB_PATH"] = LIB
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from bsdf_diffusion_sampling_amd import _lib, weights as W
    from bsdf_diffusion_sampling_amd.sampler import FlowSampler
    L = _lib.lib()
    L.bsdfd_tools_phase_read.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    dev = torch.device("cuda")
    for dom in ("disk", "spherical"):
        s = FlowSampler(W.load(W.shipped_path("aniso_miro_7_rgb", dom)), binding="ctypes")
        wi = bench.make_wi(dom, n, 1234, dev)
        wo = torch.empty((n, 3), device=dev)
        p = torch.empty(n, device=dev)
        ctx = s.new_context(n)
        s.plugin_sample(wi, None, T=8, seed=3, out=(wo, p), ctx_out=ctx)
        for T in (4, 8):
            cases = {"sample": lambda: s.plugin_sample(wi, None, T=T, seed=3, out=(torch.empty_like(wo), p)),
                     "pdf": lambda: s.plugin_pdf(wi, wo, T=T, out=p),
                     "pdf, ctx read": lambda: s.plugin_pdf(wi, wo, T=T, out=p, ctx_in=ctx)}
            for name, fn in cases.items():
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                s.set_profiling(True)
                for _ in range(10):
                    fn()
                k, ms = s.profile_read()
                out = (C.c_double * 8)()
                L.bsdfd_tools_phase_read(s._h, out)
                s.set_profiling(False)
                tot = sum(out[2:8])
                tiles = n / 16 * 10
                print(f"{dom:9s} T={T} {name:14s} {ms / k * 1e3:7.1f} us/launch | wave cycles per tile: " +
                      "  ".join(f"{PHASES[i]} {out[2 + i] / tiles:7.0f} ({out[2 + i] / tot * 100:4.1f} %)" for i in range(6)) +
                      f" | stamped {tot / out[0] * 100:5.1f} % of the waves' lifetime", flush=True)
        s.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20)
