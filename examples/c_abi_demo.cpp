// Pure C/C++ use of the drop-in boundary (include/bsdfd.h): no Python, no torch.
// Loads a neutral .bsdfw weight file, draws directions for a batch of shading queries with
// bsdfd_plugin_sample, evaluates bsdfd_plugin_pdf on them, captures both launches in a hipGraph and
// replays it, and prints a checksum.  Build (see tests/test_gpu_c_abi.py):
//   hipcc --offload-arch=gfx950 -O2 -I include examples/c_abi_demo.cpp -L bsdf_diffusion_sampling_amd -lbsdfd \
//         -Wl,-rpath,$PWD/bsdf_diffusion_sampling_amd -o c_abi_demo
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bsdfd.h"

#define CHECK(x)                                                                      \
    do {                                                                              \
        int rc__ = (x);                                                               \
        if (rc__ != 0) { std::fprintf(stderr, "%s failed (%d): %s\n", #x, rc__, bsdfd_last_error()); return 1; } \
    } while (0)
#define HIPCHECK(x)                                                                   \
    do {                                                                              \
        hipError_t e__ = (x);                                                         \
        if (e__ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s weights.bsdfw [N]\n", argv[0]); return 2; }
    const long long n = argc > 2 ? std::atoll(argv[2]) : 100000;
    bsdfd_handle h = nullptr;
    CHECK(bsdfd_create_from_file(argv[1], BSDFD_PREC_DEFAULT, &h));
    int32_t domain, width, n_hidden, prec;
    CHECK(bsdfd_get_info(h, &domain, &width, &n_hidden, &prec));
    const int T = domain == BSDFD_DOMAIN_DISK ? 4 : 8;
    std::printf("%s: domain %d, %d-wide x %d hidden, precision %d, %lld flop/query at T=%d\n", bsdfd_version(), domain,
                width, n_hidden, prec, (long long)bsdfd_flops_per_query(h, T), T);

    std::vector<float> wi(3 * n);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
    for (long long i = 0; i < n; ++i) {  // directions on the upper hemisphere
        const float r = 0.95f * std::sqrt(rnd()), a = 6.2831853f * rnd();
        wi[3 * i] = r * std::cos(a); wi[3 * i + 1] = r * std::sin(a); wi[3 * i + 2] = std::sqrt(1.0f - r * r);
    }
    float *d_wi, *d_wo, *d_ps, *d_pp;
    HIPCHECK(hipMalloc(&d_wi, 12 * n)); HIPCHECK(hipMalloc(&d_wo, 12 * n));
    HIPCHECK(hipMalloc(&d_ps, 4 * n)); HIPCHECK(hipMalloc(&d_pp, 4 * n));
    HIPCHECK(hipMemcpy(d_wi, wi.data(), 12 * n, hipMemcpyHostToDevice));
    hipStream_t st;
    HIPCHECK(hipStreamCreate(&st));

    // eager
    CHECK(bsdfd_plugin_sample(h, BSDFD_PLUGIN_MEASURED, d_wi, nullptr, /*seed*/ 7, /*offset*/ 0, n, T, d_wo, d_ps, st));
    CHECK(bsdfd_plugin_pdf(h, BSDFD_PLUGIN_MEASURED, d_wi, d_wo, n, T, d_pp, st));
    HIPCHECK(hipStreamSynchronize(st));
    std::vector<float> wo(3 * n), ps(n), pp(n);
    HIPCHECK(hipMemcpy(wo.data(), d_wo, 12 * n, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(ps.data(), d_ps, 4 * n, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(pp.data(), d_pp, 4 * n, hipMemcpyDeviceToHost));
    double sum_s = 0, sum_p = 0, norm_err = 0;
    long long pos = 0;
    for (long long i = 0; i < n; ++i) {
        sum_s += ps[i]; sum_p += pp[i]; pos += ps[i] > 0;
        norm_err = std::fmax(norm_err, std::fabs(wo[3 * i] * wo[3 * i] + wo[3 * i + 1] * wo[3 * i + 1] + wo[3 * i + 2] * wo[3 * i + 2] - 1.0));
    }
    std::printf("eager : sum pdf(sample) %.6e  sum pdf() %.6e  positive %lld/%lld  max | |wo|^2-1 | %.2e\n", sum_s, sum_p, pos, n, norm_err);

    // the same two launches captured in a hipGraph and replayed (no allocation / sync inside the calls)
    hipGraph_t graph; hipGraphExec_t exec;
    HIPCHECK(hipMemset(d_wo, 0, 12 * n)); HIPCHECK(hipMemset(d_pp, 0, 4 * n));
    HIPCHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    CHECK(bsdfd_plugin_sample(h, BSDFD_PLUGIN_MEASURED, d_wi, nullptr, 7, 0, n, T, d_wo, d_ps, st));
    CHECK(bsdfd_plugin_pdf(h, BSDFD_PLUGIN_MEASURED, d_wi, d_wo, n, T, d_pp, st));
    HIPCHECK(hipStreamEndCapture(st, &graph));
    HIPCHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    HIPCHECK(hipGraphLaunch(exec, st));
    HIPCHECK(hipStreamSynchronize(st));
    std::vector<float> pp2(n);
    HIPCHECK(hipMemcpy(pp2.data(), d_pp, 4 * n, hipMemcpyDeviceToHost));
    long long diff = 0;
    for (long long i = 0; i < n; ++i) diff += pp2[i] != pp[i];
    std::printf("graph : %lld of %lld pdf values differ from the eager run\n", diff, n);
    HIPCHECK(hipGraphExecDestroy(exec)); HIPCHECK(hipGraphDestroy(graph));

    // sample(wi) then pdf(wi, .) for the SAME intersections: hand the per-query context over (bsdfd_opts, *_ex calls)
    void* d_ctx = nullptr;
    const long long ctx_bytes = bsdfd_context_bytes(h, n, 1);
    HIPCHECK(hipMalloc(&d_ctx, (size_t)ctx_bytes));
    bsdfd_opts o = {};
    o.ctx_out = d_ctx;
    CHECK(bsdfd_plugin_sample_ex(h, BSDFD_PLUGIN_MEASURED, d_wi, nullptr, 7, 0, n, T, d_wo, d_ps, &o, st));
    o.ctx_out = nullptr; o.ctx_in = d_ctx;
    HIPCHECK(hipMemsetAsync(d_pp, 0, 4 * n, st));
    CHECK(bsdfd_plugin_pdf_ex(h, BSDFD_PLUGIN_MEASURED, d_wi, d_wo, n, T, d_pp, &o, st));
    HIPCHECK(hipStreamSynchronize(st));
    std::vector<float> pp3(n);
    HIPCHECK(hipMemcpy(pp3.data(), d_pp, 4 * n, hipMemcpyDeviceToHost));
    long long diff_ctx = 0;
    for (long long i = 0; i < n; ++i) diff_ctx += pp3[i] != pp[i];
    std::printf("ctx   : %lld B of per-query context; %lld of %lld pdf values differ from the eager run\n", ctx_bytes, diff_ctx, n);
    HIPCHECK(hipFree(d_ctx));

    // ABI 6 — bsdfd_opts.row_index: a wavefront whose lanes carry a material tag is bucketed on the device
    // (bsdfd_bucket_by_material) and the flow kernel reads wi / writes (wo, pdf) in LANE order through the bucket permutation:
    // no gathered copy, no scatter.  Here: two "materials" served by the same handle, lanes tagged 0 / 1 / 2 (2 = no material);
    // the lanes of material 0 must come out exactly as in the plain run above (Philox counter = offset + lane), the lanes of
    // tag 2 untouched.
    long long diff_row = 0, untouched_bad = 0;
    if (bsdfd_abi_version() != BSDFD_ABI_VERSION) { std::fprintf(stderr, "ABI mismatch\n"); return 1; }
    {
        std::vector<long long> ids(n);
        for (long long i = 0; i < n; ++i) ids[i] = (i * 2654435761ull >> 7) % 3;
        long long *d_ids = nullptr, *d_perm = nullptr, *d_counts = nullptr;
        void* d_ws = nullptr;
        const long long ws_bytes = bsdfd_bucket_workspace_bytes(n, 3);
        HIPCHECK(hipMalloc(&d_ids, 8 * n)); HIPCHECK(hipMalloc(&d_perm, 8 * n)); HIPCHECK(hipMalloc(&d_counts, 8 * 3));
        HIPCHECK(hipMalloc(&d_ws, (size_t)ws_bytes));
        HIPCHECK(hipMemcpy(d_ids, ids.data(), 8 * n, hipMemcpyHostToDevice));
        CHECK(bsdfd_bucket_by_material(reinterpret_cast<const int64_t*>(d_ids), n, 3, reinterpret_cast<int64_t*>(d_perm),
                                       reinterpret_cast<int64_t*>(d_counts), d_ws, ws_bytes, st));
        long long counts[3];
        HIPCHECK(hipMemcpyAsync(counts, d_counts, sizeof counts, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        const int64_t seg_end[2] = {counts[0], counts[0] + counts[1]};   // buckets 0 and 1 carry a material, bucket 2 does not
        const bsdfd_handle hs[2] = {h, h};
        HIPCHECK(hipMemsetAsync(d_wo, 0xff, 12 * n, st)); HIPCHECK(hipMemsetAsync(d_ps, 0xff, 4 * n, st));
        bsdfd_opts orow = {};
        orow.row_index = reinterpret_cast<const int64_t*>(d_perm);
        CHECK(bsdfd_plugin_sample_multi_ex(hs, 2, seg_end, BSDFD_PLUGIN_MEASURED, d_wi, nullptr, 7, 0, T, d_wo, d_ps, &orow, st));
        HIPCHECK(hipStreamSynchronize(st));
        std::vector<float> ps4(n);
        HIPCHECK(hipMemcpy(ps4.data(), d_ps, 4 * n, hipMemcpyDeviceToHost));
        for (long long i = 0; i < n; ++i) {
            unsigned bits; std::memcpy(&bits, &ps4[i], 4);
            if (ids[i] == 2) untouched_bad += bits != 0xffffffffu;
            else diff_row += ps4[i] != ps[i];
        }
        std::printf("rows  : %lld + %lld lanes through bsdfd_opts.row_index; %lld pdf values differ from the plain run, %lld untagged lanes touched\n",
                    (long long)counts[0], (long long)counts[1], diff_row, untouched_bad);
        HIPCHECK(hipFree(d_ids)); HIPCHECK(hipFree(d_perm)); HIPCHECK(hipFree(d_counts)); HIPCHECK(hipFree(d_ws));
    }
    bsdfd_destroy(h);
    return diff == 0 && diff_ctx == 0 && diff_row == 0 && untouched_bad == 0 && pos > n / 2 && norm_err < 1e-4 ? 0 : 1;
}
