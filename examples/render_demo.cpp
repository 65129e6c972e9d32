// A renderer's inner loop written against the C ABI only (include/bsdfd.h) — no Python, no torch:
//   per pass:  bsdfd_wf_primary  ->  bsdfd_plugin_sample_pdf  ->  [bsdfd_measured_eval x2]  ->  bsdfd_wf_shade
// i.e. the loop of the reference's `mi.render(scene, spp=4, seed)` passes (rendering/brdf_measured_disk.py:
// 146-155) around its plugin's sample()/pdf()/eval(), for the harness' material-ball scene.  Writes a PPM.
// Build (see tests/test_gpu_c_abi.py):
//   hipcc --offload-arch=gfx950 -O2 -I include examples/render_demo.cpp -L bsdf_diffusion_sampling_amd -lbsdfd \
//         -Wl,-rpath,$PWD/bsdf_diffusion_sampling_amd -o render_demo
//   ./render_demo weights.bsdfw [measured.bsdf|-] [out.ppm] [size] [passes]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bsdfd.h"

#define CHECK(x)                                                                                                  \
    do {                                                                                                          \
        int rc__ = (x);                                                                                           \
        if (rc__ != 0) { std::fprintf(stderr, "%s failed (%d): %s\n", #x, rc__, bsdfd_last_error()); return 1; }  \
    } while (0)
#define HIPCHECK(x)                                                                                      \
    do {                                                                                                 \
        hipError_t e__ = (x);                                                                            \
        if (e__ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__)); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s weights.bsdfw [measured.bsdf|-] [out.ppm] [size] [passes]\n", argv[0]); return 2; }
    const char* gt_path = argc > 2 && std::strcmp(argv[2], "-") != 0 ? argv[2] : nullptr;
    const char* out_path = argc > 3 ? argv[3] : "render_demo.ppm";
    const int size = argc > 4 ? std::atoi(argv[4]) : 256, passes = argc > 5 ? std::atoi(argv[5]) : 64, spp = 4;

    bsdfd_handle h = nullptr;
    CHECK(bsdfd_create_from_file(argv[1], BSDFD_PREC_DEFAULT, &h));
    int32_t domain, width, n_hidden, prec;
    CHECK(bsdfd_get_info(h, &domain, &width, &n_hidden, &prec));
    const int T = domain == BSDFD_DOMAIN_DISK ? 4 : 8;
    bsdfd_measured_handle gt = nullptr;
    if (gt_path) CHECK(bsdfd_measured_create_from_file(gt_path, &gt));

    // scene: camera at (0, 0.6, 3.2) looking at the unit ball at the origin, 40 degrees horizontal fov
    bsdfd_wf_scene sc;
    std::memset(&sc, 0, sizeof(sc));
    const float o[3] = {0.0f, 0.6f, 3.2f};
    float f[3] = {-o[0], -o[1], -o[2]};
    const float fl = std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    for (float& v : f) v /= fl;
    float r[3] = {f[1] * 0 - f[2] * 1, f[2] * 0 - f[0] * 0, f[0] * 1 - f[1] * 0};  // f x (0,1,0)
    const float rl = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    for (float& v : r) v /= rl;
    const float u[3] = {r[1] * f[2] - r[2] * f[1], r[2] * f[0] - r[0] * f[2], r[0] * f[1] - r[1] * f[0]};  // r x f
    for (int c = 0; c < 3; ++c) { sc.cam_origin[c] = o[c]; sc.cam_forward[c] = f[c]; sc.cam_right[c] = r[c]; sc.cam_up[c] = u[c]; sc.albedo[c] = 1.0f; }
    sc.tan_half_fov = std::tan(20.0f * 3.14159265f / 180.0f);
    sc.width = size; sc.height = size;
    sc.sphere_radius = 1.0f;
    // environment: horizon gradient + one warm light, lat-long, y up
    const int ew = 256, eh = 128;
    sc.env_width = ew; sc.env_height = eh;
    std::vector<float> env(3 * ew * eh);
    for (int y = 0; y < eh; ++y)
        for (int x = 0; x < ew; ++x) {
            const float th = 3.14159265f * (y + 0.5f) / eh, ph = 6.2831853f * (x + 0.5f) / ew;
            const float d[3] = {std::sin(th) * std::sin(ph), std::cos(th), -std::sin(th) * std::cos(ph)};
            const float up = std::max(d[1], 0.0f);
            const float sun = std::exp(60.0f * (d[0] * 0.5f + d[1] * 0.7f + d[2] * 0.5f - 1.0f)) * 25.0f;
            float* e = &env[3 * (y * ew + x)];
            e[0] = (d[1] < 0 ? 0.04f : 0.25f * (1 - up) + 0.27f * up) + sun;
            e[1] = (d[1] < 0 ? 0.036f : 0.24f * (1 - up) + 0.36f * up) + 0.85f * sun;
            e[2] = (d[1] < 0 ? 0.03f : 0.22f * (1 - up) + 0.6f * up) + 0.6f * sun;
        }

    const long long n = (long long)size * size * spp;
    float *d_env, *wi, *wl, *nrm, *dir, *wo, *po, *pl, *fo = nullptr, *fl_ = nullptr, *film;
    HIPCHECK(hipMalloc(&d_env, env.size() * 4));
    HIPCHECK(hipMemcpy(d_env, env.data(), env.size() * 4, hipMemcpyHostToDevice));
    for (float** p : {&wi, &wl, &nrm, &dir, &wo}) HIPCHECK(hipMalloc(p, 12 * n));
    for (float** p : {&po, &pl}) HIPCHECK(hipMalloc(p, 4 * n));
    if (gt) { HIPCHECK(hipMalloc(&fo, 12 * n)); HIPCHECK(hipMalloc(&fl_, 12 * n)); }
    HIPCHECK(hipMalloc(&film, 12ll * size * size));
    HIPCHECK(hipMemset(film, 0, 12ll * size * size));
    hipStream_t st;
    HIPCHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    HIPCHECK(hipEventRecord(e0, st));
    for (int pass = 0; pass < passes; ++pass) {
        CHECK(bsdfd_wf_primary(&sc, 0, size, spp, /*seed*/ 1, pass, wi, wl, nrm, dir, nullptr, st));
        CHECK(bsdfd_plugin_sample_pdf(h, BSDFD_PLUGIN_MEASURED, wi, nullptr, wl, /*seed*/ 1000 + pass, 0, n, T, wo, po, pl, st));
        if (gt) {
            CHECK(bsdfd_measured_eval(gt, wi, wo, n, nullptr, fo, st));
            CHECK(bsdfd_measured_eval(gt, wi, wl, n, nullptr, fl_, st));
        }
        CHECK(bsdfd_wf_shade(&sc, d_env, 0, size, spp, wo, po, wl, pl, nrm, dir, fo, fl_, wi, nullptr, film, st));
    }
    HIPCHECK(hipEventRecord(e1, st));
    HIPCHECK(hipStreamSynchronize(st));
    float ms = 0;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> img(3ull * size * size);
    HIPCHECK(hipMemcpy(img.data(), film, img.size() * 4, hipMemcpyDeviceToHost));
    double mean = 0;
    bool finite = true;
    for (float& v : img) { v /= passes; mean += v; finite = finite && std::isfinite(v); }
    mean /= img.size();
    FILE* fp = std::fopen(out_path, "wb");
    if (!fp) { std::fprintf(stderr, "cannot write %s\n", out_path); return 1; }
    std::fprintf(fp, "P6\n%d %d\n255\n", size, size);
    for (float v : img) {
        const float t = std::pow(std::max(v, 0.0f) / (1.0f + std::max(v, 0.0f)), 1.0f / 2.2f);
        std::fputc((int)(std::min(t, 1.0f) * 255.0f + 0.5f), fp);
    }
    std::fclose(fp);
    std::printf("%s: %dx%d, %d passes x %d spp, %s: %.2f ms (%.1f Mpaths/s), image mean %.4f, finite %d -> %s\n", bsdfd_version(),
                size, size, passes, spp, gt ? "ground-truth f" : "proxy f", ms, (double)n * passes / ms / 1e3, mean, (int)finite, out_path);
    if (gt) bsdfd_measured_destroy(gt);
    bsdfd_destroy(h);
    return finite ? 0 : 1;
}
