// wavefront.hip — the two streaming kernels of the wavefront harness.
//
// The reference renders through Mitsuba 3 (`mi.render(scene, spp=4, seed)` in a 128..256-pass loop,
// rendering/brdf_measured_disk.py:146-155), which calls the plugin's sample()/pdf() once per
// wavefront; its own helper restates the primary-ray generation (pixel index -> film position +
// jitter -> sensor ray, rendering/utils/mitsuba_helper.py:59-127) and the power-heuristic MIS
// weight (:130-137).  Mitsuba is not part of this build, so the harness drives the SAME plugin
// entry points from a minimal scene of its own: pinhole camera, one analytic sphere carrying the
// material (the "material ball" of matpreview/*.xml), a lat-long environment map, one bounce,
// BSDF sampling + cosine-hemisphere light sampling combined with the power heuristic — one
// sample() and one pdf() call per path per pass, as config 5 describes.
//
//   primary : pixel (row, col), sample s  ->  wi (local), wl (local light-sample direction),
//             world normal, world ray direction            [HBM-bound: 48 B written per path]
//   shade   : + wo, pdf(wo), pdf(wl) from the sampler      ->  film += mean over spp of the MIS estimate
//                                                          [HBM-bound: 80 B read per path]
// Both are a few dozen flops per path; they are written as plain coalesced streaming kernels.
// Path index inside a tile of rows [row_begin, row_end): ((row - row_begin) * width + col) * spp + s;
// the Philox counter is the GLOBAL path index (row * width + col) * spp + s, so an image does not
// depend on how its rows are split over GPUs.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <string>

#include "bsdfd.h"
#include "common.h"

namespace {

struct V3 {
    float x, y, z;
};
__host__ __device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }
__device__ __forceinline__ void st3(float* p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

// Orthonormal basis from a unit normal (Duff et al. 2017, the construction behind Mitsuba's
// coordinate_system()): s, t with (s, t, n) right-handed.
__device__ __forceinline__ void onb(V3 n, V3& s, V3& t) {
    const float sign = copysignf(1.0f, n.z);
    const float a = -1.0f / (sign + n.z);
    const float b = n.x * n.y * a;
    s = v3(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    t = v3(b, sign + n.y * n.y * a, -n.y);
}

constexpr int WF_MAX_SPHERES = 32;

struct Scene {
    V3 o, right, up, fwd;
    float tan_half_fov;
    int width, height;
    float albedo[3];
    int env_w, env_h;
    int n_sph;                       // material balls; ball k carries material k
    float sph[WF_MAX_SPHERES][4];    // centre xyz, radius
    int has_plane;                   // diffuse checkerboard ground plane y = plane_y (the matpreview scenes' floor)
    float plane_y, checker_scale, checker_c0, checker_c1;
};

__global__ __launch_bounds__(256) void primary_kernel(Scene sc, int row_begin, int row_end, int spp,
                                                      unsigned long long seed, unsigned long long pass,
                                                      float* __restrict__ wi, float* __restrict__ wl,
                                                      float* __restrict__ nrm, float* __restrict__ dir,
                                                      long long* __restrict__ mat) {
    const long long n = (long long)(row_end - row_begin) * sc.width * spp;
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const long long pix_local = p / spp;
    const int s = (int)(p - pix_local * spp);
    const int row = row_begin + (int)(pix_local / sc.width);
    const int col = (int)(pix_local % sc.width);
    const unsigned long long gp = ((unsigned long long)row * sc.width + col) * (unsigned long long)spp + s;
    unsigned u[4];
    philox4x32((unsigned)seed, (unsigned)(seed >> 32), (unsigned)gp, (unsigned)(gp >> 32), (unsigned)pass,
               0x57617665u /* "Wave" */, u);
    // film position in [0,1)^2 (pixel + jitter), mitsuba_helper.py:104-117
    const float jx = (float)(u[0] >> 8) * (1.0f / 16777216.0f), jy = (float)(u[1] >> 8) * (1.0f / 16777216.0f);
    const float fx = ((float)col + jx) / (float)sc.width, fy = ((float)row + jy) / (float)sc.height;
    const float sx = (2.0f * fx - 1.0f) * sc.tan_half_fov;
    const float sy = (1.0f - 2.0f * fy) * sc.tan_half_fov * ((float)sc.height / (float)sc.width);
    V3 d = sc.fwd + sx * sc.right + sy * sc.up;
    d = (1.0f / sqrtf(dot(d, d))) * d;
    // closest hit among the analytic balls and the ground plane
    float t_best = 3.0e38f;
    int hit_k = -1;
    for (int k = 0; k < sc.n_sph; ++k) {
        const V3 oc = sc.o - v3(sc.sph[k][0], sc.sph[k][1], sc.sph[k][2]);
        const float b = dot(oc, d);
        const V3 perp = oc - b * d;  // discriminant as R^2 - (distance of the centre from the ray)^2: no b^2 - c cancellation
        const float disc = sc.sph[k][3] * sc.sph[k][3] - dot(perp, perp);
        const float t = -b - sqrtf(fmaxf(disc, 0.0f));
        if (disc > 0.0f && t > 0.0f && t < t_best) { t_best = t; hit_k = k; }
    }
    bool plane = false;
    if (sc.has_plane && d.y < 0.0f) {
        const float t = (sc.plane_y - sc.o.y) / d.y;
        if (t > 0.0f && t < t_best) { t_best = t; plane = true; hit_k = -1; }
    }
    V3 nn = v3(0.f, 0.f, 0.f), w_in = v3(0.f, 0.f, 1.f);
    long long material = sc.n_sph + 1;  // miss
    if (hit_k >= 0) {
        const V3 oc = sc.o - v3(sc.sph[hit_k][0], sc.sph[hit_k][1], sc.sph[hit_k][2]);
        nn = (1.0f / sc.sph[hit_k][3]) * (oc + t_best * d);
        nn = (1.0f / sqrtf(dot(nn, nn))) * nn;
        V3 fs, ft;
        onb(nn, fs, ft);
        w_in = v3(-dot(d, fs), -dot(d, ft), -dot(d, nn));
        material = hit_k;
    } else if (plane) {  // the floor is not neural: its reflectance travels in the wi slot, its id is n_sph
        const V3 h = sc.o + t_best * d;
        const int cx = (int)floorf(h.x * sc.checker_scale), cz = (int)floorf(h.z * sc.checker_scale);
        const float refl = ((cx + cz) & 1) ? sc.checker_c1 : sc.checker_c0;
        nn = v3(0.f, 1.f, 0.f);
        w_in = v3(refl, refl, refl);
        material = sc.n_sph;
    }
    // cosine-weighted light-sample direction in the local frame
    const float u2 = u01_open(u[2]), u3 = (float)(u[3] >> 8) * (1.0f / 16777216.0f);
    const float r = sqrtf(u2);
    float sp, cp;
    sincosf(6.28318530717958647692f * u3, &sp, &cp);
    const V3 w_l = v3(r * cp, r * sp, sqrtf(fmaxf(1.0f - u2, 0.0f)));
    st3(wi + 3 * p, w_in);
    st3(wl + 3 * p, w_l);
    st3(nrm + 3 * p, nn);
    st3(dir + 3 * p, d);
    if (mat) mat[p] = material;
}

// lat-long radiance map, y up: u = atan2(x, -z) / 2pi (wrapped), v = acos(y) / pi; bilinear
__device__ __forceinline__ void env_lookup(const float* __restrict__ env, int w, int h, V3 d, float out[3]) {
    float uu = atan2f(d.x, -d.z) * 0.15915494309189533577f;
    uu -= floorf(uu);
    const float vv = acosf(fminf(fmaxf(d.y, -1.0f), 1.0f)) * 0.31830988618379067154f;
    const float x = uu * (float)w - 0.5f, y = vv * (float)h - 0.5f;
    const float xf = floorf(x), yf = floorf(y);
    const float ax = x - xf, ay = y - yf;
    int x0 = (int)xf, y0 = (int)yf;
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = ((x0 % w) + w) % w; x1 = ((x1 % w) + w) % w;
    y0 = min(max(y0, 0), h - 1); y1 = min(max(y1, 0), h - 1);
    const float w00 = (1.f - ax) * (1.f - ay), w10 = ax * (1.f - ay), w01 = (1.f - ax) * ay, w11 = ax * ay;
    const float* p00 = env + ((long long)y0 * w + x0) * 3;
    const float* p10 = env + ((long long)y0 * w + x1) * 3;
    const float* p01 = env + ((long long)y1 * w + x0) * 3;
    const float* p11 = env + ((long long)y1 * w + x1) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = w00 * p00[c] + w10 * p10[c] + w01 * p01[c] + w11 * p11[c];
}

__device__ __forceinline__ float mis_power(float pa, float pb) {  // mitsuba_helper.py:130-137
    // pa^2 / (pa^2 + pb^2), formed as 1 / (1 + (pb/pa)^2): no overflow for the 1e9+ densities of
    // near-specular lobes
    if (!(pa > 0.0f)) return 0.0f;
    const float q = pb / pa;
    return 1.0f / fmaf(q, q, 1.0f);
}

__global__ __launch_bounds__(256) void shade_kernel(Scene sc, const float* __restrict__ env, int row_begin,
                                                    int row_end, int spp, const float* __restrict__ wo,
                                                    const float* __restrict__ pdf_o, const float* __restrict__ wl,
                                                    const float* __restrict__ pdf_l, const float* __restrict__ nrm,
                                                    const float* __restrict__ dir, const float* __restrict__ f_o,
                                                    const float* __restrict__ f_l, const float* __restrict__ wi,
                                                    const long long* __restrict__ mat, float* __restrict__ film) {
    const long long npix = (long long)(row_end - row_begin) * sc.width;
    const long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= npix) return;
    float acc[3] = {0.f, 0.f, 0.f};
    const float inv_pi = 0.31830988618379067154f;
    for (int s = 0; s < spp; ++s) {
        const long long p = pix * spp + s;
        const V3 n = ld3(nrm + 3 * p);
        float L[3];
        if (n.x == 0.0f && n.y == 0.0f && n.z == 0.0f) {  // miss: the camera sees the environment
            env_lookup(env, sc.env_w, sc.env_h, ld3(dir + 3 * p), L);
        } else if (mat && mat[p] == sc.n_sph) {  // diffuse floor, cosine-sampled: f cos / pdf = reflectance
            V3 fs, ft;
            onb(n, fs, ft);
            const V3 l = ld3(wl + 3 * p);
            float e[3];
            env_lookup(env, sc.env_w, sc.env_h, l.x * fs + l.y * ft + l.z * n, e);
            const float refl = wi[3 * p];
#pragma unroll
            for (int c = 0; c < 3; ++c) L[c] = refl * e[c];
        } else {
            V3 fs, ft;
            onb(n, fs, ft);
            L[0] = L[1] = L[2] = 0.0f;
            // BSDF-sampled direction: weight f cos / pdf.  With the ground truth (f_o = plugin eval(), which
            // already carries the albedo tint) that is f_o / pdf; without it the proxy f cos = albedo * pdf
            // (the nets model pdf ∝ lum(f cos)) gives weight = albedo.
            // per-path opt-out: a NaN in the f arrays selects the proxy for that path (array scenes mix materials
            // with and without a ground-truth file)
            const bool gt_o = f_o && f_o[3 * p] == f_o[3 * p];
            const bool gt_l = f_l && f_l[3 * p] == f_l[3 * p];
            const V3 o = ld3(wo + 3 * p);
            float pb = pdf_o[p];
            if (!(pb > 0.0f) || !isfinite(pb)) pb = 0.0f;
            if (pb > 0.0f) {
                const float w = mis_power(pb, fmaxf(o.z, 0.0f) * inv_pi);
                float e[3];
                env_lookup(env, sc.env_w, sc.env_h, o.x * fs + o.y * ft + o.z * n, e);
#pragma unroll
                for (int c = 0; c < 3; ++c) L[c] += w * e[c] * (gt_o ? f_o[3 * p + c] / pb : sc.albedo[c]);
            }
            // light-sampled direction (cosine hemisphere, pdf cos/pi): f cos / pdf_light
            const V3 l = ld3(wl + 3 * p);
            const float pl = l.z * inv_pi;
            float pbl = pdf_l[p];
            if (!(pbl > 0.0f) || !isfinite(pbl)) pbl = 0.0f;
            if (pl > 0.0f && (pbl > 0.0f || gt_l)) {
                const float w = mis_power(pl, pbl) / pl;
                float e[3];
                env_lookup(env, sc.env_w, sc.env_h, l.x * fs + l.y * ft + l.z * n, e);
#pragma unroll
                for (int c = 0; c < 3; ++c) L[c] += w * e[c] * (gt_l ? f_l[3 * p + c] : sc.albedo[c] * pbl);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += L[c];
    }
    const float inv = 1.0f / (float)spp;
#pragma unroll
    for (int c = 0; c < 3; ++c) film[3 * pix + c] += acc[c] * inv;
}

int to_scene(const bsdfd_wf_scene* s, int row_begin, int row_end, int spp, Scene& sc) {
    if (!s) return bsdfd_fail_(BSDFD_EINVAL, "null scene");
    if (s->width <= 0 || s->height <= 0) return bsdfd_fail_(BSDFD_EINVAL, "film size must be positive");
    if (row_begin < 0 || row_end > s->height || row_begin > row_end)
        return bsdfd_fail_(BSDFD_EINVAL, "row range outside the film");
    if (spp <= 0) return bsdfd_fail_(BSDFD_EINVAL, "spp must be positive");
    if (!(s->sphere_radius > 0.0f)) return bsdfd_fail_(BSDFD_EINVAL, "sphere radius must be positive");
    if (s->n_extra_spheres < 0 || s->n_extra_spheres > WF_MAX_SPHERES - 1)
        return bsdfd_fail_(BSDFD_EINVAL, "at most 31 extra spheres");
    sc.o = v3(s->cam_origin[0], s->cam_origin[1], s->cam_origin[2]);
    sc.right = v3(s->cam_right[0], s->cam_right[1], s->cam_right[2]);
    sc.up = v3(s->cam_up[0], s->cam_up[1], s->cam_up[2]);
    sc.fwd = v3(s->cam_forward[0], s->cam_forward[1], s->cam_forward[2]);
    sc.tan_half_fov = s->tan_half_fov;
    sc.width = s->width; sc.height = s->height;
    for (int c = 0; c < 3; ++c) sc.albedo[c] = s->albedo[c];
    sc.env_w = s->env_width; sc.env_h = s->env_height;
    sc.n_sph = 1 + s->n_extra_spheres;
    for (int c = 0; c < 3; ++c) sc.sph[0][c] = s->sphere_center[c];
    sc.sph[0][3] = s->sphere_radius;
    for (int k = 0; k < s->n_extra_spheres; ++k) {
        if (!(s->extra_spheres[k][3] > 0.0f)) return bsdfd_fail_(BSDFD_EINVAL, "sphere radius must be positive");
        for (int c = 0; c < 4; ++c) sc.sph[k + 1][c] = s->extra_spheres[k][c];
    }
    sc.has_plane = s->has_plane ? 1 : 0;
    sc.plane_y = s->plane_y; sc.checker_scale = s->checker_scale;
    sc.checker_c0 = s->checker_color0; sc.checker_c1 = s->checker_color1;
    return BSDFD_OK;
}

}  // namespace

extern "C" {

int bsdfd_wf_primary(const bsdfd_wf_scene* scene, int32_t row_begin, int32_t row_end, int32_t spp, uint64_t seed,
                     uint64_t pass, float* wi, float* wl, float* nrm, float* dir, int64_t* material, void* stream) {
    Scene sc;
    if (int rc = to_scene(scene, row_begin, row_end, spp, sc)) return rc;
    const long long n = (long long)(row_end - row_begin) * sc.width * spp;
    if (n == 0) return BSDFD_OK;
    if (!wi || !wl || !nrm || !dir) return bsdfd_fail_(BSDFD_EINVAL, "null output pointer");
    const long long blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffLL) return bsdfd_fail_(BSDFD_EINVAL, "tile too large for one launch");
    hipLaunchKernelGGL(primary_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), sc,
                       row_begin, row_end, spp, (unsigned long long)seed, (unsigned long long)pass, wi, wl, nrm, dir,
                       reinterpret_cast<long long*>(material));
    HIP_TRY(hipGetLastError());
    return BSDFD_OK;
}

int bsdfd_wf_shade(const bsdfd_wf_scene* scene, const float* env, int32_t row_begin, int32_t row_end, int32_t spp,
                   const float* wo, const float* pdf_o, const float* wl, const float* pdf_l, const float* nrm,
                   const float* dir, const float* f_o, const float* f_l, const float* wi, const int64_t* material,
                   float* film, void* stream) {
    Scene sc;
    if (int rc = to_scene(scene, row_begin, row_end, spp, sc)) return rc;
    if (sc.env_w <= 0 || sc.env_h <= 0) return bsdfd_fail_(BSDFD_EINVAL, "environment map size must be positive");
    const long long npix = (long long)(row_end - row_begin) * sc.width;
    if (npix == 0) return BSDFD_OK;
    if (!env || !wo || !pdf_o || !wl || !pdf_l || !nrm || !dir || !film)
        return bsdfd_fail_(BSDFD_EINVAL, "null pointer");
    if (material && !wi) return bsdfd_fail_(BSDFD_EINVAL, "the material ids need the wi array (floor reflectance)");
    if (sc.has_plane && !material) return bsdfd_fail_(BSDFD_EINVAL, "a scene with a floor needs the material ids");
    const long long blocks = (npix + 255) / 256;
    hipLaunchKernelGGL(shade_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), sc, env,
                       row_begin, row_end, spp, wo, pdf_o, wl, pdf_l, nrm, dir, f_o, f_l, wi,
                       reinterpret_cast<const long long*>(material), film);
    HIP_TRY(hipGetLastError());
    return BSDFD_OK;
}

}  // extern "C"
