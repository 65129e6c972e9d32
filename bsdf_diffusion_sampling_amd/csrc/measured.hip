// measured.hip — ground-truth evaluator behind the plugins' eval(): the RGL measured-BSDF model.
//
// The reference's plugins build Mitsuba 3's `measured` BSDF on `measuredbsdfs/<name>.bsdf`
// (rendering/brdf_measured_disk.py:36-42) and call its eval() for the sample weight f/pdf and the
// firefly rule (:96-100, :107-110).  Mitsuba has no AMD GPU variant, so the drop-in carries its own
// evaluator of the published model (Dupuy & Jakob, SIGGRAPH Asia 2018):
//
//     f(wi, wo) cos(theta_o) = spec(s; phi_i, theta_i) * D(u_m) / (4 sigma(u_i)),
//     u = (sqrt(2 theta / pi), (phi + pi) / 2 pi),   s = VNDF^-1(u_m | phi_i, theta_i),
//
// every table a bilinear grid over [0,1]^2, linearly interpolated over the incident-direction
// parameters; VNDF^-1 is the inverse of the inverse-CDF warp of that interpolated density
// (marginal over azimuth rows, conditional over elevation columns).  The host part parses the
// tensor file and builds the normalised VNDF with its conditional / marginal CDFs in double;
// the kernel is one thread per (wi, wo) pair: ~60 gathers from tables that total < 1 MB (L2-resident),
// i.e. latency- not bandwidth-bound; it is not part of the neural hot path.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "bsdfd.h"
#include "common.h"

namespace {

struct Table {        // [slices][h][w] fp32 on the device
    const float* data;
    int w, h;
};

struct MeasuredDev {
    const float* phi_i;
    const float* theta_i;
    int n_phi, n_theta;
    int isotropic, jacobian, reduction;
    float fold_x, fold_y;  // signs of (cos, sin) at the middle of the file's phi_i range: the quadrant / half-plane stored
    Table ndf, sigma, vndf, rgb;
    const float* vndf_cond;  // [n_phi][n_theta][h][w]  cumulative row integrals (patch units, normalised)
    const float* vndf_marg;  // [n_phi][n_theta][h]
};

struct Field {
    int dtype;
    std::vector<uint64_t> shape;
    const unsigned char* ptr;
    size_t count() const {
        size_t n = 1;
        for (auto s : shape) n *= (size_t)s;
        return n;
    }
    // element count with the shape fields validated: every dimension <= 2^26 and no wrap-around of the product (a
    // crafted shape that wraps to a small product would pass the "fits in the file" test and then index out of bounds)
    bool count_checked(size_t* out) const {
        size_t n = 1;
        for (auto s : shape) {
            if (s > (1ull << 26)) return false;
            if (__builtin_mul_overflow(n, (size_t)s, &n)) return false;
        }
        *out = n;
        return true;
    }
};

__device__ __forceinline__ float elevation(float x, float y, float z) {  // 2 asin(|d - z| / 2)
    const float dist = sqrtf(x * x + y * y + (z - 1.0f) * (z - 1.0f));
    return 2.0f * asinf(fminf(0.5f * dist, 1.0f));
}

// interval i with vals[i] <= p < vals[i+1] (clamped) and the weight of its upper end
__device__ __forceinline__ void interval(const float* __restrict__ vals, int n, float p, int& i, float& t) {
    if (n == 1) { i = 0; t = 0.0f; return; }
    int lo = 0;
    for (int k = 1; k < n - 1; ++k) lo = vals[k] <= p ? k : lo;  // n <= a few dozen
    i = lo;
    const float a = vals[lo], b = vals[lo + 1];
    t = fminf(fmaxf((p - a) / (b - a), 0.0f), 1.0f);
}

struct Patch {
    int ix, iy;
    float fx, fy;
};
__device__ __forceinline__ Patch patch_of(float x, float y, int w, int h) {
    x *= (float)(w - 1); y *= (float)(h - 1);
    Patch p;
    p.ix = min(max((int)floorf(x), 0), w - 2);
    p.iy = min(max((int)floorf(y), 0), h - 2);
    p.fx = x - (float)p.ix; p.fy = y - (float)p.iy;
    return p;
}
__device__ __forceinline__ float bilerp(float v00, float v10, float v01, float v11, float fx, float fy) {
    return (1.0f - fy) * ((1.0f - fx) * v00 + fx * v10) + fy * ((1.0f - fx) * v01 + fx * v11);
}
__device__ __forceinline__ float eval_plain(const Table& t, float x, float y) {
    const Patch p = patch_of(x, y, t.w, t.h);
    const float* d = t.data + (size_t)p.iy * t.w + p.ix;
    return bilerp(d[0], d[1], d[t.w], d[t.w + 1], p.fx, p.fy);
}

// f(wi, wo) cos(theta_o) for one pair; false (and rgb = 0) on the lower hemispheres
__device__ __forceinline__ bool measured_f(const MeasuredDev& m, float wix, float wiy, float wiz, float wox, float woy,
                                           float woz, float rgb[3]) {
    rgb[0] = rgb[1] = rgb[2] = 0.0f;
    if (!(wiz > 0.0f && woz > 0.0f)) return false;
    if (m.reduction >= 2) {
        // Symmetries of an anisotropic acquisition: only phi_i in a half-plane (reduction 2: point symmetry)
        // or a quadrant (reduction 4: two mirror planes) is stored.  Mitsuba folds with mulsign_neg(v, s) =
        // -|..|, i.e. into y <= 0 (and x <= 0), which is where its files keep phi_i; here the target is read
        // off the file's own phi_i range, which is the same thing for such files and right for any other.
        const bool fy = wiy * m.fold_y < 0.0f;
        const bool fx = m.reduction == 4 ? (wix * m.fold_x < 0.0f) : fy;
        if (fx) { wix = -wix; wox = -wox; }
        if (fy) { wiy = -wiy; woy = -woy; }
    }
    float mx = wix + wox, my = wiy + woy, mz = wiz + woz;
    const float inv = 1.0f / fmaxf(sqrtf(mx * mx + my * my + mz * mz), 1e-30f);
    mx *= inv; my *= inv; mz *= inv;
    const float theta_i = elevation(wix, wiy, wiz), phi_i = atan2f(wiy, wix);
    const float theta_m = elevation(mx, my, mz), phi_m = atan2f(my, mx);
    const float inv_2pi = 0.15915494309189533577f, pi = 3.14159265358979323846f;
    // unit-square coordinates: x = elevation, y = azimuth
    const float ui_x = sqrtf(theta_i * (2.0f / pi)), ui_y = (phi_i + pi) * inv_2pi;
    const float um_x = sqrtf(theta_m * (2.0f / pi));
    float um_y = ((m.isotropic ? phi_m - phi_i : phi_m) + pi) * inv_2pi;
    um_y -= floorf(um_y);

    // incident-direction parameter slices (<= 4) and their weights
    int ip, it;
    float tp, tt;
    interval(m.phi_i, m.n_phi, phi_i, ip, tp);
    interval(m.theta_i, m.n_theta, theta_i, it, tt);
    int slice[4];
    float wgt[4];
    int ns = 0;
    for (int a = 0; a < (m.n_phi > 1 ? 2 : 1); ++a)
        for (int b = 0; b < (m.n_theta > 1 ? 2 : 1); ++b) {
            slice[ns] = (ip + a) * m.n_theta + (it + b);
            wgt[ns] = (a ? tp : 1.0f - tp) * (b ? tt : 1.0f - tt);
            ++ns;
        }

    // ---- s = VNDF^-1(u_m): invert the marginal/conditional warp of the interpolated density ----
    const int vw = m.vndf.w, vh = m.vndf.h;
    const Patch pv = patch_of(um_x, um_y, vw, vh);
    float v00 = 0.f, v10 = 0.f, v01 = 0.f, v11 = 0.f, cdf0 = 0.f, cdf1 = 0.f, r0 = 0.f, r1 = 0.f, marg = 0.f;
    for (int k = 0; k < ns; ++k) {
        const size_t base = (size_t)slice[k] * vh * vw;
        const float* d = m.vndf.data + base + (size_t)pv.iy * vw + pv.ix;
        const float* c = m.vndf_cond + base + (size_t)pv.iy * vw;
        const float w = wgt[k];
        v00 += w * d[0]; v10 += w * d[1]; v01 += w * d[vw]; v11 += w * d[vw + 1];
        cdf0 += w * c[pv.ix]; cdf1 += w * c[vw + pv.ix];
        r0 += w * c[vw - 1]; r1 += w * c[2 * vw - 1];
        marg += w * m.vndf_marg[(size_t)slice[k] * vh + pv.iy];
    }
    const float c0 = (1.0f - pv.fy) * v00 + pv.fy * v01, c1 = (1.0f - pv.fy) * v10 + pv.fy * v11;
    const float part = pv.fx * (c0 + 0.5f * pv.fx * (c1 - c0));
    const float row = (1.0f - pv.fy) * r0 + pv.fy * r1;
    const float s0 = row > 0.0f ? (part + (1.0f - pv.fy) * cdf0 + pv.fy * cdf1) / row : 0.0f;
    const float s1 = pv.fy * (r0 + 0.5f * pv.fy * (r1 - r0)) + marg;

    // ---- spectral (rgb) lookup at s ----
    const int sw = m.rgb.w, sh = m.rgb.h;
    const Patch ps = patch_of(s0, s1, sw, sh);
    for (int k = 0; k < ns; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* d = m.rgb.data + (((size_t)slice[k] * 3 + c) * sh + ps.iy) * sw + ps.ix;
            rgb[c] += wgt[k] * bilerp(d[0], d[1], d[sw], d[sw + 1], ps.fx, ps.fy);
        }
    float scale = 1.0f;
    if (m.jacobian) scale = eval_plain(m.ndf, um_x, um_y) / (4.0f * eval_plain(m.sigma, ui_x, ui_y));
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb[c] *= scale;
    return true;
}

struct Tint {
    float r, g, b;
};

// eval(): rgb_out = f cos * tint
__global__ __launch_bounds__(256) void measured_eval_kernel(MeasuredDev m, const float* __restrict__ wi,
                                                            const float* __restrict__ wo, long long n, Tint tint,
                                                            float* __restrict__ out) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    float f[3];
    measured_f(m, wi[3 * q], wi[3 * q + 1], wi[3 * q + 2], wo[3 * q], wo[3 * q + 1], wo[3 * q + 2], f);
    out[3 * q] = f[0] * tint.r; out[3 * q + 1] = f[1] * tint.g; out[3 * q + 2] = f[2] * tint.b;
}

// The tail of the plugins' sample() in one pass (rendering/brdf_measured_disk.py:89-101,
// brdf_measured_spherical.py:97-109): value = f * albedo / pdf on active lanes with pdf > 0, firefly rule
// pdf := 0 where lum(value) >= thr, weight = value where active, pdf > 0 and cos(theta_o) > 0, else 0.
__global__ __launch_bounds__(256) void measured_weight_kernel(MeasuredDev m, const float* __restrict__ wi,
                                                              const float* __restrict__ wo,
                                                              const float* __restrict__ pdf_in,
                                                              const unsigned char* __restrict__ active, long long n,
                                                              Tint tint, float thr, float* __restrict__ weight,
                                                              float* __restrict__ pdf_out) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const float wiz = wi[3 * q + 2], woz = wo[3 * q + 2];
    float f[3];
    measured_f(m, wi[3 * q], wi[3 * q + 1], wiz, wo[3 * q], wo[3 * q + 1], woz, f);
    const float pdf = pdf_in[q];
    const bool act = wiz > 0.0f && (!active || active[q] != 0);
    float v[3] = {0.f, 0.f, 0.f};
    if (act && pdf > 0.0f) {
        const float inv = 1.0f / pdf;
        v[0] = f[0] * tint.r * inv; v[1] = f[1] * tint.g * inv; v[2] = f[2] * tint.b * inv;
    }
    const float lum = 0.2126f * v[0] + 0.7152f * v[1] + 0.0722f * v[2];  // rendering/utils/mitsuba_brdf_draw.py:36-38
    const float p = lum < thr ? pdf : 0.0f;
    const bool keep = act && p > 0.0f && woz > 0.0f;
    pdf_out[q] = p;
    weight[3 * q] = keep ? v[0] : 0.0f; weight[3 * q + 1] = keep ? v[1] : 0.0f; weight[3 * q + 2] = keep ? v[2] : 0.0f;
}

}  // namespace

struct bsdfd_measured_ctx {
    MeasuredDev dev;
    std::vector<void*> allocs;
    int device;
    std::string description;
};

namespace {

int upload(bsdfd_measured_ctx* h, const std::vector<float>& v, const float** out) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, v.size() * sizeof(float)));
    h->allocs.push_back(p);
    HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    *out = static_cast<const float*>(p);
    return BSDFD_OK;
}

std::vector<float> as_f32(const Field& f) {
    std::vector<float> v(f.count());
    std::memcpy(v.data(), f.ptr, v.size() * sizeof(float));
    return v;
}

}  // namespace

extern "C" {

int bsdfd_measured_create_from_file(const char* path, bsdfd_measured_handle* out) {
    if (!path || !out) return bsdfd_fail_(BSDFD_EINVAL, "null argument");
    *out = nullptr;
    FILE* fp = std::fopen(path, "rb");
    if (!fp) return bsdfd_fail_(BSDFD_EIO, std::string("cannot open ") + path);
    std::vector<unsigned char> raw;
    {
        std::fseek(fp, 0, SEEK_END);
        const long sz = std::ftell(fp);
        std::fseek(fp, 0, SEEK_SET);
        raw.resize(sz > 0 ? (size_t)sz : 0);
        const size_t got = raw.empty() ? 0 : std::fread(raw.data(), 1, raw.size(), fp);
        std::fclose(fp);
        if (got != raw.size()) return bsdfd_fail_(BSDFD_EIO, std::string("short read: ") + path);
    }
    // Mitsuba TensorFile: "tensor_file\0", u8 major, u8 minor, u32 n_fields, then per field:
    // u16 name_len, name, u16 ndim, u8 dtype, u64 offset, u64 shape[ndim]
    if (raw.size() < 18 || std::memcmp(raw.data(), "tensor_file", 12) != 0)
        return bsdfd_fail_(BSDFD_EIO, std::string(path) + ": not a tensor file");
    if (raw[12] != 1 || raw[13] != 0) return bsdfd_fail_(BSDFD_EIO, std::string(path) + ": unsupported tensor file version");
    uint32_t nf;
    std::memcpy(&nf, raw.data() + 14, 4);
    size_t pos = 18;
    std::map<std::string, Field> fields;
    static const int dtype_size[12] = {0, 1, 1, 2, 2, 4, 4, 8, 8, 2, 4, 8};
    for (uint32_t i = 0; i < nf; ++i) {
        auto need = [&](size_t k) { return pos + k <= raw.size(); };
        uint16_t nl, nd;
        if (!need(2)) return bsdfd_fail_(BSDFD_EIO, std::string(path) + ": truncated header");
        std::memcpy(&nl, raw.data() + pos, 2); pos += 2;
        if (!need((size_t)nl + 11)) return bsdfd_fail_(BSDFD_EIO, std::string(path) + ": truncated header");
        std::string name(reinterpret_cast<const char*>(raw.data() + pos), nl); pos += nl;
        std::memcpy(&nd, raw.data() + pos, 2); pos += 2;
        Field f;
        f.dtype = raw[pos]; pos += 1;
        uint64_t off;
        std::memcpy(&off, raw.data() + pos, 8); pos += 8;
        if (!need(8 * (size_t)nd)) return bsdfd_fail_(BSDFD_EIO, std::string(path) + ": truncated header");
        f.shape.resize(nd);
        std::memcpy(f.shape.data(), raw.data() + pos, 8 * (size_t)nd); pos += 8 * (size_t)nd;
        if (f.dtype < 1 || f.dtype > 11) return bsdfd_fail_(BSDFD_EIO, std::string(path) + ": bad dtype in field " + name);
        size_t cnt = 0, bytes = 0;
        if (!f.count_checked(&cnt) || __builtin_mul_overflow(cnt, (size_t)dtype_size[f.dtype], &bytes))
            return bsdfd_fail_(BSDFD_EIO, std::string(path) + ": implausible shape in field " + name);
        if (off > raw.size() || bytes > raw.size() - off)
            return bsdfd_fail_(BSDFD_EIO, std::string(path) + ": field " + name + " exceeds the file");
        f.ptr = raw.data() + off;
        fields[name] = f;
    }
    auto want = [&](const char* name, int dtype, size_t ndim) -> const Field* {
        auto it = fields.find(name);
        if (it == fields.end() || it->second.dtype != dtype || it->second.shape.size() != ndim) return nullptr;
        return &it->second;
    };
    const Field *phi = want("phi_i", 10, 1), *theta = want("theta_i", 10, 1), *sigma = want("sigma", 10, 2),
                *ndf = want("ndf", 10, 2), *vndf = want("vndf", 10, 4), *rgb = want("rgb", 10, 5),
                *jac = want("jacobian", 1, 1);
    if (!phi || !theta || !sigma || !ndf || !vndf || !rgb || !jac)
        return bsdfd_fail_(BSDFD_EIO, std::string(path) + ": expected fp32 fields phi_i, theta_i, sigma, ndf, vndf, rgb and u8 "
                                                           "jacobian (spectral files are not supported; use the *_rgb.bsdf flavour)");
    for (const Field* f : {phi, theta, sigma, ndf, vndf, rgb})  // table extents are used as int indices below
        for (auto dim : f->shape)
            if (dim < 1 || dim > (1u << 20))
                return bsdfd_fail_(BSDFD_EIO, std::string(path) + ": table dimension out of range [1, 2^20]");
    if (jac->shape[0] < 1) return bsdfd_fail_(BSDFD_EIO, std::string(path) + ": empty jacobian field");
    const int n_phi = (int)phi->shape[0], n_theta = (int)theta->shape[0];
    if ((int)vndf->shape[0] != n_phi || (int)vndf->shape[1] != n_theta || (int)rgb->shape[0] != n_phi ||
        (int)rgb->shape[1] != n_theta || rgb->shape[2] != 3 || vndf->shape[2] < 2 || vndf->shape[3] < 2 ||
        rgb->shape[3] < 2 || rgb->shape[4] < 2 || ndf->shape[0] < 2 || ndf->shape[1] < 2 || sigma->shape[0] < 2 ||
        sigma->shape[1] < 2)
        return bsdfd_fail_(BSDFD_EIO, std::string(path) + ": inconsistent table shapes");

    int devid = -1;
    HIP_TRY(hipGetDevice(&devid));
    bsdfd_measured_ctx* h = new bsdfd_measured_ctx();
    h->device = devid;
    if (auto it = fields.find("description"); it != fields.end())
        h->description.assign(reinterpret_cast<const char*>(it->second.ptr), it->second.count());
    MeasuredDev& d = h->dev;
    d.n_phi = n_phi; d.n_theta = n_theta;
    d.isotropic = n_phi <= 2;
    d.jacobian = jac->ptr[0] ? 1 : 0;
    d.reduction = 0;
    const std::vector<float> phi_v = as_f32(*phi), theta_v = as_f32(*theta);
    d.fold_x = d.fold_y = 1.0f;
    if (!d.isotropic) {
        d.reduction = (int)std::lrint(2.0 * M_PI / ((double)phi_v[n_phi - 1] - (double)phi_v[0]));
        const double mid = 0.5 * ((double)phi_v[0] + (double)phi_v[n_phi - 1]);
        d.fold_x = std::cos(mid) < 0.0 ? -1.0f : 1.0f;
        d.fold_y = std::sin(mid) < 0.0 ? -1.0f : 1.0f;
    }

    // VNDF: per-slice normalisation and CDFs in double.  Integrals in patch units: a linear segment
    // integrates to the mean of its end points, a row of patches to the mean of its two vertex rows.
    const int vh = (int)vndf->shape[2], vw = (int)vndf->shape[3];
    const std::vector<float> vraw = as_f32(*vndf);
    std::vector<float> vdata(vraw.size()), vcond(vraw.size()), vmarg((size_t)n_phi * n_theta * vh);
    std::vector<double> cond((size_t)vh * vw), marg(vh);
    for (int s = 0; s < n_phi * n_theta; ++s) {
        const float* src = vraw.data() + (size_t)s * vh * vw;
        for (int y = 0; y < vh; ++y) {
            double acc = 0.0;
            cond[(size_t)y * vw] = 0.0;
            for (int x = 1; x < vw; ++x) {
                acc += 0.5 * ((double)src[(size_t)y * vw + x - 1] + (double)src[(size_t)y * vw + x]);
                cond[(size_t)y * vw + x] = acc;
            }
        }
        double acc = 0.0;
        marg[0] = 0.0;
        for (int y = 1; y < vh; ++y) {
            acc += 0.5 * (cond[(size_t)(y - 1) * vw + vw - 1] + cond[(size_t)y * vw + vw - 1]);
            marg[y] = acc;
        }
        const double scale = acc > 0.0 ? 1.0 / acc : 0.0;
        for (size_t i = 0; i < (size_t)vh * vw; ++i) {
            vdata[(size_t)s * vh * vw + i] = (float)((double)src[i] * scale);
            vcond[(size_t)s * vh * vw + i] = (float)(cond[i] * scale);
        }
        for (int y = 0; y < vh; ++y) vmarg[(size_t)s * vh + y] = (float)(marg[y] * scale);
    }
    int rc = BSDFD_OK;
    auto up = [&](const std::vector<float>& v, const float** dst) {
        if (rc == BSDFD_OK) rc = upload(h, v, dst);
    };
    up(phi_v, &d.phi_i);
    up(theta_v, &d.theta_i);
    up(as_f32(*ndf), &d.ndf.data);
    up(as_f32(*sigma), &d.sigma.data);
    up(vdata, &d.vndf.data);
    up(vcond, &d.vndf_cond);
    up(vmarg, &d.vndf_marg);
    up(as_f32(*rgb), &d.rgb.data);
    d.ndf.h = (int)ndf->shape[0]; d.ndf.w = (int)ndf->shape[1];
    d.sigma.h = (int)sigma->shape[0]; d.sigma.w = (int)sigma->shape[1];
    d.vndf.h = vh; d.vndf.w = vw;
    d.rgb.h = (int)rgb->shape[3]; d.rgb.w = (int)rgb->shape[4];
    if (rc != BSDFD_OK) {
        for (void* p : h->allocs) (void)hipFree(p);
        delete h;
        return rc;
    }
    *out = h;
    return BSDFD_OK;
}

void bsdfd_measured_destroy(bsdfd_measured_handle h) {
    if (!h) return;
    for (void* p : h->allocs) (void)hipFree(p);
    delete h;
}

int bsdfd_measured_get_info(bsdfd_measured_handle h, int32_t* n_phi, int32_t* n_theta, int32_t* isotropic,
                            int32_t* jacobian, int32_t* reduction) {
    if (!h) return bsdfd_fail_(BSDFD_EINVAL, "null handle");
    if (n_phi) *n_phi = h->dev.n_phi;
    if (n_theta) *n_theta = h->dev.n_theta;
    if (isotropic) *isotropic = h->dev.isotropic;
    if (jacobian) *jacobian = h->dev.jacobian;
    if (reduction) *reduction = h->dev.reduction;
    return BSDFD_OK;
}

static int measured_launch_checks(bsdfd_measured_handle h, int64_t n) {
    if (!h) return bsdfd_fail_(BSDFD_EINVAL, "null handle");
    if (n < 0) return bsdfd_fail_(BSDFD_EINVAL, "N must be >= 0");
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != h->device) return bsdfd_fail_(BSDFD_EINVAL, "measured handle belongs to another device");
    if (((long long)n + 255) / 256 > 0x7fffffffLL) return bsdfd_fail_(BSDFD_EINVAL, "N too large for one launch");
    return BSDFD_OK;
}

int bsdfd_measured_eval(bsdfd_measured_handle h, const float* wi, const float* wo, int64_t n, const float* tint,
                        float* rgb_out, void* stream) {
    if (int rc = measured_launch_checks(h, n)) return rc;
    if (n == 0) return BSDFD_OK;
    if (!wi || !wo || !rgb_out) return bsdfd_fail_(BSDFD_EINVAL, "null pointer");
    const Tint t = tint ? Tint{tint[0], tint[1], tint[2]} : Tint{1.0f, 1.0f, 1.0f};
    hipLaunchKernelGGL(measured_eval_kernel, dim3((unsigned)(((long long)n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), h->dev, wi, wo, (long long)n, t, rgb_out);
    HIP_TRY(hipGetLastError());
    return BSDFD_OK;
}

int bsdfd_measured_sample_weight(bsdfd_measured_handle h, const float* wi, const float* wo, const float* pdf_sa,
                                 const unsigned char* active, int64_t n, const float* tint, float firefly_threshold,
                                 float* weight_out, float* pdf_out, void* stream) {
    if (int rc = measured_launch_checks(h, n)) return rc;
    if (n == 0) return BSDFD_OK;
    if (!wi || !wo || !pdf_sa || !weight_out || !pdf_out) return bsdfd_fail_(BSDFD_EINVAL, "null pointer");
    const Tint t = tint ? Tint{tint[0], tint[1], tint[2]} : Tint{1.0f, 1.0f, 1.0f};
    hipLaunchKernelGGL(measured_weight_kernel, dim3((unsigned)(((long long)n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), h->dev, wi, wo, pdf_sa, active, (long long)n, t,
                       firefly_threshold, weight_out, pdf_out);
    HIP_TRY(hipGetLastError());
    return BSDFD_OK;
}

}  // extern "C"
