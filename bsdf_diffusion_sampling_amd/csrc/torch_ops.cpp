// torch_ops.cpp — PyTorch-ROCm operator library `bsdfd::*` over the C ABI of libbsdfd.so.
//
// The reference binds its only native library (tiny-cuda-nn) to PyTorch with a thin C++ layer
// (tiny-cuda-nn/bindings/torch/tinycudann/bindings.cpp:79-110): contiguity / dtype / device checks that throw,
// a device guard, the CURRENT stream of the tensors' device, outputs allocated by torch, raw pointers handed to
// the native call.  This file is that layer for bsdfd: every operator checks its tensors, takes the current HIP
// stream of their device and calls one entry point of include/bsdfd.h — no arithmetic here.  It exists next to
// the ctypes shim (sampler.py) because a ctypes call costs 9.8 us of host time — more than the 8.3 us kernel of a
// 4 Ki-query wavefront (hipGraph replay), so small wavefronts were host-bound; through the dispatcher a call issues in
// 7.9 us and the same loop runs at 8.5 us per call, i.e. kernel-bound (tools/host_overhead.py, round 2).  The operators
// are also visible to torch tooling (torch.ops.bsdfd.*, stream-capturable, no GIL-held pointer marshalling).
//
// Built in-tree (no hipify pass, no CUDA spellings): g++ against torch's headers and libbsdfd.so, loaded with
// torch.ops.load_library (bsdf_diffusion_sampling_amd/torch_ext.py).
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <optional>
#include <string>
#include <tuple>

#include "bsdfd.h"

namespace {

using at::Tensor;

bsdfd_handle as_handle(int64_t h) {
    TORCH_CHECK(h != 0, "bsdfd: null handle");
    return reinterpret_cast<bsdfd_handle>(static_cast<intptr_t>(h));
}

void ok(int rc) { TORCH_CHECK(rc == BSDFD_OK, "bsdfd error ", rc, ": ", bsdfd_last_error()); }

// the checks of bindings.cpp:54-55,73 (CHECK_INPUT + dtype + shape), with messages
const float* in2d(const Tensor& t, int64_t cols, const char* name, const at::Device& dev, int64_t n = -1) {
    TORCH_CHECK(t.device().is_cuda(), name, " must be a CUDA (HIP) tensor");
    TORCH_CHECK(t.device() == dev, name, " is on ", t.device(), ", expected ", dev);
    TORCH_CHECK(t.scalar_type() == at::kFloat, name, " must be float32, got ", t.scalar_type());
    TORCH_CHECK(t.dim() == 2 && t.size(1) == cols, name, " must have shape [N, ", cols, "], got ", t.sizes());
    TORCH_CHECK(n < 0 || t.size(0) == n, name, " has ", t.size(0), " rows, expected ", n);
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
    return t.data_ptr<float>();
}
const float* opt2d(const std::optional<Tensor>& t, int64_t cols, const char* name, const at::Device& dev, int64_t n) {
    return t.has_value() ? in2d(*t, cols, name, dev, n) : nullptr;
}

struct Launch {  // device guard + current stream of the inputs' device (bindings.cpp:95-96)
    c10::hip::HIPGuardMasqueradingAsCUDA guard;
    void* stream;
    explicit Launch(const at::Device& dev)
        : guard(dev), stream(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(dev.index()).stream()) {}
};

Tensor empty_f32(at::IntArrayRef shape, const at::Device& dev) {
    return at::empty(shape, at::TensorOptions().dtype(at::kFloat).device(dev));
}

// ---- handle management -------------------------------------------------------------------------------
int64_t create_from_file(const std::string& path, int64_t precision, int64_t device_index) {
    c10::hip::HIPGuardMasqueradingAsCUDA guard(at::Device(at::kCUDA, static_cast<c10::DeviceIndex>(device_index)));
    bsdfd_handle h = nullptr;
    ok(bsdfd_create_from_file(path.c_str(), static_cast<int32_t>(precision), &h));
    return static_cast<int64_t>(reinterpret_cast<intptr_t>(h));
}

// weights as CPU float32 tensors in nn.Linear layout ([out, in]); the library copies and packs them
int64_t create(int64_t domain, int64_t width, int64_t n_hidden, int64_t precision, const Tensor& w_in, const Tensor& w_hidden,
               const Tensor& w_out, const Tensor& base_w1, const Tensor& base_b1, const Tensor& base_w2, const Tensor& base_b2,
               int64_t device_index) {
    const Tensor* ws[] = {&w_in, &w_hidden, &w_out, &base_w1, &base_b1, &base_w2, &base_b2};
    for (const Tensor* w : ws)
        TORCH_CHECK(w->device().is_cpu() && w->scalar_type() == at::kFloat && w->is_contiguous(),
                    "bsdfd::create takes contiguous CPU float32 weight tensors");
    const int64_t in_dim = (domain == BSDFD_DOMAIN_DISK ? 2 : 3) + 1 + 2 + 4 * 5;
    TORCH_CHECK(w_in.numel() == width * in_dim && w_out.numel() == 2 * width && w_hidden.numel() == (n_hidden - 1) * width * width,
                "bsdfd::create: velocity-net weight shapes do not match (domain, width, n_hidden)");
    TORCH_CHECK(base_w1.numel() == 16 * 14 && base_b1.numel() == 16 && base_w2.numel() == 4 * 16 && base_b2.numel() == 4,
                "bsdfd::create: base-net weight shapes must be those of PE_3 -> 16 -> 4");
    bsdfd_desc d{};
    d.domain = static_cast<int32_t>(domain);
    d.width = static_cast<int32_t>(width);
    d.n_hidden = static_cast<int32_t>(n_hidden);
    d.pe_bands = 5;
    d.base_hidden = 16;
    d.base_pe_bands = 3;
    d.precision = static_cast<int32_t>(precision);
    d.w_in = w_in.data_ptr<float>();
    d.w_hidden = w_hidden.numel() ? w_hidden.data_ptr<float>() : nullptr;
    d.w_out = w_out.data_ptr<float>();
    d.base_w1 = base_w1.data_ptr<float>();
    d.base_b1 = base_b1.data_ptr<float>();
    d.base_w2 = base_w2.data_ptr<float>();
    d.base_b2 = base_b2.data_ptr<float>();
    c10::hip::HIPGuardMasqueradingAsCUDA guard(at::Device(at::kCUDA, static_cast<c10::DeviceIndex>(device_index)));
    bsdfd_handle h = nullptr;
    ok(bsdfd_create(&d, &h));
    return static_cast<int64_t>(reinterpret_cast<intptr_t>(h));
}

void destroy(int64_t h) {
    if (h != 0) bsdfd_destroy(as_handle(h));
}

int64_t flops_per_query(int64_t h, int64_t T) { return bsdfd_flops_per_query(as_handle(h), static_cast<int32_t>(T)); }

// ---- operator level: rendering/utils/mlp_brdf_sampling.py:17,69,106,144 ----------------------------
std::tuple<Tensor, Tensor> network_sampling(int64_t h, const Tensor& omega_i, const std::optional<Tensor>& x0, int64_t seed,
                                            int64_t offset, int64_t T) {
    const at::Device dev = omega_i.device();
    const float* wi = in2d(omega_i, 2, "omega_i", dev);
    const int64_t n = omega_i.size(0);
    const float* x0p = opt2d(x0, 2, "x0", dev, n);
    Launch L(dev);
    Tensor x = empty_f32({n, 2}, dev), pdf = empty_f32({n}, dev);
    ok(bsdfd_network_sampling(as_handle(h), wi, x0p, static_cast<uint64_t>(seed), static_cast<uint64_t>(offset), n,
                              static_cast<int32_t>(T), x.data_ptr<float>(), pdf.data_ptr<float>(), L.stream));
    return {x, pdf};
}

Tensor network_pdf(int64_t h, const Tensor& omega_o, const Tensor& omega_i, int64_t T) {
    const at::Device dev = omega_i.device();
    const float* wi = in2d(omega_i, 2, "omega_i", dev);
    const int64_t n = omega_i.size(0);
    const float* wo = in2d(omega_o, 2, "omega_o", dev, n);
    Launch L(dev);
    Tensor pdf = empty_f32({n}, dev);
    ok(bsdfd_network_pdf(as_handle(h), wo, wi, n, static_cast<int32_t>(T), pdf.data_ptr<float>(), L.stream));
    return pdf;
}

Tensor flow_samples_only(int64_t h, const Tensor& omega_i, const Tensor& x0, int64_t T) {
    const at::Device dev = omega_i.device();
    const float* wi = in2d(omega_i, 2, "omega_i", dev);
    const int64_t n = omega_i.size(0);
    const float* x0p = in2d(x0, 2, "x0", dev, n);
    Launch L(dev);
    Tensor x = empty_f32({n, 2}, dev);
    ok(bsdfd_flow_samples_only(as_handle(h), wi, x0p, n, static_cast<int32_t>(T), x.data_ptr<float>(), L.stream));
    return x;
}

// ---- plugin level: tensor core of MyBSDF.sample / MyBSDF.pdf ----------------------------------------
std::tuple<Tensor, Tensor> plugin_sample(int64_t h, int64_t variant, const Tensor& wi, const std::optional<Tensor>& x0,
                                         int64_t seed, int64_t offset, int64_t T) {
    const at::Device dev = wi.device();
    const float* wip = in2d(wi, 3, "wi", dev);
    const int64_t n = wi.size(0);
    const float* x0p = opt2d(x0, 2, "x0", dev, n);
    Launch L(dev);
    Tensor wo = empty_f32({n, 3}, dev), pdf = empty_f32({n}, dev);
    ok(bsdfd_plugin_sample(as_handle(h), static_cast<int32_t>(variant), wip, x0p, static_cast<uint64_t>(seed),
                           static_cast<uint64_t>(offset), n, static_cast<int32_t>(T), wo.data_ptr<float>(),
                           pdf.data_ptr<float>(), L.stream));
    return {wo, pdf};
}

Tensor plugin_pdf(int64_t h, int64_t variant, const Tensor& wi, const Tensor& wo, int64_t T) {
    const at::Device dev = wi.device();
    const float* wip = in2d(wi, 3, "wi", dev);
    const int64_t n = wi.size(0);
    const float* wop = in2d(wo, 3, "wo", dev, n);
    Launch L(dev);
    Tensor pdf = empty_f32({n}, dev);
    ok(bsdfd_plugin_pdf(as_handle(h), static_cast<int32_t>(variant), wip, wop, n, static_cast<int32_t>(T),
                        pdf.data_ptr<float>(), L.stream));
    return pdf;
}

std::tuple<Tensor, Tensor, Tensor> plugin_sample_pdf(int64_t h, int64_t variant, const Tensor& wi, const Tensor& wl,
                                                     const std::optional<Tensor>& x0, int64_t seed, int64_t offset, int64_t T) {
    const at::Device dev = wi.device();
    const float* wip = in2d(wi, 3, "wi", dev);
    const int64_t n = wi.size(0);
    const float* wlp = in2d(wl, 3, "wl", dev, n);
    const float* x0p = opt2d(x0, 2, "x0", dev, n);
    Launch L(dev);
    Tensor wo = empty_f32({n, 3}, dev), pdf_o = empty_f32({n}, dev), pdf_l = empty_f32({n}, dev);
    ok(bsdfd_plugin_sample_pdf(as_handle(h), static_cast<int32_t>(variant), wip, x0p, wlp, static_cast<uint64_t>(seed),
                               static_cast<uint64_t>(offset), n, static_cast<int32_t>(T), wo.data_ptr<float>(),
                               pdf_o.data_ptr<float>(), pdf_l.data_ptr<float>(), L.stream));
    return {wo, pdf_o, pdf_l};
}

// out-variants for callers that own their buffers (a renderer's wavefront arrays): nothing is allocated
void plugin_sample_out(int64_t h, int64_t variant, const Tensor& wi, const std::optional<Tensor>& x0, int64_t seed, int64_t offset,
                       int64_t T, Tensor wo, Tensor pdf) {
    const at::Device dev = wi.device();
    const float* wip = in2d(wi, 3, "wi", dev);
    const int64_t n = wi.size(0);
    const float* x0p = opt2d(x0, 2, "x0", dev, n);
    in2d(wo, 3, "wo (out)", dev, n);
    TORCH_CHECK(pdf.device() == dev && pdf.scalar_type() == at::kFloat && pdf.dim() == 1 && pdf.size(0) == n && pdf.is_contiguous(),
                "pdf (out) must be a contiguous float32 tensor of shape [", n, "] on ", dev);
    Launch L(dev);
    ok(bsdfd_plugin_sample(as_handle(h), static_cast<int32_t>(variant), wip, x0p, static_cast<uint64_t>(seed),
                           static_cast<uint64_t>(offset), n, static_cast<int32_t>(T), wo.data_ptr<float>(),
                           pdf.data_ptr<float>(), L.stream));
}

void plugin_pdf_out(int64_t h, int64_t variant, const Tensor& wi, const Tensor& wo, int64_t T, Tensor pdf) {
    const at::Device dev = wi.device();
    const float* wip = in2d(wi, 3, "wi", dev);
    const int64_t n = wi.size(0);
    const float* wop = in2d(wo, 3, "wo", dev, n);
    TORCH_CHECK(pdf.device() == dev && pdf.scalar_type() == at::kFloat && pdf.dim() == 1 && pdf.size(0) == n && pdf.is_contiguous(),
                "pdf (out) must be a contiguous float32 tensor of shape [", n, "] on ", dev);
    Launch L(dev);
    ok(bsdfd_plugin_pdf(as_handle(h), static_cast<int32_t>(variant), wip, wop, n, static_cast<int32_t>(T),
                        pdf.data_ptr<float>(), L.stream));
}

// ---- per-query context (bsdfd_context_bytes): whichever of sample() / pdf() sees a wi array first writes it, the later
// calls for the same wi read it (`ctx_read` / `ctx_write` select the direction; the defaults are sample-writes, pdf-reads) ------
void* ctx_ptr(const Tensor& ctx, int64_t h, int64_t n, const at::Device& dev) {
    const int64_t need = bsdfd_context_bytes(as_handle(h), n, 1);
    TORCH_CHECK(need >= 0, "bsdfd_context_bytes: bad arguments");
    TORCH_CHECK(ctx.device() == dev && ctx.scalar_type() == at::kFloat && ctx.dim() == 1 && ctx.is_contiguous() &&
                    ctx.numel() * 4 >= need && reinterpret_cast<uintptr_t>(ctx.data_ptr()) % 16 == 0,
                "context must be a contiguous, 16-byte aligned float32 tensor of >= ", need / 4, " elements on ", dev);
    return ctx.data_ptr();
}

int64_t context_floats(int64_t h, int64_t n, int64_t n_segments) {
    const int64_t b = bsdfd_context_bytes(as_handle(h), n, static_cast<int32_t>(n_segments));
    TORCH_CHECK(b >= 0, "bsdfd_context_bytes: bad arguments");
    return b / 4;
}

// row_index (bsdfd_opts.row_index, ABI 6): int64 [n] — the call processes n rows and reads / writes row row_index[i] of the
// [M, .] arrays (M >= n rows; entries must be distinct and < M — not checked, they live on the device)
const int64_t* index_ptr(const std::optional<Tensor>& t, const char* name, const at::Device& dev, int64_t n = -1) {
    if (!t.has_value()) return nullptr;
    const Tensor& r = *t;
    TORCH_CHECK(r.device() == dev && r.scalar_type() == at::kLong && r.dim() == 1 && (n < 0 || r.size(0) == n) && r.is_contiguous(),
                name, " must be a contiguous int64 tensor", n < 0 ? "" : " of the call's row count", " on ", dev);
    return r.data_ptr<int64_t>();
}

void plugin_sample_ex_out(int64_t h, int64_t variant, const Tensor& wi, const std::optional<Tensor>& x0, int64_t seed,
                          int64_t offset, int64_t T, Tensor wo, Tensor pdf, const std::optional<Tensor>& ctx,
                          const std::optional<Tensor>& rng_index, bool ctx_read, const std::optional<Tensor>& row_index) {
    const at::Device dev = wi.device();
    const float* wip = in2d(wi, 3, "wi", dev);
    const int64_t m = wi.size(0);
    bsdfd_opts o{};
    o.row_index = index_ptr(row_index, "row_index", dev);
    const int64_t n = row_index.has_value() ? row_index->size(0) : m;
    TORCH_CHECK(n <= m, "row_index names ", n, " rows, wi has ", m);
    const float* x0p = opt2d(x0, 2, "x0", dev, m);
    in2d(wo, 3, "wo (out)", dev, m);
    TORCH_CHECK(pdf.device() == dev && pdf.scalar_type() == at::kFloat && pdf.dim() == 1 && pdf.size(0) == m && pdf.is_contiguous(),
                "pdf (out) must be a contiguous float32 tensor of shape [", m, "] on ", dev);
    if (ctx.has_value()) {
        if (ctx_read) o.ctx_in = ctx_ptr(*ctx, h, n, dev);
        else o.ctx_out = ctx_ptr(*ctx, h, n, dev);
    }
    o.rng_index = index_ptr(rng_index, "rng_index", dev, n);
    Launch L(dev);
    ok(bsdfd_plugin_sample_ex(as_handle(h), static_cast<int32_t>(variant), wip, x0p, static_cast<uint64_t>(seed),
                              static_cast<uint64_t>(offset), n, static_cast<int32_t>(T), wo.data_ptr<float>(),
                              pdf.data_ptr<float>(), &o, L.stream));
}

void plugin_pdf_ex_out(int64_t h, int64_t variant, const Tensor& wi, const Tensor& wo, int64_t T, Tensor pdf,
                       const std::optional<Tensor>& ctx, bool ctx_write, const std::optional<Tensor>& row_index) {
    const at::Device dev = wi.device();
    const float* wip = in2d(wi, 3, "wi", dev);
    const int64_t m = wi.size(0);
    bsdfd_opts o{};
    o.row_index = index_ptr(row_index, "row_index", dev);
    const int64_t n = row_index.has_value() ? row_index->size(0) : m;
    TORCH_CHECK(n <= m, "row_index names ", n, " rows, wi has ", m);
    const float* wop = in2d(wo, 3, "wo", dev, m);
    TORCH_CHECK(pdf.device() == dev && pdf.scalar_type() == at::kFloat && pdf.dim() == 1 && pdf.size(0) == m && pdf.is_contiguous(),
                "pdf (out) must be a contiguous float32 tensor of shape [", m, "] on ", dev);
    if (ctx.has_value()) {
        if (ctx_write) o.ctx_out = ctx_ptr(*ctx, h, n, dev);
        else o.ctx_in = ctx_ptr(*ctx, h, n, dev);
    }
    Launch L(dev);
    ok(bsdfd_plugin_pdf_ex(as_handle(h), static_cast<int32_t>(variant), wip, wop, n, static_cast<int32_t>(T),
                           pdf.data_ptr<float>(), &o, L.stream));
}

}  // namespace

TORCH_LIBRARY(bsdfd, m) {
    m.def("create_from_file(str path, int precision, int device_index) -> int", &create_from_file);
    m.def("create(int domain, int width, int n_hidden, int precision, Tensor w_in, Tensor w_hidden, Tensor w_out, Tensor base_w1, "
          "Tensor base_b1, Tensor base_w2, Tensor base_b2, int device_index) -> int", &create);
    m.def("destroy(int handle) -> ()", &destroy);
    m.def("flops_per_query(int handle, int T) -> int", &flops_per_query);
    m.def("network_sampling(int handle, Tensor omega_i, Tensor? x0, int seed, int offset, int T) -> (Tensor, Tensor)", &network_sampling);
    m.def("network_pdf(int handle, Tensor omega_o, Tensor omega_i, int T) -> Tensor", &network_pdf);
    m.def("flow_samples_only(int handle, Tensor omega_i, Tensor x0, int T) -> Tensor", &flow_samples_only);
    m.def("plugin_sample(int handle, int variant, Tensor wi, Tensor? x0, int seed, int offset, int T) -> (Tensor, Tensor)", &plugin_sample);
    m.def("plugin_pdf(int handle, int variant, Tensor wi, Tensor wo, int T) -> Tensor", &plugin_pdf);
    m.def("plugin_sample_pdf(int handle, int variant, Tensor wi, Tensor wl, Tensor? x0, int seed, int offset, int T) -> (Tensor, Tensor, Tensor)",
          &plugin_sample_pdf);
    m.def("plugin_sample_out(int handle, int variant, Tensor wi, Tensor? x0, int seed, int offset, int T, Tensor(a!) wo, Tensor(b!) pdf) -> ()",
          &plugin_sample_out);
    m.def("plugin_pdf_out(int handle, int variant, Tensor wi, Tensor wo, int T, Tensor(a!) pdf) -> ()", &plugin_pdf_out);
    m.def("context_floats(int handle, int n, int n_segments) -> int", &context_floats);
    m.def("plugin_sample_ex_out(int handle, int variant, Tensor wi, Tensor? x0, int seed, int offset, int T, Tensor(a!) wo, "
          "Tensor(b!) pdf, Tensor(c!)? ctx, Tensor? rng_index, bool ctx_read=False, Tensor? row_index=None) -> ()", &plugin_sample_ex_out);
    m.def("plugin_pdf_ex_out(int handle, int variant, Tensor wi, Tensor wo, int T, Tensor(a!) pdf, Tensor(b!)? ctx, bool ctx_write=False, "
          "Tensor? row_index=None) -> ()", &plugin_pdf_ex_out);
}
