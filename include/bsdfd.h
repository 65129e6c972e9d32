/*
 * bsdfd.h — C ABI of the MI355X-native neural-BSDF flow sampler ("bsdfd").
 *
 * This is the drop-in boundary for the ONE hot path of fzy28/BSDF_diffusion_sampling:
 * the per-query rectified-flow sampler and its change-of-variables PDF behind the
 * Mitsuba plugin methods sample()/pdf().  Plain pointers and sizes only — no torch
 * types.  Every entry point names the reference interface it replaces
 * (paths relative to the reference repo root).
 *
 * Conventions (modelled on how the reference binds its only native library,
 * tiny-cuda-nn/bindings/torch/tinycudann/bindings.cpp:79-110):
 *   - all data pointers are DEVICE pointers to contiguous row-major fp32, caller-owned;
 *   - no allocation inside sample/pdf calls; any N >= 0 (ragged tail handled in-kernel);
 *   - work is enqueued on the HIP stream the caller passes (NULL = default stream)
 *     and the call returns without synchronising;
 *   - return value 0 = ok, otherwise a BSDFD_E* code; bsdfd_last_error() returns a
 *     thread-local message (the Python host raises RuntimeError with it);
 *   - a handle is immutable after create and bound to the device current at create;
 *     calls are re-entrant across host threads and streams (the optional launch-timing
 *     counters of bsdfd_set_profiling are the only mutable state and are mutex-guarded).
 */
#ifndef BSDFD_H
#define BSDFD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version of this header: bumped whenever a struct grows or an entry point changes.  Structs (bsdfd_desc, bsdfd_opts,
 * bsdfd_wf_scene) may grow AT THE END between versions and MUST be zero-initialised by the caller (memset / `= {0}`): a zero
 * field always selects the behaviour of the version that did not have it.  A host checks bsdfd_abi_version() ==
 * BSDFD_ABI_VERSION once after loading the library (the Python hosts do: _lib.lib()).
 *   5: bsdfd_desc.reserved became bsdfd_desc.tile (values other than 0 / 16 / 32 are rejected, 0 = library default).
 *   6: bsdfd_opts.row_index appended; bsdfd_abi_version() added. */
#define BSDFD_ABI_VERSION 6

#define BSDFD_OK 0
#define BSDFD_EINVAL 1   /* bad argument / unsupported architecture */
#define BSDFD_EHIP 2     /* a HIP runtime call failed */
#define BSDFD_EIO 3      /* weight file unreadable / malformed */

#define BSDFD_DOMAIN_DISK 0       /* state = 2-D point on the unit disk          */
#define BSDFD_DOMAIN_SPHERICAL 1  /* state = (theta, phi); net input [theta, sin phi, cos phi] */

/* Arithmetic of the dense layer contractions.  Activations, Jacobian determinant, base density and warps are fp32 VALU —
 * with ONE exception: bsdfd_flow_samples_only in BSDFD_PREC_F16 (both tilings) evaluates the hidden layers' sigmoids in
 * packed fp16 as well (see BSDFD_PREC_F16). */
#define BSDFD_PREC_DEFAULT 0  /* = BSDFD_PREC_SPLIT3 */
#define BSDFD_PREC_F32 1      /* v_mfma_f32_16x16x4_f32: exact fp32 FMA chains (validation mode) */
#define BSDFD_PREC_SPLIT3 2   /* fp16 MFMA, operands split hi+lo, 3 products, fp32 accumulate (<=1e-4 parity).
                               * Range: hidden activations and tangents must stay below fp16's 65504 (they are
                               * O(1..100) for the reference's nets and inputs); beyond it the result is inf/NaN,
                               * never a silently wrong finite number. BSDFD_PREC_F32 has fp32 range. */
#define BSDFD_PREC_F16 3      /* single fp16 MFMA pass (tcnn-class 1e-2 tolerance; reflow teacher sampling).  The kernels of
                               * bsdfd_flow_samples_only in this precision — 16- AND 32-query tiles — also evaluate the hidden
                               * layers' sigmoids in packed fp16 (the 16-query kernels: every hidden layer; the 32-query
                               * kernels keep the last hidden layer's in fp32): that call is its own fp16-class evaluation and
                               * does NOT walk the trajectory of bsdfd_network_sampling(BSDFD_PREC_F16) bit for bit (agreement
                               * ~2e-2, the class's tolerance).  Range: |pre-activation| must stay below ~4.5e4 (fp16's
                               * largest finite value); beyond it sigma evaluates inf * 0 = NaN, never a wrong finite number. */

/* Plugin post-processing variants (which MyBSDF the call mirrors). */
#define BSDFD_PLUGIN_MEASURED 0    /* rendering/brdf_measured_{disk,spherical}.py */
#define BSDFD_PLUGIN_FULLSPHERE 1  /* rendering/bsdf_myresult.py (spherical only) */

typedef struct bsdfd_ctx* bsdfd_handle;

/* The four kinds of flow-kernel launch (bsdfd_get_tile, bsdfd_profile_read_op). */
#define BSDFD_OP_SAMPLE 0        /* network_sampling, plugin_sample      */
#define BSDFD_OP_PDF 1           /* network_pdf, plugin_pdf              */
#define BSDFD_OP_SAMPLES_ONLY 2  /* flow_samples_only                    */
#define BSDFD_OP_SAMPLE_PDF 3    /* plugin_sample_pdf                    */

/* Host pointers to fp32 row-major [out, in] matrices exactly as nn.Linear.weight
 * stores them (reference checkpoints: rendering/checkpoints_new/.../brdf_rectify_network*.pth
 * and brdf_pretrain_network*.pth, loaded at rendering/brdf_measured_disk.py:43-51).
 * The library copies and re-packs them; the caller may free them after create. */
typedef struct bsdfd_desc {
    int32_t domain;         /* BSDFD_DOMAIN_*                                            */
    int32_t width;          /* hidden width of the velocity net: 32 or 64                */
    int32_t n_hidden;       /* hidden layers: 3 (disk), 4 (spherical), 6 (64-wide teacher) */
    int32_t pe_bands;       /* positional-encoding bands of the velocity net (5)         */
    int32_t base_hidden;    /* hidden width of the base-density net (16)                 */
    int32_t base_pe_bands;  /* positional-encoding bands of the base net (3)             */
    int32_t precision;      /* BSDFD_PREC_*                                              */
    int32_t tile;           /* queries per wave64 tile: 0 = library default (32 where such a kernel exists), 16 = the
                             * 16x16x32-MFMA kernels (every net), 32 = the 32x32x16-MFMA kernels, which exist for
                             *   (1) the reference's two plugin nets — disk 32x3, spherical 32x4 — in BSDFD_PREC_SPLIT3 (all calls),
                             *   (2) bsdfd_flow_samples_only of the 64 x 6 spherical teacher in BSDFD_PREC_F16,
                             *   (3) bsdfd_flow_samples_only of those two 32-wide nets in BSDFD_PREC_F16,
                             *   (4) the Jacobian calls (sampling / pdf, operator and plugin level) of the 64 x 6 spherical net in
                             *       BSDFD_PREC_SPLIT3 — OPT-IN: served only when 32 is asked for explicitly (it measured 4 % slower
                             *       than the 16-query kernel, which stays this net's default).
                             * An explicit 32 for a (net, precision) with no such kernel at all is REJECTED by bsdfd_create
                             * (BSDFD_EINVAL); where only some calls have one — (2), (3), (4) — the others run 16-query tiles and
                             * bsdfd_get_tile says which.  Product behaviour depends on this field alone: the library reads
                             * no environment variable (the Python hosts map $BSDFD_TILE onto it for A/B runs).
                             * Both tilings implement the same operators to the same tolerance.  Was `reserved` before ABI 5:
                             * zero-initialise the struct. */
    const float* w_in;      /* [width, state_dim + 1 + 2 + 4*pe_bands], cols [state|alpha|PE(omega_i)] */
    const float* w_hidden;  /* [n_hidden-1, width, width]                                */
    const float* w_out;     /* [2, width]                                                */
    const float* base_w1;   /* [base_hidden, 2 + 4*base_pe_bands]                        */
    const float* base_b1;   /* [base_hidden]                                             */
    const float* base_w2;   /* [4, base_hidden]                                          */
    const float* base_b2;   /* [4]                                                       */
} bsdfd_desc;

/* Replaces MyBSDF.__init__'s network construction + load_state_dict
 * (rendering/brdf_measured_disk.py:43-51, brdf_measured_spherical.py:53-59,
 * bsdf_myresult.py:49-54). */
int bsdfd_create(const bsdfd_desc* desc, bsdfd_handle* out);

/* Same, from a neutral .bsdfw file (format: bsdf_diffusion_sampling_amd/weights.py);
 * replaces torch.load of the pickle checkpoints. precision = BSDFD_PREC_*. */
int bsdfd_create_from_file(const char* path, int32_t precision, bsdfd_handle* out);

void bsdfd_destroy(bsdfd_handle h);

/* Introspection (host side mirrors / bench): domain, width, n_hidden, precision in
 * effect; algorithmic flop per query for T Euler steps. */
int bsdfd_get_info(bsdfd_handle h, int32_t* domain, int32_t* width, int32_t* n_hidden,
                   int32_t* precision);
int64_t bsdfd_flops_per_query(bsdfd_handle h, int32_t T);
/* Queries per wave64 tile (16 or 32, see bsdfd_desc.tile) of the kernel this handle launches for `op` (BSDFD_OP_*).  For
 * BSDFD_OP_SAMPLE / _PDF it is the granularity of the opaque per-query context below.  (A handle may mix tilings: the 64 x 6 teacher
 * in BSDFD_PREC_F16 has a 32-query-tile kernel for bsdfd_flow_samples_only alone.)  No counterpart in the reference. */
int bsdfd_get_tile(bsdfd_handle h, int32_t op, int32_t* tile);

/* network_sampling_disk / network_sampling_spherical
 * (rendering/utils/mlp_brdf_sampling.py:17-51, :106-140).
 *   omega_i [N,2]  condition: disk coords of wi, or (theta_i, phi_i)
 *   x0      [N,2]  base draw, or NULL: drawn in-kernel from D_base with a Philox4x32-10
 *                  stream keyed (seed, offset + query index) — statistically, not
 *                  bit-wise, equal to torch's RNG
 *   x_out   [N,2]  flowed sample;  pdf_out [N]  p0(x0) * prod 1/det(I + J/T), signed. */
int bsdfd_network_sampling(bsdfd_handle h, const float* omega_i, const float* x0, uint64_t seed,
                           uint64_t offset, int64_t N, int32_t T, float* x_out, float* pdf_out,
                           void* hip_stream);

/* network_pdf_disk / network_pdf_spherical (mlp_brdf_sampling.py:69-103, :144-181). */
int bsdfd_network_pdf(bsdfd_handle h, const float* omega_o, const float* omega_i, int64_t N,
                      int32_t T, float* pdf_out, void* hip_stream);

/* Tensor core of MyBSDF.sample with the domain warp and guards fused
 * (rendering/brdf_measured_disk.py:59-82; brdf_measured_spherical.py:69-91;
 *  bsdf_myresult.py:59-84 when variant = BSDFD_PLUGIN_FULLSPHERE):
 *   wi [N,3] unit vectors in the local frame -> wo [N,3], pdf_sa [N] (solid-angle pdf).
 * The measured.eval()-dependent firefly rule (:97-100) is NOT applied here. */
int bsdfd_plugin_sample(bsdfd_handle h, int32_t variant, const float* wi, const float* x0,
                        uint64_t seed, uint64_t offset, int64_t N, int32_t T, float* wo,
                        float* pdf_sa, void* hip_stream);

/* Tensor core of MyBSDF.pdf (rendering/brdf_measured_disk.py:112-124;
 * brdf_measured_spherical.py:122-137; bsdf_myresult.py:115-133). */
int bsdfd_plugin_pdf(bsdfd_handle h, int32_t variant, const float* wi, const float* wo, int64_t N,
                     int32_t T, float* pdf_sa, void* hip_stream);

/* sample(wi) and pdf(wi, wl) for the SAME intersections in one launch: the per-query prologue
 * (positional encoding, conditioning term of layer 1, base-density net) is evaluated once and the flow
 * runs twice (forward from the base draw, reverse from wl).  This is the call pattern of a renderer
 * with next-event estimation (one sample() and one pdf() per path and bounce); results are identical
 * to bsdfd_plugin_sample followed by bsdfd_plugin_pdf(wi, wl). */
int bsdfd_plugin_sample_pdf(bsdfd_handle h, int32_t variant, const float* wi, const float* x0, const float* wl,
                            uint64_t seed, uint64_t offset, int64_t N, int32_t T, float* wo, float* pdf_wo,
                            float* pdf_wl, void* hip_stream);

/* Per-query context: what the kernel derives from wi ALONE — the conditioning term of layer 1
 * (W1[:, PE(omega_i)] PE(omega_i), once per query instead of once per Euler step as rendering/utils/model.py:494
 * recomputes it) and the base-density net's four outputs (evaluated twice per sample() by
 * rendering/utils/mlp_brdf_sampling.py:20,24) — 144 B per query for the 32-wide nets, 272 B for 64-wide.
 * A renderer calls sample(si) and pdf(si, wl) for the SAME intersections (rendering/brdf_measured_disk.py:59,112 are
 * both driven by one `si`) — Mitsuba's path integrator in the order eval_pdf() (emitter sampling, :126 -> :112) first,
 * sample() (:59) second — so EITHER call may write the context while it runs (opts->ctx_out) and either may read it
 * (opts->ctx_in) instead of re-evaluating the prologue (cart_to_spher(wi), 22 sin/cos, the conditioning contraction and the
 * base net: 20 fp32 MFMAs per 16 queries).  A call takes at most one of the two.
 * Results are BIT-IDENTICAL to the calls without a context.  The buffer is opaque device memory of
 * bsdfd_context_bytes(h, N, n_segments) bytes (n_segments = 1 for single-material calls, = n_handles for *_multi
 * calls), 16-byte aligned, valid only for the handle(s), the wi array and — for *_multi — the seg_end layout it
 * was written with; T may differ between the two calls.  NULL context = the plain call. */
int64_t bsdfd_context_bytes(bsdfd_handle h, int64_t N, int32_t n_segments);

/* Optional arguments of the plugin-level calls (NULL pointer / all-zero struct = the plain call). */
typedef struct bsdfd_opts {
    void* ctx_out;             /* sample / pdf calls: also write the per-query context of this wi array here        */
    const void* ctx_in;        /* sample / pdf calls: read the context an earlier call wrote for the same wi array  */
    const int64_t* rng_index;  /* sample calls: device array [N]; the Philox counter of row i is offset +
                                * rng_index[i] instead of offset + i.  A wavefront that was bucketed by material
                                * passes the rows' ORIGINAL lane indices here: the base draws then depend on neither
                                * the bucketing nor the sharding / GPU count.                */
    const int64_t* row_index;  /* sample / pdf / sample_pdf calls (ABI 6): device array [N]; row i of the call READS its
                                * inputs (wi, x0, wo / wl) at row row_index[i] of the callers' arrays and WRITES its outputs
                                * there.  A wavefront bucketed by material hands over the bucket permutation
                                * (bsdfd_bucket_by_material's perm) with the callers' lane-ordered arrays: no gathered copy of
                                * the inputs, no scatter of the results (the dispatch it replaces: one plugin instance per
                                * material called on its lanes, rendering/matpreview/disney_bsdf_array0_envmap.xml +
                                * rendering/brdf_measured_disk.py:140).  seg_end, the per-query context and rng_index stay
                                * indexed by i; without rng_index the Philox counter of row i is offset + row_index[i].
                                * Rows not named by row_index are not touched.  Entries must be distinct and inside the
                                * callers' arrays; the library cannot check either (it is not told the arrays' lengths;
                                * the Python hosts do under BSDFD_CHECK_INDEX=1).                                      */
} bsdfd_opts;
int bsdfd_plugin_sample_ex(bsdfd_handle h, int32_t variant, const float* wi, const float* x0, uint64_t seed,
                           uint64_t offset, int64_t N, int32_t T, float* wo, float* pdf_sa, const bsdfd_opts* opts,
                           void* hip_stream);
int bsdfd_plugin_pdf_ex(bsdfd_handle h, int32_t variant, const float* wi, const float* wo, int64_t N, int32_t T,
                        float* pdf_sa, const bsdfd_opts* opts, void* hip_stream);

/* Mixed-material batches (BASELINE.json configs[3]; the reference binds one plugin instance per material,
 * rendering/matpreview/disney_bsdf_array0_envmap.xml, and Mitsuba calls each instance on its lanes).
 * The query arrays are BUCKETED by material: bucket i = rows [seg_end[i-1], seg_end[i]) (seg_end is a HOST
 * array of n_handles cumulative ends; empty buckets allowed) is served by handles[i].  All handles must
 * share domain, width, depth, precision and device; ONE launch serves up to 64 buckets (workgroups are
 * dealt to buckets in proportion to their sizes, each loads its material's weight image into LDS).
 * The Philox counter of a row is offset + its row index in the bucketed array. */
int bsdfd_plugin_sample_multi(const bsdfd_handle* handles, int32_t n_handles, const int64_t* seg_end,
                              int32_t variant, const float* wi, const float* x0, uint64_t seed,
                              uint64_t offset, int32_t T, float* wo, float* pdf_sa, void* hip_stream);
int bsdfd_plugin_pdf_multi(const bsdfd_handle* handles, int32_t n_handles, const int64_t* seg_end,
                           int32_t variant, const float* wi, const float* wo, int32_t T, float* pdf_sa,
                           void* hip_stream);
/* bsdfd_plugin_sample_pdf over material buckets (same bucket layout as bsdfd_plugin_sample_multi) */
int bsdfd_plugin_sample_pdf_multi(const bsdfd_handle* handles, int32_t n_handles, const int64_t* seg_end,
                                  int32_t variant, const float* wi, const float* x0, const float* wl,
                                  uint64_t seed, uint64_t offset, int32_t T, float* wo, float* pdf_wo,
                                  float* pdf_wl, void* hip_stream);

/* the *_multi calls with optional arguments (bsdfd_opts; a context spans n_segments = n_handles buckets) */
int bsdfd_plugin_sample_multi_ex(const bsdfd_handle* handles, int32_t n_handles, const int64_t* seg_end,
                                 int32_t variant, const float* wi, const float* x0, uint64_t seed, uint64_t offset,
                                 int32_t T, float* wo, float* pdf_sa, const bsdfd_opts* opts, void* hip_stream);
int bsdfd_plugin_pdf_multi_ex(const bsdfd_handle* handles, int32_t n_handles, const int64_t* seg_end,
                              int32_t variant, const float* wi, const float* wo, int32_t T, float* pdf_sa,
                              const bsdfd_opts* opts, void* hip_stream);
/* bsdfd_plugin_sample_pdf_multi with opts->rng_index (the fused call takes no context: its prologue stays in registers) */
int bsdfd_plugin_sample_pdf_multi_ex(const bsdfd_handle* handles, int32_t n_handles, const int64_t* seg_end,
                                     int32_t variant, const float* wi, const float* x0, const float* wl, uint64_t seed,
                                     uint64_t offset, int32_t T, float* wo, float* pdf_wo, float* pdf_wl,
                                     const bsdfd_opts* opts, void* hip_stream);

/* Reflow teacher sampling without the Jacobian: x <- x + v(x, t/T | omega_i)/T for T steps
 * (learning_repo_cleanup/spherical_domain_sampling.py:147-166, disk_domain_sampling.py:93-110 —
 * the reference's only tiny-cuda-nn call site). x0 [N,2] in, x_out [N,2] out. */
int bsdfd_flow_samples_only(bsdfd_handle h, const float* omega_i, const float* x0, int64_t N,
                            int32_t T, float* x_out, void* hip_stream);

/* Kernel timing with HIP events ON THE LAUNCH STREAM: while profiling is enabled every launch
 * through this handle is bracketed by an event pair recorded on the stream the kernel is
 * launched on (a ring of 64 pairs; no host synchronisation unless the ring wraps onto a launch
 * that is still running).  bsdfd_set_profiling resets the counters.
 * bsdfd_profile_read synchronises on the outstanding stop events and returns the number of
 * launches and the sum of their durations (ms) since profiling was enabled;
 * bsdfd_last_kernel_ms returns the duration of the most recent launch (negative if none).
 * bsdfd_set_profiling / bsdfd_profile_* synchronise and touch device memory from the host: not while a stream is being captured
 * (launches THROUGH a profiling handle are capture-safe only with profiling off: event records inside a capture become graph nodes). */
int bsdfd_set_profiling(bsdfd_handle h, int32_t enable);
int bsdfd_profile_read(bsdfd_handle h, int64_t* n_launches, double* total_ms);
float bsdfd_last_kernel_ms(bsdfd_handle h);
/* The same totals for ONE kind of launch (bench.py's sample / pdf split of the timed region itself): `op` is
 * BSDFD_OP_SAMPLE (network_sampling, plugin_sample), BSDFD_OP_PDF (network_pdf, plugin_pdf), BSDFD_OP_SAMPLES_ONLY
 * (flow_samples_only) or BSDFD_OP_SAMPLE_PDF (plugin_sample_pdf).  No counterpart in the reference. */
int bsdfd_profile_read_op(bsdfd_handle h, int32_t op, int64_t* n_launches, double* total_ms);
/* Shader clock (MHz) the chip sustained UNDER THE PROFILED LAUNCHES THEMSELVES: while profiling is enabled every wave adds its
 * lifetime in shader cycles (s_memtime) and in ticks of the constant-rate wall clock (s_memrealtime,
 * hipDeviceAttributeWallClockRate) to two words of the handle — both read by the same wave, so counter offsets between CUs
 * cancel; the ratio of the sums x the wall-clock rate is the clock, averaged over the waves' lifetimes.  This is what
 * bench.py's issue-bound entry divides by (the stand-alone probe below runs a similar instruction mix, not the kernel itself,
 * and may draw a different clock: "DVFS give-back", MI355X_MICROARCH.md).  0 if nothing was recorded.  No counterpart in the
 * reference. */
int bsdfd_profile_clock_mhz(bsdfd_handle h, double* mhz);

/* Shader clock (MHz) the chip sustains under the flow kernel's instruction mix: a ~6 ms probe launch on every
 * CU, shader-cycle count of the LONGEST-lived wave (maximum over all waves) / HIP-event duration (csrc/clock.hip).
 * The probe allocates, copies back and SYNCHRONISES: it must not be called while a stream is being captured
 * (the rest of the API is capture-safe).  Measurement only: bench.py
 * uses it to state kernel time in shader cycles per (16-query tile x Euler step) next to the instruction-issue
 * model of that loop (the "issue-bound" roofline entry).  No counterpart in the reference. */
int bsdfd_shader_clock_mhz(double* mhz, void* hip_stream);

/* Stand-alone positional encoding = the reference's positional_encoding_1 (rendering/utils/model.py:9-57):
 * x [N,dim] -> out [N, dim * (include_input + 2 * bands)] = cat([x], sin(f_0 x), cos(f_0 x), ...), f = 2^b
 * (log_sampling) or linspace(1, 2^(bands-1), bands).  Fused (and free) inside the flow kernel; this
 * un-fused pass is the drop-in for the reference function and the HBM-bound "encoding pass". */
int bsdfd_positional_encoding(const float* x, int64_t N, int32_t dim, int32_t bands, int32_t include_input,
                              int32_t log_sampling, float* out, void* hip_stream);

/* ---- bucketing of a material-tagged wavefront (config 4) ---------------------------------------------
 * Stable counting sort of N lanes by material id (int64, 0 <= id < n_materials <= 64), on the device:
 * perm [N] gathers lanes into bucket order (torch.argsort(stable=True) semantics), counts [n_materials]
 * are the bucket sizes (device memory).  `workspace`: device scratch of bsdfd_bucket_workspace_bytes().
 * Lanes with an id outside the range are left out of perm (counts then sum to less than N). */
int64_t bsdfd_bucket_workspace_bytes(int64_t N, int32_t n_materials);
int bsdfd_bucket_by_material(const int64_t* material_id, int64_t N, int32_t n_materials, int64_t* perm,
                             int64_t* counts, void* workspace, int64_t workspace_bytes, void* hip_stream);

/* Bucket order <-> lane order around a run of bucketed calls (a renderer keeps a wavefront in bucket order across its
 * sample() and pdf() calls): wi_b[i] = wi[perm[i]] for the first n entries of perm (the lanes that carry a material), and
 * the inverse for the results, all arrays of a wavefront in ONE pass (NULL = skip that array): wo[perm[i]] = wo_b[i],
 * pdf[perm[i]] = pdf_b[i], pdf2[perm[i]] = pdf2_b[i].  Lanes not named by perm[0..n) are left untouched. */
int bsdfd_gather_lanes(const int64_t* perm, int64_t n, const float* wi, float* wi_b, void* hip_stream);
int bsdfd_scatter_lanes(const int64_t* perm, int64_t n, const float* wo_b, const float* pdf_b, const float* pdf2_b,
                        float* wo, float* pdf, float* pdf2, void* hip_stream);

/* ---- ground-truth evaluator for eval(): RGL measured BSDF (rgb tensor files) -----------------------
 * Replaces, for the plugins' eval() / sample-weight / firefly rule, the Mitsuba `measured` BSDF the
 * reference builds in rendering/brdf_measured_disk.py:36-42 (`mi.load_dict({'type': 'measured',
 * 'filename': 'measuredbsdfs/<name>.bsdf'})`) and evaluates at :96,107.  Model: Dupuy & Jakob 2018
 * (csrc/measured.hip).  rgb_out [N,3] = f(wi, wo) * cos(theta_o), 0 where cos(theta_i) <= 0 or
 * cos(theta_o) <= 0 — Mitsuba's eval() convention. */
typedef struct bsdfd_measured_ctx* bsdfd_measured_handle;
int bsdfd_measured_create_from_file(const char* path, bsdfd_measured_handle* out);
void bsdfd_measured_destroy(bsdfd_measured_handle h);
int bsdfd_measured_get_info(bsdfd_measured_handle h, int32_t* n_phi, int32_t* n_theta, int32_t* isotropic,
                            int32_t* jacobian, int32_t* reduction);
/* tint: host pointer to 3 floats (the plugin's albedo) or NULL */
int bsdfd_measured_eval(bsdfd_measured_handle h, const float* wi, const float* wo, int64_t N, const float* tint,
                        float* rgb_out, void* hip_stream);
/* The tail of the plugins' sample() in one pass (rendering/brdf_measured_disk.py:89-101,
 * brdf_measured_spherical.py:97-109): value = f * tint / pdf_sa on active lanes (cos(theta_i) > 0, and
 * active[q] != 0 if a mask is given) with pdf_sa > 0; firefly rule pdf_out = lum(value) < threshold ? pdf_sa : 0;
 * weight_out [N,3] = value where active, pdf_out > 0 and cos(theta_o) > 0, else 0. */
int bsdfd_measured_sample_weight(bsdfd_measured_handle h, const float* wi, const float* wo, const float* pdf_sa,
                                 const unsigned char* active, int64_t N, const float* tint, float firefly_threshold,
                                 float* weight_out, float* pdf_out, void* hip_stream);

/* ---- wavefront harness ------------------------------------
 * The reference renders through Mitsuba 3 (rendering/brdf_measured_disk.py:146-155: passes of
 * `mi.render(scene, spp=4, seed)`), whose integrator calls the plugin's sample()/pdf() once per
 * wavefront; rendering/utils/mitsuba_helper.py:59-127 restates the primary-ray generation and
 * :130-137 the power-heuristic MIS weight.  These two streaming kernels are the part of that loop
 * around the plugin calls for a minimal scene of the harness' own: pinhole camera, one analytic
 * sphere carrying the material, a lat-long environment map (y up), one bounce.  Path index inside a
 * tile of rows [row_begin, row_end): ((row - row_begin) * width + col) * spp + s; the RNG counter is
 * the global path index, so an image does not depend on the row split over GPUs. */
typedef struct bsdfd_wf_scene {
    float cam_origin[3], cam_right[3], cam_up[3], cam_forward[3]; /* orthonormal camera basis */
    float tan_half_fov;                                            /* of the horizontal field of view */
    int32_t width, height;                                         /* film size in pixels */
    float sphere_center[3];
    float sphere_radius;
    float albedo[3];                                               /* props["albedo"] of the plugin */
    int32_t env_width, env_height;                                 /* environment map [H,W,3] fp32 */
    /* array scenes (matpreview/disney_bsdf_array*.xml: 12 balls with one `mybsdf` material each over a
     * checkerboard floor): ball 0 is the sphere above, balls 1..n_extra_spheres follow; ball k carries
     * material k.  All zero = the single-ball scene. */
    int32_t n_extra_spheres;                                       /* 0..31 */
    float extra_spheres[31][4];                                    /* centre xyz, radius */
    int32_t has_plane;                                             /* diffuse checkerboard floor y = plane_y */
    float plane_y, checker_scale, checker_color0, checker_color1;
} bsdfd_wf_scene;

/* Primary rays of rows [row_begin,row_end), spp jittered samples per pixel, pass index `pass`:
 * wi [N,3] local incoming direction ((0,0,1) for rays that miss the sphere), wl [N,3] cosine-weighted
 * light-sample direction (local), nrm [N,3] world normal (0 for a miss), dir [N,3] world ray direction,
 * material [N] (or NULL): ball index = material index, n_balls for the floor (its reflectance is then
 * in the wi slot), n_balls + 1 for a miss — the ids bsdfd_bucket_by_material sorts. */
int bsdfd_wf_primary(const bsdfd_wf_scene* scene, int32_t row_begin, int32_t row_end, int32_t spp,
                     uint64_t seed, uint64_t pass, float* wi, float* wl, float* nrm, float* dir,
                     int64_t* material, void* hip_stream);
/* One-bounce MIS estimate: wo/pdf_o from plugin sample(), pdf_l = plugin pdf(wi, wl);
 * f_o / f_l [N,3] = plugin eval(wi, wo) / eval(wi, wl) (f cos, albedo included) or both NULL: then the
 * proxy f cos = albedo * pdf is used; a NaN in a path's f_o / f_l entry selects the proxy for that path.  wi / material: the arrays of bsdfd_wf_primary, needed (non-NULL)
 * for scenes with a floor.  film [row_end-row_begin, width, 3] += mean over the spp samples. */
int bsdfd_wf_shade(const bsdfd_wf_scene* scene, const float* env, int32_t row_begin, int32_t row_end,
                   int32_t spp, const float* wo, const float* pdf_o, const float* wl, const float* pdf_l,
                   const float* nrm, const float* dir, const float* f_o, const float* f_l, const float* wi,
                   const int64_t* material, float* film, void* hip_stream);

const char* bsdfd_last_error(void);
const char* bsdfd_version(void);
/* BSDFD_ABI_VERSION the library was compiled with (see the top of this header). */
int32_t bsdfd_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BSDFD_H */
