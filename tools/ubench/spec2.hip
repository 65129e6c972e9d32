// Wave specialisation, priced with the flow kernel's own per-layer instruction mix (round 3).
//
// Question: if one wave per SIMD issues ONLY the 18 MFMAs of a (hidden layer x 16-query tile)
// and partner wave(s) on the same SIMD issue ONLY the ~128 VALU of the activation / hi-lo split work, does the SIMD
// retire more layer-tiles per cycle than when every wave does both (the shipped kernel, `phase.hip` mode 0)?
//
// Every wave runs for a fixed budget of shader cycles and counts the layer-tiles' worth of its own role it
// completed; per SIMD the pipeline could retire min(MFMA-tiles, VALU-tiles).  Roles are assigned from the HARDWARE
// SIMD id (HW_REG_HW_ID[5:4]) and the arrival rank of the wave on that SIMD, not from the wave index.
//   mono      : every wave alternates 18 MFMA / 8 VALU units                    (the shipped structure)
//   spec      : rank 0 of each SIMD = MFMA-only, ranks 1.. = VALU-only           (no hand-off traffic: upper bound)
//   spec+lds  : same, plus the LDS traffic a hand-off needs per layer-tile (no waiting on the partner: still an
//               upper bound): MFMA wave reads 6 B fragments (ds_read_b128) and writes 6 fp32 accumulators
//               (ds_write_b128); VALU wave reads 6 accumulators and writes 6 fragments
//   pipeline  : spec+lds with real flow control through LDS counters: the MFMA wave may run at most D tiles ahead of
//               the VALU waves' completions, a VALU wave starts tile n only once the MFMA wave has finished it
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define MFMA6(c0, c1, c2, c3, c4, c5)                                                       \
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %6, %7, %0\n v_mfma_f32_16x16x32_f16 %1, %6, %7, %1\n" \
                 "v_mfma_f32_16x16x32_f16 %2, %6, %7, %2\n v_mfma_f32_16x16x32_f16 %3, %6, %7, %3\n" \
                 "v_mfma_f32_16x16x32_f16 %4, %6, %7, %4\n v_mfma_f32_16x16x32_f16 %5, %6, %7, %5\n" \
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5) : "v"(h0), "v"(h1))
// one "unit" = the VALU work of 2 hidden units x 3 vectors: sigmoid, silu', tangent scaling, hi/lo split, pack
#define UNIT(a, b, c, d)                                                             \
    asm volatile("v_exp_f32 %0, %0\n v_add_f32 %0, 1.0, %0\n v_rcp_f32 %0, %0\n"    \
                 "v_mul_f32 %1, %0, %1\n v_fma_f32 %2, %0, %1, %2\n v_fma_f32 %3, %1, %2, %0\n" \
                 "v_mul_f32 %1, %3, %1\n v_mul_f32 %2, %3, %2\n"                     \
                 "v_and_b32 %0, 0xffffe000, %1\n v_and_b32 %3, 0xffffe000, %2\n v_and_b32 %1, 0xffffe000, %1\n" \
                 "v_sub_f32 %2, %2, %3\n v_sub_f32 %1, %1, %0\n"                     \
                 "v_cvt_pk_f16_f32 %0, %0, %3\n v_cvt_pk_f16_f32 %1, %1, %2\n v_cvt_pk_f16_f32 %2, %2, %3\n" \
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

struct Res { unsigned long long mfma_tiles, valu_tiles; unsigned simd_rank_hist[16]; };

// MODE 0 mono, 1 spec, 2 spec + LDS traffic, 3 pipeline (flow-controlled), 3 pipeline (flow-controlled)
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, long long budget, Res* res) {
    __shared__ unsigned rank_ctr[4];
    __shared__ unsigned m_done[4], v_done[4];
    if (threadIdx.x < 4) { m_done[threadIdx.x] = 0; v_done[threadIdx.x] = 0; }
    __shared__ __attribute__((aligned(16))) char xfer[16][8 * 1024];  // per wave: 6 KB fragments, 6 KB accumulators (overlapping: only the traffic matters)
    if (threadIdx.x < 4) rank_ctr[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned hwid = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);  // HW_REG_HW_ID bits [5:4] = SIMD id
    const unsigned simd = hwid & 3u;
    unsigned rank = 0;
    if (lane == 0) rank = atomicAdd(&rank_ctr[simd], 1u);
    rank = __builtin_amdgcn_readfirstlane(rank);
    __syncthreads();

    float a[16];
    for (int j = 0; j < 16; ++j) a[j] = threadIdx.x * 1e-3f + j * 0.01f;
    f32x4 c[6];
    for (int j = 0; j < 6; ++j) c[j] = (f32x4){a[0], a[1], a[2], a[3]};
    f16x8 h0, h1;
    for (int j = 0; j < 8; ++j) { h0[j] = (_Float16)(a[0] + j); h1[j] = (_Float16)(a[1] - j); }
    for (int j = 0; j < 6; ++j) asm volatile("" : "+v"(c[j]));  // opaque initial values: no re-materialisation inside the loops
    char* my = xfer[wave] + lane * 16;
    const bool is_mfma = (MODE == 0) || rank == 0;

    unsigned long long nm = 0, nv = 0;
    const unsigned nvw = (blockDim.x >> 8) - 1;  // VALU waves per SIMD in the specialised modes
    const unsigned D = 2 * nvw + 1;
    const long long t0 = __builtin_readcyclecounter();
    // each role runs its OWN loop (no per-iteration merge of the two roles' register state: an earlier version that
    // branched on the role inside one loop made the compiler shuffle the 24 accumulator registers with ~44 v_mov per tile)
#define TIME_UP() ((long long)__builtin_readcyclecounter() - t0 >= budget)
    // wave-uniform poll of an LDS counter (all lanes read the same word; readfirstlane keeps the loop's control flow scalar)
#define LDS_LOAD(p) ((unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)))
#define MFMA_TILE()                                                                                        \
    do {                                                                                                   \
        if (MODE >= 2) { /* B fragments in */                                                              \
            f16x8 f[6];                                                                                    \
            _Pragma("unroll") for (int j = 0; j < 6; ++j) f[j] = *reinterpret_cast<const f16x8*>(my + j * 1024); \
            _Pragma("unroll") for (int j = 0; j < 6; ++j) asm volatile("" : "+v"(f[j]));                  \
            h0 = f[0]; h1 = f[1];                                                                          \
        }                                                                                                  \
        MFMA6(c[0], c[1], c[2], c[3], c[4], c[5]);                                                         \
        MFMA6(c[0], c[1], c[2], c[3], c[4], c[5]);                                                         \
        MFMA6(c[0], c[1], c[2], c[3], c[4], c[5]);                                                         \
        if (MODE >= 2) { /* accumulators out */                                                            \
            _Pragma("unroll") for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x4*>(my + 2048 + j * 1024) = c[j]; \
        }                                                                                                  \
        ++nm;                                                                                              \
    } while (0)
#define VALU_TILE()                                                                                        \
    do {                                                                                                   \
        if (MODE >= 2) { /* accumulators in */                                                             \
            _Pragma("unroll") for (int j = 0; j < 6; ++j) {                                                \
                f32x4 v = *reinterpret_cast<const f32x4*>(my + 2048 + j * 1024);                           \
                asm volatile("" : "+v"(v));                                                                \
                a[(2 * j) & 15] += v[0]; a[(2 * j + 1) & 15] += v[3];                                      \
            }                                                                                              \
        }                                                                                                  \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) UNIT(a[(4 * u) & 15], a[(4 * u + 1) & 15], a[(4 * u + 2) & 15], a[(4 * u + 3) & 15]); \
        if (MODE >= 2) { /* fragments out */                                                               \
            _Pragma("unroll") for (int j = 0; j < 6; ++j)                                                  \
                *reinterpret_cast<f32x4*>(my + j * 1024) = (f32x4){a[(2 * j) & 15], a[(2 * j + 1) & 15], a[(2 * j + 2) & 15], a[(2 * j + 3) & 15]}; \
        }                                                                                                  \
        ++nv;                                                                                              \
    } while (0)
    if (MODE == 0) {
        do {
#pragma unroll 1
            for (int rep = 0; rep < 4; ++rep) { MFMA_TILE(); VALU_TILE(); }
        } while (!TIME_UP());
    } else if (is_mfma) {
        bool stop = false;
        do {
#pragma unroll 1
            for (int rep = 0; rep < 4 && !stop; ++rep) {
                if (MODE == 3) {  // flow control: at most D tiles ahead of the consumers
                    while ((unsigned)nm >= LDS_LOAD(&v_done[simd]) + D) {
                        __builtin_amdgcn_s_sleep(1);
                        if (TIME_UP()) { stop = true; break; }
                    }
                    if (stop) break;
                }
                MFMA_TILE();
                if (MODE == 3) {
                    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the accumulators are in LDS
                    if (lane == 0) __hip_atomic_store(&m_done[simd], (unsigned)nm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        } while (!stop && !TIME_UP());
    } else {
        bool stop = false;
        do {
#pragma unroll 1
            for (int rep = 0; rep < 4 && !stop; ++rep) {
                if (MODE == 3) {  // this wave's next tile: index nv * nvw + (rank - 1)
                    const unsigned idx = (unsigned)nv * nvw + (rank - 1);
                    while (LDS_LOAD(&m_done[simd]) <= idx) {
                        __builtin_amdgcn_s_sleep(1);
                        if (TIME_UP()) { stop = true; break; }
                    }
                    if (stop) break;
                }
                VALU_TILE();
                if (MODE == 3) {
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    if (lane == 0) __hip_atomic_fetch_add(&v_done[simd], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        } while (!stop && !TIME_UP());
    }
    float s = 0;
    for (int j = 0; j < 16; ++j) s += a[j];
    for (int j = 0; j < 6; ++j) s += c[j][0] + c[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && lane == 0) {
        atomicAdd(&res->mfma_tiles, nm);
        atomicAdd(&res->valu_tiles, nv);
        atomicAdd(&res->simd_rank_hist[simd * 4 + (rank < 3 ? rank : 3)], 1u);
    }
}

template <int MODE>
void run(const char* name, int waves_per_simd, long long budget) {
    static float* out = nullptr; static Res* res = nullptr;
    if (!out) { hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&res, sizeof(Res)); }
    k<MODE><<<256, 256 * waves_per_simd>>>(out, 20000, res);
    hipDeviceSynchronize();
    hipMemset(res, 0, sizeof(Res));
    k<MODE><<<256, 256 * waves_per_simd>>>(out, budget, res);
    hipDeviceSynchronize();
    Res r; hipMemcpy(&r, res, sizeof r, hipMemcpyDeviceToHost);
    const double per_simd_m = (double)r.mfma_tiles / 4, per_simd_v = (double)r.valu_tiles / 4;
    const double pipe = MODE == 0 ? per_simd_m : (per_simd_m < per_simd_v ? per_simd_m : per_simd_v);
    printf("%-10s waves/SIMD=%d : MFMA-tiles/SIMD %8.1f (%.0f cyc each)  VALU-tiles/SIMD %8.1f (%.0f cyc each)  => %.0f cycles per layer-tile per SIMD   [simd x rank:",
           name, waves_per_simd, per_simd_m, budget / per_simd_m, per_simd_v, budget / per_simd_v, budget / pipe);
    for (int i = 0; i < 16; ++i) printf("%s%u", i % 4 ? "," : " ", r.simd_rank_hist[i]);
    printf("]\n");
}

int main(int argc, char** argv) {
    const long long budget = argc > 1 ? atoll(argv[1]) : 2000000;
    for (int w : {1, 2, 3, 4}) run<0>("mono", w, budget);
    for (int w : {2, 3, 4}) run<1>("spec", w, budget);
    for (int w : {2, 3, 4}) run<2>("spec+lds", w, budget);
    for (int w : {2, 3, 4}) run<3>("pipeline", w, budget);
    return 0;
}
