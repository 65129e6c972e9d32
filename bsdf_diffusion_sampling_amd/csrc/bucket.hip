// bucket.hip — stable bucketing of a material-tagged wavefront (BASELINE.json config 4: "all paper
// measured BSDFs, mixed queries ... bucketed by id").
//
// A Mitsuba scene with several `mybsdf` instances dispatches every wavefront lane to its instance; the
// batched equivalent is to sort the lanes by material id, run one segmented launch per kernel signature
// (bsdfd_plugin_*_multi) and scatter the results back.  The sort is a STABLE counting sort — stability
// keeps the Philox counter of a lane (its row in the bucketed order) independent of how the sort is
// implemented — over <= 64 materials, three small kernels, all HBM-streaming:
//   count   : per 4096-row block, per-thread private columns of an LDS histogram (no atomics)
//   scan    : per material, exclusive scan of its per-block totals; bucket sizes
//   scatter : each block recounts, sorts its 4096 rows locally in LDS and writes every
//             (material, block) run of perm with coalesced stores.  `perm` gathers rows into bucket order (torch.argsort
//             semantics: bucketed[k] = rows[perm[k]]).
// 16 Mi ids: 128 MB read twice + 128 MB written.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "bsdfd.h"
#include "common.h"

namespace {

constexpr int BK_THREADS = 256;
constexpr int BK_ROWS = 16;                        // consecutive rows per thread
constexpr int BK_CHUNK = BK_THREADS * BK_ROWS;      // rows per block
constexpr int BK_MAX_MATERIALS = 64;
// row strides of the per-material LDS tables, padded so that the per-material serial scans (lane = material,
// same column) fall into different banks: 260 B -> bank (m + t/4) % 64, 258 u16 -> bank (m + t/2) % 64
constexpr int BK_CNT_STRIDE = BK_THREADS + 4;
constexpr int BK_BASE_STRIDE = BK_THREADS + 2;

// Stage the block's ids into LDS with coalesced loads (u8; 255 = out of range), then count:
// cnt[m][t] = number of rows with id m among thread t's 16 consecutive rows (each thread owns its column).
__device__ __forceinline__ void stage_and_count(const long long* __restrict__ ids, long long n, long long row0, int M,
                                                unsigned char* lid, unsigned char* cnt) {
    for (int i = threadIdx.x; i < M * BK_CNT_STRIDE / 4; i += BK_THREADS) reinterpret_cast<unsigned*>(cnt)[i] = 0u;
#pragma unroll
    for (int k = 0; k < BK_ROWS; ++k) {
        const int j = k * BK_THREADS + threadIdx.x;
        const long long r = row0 + j;
        long long m = r < n ? ids[r] : -1;
        lid[j] = (m >= 0 && m < M) ? (unsigned char)m : (unsigned char)255;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BK_ROWS; ++k) {
        const unsigned char m = lid[threadIdx.x * BK_ROWS + k];
        if (m != 255) cnt[m * BK_CNT_STRIDE + threadIdx.x]++;
    }
    __syncthreads();
}

__global__ __launch_bounds__(BK_THREADS) void bucket_count_kernel(const long long* __restrict__ ids, long long n, int M,
                                                                  long long nblocks, int* __restrict__ blockhist) {
    extern __shared__ unsigned char smem[];
    unsigned char* lid = smem;
    unsigned char* cnt = smem + BK_CHUNK;
    stage_and_count(ids, n, (long long)blockIdx.x * BK_CHUNK, M, lid, cnt);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int m = wave; m < M; m += BK_THREADS / 64) {  // four u8 counters per word, each <= 16
        const unsigned v = *reinterpret_cast<const unsigned*>(cnt + m * BK_CNT_STRIDE + lane * 4);
        int total = (int)((v & 0xff) + ((v >> 8) & 0xff) + ((v >> 16) & 0xff) + (v >> 24));
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) total += __shfl_xor(total, d, 64);
        if (lane == 0) blockhist[(long long)m * nblocks + blockIdx.x] = total;
    }
}

// one block per material: exclusive scan of the material's per-block totals (row-local), bucket size
__global__ __launch_bounds__(1024) void bucket_scan_kernel(const int* __restrict__ blockhist, long long nblocks,
                                                           long long* __restrict__ offs, long long* __restrict__ counts) {
    __shared__ long long part[1024];
    const int* row = blockhist + (long long)blockIdx.x * nblocks;
    long long* out = offs + (long long)blockIdx.x * nblocks;
    const long long per = (nblocks + 1023) / 1024;
    const long long b = min((long long)threadIdx.x * per, nblocks), e = min(b + per, nblocks);
    long long s = 0;
    for (long long i = b; i < e; ++i) s += row[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan of the 1024 partial sums
        const long long v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    long long acc = part[threadIdx.x] - s;
    for (long long i = b; i < e; ++i) { out[i] = acc; acc += row[i]; }
    if (threadIdx.x == 1023) counts[blockIdx.x] = part[1023];
}

__global__ __launch_bounds__(BK_THREADS) void bucket_scatter_kernel(const long long* __restrict__ ids, long long n, int M,
                                                                    long long nblocks, const long long* __restrict__ offs,
                                                                    const long long* __restrict__ counts,
                                                                    long long* __restrict__ perm) {
    extern __shared__ unsigned char smem[];
    unsigned char* lid = smem;                                                    // [4096] staged ids
    unsigned char* cnt = lid + BK_CHUNK;                                          // [M][256]
    unsigned short* base = reinterpret_cast<unsigned short*>(cnt + M * BK_CNT_STRIDE);  // [M][256] -> local positions
    unsigned short* srow = base + M * BK_BASE_STRIDE;                             // [4096] rows in bucket order
    unsigned char* sbin = reinterpret_cast<unsigned char*>(srow + BK_CHUNK);      // [4096] their materials
    long long* gbase = reinterpret_cast<long long*>(sbin + BK_CHUNK);             // [M] first slot of (material, block)
    int* lstart = reinterpret_cast<int*>(gbase + BK_MAX_MATERIALS);               // [M+1] local start of a material
    const long long row0 = (long long)blockIdx.x * BK_CHUNK;
    stage_and_count(ids, n, row0, M, lid, cnt);
    // exclusive scan of every material's column counts over the 256 threads: one wave per material at a
    // time, a lane takes 4 adjacent columns (one LDS word) and the lanes' sums are scanned with shuffles
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int m = wave; m < M; m += BK_THREADS / 64) {
        const unsigned v = *reinterpret_cast<const unsigned*>(cnt + m * BK_CNT_STRIDE + lane * 4);
        const unsigned c0 = v & 0xff, c1 = (v >> 8) & 0xff, c2 = (v >> 16) & 0xff, c3 = v >> 24;
        const unsigned tot = c0 + c1 + c2 + c3;
        unsigned incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        const unsigned excl = incl - tot;
        unsigned short* b = base + m * BK_BASE_STRIDE + lane * 4;
        b[0] = (unsigned short)excl; b[1] = (unsigned short)(excl + c0);
        b[2] = (unsigned short)(excl + c0 + c1); b[3] = (unsigned short)(excl + c0 + c1 + c2);
        if (lane == 63) lstart[m + 1] = (int)incl;
    }
    if (wave == 0) {  // first slot of (material, block) = buckets before the material + blocks before this one
        const long long c = lane < M ? counts[lane] : 0;
        long long incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const long long o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane < M) gbase[lane] = incl - c + offs[(long long)lane * nblocks + blockIdx.x];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        lstart[0] = 0;
        for (int m = 0; m < M; ++m) lstart[m + 1] += lstart[m];
    }
    __syncthreads();
    // locally sorted order (stable: threads own consecutive rows, columns were scanned in thread order)
#pragma unroll
    for (int k = 0; k < BK_ROWS; ++k) {
        const int j = threadIdx.x * BK_ROWS + k;
        const unsigned char m = lid[j];
        if (m != 255) {
            const int pos = lstart[m] + base[m * BK_BASE_STRIDE + threadIdx.x]++;
            srow[pos] = (unsigned short)j;
            sbin[pos] = m;
        }
    }
    __syncthreads();
    // coalesced write-out: consecutive local positions of a material are consecutive slots of perm
    const int total = lstart[M];
    for (int j = threadIdx.x; j < total; j += BK_THREADS) {
        const int m = sbin[j];
        perm[gbase[m] + (j - lstart[m])] = row0 + srow[j];
    }
}

}  // namespace

// Lane-order gather / scatter around the bucketed flow (HBM-bound streaming passes; 12-B rows, so one side of each is
// uncoalesced by nature — one pass moving all arrays of a wavefront beats one torch index kernel per array).
__global__ __launch_bounds__(256) void gather_wi_kernel(const long long* __restrict__ perm, long long n,
                                                        const float* __restrict__ wi, float* __restrict__ wi_b) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long s = perm[i];
    const float a = wi[s * 3 + 0], b = wi[s * 3 + 1], c = wi[s * 3 + 2];
    wi_b[i * 3 + 0] = a; wi_b[i * 3 + 1] = b; wi_b[i * 3 + 2] = c;
}
__global__ __launch_bounds__(256) void scatter_results_kernel(const long long* __restrict__ perm, long long n,
                                                              const float* __restrict__ wo_b, const float* __restrict__ pdf_b,
                                                              const float* __restrict__ pdf2_b, float* __restrict__ wo,
                                                              float* __restrict__ pdf, float* __restrict__ pdf2) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long d = perm[i];
    if (wo_b) {
        const float a = wo_b[i * 3 + 0], b = wo_b[i * 3 + 1], c = wo_b[i * 3 + 2];
        wo[d * 3 + 0] = a; wo[d * 3 + 1] = b; wo[d * 3 + 2] = c;
    }
    if (pdf_b) pdf[d] = pdf_b[i];
    if (pdf2_b) pdf2[d] = pdf2_b[i];
}

extern "C" {

int bsdfd_gather_lanes(const int64_t* perm, int64_t n, const float* wi, float* wi_b, void* stream) {
    if (n < 0) return bsdfd_fail_(BSDFD_EINVAL, "N must be >= 0");
    if (n == 0) return BSDFD_OK;
    if (!perm || !wi || !wi_b) return bsdfd_fail_(BSDFD_EINVAL, "null pointer");
    if ((n + 255) / 256 > 0x7fffffffLL) return bsdfd_fail_(BSDFD_EINVAL, "N too large");
    hipLaunchKernelGGL(gather_wi_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const long long*>(perm), (long long)n, wi, wi_b);
    HIP_TRY(hipGetLastError());
    return BSDFD_OK;
}

int bsdfd_scatter_lanes(const int64_t* perm, int64_t n, const float* wo_b, const float* pdf_b, const float* pdf2_b, float* wo,
                        float* pdf, float* pdf2, void* stream) {
    if (n < 0) return bsdfd_fail_(BSDFD_EINVAL, "N must be >= 0");
    if (n == 0) return BSDFD_OK;
    if (!perm || (wo_b && !wo) || (pdf_b && !pdf) || (pdf2_b && !pdf2)) return bsdfd_fail_(BSDFD_EINVAL, "null pointer");
    if ((n + 255) / 256 > 0x7fffffffLL) return bsdfd_fail_(BSDFD_EINVAL, "N too large");
    hipLaunchKernelGGL(scatter_results_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const long long*>(perm), (long long)n, wo_b, pdf_b, pdf2_b, wo, pdf, pdf2);
    HIP_TRY(hipGetLastError());
    return BSDFD_OK;
}

int64_t bsdfd_bucket_workspace_bytes(int64_t n, int32_t n_materials) {
    if (n < 0 || n_materials < 1) return 0;
    const long long nblocks = (n + BK_CHUNK - 1) / BK_CHUNK;
    const long long entries = (long long)n_materials * (nblocks > 0 ? nblocks : 1);
    return entries * (long long)(sizeof(int) + sizeof(long long)) + 64;
}

int bsdfd_bucket_by_material(const int64_t* material_id, int64_t n, int32_t n_materials, int64_t* perm, int64_t* counts,
                             void* workspace, int64_t workspace_bytes, void* stream) {
    if (n < 0) return bsdfd_fail_(BSDFD_EINVAL, "N must be >= 0");
    if (n_materials < 1 || n_materials > BK_MAX_MATERIALS)
        return bsdfd_fail_(BSDFD_EINVAL, "n_materials must be in [1, 64]");
    if (!counts) return bsdfd_fail_(BSDFD_EINVAL, "null counts pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n == 0) {
        HIP_TRY(hipMemsetAsync(counts, 0, sizeof(int64_t) * n_materials, st));
        return BSDFD_OK;
    }
    if (!material_id || !perm || !workspace) return bsdfd_fail_(BSDFD_EINVAL, "null pointer");
    if (workspace_bytes < bsdfd_bucket_workspace_bytes(n, n_materials))
        return bsdfd_fail_(BSDFD_EINVAL, "workspace smaller than bsdfd_bucket_workspace_bytes()");
    const long long nblocks = (n + BK_CHUNK - 1) / BK_CHUNK;
    if (nblocks > 0x7fffffffLL) return bsdfd_fail_(BSDFD_EINVAL, "N too large");
    const long long entries = (long long)n_materials * nblocks;
    // workspace: [offs: entries x i64][blockhist: entries x i32]
    long long* offs = static_cast<long long*>(workspace);
    int* blockhist = reinterpret_cast<int*>(offs + entries);
    const long long* ids = reinterpret_cast<const long long*>(material_id);
    const size_t lds_count = (size_t)BK_CHUNK + (size_t)n_materials * BK_CNT_STRIDE;
    const size_t lds_scatter = (size_t)BK_CHUNK + (size_t)n_materials * (BK_CNT_STRIDE + 2 * BK_BASE_STRIDE) + (size_t)BK_CHUNK * 3 +
                               BK_MAX_MATERIALS * sizeof(long long) + (BK_MAX_MATERIALS + 1) * sizeof(int) + 16;
    // up to ~67 KB of dynamic LDS at 64 materials: above the default cap, needs the attribute (once per process)
    static const hipError_t attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_scatter_kernel),
                                                          hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    HIP_TRY(attr_rc);
    hipLaunchKernelGGL(bucket_count_kernel, dim3((unsigned)nblocks), dim3(BK_THREADS), lds_count, st, ids, (long long)n,
                       (int)n_materials, nblocks, blockhist);
    hipLaunchKernelGGL(bucket_scan_kernel, dim3((unsigned)n_materials), dim3(1024), 0, st, blockhist, nblocks, offs,
                       reinterpret_cast<long long*>(counts));
    hipLaunchKernelGGL(bucket_scatter_kernel, dim3((unsigned)nblocks), dim3(BK_THREADS), lds_scatter, st, ids,
                       (long long)n, (int)n_materials, nblocks, offs, reinterpret_cast<const long long*>(counts),
                       reinterpret_cast<long long*>(perm));
    HIP_TRY(hipGetLastError());
    return BSDFD_OK;
}

}  // extern "C"
