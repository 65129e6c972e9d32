// common.h — pieces shared by the translation units of libbsdfd.so (bsdfd.hip: the flow sampler,
// wavefront.hip: the wavefront harness kernels): the counter-based RNG and the error plumbing.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "bsdfd.h"

// Philox4x32-10 counter-based RNG (Salmon et al., SC'11); key = seed, counter = (index, stream).
__device__ __forceinline__ void philox4x32(unsigned k0, unsigned k1, unsigned c0, unsigned c1, unsigned c2,
                                           unsigned c3, unsigned out[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u01_open(unsigned x) {  // (0, 1]
    return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

// records the thread-local message behind bsdfd_last_error() and returns `code` (defined in bsdfd.hip)
int bsdfd_fail_(int code, const std::string& msg);

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e__ = (expr);                                                                    \
        if (e__ != hipSuccess)                                                                      \
            return bsdfd_fail_(BSDFD_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__));     \
    } while (0)
