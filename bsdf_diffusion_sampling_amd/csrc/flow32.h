// flow32.h — what bsdfd.hip (the host side of the library) needs from flow32.hip (the 32-query-tile flow kernels).
#pragma once
#include <vector>

#include "bsdfd.h"

// true for the two nets the reference's plugins load — disk 25-32x3-2 (rendering/utils/model.py:479-501) and spherical
// 26-32x4-2 (:422-446) — in precision split3: the shapes the 32-query-tile kernels are instantiated for
bool bsdfd_tile32_supported(const bsdfd_desc& d, int prec);
// the weight image of those kernels (fragment order of v_mfma_f32_32x32x16_f16; compile-time offsets, see L32 in flow32.hip)
std::vector<char> bsdfd_build_image32(const bsdfd_desc& d);
// mode 0: no Jacobian (flow_samples_only), 1: Jacobian (network_sampling / network_pdf / plugin sample / plugin pdf), 2: fused
// sample+pdf; nullptr = no such kernel
const void* bsdfd_kernel32(int domain, int mode);
// dynamic LDS of those kernels: the image + per-wave scratch
int bsdfd_kernel32_lds_bytes(int domain, int mode);
