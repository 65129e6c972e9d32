// flow32.h — what bsdfd.hip (the host side of the library) needs from flow32.hip (the 32-query-tile flow kernels).
#pragma once
#include <vector>

#include "bsdfd.h"

// true where a 32-query-tile kernel exists for SOME mode of this net: the reference's two plugin nets — disk 25-32x3-2
// (rendering/utils/model.py:479-501) and spherical 26-32x4-2 (:422-446) — in precision split3 (all modes), and the 64 x 6
// spherical teacher (:449-477) in precision f16 (samples-only mode)
bool bsdfd_tile32_supported(const bsdfd_desc& d, int prec);
// the weight image of those kernels (fragment order of v_mfma_f32_32x32x16_f16; compile-time offsets, see L32 / L32W in flow32.hip)
std::vector<char> bsdfd_build_image32(const bsdfd_desc& d, int prec);
// mode 0: no Jacobian (flow_samples_only), 1: Jacobian (network_sampling / network_pdf / plugin sample / plugin pdf), 2: fused
// sample+pdf; nullptr = no such kernel for this net / precision (the 16-query-tile kernel of csrc/bsdfd.hip serves the mode)
const void* bsdfd_kernel32(const bsdfd_desc& d, int prec, int mode);
// true for a kernel that serves the mode only when the caller asked for 32-query tiles EXPLICITLY (bsdfd_desc.tile = 32), not under
// the library's default: the 64 x 6 Jacobian kernel, which measured slower than its 16-query counterpart
bool bsdfd_kernel32_opt_in(const bsdfd_desc& d, int prec, int mode);
// dynamic LDS of that kernel (the image + per-wave scratch) and its workgroup size
int bsdfd_kernel32_lds_bytes(const bsdfd_desc& d, int prec, int mode);
int bsdfd_kernel32_threads(const bsdfd_desc& d, int prec, int mode);
// f32x4 records per 32-query tile of the per-query context the mode-1 kernel of this net writes / reads (bsdfd_context_bytes)
int bsdfd_tile32_context_v4(const bsdfd_desc& d, int prec);
