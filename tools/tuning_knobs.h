// Grid-shape knobs for tools/tscan.py and tools/nscan.py.  NOT part of the product: a tools build force-includes this header
//     tools/ab_build.sh tune "-include tools/tuning_knobs.h"
// and csrc/bsdfd.hip expands BSDFD_TOOLS_KNOBS(per_cu) inside its launch routine.
#pragma once
#include <cstdlib>
#define BSDFD_TOOLS_KNOBS(per_cu)                                          \
    static const int per_cu_override = [] {                                \
        const char* ov = std::getenv("BSDFD_BLOCKS_PER_CU");               \
        return ov ? std::atoi(ov) : 0;                                     \
    }();                                                                   \
    static const int cl_override = [] {                                    \
        const char* ov = std::getenv("BSDFD_CHUNK_LOG2");                  \
        return ov ? std::atoi(ov) : -1;                                    \
    }();                                                                   \
    if (per_cu_override > 0) per_cu = per_cu_override;
// $BSDFD_LDS_PAD: extra bytes of dynamic LDS per workgroup (fewer resident workgroups per CU: occupancy sweeps)
#define BSDFD_TOOLS_LDS_PAD                                                \
    static const size_t pad = [] {                                         \
        const char* ov = std::getenv("BSDFD_LDS_PAD");                     \
        return ov ? (size_t)std::atoll(ov) : (size_t)0;                    \
    }();                                                                   \
    return pad;
