// Gather-ceiling probe (spx_probe.hip): parameter block and launch wrappers. Measurement infrastructure only.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "spx_kernels.h"

namespace spx {

struct ProbeParams {
    const void* positions;  // spx_packed_pos[nPositions]
    uint32_t nPositions;
    const uint32_t* order;  // king-bucket-sorted perspective ids (the product path's order), or nullptr
    FtTables t;
    uint32_t* lists;        // [2 * nPositions][328] row-offset lists, indexed by position in `order`
    uint8_t* sink;          // [2 * nPositions][512] what the loads xor to (keeps them alive; equal across variants)
};

struct ProbeVariant {
    const char* name;
    int path;          // 0 = global_load_dwordx4 into VGPRs, 1 = LDS-DMA (global_load_lds_dwordx4) + ds_read_b128
    int ringKiB;       // LDS ring per wave (path 1)
    int wavesPerSimd;  // launch bound
};

constexpr uint32_t kProbeListWords = 328;

int probeVariantCount();
const ProbeVariant& probeVariant(int i);
hipError_t launchProbeLists(const ProbeParams& p, uint32_t gridBlocks, hipStream_t stream);
hipError_t launchProbeGather(const ProbeParams& p, int variant, uint32_t gridBlocks, hipStream_t stream);

}  // namespace spx
