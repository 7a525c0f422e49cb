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
    // column-sliced variants
    const uint8_t* sliced;  // [8 slices][u8 rows + 1][128 B]: slice x of row r = bytes [128 x, 128 x + 128) of the row; last row zero
    uint32_t sliceStride;   // bytes per slice
    uint32_t* groupLists;   // [ceil(2 * nPositions / 8)][kProbeGroupWords] interleaved lists of 8 perspectives
    uint32_t* wideRows;     // counter: wide (i16) rows met while packing (the sliced replay covers u8 rows only)
    uint32_t nItems;        // entries of `order` (0 = 2 * nPositions); an entry ~0 is a hole (slab variants pad buckets to groups)
    const uint32_t* plan;   // slab variants: [c] first segment of CU slot c (33 entries), from word 64: {bucket, first group, end group} per segment
};

struct ProbeVariant {
    const char* name;
    int path;          // 0 = global_load_dwordx4 into VGPRs, 1 = LDS-DMA (global_load_lds_dwordx4) + ds_read_b128
    int ringKiB;       // LDS ring per wave (path 1)
    int wavesPerSimd;  // launch bound
    int order = 0;     // path 2: 0 = king-bucket sorted perspectives, 1 = positions as they come, 2 = output-bucket order
};

constexpr uint32_t kProbeListWords = 328;
constexpr uint32_t kProbeGroupWords = 32 + (288 / 4 + 2) * 8 * 4;  // header + (72 + 2) chunks x 8 perspectives x 4 entries

int probeVariantCount();
const ProbeVariant& probeVariant(int i);
hipError_t launchProbeLists(const ProbeParams& p, uint32_t gridBlocks, hipStream_t stream);
hipError_t launchProbeSliceTable(const uint8_t* table, uint8_t* sliced, uint32_t nRows, hipStream_t stream);
hipError_t launchProbePerspOrder(const uint32_t* posOrder, uint32_t* perspOrder, uint32_t nPositions, hipStream_t stream);
hipError_t launchProbePackSlabGroups(const ProbeParams& p, uint32_t zeroRowOffset, hipStream_t stream);
hipError_t launchProbePackGroups(const ProbeParams& p, uint32_t zeroRowOffset, hipStream_t stream);
hipError_t launchProbeGather(const ProbeParams& p, int variant, uint32_t gridBlocks, hipStream_t stream);

}  // namespace spx
