// Column-sliced full refresh (spx_ftx.hip): parameter block, buffer sizes and launch wrappers. Device pointers only.
//
// The big-batch full refresh (NnueState::evaluateOnce for >= kFtxMinPositions positions) as a pipeline of small kernels
// around one gather kernel - see spx_ftx.hip for the design and the measurements that led to it.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "spx_arch.h"
#include "spx_kernels.h"

namespace spx {

// ---- sliced row table: [8 slices][kFtxRows][128 B] ----
// slice x of row r = the 128 columns {64 x .. 64 x + 63} U {512 + 64 x ..} of the row, as 8 chunks of 16 bytes; byte m of
// chunk t is column 64 x + 8 t + 2 (m >> 2) + (m & 1) (+ 512 if m & 2): a lane of the gather that ends up with D register
// pairs (0, 1) / (2, 3) of chunk t holds columns (c, c + 1) and their pairwise partners (c + 512, c + 513).
// rows: [0, 64368) threat / pawn-pair rows (i8 as in the net file); [64368, +11264) piece-square rows, LOW-byte plane
// l = int8(w) (the row itself when it fits i8); [75632, +11264) piece-square rows, HIGH-byte plane h = int8((w - l) >> 8)
// (all zero for a row that fits i8); 86896: an all-zero row (list padding).
constexpr uint32_t kFtxPsqLoBase = kThreatRows;
constexpr uint32_t kFtxPsqHiBase = kThreatRows + kPsqRows;
constexpr uint32_t kFtxZeroRow = kThreatRows + 2 * kPsqRows;
constexpr uint32_t kFtxRows = kFtxZeroRow + 1;
constexpr uint32_t kFtxSliceStride = kFtxRows * 128u;
constexpr size_t kFtxTableBytes = size_t(8) * kFtxSliceStride;
constexpr uint32_t kFtxSlabRows = 704;  // piece-square rows of one king bucket: 88 KiB per slice, LDS resident in the gather

// ---- per-perspective lists written by the extraction pass: [perspective][kFtxListStride] words ----
// [0, 32) piece-square rows as slab offsets ((row - 704 bucket) * 128); [32, 64) high-byte planes of the wide ones and
// [64, 320) threat / pawn-pair rows as slice offsets (row index * 128). Words behind a section's count are undefined.
// heads[perspective] = {nHi | nPsq << 8 | nThr << 16, 2 * position + (0 = side-to-move half, 1 = other half)}
constexpr uint32_t kFtxListStride = 320, kFtxListPsq = 0, kFtxListHi = 32, kFtxListThr = 64;
// sort key of a perspective: king bucket * 80 + (row quartets - 1): groups of 8 neighbours in this order share a bucket
// (one LDS slab) and have almost equal list lengths (one wave walks the 8 lists in lockstep)
constexpr uint32_t kFtxQuartetBins = 80, kFtxBins = 16 * kFtxQuartetBins;

// ---- sorted[position in the sorted order] = {head word 0, output slot (~0 = hole), list offset in bytes, -} ----
// the gather's wave reads the 8 entries of its group and then the 8 lists themselves, a stage of 8 steps at a time

// ---- plan: [0, 33) first segment of CU slot c ([32] = number of segments); [33] number of groups; from word 64:
// {bucket, first group, end group} per segment ----
#ifndef SPX_FTX_GROUP_COST
#define SPX_FTX_GROUP_COST 1  // (A/B over 0, 1, 3, 6, 10: 1 is best by 1 %; many of a group's steps are cheap LDS steps)
#endif
constexpr uint32_t kFtxGroupCost = SPX_FTX_GROUP_COST;  // plan: a group costs its steps + this
#ifndef SPX_FTX_SEGMENT_COST
#define SPX_FTX_SEGMENT_COST 256  // (A/B over 0 .. 1 500: flat optimum between 200 and 440, +2 % stream-ordered, +3.5 % pipelined over 0)
#endif
constexpr uint32_t kFtxSegmentCost = SPX_FTX_SEGMENT_COST;                  // plan: a slab reload inside a CU slot's range, in steps (a group costs its steps + 3)
constexpr uint32_t kFtxPlanTimes = 64 + 3 * 64;            // per workgroup of the last gather: start / end timestamps (diagnostics)
constexpr uint32_t kFtxPlanWords = kFtxPlanTimes + 4 * 256;

constexpr size_t kFtxMinPositions = 16384;    // smaller full refreshes keep the one-kernel path (the paths cross at ~14 Ki: profiles/r04_sliced_pipeline_crossover.txt)
constexpr size_t kFtxMinPositionsPipelined = 12288;  // ... of spx_eval_full_device_async (the preparation runs beside the other lane's gather)
constexpr size_t kFtxMaxPositions = 65536;    // positions per pass (scratch: ~2.6 KB each); larger batches walk in passes

struct FtxParams {
    const void* positions;   // spx_packed_pos[nPositions]
    uint32_t nPositions;
    FtTables t;              // lut, deltaTab (pseudo-attack sets), ftBias
    const uint8_t* rowS;     // the sliced row table
    uint32_t* lists;         // [2 n][kFtxListStride]
    uint32_t* heads;         // [2 n][2]
    uint32_t* keys;          // [2 n] sort keys
    uint32_t* ranks;         // [2 n] rank inside the key's bin
    uint32_t* hist;          // [kFtxBins] counts per key; zero on entry of the rank kernel, zeroed again by the plan kernel
    uint32_t* binStart;      // [kFtxBins + 17] first sorted position of each bin; then bucketStart[17]
    uint32_t* sorted;        // [2 n + 128][4]
    uint32_t* plan;          // [kFtxPlanWords]
    uint8_t* ftOut;          // [n][1024] activations (side-to-move half first)
};

inline size_t ftxListBytes(size_t n) { return 2 * n * size_t(kFtxListStride) * 4; }

hipError_t launchFtxBuildTable(const uint8_t* thrU8, const int16_t* psqW, const uint32_t* lut, uint8_t* rowS, hipStream_t stream);
// everything before the gather (extract, rank, plan, scatter): may overlap another batch's gather
hipError_t launchFtxPrepare(const FtxParams& p, hipStream_t stream);
hipError_t launchFtxGather(const FtxParams& p, hipStream_t stream);
hipError_t launchFtxSortAndPlan(const FtxParams& p, hipStream_t stream);

}  // namespace spx
