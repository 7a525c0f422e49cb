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
// rows: [0, 64368) threat / pawn-pair rows (i8 as in the net file); [64369, +11264) piece-square rows, LOW-byte plane
// l = int8(w) (the row itself when it fits i8); [75633, +11264) piece-square rows, HIGH-byte plane h = int8((w - l) >> 8)
// (all zero for a row that fits i8); an all-zero row (list padding) behind the threat rows and one at the very end.
// hiMask[piece-square row] (round 6): bit x = the row's high-byte plane has a non-zero byte in slice x. A heavy-tailed net's wide rows
// mostly have a handful of weights outside i8 (the `realistic` preset: 4 282 of its 6 553 wide rows have <= 8), so a row's plane is
// all zero in most slices: the gather of XCD x drops those rows from its walk (spx_ftx_gather_kernel: the high-byte stage is
// compacted per slice; 108 -> 92 row loads per position on that net, tools/sim_hi_slices.py).
// (round 6: a second all-zero row right behind the threat rows, at an index that fits 16 bits - the padding of the COLD sections, whose
// walk entries are 16-bit row indices)
constexpr uint32_t kFtxColdZeroRow = kThreatRows;
constexpr uint32_t kFtxPsqLoBase = kThreatRows + 1;
constexpr uint32_t kFtxPsqHiBase = kFtxPsqLoBase + kPsqRows;
constexpr uint32_t kFtxZeroRow = kFtxPsqHiBase + kPsqRows;
constexpr uint32_t kFtxRows = kFtxZeroRow + 1;
static_assert(kFtxColdZeroRow < 65536, "cold walk entries are 16-bit row indices");
constexpr uint32_t kFtxSliceStride = kFtxRows * 128u;
constexpr size_t kFtxTableBytes = size_t(8) * kFtxSliceStride;
constexpr uint32_t kFtxSlabRows = 704;  // piece-square rows of one king bucket: 88 KiB per slice, LDS resident in the gather
// ---- the gather's LDS: [slab: 704 rows + an all-zero row][hot rows: the context's most popular threat / pawn-pair rows][ring] ----
// Round 5: a wave load holds the CU's texture path ~17 cycles whatever its width or exec mask
// (tools/probes/tcp_mask_probe.hip), an LDS read of the same 1 KiB 8.8 - so every row that is LDS resident leaves the binding unit.
// The hot set comes from DATA (a histogram over the first big batch, or spx_ctx_calibrate): on the bench distribution the 256 most
// popular of the 64 368 rows serve 35 % of the threat / pawn-pair fetches, 320 serve 40 % (tools/sim_gather_steps.py). Results do not
// depend on the set: a row is added from wherever it lives.
constexpr uint32_t kFtxSlabBytes = (kFtxSlabRows + 1) * 128;
constexpr uint32_t kFtxHotRowsMax = 384;       // capacity of the tables; what a context uses: option ftx_hot_rows
constexpr uint32_t kFtxHotRowsDefault = 256;   // slab 88.1 + hot 32 + ring 16 = 136.1 KiB: one 14 / 16 KiB co-runner workgroup fits beside it
constexpr uint32_t kFtxHotHashWords = 1024;    // the extraction's LDS copy of the set: 256 buckets x 4 entries
constexpr uint32_t kFtxRingBytesPerWave = 1024 + 64;  // one stage (16 steps x 8 perspectives x 4 entries of 16 bits; high-byte planes: 8 steps of 32-bit entries) + the group's head

// ---- per-perspective lists written by the extraction pass: [perspective][kFtxListStride] words ----
// [0, 288) the LDS section: the piece-square rows ((row - 704 bucket) * 128, into the slab), then the HOT threat / pawn-pair rows
// (kFtxSlabBytes + slot * 128) - byte offsets into the gather's LDS; [288, 320) high-byte planes of the wide piece-square rows and
// [320, 576) the COLD threat / pawn-pair rows as slice offsets (row index * 128). Words behind a section's count are undefined.
// heads[perspective] = {nHi | nLds << 6 | nCold << 15, 2 * position + (0 = side-to-move half, 1 = other half), sort key, -}
constexpr uint32_t kFtxListStride = 576, kFtxListLds = 0, kFtxListHi = 288, kFtxListCold = 320;
// sort key of a perspective: king bucket * 80 + min(global quartets >> coldShift, 15) * 5 + min(LDS quartets >> 2, 4) - the quartets
// fetched through the texture path (high planes + cold rows) first: they cost twice an LDS step, and a group of 8 neighbours
// walks as many of them as its longest list has. With the round-4 key (total quartets) the hot / cold split of the lists pads
// away half of what the hot rows save: 65.5 instead of 52.0 wave loads per position at 320 hot rows (tools/sim_gather_steps.py).
constexpr uint32_t kFtxQuartetBins = 80, kFtxBins = 16 * kFtxQuartetBins, kFtxLdsClasses = 5;
__host__ __device__ inline uint32_t ftxSortKey(uint32_t bucket, uint32_t globalQ, uint32_t ldsQ, uint32_t coldShift) {
    const uint32_t gc = globalQ >> coldShift, lc = ldsQ >> 2;
    return bucket * kFtxQuartetBins + (gc < kFtxQuartetBins / kFtxLdsClasses - 1 ? gc : kFtxQuartetBins / kFtxLdsClasses - 1) * kFtxLdsClasses +
           (lc < kFtxLdsClasses - 1 ? lc : kFtxLdsClasses - 1);
}
// cost of a group whose last member lies in bin `kk` of its bucket, in LDS steps (a global step counts 2)
__host__ __device__ inline uint32_t ftxBinCost(uint32_t kk, uint32_t coldShift) {
    const uint32_t cq = kk / kFtxLdsClasses, lc = kk % kFtxLdsClasses;
    return 2u * (cq << coldShift) + coldShift + 4u * lc + 2u;
}

// ---- sorted[position in the sorted order] = {head word 0, output slot (~0 = hole), list offset in bytes, -} ----
// ---- the packed walk of a group of 8 neighbours of that order (spx_ftx_pack_kernel; what the gather reads) ----
// A group's walk has three SECTIONS - high-byte planes (global), the LDS section, cold rows (global) -, each as long as the
// longest of the 8 lists there (in quartets of rows = steps), cut into STAGES of 1 KiB. groupHead[G] = 16 words: {hiQ | ldsQ << 8
// | coldQ << 16, output slots of the 8 perspectives (~0 = hole), then what the group costs (spx_debug_ftx_walk sums these): stages, steps
// of the cold and of the LDS section as walked, rows through the texture path (high planes + cold), rows from LDS, -}; words 10 / 12 count
// the COLD section only, 14 / 15 the high-byte section as packed (steps, rows): what an XCD walks of it after dropping the planes that are
// zero in its slice is smaller (spx_debug_ftx_walk recounts it on the host). stages[G][q], stage q in the order of the sections:
//   * high-byte section: 8 steps a stage, 256 words: word 32 k + 4 e + pr = the plane (slice byte offset | the row's hiMask << 24) that
//     row kb of step k adds to perspective 2 pr + u, e = 2 kb + u; behind a list's end the end-of-table zero row (mask 0);
//   * LDS and cold sections (round 6): SIXTEEN steps a stage, 512 halfwords: halfword ((k >> 1) * 8 + e) * 8 + (k & 1) * 4 + pr = the
//     row index from the section's base (the gather's LDS: slab row or 705 + hot slot; the slice: threat row) - the 16 bytes lane e
//     of the gather reads for a PAIR of steps -, behind a list's end the section's zero row (704 / kFtxColdZeroRow). A stage with an
//     odd number of steps holds them in steps 1 .. n behind one step of padding (the walk enters it through its second half).
// Every XCD's gather walks every group: packing once what round 4 made each of the eight find out for itself (section boundaries
// per lane, list gathers, padding) took 40 % of the gather's instructions off it (profiles/r05_gather_anatomy.txt); the 16-bit
// entries halve the gather's entry reads and stage loads and this kernel's stores.
constexpr uint32_t kFtxMaxStages = 1 + 5 + 4;  // <= 32 high planes (8 steps a stage), <= 32 + 256 LDS rows, <= 256 cold rows (16 steps a stage)
constexpr uint32_t kFtxGroupHeadWords = 16;

// ---- plan: [0, 33) first segment of CU slot c ([32] = number of segments); [33] number of groups; from word 64:
// {bucket, first group, end group} per segment ----
#ifndef SPX_FTX_GROUP_COST
#define SPX_FTX_GROUP_COST 6  // in LDS steps (A/B over 2, 6, 12 on the packed walk: 6 is best by 1.5-2 %; round 4, in its unit: 1 of 0 .. 10)
#endif
constexpr uint32_t kFtxGroupCost = SPX_FTX_GROUP_COST;  // plan: a group costs its steps + this
#ifndef SPX_FTX_SEGMENT_COST
#define SPX_FTX_SEGMENT_COST 512  // in LDS steps (round 4, in steps: A/B over 0 .. 1 500, flat optimum between 200 and 440, +2 % stream-ordered, +3.5 % pipelined over 0)
#endif
constexpr uint32_t kFtxSegmentCost = SPX_FTX_SEGMENT_COST;                  // plan: a slab reload inside a CU slot's range, in steps (a group costs its steps + 3)
constexpr uint32_t kFtxPlanTimes = 64 + 3 * 64;            // per workgroup of the last gather: start / end timestamps (diagnostics)
constexpr uint32_t kFtxPlanWords = kFtxPlanTimes + 4 * 256;

// smaller full refreshes keep the one-kernel path. Round 6 (profiles/r06_sliced_pipeline_small_batches.txt; rounds 4-5: 16 384 / 12 288):
// stream-ordered calls cross at ~9.5 Ki positions (8 704 / 9 216: equal, 10 240: +5 %), pipelined calls - the preparation runs beside
// the other lanes' gathers - gain from 6 Ki on (6 144: 1.04 against 0.87e8 evals/s, 8 192: 1.30 against 0.96, 10 240: 1.58 against 1.04)
constexpr size_t kFtxMinPositions = 10240;
constexpr size_t kFtxMinPositionsPipelined = 6144;
constexpr size_t kFtxMaxPositions = 65536;    // positions per pass (scratch: ~2.6 KB each); larger batches walk in passes

struct FtxParams {
    const void* positions;   // spx_packed_pos[nPositions]
    uint32_t nPositions;
    FtTables t;              // lut, deltaTab (pseudo-attack sets), ftBias
    const uint8_t* rowS;     // the sliced row table
    uint32_t* lists;         // [2 n][kFtxListStride]
    uint32_t* heads;         // [2 n][4]
    uint32_t* ranks;         // [2 n] rank inside the key's bin
    uint32_t* hist;          // [kFtxBins] counts per key; zero on entry of the rank kernel, zeroed again by the plan kernel
    uint32_t* binStart;      // [kFtxBins + 17 + 8] first sorted position of each bin; then bucketStart[17]; then the output buckets' starts in posOrder
    uint32_t* sorted;        // [2 n + 128][4]
    uint32_t* plan;          // [kFtxPlanWords]
    uint32_t* groupHead;     // [(2 n + 128) / 8][kFtxGroupHeadWords]
    uint32_t* stages;        // [(2 n + 128) / 8][kFtxMaxStages][256]
    uint8_t* ftOut;          // [n][1024] activations (side-to-move half first)
    const uint32_t* hotHash; // [kFtxHotHashWords] the hot set as a hash: bucket ((row * hotHashMul) >> 16) & 255, four entries
                             // row | slot << 16 (0xFFFFFFFF = empty); a row that is in no entry is cold
    uint32_t hotHashMul;
    const uint8_t* hotS;     // [8 slices][hotRows][128 B] the hot rows' slices, in slot order
    uint32_t hotRows;        // rows of the hot set (0: none - every row is fetched through the texture path)
    uint32_t coldShift;      // sort key: global quartets >> this (1 for nets / sets with long cold sections)
    const uint8_t* hiMask;   // [kPsqRows] slices in which a piece-square row's high-byte plane is not all zero
    // The MLP's output-bucket order of a ONE-PASS batch, folded into this pipeline (round 6: two sort launches less per step): the
    // extraction leaves a position's bucket in its heads' fourth word, the rank kernel ranks the positions inside their bucket, the
    // plan kernel turns the counts into starts (and into the counts spx_mlp_kernel maps its tiles from), the scatter kernel writes
    // the order. posOrder == nullptr: not wanted (multi-pass calls sort over the whole call, spx_sort_*)
    uint32_t* posOrder;      // [n] out: position ids grouped by output bucket
    uint32_t* outCounts;     // [8] zero on entry; zeroed again by the plan kernel
    uint32_t* mlpHist;       // [kHistOut + 8] out: the counts at [kHistOut + bucket]
};

inline size_t ftxListBytes(size_t n) { return 2 * n * size_t(kFtxListStride) * 4; }
inline size_t ftxGroups(size_t n) { return (2 * n + 128) / 8; }
inline size_t ftxStageBytes(size_t n) { return ftxGroups(n) * kFtxMaxStages * 1024; }

hipError_t launchFtxBuildTable(const uint8_t* thrU8, const int16_t* psqW, const uint32_t* lut, uint8_t* rowS, hipStream_t stream);
hipError_t launchFtxBuildHiMask(const uint8_t* rowS, uint8_t* hiMask, hipStream_t stream);  // from the finished table
// the hot set: counts[row] += fetches of threat / pawn-pair row `row` in the lists of p (extracted with hotRows = 0), stats[0] +=
// high-byte planes fetched; then the slices of a chosen set in slot order (its hash is built on the host: spx_api.cpp)
hipError_t launchFtxExtract(const FtxParams& p, hipStream_t stream);
hipError_t launchFtxHistogram(const FtxParams& p, uint32_t* counts, uint32_t* stats, hipStream_t stream);
hipError_t launchFtxBuildHot(const uint8_t* rowS, const uint32_t* hotIds, uint32_t hotRows, uint8_t* hotS, hipStream_t stream);
hipError_t prepareFtxGather(int device);  // allows the gather its dynamic LDS on this device (once; > 64 KiB needs the attribute)
// everything before the gather (extract, rank, plan, scatter): may overlap another batch's gather
hipError_t launchFtxPrepare(const FtxParams& p, hipStream_t stream);
hipError_t launchFtxGather(const FtxParams& p, hipStream_t stream);
hipError_t launchFtxSortAndPlan(const FtxParams& p, hipStream_t stream);

}  // namespace spx
