// Column-sliced full refresh for gfx950 (rounds 4-5): the feature-transformer pass of NnueState::evaluateOnce for big batches.
//
// What the round-3 kernel (spx_ft_kernel: one wave per perspective, whole 1 KiB rows) left on the table, measured with the
// load-only probes of spx_probe.hip on the bench batch (profiles/r04_sliced_probe_*):
//   * all eight XCDs fetch whole rows, so the eight private 4 MiB L2s cache eight copies of the same hot rows: L2 hit rate
//     74 %, 2.2 GB per launch over the fabric, 367 us for the loads alone;
//   * giving XCD x only the 128-byte slice x of every row lifts the hit rate to 88 % and cuts the fabric bytes to a third,
//     but the loads then run into the CU's texture-address path (16 cycles per wave load): 342 us;
//   * a slice of the 704 piece-square rows of ONE king bucket is 88 KiB - it fits the CU's LDS. With the perspectives
//     ordered by king bucket and that slab in LDS, 37 % of the row fetches never touch the texture path: 269 us;
//   * and the VALU work of the old gather (12 instructions per row to zero-extend and add u8 columns, 850 of the kernel's
//     1 407 instructions per perspective) disappears onto the matrix pipe: ONE v_mfma_i32_16x16x64_i8 with a constant
//     selection matrix widens AND adds up four gathered i8 rows (tools/probes/mfma_rowsum_probe.hip).
//
// Round 5: a 16-byte-per-lane wave load holds that texture path ~17.5 cycles whatever its exec mask or
// width (tools/probes/tcp_mask_probe.hip), a ds_read_b128 of the same 1 KiB 8.8: rows must LEAVE the path, not be fetched more
// cleverly. The context's ~256 most popular threat / pawn-pair rows (chosen from data: a histogram over its first big batch, or
// spx_ctx_calibrate) sit in LDS beside the slab - 35 % of those fetches -, and the walks of the groups are PACKED once per batch
// instead of found out by each of the 8 XCDs (section boundaries, list searching, padding: 160 of the round-4 gather's 280 us were
// that skeleton, profiles/r05_gather_anatomy.txt).
//
// Pipeline (all on one stream; on the three lanes of the pipelined entry point the preparation of two batches runs beside a gather):
//   spx_ftx_extract_kernel  one wave per POSITION: board decode, attack sets and the feature candidates once, the row lists of both
//                           perspectives (nnue_state.cpp:309-354, 440-449) to HBM - LDS section (piece-square rows, hot rows),
//                           high-byte planes, cold rows -, a head and a sort key (king bucket, global quartets, LDS quartets)
//   spx_ftx_rank_kernel     counting sort, part 1: rank of every perspective inside its key's bin (and, round 6, of every position
//                           inside its OUTPUT bucket: a one-pass batch gets the MLP's order - posOrder - from this sort as well)
//   spx_ftx_plan_kernel     bin starts (each bucket padded to whole groups of 8), and the PLAN: the groups cut into 32
//                           contiguous, equally heavy ranges - one per CU of an XCD -, each a list of one-bucket segments
//   spx_ftx_scatter_kernel  counting sort, part 2: every perspective's head at its place in the sorted order
//   spx_ftx_pack_kernel     one wave per group of 8 neighbours: the group's head (section lengths, output slots) and its walk as
//                           stages of 8 steps x 8 perspectives x 4 rows (1 KiB each, padded with the zero row, sections stage-aligned)
//   spx_ftx_gather_kernel   256 workgroups of 16 waves (workgroup b on XCD b % 8 = slice b % 8, CU slot b / 8): per segment
//                           the bucket's piece-square slab slice into LDS (the hot rows' slice once), then one wave per group:
//                           stages through its own 1 KiB of LDS, steps walked in pairs as a rolling window, 2 perspectives x 4 rows
//                           x 128 B per wave load / LDS read, one MFMA each, pairwise activation (multilayer.h:92-152) from the
//                           i32 sums, outputs transposed through the stage: 8 bytes per lane.
// (Round 4 also built the INCREMENTAL path on the same tables - spx_ftu_derive_kernel -> rank / plan / scatter ->
// spx_ftu_apply_kernel -: bit-exact and slower than spx_update_kernel, 1.32 vs 2.15e8 updates+evals/s; retired in round 5 to
// experiments/r04_incremental_pipeline_column_sliced.hip.txt.)
// Round 6: the walk of a stage after a look at its ISA (walkStage), high-byte planes dropped per column slice (hiMask).
// Results are bit-identical to spx_ft_kernel (sums of rows mod 2^16; tests/test_gpu_parity.py runs both).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>

#include "spx_ft_device.h"
#ifndef SPX_CORUNNER_PRIO
#define SPX_CORUNNER_PRIO 0
#endif
#include "spx_ftx.h"

namespace spx {

namespace {

#ifndef SPX_FTX_GATHER_WAVES
#define SPX_FTX_GATHER_WAVES 16
#endif
#ifndef SPX_FTX_SKIP
#define SPX_FTX_SKIP 0  // measurement builds only (wrong sums): 1 no LDS row reads, 2 no global row loads, 4 no MFMAs, 8 no output stores, 64 LDS reads without bank conflicts,
#endif                  // 32 no ring writes (tools/build_variants.sh; profiles/r05_gather_anatomy.txt)

#ifndef SPX_FTX_NT
#define SPX_FTX_NT 0  // A/B, non-temporal (streaming) accesses: 2 the gather's stage loads (-13 %), 4 its output stores (nothing), 8 the
#endif                // extraction's list stores (nothing); (1 / 16 belonged to the pack kernel variant without LDS: docs/experiments.md 6)
#ifndef SPX_FTX_GATHER_WAVES_PER_SIMD
#define SPX_FTX_GATHER_WAVES_PER_SIMD 5  // register budget: 512 / this = 96 (a workgroup brings 4 waves per SIMD: the rest is room for two of the extraction's)
#endif
constexpr uint32_t kGatherWaves = SPX_FTX_GATHER_WAVES;
// LDS of the gather: slab + hot rows + one ring stage per wave
constexpr uint32_t gatherLdsBytes(uint32_t hotRows) { return kFtxSlabBytes + hotRows * 128u + kGatherWaves * kFtxRingBytesPerWave; }

// column of byte m of chunk t of slice x (see spx_ftx.h)
__device__ __forceinline__ uint32_t sliceColumn(uint32_t x, uint32_t t, uint32_t m) {
    return 64 * x + 8 * t + 2 * (m >> 2) + (m & 1u) + ((m & 2u) ? 512u : 0u);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// Sliced row table from the tables the context already holds: the u8 row table (threat rows: value + 128, columns
// interleaved per lane - relayoutThreatRow in spx_api.cpp) and the i16 piece-square table. One thread per 16-byte chunk.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void spx_ftx_build_table_kernel(const uint8_t* thrU8, const int16_t* psqW, const uint32_t* lut, uint8_t* rowS) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= kFtxRows * 64u) return;
    const uint32_t r = idx >> 6, x = (idx >> 3) & 7u, t = idx & 7u;
    uint8_t out[16];
#pragma unroll
    for (uint32_t m = 0; m < 16; ++m) {
        const uint32_t c = sliceColumn(x, t, m);
        int v = 0;
        if (r < kThreatRows) {
            // inverse of relayoutThreatRow: column c sits in lane l = (c & 511) >> 3 at byte k(j) (+ 8 for the upper half)
            const uint32_t l = (c & 511u) >> 3, j = c & 7u, k = (j & 4u) | ((j & 1u) << 1) | ((j & 2u) >> 1);
            v = int(thrU8[size_t(r) * kL1 + 16 * l + (c >= 512 ? 8 : 0) + k] ^ 0x80u);
        } else if (r >= kFtxPsqLoBase && r < kFtxZeroRow) {
            const uint32_t row = (r - kFtxPsqLoBase) % kPsqRows;
            const int w = psqW[size_t(row) * kL1 + c];
            const bool fits = (lut[kLutCompactBase + (row >> 5)] >> (row & 31)) & 1u;  // (then l = w, h = 0)
            const int lo = int(int8_t(uint8_t(w & 0xFF)));
            v = r < kFtxPsqHiBase ? lo : (fits ? 0 : (w - lo) >> 8);
        }
        out[m] = uint8_t(v);
    }
    *reinterpret_cast<u32x4*>(rowS + (size_t(x) * kFtxRows + r) * 128 + 16 * t) = *reinterpret_cast<const u32x4*>(out);
}

// hiMask[row] bit x = slice x of the row's high-byte plane holds a non-zero byte (spx_ftx.h). One thread per (row, slice).
__global__ void spx_ftx_build_himask_kernel(const uint8_t* rowS, uint32_t* hiMaskWords) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= kPsqRows * 8u) return;
    const uint32_t row = idx >> 3, x = idx & 7u;
    const u32x4* src = reinterpret_cast<const u32x4*>(rowS + (size_t(x) * kFtxRows + kFtxPsqHiBase + row) * 128);
    uint32_t any = 0;
#pragma unroll
    for (uint32_t t = 0; t < 8; ++t) {
        const u32x4 v = src[t];
        any |= v[0] | v[1] | v[2] | v[3];
    }
    if (any) atomicOr(&hiMaskWords[row >> 2], 1u << (8 * (row & 3u) + x));  // (byte `row` of the table, zeroed by the caller)
}

// ---------------------------------------------------------------------------------------------------------------------
// Extraction: one wave per position, lane = square. Everything that does not depend on the perspective is done once (VERDICT
// r3 item 5): the record decode, the attack sets, and the ENUMERATION of the feature candidates - every (attacker, victim)
// pair and every pawn pair becomes one item of a flat list in LDS (a wave scan of the lanes' counts places them; each lane then
// writes its own, as many rounds as the busiest square has targets). Each perspective then turns the items into rows 64 at a
// time - typically one round per kind instead of one per target of the busiest square - and writes them straight to its list.
// ---------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kItemCap = 256;  // per kind: <= 30 attackers x 8 targets; malformed records are cut off here
// threat items: attacker square | victim square << 6; pawn items: from | to << 6 | (1 = same colour (from < to), 0 = different
// colours: a feature of the perspective that owns `from`, whose mask it is - nnue_state.cpp:330-351) << 12

// Beside a gather workgroup (its 16 waves x 96 registers and ~136 KiB of LDS) a CU has room for 128 registers per SIMD lane and
// ~23 KB of LDS: one workgroup of EIGHT waves here (two per SIMD), sharing one copy of the tables - 20.9 KB. The kernel is bound by
// its chains of dependent round trips, not by the VALU (a quarter of the resident waves: 3.4 x the time, profiles/
// r05_timeline_pipelined_steps_first_version.txt), so nothing a wave waits for may come from further away than LDS: the next
// record travels while this one is taken apart, and the hot set is looked up in an LDS hash (256 buckets of four row | slot << 16
// entries, built on the host; a round-5 first version asked a 64 368-entry table in memory once per 64 rows).
constexpr uint32_t kExtractWaves = 8;

__global__ __launch_bounds__(64 * kExtractWaves, 2) void spx_ftx_extract_kernel(FtxParams p) {
#if SPX_CORUNNER_PRIO
    __builtin_amdgcn_s_setprio(SPX_CORUNNER_PRIO);  // (A/B: the kernels that run beside the gather ask for issue priority)
#endif
    __shared__ uint32_t sLut[kLutCompactBase + kLutCompactWords];  // threat LUT + the compact-row bitmap
    __shared__ uint64_t sPseudo[kDeltaPseudoWords];
    __shared__ __align__(16) uint32_t sHot[kFtxHotHashWords];
    __shared__ uint16_t sItems[kExtractWaves][2 * kItemCap];  // threat items, behind them the pawn items (bit 15)
    for (int i = threadIdx.x; i < kLutCompactBase + kLutCompactWords; i += blockDim.x) sLut[i] = p.t.lut[i];
    for (int i = threadIdx.x; i < kDeltaPseudoWords; i += blockDim.x) sPseudo[i] = p.t.deltaTab[kDeltaRayWords + i];
    for (int i = threadIdx.x; i < int(kFtxHotHashWords); i += blockDim.x) sHot[i] = p.hotRows ? p.hotHash[i] : 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t lane = laneId(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint16_t* const items = sItems[wave];
    // records through the scalar cache, one position ahead: the vector-memory counter is in order, a record asked for through it
    // could only be waited for together with all the list stores of the position before (their round trip, once per position)
    const uint32_t posStride = gridDim.x * kExtractWaves;
    uint32_t pos = blockIdx.x * kExtractWaves + wave;
    const uint8_t* const records = reinterpret_cast<const uint8_t*>(p.positions);
    u32x8 record = scalarLoadRecord(records + size_t(min(pos, p.nPositions - 1)) * 32);
    for (; pos < p.nPositions; pos += posStride) {
        scalarLoadWait(record);
        const LaneBoard b = decodeBoardScalar(record, lane);
        record = scalarLoadRecord(records + size_t(min(pos + posStride, p.nPositions - 1)) * 32);
        const int piece = b.piece;
        const bool occupied = piece != kNoPiece;
        uint32_t nThreatItems, nPawnItems;
        {
            uint64_t targets = laneTargets(b, lane);  // the occupied non-king squares this lane's piece attacks
            uint64_t partners = 0, same = 0;
            if ((b.pawnsBb >> lane) & 1) {
                const uint64_t mine = b.pawnsBb & ((piece & 1) ? b.whiteBb : ~b.whiteBb);
                same = mine & ~((2ull << lane) - 1);
                partners = (same | (b.pawnsBb & ~mine)) & ppMask(int(lane));
                same &= partners;
            }
            // every lane's items go behind those of the lanes below it: one scan for both kinds, then no lane waits for another
            const uint32_t mineCount = uint32_t(popc64(targets)) | (uint32_t(popc64(partners)) << 16);
            const uint32_t incl = waveInclusiveScan(mineCount);
            const uint32_t total = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
            nThreatItems = min(total & 0xFFFFu, kItemCap);
            nPawnItems = min(total >> 16, kItemCap);
            uint32_t atThreat = (incl - mineCount) & 0xFFFFu, atPawn = (incl - mineCount) >> 16;
            // (two plain loops, one kind each: as ONE loop over both kinds the body was four nested exec-mask branches, 40 VALU
            // instructions per iteration - a third of the kernel's instructions)
            while (targets) {
                const uint32_t to = uint32_t(ctz64(targets));
                targets &= targets - 1;
                if (atThreat < kItemCap) items[atThreat] = uint16_t(lane | (to << 6));
                ++atThreat;
            }
            uint16_t* const pawnItems = items + nThreatItems;
            while (partners) {
                const uint32_t to = uint32_t(ctz64(partners));
                partners &= partners - 1;
                if (atPawn < kItemCap) pawnItems[atPawn] = uint16_t(lane | (to << 6) | (uint32_t((same >> to) & 1) << 12) | 0x8000u);
                ++atPawn;
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (int c = 0; c < 2; ++c) {
            const uint64_t ownKing = __ballot(piece == (10 | c));
            const int kingSq = ownKing ? ctz64(ownKing) : 0;  // a record without that king is malformed: stay in bounds
            const int x = perspXor(c, kingSq);
            const int flipColour = (c == 0) ? 1 : 0;
            const uint32_t bucket = uint32_t(kingBucket(c == 0 ? (kingSq ^ 56) : kingSq));
            const uint32_t q = 2 * pos + uint32_t(c);
            uint32_t* out = p.lists + size_t(q) * kFtxListStride;
            // piece-square rows (resetPsqAccumulator, nnue_state.cpp:440-449): every one from the bucket's LDS slab; a row
            // with weights outside i8 also has a high-byte plane to fetch
            uint32_t row = 0;
            bool wide = false;
            if (occupied) {
                row = psqRow(c, piece, int(lane), kingSq);
                wide = !((sLut[kLutCompactBase + (row >> 5)] >> (row & 31)) & 1u);
            }
            const uint64_t wideMask = __ballot(wide);
            const uint32_t slot = prefixCount(b.occ), wideSlot = prefixCount(wideMask);
            if (occupied && slot < kPsqCap) {
                out[kFtxListLds + slot] = (row - bucket * kFtxSlabRows) * 128u;  // (an LDS offset: the slab heads the gather's LDS)
                if (wide && wideSlot < kPsqCap) out[kFtxListHi + wideSlot] = (kFtxPsqHiBase + row) * 128u;
            }
            const uint32_t nPsq = min(uint32_t(popc64(b.occ)), uint32_t(kPsqCap));
            const uint32_t nHi = min(uint32_t(popc64(wideMask)), uint32_t(kPsqCap));
            // threat / pawn-pair rows: a row of the context's hot set (LDS resident in the gather) joins the piece-square rows in the
            // LDS section as an LDS offset, any other goes to the cold section as a slice offset; 256 rows in all, in the order of
            // enumeration, like the reference's StaticVector<u16, 256> (nnue_state.cpp:315)
            uint32_t nThr = 0, nHot = 0, nCold = 0;
            auto emit = [&](int32_t r) {
                const uint64_t valid = __ballot(r >= 0);
                const bool taken = r >= 0 && nThr + prefixCount(valid) < kThreatCap;
                uint32_t slot = 0xFFFFu;
                if (taken) {  // the hot set's hash: bucket (row * multiplier) >> 24 mod 256, four entries row | slot << 16
                    const u32x4 e = *reinterpret_cast<const u32x4*>(sHot + 4 * ((__umul24(uint32_t(r), p.hotHashMul) >> 16) & (kFtxHotHashWords / 4 - 1)));
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if ((e[i] & 0xFFFFu) == uint32_t(r)) slot = e[i] >> 16;
                    }
                }
                const bool hot = taken && slot != 0xFFFFu, cold = taken && !hot;
                const uint64_t hotMask = __ballot(hot), coldMask = __ballot(cold);
                // ONE store for both kinds (beside a gather every vector-memory instruction of this kernel queues behind the gather's
                // row loads: without the list stores the pass took 223 instead of 272 us there, profiles/r05_gather_anatomy.txt)
                if (taken) {
                    uint32_t* at = out + (hot ? kFtxListLds + nPsq + nHot + prefixCount(hotMask) : kFtxListCold + nCold + prefixCount(coldMask));
                    const uint32_t v = hot ? kFtxSlabBytes + slot * 128u : uint32_t(r) * 128u;
                    if (SPX_FTX_NT & 8) __builtin_nontemporal_store(v, at);
                    else *at = v;
                }
                nHot += uint32_t(popc64(hotMask));
                nCold += uint32_t(popc64(coldMask));
                nThr = min(nThr + uint32_t(popc64(valid)), uint32_t(kThreatCap));
            };
            // threat rows (addThreatFeatures, nnue_state.cpp:309-328: the reference drops the pairs whose index is negative), then
            // the pawn pairs (:330-351) - ONE walk over the items of both kinds, 64 at a time (a position has ~55: one round)
            const uint32_t nItems = nThreatItems + nPawnItems;
            for (uint32_t base = 0; base < nItems; base += 64) {
                const bool active = base + lane < nItems;
                const uint32_t item = active ? items[base + lane] : 0u;
                const int from = item & 63u, to = (item >> 6) & 63u;
                const int pieceFrom = __shfl(piece, from, 64), pieceTo = __shfl(piece, to, 64);
                int32_t r = -1;
                if (active && !(item & 0x8000u)) {
                    const int pieceRel = pieceFrom ^ flipColour, victimRel = pieceTo ^ flipColour, sqRel = from ^ x;
                    const uint64_t pseudoRel = sPseudo[(pieceRel >= 2 ? (pieceRel >> 1) + 1 : pieceRel) * 64 + sqRel];
                    r = threatRow(sLut, pieceRel, sqRel, pseudoRel, victimRel, to ^ x);
                } else if (active) {
                    const bool sameColour = (item >> 12) & 1u, own = (pieceFrom & 1) == c;
                    if (sameColour || own) r = int32_t(ppRow(ppId(from ^ x, !own), ppId(to ^ x, sameColour ? !own : true)));
                }
                emit(r);
            }
            if (lane == 0) {
                u32x4 head;
                head[0] = nHi | ((nPsq + nHot) << 6) | (nCold << 15);
                head[1] = 2 * pos + ((c == b.stm) ? 0u : 1u);  // stm half first (nnue_state.cpp:396-438)
                const uint32_t globalQ = (nHi + 3) / 4 + (nCold + 3) / 4, ldsQ = (nPsq + nHot + 3) / 4;
                head[2] = ftxSortKey(bucket, globalQ, ldsQ, p.coldShift);
                head[3] = min((uint32_t(popc64(b.occ)) - 2u) / 4u, 7u);  // output bucket (MaterialCount<8>, output.h:44-55): FtxParams::posOrder
                *reinterpret_cast<u32x4*>(p.heads + 4 * size_t(q)) = head;
            }
        }
        __builtin_amdgcn_wave_barrier();  // (the next position's items overwrite these)
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Counting sort by key, part 1: block-local ranks through an LDS histogram, one global atomic per block and non-empty bin.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void spx_ftx_rank_kernel(FtxParams p) {
#if SPX_CORUNNER_PRIO
    __builtin_amdgcn_s_setprio(SPX_CORUNNER_PRIO);  // (A/B: the kernels that run beside the gather ask for issue priority)
#endif
    __shared__ uint32_t sCount[kFtxBins], sBase[kFtxBins], sOut[8], sOutBase[8];
    for (uint32_t k = threadIdx.x; k < kFtxBins; k += blockDim.x) sCount[k] = 0;
    if (threadIdx.x < 8) sOut[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x, nPersp = 2 * p.nPositions;
    uint32_t key = 0, local = 0, outKey = 0, outLocal = 0;
    const bool mine = q < nPersp, ranksOut = mine && p.posOrder && !(q & 1u);  // (a position's first perspective speaks for it)
    if (mine) key = p.heads[4 * size_t(q) + 2];
    if (mine) local = atomicAdd(&sCount[key], 1u);
    if (ranksOut) {
        outKey = p.heads[4 * size_t(q) + 3] & 7u;
        outLocal = atomicAdd(&sOut[outKey], 1u);
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < kFtxBins; k += blockDim.x) {
        if (sCount[k]) sBase[k] = atomicAdd(&p.hist[k], sCount[k]);
    }
    if (threadIdx.x < 8 && sOut[threadIdx.x]) sOutBase[threadIdx.x] = atomicAdd(&p.outCounts[threadIdx.x], sOut[threadIdx.x]);
    __syncthreads();
    if (mine) p.ranks[q] = sBase[key] + local;
    if (ranksOut) p.heads[4 * size_t(q) + 3] = outKey | ((sOutBase[outKey] + outLocal) << 3);  // (read again by the scatter kernel)
}

// ---------------------------------------------------------------------------------------------------------------------
// Bin starts and the plan. One workgroup of FOUR waves: it has to find room beside a gather workgroup when the other lane's
// gather is running (16 waves x 96 registers leave 128 registers per SIMD lane: one wave of this kernel each). As 16 waves x 39
// registers (round 4) it waited for the first gather workgroup to exit: 190 us per step
// (profiles/r05_timeline_pipelined_steps_first_version.txt).
// ---------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kPlanThreads = 256, kPlanWaves = kPlanThreads / 64;
constexpr uint32_t kPlanBinsPerLane = (kFtxQuartetBins + 63) / 64, kPlanBinsPerThread = (kFtxBins + kPlanThreads - 1) / kPlanThreads;

__global__ __launch_bounds__(kPlanThreads) void spx_ftx_plan_kernel(FtxParams p) {
#if SPX_CORUNNER_PRIO
    __builtin_amdgcn_s_setprio(SPX_CORUNNER_PRIO);  // (A/B: the kernels that run beside the gather ask for issue priority)
#endif
    __shared__ uint32_t sBin[kFtxBins + 1];    // counts, then starts
    __shared__ uint32_t sBucketStart[17], sBucketCount[16];
    __shared__ uint32_t sGroups[kFtxBins];     // per bin: groups whose longest list ends in it; then the index of the first of them
    __shared__ uint32_t sCost[kFtxBins];       // per bin: their cost; then the exclusive prefix of it
    __shared__ uint32_t sWaveSum[2 * kPlanWaves];
    __shared__ uint8_t sHead[kFtxBins];
    __shared__ uint32_t sCut[33];
    const uint32_t tid = threadIdx.x;
    for (uint32_t k = tid; k < kFtxBins; k += kPlanThreads) {
        sBin[k] = p.hist[k];
        p.hist[k] = 0;  // ready for the next batch's rank kernel
    }
    __syncthreads();
    // bucket totals and the bins' starts inside their bucket: a wave per bucket (four buckets each), kPlanBinsPerLane bins per lane
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    for (uint32_t b = wave; b < 16; b += kPlanWaves) {
        uint32_t c[kPlanBinsPerLane], sum = 0;
#pragma unroll
        for (uint32_t i = 0; i < kPlanBinsPerLane; ++i) {
            const uint32_t kk = kPlanBinsPerLane * lane + i;
            c[i] = kk < kFtxQuartetBins ? sBin[b * kFtxQuartetBins + kk] : 0u;
            sum += c[i];
        }
        const uint32_t incl = waveInclusiveScan(sum);
        uint32_t at = incl - sum;
#pragma unroll
        for (uint32_t i = 0; i < kPlanBinsPerLane; ++i) {
            const uint32_t kk = kPlanBinsPerLane * lane + i;
            if (kk < kFtxQuartetBins) sBin[b * kFtxQuartetBins + kk] = at;
            at += c[i];
        }
        if (lane == 63) sBucketCount[b] = incl;
    }
    __syncthreads();
    if (wave == 0) {
        const uint32_t padded = lane < 16 ? (sBucketCount[lane] + 7u) & ~7u : 0u;  // every bucket starts a new group of 8
        const uint32_t incl = waveInclusiveScan(padded);
        if (lane < 16) sBucketStart[lane] = incl - padded;
        if (lane == 16) sBucketStart[16] = incl;
    }
    __syncthreads();
    for (uint32_t k = tid; k < kFtxBins; k += kPlanThreads) {
        const uint32_t start = sBin[k] + sBucketStart[k / kFtxQuartetBins];
        sBin[k] = start;
        p.binStart[k] = start;
    }
    if (tid < 16 * 8) {  // holes that pad a bucket to whole groups
        const uint32_t b = tid >> 3, hole = sBucketStart[b] + sBucketCount[b] + (tid & 7u);
        if (hole < sBucketStart[b + 1]) reinterpret_cast<u32x4*>(p.sorted)[hole] = u32x4{0u, 0xFFFFFFFFu, 0u, 0u};
    }
    if (tid < 17) p.binStart[kFtxBins + tid] = sBucketStart[tid];
    if (p.posOrder && tid < 8) {  // the MLP's output buckets: counts for its tile map, starts for the scatter kernel
        uint32_t start = 0;
        for (uint32_t b2 = 0; b2 < tid; ++b2) start += p.outCounts[b2];
        const uint32_t count = p.outCounts[tid];
        p.mlpHist[kHistOut + tid] = count;
        p.binStart[kFtxBins + 17 + tid] = start;
    }
    __syncthreads();
    if (p.posOrder && tid < 8) p.outCounts[tid] = 0;  // (ready for the next batch's rank kernel; every thread above has read them)

    // Cost of a group = the steps of its last valid perspective's bin (ftxBinCost: global steps count double; bins ascend inside a
    // bucket, global quartets first - the section every member pads to the group's longest) + a constant for the group's fixed work -
    // piecewise constant over the BINS, so everything below runs over the bins, not the 16 K groups.
    // Inside bucket b (relative positions, groups of 8 from its start) the groups whose last member lies in [s, e) are
    // G' = s / 8 .. e / 8 - 1, plus the bucket's final partial group, whose last member is the bucket's last perspective.
    const uint32_t nGroups = sBucketStart[16] / 8;
    for (uint32_t k = tid; k < kFtxBins; k += kPlanThreads) {
        const uint32_t b = k / kFtxQuartetBins, kk = k % kFtxQuartetBins, base = sBucketStart[b], count = sBucketCount[b];
        const uint32_t s0 = sBin[k] - base, e0 = (kk + 1 < kFtxQuartetBins ? sBin[k + 1] : base + count) - base;
        uint32_t ng = e0 / 8 - s0 / 8;
        if (e0 > s0 && e0 == count && (count & 7u)) ++ng;  // (the partial group: the last one of the bucket's last bin)
        sGroups[k] = ng;
        // a bucket that starts INSIDE a CU slot's range costs that slot a slab reload behind a barrier - its 16 waves wait for the last
        // group of the previous bucket (tools/gpu_ftx_block_times.py: ~15 us per extra segment; kFtxSegmentCost, in steps, from an A/B). The bin that holds the
        // bucket's first group carries it; a cut that falls into it goes to the bucket's start (below)
        const bool bucketHead = ng > 0 && s0 / 8 == 0 && base > 0;
        sHead[k] = bucketHead;
        sCost[k] = ng * (ftxBinCost(kk, p.coldShift) + kFtxGroupCost) + (bucketHead ? kFtxSegmentCost : 0u);
    }
    __syncthreads();
    // exclusive prefixes of the costs and of the group counts over the bins: kPlanBinsPerThread consecutive bins per thread, wave
    // scans of the threads' sums, the waves' totals
    uint32_t gMine[kPlanBinsPerThread], wMine[kPlanBinsPerThread], gSum = 0, wSum = 0;
#pragma unroll
    for (uint32_t i = 0; i < kPlanBinsPerThread; ++i) {
        const uint32_t k = kPlanBinsPerThread * tid + i;
        gMine[i] = k < kFtxBins ? sGroups[k] : 0u;
        wMine[i] = k < kFtxBins ? sCost[k] : 0u;
        gSum += gMine[i];
        wSum += wMine[i];
    }
    const uint32_t wIncl = waveInclusiveScan(wSum), gIncl = waveInclusiveScan(gSum);
    if (lane == 63) sWaveSum[wave] = wIncl, sWaveSum[kPlanWaves + wave] = gIncl;
    __syncthreads();
    uint32_t wBase = 0, gBase = 0, total = 0;
    for (uint32_t i = 0; i < kPlanWaves; ++i) {
        const uint32_t w = sWaveSum[i], g = sWaveSum[kPlanWaves + i];
        total += w;
        if (i < wave) wBase += w, gBase += g;
    }
    {
        uint32_t costBefore = wBase + wIncl - wSum, groupsBefore = gBase + gIncl - gSum;
#pragma unroll
        for (uint32_t i = 0; i < kPlanBinsPerThread; ++i) {
            const uint32_t k = kPlanBinsPerThread * tid + i;
            if (k < kFtxBins) {
                sCost[k] = costBefore;
                sGroups[k] = groupsBefore;  // (bins follow the sorted order, so the running count IS the group index)
            }
            costBefore += wMine[i];
            groupsBefore += gMine[i];
        }
    }
    __syncthreads();
    // CU slot c - 1 ends with the group at which the running cost reaches total * c / 32: bisect the bins' cost prefix
    if (tid <= 32) {
        uint32_t cut = tid == 32 ? nGroups : 0u;
        if (tid >= 1 && tid < 32 && total) {
            const uint64_t target32 = uint64_t(total) * tid;  // compare 32 * cost with it
            uint32_t lo = 0, hi = kFtxBins - 1;             // the last bin whose prefix is below the target
            while (lo < hi) {
                const uint32_t mid = (lo + hi + 1) / 2;
                if (uint64_t(sCost[mid]) * 32 < target32) lo = mid; else hi = mid - 1;
            }
            const uint32_t w = ftxBinCost(lo % kFtxQuartetBins, p.coldShift) + kFtxGroupCost;
            const uint64_t need = target32 - uint64_t(sCost[lo]) * 32;  // > 0
            const uint64_t extra = sHead[lo] ? 32ull * kFtxSegmentCost : 0ull;
            // groups of this bin up to and including the one that reaches it; inside a bucket head's surcharge: none - the bucket
            // starts the next slot, whose first slab load is the one every slot pays
            const uint32_t within = need <= extra ? 0u : uint32_t((need - extra + 32ull * w - 1) / (32ull * w));
            const uint32_t binGroups = (lo + 1 < kFtxBins ? sGroups[lo + 1] : nGroups) - sGroups[lo];
            cut = sGroups[lo] + min(within, binGroups);
        }
        sCut[tid] = cut;  // (ascending: the targets ascend)
    }
    __syncthreads();
    // segments: CU slot c's range [cut c, cut c + 1) cut again where the bucket changes - one thread per slot
    if (wave == 0) {
        uint32_t start = 0, end = 0, mine = 0;
        if (lane < 32) {
            start = sCut[lane], end = max(sCut[lane + 1], start);
            if (end > start) {
                mine = 1;
                for (uint32_t bb = 1; bb < 16; ++bb) {  // bucket starts strictly inside (an empty bucket's start is its successor's)
                    const uint32_t at = sBucketStart[bb] / 8;
                    if (at > start && at < end && at != sBucketStart[bb + 1] / 8) ++mine;
                }
            }
        }
        const uint32_t incl = waveInclusiveScan(mine);
        uint32_t k = incl - mine;
        if (lane < 32) {
            p.plan[lane] = k;
            uint32_t G = start, bb = 0;
            while (G < end) {
                while (bb < 15 && G >= sBucketStart[bb + 1] / 8) ++bb;  // the bucket of group G
                const uint32_t e = min(end, bb < 15 ? sBucketStart[bb + 1] / 8 : end);
                p.plan[64 + 3 * k] = bb;
                p.plan[64 + 3 * k + 1] = G;
                p.plan[64 + 3 * k + 2] = e;
                ++k;
                G = e;
            }
        }
        if (lane == 63) {
            p.plan[32] = incl;
            p.plan[33] = nGroups;
        }
    }
}

// counting sort, part 2: the perspective's head and list at its place in the sorted order
__global__ void spx_ftx_scatter_kernel(FtxParams p) {
#if SPX_CORUNNER_PRIO
    __builtin_amdgcn_s_setprio(SPX_CORUNNER_PRIO);  // (A/B: the kernels that run beside the gather ask for issue priority)
#endif
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= 2 * p.nPositions) return;
    const u32x4 head = *reinterpret_cast<const u32x4*>(p.heads + 4 * size_t(q));
    reinterpret_cast<u32x4*>(p.sorted)[p.binStart[head[2]] + p.ranks[q]] = u32x4{head[0], head[1], q * (kFtxListStride * 4u), q};
    if (p.posOrder && !(q & 1u)) p.posOrder[p.binStart[kFtxBins + 17 + (head[3] & 7u)] + (head[3] >> 3)] = q >> 1;
}

// ---------------------------------------------------------------------------------------------------------------------
// Pack: one wave per group of 8 neighbours of the sorted order writes the group's walk (spx_ftx.h) - the sections' lengths, the
// output slots, and the lists cut into interleaved stages with their padding. Lane (g = lane >> 3, ks = lane & 7) carries the
// four rows of step ks of perspective g: 16 bytes of its list, 8 consecutive lanes one 128-byte line. (A variant without LDS - a lane
// loads the four words it stores, 4-byte loads from four lists, 30 registers - and non-temporal stage stores were measured: equal
// beside a gather, 1-2 % slower stream-ordered: docs/experiments.md 6. Measurement builds that SKIP this kernel run the pipelined
// step 8 % faster, those that skip the extraction 6.5 %.)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * kWavesPerBlock) void spx_ftx_pack_kernel(FtxParams p) {
#if SPX_CORUNNER_PRIO
    __builtin_amdgcn_s_setprio(SPX_CORUNNER_PRIO);  // (A/B: the kernels that run beside the gather ask for issue priority)
#endif
    __shared__ __align__(16) uint32_t sStage[kWavesPerBlock][256];
    const uint32_t lane = laneId(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), G = blockIdx.x * kWavesPerBlock + wave;
    if (G >= p.plan[33]) return;
    const uint32_t g = lane >> 3, ks = lane & 7u;
    const uint32_t fillAt = 32 * ks + 4 * (g & 1u) + (g >> 1);  // (+ 8 i for row i of the step)
    const u32x4 head = reinterpret_cast<const u32x4*>(p.sorted)[8 * G + g];
    const uint32_t cHi = head[0] & 0x3Fu, cLds = (head[0] >> 6) & 0x1FFu, cCold = (head[0] >> 15) & 0x1FFu;
    // the sections' lengths: the longest of the 8 lists, in quartets (9 + 7 + 7 bits of one word per lane, three shuffles)
    uint32_t dims = ((cHi + 3) >> 2) | (((cLds + 3) >> 2) << 8) | (((cCold + 3) >> 2) << 16);
#pragma unroll
    for (int dlt = 8; dlt < 64; dlt <<= 1) {
        const uint32_t other = uint32_t(__shfl_xor(int(dims), dlt, 64));
        dims = max(dims & 0xFFu, other & 0xFFu) | max(dims & 0xFF00u, other & 0xFF00u) | max(dims & 0xFF0000u, other & 0xFF0000u);
    }
    dims = __builtin_amdgcn_readfirstlane(dims);
    const uint32_t hiQ = dims & 0xFFu, ldsQ = (dims >> 8) & 0xFFu, coldQ = dims >> 16;
    uint32_t* gh = p.groupHead + size_t(G) * kFtxGroupHeadWords;
    if (lane == 0) gh[0] = dims;
    if (ks == 0) gh[1 + g] = head[1];
    const uint32_t H = (hiQ + 7) >> 3, L = (ldsQ + 15) >> 4, Q = H + L + ((coldQ + 15) >> 4);
    uint32_t* out = p.stages + size_t(G) * (kFtxMaxStages * 256);
    {   // what the gather will walk, in the spare words of the group's head (spx_debug_ftx_walk sums them on the host: bench.py's
        // instruction counts. Six atomic adds per group on one cache line made this kernel the pipeline's longest - 0.4 ms)
        uint32_t rowsG = ks == 0 ? cCold | (cHi << 16) : 0u, rowsL = ks == 0 ? cLds : 0u;  // (cold | high planes << 16: <= 2 048 | 256 per group)
#pragma unroll
        for (int dlt = 8; dlt < 64; dlt <<= 1) {
            rowsG += uint32_t(__shfl_xor(int(rowsG), dlt, 64));
            rowsL += uint32_t(__shfl_xor(int(rowsL), dlt, 64));
        }
        if (lane == 0) {
            gh[9] = Q;
            gh[10] = coldQ;
            gh[11] = ldsQ;
            gh[12] = rowsG & 0xFFFFu;
            gh[13] = rowsL;
            gh[14] = hiQ;  // as packed: an XCD walks what is left after dropping the planes that are zero in its slice
            gh[15] = rowsG >> 16;
        }
    }
    const uint8_t* lists = reinterpret_cast<const uint8_t*>(p.lists);
    // ---- the high-byte section: stages of 8 steps, 32-bit entries (slice offset | hiMask << 24): the gather compacts them per slice ----
    for (uint32_t q = 0; q < H; ++q) {
        const uint32_t first = 4 * (8 * q + ks), left = cHi > first ? cHi - first : 0u;
        u32x4 v = {0, 0, 0, 0};
        if (left) v = *reinterpret_cast<const u32x4*>(lists + head[2] + 4 * (kFtxListHi + first));
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {  // in which slices the plane is not all zero (bits 24-31; the entry itself < 2^24)
            if (left > i) v[i] |= uint32_t(p.hiMask[(v[i] >> 7) - kFtxPsqHiBase]) << 24;
        }
        uint32_t* st = &sStage[wave][fillAt];
        st[0] = left > 0 ? v[0] : kFtxZeroRow * 128u;
        st[8] = left > 1 ? v[1] : kFtxZeroRow * 128u;
        st[16] = left > 2 ? v[2] : kFtxZeroRow * 128u;
        st[24] = left > 3 ? v[3] : kFtxZeroRow * 128u;
        __builtin_amdgcn_wave_barrier();
        const u32x4 line = *reinterpret_cast<const u32x4*>(&sStage[wave][4 * lane]);
        __builtin_amdgcn_wave_barrier();
        *reinterpret_cast<u32x4*>(out + size_t(q) * 256 + 4 * lane) = line;
    }
    // ---- the LDS section and the cold section: stages of SIXTEEN steps, 16-bit entries = row index from the section's base (the
    // gather's LDS / the slice), spx_ftx.h. This lane carries steps ks and ks + 8 of perspective g: two 16-byte pieces of its list.
    // The next stage's pieces travel while this one is written; a stage passes through LDS so that it leaves as ONE coalesced 1 KiB
    // store (beside a gather every vector-memory instruction queues behind the gather's row loads)
    // (a stage with an ODD number of steps - a section's last - holds them in its steps 1 .. n, behind a step of padding the gather
    // does not fetch: its walk takes the steps in aligned pairs and enters an odd stage through the second half of the first pair)
    auto place = [&](uint32_t q, uint32_t c, uint32_t& at, uint32_t& left) {
        const uint32_t isCold = q >= H + L ? 1u : 0u, s = q - H - isCold * L;
        const uint32_t shift = min((isCold ? coldQ : ldsQ) - 16 * s, 16u) & 1u, k = ks + 8 * c;
        const uint32_t count = isCold ? cCold : cLds, first = 4 * (16 * s + k - shift);
        left = (k >= shift && count > first) ? count - first : 0u;
        at = head[2] + 4 * ((isCold ? kFtxListCold : kFtxListLds) + first);
    };
    uint32_t at[2] = {0, 0}, left[2] = {0, 0};
    u32x4 next[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    if (Q > H) {
#pragma unroll
        for (uint32_t c = 0; c < 2; ++c) {
            place(H, c, at[c], left[c]);
            if (left[c]) next[c] = *reinterpret_cast<const u32x4*>(lists + at[c]);
        }
    }
    uint16_t* const sHalf = reinterpret_cast<uint16_t*>(sStage[wave]);
    const uint32_t u = g & 1u, pr = g >> 1;
    for (uint32_t q = H; q < Q; ++q) {
        const u32x4 v[2] = {next[0], next[1]};
        const uint32_t leftNow[2] = {left[0], left[1]};
        const uint32_t zero = q >= H + L ? kFtxColdZeroRow : kFtxSlabRows;  // (the table's zero row behind the threat rows / the slab's)
        if (q + 1 < Q) {
#pragma unroll
            for (uint32_t c = 0; c < 2; ++c) {
                place(q + 1, c, at[c], left[c]);
                next[c] = u32x4{0, 0, 0, 0};
                if (left[c]) next[c] = *reinterpret_cast<const u32x4*>(lists + at[c]);
            }
        }
#pragma unroll
        for (uint32_t c = 0; c < 2; ++c) {
            const uint32_t k = ks + 8 * c;  // halfword ((k >> 1) * 8 + 2 kb + u) * 8 + (k & 1) * 4 + pr: row kb of step k, perspective 2 pr + u
            uint16_t* st = sHalf + ((k >> 1) * 8 + u) * 8 + (k & 1u) * 4 + pr;
#pragma unroll
            for (uint32_t kb = 0; kb < 4; ++kb) st[16 * kb] = uint16_t(leftNow[c] > kb ? v[c][kb] >> 7 : zero);
        }
        __builtin_amdgcn_wave_barrier();
        const u32x4 line = *reinterpret_cast<const u32x4*>(&sStage[wave][4 * lane]);
        __builtin_amdgcn_wave_barrier();
        *reinterpret_cast<u32x4*>(out + size_t(q) * 256 + 4 * lane) = line;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Gather.
// ---------------------------------------------------------------------------------------------------------------------
// 256 persistent workgroups: CU slot `cu` of XCD `xcd` walks its planned share, segment by segment (a segment = one bucket:
// its slab slice goes to LDS once; the hot rows' slices went there when the workgroup started), its 16 waves striding over the
// segment's groups. A wave walks a group's packed stages (spx_ftx_pack_kernel): one coalesced 1 KiB load per stage of 8 steps -
// the next stage's, at a group's last stage the next group's first, travels while the stage is walked - parked in the wave's
// 1 KiB of LDS, from where every step's lanes pick the four rows they fetch (lane (n = 8 u + t, kb): row kb of perspectives
// 2 pr + u, 16-byte chunk t): 2 perspectives x 4 rows x 128 B per wave load / LDS read, ONE v_mfma_i32_16x16x64_i8 each.
// Round 5 rebuilt the walk after measuring what it is made of (profiles/r05_gather_anatomy.txt: with its row loads AND its
// MFMAs removed the round-4 kernel still took 160 of 281 us; without its global row loads as long as with them):
//   * every XCD found out for itself what a group's walk looks like - section boundaries searched per lane and stage, list
//     gathers, padding: now packed once per batch by a pass of its own;
//   * the loop issued 8 row loads, waited for all of them, fed 8 MFMAs and started over: now a stage's steps run as a ROLLING
//     window of two (the loads of step k + 2 are issued right behind the MFMAs of step k: 8 in flight all the time), and every
//     loop body has ONE kind of load, so the compiler's s_waitcnt are exact (a join of paths with different numbers of loads in
//     flight waits for the smaller number);
//   * a group ended with four 2-byte stores per lane behind four cross-lane shuffles: now the wave's LDS transposes its 512
//     output bytes and every lane stores 8.
// (Round-4 measurements that still shape it, profiles/r04_sliced_pipeline_overlap_attempts.txt: groups claimed from work queues
// with finished workgroups helping - no gain, every workgroup slows down alike, there is no tail; chunks of groups through the
// hardware dispatcher - a slab reload per chunk, 18-34 % slower; s_setprio, a start gate, a high-priority stream - nothing or worse.)
namespace {

// one stage - n <= 8 steps, 32-bit entries - of a group's walk (since the 16-bit entries of round 6: the HIGH-BYTE sections' stages, as
// compacted per slice; the LDS and cold sections walk walkStage16 below): rows from LDS (kLds) or through the texture path. The steps run in
// PAIRS as a rolling window: the loads of step k + 2 are issued right behind the MFMAs of step k, so four to eight loads are in
// flight all the time, and the loop body has no branch. Round 5 padded an odd section with a step of zero rows (with a branch for the
// odd step the compiler's s_waitcnt at the join wait for ALL loads and the accumulators travel through copies); round 6 enters the
// loop through its second half instead (below).
template <bool kLds>
__device__ __forceinline__ void walkStage(uint32_t n, const uint32_t* stage, const uint8_t* ldsRows, const uint8_t* slice, uint32_t e,
                                          uint32_t laneOff, const i32x4& sel, i32x4 (&d)[4]) {
    auto entries = [&](uint32_t k) { return *reinterpret_cast<const u32x4*>(stage + 4 * (8 * k + e)); };
    auto issue = [&](const u32x4& en, i32x4 (&w)[4]) {
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            if constexpr (kLds) {
                if (SPX_FTX_SKIP & 1) w[pr] = i32x4{int(en[pr]), 0, 0, 0};
                else if (SPX_FTX_SKIP & 64)  // (measurement: every LDS row read forced onto the bank half of its kb - no bank conflicts, wrong rows)
                    w[pr] = *reinterpret_cast<const i32x4*>(ldsRows + ((en[pr] & ~128u) | (((laneId() >> 4) & 1u) << 7)) + laneOff);
                else w[pr] = *reinterpret_cast<const i32x4*>(ldsRows + en[pr] + laneOff);
            } else {
                if (SPX_FTX_SKIP & 2) w[pr] = i32x4{int(en[pr]), 0, 0, 0};
                else w[pr] = *reinterpret_cast<const i32x4*>(slice + size_t(en[pr] + laneOff));
            }
        }
    };
    auto add = [&](const i32x4 (&w)[4]) {
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            if (SPX_FTX_SKIP & 4) d[pr] += w[pr];
            else d[pr] = __builtin_amdgcn_mfma_i32_16x16x64_i8(sel, w[pr], d[pr], 0, 0, 0);
        }
    };
    i32x4 wa[4], wb[4];
    u32x4 en;
    // Round 6. (a) An odd section does NOT walk a step of zero rows: its first step goes to the window's second half while the first
    // holds zeros - four MFMAs that add nothing instead of four row loads / LDS reads of the all-zero row (tools/sim_gather_steps.py:
    // the padding to pairs was 7 % of the wave loads and 4 % of the LDS reads); both entries into the loop leave `wb` in flight LAST, so
    // the waits at the join stay exact. (b) The halves of the loop are FENCED: left alone the scheduler clustered the loop's eight
    // loads behind its eight MFMAs, and the window drained to nothing once per pair of steps. (c) A step's entries are read one half
    // ahead: the LDS counter is in order, so an entries read issued behind the other half's row reads could only be waited for
    // together with them (the LDS sections drained at every step), and in the global sections its latency sat between the MFMAs and
    // the loads they make room for.
    u32x4 enN;
    uint32_t k;
    if (n & 1u) {
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) wa[pr] = i32x4{0, 0, 0, 0};
        en = entries(0);
        enN = entries(1);
        issue(en, wb);
        k = 1;
    } else {
        en = entries(0);
        enN = entries(1);
        issue(en, wa);
        en = enN;
        enN = entries(2);
        issue(en, wb);
        k = 2;
    }
    for (; k < n; k += 2) {  // steps k - 2, k - 1 are in flight, enN = the entries of step k
        en = enN;
        enN = entries(k + 1);
        __builtin_amdgcn_sched_barrier(0);
        add(wa);
        issue(en, wa);
        __builtin_amdgcn_sched_barrier(0);
        en = enN;
        enN = entries(min(k + 2, 7u));
        __builtin_amdgcn_sched_barrier(0);
        add(wb);
        issue(en, wb);
        __builtin_amdgcn_sched_barrier(0);
    }
    add(wa);
    add(wb);
}

// The same for the LDS and the cold sections, whose stages hold SIXTEEN steps of 16-bit entries (round 6; spx_ftx.h): one 16-byte LDS
// read brings the entries of a PAIR of steps (words 0, 1: the four rows of step 2 p, perspectives 2 pr + u; words 2, 3: step 2 p + 1) -
// half the entry reads of the 32-bit form, half the stage loads and the pack kernel's stores. A row's address is base + (entry << 7)
// + the lane's 16 bytes. The window rolls as above; a pair's entries are read while the pair before it is added up; a stage with an
// odd number of steps keeps them in steps 1 .. n and is entered through the second half of its first pair.
template <bool kLds>
__device__ __forceinline__ void walkStage16(uint32_t n, const uint32_t* stage, const uint8_t* ldsRows, const uint8_t* slice, uint32_t e,
                                            uint32_t laneOff, const i32x4& sel, i32x4 (&d)[4]) {
    auto entries = [&](uint32_t pair) { return *reinterpret_cast<const u32x4*>(stage + 4 * (8 * pair + e)); };
    auto issue = [&](uint32_t lo, uint32_t hi, i32x4 (&w)[4]) {  // (lo: rows of perspectives u, 2 + u; hi: 4 + u, 6 + u)
        // offset = (16-bit entry << 7) + base: ONE v_mad_u32_u16 each (op_sel picks the half; as C the compiler spends 2-3 instructions)
        uint32_t off[4];
        const uint32_t base = kLds ? uint32_t(reinterpret_cast<uintptr_t>(ldsRows)) + laneOff : laneOff;
        asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(off[0]) : "v"(lo), "s"(128u), "v"(base));
        asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(off[1]) : "v"(lo), "s"(128u), "v"(base));
        asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(off[2]) : "v"(hi), "s"(128u), "v"(base));
        asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(off[3]) : "v"(hi), "s"(128u), "v"(base));
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            if constexpr (kLds) w[pr] = *reinterpret_cast<const __attribute__((address_space(3))) i32x4*>(off[pr]);
            else w[pr] = *reinterpret_cast<const i32x4*>(slice + size_t(off[pr]));
        }
    };
    auto add = [&](const i32x4 (&w)[4]) {
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) d[pr] = __builtin_amdgcn_mfma_i32_16x16x64_i8(sel, w[pr], d[pr], 0, 0, 0);
    };
    i32x4 wa[4], wb[4];
    const uint32_t pairs = (n + 1) >> 1;
    u32x4 en = entries(0), enN = entries(1);
    if (n & 1u) {  // (an odd stage's first step is padding, spx_ftx_pack_kernel: zeros instead of four row fetches)
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) wa[pr] = i32x4{0, 0, 0, 0};
    } else {
        issue(en[0], en[1], wa);
    }
    issue(en[2], en[3], wb);
    for (uint32_t pair = 1; pair < pairs; ++pair) {  // the pair before is in flight, enN = this pair's entries
        en = enN;
        enN = entries(min(pair + 1, 7u));
        __builtin_amdgcn_sched_barrier(0);
        add(wa);
        issue(en[0], en[1], wa);
        __builtin_amdgcn_sched_barrier(0);
        add(wb);
        issue(en[2], en[3], wb);
        __builtin_amdgcn_sched_barrier(0);
    }
    add(wa);
    add(wb);
}

}  // namespace

__global__ __attribute__((amdgpu_flat_work_group_size(64 * kGatherWaves, 64 * kGatherWaves), amdgpu_waves_per_eu(SPX_FTX_GATHER_WAVES_PER_SIMD, 8))) void spx_ftx_gather_kernel(FtxParams p) {
    // LDS is DYNAMIC on purpose: its size depends on the context's hot set - and with a static 120 KiB the compiler knows that
    // only four waves per SIMD can be resident and pads the kernel's register count up to that occupancy's floor
    extern __shared__ __align__(16) uint8_t sDyn[];
    uint8_t* const sSlab = sDyn;                                   // the bucket's slab slice + an all-zero row, then the hot rows
    // (the wave's index as a SCALAR: everything derived from it - group indices, stage addresses - stays in SGPRs)
    const uint32_t lane = laneId(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // per wave: the current stage (1 KiB; at a group's end its 512 output bytes pass through it), then the group's head (64 B)
    uint32_t* const stage = reinterpret_cast<uint32_t*>(sDyn + kFtxSlabBytes + p.hotRows * 128u) + wave * (kFtxRingBytesPerWave / 4);
    uint32_t* const sHead = stage + 256;
    const uint32_t xcd = blockIdx.x & 7u, cu = blockIdx.x >> 3;
    const uint32_t n = lane & 15u, kb = lane >> 4, u = n >> 3, t = n & 7u, e = 2 * kb + u;
    // diagnostics (spx_debug_ftx_block_times): when did this workgroup start and end (constant 100 MHz clock)
    if (threadIdx.x == 0) {
        reinterpret_cast<unsigned long long*>(p.plan + kFtxPlanTimes)[2 * blockIdx.x] = wall_clock64();
        reinterpret_cast<unsigned long long*>(p.plan + kFtxPlanTimes)[2 * blockIdx.x + 1] = 0;  // (end: the LAST wave's, below)
    }
    const uint8_t* slice = p.rowS + size_t(xcd) * kFtxSliceStride;
    const uint32_t laneOff = 16 * t;
    const i32x4 sel = mfmaSelector(lane);
    // this lane's output columns: D registers (0, 1) = columns (c, c + 1), (2, 3) = their partners (c + 512, c + 513)
    const uint32_t col = 64 * xcd + 8 * t + 2 * kb;
    for (uint32_t i = threadIdx.x; i < 8; i += blockDim.x) reinterpret_cast<u32x4*>(sSlab + kFtxSlabRows * 128)[i] = u32x4{0, 0, 0, 0};
    {   // the hot rows' slices: once per workgroup (the first segment's barrier publishes them)
        const u32x4* src = reinterpret_cast<const u32x4*>(p.hotS + size_t(xcd) * p.hotRows * 128u);
        for (uint32_t i = threadIdx.x; i < p.hotRows * 8; i += blockDim.x) reinterpret_cast<u32x4*>(sSlab + kFtxSlabBytes)[i] = src[i];
    }
    // (32-bit byte offsets from a scalar base: one address register per load instead of two)
    auto stageOfGroup = [&](uint32_t G, uint32_t q) {
        const u32x4* src = reinterpret_cast<const u32x4*>(reinterpret_cast<const uint8_t*>(p.stages) + (size_t(G) * kFtxMaxStages + q) * 1024 + 16 * lane);
        return (SPX_FTX_NT & 2) ? __builtin_nontemporal_load(src) : *src;
    };
    auto headOfGroup = [&](uint32_t G) { return lane < 9 ? p.groupHead[size_t(G) * kFtxGroupHeadWords + lane] : 0u; };
    uint32_t loaded = 0xFFFFFFFFu;
    const uint32_t ownEnd = p.plan[cu + 1];
    for (uint32_t seg = p.plan[cu]; seg < ownEnd; ++seg) {
        const uint32_t bucket = p.plan[64 + 3 * seg], gFirst = p.plan[64 + 3 * seg + 1], gEnd = p.plan[64 + 3 * seg + 2];
        if (bucket != loaded) {
            __syncthreads();  // the previous segment's readers are done with the slab
            const u32x4* src = reinterpret_cast<const u32x4*>(slice + size_t(kFtxPsqLoBase + bucket * kFtxSlabRows) * 128);
            for (uint32_t i = threadIdx.x; i < kFtxSlabRows * 8; i += blockDim.x) reinterpret_cast<u32x4*>(sSlab)[i] = src[i];
            __syncthreads();
            loaded = bucket;
        }
        uint32_t G = gFirst + wave;
        if (G >= gEnd) continue;
        uint32_t headNext = headOfGroup(G);
        u32x4 ents = stageOfGroup(G, 0);
        while (true) {
            // the group's head: from the registers it travelled in to the wave's LDS (the lanes' LDS operations are in order)
            if (lane < 9) sHead[lane] = headNext;
            __builtin_amdgcn_wave_barrier();
            const uint32_t dims = __builtin_amdgcn_readfirstlane(sHead[0]);
            const uint32_t hiQ = dims & 0xFFu, ldsQ = (dims >> 8) & 0xFFu, coldQ = dims >> 16;
            const uint32_t H = (hiQ + 7) >> 3, L = (ldsQ + 15) >> 4, Q = H + L + ((coldQ + 15) >> 4);
            const uint32_t nextG = G + kGatherWaves;
            const bool haveNext = nextG < gEnd;
            if (haveNext) headNext = headOfGroup(nextG);
            i32x4 d[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            // what travels while a stage is walked: the next stage - at the last one: the next group's first
            auto prefetch = [&](uint32_t q) {
                if (q + 1 < Q) {
                    ents = stageOfGroup(G, q + 1);
                } else if (haveNext) {
                    ents = stageOfGroup(nextG, 0);
                }
            };
            // (three loops, one per section, instead of one loop with a three-way branch: the accumulators then flow through ONE walk
            // each - as one loop they travelled through copies at every join, and the copies' registers were what spilled)
            uint32_t q = 0;
            for (; q < H; ++q) {
                // A stage of high-byte planes, compacted for THIS slice: a plane that is all zero in slice `xcd` (bit 24 + xcd of its
                // entry, spx_ftx_pack_kernel) is dropped, the perspective's remaining planes move up. Lane 8 k + 2 kb + u holds the
                // entries of step k, row kb of perspectives 2 pr + u (pr = 0 .. 3): the j-th kept entry of a perspective goes to
                // step j >> 2, row j & 3 - word 8 j + 4 u + pr. (Most wide rows of a heavy-tailed net have a handful of weights
                // outside i8: their plane is zero in most slices - 108 -> 92 row loads per position on the `realistic` preset.)
                *reinterpret_cast<u32x4*>(stage + 4 * lane) = u32x4{kFtxZeroRow * 128u, kFtxZeroRow * 128u, kFtxZeroRow * 128u, kFtxZeroRow * 128u};
                uint32_t longest = 0;
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    const bool keep = (ents[pr] >> (24u + xcd)) & 1u;
                    const uint64_t kept = __ballot(keep);
                    const uint64_t even = kept & 0x5555555555555555ull, odd = kept & 0xAAAAAAAAAAAAAAAAull;
                    const uint32_t j = prefixCount((lane & 1u) ? odd : even);
                    if (keep) stage[8 * j + 4 * (lane & 1u) + pr] = ents[pr] & 0xFFFFFFu;
                    longest = max(longest, uint32_t(max(popc64(even), popc64(odd))));
                }
                const uint32_t hiSteps = (longest + 3) >> 2;
                __builtin_amdgcn_wave_barrier();
                prefetch(q);
                if (hiSteps) walkStage<false>(hiSteps, stage, sSlab, slice, e, laneOff, sel, d);
                if (q + 1 == H) {  // the high-byte planes' sums count 256-fold
#pragma unroll
                    for (int pr = 0; pr < 4; ++pr) d[pr] = d[pr] << 8;
                }
            }
            for (; q < H + L; ++q) {
                if (!(SPX_FTX_SKIP & 32)) *reinterpret_cast<u32x4*>(stage + 4 * lane) = ents;  // (the last stage's reads came first)
                __builtin_amdgcn_wave_barrier();
                prefetch(q);
                walkStage16<true>(min(ldsQ - 16 * (q - H), 16u), stage, sSlab, slice, e, laneOff, sel, d);
            }
            for (; q < Q; ++q) {
                if (!(SPX_FTX_SKIP & 32)) *reinterpret_cast<u32x4*>(stage + 4 * lane) = ents;
                __builtin_amdgcn_wave_barrier();
                prefetch(q);
                walkStage16<false>(min(coldQ - 16 * (q - H - L), 16u), stage, sSlab, slice, e, laneOff, sel, d);
            }
            if (Q == 0 && haveNext) ents = stageOfGroup(nextG, 0);  // (a group of records without a single piece: malformed input)
            // pairwise activation (multilayer.h:108-145) of this lane's two columns of perspectives 2 pr + u; the 2 output bytes go
            // to byte 64 (2 pr + u) + 8 t + 2 kb of the wave's LDS, from where lane l stores bytes 8 l .. 8 l + 7: perspective l >> 3
            {
                const uint32_t biasA = *reinterpret_cast<const uint32_t*>(p.t.ftBias + col);
                const uint32_t biasB = *reinterpret_cast<const uint32_t*>(p.t.ftBias + 512 + col);
                uint16_t* const outBytes = reinterpret_cast<uint16_t*>(stage);
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    const uint32_t a = pkAdd16(biasA, __builtin_amdgcn_perm(uint32_t(d[pr][1]), uint32_t(d[pr][0]), 0x05040100u));
                    const uint32_t b = pkAdd16(biasB, __builtin_amdgcn_perm(uint32_t(d[pr][3]), uint32_t(d[pr][2]), 0x05040100u));
                    const i16x2 zero = {0, 0}, top = {255, 255};
                    const u16x2 i1 = __builtin_bit_cast(u16x2, __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(i16x2, a), zero), top));
                    const u16x2 i2 = __builtin_bit_cast(u16x2, __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(i16x2, b), zero), top));
                    const uint32_t o = __builtin_bit_cast(uint32_t, u16x2((i1 * i2) >> 9));
                    outBytes[32 * (2 * pr + u) + 4 * t + kb] = uint16_t((o & 0xFFu) | ((o >> 8) & 0xFF00u));
                }
                __builtin_amdgcn_wave_barrier();
                const u32x2 mine = *reinterpret_cast<const u32x2*>(stage + 2 * lane);
                const uint32_t dst = sHead[1 + (lane >> 3)];
                if ((SPX_FTX_SKIP & 8) ? (mine[0] == 0x12345678u && dst == 77u) : (dst != 0xFFFFFFFFu)) {
                    if (SPX_FTX_NT & 4) __builtin_nontemporal_store(mine, reinterpret_cast<u32x2*>(p.ftOut + size_t(dst) * kPairs + 64 * xcd + 8 * (lane & 7u)));
                    else *reinterpret_cast<u32x2*>(p.ftOut + size_t(dst) * kPairs + 64 * xcd + 8 * (lane & 7u)) = mine;
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (!haveNext) break;
            G = nextG;
        }
    }
    if (lane == 0) atomicMax(reinterpret_cast<unsigned long long*>(p.plan + kFtxPlanTimes) + 2 * blockIdx.x + 1, (unsigned long long)wall_clock64());
}

// ---------------------------------------------------------------------------------------------------------------------
// The hot set (round 5). spx_ftx_hist_kernel counts how often every threat / pawn-pair row is fetched by the lists of a batch
// that was extracted with an EMPTY hot set (all such rows sit in the cold sections); the host picks the most popular rows
// (spx_api.cpp: calibrateHotRows) and spx_ftx_build_hot_kernel lays their slices out in slot order and fills the row -> slot map.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void spx_ftx_hist_kernel(FtxParams p, uint32_t* counts, uint32_t* stats) {
    const uint32_t lane = laneId(), nPersp = 2 * p.nPositions;
    uint32_t hi = 0;
    for (uint32_t q = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; q < nPersp; q += (gridDim.x * blockDim.x) >> 6) {
        const uint32_t head = p.heads[4 * size_t(q)];
        const uint32_t nCold = head >> 15;
        const uint32_t* list = p.lists + size_t(q) * kFtxListStride + kFtxListCold;
        for (uint32_t i = lane; i < nCold; i += 64) atomicAdd(&counts[list[i] >> 7], 1u);
        if (lane == 0) hi += head & 0x3Fu;
    }
    if (hi) atomicAdd(&stats[0], hi);
}

__global__ void spx_ftx_build_hot_kernel(const uint8_t* rowS, const uint32_t* hotIds, uint32_t hotRows, uint8_t* hotS) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte chunk each
    if (idx >= hotRows * 64u) return;
    const uint32_t slot = idx >> 6, x = (idx >> 3) & 7u, t = idx & 7u, row = hotIds[slot];
    *reinterpret_cast<u32x4*>(hotS + (size_t(x) * hotRows + slot) * 128 + 16 * t) =
        *reinterpret_cast<const u32x4*>(rowS + (size_t(x) * kFtxRows + row) * 128 + 16 * t);
}

hipError_t launchFtxHistogram(const FtxParams& p, uint32_t* counts, uint32_t* stats, hipStream_t stream) {
    hipLaunchKernelGGL(spx_ftx_hist_kernel, dim3(1024), dim3(256), 0, stream, p, counts, stats);
    return hipGetLastError();
}

hipError_t launchFtxBuildHot(const uint8_t* rowS, const uint32_t* hotIds, uint32_t hotRows, uint8_t* hotS, hipStream_t stream) {
    if (!hotRows) return hipSuccess;
    hipLaunchKernelGGL(spx_ftx_build_hot_kernel, dim3((hotRows * 64u + 255) / 256), dim3(256), 0, stream, rowS, hotIds, hotRows, hotS);
    return hipGetLastError();
}

hipError_t launchFtxBuildTable(const uint8_t* thrU8, const int16_t* psqW, const uint32_t* lut, uint8_t* rowS, hipStream_t stream) {
    hipLaunchKernelGGL(spx_ftx_build_table_kernel, dim3((kFtxRows * 64u + 255) / 256), dim3(256), 0, stream, thrU8, psqW, lut, rowS);
    return hipGetLastError();
}

hipError_t launchFtxBuildHiMask(const uint8_t* rowS, uint8_t* hiMask, hipStream_t stream) {
    const hipError_t e = hipMemsetAsync(hiMask, 0, kPsqRows, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(spx_ftx_build_himask_kernel, dim3((kPsqRows * 8u + 255) / 256), dim3(256), 0, stream, rowS, reinterpret_cast<uint32_t*>(hiMask));
    return hipGetLastError();
}

#ifndef SPX_FTX_EXTRACT_GRID
#define SPX_FTX_EXTRACT_GRID 1024  // workgroups (8 waves each; A/B round 6: 512 and fewer lose 15 % pipelined - long-lived workgroups keep the next gather's off their CUs)
#endif
hipError_t launchFtxExtract(const FtxParams& p, hipStream_t stream) {
    const uint32_t extractBlocks = min((p.nPositions + kExtractWaves - 1) / kExtractWaves, uint32_t(SPX_FTX_EXTRACT_GRID));
    hipLaunchKernelGGL(spx_ftx_extract_kernel, dim3(extractBlocks), dim3(64 * kExtractWaves), 0, stream, p);
    return hipGetLastError();
}

#ifndef SPX_MEASURE_SKIP
#define SPX_MEASURE_SKIP 0  // measurement builds only (tools/build_variants.sh): after the first 12 batches of the process 1 = no extraction
#endif                      // (the lists of the scratch set's last batch stay), 2 = no rank / plan / scatter / pack, 4 = no MLP (spx_kernels.hip)

hipError_t launchFtxPrepare(const FtxParams& p, hipStream_t stream) {
    static std::atomic<uint32_t> calls{0};
    const bool warm = SPX_MEASURE_SKIP != 0 && calls.fetch_add(1) >= 12;
    if (!(warm && (SPX_MEASURE_SKIP & 1))) {
        const hipError_t e = launchFtxExtract(p, stream);
        if (e != hipSuccess) return e;
    }
    if (warm && (SPX_MEASURE_SKIP & 2)) return hipSuccess;
    if (warm && (SPX_MEASURE_SKIP & 24)) {  // 8 = no pack, 16 = no rank / plan / scatter (the sorted order of the last batch stays)
        const uint32_t nPersp = 2 * p.nPositions;
        if (!(SPX_MEASURE_SKIP & 16)) {
            hipLaunchKernelGGL(spx_ftx_rank_kernel, dim3((nPersp + 1023) / 1024), dim3(1024), 0, stream, p);
            hipLaunchKernelGGL(spx_ftx_plan_kernel, dim3(1), dim3(kPlanThreads), 0, stream, p);
            hipLaunchKernelGGL(spx_ftx_scatter_kernel, dim3((nPersp + 255) / 256), dim3(256), 0, stream, p);
        }
        if (!(SPX_MEASURE_SKIP & 8)) {
            hipLaunchKernelGGL(spx_ftx_pack_kernel, dim3((uint32_t(ftxGroups(p.nPositions)) + kWavesPerBlock - 1) / kWavesPerBlock), dim3(64 * kWavesPerBlock), 0, stream, p);
        }
        return hipGetLastError();
    }
    return launchFtxSortAndPlan(p, stream);
}

hipError_t launchFtxSortAndPlan(const FtxParams& p, hipStream_t stream) {
    const uint32_t nPersp = 2 * p.nPositions;
    hipLaunchKernelGGL(spx_ftx_rank_kernel, dim3((nPersp + 1023) / 1024), dim3(1024), 0, stream, p);
    hipLaunchKernelGGL(spx_ftx_plan_kernel, dim3(1), dim3(kPlanThreads), 0, stream, p);
    hipLaunchKernelGGL(spx_ftx_scatter_kernel, dim3((nPersp + 255) / 256), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(spx_ftx_pack_kernel, dim3((uint32_t(ftxGroups(p.nPositions)) + kWavesPerBlock - 1) / kWavesPerBlock), dim3(64 * kWavesPerBlock), 0, stream, p);
    return hipGetLastError();
}

// more than 64 KiB of dynamic LDS has to be allowed once per device (ensureFtx calls this: a failure there leaves the context on
// the one-kernel path)
hipError_t prepareFtxGather(int device) {
    static std::atomic<uint64_t> allowed{0};
    if (allowed.load() >> (device & 63) & 1u) return hipSuccess;
    const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&spx_ftx_gather_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, int(gatherLdsBytes(kFtxHotRowsMax)));
    if (attr != hipSuccess) return attr;
    allowed.fetch_or(uint64_t(1) << (device & 63));
    return hipSuccess;
}

hipError_t launchFtxGather(const FtxParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(spx_ftx_gather_kernel, dim3(256), dim3(64 * kGatherWaves), gatherLdsBytes(p.hotRows), stream, p);
    return hipGetLastError();
}

}  // namespace spx
