// gfx950 (MI355X, CDNA4) kernels of the batched NNUE evaluator. Hand-written HIP; wave64 throughout.
//
//   spx_ft_kernel    feature extraction + feature-transformer accumulation + pairwise activation.
//                    One wavefront per (position, perspective); lane l IS square l during extraction and owns
//                    accumulator columns {8l..8l+7} U {512+8l..512+8l+7} during accumulation, so the pairwise
//                    product (column j with j+512) is lane-local. A 1 KiB threat row is ONE coalesced 16 B/lane
//                    wave load, a 2 KiB piece-square row is two. L2-bound gather: this is the roofline kernel.
//   spx_mlp_kernel   int8 L1 (1024 -> 32, x8 output buckets) on v_mfma_i32_16x16x64_i8, then the i32 tail
//                    (dual activation, L2 64x64, L3 + skip, scale). < 1 % of int8 MFMA peak by construction.
//   spx_update_kernel[_v1|_observed], spx_slot_act_kernel, spx_adjust_kernel, spx_sort_*: see each kernel.
// The wave-level building blocks (board decode, row lists, gather, activation) live in spx_ft_device.h. Variants that were
// measured and rejected are kept, unbuilt, under experiments/ (numbers in DESIGN.md 4.1).
//
// Reference semantics (paths relative to /root/reference/src/eval):
//   nnue_state.cpp:612-634 evaluateOnce; :440-449 resetPsqAccumulator; :309-354 addThreatFeatures;
//   nnue/input.h:72-75,283-293 Accumulator::initBoth/add; nnue_state.cpp:89-145 applyThreatRows (i8 -> i16 widening);
//   nnue/arch/multilayer.h:92-152 activateFt; :154-257 propagateL1; :261-343 propagateL2; :345-447 propagateL3;
//   :484-489 final scale; nnue/output.h:51-54 output bucket.
#include <hip/hip_runtime.h>

#include <atomic>

#include <cstdint>

#include "spx_ft_device.h"
#ifndef SPX_CORUNNER_PRIO
#define SPX_CORUNNER_PRIO 0
#endif
#include "spx_ftx.h"

namespace spx {

// ---------------------------------------------------------------------------------------------------------------------
// Feature transformer kernel (full refresh).
// One wavefront per (position, perspective): grid-stride over perspectives q = 2*position + colour; `order` (optional)
// is a permutation of perspective ids (king-bucket sorted for L2 locality) - results are written by q, so any order
// gives identical output. Two output modes:
//   ftOut   != nullptr: pairwise-activated u8[512] halves (stm first) for the MLP kernel     == evaluateOnce
//   accOut  != nullptr: raw i16 accumulators into arena slot slots[position] + the record     == NnueState::reset
// ---------------------------------------------------------------------------------------------------------------------
// kNear: the net has near-compact piece-square rows (FtTables::outlierTab): they take the 1 KiB path, their wide weights'
//        remainders are summed through 4 KiB of LDS per wave (buildFullLists). Nets without such rows run the kNear = false
//        instantiation - the same code as before the feature existed.
// (Round 4 also ran this kernel's gather on the matrix pipe - gatherFullMfma, SPX_FT_MFMA_GATHER=1: half the VALU instructions, the
// same time, 88 MB of extra tables; retired in round 5 to experiments/r04_ft_kernel_gather_on_the_matrix_pipe.hip.txt. The
// column-sliced pipeline, spx_ftx.hip, is where that idea pays.)
template <bool kNear>
__global__ __launch_bounds__(64 * kWavesPerBlock, SPX_FT_WAVES_PER_SIMD) void spx_ft_kernel(FtParams p) {
    __shared__ uint32_t sLut[kLutWords];
    __shared__ __align__(16) int32_t sNear[kNear ? kWavesPerBlock : 1][kNear ? int(kL1) : 4];  // per-column remainder sums
    __shared__ uint32_t sThr[kWavesPerBlock][kU8Cap];  // byte offsets into the threat table
    __shared__ uint32_t sPsq[kWavesPerBlock][kPsqCap];     // byte offsets into the psq table
    __shared__ uint64_t sPseudo[kDeltaPseudoWords];        // pseudo-attack sets per (piece kind, square), 3 KiB

    if (p.clearWord && blockIdx.x == 0 && threadIdx.x == 0) *p.clearWord = 0;
    // nPerspPtr: the list in `order` was produced on the device (deferred refreshes of the update kernel) and so was its length
    const uint32_t nPersp = p.nPerspPtr ? min(*p.nPerspPtr, p.nPositions * 2) : p.nPositions * 2;
    // XCD-aware traversal: workgroup b runs on XCD b % 8 (observed dispatch order; affects speed only). The (king-bucket
    // sorted) perspective order is dealt to the XCDs in chunks of SPX_FT_CHUNK perspectives, round robin: all eight walk
    // the order side by side, so at any moment each private 4 MiB L2 holds the 0.7-1.4 MiB piece-square slab of the SAME
    // current bucket plus the hot threat rows - and every XCD gets the same mix of light and heavy buckets. Round 1 gave
    // each XCD one contiguous eighth: the slices differ in WORK (castled-king buckets average 76 rows per perspective,
    // advanced-king endgame buckets 40-50: the heaviest eighth of the bench batch carries 17 % more rows than the mean)
    // and the kernel waited for the slowest XCD: FT kernel 0.4407 -> 0.4210 ms.
    const uint32_t xcd = blockIdx.x & 7, blockInXcd = blockIdx.x >> 3, blocksPerXcd = gridDim.x >> 3;
    const uint32_t stride = blocksPerXcd * kWavesPerBlock;
    // (batches too small to give every XCD several chunks are dealt perspective by perspective: chunk = 1)
    const uint32_t chunkShift = nPersp >= 64u * SPX_FT_CHUNK ? uint32_t(__builtin_ctz(SPX_FT_CHUNK)) : 0u;
    const uint32_t nChunks = (nPersp + (1u << chunkShift) - 1) >> chunkShift;
    const uint32_t myItems = ((nChunks + 7 - xcd) / 8) << chunkShift;  // chunks xcd, xcd + 8, ...
    // A workgroup without items leaves BEFORE staging the tables: the grid of a rebuild pass is sized on the host for a list
    // whose length only the device knows (3-4 % of the perspectives in play - in self-play 90 % of its 12 288 workgroups find
    // nothing, and each read 14 KB of tables and sat at a barrier beside the other half's kernels): self-play +3.4 % at 4 096
    // seats, +2.3 % at 1 024; the incremental bench (a quarter of the grid, nothing running beside it) unchanged
    // (profiles/r03_ab_idle_workgroups_exit_before_staging.txt).
    if (blockInXcd * kWavesPerBlock >= myItems) return;
    for (int i = threadIdx.x; i < kLutWords; i += blockDim.x) {
        sLut[i] = p.t.lut[i];
    }
    for (int i = threadIdx.x; i < kDeltaPseudoWords; i += blockDim.x) {
        sPseudo[i] = p.t.deltaTab[kDeltaRayWords + i];
    }
    __syncthreads();

    const uint32_t lane = laneId();
    const uint32_t wave = threadIdx.x >> 6;
    for (uint32_t t = blockInXcd * kWavesPerBlock + wave; t < myItems; t += stride) {
        const uint32_t it = ((((t >> chunkShift) * 8 + xcd)) << chunkShift) + (t & ((1u << chunkShift) - 1));
        if (it >= nPersp) continue;  // the last chunk may be partial
        const uint32_t q = __builtin_amdgcn_readfirstlane(p.order ? p.order[it] : it);
        const uint32_t posIdx = q >> 1;
        const int c = int(q & 1);

        const uint8_t* rec = reinterpret_cast<const uint8_t*>(p.positions) + size_t(posIdx) * 32;
        const LaneBoard board = decodeBoard(rec, lane);
        uint32_t nPsq, nThr;
        const bool hasNear = buildFullLists<kNear>(board, c, lane, sLut, sPsq[wave], sThr[wave], nPsq, nThr, sPseudo,
                                                          p.t.outlierTab, sNear[kNear ? wave : 0]);
        uint32_t acc[8];
        gatherFull(p.t, lane, sPsq[wave], nPsq, sThr[wave], nThr, acc, (kNear && hasNear) ? sNear[kNear ? wave : 0] : nullptr);

        if (p.accOut) {
            const uint32_t slot = __builtin_amdgcn_readfirstlane(p.slots[posIdx]);
            storeAcc(p.accOut, slot, c, lane, acc);
            if (c == 0 && lane < 8) {  // the record travels with the slot (parent of later incremental updates)
                reinterpret_cast<uint32_t*>(p.slotRecords + size_t(slot) * 32)[lane] =
                    reinterpret_cast<const uint32_t*>(rec)[lane];
            }
        }
        if (p.ftOut) {
            const uint32_t half = (c == board.stm) ? 0u : 1u;  // stm half first (nnue_state.cpp:396-438)
            *reinterpret_cast<u32x2*>(p.ftOut + size_t(posIdx) * kL1 + half * kPairs + 8 * lane) = activate(acc);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Team variant of the feature-transformer kernel, for full refreshes of fewer perspectives than the chip has resident
// workgroups (<= 512: evaluate_once of up to 256 positions). Such a launch is bound by the LATENCY of one perspective's serial
// chain (lists, then ~65 rows in bursts of 8), so here the kWavesPerBlock waves of a workgroup share ONE perspective: every
// wave builds the (identical) row lists for itself - no barrier before the gather -, fetches its quarter of the rows, and
// wave 0 adds the partial accumulators up through LDS. A partial accumulator is ftBias + its rows - 128 per u8 row
// (gatherFull), so the sum of kWavesPerBlock of them carries the bias kWavesPerBlock times: the surplus is taken off again;
// everything is mod 2^16 per column, hence bit-identical to the one-wave kernel (test_team_kernel_matches_the_wave_kernel).
// NOT used for the update kernels' rebuild pass: with more items than resident workgroups the pass runs in rounds and pays
// table staging and list building four times over (measured: 44 -> 72 us, profiles/r03_ab_rebuild_pass_team_kernel.txt).
// ---------------------------------------------------------------------------------------------------------------------
template <bool kNear>
__global__ __launch_bounds__(64 * kWavesPerBlock, 4) void spx_ft_team_kernel(FtParams p) {  // (5 waves/SIMD: 96 VGPRs spill)
    __shared__ uint32_t sLut[kLutWords];
    __shared__ __align__(16) int32_t sNear[kNear ? int(kL1) : 4];  // remainders of near-compact rows: summed by wave 0 alone
    __shared__ uint32_t sThr[kWavesPerBlock][kU8Cap];
    __shared__ uint32_t sPsq[kWavesPerBlock][kPsqCap];
    __shared__ uint64_t sPseudo[kDeltaPseudoWords];
    __shared__ uint32_t sPart[kWavesPerBlock - 1][8][64];  // partial accumulators of waves 1.., 2 KiB each

    if (p.clearWord && blockIdx.x == 0 && threadIdx.x == 0) *p.clearWord = 0;
    const uint32_t nPersp = p.nPerspPtr ? min(*p.nPerspPtr, p.nPositions * 2) : p.nPositions * 2;
    if (blockIdx.x >= nPersp) return;  // (a device-produced list is usually much shorter than the grid)
    for (int i = threadIdx.x; i < kLutWords; i += blockDim.x) {
        sLut[i] = p.t.lut[i];
    }
    for (int i = threadIdx.x; i < kDeltaPseudoWords; i += blockDim.x) {
        sPseudo[i] = p.t.deltaTab[kDeltaRayWords + i];
    }
    __syncthreads();

    const uint32_t lane = laneId();
    const uint32_t wave = threadIdx.x >> 6;
    for (uint32_t it = blockIdx.x; it < nPersp; it += gridDim.x) {  // (block-uniform trip count: the barriers below are safe)
        const uint32_t q = __builtin_amdgcn_readfirstlane(p.order ? p.order[it] : it);
        const uint32_t posIdx = q >> 1;
        const int c = int(q & 1);
        const uint8_t* rec = reinterpret_cast<const uint8_t*>(p.positions) + size_t(posIdx) * 32;
        const LaneBoard board = decodeBoard(rec, lane);
        uint32_t nPsq, nThr;
        const bool hasNear = buildFullLists<kNear>(board, c, lane, sLut, sPsq[wave], sThr[wave], nPsq, nThr, sPseudo,
                                                   p.t.outlierTab, wave == 0 ? sNear : nullptr);
        // this wave's slice of both lists (the near-compact rows' remainders go in once: wave 0)
        const uint32_t p0 = nPsq * wave / kWavesPerBlock, p1 = nPsq * (wave + 1) / kWavesPerBlock;
        const uint32_t t0 = nThr * wave / kWavesPerBlock, t1 = nThr * (wave + 1) / kWavesPerBlock;
        uint32_t acc[8];
        gatherFull(p.t, lane, sPsq[wave] + p0, p1 - p0, sThr[wave] + t0, t1 - t0, acc,
                   (kNear && hasNear && wave == 0) ? sNear : nullptr);
        if (wave != 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) sPart[wave - 1][r][lane] = acc[r];
        }
        __syncthreads();
        if (wave == 0) {
            const u32x4 b0 = *reinterpret_cast<const u32x4*>(p.t.ftBias + 8 * lane);
            const u32x4 b1 = *reinterpret_cast<const u32x4*>(p.t.ftBias + 512 + 8 * lane);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint32_t bias = r < 4 ? b0[r & 3] : b1[r & 3];
#pragma unroll
                for (int w = 0; w < kWavesPerBlock - 1; ++w) acc[r] = pkSub16(pkAdd16(acc[r], sPart[w][r][lane]), bias);
            }
            if (p.accOut) {
                const uint32_t slot = __builtin_amdgcn_readfirstlane(p.slots[posIdx]);
                storeAcc(p.accOut, slot, c, lane, acc);
                if (c == 0 && lane < 8) {
                    reinterpret_cast<uint32_t*>(p.slotRecords + size_t(slot) * 32)[lane] =
                        reinterpret_cast<const uint32_t*>(rec)[lane];
                }
            }
            if (p.ftOut) {
                const uint32_t half = (c == board.stm) ? 0u : 1u;
                *reinterpret_cast<u32x2*>(p.ftOut + size_t(posIdx) * kL1 + half * kPairs + 8 * lane) = activate(acc);
            }
        }
        __syncthreads();  // the partial sums are consumed before the next perspective's are written
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Incremental update kernel: child accumulator = parent accumulator + added rows - removed rows.
// One wavefront per (parent slot -> child slot) record, both perspectives. Replaces, for a batch of independent
// records, what the reference does per ply in ensureUpToDate (nnue_state.cpp:636-697): updatePsq (:34-87),
// applyThreatUpdates (:356-394) with generatePpRows (:163-307), and the refreshes (:458-536).
//
// The reference captures deltas on the HOST while the move is made (BoardObserver, nnue_state.h:118-186). Here the
// wave derives the same delta on the DEVICE from the parent and child boards (lane = square): changed squares give the
// piece-square rows; for threats each lane compares its parent and child target sets and only emits the symmetric
// difference (kept pairs: same attacker, same victim, still attacking); pawn pairs likewise. Accumulators are sums of
// rows mod 2^16, so any exact delta yields bit-identical results to the reference's event-driven one
// (the invariant Stormphrax asserts itself, datagen.cpp:262). A perspective whose king changed piece-square bucket or
// crossed the d/e mirror line (psq.h:264-283, nnue_state.h:118-128) is rebuilt from scratch, as the reference does.
// ---------------------------------------------------------------------------------------------------------------------
// kSplit = false: one wavefront per record does both perspectives (board decoding and attack generation shared);
// kSplit = true: one wavefront per (record, perspective) - twice the waves, half the serial latency - for batches too
// small to fill the chip (the kernel is latency-bound there: 4 096 records = 36 us unsplit).
// Workgroup b runs on XCD b % 8 (observed dispatch order; speed only). The update kernels deal their items to the XCDs in
// round-robin CHUNKS of 64 consecutive items (as the feature-transformer kernel deals perspectives): the ~35 children of one
// parent - contiguous in self-play's batches - then meet in one private L2, so the parent's accumulators cross the fabric
// once instead of up to eight times and the delta rows siblings share (the moved piece leaving its square) hit; every XCD
// gets the same share whatever the (possibly device-resident) item count. Independent records are unaffected.
// Measured (profiles/r03_ab_update_xcd_chunks.txt): self-play at 4 096 seats +1.5 %, 16 384 seats and independent records
// unchanged. The single-launch kernel for tiny batches keeps the plain order (kChunked = false).
// kStream launches (materialising batches from 32 768 records on) read the parent accumulators with non-temporal loads
// too: each is used once, and streaming them keeps the 4 KiB per record from evicting weight rows (round 2 measured +5 % for
// one child per parent but -15 % in self-play, whose ~35 siblings share a parent; since round 3 self-play's big update is
// eval-only and never a kStream launch, so the hint is on). SPX_UPDATE_STREAM_PARENTS=0: cached parent loads.
#ifndef SPX_UPDATE_STREAM_PARENTS
#define SPX_UPDATE_STREAM_PARENTS 1
#endif
template <bool kChunked>
struct ItemWalk {
    uint32_t t, tEnd, stride, xcd;
    __device__ ItemWalk(uint32_t nItems, uint32_t wave) {
        if constexpr (kChunked) {
            xcd = blockIdx.x & 7u;
            t = (blockIdx.x >> 3) * kWavesPerBlock + wave;
            stride = (gridDim.x >> 3) * kWavesPerBlock;  // gridDim.x is a multiple of 8 (cappedGrid)
            tEnd = ((nItems + 511u) >> 9) << 6;          // per-XCD index space: chunks of 64, one chunk in eight is this XCD's
        } else {
            xcd = 0;
            t = blockIdx.x * kWavesPerBlock + wave;
            stride = gridDim.x * kWavesPerBlock;
            tEnd = nItems;
        }
    }
    __device__ uint32_t item() const {
        if constexpr (kChunked) return ((((t >> 6) << 3) + xcd) << 6) | (t & 63u);
        return t;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// Incremental update kernel, second generation (round 2; the round-1 kernel - two full attack generations, rebuilds inline - is retired to
// experiments/r01_update_kernel_board_diff.hip.txt): child
// accumulator = parent accumulator + added rows - removed rows, delta derived on the device from the two boards - but
// the threat delta comes from RAY WALKS around the changed squares (deltaCandidates, spx_device_math.h: one lane per
// (board, changed square, ray / knight slot), no loops) instead of two full attack generations and per-lane victim
// loops; pawn pairs only from the pawns that left or arrived; the parent accumulator is requested before the delta is
// derived and the delta rows are fetched four at a time. PMC of v1 at 65 536 records (profiles/r02_pmc_incremental_v1.txt):
// 1 065 VALU instructions per (record, perspective) wave and 64 % of the wave cycles parked on memory (one row load at a
// time); this kernel: see DESIGN.md 4.3.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

// u8 delta rows: `nAdd` rows are added, `nSub` rows subtracted. A subtracted row is accumulated as its byte-wise
// complement (255 - b per column), so ONE non-overflowing accumulator serves both signs:
// sum = sum(add) + 255 * nSub - sum(sub); with the +128 storage bias of the table the correction is
// 128 * nAdd + 127 * nSub per column (mod 2^16). nAdd + nSub <= 256 (16-bit fields: 256 * 255 < 2^16).
// Rows are fetched kN at a time - the round-1 kernel waited for every row before asking for the next.
template <int kN>
__device__ __forceinline__ void loadAddRows(const RowTable& table, uint32_t laneOff, const uint32_t* list, uint32_t flip,
                                            uint32_t (&tacc)[8]) {
    u32x4 w[kN];
#pragma unroll
    for (int u = 0; u < kN; ++u) {
        w[u] = loadRow16(table, 0u, list[u] + laneOff);  // row offset folded into the lane's VGPR offset: no v_readfirstlane (+1-1.5 %)
    }
#pragma unroll
    for (int u = 0; u < kN; ++u) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t v = w[u][d] ^ flip;
            tacc[2 * d] += unpackLo(v);
            tacc[2 * d + 1] += unpackHi(v);
        }
        __builtin_amdgcn_sched_barrier(0);  // consume row by row: hoisting all 8 * kN widenings costs 32 VGPRs
    }
}

// Four rows in flight per wave. 5 / 6 / 8 at a time (28-64 B of spills at 96 VGPRs) and 8 at 4 waves/SIMD were measured and
// all LOSE: update + rebuild pass per 65 536-record ply +4.6 % / +5.8 % / +13 % / +4.8 % (profiles/r03_ab_update_burst_depth.txt) -
// the kernel is bound by its L2-miss traffic, not short of loads in flight.
__device__ __forceinline__ void accumulateRows(const RowTable& table, uint32_t laneOff, const uint32_t* list, uint32_t n,
                                               uint32_t flip, uint32_t (&tacc)[8]) {
    uint32_t i = 0;
#pragma unroll 1
    for (; i + 4 <= n; i += 4) loadAddRows<4>(table, laneOff, list + i, flip, tacc);
    const uint32_t rest = n - i;  // wave-uniform
    if (rest == 3) {
        loadAddRows<3>(table, laneOff, list + i, flip, tacc);
    } else if (rest == 2) {
        loadAddRows<2>(table, laneOff, list + i, flip, tacc);
    } else if (rest == 1) {
        loadAddRows<1>(table, laneOff, list + i, flip, tacc);
    }
}

static_assert(2 * kDeltaCap <= 256, "applyU8Delta sums add + sub rows in 16-bit fields: at most 256 rows x 255");
__device__ __forceinline__ void applyU8Delta(const FtTables& t, uint32_t lane, const uint32_t* addList, uint32_t nAdd,
                                             const uint32_t* subList, uint32_t nSub, uint32_t (&acc)[8]) {
    const RowTable table = makeRowTable(t.thrW, kU8TableBytes);
    uint32_t tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    accumulateRows(table, 16 * lane, addList, nAdd, 0u, tacc);
    accumulateRows(table, 16 * lane, subList, nSub, 0xFFFFFFFFu, tacc);
    const uint32_t corr = (nAdd * 128u + nSub * 127u) & 0xFFFFu;
    const uint32_t corr2 = corr | (corr << 16);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        acc[r] = pkSub16(pkAdd16(acc[r], tacc[r]), corr2);
    }
}

// wide (i16) piece-square delta rows - only nets whose piece-square rows do not all fit i8 have any
__device__ __forceinline__ void applyWidePsqDelta(const FtTables& t, uint32_t lane, const uint32_t* subList, uint32_t nSub,
                                                  const uint32_t* addList, uint32_t nAdd, uint32_t (&acc)[8]) {
    const uint8_t* psqBase = reinterpret_cast<const uint8_t*>(t.psqW) + 16 * lane;
#pragma unroll 1
    for (uint32_t i = 0; i < nSub; ++i) {
        const uint8_t* row = psqBase + __builtin_amdgcn_readfirstlane(subList[i]);
        const u32x4 lo = *reinterpret_cast<const u32x4*>(row);
        const u32x4 hi = *reinterpret_cast<const u32x4*>(row + 1024);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[r] = pkSub16(acc[r], lo[r]);
            acc[4 + r] = pkSub16(acc[4 + r], hi[r]);
        }
    }
#pragma unroll 1
    for (uint32_t i = 0; i < nAdd; ++i) {
        const uint8_t* row = psqBase + __builtin_amdgcn_readfirstlane(addList[i]);
        const u32x4 lo = *reinterpret_cast<const u32x4*>(row);
        const u32x4 hi = *reinterpret_cast<const u32x4*>(row + 1024);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[r] = pkAdd16(acc[r], lo[r]);
            acc[4 + r] = pkAdd16(acc[4 + r], hi[r]);
        }
    }
}

// Appends the pawn-pair rows that involve the pawns in `moved` (pawns of this board the other board lacks) to `list`:
// lane = partner square. All masks are wave-uniform, so positions come from popcounts, not ballots.
__device__ __forceinline__ uint32_t emitPawnPairDelta(uint32_t* list, uint32_t n, uint64_t moved, uint64_t pawns,
                                                      uint64_t ownPawns, uint32_t lane, int x) {
    uint64_t done = 0;
    while (moved) {
        const int a = ctz64(moved);
        moved &= moved - 1;
        done |= 1ull << a;
        const uint64_t partners = pawns & ppMask(a) & ~done;
        const uint32_t idA = ppId(a ^ x, !((ownPawns >> a) & 1));
        if ((partners >> lane) & 1) {
            const uint32_t slot = n + uint32_t(popc64(partners & ((1ull << lane) - 1)));
            if (slot < uint32_t(kDeltaCap)) list[slot] = ppRow(idA, ppId(int(lane) ^ x, !((ownPawns >> lane) & 1))) * kL1;
        }
        n += uint32_t(popc64(partners));  // may exceed the capacity: the caller then rebuilds the perspective
    }
    return n;
}

}  // namespace

#ifndef SPX_UPDATE_WAVES
#define SPX_UPDATE_WAVES 5  // 96 VGPRs: no spills (6 -> 80 VGPRs spills 14-25)
#endif

// The delta row lists of one record's perspective(s) [cFirst, cLast), into LDS (phase 1 of spx_update_kernel; also the whole of
// spx_ftu_derive_kernel): piece-square rows of the changed squares (updatePsq, nnue_state.cpp:34-87), threat rows from ray walks
// around them (applyThreatUpdates :356-394), pawn pairs of the pawns that left / arrived (generatePpRows :163-307). subList /
// addList: u8-table rows (compact piece-square rows first), wideList[c][0 / 1]: wide piece-square rows to subtract / add.
// refresh[c]: the perspective must be rebuilt instead. Returns the child's side to move.
__device__ __forceinline__ int deriveDeltaLists(const uint8_t* parentRec, const uint8_t* childRec, uint32_t lane, int cFirst, int cLast,
                                                const uint32_t* sLut, const uint64_t* sTab, uint8_t (*mail)[64],
                                                uint32_t (*addList)[kDeltaCap], uint32_t (*subList)[kDeltaCap],
                                                uint32_t (*wideList)[2][8], uint32_t (&nAdd)[2], uint32_t (&nSub)[2],
                                                uint32_t (&nWideSub)[2], uint32_t (&nWideAdd)[2], bool (&refresh)[2]) {
        int childStm;
        {
            const LaneBoard pb = decodeBoard(parentRec, lane);
            const LaneBoard cb = decodeBoard(childRec, lane);
            childStm = cb.stm;
            mail[0][lane] = uint8_t(pb.piece);
            mail[1][lane] = uint8_t(cb.piece);
            const bool changedSq = pb.piece != cb.piece;
            const uint64_t changed = __ballot(changedSq);
            const uint32_t nChanged = uint32_t(popc64(changed));
            __builtin_amdgcn_wave_barrier();

            int x[2], kingC[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const uint64_t kingMaskP = pb.kingsBb & (c ? pb.whiteBb : ~pb.whiteBb);
                const uint64_t kingMaskC = cb.kingsBb & (c ? cb.whiteBb : ~cb.whiteBb);
                const int kingP = kingMaskP ? ctz64(kingMaskP) : 0;
                kingC[c] = kingMaskC ? ctz64(kingMaskC) : 0;
                const int relP = c == 0 ? (kingP ^ 56) : kingP, relC = c == 0 ? (kingC[c] ^ 56) : kingC[c];
                // a legal move changes at most 4 squares (castling); anything larger is not a one-move delta (the
                // caller paired unrelated boards) and is rebuilt, like a king that changed bucket or mirror half
                refresh[c] = kingBucket(relP) != kingBucket(relC) || ((kingP & 7) >= 4) != ((kingC[c] & 7) >= 4) ||
                             nChanged > 4;
                x[c] = perspXor(c, kingC[c]);  // bucket and mirror half are those of the parent too
            }

            // ---- piece-square delta: changed squares (updatePsq: <= 2 subs, <= 2 adds per move) ----
            const bool subLane = changedSq && pb.piece != kNoPiece, addLane = changedSq && cb.piece != kNoPiece;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (c < cFirst || c >= cLast || refresh[c]) continue;
                nSub[c] = emitPsqDeltaRows(subLane, subLane ? psqRow(c, pb.piece, int(lane), kingC[c]) : 0u, sLut,
                                           wideList[c][0], subList[c], nWideSub[c]);
                nAdd[c] = emitPsqDeltaRows(addLane, addLane ? psqRow(c, cb.piece, int(lane), kingC[c]) : 0u, sLut,
                                           wideList[c][1], addList[c], nWideAdd[c]);
            }

            // ---- threat delta: ray walks around the changed squares, lane = board << 5 | square index << 4 | slot;
            //      two changed squares per pass, so a second pass only for castling / en passant ----
            {
                uint64_t m = nChanged <= 4 ? changed : 0;
                const int b = int(lane >> 5);
                const uint64_t occB = b ? cb.occ : pb.occ;
#pragma unroll 1
                while (m) {
                    const int fA = ctz64(m);
                    m &= m - 1;
                    const int fB = m ? ctz64(m) : -1;
                    m &= m - 1;
                    const int f = (lane & 16) ? fB : fA;
                    uint32_t desc[2];
                    deltaCandidates(sTab, mail[b], occB, changed, max(f, 0), int(lane & 15), desc[0], desc[1]);
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        if (c < cFirst || c >= cLast || refresh[c]) continue;
                        const int flipColour = (c == 0) ? 1 : 0;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const bool have = f >= 0 && desc[j] != kNoDesc;
                            const int32_t row = descRow(sLut, sTab, have ? desc[j] : 0u, x[c], flipColour);
                            const uint64_t valid = __ballot(have && row >= 0);
                            const uint32_t lo = uint32_t(valid), hi = uint32_t(valid >> 32);
                            // parent-board lanes (0..31) -> rows to subtract, child-board lanes (32..63) -> rows to add
                            const uint32_t slot = lane < 32 ? nSub[c] + __builtin_amdgcn_mbcnt_lo(lo, 0u)
                                                            : nAdd[c] + __builtin_amdgcn_mbcnt_hi(hi, 0u);
                            if (((valid >> lane) & 1) && slot < uint32_t(kDeltaCap)) {
                                (lane < 32 ? subList[c] : addList[c])[slot] = uint32_t(row) * kL1;
                            }
                            nSub[c] += uint32_t(__builtin_popcount(lo));
                            nAdd[c] += uint32_t(__builtin_popcount(hi));
                        }
                    }
                }
            }

            // ---- pawn-pair delta (generatePpRows): pairs of the pawns that left / arrived ----
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (c < cFirst || c >= cLast || refresh[c]) continue;
                const uint64_t ownP = pb.pawnsBb & (c ? pb.whiteBb : ~pb.whiteBb), theirP = pb.pawnsBb & ~ownP;
                const uint64_t ownC = cb.pawnsBb & (c ? cb.whiteBb : ~cb.whiteBb), theirC = cb.pawnsBb & ~ownC;
                nSub[c] = emitPawnPairDelta(subList[c], nSub[c], (ownP & ~ownC) | (theirP & ~theirC), pb.pawnsBb, ownP,
                                            lane, x[c]);
                nAdd[c] = emitPawnPairDelta(addList[c], nAdd[c], (ownC & ~ownP) | (theirC & ~theirP), cb.pawnsBb, ownC,
                                            lane, x[c]);
            }
            // every emitter above has run: a list that outgrew its capacity (never in legal play) is not applied - its
            // perspective is rebuilt instead (applyU8Delta relies on nAdd + nSub <= 2 * kDeltaCap <= 256 rows)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (nSub[c] > uint32_t(kDeltaCap) || nAdd[c] > uint32_t(kDeltaCap)) refresh[c] = true;
            }
            __builtin_amdgcn_wave_barrier();
        }

        return childStm;
}

// kSplit = false: one wavefront per record does both perspectives (board decoding and the ray walks shared);
// kSplit = true: one wavefront per (record, perspective) - twice the waves for batches too small to fill the chip.
// Perspectives that must be REBUILT (king changed bucket / mirror half, boards more than one move apart) are not
// handled here: their ids (2 * record + colour) are appended to p.refreshList and the feature-transformer kernel,
// launched right behind this one on the same stream, rebuilds exactly those (3-4 % of the perspectives in play).
template <bool kSplit, bool kStream>
// (kSplit = small batches, bound by one record's latency: 4 waves/SIMD leave it the registers it spilled at 5)
__global__ __launch_bounds__(64 * kWavesPerBlock, kSplit ? (SPX_UPDATE_WAVES > 4 ? 4 : SPX_UPDATE_WAVES) : SPX_UPDATE_WAVES) void spx_update_kernel(UpdateParams p) {
    __shared__ uint32_t sLut[kLutWords];
    __shared__ uint64_t sTab[kDeltaTabWords];                  // ray / knight masks + pseudo-attack sets (11 KiB)
    __shared__ uint32_t sAdd[kWavesPerBlock][2][kDeltaCap];    // per perspective: u8 rows to add ...
    __shared__ uint32_t sSub[kWavesPerBlock][2][kDeltaCap];    // ... and to subtract (compact piece-square rows first)
    __shared__ uint32_t sWide[kWavesPerBlock][2][2][8];        // per perspective: wide piece-square rows to subtract / add
    __shared__ uint8_t sMail[kWavesPerBlock][2][64];           // piece per square of the parent / child board

    const uint32_t nRecords = p.nRecordsPtr ? min(*p.nRecordsPtr, p.nRecords) : p.nRecords;
    const uint32_t nItems = kSplit ? nRecords * 2 : nRecords;
    // (a counted launch is sized for its capacity: workgroups beyond the count leave before staging 19 KB of tables)
    if (ItemWalk<true>(nItems, 0).t >= ItemWalk<true>(nItems, 0).tEnd) return;
    for (int i = threadIdx.x; i < kLutWords; i += blockDim.x) {
        sLut[i] = p.t.lut[i];
    }
    for (int i = threadIdx.x; i < kDeltaTabWords; i += blockDim.x) {
        sTab[i] = p.t.deltaTab[i];
    }
    __syncthreads();

    const uint32_t lane = laneId();
    const uint32_t wave = threadIdx.x >> 6;

    for (ItemWalk<true> walk(nItems, wave); walk.t < walk.tEnd; walk.t += walk.stride) {
        const uint32_t item = walk.item();
        if (item >= nItems) continue;  // (the last chunk round may be partial)
        const uint32_t it = kSplit ? item >> 1 : item;
        const int cFirst = kSplit ? int(item & 1) : 0, cLast = kSplit ? cFirst + 1 : 2;
        const uint32_t parentSlot = __builtin_amdgcn_readfirstlane(p.parentSlots[it]);
        // childSlots == nullptr: EVAL-ONLY children - the activations leave through ftOut, no accumulator and no record is
        // stored (the reference never keeps accumulators of nodes it only evaluates, nnue_state.cpp:598-610)
        const uint32_t childSlot = p.childSlots ? __builtin_amdgcn_readfirstlane(p.childSlots[it]) : 0u;
        const uint8_t* childRec = reinterpret_cast<const uint8_t*>(p.childPositions) + size_t(it) * 32;
        const uint8_t* parentRec = p.slotRecords + size_t(parentSlot) * 32;

        // ================= phase 1: the delta row lists of the perspective(s), into LDS =================
        uint32_t nAdd[2] = {0, 0}, nSub[2] = {0, 0}, nWideSub[2] = {0, 0}, nWideAdd[2] = {0, 0};
        bool refresh[2] = {false, false};
        const int childStm = deriveDeltaLists(parentRec, childRec, lane, cFirst, cLast, sLut, sTab, sMail[wave], sAdd[wave], sSub[wave],
                                              sWide[wave], nAdd, nSub, nWideSub, nWideAdd, refresh);

        // ================= phase 2: child = parent - removed rows + added rows =================
#pragma unroll 1
        for (int c = cFirst; c < cLast; ++c) {
            // (selects, not indexing: a dynamically indexed register array would live in scratch memory)
            if (c ? refresh[1] : refresh[0]) {
                if (lane == 0) p.refreshList[atomicAdd(p.refreshCount, 1u)] = 2 * it + uint32_t(c);
                continue;
            }
            const uint32_t na = c ? nAdd[1] : nAdd[0], ns = c ? nSub[1] : nSub[0];
            const uint32_t nws = c ? nWideSub[1] : nWideSub[0], nwa = c ? nWideAdd[1] : nWideAdd[0];
            uint32_t acc[8];
            loadAcc<kStream && SPX_UPDATE_STREAM_PARENTS>(p.arena, parentSlot, c, lane, acc);
            applyWidePsqDelta(p.t, lane, sWide[wave][c][0], nws, sWide[wave][c][1], nwa, acc);
            applyU8Delta(p.t, lane, sAdd[wave][c], na, sSub[wave][c], ns, acc);
            if (p.childSlots) storeAcc<kStream>(p.arena, childSlot, c, lane, acc);
            if (p.ftOut) {  // fused evaluation of the child: activations straight from the registers
                const uint32_t half = (c == childStm) ? 0u : 1u;
                *reinterpret_cast<u32x2*>(p.ftOut + size_t(it) * kL1 + half * kPairs + 8 * lane) = activate(acc);
            }
        }
        __builtin_amdgcn_wave_barrier();  // this record's lists are dead before the next record's are written
        if (lane < 8 && cFirst == 0) {
            const uint32_t word = reinterpret_cast<const uint32_t*>(childRec)[lane];
            if (p.childSlots) reinterpret_cast<uint32_t*>(p.slotRecords + size_t(childSlot) * 32)[lane] = word;
            if (p.ftOut) reinterpret_cast<uint32_t*>(p.stagedRecords + size_t(it) * 32)[lane] = word;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// A whole pending PATH per wavefront pair: NnueState::ensureUpToDate (nnue_state.cpp:636-697) walks forward from the last
// clean ancestor and applies one ply after the other; here one wavefront per (path, perspective) does the same inside ONE
// launch, the accumulator staying in registers from ply to ply (updatePsq :34-87, applyThreatUpdates :356-394, rebuilds
// :458-536 inline) and every ply's accumulator written to its slot on the way, as the reference leaves every stack entry on
// the path clean. The drop-in stack (include/spx_nnue.hpp) used to pay one ~29 us synchronous call per pending ply.
// Delta derivation as in spx_update_kernel (ray walks around the changed squares), for the one perspective of the wave:
// round 3 replaced the two full attack generations of the round-1 kernel here - a path is one wave pair deep, so every
// instruction and every LDS round trip of the derivation is latency (6.1 -> see DESIGN.md 4.8 us per ply).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * kWavesPerBlock, 3) void spx_update_chain_kernel(ChainParams p) {  // (4 waves/SIMD: 128 VGPRs spill 52 B; the kernel runs paths a wave pair deep)
    __shared__ uint32_t sLut[kLutWords];
    __shared__ uint64_t sTab[kDeltaTabWords];               // ray / knight masks + pseudo-attack sets (11 KiB)
    __shared__ uint32_t sThr[kWavesPerBlock][kU8Cap];       // rebuilds: the full row lists
    __shared__ uint32_t sPsq[kWavesPerBlock][kPsqCap];
    __shared__ uint32_t sAdd[kWavesPerBlock][kDeltaCap];    // deltas: u8 rows to add (compact piece-square rows first) ...
    __shared__ uint32_t sSub[kWavesPerBlock][kDeltaCap];    // ... and to subtract
    __shared__ uint32_t sWide[kWavesPerBlock][2][8];        // wide piece-square rows to subtract / add
    __shared__ uint8_t sMail[kWavesPerBlock][2][64];        // piece per square of the parent / child board
    for (int i = threadIdx.x; i < kLutWords; i += blockDim.x) sLut[i] = p.t.lut[i];
    for (int i = threadIdx.x; i < kDeltaTabWords; i += blockDim.x) sTab[i] = p.t.deltaTab[i];
    __syncthreads();
    const uint32_t lane = laneId();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t item = blockIdx.x * kWavesPerBlock + wave;
    if (item >= 2 * p.nChains) return;
    const uint32_t chain = item >> 1;
    const int c = int(item & 1);
    // (first == nullptr: UNIT paths - path i is record i alone: a batch of independent one-ply updates, what the small-batch
    // entry points launch)
    const uint32_t first = p.first ? __builtin_amdgcn_readfirstlane(p.first[chain]) : chain;
    const uint32_t n = p.first ? __builtin_amdgcn_readfirstlane(p.count[chain]) : 1u;
    const uint32_t parentSlot = __builtin_amdgcn_readfirstlane(p.parentSlots[chain]);
    uint32_t acc[8];
    loadAcc(p.arena, parentSlot, c, lane, acc);
    LaneBoard pb = decodeBoard(p.slotRecords + size_t(parentSlot) * 32, lane);
    int childStm = pb.stm;
    const uint8_t* records = reinterpret_cast<const uint8_t*>(p.childPositions) + size_t(first) * 32;
    uint32_t nextWord = n ? loadRecordWord(records, lane) : 0u;  // a ply's record is requested one ply ahead (long paths of a
                                                                 // tree replay run one wave pair deep: every round trip shows)
    const int flipColour = (c == 0) ? 1 : 0;
#pragma unroll 1
    for (uint32_t k = 0; k < n; ++k) {
        const uint8_t* childRec = records + size_t(k) * 32;
        const uint32_t word = nextWord;  // (lane l < 8 holds dword l of the record: what the slot's record copy needs)
        const LaneBoard cb = decodeBoardWord(word, lane);
        if (k + 1 < n) nextWord = loadRecordWord(childRec + 32, lane);
        childStm = cb.stm;
        sMail[wave][0][lane] = uint8_t(pb.piece);
        sMail[wave][1][lane] = uint8_t(cb.piece);
        const bool changedSq = pb.piece != cb.piece;
        const uint64_t changed = __ballot(changedSq);
        const uint32_t nChanged = uint32_t(popc64(changed));
        __builtin_amdgcn_wave_barrier();
        const uint64_t kingMaskP = pb.kingsBb & (c ? pb.whiteBb : ~pb.whiteBb), kingMaskC = cb.kingsBb & (c ? cb.whiteBb : ~cb.whiteBb);
        const int kingP = kingMaskP ? ctz64(kingMaskP) : 0, kingC = kingMaskC ? ctz64(kingMaskC) : 0;
        const int relP = c == 0 ? (kingP ^ 56) : kingP, relC = c == 0 ? (kingC ^ 56) : kingC;
        // as in spx_update_kernel: rebuilt when the king changed bucket or mirror half, or the boards are not one move apart
        bool refresh = kingBucket(relP) != kingBucket(relC) || ((kingP & 7) >= 4) != ((kingC & 7) >= 4) || nChanged > 4;
        uint32_t nAdd = 0, nSub = 0, nWideSub = 0, nWideAdd = 0;
        if (!refresh) {
            // the delta of ONE perspective, derived like spx_update_kernel does it for two (ray walks around the changed squares:
            // a fifth of the instructions of two full attack generations - on a path one wave deep every instruction is latency)
            const int x = perspXor(c, kingC);
            const bool subLane = changedSq && pb.piece != kNoPiece, addLane = changedSq && cb.piece != kNoPiece;
            nSub = emitPsqDeltaRows(subLane, subLane ? psqRow(c, pb.piece, int(lane), kingC) : 0u, sLut, sWide[wave][0], sSub[wave],
                                    nWideSub);
            nAdd = emitPsqDeltaRows(addLane, addLane ? psqRow(c, cb.piece, int(lane), kingC) : 0u, sLut, sWide[wave][1], sAdd[wave],
                                    nWideAdd);
            {
                uint64_t m = changed;
                const int b = int(lane >> 5);  // lanes 0..31 walk the parent board, 32..63 the child board
                const uint64_t occB = b ? cb.occ : pb.occ;
#pragma unroll 1
                while (m) {
                    const int fA = ctz64(m);
                    m &= m - 1;
                    const int fB = m ? ctz64(m) : -1;
                    m &= m - 1;
                    const int f = (lane & 16) ? fB : fA;
                    uint32_t desc[2];
                    deltaCandidates(sTab, sMail[wave][b], occB, changed, max(f, 0), int(lane & 15), desc[0], desc[1]);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const bool have = f >= 0 && desc[j] != kNoDesc;
                        const int32_t row = descRow(sLut, sTab, have ? desc[j] : 0u, x, flipColour);
                        const uint64_t valid = __ballot(have && row >= 0);
                        const uint32_t lo = uint32_t(valid), hi = uint32_t(valid >> 32);
                        const uint32_t slot = lane < 32 ? nSub + __builtin_amdgcn_mbcnt_lo(lo, 0u) : nAdd + __builtin_amdgcn_mbcnt_hi(hi, 0u);
                        if (((valid >> lane) & 1) && slot < uint32_t(kDeltaCap)) {
                            (lane < 32 ? sSub[wave] : sAdd[wave])[slot] = uint32_t(row) * kL1;
                        }
                        nSub += uint32_t(__builtin_popcount(lo));
                        nAdd += uint32_t(__builtin_popcount(hi));
                    }
                }
            }
            {
                const uint64_t ownP = pb.pawnsBb & (c ? pb.whiteBb : ~pb.whiteBb), theirP = pb.pawnsBb & ~ownP;
                const uint64_t ownC = cb.pawnsBb & (c ? cb.whiteBb : ~cb.whiteBb), theirC = cb.pawnsBb & ~ownC;
                nSub = emitPawnPairDelta(sSub[wave], nSub, (ownP & ~ownC) | (theirP & ~theirC), pb.pawnsBb, ownP, lane, x);
                nAdd = emitPawnPairDelta(sAdd[wave], nAdd, (ownC & ~ownP) | (theirC & ~theirP), cb.pawnsBb, ownC, lane, x);
            }
            if (nSub > uint32_t(kDeltaCap) || nAdd > uint32_t(kDeltaCap)) refresh = true;  // (never in legal play)
        }
        __builtin_amdgcn_wave_barrier();
        if (refresh) {
            uint32_t nPsq, nThr;
            buildFullLists(cb, c, lane, sLut, sPsq[wave], sThr[wave], nPsq, nThr, sTab + kDeltaRayWords);
            gatherFull(p.t, lane, sPsq[wave], nPsq, sThr[wave], nThr, acc);
        } else {
            applyWidePsqDelta(p.t, lane, sWide[wave][0], nWideSub, sWide[wave][1], nWideAdd, acc);
            applyU8Delta(p.t, lane, sAdd[wave], nAdd, sSub[wave], nSub, acc);
        }
        if (p.childSlots) {  // (nullptr: eval-only - the activations below are all that leaves)
            const uint32_t childSlot = __builtin_amdgcn_readfirstlane(p.childSlots[first + k]);
            storeAcc(p.arena, childSlot, c, lane, acc);
            if (lane < 8 && c == 0) {
                reinterpret_cast<uint32_t*>(p.slotRecords + size_t(childSlot) * 32)[lane] = word;
            }
        }
        __builtin_amdgcn_wave_barrier();  // this ply's lists are dead before the next ply's are written
        pb = cb;
    }
    if (p.ftOut && n) {
        const uint32_t half = (c == childStm) ? 0u : 1u;
        *reinterpret_cast<u32x2*>(p.ftOut + size_t(chain) * kL1 + half * kPairs + 8 * lane) = activate(acc);
        if (lane < 8 && c == 0) {
            const uint8_t* last = records + size_t(n - 1) * 32;
            reinterpret_cast<uint32_t*>(p.stagedRecords + size_t(chain) * 32)[lane] = reinterpret_cast<const uint32_t*>(last)[lane];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Incremental update from HOST-CAPTURED deltas: the reference's own bookkeeping, applied on the device. Each record
// carries the UpdateContext a BoardObserver captured while the move was made (spx_move_delta: piece-square subs/adds,
// threat descriptors added/removed, pawn bitboards before/after, refresh flags, kings). Per perspective, exactly as
// ensureUpToDate does (nnue_state.cpp:636-697): refresh -> rebuild from the child board; otherwise updatePsq (:34-87)
// and applyThreatUpdates (:356-394: descriptors -> threatFeatureIndex, negatives dropped; generatePpRows :163-307 when
// the pawn structure changed). One lane maps one descriptor. Cheaper than spx_update_kernel (no attack generation)
// but it needs ~1 KB of host-built delta per move; both produce identical accumulators.
// spx_move_delta layout (bytes): 0 n_sub, 1 n_add, 2 n_added, 3 n_removed, 4 sub_piece[2], 6 sub_sq[2], 8 add_piece[2],
// 10 add_sq[2], 12 psq_refresh[2], 14 threat_refresh[2], 16 kings[2], 24 pawns_before[2] (u64), 40 pawns_after[2],
// 56 threats_added[128] (4 B each), 568 threats_removed[128]; sizeof = 1080.
// ---------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kDeltaBytes = 1080;

// One wavefront per (record, perspective): twice the waves, half the serial latency, no spills (as in spx_update_kernel).
__global__ __launch_bounds__(64 * kWavesPerBlock, 3) void spx_update_observed_kernel(UpdateParams p) {  // (4 waves/SIMD: 4 VGPRs spilled)
    __shared__ uint32_t sLut[kLutWords];
    __shared__ uint32_t sThr[kWavesPerBlock][kU8Cap];
    __shared__ uint32_t sPsq[kWavesPerBlock][kPsqCap];
    __shared__ uint32_t sSub[kWavesPerBlock][kU8Cap];
    __shared__ uint32_t sPsqDelta[kWavesPerBlock][2][8];

    for (int i = threadIdx.x; i < kLutWords; i += blockDim.x) {
        sLut[i] = p.t.lut[i];
    }
    __syncthreads();

    const uint32_t lane = laneId();
    const uint32_t wave = threadIdx.x >> 6;
    for (uint32_t item = blockIdx.x * kWavesPerBlock + wave; item < 2 * p.nRecords; item += gridDim.x * kWavesPerBlock) {
        const uint32_t it = item >> 1;
        const int cOnly = int(item & 1);
        const uint32_t parentSlot = __builtin_amdgcn_readfirstlane(p.parentSlots[it]);
        const uint32_t childSlot = __builtin_amdgcn_readfirstlane(p.childSlots[it]);
        const uint8_t* childRec = reinterpret_cast<const uint8_t*>(p.childPositions) + size_t(it) * 32;
        const uint8_t* delta = p.deltas + size_t(it) * kDeltaBytes;
        const uint32_t nDeltaSub = min(uint32_t(delta[0]), 2u), nDeltaAdd = min(uint32_t(delta[1]), 2u);
        const uint32_t nAdded = min(uint32_t(delta[2]), 128u), nRemoved = min(uint32_t(delta[3]), 128u);
        const uint64_t* pawnBbs = reinterpret_cast<const uint64_t*>(delta + 24);
        const uint64_t blackBefore = pawnBbs[0], whiteBefore = pawnBbs[1], blackAfter = pawnBbs[2], whiteAfter = pawnBbs[3];
        const bool pawnsChanged = blackBefore != blackAfter || whiteBefore != whiteAfter;
        const int childStm = (childRec[24] & 0x80) ? 0 : 1;

        {
            const int c = cOnly;
            uint32_t acc[8];
            if (delta[12 + c] || delta[14 + c]) {  // requiresPsqRefresh / requiresThreatRefresh
                const LaneBoard cb = decodeBoard(childRec, lane);
                uint32_t nPsq, nThr;
                buildFullLists(cb, c, lane, sLut, sPsq[wave], sThr[wave], nPsq, nThr);
                gatherFull(p.t, lane, sPsq[wave], nPsq, sThr[wave], nThr, acc);
            } else {
                const int kingSq = delta[16 + c];
                const int x = perspXor(c, kingSq);
                const int flipColour = (c == 0) ? 1 : 0;
                // updatePsq: <= 2 subs, <= 2 adds
                uint32_t nPsqSub, nPsqAdd;
                const bool subLane = lane < nDeltaSub, addLane = lane < nDeltaAdd;
                const uint32_t nSubCompact =
                    emitPsqDeltaRows(subLane, subLane ? psqRow(c, delta[4 + lane] % 12, delta[6 + lane] & 63, kingSq) : 0u,
                                     sLut, sPsqDelta[wave][0], sSub[wave], nPsqSub);
                const uint32_t nAddCompact =
                    emitPsqDeltaRows(addLane, addLane ? psqRow(c, delta[8 + lane] % 12, delta[10 + lane] & 63, kingSq) : 0u,
                                     sLut, sPsqDelta[wave][1], sThr[wave], nPsqAdd);
                // applyThreatUpdates: one lane per descriptor, two passes of 64 per list
                uint32_t nAdd = nAddCompact, nSub = nSubCompact;
#pragma unroll 1
                for (int list = 0; list < 2; ++list) {
                    const uint8_t* descs = delta + (list == 0 ? 56 : 568);
                    const uint32_t count = list == 0 ? nAdded : nRemoved;
                    uint32_t* out = list == 0 ? sThr[wave] : sSub[wave];
                    uint32_t n = list == 0 ? nAddCompact : nSubCompact;
                    for (uint32_t base = 0; base < count; base += 64) {
                        int32_t row = -1;
                        if (base + lane < count) {
                            const uint32_t d = reinterpret_cast<const uint32_t*>(descs)[base + lane];
                            const int attacker = int(d & 0xFF) ^ flipColour, asq = int((d >> 8) & 0xFF) ^ x;
                            const int attacked = int((d >> 16) & 0xFF) ^ flipColour, vsq = int(d >> 24) ^ x;
                            if (attacker < 12 && attacked < 12 && (attacker >> 1) != 5) {
                                row = threatRow(sLut, attacker, asq, piecePseudoAttacks(attacker, asq), attacked, vsq);
                            }
                        }
                        const uint64_t valid = __ballot(row >= 0);
                        if (row >= 0) out[n + prefixCount(valid)] = uint32_t(row) * kL1;
                        n += popc64(valid);
                    }
                    (list == 0 ? nAdd : nSub) = n;
                }
                // generatePpRows: pairs that exist on one side only (lane = square)
                if (pawnsChanged) {
                    const uint64_t ownP = c ? whiteBefore : blackBefore, theirP = c ? blackBefore : whiteBefore;
                    const uint64_t ownC = c ? whiteAfter : blackAfter, theirC = c ? blackAfter : whiteAfter;
                    const bool ownSideP = (ownP >> lane) & 1, ownSideC = (ownC >> lane) & 1;
                    const bool pawnP = ownSideP || ((theirP >> lane) & 1), pawnC = ownSideC || ((theirC >> lane) & 1);
                    const uint64_t partP = pawnPartners(pawnP, ownSideP, lane, ownP, theirP);
                    const uint64_t partC = pawnPartners(pawnC, ownSideC, lane, ownC, theirC);
                    const uint64_t unchanged = (ownP & ownC) | (theirP & theirC);
                    const bool same = pawnP && pawnC && ownSideP == ownSideC;
                    const uint64_t kept = same ? (partP & partC & unchanged) : 0;
                    nSub = emitPawnPairRows(sSub[wave], nSub, partP & ~kept, ppId(int(lane) ^ x, !ownSideP), ownP, x);
                    nAdd = emitPawnPairRows(sThr[wave], nAdd, partC & ~kept, ppId(int(lane) ^ x, !ownSideC), ownC, x);
                }
                __builtin_amdgcn_wave_barrier();
                applyDelta(p.t, p.arena, parentSlot, c, lane, sPsqDelta[wave][0], nPsqSub, sPsqDelta[wave][1], nPsqAdd,
                           sThr[wave], nAdd, sSub[wave], nSub, acc);
            }
            storeAcc(p.arena, childSlot, c, lane, acc);
            if (p.ftOut) {
                const uint32_t half = (c == childStm) ? 0u : 1u;
                *reinterpret_cast<u32x2*>(p.ftOut + size_t(it) * kL1 + half * kPairs + 8 * lane) = activate(acc);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (lane < 8 && cOnly == 0) {
            const uint32_t word = reinterpret_cast<const uint32_t*>(childRec)[lane];
            reinterpret_cast<uint32_t*>(p.slotRecords + size_t(childSlot) * 32)[lane] = word;
            if (p.ftOut) reinterpret_cast<uint32_t*>(p.stagedRecords + size_t(it) * 32)[lane] = word;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Activation from arena slots: evaluateNetwork's front half (nnue_state.cpp:396-438 + multilayer.h:92-152) for
// already-materialised accumulators. One wavefront per slot; also stages the slot's record for the MLP's bucket sort.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spx_slot_act_kernel(SlotActParams p) {
    const uint32_t lane = laneId();
    const uint32_t wave = threadIdx.x >> 6;
    for (uint32_t i = blockIdx.x * 4 + wave; i < p.nSlots; i += gridDim.x * 4) {
        const uint32_t slot = __builtin_amdgcn_readfirstlane(p.slots[i]);
        const uint8_t* rec = p.slotRecords + size_t(slot) * 32;
        const int stm = (rec[24] & 0x80) ? 0 : 1;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            uint32_t acc[8];
            loadAcc(p.arena, slot, c, lane, acc);
            const uint32_t half = (c == stm) ? 0u : 1u;
            *reinterpret_cast<u32x2*>(p.ftOut + size_t(i) * kL1 + half * kPairs + 8 * lane) = activate(acc);
        }
        if (lane < 8) {
            reinterpret_cast<uint32_t*>(p.stagedRecords + size_t(i) * 32)[lane] =
                reinterpret_cast<const uint32_t*>(rec)[lane];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Post-processing of raw evals: eval::adjustStatic (eval.cpp:24-27) and eval::adjustEval (eval.cpp:30-67) - one thread
// per position, in place. Piece counts, side to move and the halfmove clock come from the 32-byte record
// (marlinformat.h:32-84: nibbles at +8, stm bit 7 of byte 24, halfmove byte 25). i32 arithmetic, wrapping where the
// reference's would overflow; '/' truncates toward zero as in C++.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t clampScore(int32_t v) {
    return min(max(v, -(kScoreWin - 1)), kScoreWin - 1);
}
__device__ __forceinline__ int32_t wrapMul(int32_t a, int32_t b) {
    return int32_t(uint32_t(a) * uint32_t(b));
}
__device__ __forceinline__ int32_t wrapAdd(int32_t a, int32_t b) {
    return int32_t(uint32_t(a) + uint32_t(b));
}

__global__ __launch_bounds__(256) void spx_adjust_kernel(AdjustParams p) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.nPositions) return;
    const uint64_t* rec = p.positions + size_t(i) * 4;
    const uint64_t occ = rec[0], nibLo = rec[1], nibHi = rec[2];
    const uint32_t tail = uint32_t(rec[3]);  // byte 24 = stm | ep, byte 25 = halfmove clock
    const int stm = (tail & 0x80u) ? 0 : 1;
    const int32_t halfmove = int32_t((tail >> 8) & 0xFFu);
    int32_t eval = p.evals[i];
    if (p.stages & 1u) {
        eval = clampScore(wrapAdd(eval, p.contempt[stm]));
    }
    const uint32_t count = min(uint32_t(popc64(occ)), 32u);
    if (p.stages & 2u) {
        int32_t npMaterial = 0;
        for (uint32_t k = 0; k < count; ++k) {
            const int type = nibbleToPiece(int(((k < 16 ? nibLo : nibHi) >> ((k & 15) * 4)) & 0xF)) >> 1;
            if (type < 5) npMaterial += p.scalingValue[type];
        }
        const int32_t scaled = wrapMul(eval, wrapAdd(p.materialScalingBase, npMaterial));
        const int32_t optimism =
            wrapMul(p.optimism[stm], wrapAdd(p.optimismBase, wrapMul(npMaterial, p.optimismMaterialScale) / 1024));
        eval = wrapAdd(scaled, optimism) / 32768;
        eval = wrapMul(eval, 200 - halfmove) / 200;
        if (p.corrections) {
            eval = wrapAdd(eval, p.corrections[i] / 2048);
        }
        eval = clampScore(eval);
    }
    if (p.stages & 12u) {  // SPX_ADJUST_WHITE_POV, SPX_ADJUST_WDL: what runDatagenSearch returns (search.cpp:237-238)
        if ((p.stages & 4u) && stm == 0) eval = int32_t(0u - uint32_t(eval));
        if (p.stages & 8u) {
            int32_t material = 0;
            for (uint32_t k = 0; k < count; ++k) {
                material += classicalMaterialOfNibble(int(((k < 16 ? nibLo : nibHi) >> ((k & 15) * 4)) & 0xF));
            }
            eval = wdlNormalize(eval, material);
        }
    }
    p.evals[i] = eval;
}

// ---------------------------------------------------------------------------------------------------------------------
// Counting sorts (2 tiny kernels, both keys in one pass). Purely locality / tiling optimisations - results are written
// by perspective id / position id, so any permutation gives identical output.
//   perspectives by piece-square KING BUCKET (16 keys, arch.h:53-65): all perspectives of one bucket gather from the
//     same 1.4 MiB slab of the piece-square table (L2-resident per XCD);
//   positions by OUTPUT BUCKET (8 keys, output.h:51-54): every 16-position MFMA tile of the MLP kernel shares one
//     set of L1/L2/L3 weights.
// hist layout: see the constants below
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kKingKeys = 16;
constexpr int kPairKeys = 256;  // capacity of the first-key histogram (16 king keys in use)
constexpr int kOutKeys = 8;
// hist layout (kHistWords u32 words per buffer): [0, 256) first-key counts (16 king keys or 256 pair keys),
// [256, 264) output-bucket counts, [512, 768) first-key cursors, [768, 776) output-bucket cursors
constexpr int kCursorKing = 512, kCursorOut = 768;  // (kHistOut = 256: spx_kernels.h)

__global__ __launch_bounds__(256) void spx_sort_hist_kernel(SortParams p) {
#if SPX_CORUNNER_PRIO
    __builtin_amdgcn_s_setprio(SPX_CORUNNER_PRIO);  // (A/B: the kernels that run beside the gather ask for issue priority)
#endif
    __shared__ uint32_t sHist[kPairKeys + kOutKeys];
    for (uint32_t i = threadIdx.x; i < kPairKeys + kOutKeys; i += blockDim.x) sHist[i] = 0;
    __syncthreads();
    const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nPositions = p.nPositionsPtr ? min(*p.nPositionsPtr, p.nPositions) : p.nPositions;
    if (pos < nPositions && p.outOnly) {  // arena paths: only the MLP's output-bucket order is needed
        const uint32_t outKey = min((uint32_t(popc64(p.positions[size_t(pos) * 4])) - 2u) / 4u, uint32_t(kOutKeys - 1));
        p.outKeys[pos] = uint8_t(outKey);
        atomicAdd(&sHist[kPairKeys + outKey], 1u);
    } else if (pos < nPositions) {
        const uint64_t* rec = p.positions + size_t(pos) * 4;
        uint64_t occ = rec[0];
        const uint64_t nibLo = rec[1], nibHi = rec[2];
        const uint32_t outKey = min((uint32_t(popc64(occ)) - 2u) / 4u, uint32_t(kOutKeys - 1));  // MaterialCount<8>
        int kingSq[2] = {0, 0};
        uint32_t idx = 0;
        while (occ) {
            const int sq = ctz64(occ);
            occ &= occ - 1;
            const uint32_t nib = uint32_t(((idx < 16 ? nibLo : nibHi) >> ((idx & 15) * 4)) & 0xF);
            ++idx;
            if ((nib & 7) == 5) kingSq[(nib & 8) ? 0 : 1] = sq;
        }
        const uint32_t keyB = uint32_t(kingBucket(kingSq[0] ^ 56)), keyW = uint32_t(kingBucket(kingSq[1]));
        p.kingKeys[2 * pos] = uint8_t(keyB);
        p.kingKeys[2 * pos + 1] = uint8_t(keyW);
        atomicAdd(&sHist[keyB], 1u);
        atomicAdd(&sHist[keyW], 1u);
        p.outKeys[pos] = uint8_t(outKey);
        atomicAdd(&sHist[kPairKeys + outKey], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kPairKeys + kOutKeys; i += blockDim.x) {
        if (sHist[i]) atomicAdd(&p.hist[i], sHist[i]);  // (i >= 256: the output-bucket counts sit at kHistOut = 256)
    }
}

// blocks [0, nb) scatter the first key (perspectives by king key); blocks [nb, nb + nb2) scatter positions by output key
__global__ __launch_bounds__(256) void spx_sort_scatter_kernel(SortParams p, uint32_t firstBlocks) {
#if SPX_CORUNNER_PRIO
    __builtin_amdgcn_s_setprio(SPX_CORUNNER_PRIO);  // (A/B: the kernels that run beside the gather ask for issue priority)
#endif
    __shared__ uint32_t sCount[kPairKeys];
    __shared__ uint32_t sBase[kPairKeys];
    sCount[threadIdx.x] = 0;  // blockDim.x == kPairKeys
    const bool first = blockIdx.x < firstBlocks;
    const uint32_t id = (first ? blockIdx.x : blockIdx.x - firstBlocks) * blockDim.x + threadIdx.x;
    const uint32_t nPositions = p.nPositionsPtr ? min(*p.nPositionsPtr, p.nPositions) : p.nPositions;
    const uint32_t count = first ? nPositions * 2 : nPositions;
    const uint32_t nKeys = first ? kKingKeys : kOutKeys;
    const uint32_t histOff = first ? 0 : kHistOut, cursorOff = first ? kCursorKing : kCursorOut;
    sBase[threadIdx.x] = threadIdx.x < nKeys ? p.hist[histOff + threadIdx.x] : 0u;  // counts, turned into bases below
    __syncthreads();
    uint32_t key = 0, rank = 0;
    if (id < count) {
        key = first ? p.kingKeys[id] : p.outKeys[id];
        rank = atomicAdd(&sCount[key], 1u);
    }
    uint32_t prefix = 0;
    for (uint32_t k = 0; k < threadIdx.x && k < nKeys; ++k) prefix += sBase[k];
    __syncthreads();
    if (threadIdx.x < nKeys) {
        const uint32_t mine = sCount[threadIdx.x];
        sBase[threadIdx.x] = prefix + (mine ? atomicAdd(&p.hist[cursorOff + threadIdx.x], mine) : 0u);
    }
    __syncthreads();
    if (id < count) (first ? p.perspOrder : p.posOrder)[sBase[key] + rank] = id;
    if (blockIdx.x == 0) {
        for (uint32_t i = threadIdx.x; i < uint32_t(kHistWords); i += blockDim.x) p.histNext[i] = 0;
    }
}

// Small batches (<= kSmallSortMax positions, e.g. the one-position drop-in call): the whole two-key counting sort in
// ONE workgroup and one launch - at these sizes the three-launch version is pure launch latency.
constexpr uint32_t kSmallSortMax = 1024;  // measured: one workgroup beats three launches only up to ~1K positions

__global__ __launch_bounds__(1024) void spx_sort_small_kernel(SortParams p) {
    __shared__ uint32_t sHist[kKingKeys + kOutKeys];
    __shared__ uint32_t sBase[kKingKeys + kOutKeys];
    __shared__ uint32_t sCursor[kKingKeys + kOutKeys];
    __shared__ uint8_t sKing[2 * kSmallSortMax];
    __shared__ uint8_t sOut[kSmallSortMax];
    if (threadIdx.x < kKingKeys + kOutKeys) {
        sHist[threadIdx.x] = 0;
        sCursor[threadIdx.x] = 0;
    }
    __syncthreads();
    for (uint32_t pos = threadIdx.x; pos < p.nPositions; pos += blockDim.x) {
        const uint64_t* rec = p.positions + size_t(pos) * 4;
        uint64_t occ = rec[0];
        const uint64_t nibLo = rec[1], nibHi = rec[2];
        const uint32_t outKey = min((uint32_t(popc64(occ)) - 2u) / 4u, uint32_t(kOutKeys - 1));
        int kingSq[2] = {0, 0};
        uint32_t idx = 0;
        while (occ) {
            const int sq = ctz64(occ);
            occ &= occ - 1;
            const uint32_t nib = uint32_t(((idx < 16 ? nibLo : nibHi) >> ((idx & 15) * 4)) & 0xF);
            ++idx;
            if ((nib & 7) == 5) kingSq[(nib & 8) ? 0 : 1] = sq;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const uint32_t key = uint32_t(kingBucket(c == 0 ? (kingSq[c] ^ 56) : kingSq[c]));
            sKing[2 * pos + c] = uint8_t(key);
            atomicAdd(&sHist[key], 1u);
        }
        sOut[pos] = uint8_t(outKey);
        atomicAdd(&sHist[kKingKeys + outKey], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kKingKeys + kOutKeys) {
        const uint32_t first = threadIdx.x < kKingKeys ? 0 : kKingKeys;
        uint32_t prefix = 0;
        for (uint32_t k = first; k < threadIdx.x; ++k) prefix += sHist[k];
        sBase[threadIdx.x] = prefix;
        // the MLP kernel maps tiles from the output-bucket counts (at kHistOut in the global layout)
        p.hist[threadIdx.x < kKingKeys ? threadIdx.x : kHistOut + (threadIdx.x - kKingKeys)] = sHist[threadIdx.x];
    }
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < 2 * p.nPositions; q += blockDim.x) {
        const uint32_t key = sKing[q];
        p.perspOrder[sBase[key] + atomicAdd(&sCursor[key], 1u)] = q;
    }
    for (uint32_t pos = threadIdx.x; pos < p.nPositions; pos += blockDim.x) {
        const uint32_t key = kKingKeys + sOut[pos];
        p.posOrder[sBase[key] + atomicAdd(&sCursor[key], 1u)] = pos;
    }
}

// p.hist must be all-zero on entry of the multi-launch path; its scatter kernel clears p.histNext (the buffer the NEXT
// large sort will use), so no memset launch is ever needed (spx_api alternates two buffers; small sorts use a third).
hipError_t launchSort(const SortParams& p, hipStream_t stream) {
    if (p.nPositions <= kSmallSortMax && !p.nPositionsPtr) {
        hipLaunchKernelGGL(spx_sort_small_kernel, dim3(1), dim3(1024), 0, stream, p);
        return hipGetLastError();
    }
    const uint32_t b1 = (p.nPositions + 255) / 256, b2 = p.outOnly ? 0u : (2 * p.nPositions + 255) / 256;
    hipLaunchKernelGGL(spx_sort_hist_kernel, dim3(b1), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(spx_sort_scatter_kernel, dim3(b2 + b1), dim3(256), 0, stream, p, b2);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// MLP kernel. One wavefront per TILE of 16 positions that share an output bucket (positions arrive sorted by bucket,
// tiles never straddle buckets):
//   L1   D[16 pos][32] = A[16][1024] (u8 activations, <= 127 so u8 == i8) x B[1024][32] (i8) on
//        v_mfma_i32_16x16x64_i8: 16 k-steps x 2 n-tiles = 32 MFMAs, exact i32 sums (multilayer.h:154-217)
//   tail per position, lane = output neuron: shift/bias/dual activation (multilayer.h:219-257), L2 64x64 in wrapping
//        i32 (multilayer.h:261-343) with this lane's 64 weights held in registers for the whole tile, L3 + skip
//        connection (multilayer.h:345-447) reduced across the wave, i64 scale with truncating division (:484-489).
// kSmallL2W: every |l2W| < 2^23 (checked on the host at context creation), so with the L2 inputs always inside
//   (-2^20, 2^12] the product is one full-rate v_mad_i32_i24; otherwise the exact-but-slow v_mul_lo_u32 path runs.
// ---------------------------------------------------------------------------------------------------------------------
// kTiling: kMlpTileSorted = one wavefront per 16-position tile of the bucket-sorted order; kMlpTileShared = the four
// waves of a workgroup share ONE such tile (each repeats the cheap MFMA part and takes every fourth position of the
// serial tail) - for batches too small to fill the chip, where the kernel is bound by the latency of a single tile
// (4 096 positions: 16 us unshared); kMlpTilePerPosition = no sort at all, every position is its own tile and finds
// its bucket from its record - the handful-of-positions drop-in call, where a sort launch costs more than it saves.
template <bool kSmallL2W, int kTiling>
__global__ __launch_bounds__(256, kTiling == kMlpTileSorted ? SPX_MLP_SORTED_WAVES_PER_SIMD : SPX_MLP_WAVES_PER_SIMD) void spx_mlp_kernel(MlpParams p) {
#if SPX_CORUNNER_PRIO
    __builtin_amdgcn_s_setprio(SPX_CORUNNER_PRIO);  // (A/B: the kernels that run beside the gather ask for issue priority)
#endif
    constexpr bool kShareTile = kTiling == kMlpTileShared;
    // Per wave 4 KiB of LDS, used twice: the L1 sums of the wave's tile ([16][33], padded against bank conflicts) sit at its END, the
    // L2 inputs of the tile's positions ([16][64]; broadcast reads; later the L3 terms) grow from its start while the sums are
    // consumed row by row - row k of the inputs ends at byte 256 (k + 1), row k + 1 of the sums starts at 1984 + 132 (k + 1), never
    // below it. (Round 5: 16 instead of 24.3 KiB per workgroup, so that one fits beside the column-sliced gather's 136 KiB.)
    __shared__ __align__(16) int32_t sBuf[4][16 * kL2Full];
    static_assert(kL2Full * 16 - (kL2 + 1) * 16 + (kL2 + 1) * 16 == kL2Full * 16 && (kL2Full - (kL2 + 1)) * 16 >= 0, "the sums fit");
    static_assert(kL2Full * 16 >= (kL2Full - (kL2 + 1)) * 16 + (kL2 + 1) * 16 && kL2Full * 15 <= (kL2Full - (kL2 + 1)) * 16 + (kL2 + 1) * 15,
                  "input row k never reaches sum row k + 1");

    const uint32_t lane = laneId();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t tile = kShareTile ? blockIdx.x : blockIdx.x * 4 + wave;
    int32_t (*const sSumW)[kL2 + 1] = reinterpret_cast<int32_t (*)[kL2 + 1]>(sBuf[wave] + 16 * (kL2Full - (kL2 + 1)));
    int32_t (*const sInW)[kL2Full] = reinterpret_cast<int32_t (*)[kL2Full]>(sBuf[wave]);

    // ---- locate this tile: bucket, first sorted index, number of real positions ----
    uint32_t bucket = 0, sortedBase = 0, count = 0;
    if constexpr (kTiling == kMlpTilePerPosition) {
        if (tile < p.nPositions) {
            const uint32_t pieces = uint32_t(popc64(p.records[size_t(tile) * 4]));
            bucket = min((pieces - 2u) / 4u, uint32_t(kOutputBuckets - 1));  // MaterialCount<8> (output.h:44-55)
            sortedBase = tile;
            count = 1;
        }
    } else {
        uint32_t tileStart = 0, posStart = 0;
        bool found = false;
#pragma unroll
        for (uint32_t b = 0; b < kOutputBuckets; ++b) {
            const uint32_t cnt = p.hist[kHistOut + b];
            const uint32_t tiles = (cnt + 15u) >> 4;
            if (!found && tile < tileStart + tiles) {
                found = true;
                bucket = b;
                const uint32_t j = tile - tileStart;
                sortedBase = posStart + j * 16;
                count = min(16u, cnt - j * 16);
            }
            tileStart += tiles;
            posStart += cnt;
        }
        if (!found) {
            count = 0;  // grid is sized for the worst-case number of tiles: nothing to do for this wave
        }
    }
    if (count == 0) {
        return;
    }
    bucket = __builtin_amdgcn_readfirstlane(bucket);
    sortedBase = __builtin_amdgcn_readfirstlane(sortedBase);
    count = __builtin_amdgcn_readfirstlane(count);

    // ---- L1 on MFMA ----
    const uint32_t rowInTile = lane & 15, kGroup = lane >> 4;
    const uint32_t myPos = kTiling == kMlpTilePerPosition
                               ? sortedBase
                               : p.posOrder[sortedBase + min(rowInTile, count - 1)];  // rows past `count` replicate the last
    const uint8_t* aRow = p.ftOut + size_t(myPos) * kL1 + kGroup * 16;
    const int8_t* bBase = p.l1W + size_t(bucket) * (kL1 * kL2) + lane * 16;
    i32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const i32x4 a = *reinterpret_cast<const i32x4*>(aRow + ks * 64);
        // device layout [bucket][kstep][ntile][lane][16 B]: one coalesced 1 KiB wave load per B fragment
        const i32x4 w0 = *reinterpret_cast<const i32x4*>(bBase + (ks * 2 + 0) * 1024);
        const i32x4 w1 = *reinterpret_cast<const i32x4*>(bBase + (ks * 2 + 1) * 1024);
        acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, w0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, w1, acc1, 0, 0, 0);
    }
    // C/D layout of the 16x16 MFMA: lane holds column (lane & 15), rows (lane >> 4) * 4 + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sSumW[kGroup * 4 + r][rowInTile] = acc0[r];
        sSumW[kGroup * 4 + r][16 + rowInTile] = acc1[r];
    }

    // ---- per-lane constants of this bucket: lane = L2/L3 output neuron o, L1 neuron = lane & 31 ----
    const uint32_t o1 = lane & 31;
    const int32_t l1Bias = p.l1B[bucket * kL2 + o1];
    const int32_t l2Bias = p.l2B[bucket * kL3 + lane];
    const int32_t l3Weight = p.l3W[bucket * kL3 + lane];
    const int32_t l3Bias = p.l3B[bucket];
    // L2 weights of this lane's output: [bucket][input quartet][o][4] (relayoutL2), one 16-byte load per four inputs. The small
    // tilings hold all 64 in registers; the big-batch tiling STREAMS them, a quartet ahead of its use (round 6: 16 loads per tile
    // instead of 64 four-byte ones, 64 registers less; alone 37.6 -> 28 us per 65 536 positions, and the gather this kernel runs
    // beside 327 -> 305 us). More resident waves do NOT help it (5 / 6 waves per SIMD: 31.7 / 36.5 us): DESIGN.md 4.9c.
    constexpr bool kStreamW2 = kTiling == kMlpTileSorted;
    const i32x4* const w2q = reinterpret_cast<const i32x4*>(p.l2W) + size_t(bucket) * (kL2Full / 4) * kL3 + lane;
    i32x4 w2[kStreamW2 ? 1 : kL2Full / 4];
    i32x4 w2Next = w2q[0];
    if constexpr (!kStreamW2) {
        w2[0] = w2Next;
#pragma unroll
        for (int i = 1; i < int(kL2Full / 4); ++i) w2[i] = w2q[i * kL3];
    }
    __builtin_amdgcn_wave_barrier();

    // The tile's positions move through the tail TOGETHER, layer by layer (round 1 took them one at a time: per position a
    // store -> barrier -> 16 broadcast reads -> 64-long multiply-add -> 6-step wave reduction, all latency, 16 times):
    //   A  lane = L1 output o: dual activation of every position's sum; L2 inputs to LDS
    //   B  lane = L2 output o: kRows independent accumulators over the 64 inputs (weights in registers, inputs broadcast)
    //   C  lane = L3 input o:  (clamp(l2) + l1o) * W3 per position to LDS; four lanes per position add 16 terms each
    constexpr int kRows = kShareTile ? 4 : (kTiling == kMlpTilePerPosition ? 1 : 16);  // positions this wave finishes
    const uint32_t rowBegin = kShareTile ? wave : 0u, rowStep = kShareTile ? 4u : 1u;
    int32_t mine[kRows];
#pragma unroll
    for (int k = 0; k < kRows; ++k) {
        const uint32_t r = rowBegin + uint32_t(k) * rowStep;
        mine[k] = 0;
        if (r < count) {
            const int32_t s = sSumW[r][o1];
            const uint32_t t = uint32_t(s >> kL1Shift) + uint32_t(l1Bias);  // wraps
            const int32_t ts = int32_t(t);
            const int32_t c0 = min(max(ts, 0), 4096) << kQBits;  // CReLU side, pre-shifted for the skip connection
            const int32_t sq = int32_t(t * t);                    // mullo wraps BEFORE the signed min
            const int32_t c1 = min(sq, 1 << 24) >> kQBits;        // SCReLU side
            mine[k] = lane < kL2 ? c0 : c1;                       // l1o[lane] = [CReLU(32) | SCReLU(32)]
            sInW[k][lane] = mine[k] >> kQBits;               // L2 input (multilayer.h:281-283), in (-2^20, 2^12]
        }
    }
    __builtin_amdgcn_wave_barrier();
    // L2: l2[o] = bias + sum_i in[i] * W2[b][i][o], wrapping i32
    uint32_t acc2[kRows];
#pragma unroll
    for (int k = 0; k < kRows; ++k) acc2[k] = uint32_t(l2Bias);
    auto mac = [&](uint32_t& acc, int32_t in, int32_t w) {
        if constexpr (kSmallL2W) {
            // ONE v_mad_i32_i24 (left to itself the compiler multiplies twice and folds two products per v_add3_u32: 6 instructions
            // per 4 terms instead of 4)
            asm("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(acc) : "v"(in), "v"(w));
        } else {
            acc += uint32_t(in) * uint32_t(w);
        }
    };
    if constexpr (kStreamW2) {
        // (the rows past `count` of a partial tile are taken along - whatever their LDS words hold, the sums wrap and are never
        // stored -: sixteen wave-uniform branches inside the loop cost it 100 bytes of spills. Four rows' inputs are asked for before
        // the first is used: left to the compiler every broadcast read was waited for on the spot.)
        static_assert(kRows % 4 == 0 || !kStreamW2, "rows in fours");
#pragma unroll 1
        for (int i4 = 0; i4 < int(kL2Full / 4); ++i4) {
            const i32x4 w = w2Next;
            w2Next = w2q[min(i4 + 1, int(kL2Full / 4) - 1) * int(kL3)];
            i32x4 in4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) in4[j] = *reinterpret_cast<const i32x4*>(&sInW[j][4 * i4]);
#pragma unroll
            for (int kb = 0; kb < kRows; kb += 4) {
                i32x4 cur[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) cur[j] = in4[j];
                if (kb + 4 < kRows) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) in4[j] = *reinterpret_cast<const i32x4*>(&sInW[(kb + 4 + j) % kRows][4 * i4]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) mac(acc2[(kb + j) % kRows], cur[j][t], w[t]);
                }
            }
        }
    } else {
#pragma unroll
        for (int i4 = 0; i4 < int(kL2Full / 4); ++i4) {
#pragma unroll
            for (int k = 0; k < kRows; ++k) {
                if (rowBegin + uint32_t(k) * rowStep < count) {  // wave-uniform
                    const i32x4 in4 = *reinterpret_cast<const i32x4*>(&sInW[k][4 * i4]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) mac(acc2[k], in4[j], w2[i4][j]);
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();  // every lane is done reading the inputs: the buffer now takes the L3 terms
    // L3 with skip connection: (clamp(l2, 0, Q^3) + l1o) * W3, wrapping
#pragma unroll
    for (int k = 0; k < kRows; ++k) {
        const int32_t l2v = min(max(int32_t(acc2[k]), 0), 262144);
        sInW[k][lane] = int32_t((uint32_t(l2v) + uint32_t(mine[k])) * uint32_t(l3Weight));
    }
    __builtin_amdgcn_wave_barrier();
    {
        const uint32_t k = lane >> 2, q = lane & 3;  // four lanes per position, 16 terms each
        uint32_t sum = 0;
        if (k < uint32_t(kRows)) {
#pragma unroll
            for (int j = 0; j < 16; j += 4) {
                const i32x4 t4 = *reinterpret_cast<const i32x4*>(&sInW[k][q * 16 + j]);
                sum += uint32_t(t4[0]) + uint32_t(t4[1]) + uint32_t(t4[2]) + uint32_t(t4[3]);
            }
        }
        sum += uint32_t(__shfl_xor(int32_t(sum), 1, 64));
        sum += uint32_t(__shfl_xor(int32_t(sum), 2, 64));
        const uint32_t r = rowBegin + k * rowStep;
        if (q == 0 && k < uint32_t(kRows) && r < count) {
            const int32_t l3 = int32_t(sum + uint32_t(l3Bias));
            const int64_t scaled = int64_t(l3) * kScale / (int64_t(1) << (4 * kQBits));  // truncating division
            p.out[kTiling == kMlpTilePerPosition ? sortedBase : p.posOrder[sortedBase + r]] = int32_t(scaled);
        }
    }
}


hipError_t launchFtTeam(const FtParams& p, uint32_t gridBlocks, hipStream_t stream) {
    if (p.t.outlierTab) {
        hipLaunchKernelGGL((spx_ft_team_kernel<true>), dim3(gridBlocks), dim3(64 * kWavesPerBlock), 0, stream, p);
    } else {
        hipLaunchKernelGGL((spx_ft_team_kernel<false>), dim3(gridBlocks), dim3(64 * kWavesPerBlock), 0, stream, p);
    }
    return hipGetLastError();
}

hipError_t launchFt(const FtParams& p, uint32_t gridBlocks, hipStream_t stream) {
    const dim3 grid(gridBlocks), block(64 * kWavesPerBlock);
    if (p.t.outlierTab) {
        hipLaunchKernelGGL((spx_ft_kernel<true>), grid, block, 0, stream, p);
    } else {
        hipLaunchKernelGGL((spx_ft_kernel<false>), grid, block, 0, stream, p);
    }
    return hipGetLastError();
}

hipError_t launchUpdate(const UpdateParams& p, uint32_t gridBlocks, bool splitPerspectives, bool streamAccumulators,
                        hipStream_t stream) {
    const dim3 grid(gridBlocks), block(64 * kWavesPerBlock);
    if (splitPerspectives && streamAccumulators) {
        hipLaunchKernelGGL((spx_update_kernel<true, true>), grid, block, 0, stream, p);
    } else if (splitPerspectives) {
        hipLaunchKernelGGL((spx_update_kernel<true, false>), grid, block, 0, stream, p);
    } else if (streamAccumulators) {
        hipLaunchKernelGGL((spx_update_kernel<false, true>), grid, block, 0, stream, p);
    } else {
        hipLaunchKernelGGL((spx_update_kernel<false, false>), grid, block, 0, stream, p);
    }
    return hipGetLastError();
}

hipError_t launchUpdateChain(const ChainParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(spx_update_chain_kernel, dim3((2 * p.nChains + kWavesPerBlock - 1) / kWavesPerBlock), dim3(64 * kWavesPerBlock),
                       0, stream, p);
    return hipGetLastError();
}

hipError_t launchUpdateObserved(const UpdateParams& p, uint32_t gridBlocks, hipStream_t stream) {
    hipLaunchKernelGGL(spx_update_observed_kernel, dim3(gridBlocks), dim3(64 * kWavesPerBlock), 0, stream, p);
    return hipGetLastError();
}

hipError_t launchAdjust(const AdjustParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(spx_adjust_kernel, dim3((p.nPositions + 255) / 256), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launchSlotAct(const SlotActParams& p, uint32_t gridBlocks, hipStream_t stream) {
    hipLaunchKernelGGL(spx_slot_act_kernel, dim3(gridBlocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

template <bool kSmallL2W>
static void launchMlpTiling(const MlpParams& p, MlpTiling tiling, hipStream_t stream) {
    const uint32_t tiles = (p.nPositions + 15) / 16 + kOutputBuckets;  // worst case: every bucket ends in a partial tile
    if (tiling == kMlpTilePerPosition) {
        hipLaunchKernelGGL((spx_mlp_kernel<kSmallL2W, kMlpTilePerPosition>), dim3((p.nPositions + 3) / 4), dim3(256), 0,
                           stream, p);
    } else if (tiling == kMlpTileShared) {
        hipLaunchKernelGGL((spx_mlp_kernel<kSmallL2W, kMlpTileShared>), dim3(tiles), dim3(256), 0, stream, p);
    } else {
        hipLaunchKernelGGL((spx_mlp_kernel<kSmallL2W, kMlpTileSorted>), dim3((tiles + 3) / 4), dim3(256), 0, stream, p);
    }
}

#ifndef SPX_MEASURE_SKIP
#define SPX_MEASURE_SKIP 0  // (measurement builds, spx_ftx.hip: bit 2 of it = no MLP after the first 12 launches)
#endif

hipError_t launchMlp(const MlpParams& p, bool smallL2Weights, MlpTiling tiling, hipStream_t stream) {
    if (SPX_MEASURE_SKIP & 4) {
        static std::atomic<uint32_t> calls{0};
        if (calls.fetch_add(1) >= 12) return hipSuccess;
    }
    if (smallL2Weights) {
        launchMlpTiling<true>(p, tiling, stream);
    } else {
        launchMlpTiling<false>(p, tiling, stream);
    }
    return hipGetLastError();
}

uint32_t ftWavesPerBlock() {
    return kWavesPerBlock;
}

}  // namespace spx
