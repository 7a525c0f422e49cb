// Wave-level building blocks of the gfx950 feature-transformer kernels (wave64; one wavefront = one board, lane = square
// during extraction, lane = 16 accumulator columns during accumulation). Shared by spx_kernels.hip (the product kernels)
// and spx_probe.hip (the load-only gather-ceiling probe), so both run exactly the same list construction.
//
// Reference semantics (paths relative to /root/reference/src/eval):
//   nnue_state.cpp:440-449 resetPsqAccumulator; :309-354 addThreatFeatures; :89-145 applyThreatRows (i8 -> i16 widening);
//   nnue/input.h:72-75,283-293 Accumulator::initBoth/add; nnue/arch/multilayer.h:92-152 activateFt.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "spx_device_math.h"
#include "spx_kernels.h"

namespace spx {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kWavesPerBlock = 4;
#ifndef SPX_FT_CHUNK
#define SPX_FT_CHUNK 128  // perspectives per round-robin chunk of the XCD traversal, a power of two
#endif
#ifndef SPX_FT_WAVES_PER_SIMD
#define SPX_FT_WAVES_PER_SIMD 5  // launch_bounds 2nd arg = min waves per SIMD. A/B on MI355X: 4 -> 0.557 ms, 5 (96 VGPRs, no spill) -> 0.548, 6 (spills) -> 0.663
#endif
#ifndef SPX_UPDATE_SPLIT_WAVES
#define SPX_UPDATE_SPLIT_WAVES 4
#endif
#ifndef SPX_MLP_WAVES_PER_SIMD
#define SPX_MLP_WAVES_PER_SIMD 4  // A/B on MI355X with the batched tail: 3 -> 39.0 us, 4 -> 35.5 us per 65 536 positions (round-1 tail: 38.3)
#endif
#ifndef SPX_MLP_SORTED_WAVES_PER_SIMD
#define SPX_MLP_SORTED_WAVES_PER_SIMD 4  // the big-batch tiling (L2 weights streamed, not held). A/B on MI355X, us per 65 536 positions alone: 4 -> 28.0, 5 -> 31.7, 6 -> 36.5 (round 5, weights held: 37.6)
#endif
constexpr int kThreatCap = 256;  // StaticVector<u16, 256> in addThreatFeatures (nnue_state.cpp:315)
constexpr int kPsqCap = 32;
constexpr int kU8Cap = kThreatCap + kPsqCap;  // u8-row list: compact piece-square rows first, then <= 256 threat rows
constexpr int kDeltaCap = 96;  // rows per delta list of the update kernel; a legal move stays far below (<= 64 threat rows
                               // per board for two changed squares, + pawn pairs); a list that would overflow is rebuilt

__device__ __forceinline__ uint32_t laneId() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ uint32_t prefixCount(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}
// inclusive prefix sum over the wave's 64 lanes, on the DPP cross-lane path (no LDS): row_shr 1, 2, 4, 8 scan every row of
// 16 lanes, row_bcast:15 / row_bcast:31 carry the row totals on (rows 1 and 3, then rows 2 and 3)
__device__ __forceinline__ uint32_t waveInclusiveScan(uint32_t v) {
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xF, 0xF, false));
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xF, 0xF, false));
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xF, 0xF, false));
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xF, 0xF, false));
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xA, 0xF, false));
    v += uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xC, 0xF, false));
    return v;
}
__device__ __forceinline__ uint32_t pkAdd16(uint32_t a, uint32_t b) {
    // two independent wrapping 16-bit adds (v_pk_add_u16): exactly the reference's add_epi16 semantics
    const u16x2 r = __builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pkSub16(uint32_t a, uint32_t b) {
    const u16x2 r = __builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b);
    return __builtin_bit_cast(uint32_t, r);
}
// Row fetch: every row load of the gather is `table + wave-uniform row offset + this lane's 16 * lane`. As a BUFFER load the
// three terms map onto the instruction itself - resource descriptor (table base, SGPRs, built once), soffset (the row
// offset straight from v_readfirstlane), voffset (the lane's constant) - so a row costs no address arithmetic at all;
// the flat-global form pays a 64-bit v_lshl_add_u64 (or s_add_u32 + s_addc_u32) per load. Out-of-range offsets (malformed
// lists cannot produce them; belt and braces) read zeros instead of faulting. Used by the UPDATE kernel (+5 % there);
// the full-refresh gather keeps global loads (see gatherFull).
struct RowTable {
    __amdgpu_buffer_rsrc_t rsrc;
};
__device__ __forceinline__ RowTable makeRowTable(const void* base, uint32_t bytes) {
    RowTable t;
    t.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, int(bytes), 0x00020000);  // raw buffer, gfx9 DATA_FORMAT_32
    return t;
}
__device__ __forceinline__ u32x4 loadRow16(const RowTable& t, uint32_t rowOffset, uint32_t laneOffset) {
    return __builtin_amdgcn_raw_buffer_load_b128(t.rsrc, int(laneOffset), int(rowOffset), 0);
}
constexpr uint32_t kU8TableBytes = (kThreatRows + kPsqRows) * kL1;  // threat rows + the compact piece-square slots

// The 16 bytes a lane holds of a u8 row are widened into two accumulator words per dword. Storage order of the table
// (relayoutThreatRow, spx_api.cpp): within a dword the bytes are columns (c, c + 2, c + 1, c + 3) - i.e. the EVEN bytes
// 0, 2 are one packed-i16 accumulator word (columns c, c + 1) and the ODD bytes 1, 3 the next one (c + 2, c + 3). So:
//   unpackEven: bytes 0, 2 -> 16-bit fields = x & 0x00FF00FF      (v_and_b32: 2 SIMD cycles, tools/probes/valu_rate_probe)
//   unpackOdd:  bytes 1, 3 -> 16-bit fields = v_perm_b32          (4 cycles; a shift + and would be 6)
// Round 1 stored (c, c + 1, c + 2, c + 3) and paid two v_perm_b32 per dword.
__device__ __forceinline__ uint32_t unpackLo(uint32_t x) {
    return x & 0x00FF00FFu;
}
__device__ __forceinline__ uint32_t unpackHi(uint32_t x) {
    return __builtin_amdgcn_perm(0u, x, 0x0C030C01u);
}



// ---------------------------------------------------------------------------------------------------------------------
// Shared wave-level building blocks (one wavefront = one board, lane = square).
// ---------------------------------------------------------------------------------------------------------------------
struct LaneBoard {
    uint64_t occ, kingsBb, whiteBb, pawnsBb;  // wave-uniform bitboards
    int piece;                                // this lane's piece (type<<1|colour) or kNoPiece
    int stm;                                  // side to move, 1 = white
};

// the record as decodeBoard wants it: lane l holds dword l & 7 (one coalesced load; a caller that walks a chain of records
// asks for the next one a ply ahead)
__device__ __forceinline__ uint32_t loadRecordWord(const uint8_t* rec, uint32_t lane) {
    return reinterpret_cast<const uint32_t*>(rec)[lane & 7];
}

__device__ __forceinline__ LaneBoard decodeBoardWord(uint32_t w, uint32_t lane) {
    LaneBoard b;
    // ONE coalesced load for the whole 32-byte record (lane l fetches dword l & 7); the wave-uniform fields come out of
    // v_readlane into SGPRs - so every mask derived from the occupancy is scalar arithmetic - and a lane's nibble comes
    // from the lane that holds its dword (ds_bpermute) instead of a second, dependent global load
    b.occ = (uint64_t(uint32_t(__builtin_amdgcn_readlane(int(w), 1))) << 32) | uint32_t(__builtin_amdgcn_readlane(int(w), 0));
    b.stm = (uint32_t(__builtin_amdgcn_readlane(int(w), 6)) & 0x80u) ? 0 : 1;
    const bool occupied = (b.occ >> lane) & 1;
    // malformed records (> 32 pieces) must not index past the 16 nibble bytes: results are unspecified, accesses are not
    const uint32_t nibIdx = min(prefixCount(b.occ), 31u);
    const uint32_t word = uint32_t(__shfl(int(w), int(2 + (nibIdx >> 3)), 64));
    b.piece = occupied ? nibbleToPiece(int((word >> ((nibIdx & 7) * 4)) & 0xF)) : int(kNoPiece);
    const int type = b.piece >> 1;  // 6 for empty
    b.kingsBb = __ballot(type == 5);
    b.whiteBb = __ballot(occupied && (b.piece & 1) == 1);
    // pawn-pair ids are (square - 8): pawns on the back ranks exist only in malformed records and are left out of the
    // pawn-pair features (kPpMasks is empty for those squares anyway, threats.h:109)
    b.pawnsBb = __ballot(type == 0) & 0x00FFFFFFFFFFFF00ull;
    return b;
}

// The same from a record that sits in SGPRs (a wave that walks records by a wave-uniform index fetches them through the SCALAR
// cache - s_load_dwordx8, its own counter: the vector-memory counter is in order, so a record asked for through it cannot be
// waited for without waiting for every store issued since).
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ u32x8 scalarLoadRecord(const void* rec) {  // (rec must be wave-uniform; returns before the data does)
    u32x8 r;
    asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(r) : "s"(rec));
    return r;
}
__device__ __forceinline__ void scalarLoadWait(u32x8& r) {  // (every outstanding scalar / LDS operation of the wave, in fact)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r));
}
__device__ __forceinline__ LaneBoard decodeBoardScalar(const u32x8& r, uint32_t lane) {
    LaneBoard b;
    b.occ = (uint64_t(r[1]) << 32) | r[0];
    b.stm = (r[6] & 0x80u) ? 0 : 1;
    const bool occupied = (b.occ >> lane) & 1;
    const uint32_t nibIdx = min(prefixCount(b.occ), 31u), k = nibIdx >> 3;  // (> 32 pieces: malformed, stay inside the 16 bytes)
    const uint32_t word = k == 0 ? r[2] : (k == 1 ? r[3] : (k == 2 ? r[4] : r[5]));
    b.piece = occupied ? nibbleToPiece(int((word >> ((nibIdx & 7) * 4)) & 0xF)) : int(kNoPiece);
    const int type = b.piece >> 1;  // 6 for empty
    b.kingsBb = __ballot(type == 5);
    b.whiteBb = __ballot(occupied && (b.piece & 1) == 1);
    b.pawnsBb = __ballot(type == 0) & 0x00FFFFFFFFFFFF00ull;  // (see decodeBoardWord)
    return b;
}

__device__ __forceinline__ LaneBoard decodeBoard(const uint8_t* rec, uint32_t lane) {
    return decodeBoardWord(loadRecordWord(rec, lane), lane);
}

// Appends one threat row per set bit of this lane's `targets` (victims popped one per wave iteration):
// attacker = this lane's `piece` on square `lane`, victim piece fetched from the lane that owns the target square.
// Rows the reference excludes (threatFeatureIndex < 0) are dropped. Returns the new list length (capacity kThreatCap).
__device__ __forceinline__ uint32_t emitThreatRows(uint32_t* list, uint32_t n, uint64_t targets, int piece,
                                                   uint32_t lane, int x, int flipColour, const uint32_t* lut,
                                                   const uint64_t* pseudoTab = nullptr) {
    const int pieceRel = piece ^ flipColour;
    const int sqRel = int(lane) ^ x;
    uint64_t pseudoRel = 0;
    if (targets) {  // (only non-king pieces have targets)
        // pseudo-attack set of the attacker in the perspective's frame: one LDS read where the table is staged
        // (pseudo[k][sq], spx_device_math.h), else the per-lane arithmetic (~40 instructions for the union of piece types)
        if (pseudoTab) {
            pseudoRel = pseudoTab[(pieceRel >= 2 ? (pieceRel >> 1) + 1 : pieceRel) * 64 + sqRel];
        } else {
            pseudoRel = piecePseudoAttacks(pieceRel, sqRel);
        }
    }
    while (__ballot(targets != 0)) {
        const bool active = targets != 0;
        const int to = active ? ctz64(targets) : 0;
        targets &= targets - 1;
        const int victim = __shfl(piece, to, 64);
        int32_t row = -1;
        if (active) {
            row = threatRow(lut, pieceRel, sqRel, pseudoRel, victim ^ flipColour, to ^ x);
        }
        const uint64_t valid = __ballot(row >= 0);
        const uint32_t slot = n + prefixCount(valid);
        if (row >= 0 && slot < kThreatCap) {
            list[slot] = uint32_t(row) * kL1;
        }
        n = min(n + uint32_t(popc64(valid)), uint32_t(kThreatCap));
    }
    return n;
}

// Appends one pawn-pair row per set bit of this lane's `partners`; `ownPawns` classifies the partner's side.
__device__ __forceinline__ uint32_t emitPawnPairRows(uint32_t* list, uint32_t n, uint64_t partners, uint32_t idA,
                                                     uint64_t ownPawns, int x) {
    while (__ballot(partners != 0)) {
        const bool active = partners != 0;
        const int b = active ? ctz64(partners) : 0;
        partners &= partners - 1;
        const bool bEnemy = !((ownPawns >> b) & 1);
        const uint64_t valid = __ballot(active);
        const uint32_t slot = n + prefixCount(valid);
        if (active && slot < kThreatCap) {
            list[slot] = ppRow(idA, ppId(b ^ x, bEnemy)) * kL1;
        }
        n = min(n + uint32_t(popc64(valid)), uint32_t(kThreatCap));
    }
    return n;
}

// this lane's pawn-pair partner set for perspective c (nnue_state.cpp:330-351): own pawns pair with own pawns on
// higher squares and with every enemy pawn inside kPpMasks; enemy pawns pair with enemy pawns on higher squares
__device__ __forceinline__ uint64_t pawnPartners(bool isPawn, bool own, uint32_t lane, uint64_t ownPawns,
                                                 uint64_t theirPawns) {
    if (!isPawn) {
        return 0;
    }
    const uint64_t above = ~((2ull << lane) - 1);
    return own ? (((ownPawns & above) | theirPawns) & ppMask(int(lane))) : (theirPawns & above & ppMask(int(lane)));
}

// One piece-square delta row per `active` lane: rows with a compact (u8) copy are appended at the head of `u8List`,
// the others go to `wideList` (i16 table). Capacities 8 each. Returns the number of compact rows; nWide by reference.
__device__ __forceinline__ uint32_t emitPsqDeltaRows(bool active, uint32_t row, const uint32_t* lut, uint32_t* wideList,
                                                     uint32_t* u8List, uint32_t& nWide) {
    const bool compact = active && ((lut[kLutCompactBase + (row >> 5)] >> (row & 31)) & 1u);
    const uint64_t compactMask = __ballot(compact), wideMask = __ballot(active && !compact);
    const uint32_t slot = prefixCount(compact ? compactMask : wideMask);
    if (active && slot < 8) {
        if (compact) {
            u8List[slot] = (kThreatRows + row) * kL1;
        } else {
            wideList[slot] = row * (kL1 * 2);
        }
    }
    nWide = min(uint32_t(popc64(wideMask)), 8u);
    return min(uint32_t(popc64(compactMask)), 8u);
}

// Row lists of one perspective of one board (the full-refresh feature set): psqList (capacity kPsqCap) = byte offsets
// into the i16 piece-square table, thrList (capacity kU8Cap) = byte offsets into the u8 row table; nThr counts both the
// compact piece-square rows and the threat / pawn-pair rows in it.
// this lane's threat targets (perspective independent): the occupied non-king squares its piece attacks
__device__ __forceinline__ uint64_t laneTargets(const LaneBoard& b, uint32_t lane) {
    const int type = b.piece >> 1;
    uint64_t targets = 0;
    if (b.piece != kNoPiece && type != 5) {
        targets = pieceAttacks(b.piece, int(lane), b.occ) & b.occ & ~b.kingsBb;
    }
    return targets;
}


// kNear (full-refresh kernel of a net that has near-compact rows): such rows take the 1 KiB path too; the remainders of
// their <= kOutlierCap wide weights (FtTables::outlierTab) are summed per column into `nearAcc` (this wave's 1 024 i32 in
// LDS; lane = the row's square, one LDS atomic per remainder) and folded in after the gather. Returns whether any was.
template <bool kNear = false>
__device__ __forceinline__ bool buildFullLists(const LaneBoard& b, int c, uint32_t lane, const uint32_t* lut,
                                               uint32_t* psqList, uint32_t* thrList, uint32_t& nPsq, uint32_t& nThr,
                                               const uint64_t* pseudoTab = nullptr,
                                               const uint32_t* outlierTab = nullptr, int32_t* nearAcc = nullptr) {
    const int piece = b.piece;
    const bool occupied = piece != kNoPiece;
    const int type = piece >> 1;
    const uint64_t ownKing = __ballot(piece == (10 | c));
    const int kingSq = ownKing ? ctz64(ownKing) : 0;  // a record without that king is malformed: stay in bounds
    const uint64_t ownPawns = b.pawnsBb & (c ? b.whiteBb : ~b.whiteBb);
    const uint64_t theirPawns = b.pawnsBb & ~ownPawns;
    const int x = perspXor(c, kingSq);
    const int flipColour = (c == 0) ? 1 : 0;
    bool hasNear = false;

    // piece-square rows: one per occupied square (resetPsqAccumulator, nnue_state.cpp:440-449). Rows whose weights all
    // fit i8 have a 1 KiB copy in the u8 table: those go to the head of the u8 list, the rest to the i16 list.
    uint32_t nCompact;
    {
        uint32_t row = 0;
        bool compact = false, near = false;
        if (occupied) {
            row = psqRow(c, piece, int(lane), kingSq);
            compact = (lut[kLutCompactBase + (row >> 5)] >> (row & 31)) & 1u;
            if constexpr (kNear) {
                near = (lut[kLutNearBase + (row >> 5)] >> (row & 31)) & 1u;
                compact = compact || near;
            }
        }
        if constexpr (kNear) {
            hasNear = __ballot(near) != 0;
            if (hasNear && nearAcc) {  // (nearAcc == nullptr: another wave of the team sums the remainders)
#pragma unroll
                for (int k = 0; k < 4; ++k) {  // 1 024 sums back to zero: 4 x 16 bytes per lane
                    *reinterpret_cast<u32x4*>(nearAcc + 256 * k + 4 * lane) = u32x4{0, 0, 0, 0};
                }
                __builtin_amdgcn_wave_barrier();
                // entries are packed from the front: four at a time, until no row of this board has any left (a net whose rows
                // carry a few remainders each pays one 16-byte load per lane, not four)
#pragma unroll 1
                for (int k = 0; k < kOutlierCap / 4; ++k) {
                    const u32x4 e = near ? *reinterpret_cast<const u32x4*>(outlierTab + size_t(row) * kOutlierCap + 4 * k)
                                         : u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                    if (!__ballot(e[0] != 0xFFFFFFFFu)) break;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (e[j] != 0xFFFFFFFFu) atomicAdd(nearAcc + (e[j] & 0xFFFFu), int32_t(int16_t(e[j] >> 16)));
                    }
                }
            }
        }
        const uint64_t compactMask = __ballot(occupied && compact), wideMask = b.occ & ~compactMask;
        const uint32_t slot = prefixCount(compact ? compactMask : wideMask);
        if (occupied && slot < kPsqCap) {
            if (compact) {
                thrList[slot] = (kThreatRows + row) * kL1;
            } else {
                psqList[slot] = row * (kL1 * 2);
            }
        }
        nCompact = min(uint32_t(popc64(compactMask)), uint32_t(kPsqCap));
        nPsq = min(uint32_t(popc64(wideMask)), uint32_t(kPsqCap));
    }
    uint32_t* threatList = thrList + nCompact;  // the reference's <= 256-entry threat list proper

    // threat rows (addThreatFeatures, nnue_state.cpp:309-328)
    nThr = emitThreatRows(threatList, 0, laneTargets(b, lane), piece, lane, x, flipColour, lut, pseudoTab);

    // pawn-pair rows (nnue_state.cpp:330-351)
    const bool isPawn = type == 0;
    const bool own = isPawn && (piece & 1) == c;
    nThr = nCompact + emitPawnPairRows(threatList, nThr, pawnPartners(isPawn, own, lane, ownPawns, theirPawns),
                                       ppId(int(lane) ^ x, !own), ownPawns, x);
    __builtin_amdgcn_wave_barrier();  // lists are produced and consumed by the same wave: LDS order suffices
    return hasNear;
}

// One lane's 16 bytes of a table row for the full-refresh gather. The row's byte offset stays in a VGPR (every lane reads
// the list entry from LDS itself) and goes into the 32-bit offset operand of `global_load_dwordx4 v, voff, s[base]` next to
// the table's SGPR base: ONE v_add_u32 per row (2.6 SIMD cycles) where round 1 / early round 2 paid v_readfirstlane + a
// 64-bit per-lane pointer add (v_lshl_add_u64), 4.1-4.2 cycles each (tools/probes/valu_rate_probe). +5.3 % on the whole
// bench in the same-run A/B (profiles/r02_ab_variants.txt).
__device__ __forceinline__ u32x4 loadGatherRow(const uint8_t* table, uint32_t rowOffset, uint32_t laneOff) {
    return *reinterpret_cast<const u32x4*>(table + size_t(rowOffset + laneOff));
}

// acc = ftBias + sum(piece-square rows) + sum(threat rows), all mod 2^16 per column.
// acc[r], r < 4: columns 8l+2r, 8l+2r+1 ; acc[4+r]: columns 512+8l+2r, 512+8l+2r+1  (lane l)
__device__ __forceinline__ void gatherFull(const FtTables& t, uint32_t lane, const uint32_t* psqList, uint32_t nPsq,
                                           const uint32_t* thrList, uint32_t nThr, uint32_t (&acc)[8],
                                           const int32_t* nearAcc = nullptr) {
    // Full-refresh rows come in through plain global loads (loadGatherRow). The buffer-load form the update kernel uses
    // (RowTable: SGPR row offset, no address arithmetic at all) was A/B-measured here too and LOSES 10 % (FT kernel 0.469
    // -> 0.514 ms, profiles/r02_ab_variants.txt): with 8 x 1 KiB in flight per wave the kernel is bound by the
    // vector-memory return path, and buffer loads sit longer in it; in the update kernel (4 loads in flight,
    // latency-bound) they win 5 %.
    const uint8_t* psqTable = reinterpret_cast<const uint8_t*>(t.psqW);
    const uint32_t laneOff = 16 * lane;

    // (1) u8 rows (threat, pawn-pair and compact piece-square rows, stored +128) FIRST, into their own accumulator while
    // the packed-i16 one is not live yet (8 VGPRs less in the hot loop): <= 256 rows x 255 never overflow a 16-bit
    // field, so plain 32-bit adds (v_add3_u32: two rows per add) are exact and no carry crosses fields.
    const uint32_t nFirst = min(nThr, uint32_t(kThreatCap));
    uint32_t tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    {
        uint32_t i = 0;
        for (; i + 8 <= nFirst; i += 8) {  // 8 x 1 KiB wave loads in flight
            u32x4 w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                w[u] = loadGatherRow(t.thrW, thrList[i + u], laneOff);
            }
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    tacc[2 * d] = tacc[2 * d] + unpackLo(w[u][d]) + unpackLo(w[u + 1][d]);
                    tacc[2 * d + 1] = tacc[2 * d + 1] + unpackHi(w[u][d]) + unpackHi(w[u + 1][d]);
                }
            }
        }
        for (; i + 2 <= nFirst; i += 2) {
            const u32x4 w0 = loadGatherRow(t.thrW, thrList[i], laneOff);
            const u32x4 w1 = loadGatherRow(t.thrW, thrList[i + 1], laneOff);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                tacc[2 * d] = tacc[2 * d] + unpackLo(w0[d]) + unpackLo(w1[d]);
                tacc[2 * d + 1] = tacc[2 * d + 1] + unpackHi(w0[d]) + unpackHi(w1[d]);
            }
        }
        if (i < nFirst) {
            const u32x4 w0 = loadGatherRow(t.thrW, thrList[i], laneOff);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                tacc[2 * d] += unpackLo(w0[d]);
                tacc[2 * d + 1] += unpackHi(w0[d]);
            }
        }
    }

    // (2) bias + wide (i16) piece-square rows
    {
        const u32x4 b0 = *reinterpret_cast<const u32x4*>(t.ftBias + 8 * lane);
        const u32x4 b1 = *reinterpret_cast<const u32x4*>(t.ftBias + 512 + 8 * lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[r] = b0[r];
            acc[4 + r] = b1[r];
        }
    }
    {
        uint32_t i = 0;
        for (; i + 4 <= nPsq; i += 4) {  // 8 x 1 KiB wave loads in flight
            u32x4 lo[4], hi[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                lo[u] = loadGatherRow(psqTable, psqList[i + u], laneOff);
                hi[u] = loadGatherRow(psqTable, psqList[i + u], laneOff + 1024);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[r] = pkAdd16(acc[r], lo[u][r]);
                    acc[4 + r] = pkAdd16(acc[4 + r], hi[u][r]);
                }
            }
        }
        for (; i < nPsq; ++i) {
            const u32x4 lo = loadGatherRow(psqTable, psqList[i], laneOff);
            const u32x4 hi = loadGatherRow(psqTable, psqList[i], laneOff + 1024);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[r] = pkAdd16(acc[r], lo[r]);
                acc[4 + r] = pkAdd16(acc[4 + r], hi[r]);
            }
        }
    }

    // (3) fold the u8 sums in (mod 2^16), removing the +128 storage bias: every u8 row contributed 128 to every column
    {
        const uint32_t corr = (nThr * 128u) & 0xFFFFu;
        const uint32_t corr2 = corr | (corr << 16);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            acc[r] = pkSub16(pkAdd16(acc[r], tacc[r]), corr2);
        }
    }
    // (3b) remainders of the near-compact rows' wide weights (buildFullLists<.., kNear>): this lane's 16 column sums, mod 2^16
    if (nearAcc) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const i32x4 lo = *reinterpret_cast<const i32x4*>(nearAcc + 512 * h + 8 * lane);
            const i32x4 hi = *reinterpret_cast<const i32x4*>(nearAcc + 512 * h + 8 * lane + 4);
            acc[4 * h + 0] = pkAdd16(acc[4 * h + 0], __builtin_amdgcn_perm(uint32_t(lo[1]), uint32_t(lo[0]), 0x05040100u));
            acc[4 * h + 1] = pkAdd16(acc[4 * h + 1], __builtin_amdgcn_perm(uint32_t(lo[3]), uint32_t(lo[2]), 0x05040100u));
            acc[4 * h + 2] = pkAdd16(acc[4 * h + 2], __builtin_amdgcn_perm(uint32_t(hi[1]), uint32_t(hi[0]), 0x05040100u));
            acc[4 * h + 3] = pkAdd16(acc[4 * h + 3], __builtin_amdgcn_perm(uint32_t(hi[3]), uint32_t(hi[2]), 0x05040100u));
        }
    }
    // (4) rows beyond 256 exist only when compact piece-square rows sit in front of a near-full threat list: one at a
    // time, straight into the wrapping accumulator
    for (uint32_t i = nFirst; i < nThr; ++i) {
        const u32x4 w0 = loadGatherRow(t.thrW, thrList[i], laneOff);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            acc[2 * d] = pkAdd16(acc[2 * d], unpackLo(w0[d]));
            acc[2 * d + 1] = pkAdd16(acc[2 * d + 1], unpackHi(w0[d]));
        }
    }
}

// pairwise activation of one perspective's accumulator -> 8 bytes per lane (columns 8l..8l+7 of the 512 outputs)
typedef int16_t i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2 activate(const uint32_t (&acc)[8]) {
    // Two columns per instruction on the packed-16-bit pipe (multilayer.h:108-145): i1 = clamp(a, 0, 255),
    // i2 = min(b, 255); ((i1 << 7) * i2) >> 16 floors i1 * i2 / 512 and negatives saturate to 0 - and since i1 >= 0 the
    // product is negative exactly when i2 is, so clamping i2 at 0 first gives the same byte. Then both factors are
    // 0..255, the product fits 16 bits: v_pk_max/min_i16, v_pk_mul_lo_u16, v_pk_lshrrev_b16.
    const i16x2 zero = {0, 0}, top = {255, 255};
    uint32_t q[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const i16x2 a = __builtin_bit_cast(i16x2, acc[r]), b = __builtin_bit_cast(i16x2, acc[4 + r]);
        const u16x2 i1 = __builtin_bit_cast(u16x2, __builtin_elementwise_min(__builtin_elementwise_max(a, zero), top));
        const u16x2 i2 = __builtin_bit_cast(u16x2, __builtin_elementwise_min(__builtin_elementwise_max(b, zero), top));
        q[r] = __builtin_bit_cast(uint32_t, u16x2((i1 * i2) >> 9));  // [col 2r | col 2r + 1 << 16], each 0..127
    }
    u32x2 o;  // bytes 0, 2 of each pair of dwords
    o[0] = __builtin_amdgcn_perm(q[1], q[0], 0x06040200u);
    o[1] = __builtin_amdgcn_perm(q[3], q[2], 0x06040200u);
    return o;
}

// ---------------------------------------------------------------------------------------------------------------------
// The gather on the MATRIX PIPE (round 4). ONE v_mfma_i32_16x16x64_i8 widens AND adds up four gathered i8 rows:
//   B operand = the loaded bytes: lane (n = lane & 15, kb = lane >> 4) holds 16 consecutive columns of row kb of the
//               current four rows (B[16 kb + i][n] = byte i);
//   A operand = a constant selection matrix: lane (m = lane & 15, kb) holds 16 bytes, byte m = 1, the rest 0
//               (A[m][16 kb + i] = (i == m)), so D[m][n] = sum over kb of byte m of lane (n, kb);
//   D         = lane (n, mb = lane >> 4), register r: m = 4 mb + r - the exact i32 sum of the four rows' column
//               16 n + 4 mb + r of the 256 columns the instruction covers (checked on the device: tools/probes/mfma_rowsum_probe).
// No VALU at all where gatherFull spends 12 instructions per row on zero-extension and adds. Rows are stored as plain i8; a
// piece-square row with weights outside i8 is two planes, v = 256 h + l with l = int8(v), h = int8((v - l) >> 8) (exact mod 2^16),
// the high planes summed first and shifted up. Sums are exact in i32 (<= 288 rows x 128), reduced mod 2^16 with the bias at the
// end: the reference's wrapping i16 accumulators. Used by the column-sliced gather (spx_ftx.hip); the whole-row form of round 4
// (gatherFullMfma in spx_ft_kernel) is retired to experiments/r04_ft_kernel_gather_on_the_matrix_pipe.hip.txt.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ i32x4 mfmaSelector(uint32_t lane) {
    const uint32_t m = lane & 15u, one = 1u << (8 * (m & 3u));
    i32x4 sel;
    sel[0] = (m >> 2) == 0 ? int32_t(one) : 0;
    sel[1] = (m >> 2) == 1 ? int32_t(one) : 0;
    sel[2] = (m >> 2) == 2 ? int32_t(one) : 0;
    sel[3] = (m >> 2) == 3 ? int32_t(one) : 0;
    return sel;
}

// Accumulator arena slot: [colour 0: i16[1024]][colour 1: i16[1024]] = 4 KiB, natural column order. Lane l owns
// columns {8l..8l+7} (16 B at 16l) and {512+8l..} (16 B at 1024+16l) - the same split as a piece-square row.
// kStream: child accumulators of a big batch are written once and, if at all, read much later - non-temporal stores
// keep those 4 KiB per update from evicting the weight rows out of L2 (65 536 updates: 454 -> 444 us per ply, self-play
// +3-4 %); small batches (<= 16 384: -4 %) are better off with their slots cached. The parent loads stay cached
// (streaming them too was measured: +5 % when every parent has one child, -15 % in self-play where ~35
// siblings share a parent: rejected). Compile-time, because the hint does not survive a run-time select between the two kinds
// of access.
template <bool kStream = false>
__device__ __forceinline__ void storeAcc(uint8_t* arena, uint32_t slot, int c, uint32_t lane, const uint32_t (&acc)[8]) {
    uint8_t* base = arena + size_t(slot) * kAccSlotBytes + size_t(c) * (kL1 * 2) + 16 * lane;
    u32x4 lo, hi;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        lo[r] = acc[r];
        hi[r] = acc[4 + r];
    }
    if constexpr (kStream) {
        __builtin_nontemporal_store(lo, reinterpret_cast<u32x4*>(base));
        __builtin_nontemporal_store(hi, reinterpret_cast<u32x4*>(base + 1024));
    } else {
        *reinterpret_cast<u32x4*>(base) = lo;
        *reinterpret_cast<u32x4*>(base + 1024) = hi;
    }
}
template <bool kStream = false>
__device__ __forceinline__ void loadAcc(const uint8_t* arena, uint32_t slot, int c, uint32_t lane, uint32_t (&acc)[8]) {
    const uint8_t* base = arena + size_t(slot) * kAccSlotBytes + size_t(c) * (kL1 * 2) + 16 * lane;
    u32x4 lo, hi;
    if constexpr (kStream) {
        lo = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base));
        hi = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(base + 1024));
    } else {
        lo = *reinterpret_cast<const u32x4*>(base);
        hi = *reinterpret_cast<const u32x4*>(base + 1024);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        acc[r] = lo[r];
        acc[4 + r] = hi[r];
    }
}

// child accumulator = parent accumulator - removed rows + added rows (updatePsq, nnue_state.cpp:34-87;
// applyThreatRows, :89-145). Lists hold byte offsets; wrapping i16; threat sums kept in non-overflowing 32-bit fields.
// (applyDeltaRows: the parent accumulator is already in `acc` - the chain kernel carries it from ply to ply in registers)
__device__ __forceinline__ void applyDeltaRows(const FtTables& t, uint32_t lane, const uint32_t* psqSub, uint32_t nPsqSub,
                                               const uint32_t* psqAdd, uint32_t nPsqAdd, const uint32_t* thrAdd,
                                               uint32_t nAdd, const uint32_t* thrSub, uint32_t nSub, uint32_t (&acc)[8]);
template <bool kStream = false>
__device__ __forceinline__ void applyDelta(const FtTables& t, const uint8_t* arena, uint32_t parentSlot, int c,
                                           uint32_t lane, const uint32_t* psqSub, uint32_t nPsqSub,
                                           const uint32_t* psqAdd, uint32_t nPsqAdd, const uint32_t* thrAdd,
                                           uint32_t nAdd, const uint32_t* thrSub, uint32_t nSub, uint32_t (&acc)[8]) {
    loadAcc(arena, parentSlot, c, lane, acc);
    applyDeltaRows(t, lane, psqSub, nPsqSub, psqAdd, nPsqAdd, thrAdd, nAdd, thrSub, nSub, acc);
}
__device__ __forceinline__ void applyDeltaRows(const FtTables& t, uint32_t lane, const uint32_t* psqSub, uint32_t nPsqSub,
                                               const uint32_t* psqAdd, uint32_t nPsqAdd, const uint32_t* thrAdd,
                                               uint32_t nAdd, const uint32_t* thrSub, uint32_t nSub, uint32_t (&acc)[8]) {
    const uint8_t* psqBase = reinterpret_cast<const uint8_t*>(t.psqW) + 16 * lane;
    for (uint32_t i = 0; i < nPsqSub; ++i) {
        const uint8_t* row = psqBase + __builtin_amdgcn_readfirstlane(psqSub[i]);
        const u32x4 lo = *reinterpret_cast<const u32x4*>(row);
        const u32x4 hi = *reinterpret_cast<const u32x4*>(row + 1024);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[r] = pkSub16(acc[r], lo[r]);
            acc[4 + r] = pkSub16(acc[4 + r], hi[r]);
        }
    }
    for (uint32_t i = 0; i < nPsqAdd; ++i) {
        const uint8_t* row = psqBase + __builtin_amdgcn_readfirstlane(psqAdd[i]);
        const u32x4 lo = *reinterpret_cast<const u32x4*>(row);
        const u32x4 hi = *reinterpret_cast<const u32x4*>(row + 1024);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[r] = pkAdd16(acc[r], lo[r]);
            acc[4 + r] = pkAdd16(acc[4 + r], hi[r]);
        }
    }
    const uint8_t* thrBase = t.thrW + 16 * lane;
    uint32_t tadd[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tsub[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t i = 0; i < nAdd; ++i) {
        const u32x4 w = *reinterpret_cast<const u32x4*>(thrBase + __builtin_amdgcn_readfirstlane(thrAdd[i]));
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            tadd[2 * d] += unpackLo(w[d]);
            tadd[2 * d + 1] += unpackHi(w[d]);
        }
    }
    for (uint32_t i = 0; i < nSub; ++i) {
        const u32x4 w = *reinterpret_cast<const u32x4*>(thrBase + __builtin_amdgcn_readfirstlane(thrSub[i]));
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            tsub[2 * d] += unpackLo(w[d]);
            tsub[2 * d + 1] += unpackHi(w[d]);
        }
    }
    // +128 storage bias: (nAdd - nSub) * 128 per column, mod 2^16
    const uint32_t corr = ((nAdd - nSub) * 128u) & 0xFFFFu;
    const uint32_t corr2 = corr | (corr << 16);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        acc[r] = pkSub16(pkSub16(pkAdd16(acc[r], tadd[r]), tsub[r]), corr2);
    }
}

}  // namespace

}  // namespace spx
