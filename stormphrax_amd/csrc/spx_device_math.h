// Per-lane integer math of the feature extractor, written once for host and device (SPX_HD).
//
// The gfx950 kernels map one wavefront to one (position, perspective) and one LANE to one SQUARE (64 lanes = 64
// squares). Everything a lane needs - its piece's attack set, the threat / pawn-pair / piece-square row ids - is
// pure register arithmetic on the 64-bit occupancy; the only tables are the two small threat LUTs (4.2 KB, staged in
// LDS). Because the functions are SPX_HD the same code is exercised lane-by-lane on the CPU by spx_debug_features()
// (host logic tests, no GPU needed).
//
// Reference semantics restated here (paths relative to /root/reference/src):
//   attacks::getAttacks / getPseudoAttacks        attacks/attacks.h:130-170
//   psq::featureIndex + KingBucketsMergedMirrored eval/nnue/features/psq.h:204-284,317-365 ; eval/arch.h:53-65
//   threats::threatFeatureIndex                   eval/nnue/features/threats.cpp:170-198
//   threats::ppPawnId / ppFeatureIndex, kPpMasks  eval/nnue/features/threats.cpp:200-221 ; threats.h:106-123
//   marlinformat::PackedBoard                     datagen/marlinformat.h:32-84
#pragma once

#include <cstdint>

#include "spx_arch.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SPX_HD __host__ __device__ inline
#else
#define SPX_HD inline
#endif

namespace spx {

constexpr uint64_t kFileA = 0x0101010101010101ull;
constexpr uint64_t kFileH = 0x8080808080808080ull;

SPX_HD int popc64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}
SPX_HD int ctz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll(static_cast<unsigned long long>(x)) - 1;
#else
    return __builtin_ctzll(x);
#endif
}
SPX_HD int clz64(uint64_t x) {  // x != 0
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll(static_cast<long long>(x));
#else
    return __builtin_clzll(x);
#endif
}
SPX_HD uint64_t brev64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(x);
#else
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    return __builtin_bswap64(x);
#endif
}

// ---- line masks through a square (including the square itself) ----
SPX_HD uint64_t fileMask(int sq) {
    return kFileA << (sq & 7);
}
SPX_HD uint64_t rankMask(int sq) {
    return 0xFFull << (sq & 56);
}
SPX_HD uint64_t diagMask(int sq) {  // a1-h8 direction
    const int d = 8 * ((sq & 7) - (sq >> 3));
    const uint64_t m = 0x8040201008040201ull;
    return d >= 0 ? (m >> d) : (m << (-d));
}
SPX_HD uint64_t antiMask(int sq) {  // a8-h1 direction
    const int d = 8 * (7 - (sq & 7) - (sq >> 3));
    const uint64_t m = 0x0102040810204080ull;
    return d >= 0 ? (m >> d) : (m << (-d));
}

// Hyperbola quintessence with a full 64-bit bit reversal: sliding attacks along one line, blockers included.
SPX_HD uint64_t lineAttacks(uint64_t occ, uint64_t bit, uint64_t lineIncl) {
    const uint64_t maskEx = lineIncl & ~bit;
    uint64_t fwd = occ & maskEx;
    uint64_t rev = brev64(fwd);
    fwd -= bit;
    rev -= brev64(bit);
    return (fwd ^ brev64(rev)) & maskEx;
}

SPX_HD uint64_t knightAttacks(uint64_t b) {
    const uint64_t notA = ~kFileA, notAB = ~(kFileA | (kFileA << 1));
    const uint64_t notH = ~kFileH, notGH = ~(kFileH | (kFileH >> 1));
    return ((b << 17) & notA) | ((b << 10) & notAB) | ((b >> 6) & notAB) | ((b >> 15) & notA) | ((b << 15) & notH) |
           ((b << 6) & notGH) | ((b >> 10) & notGH) | ((b >> 17) & notH);
}
// colour: 1 = white (attacks up the board), 0 = black
SPX_HD uint64_t pawnAttacks(uint64_t b, int colour) {
    return colour ? (((b << 7) & ~kFileH) | ((b << 9) & ~kFileA)) : (((b >> 9) & ~kFileH) | ((b >> 7) & ~kFileA));
}

// attacks::getAttacks for the non-king piece types (kings never attack or get attacked in the threat features,
// nnue_state.cpp:319-323). `piece` = type<<1|colour.
SPX_HD uint64_t pieceAttacks(int piece, int sq, uint64_t occ) {
    const uint64_t bit = 1ull << sq;
    const int type = piece >> 1;
    uint64_t att = 0;
    if (type == 0) {
        att = pawnAttacks(bit, piece & 1);
    } else if (type == 1) {
        att = knightAttacks(bit);
    } else if (type <= 4) {
        if (type != 3) {  // bishop, queen
            att |= lineAttacks(occ, bit, diagMask(sq)) | lineAttacks(occ, bit, antiMask(sq));
        }
        if (type != 2) {  // rook, queen
            att |= lineAttacks(occ, bit, fileMask(sq)) | lineAttacks(occ, bit, rankMask(sq));
        }
    }
    return att;
}

// attacks::getPseudoAttacks (empty board) for the non-king types.
SPX_HD uint64_t piecePseudoAttacks(int piece, int sq) {
    const uint64_t bit = 1ull << sq;
    const int type = piece >> 1;
    uint64_t att = 0;
    if (type == 0) {
        att = pawnAttacks(bit, piece & 1);
    } else if (type == 1) {
        att = knightAttacks(bit);
    } else if (type <= 4) {
        if (type != 3) {
            att |= diagMask(sq) | antiMask(sq);
        }
        if (type != 2) {
            att |= fileMask(sq) | rankMask(sq);
        }
        att &= ~bit;
    }
    return att;
}

// kPpMasks[sq]: own file and both neighbours, whole files (threats.h:106-123)
SPX_HD uint64_t ppMask(int sq) {
    const uint64_t f = fileMask(sq);
    return f | ((f << 1) & ~kFileA) | ((f >> 1) & ~kFileH);
}

// ---- perspective transform ----
// xorMask = (c == black ? 56 : 0) ^ (king on files e-h ? 7 : 0); colour flip = (c == black)
SPX_HD int perspXor(int c, int kingSq) {
    return (c == 0 ? 56 : 0) ^ ((kingSq & 7) >= 4 ? 7 : 0);
}

// arch.h:53-65 half-board buckets (a1 = 0), expanded by KingBucketsMirrored::kBuckets (psq.h:209-226).
// Packed as 32 nibbles: index = rank*4 + min(file, 7-file).
SPX_HD int kingBucket(int kingSqRel) {  // kingSqRel: king square already rank-flipped for black
    const int rank = kingSqRel >> 3, file = kingSqRel & 7;
    const int f = file < 4 ? file : 7 - file;
    const int i = rank * 4 + f;
    // rows: 0 1 2 3 | 4 5 6 7 | 8 9 10 11 | 8 9 10 11 | 12 12 13 13 | 12 12 13 13 | 14 14 15 15 | 14 14 15 15
    const uint64_t lo = 0xBA98BA9876543210ull;  // entries 0..15
    const uint64_t hi = 0xFFEEFFEEDDCCDDCCull;  // entries 16..31
    return int(((i < 16 ? lo : hi) >> ((i & 15) * 4)) & 0xF);
}

// psq::featureIndex (psq.h:338-365): row of (piece, sq) for perspective c whose own king stands on kingSq.
SPX_HD uint32_t psqRow(int c, int piece, int sq, int kingSq) {
    const uint32_t type = uint32_t(piece >> 1);
    const uint32_t colour = (type == 5) ? 0u : (((piece & 1) == c) ? 0u : 1u);  // merged kings
    const int x = perspXor(c, kingSq);
    const int bucket = kingBucket(c == 0 ? (kingSq ^ 56) : kingSq);
    return uint32_t(bucket) * kPsqInputSize + colour * 384u + type * 64u + uint32_t(sq ^ x);
}

// Threat LUT layout shared by host builder and kernels (u32 words):
//   [0, 768)      offsets[piece][sq]            kOffsets.offsets   (threats.cpp:108-136)
//   [768, 1056)   attackIdx[attacker][attacked][forwards]  kAttackIndices (threats.cpp:138-167), INT_MIN = excluded
//   [1056, 1408)  one bit per piece-square row: 1 = every weight of the row fits i8, so the row is ALSO stored in
//                 the u8 row table (at row kThreatRows + r) and the kernels fetch that 1 KiB copy instead of the
//                 2 KiB i16 row. Derived from the loaded net (spx_ctx_create); lossless by construction.
constexpr int kLutOffsetsWords = 12 * 64;
constexpr int kLutAttackWords = 12 * 12 * 2;
constexpr int kLutCompactBase = kLutOffsetsWords + kLutAttackWords;
constexpr int kLutCompactWords = int(kPsqRows) / 32;
// [kLutNearBase, +kLutCompactWords): bitmap of the NEAR-compact piece-square rows: all but at most kOutlierCap of the 1 024
//                 weights fit i8. The u8 table holds the row with those weights clamped, FtTables::outlierTab the exact
//                 remainders (column | (weight - clamped) << 16, unused entries 0xFFFFFFFF); only the full-refresh kernel
//                 uses them (the update kernels see such a row as a wide row - the i16 table always holds every row)
constexpr int kLutNearBase = kLutCompactBase + kLutCompactWords;
constexpr int kLutWords = kLutNearBase + kLutCompactWords;
constexpr int kOutlierCap = 32;

// threats::threatFeatureIndex. pseudoRel = piecePseudoAttacks(attacker', asq') precomputed by the caller in the
// transformed frame (it replaces the 48 KB kPieceIndices table: popcount of pseudo-attacked squares below vsq').
SPX_HD int32_t threatRow(const uint32_t* lut, int attackerRel, int asqRel, uint64_t pseudoRel, int attackedRel,
                         int vsqRel) {
    const int forwards = asqRel < vsqRel;
    const int32_t attackIdx = int32_t(lut[kLutOffsetsWords + (attackerRel * 12 + attackedRel) * 2 + forwards]);
    const int32_t offset = int32_t(lut[attackerRel * 64 + asqRel]);
    const int32_t pieceIdx = popc64(pseudoRel & ((1ull << vsqRel) - 1));
    // excluded pairs carry INT_MIN: the sum stays negative exactly as in the reference's i32 arithmetic
    return int32_t(kPpRows) + attackIdx + offset + pieceIdx;
}

// threats::ppPawnId / ppFeatureIndex
SPX_HD uint32_t ppId(int sqRel, bool enemy) {
    return uint32_t((enemy ? 48 : 0) + sqRel - 8);
}
SPX_HD uint32_t ppRow(uint32_t a, uint32_t b) {
    const uint32_t hi = a > b ? a : b, lo = a > b ? b : a;
    return hi * (hi - 1) / 2 + lo;
}

// ---------------------------------------------------------------------------------------------------------------------
// Threat DELTA between two boards that differ on a small set S of squares (one move: |S| <= 4).
//
// The reference captures the delta while the move is made (BoardObserver + geometry ray walks, nnue.cpp:380-599,
// geometry.h:44-141). Here it is derived from the two boards, but with the same ray geometry instead of two full attack
// generations: a threat pair (attacker a on x -> victim v on y) differs between the boards only if it TOUCHES S - x in S,
// y in S, or a square of S strictly between x and y. For each board B the pairs that touch S are enumerated exactly once
// from the squares f of S, one (f, slot) per lane, slot = one of 8 rays or 8 knight jumps:
//   (A) attacker on f:  f -> nearest piece X on the slot, if the piece on f attacks along it        [owned by x = f]
//   (B) victim on f:    X -> f, if X attacks towards f and X is NOT in S (else (A) from X has it)   [owned by y = f]
//   (C) f empty in B:   slider X on the ray -> nearest piece V on the opposite ray, if neither end is in S and no other
//                       square of S lies between X and f (the S-square nearest the attacker owns the pair)
// sub = pairs of the parent board, add = pairs of the child board; pairs present in both cancel in the accumulator
// (sums mod 2^16), pairs that do not touch S are identical on both boards. Kings neither attack nor are attacked in
// the threat features (nnue_state.cpp:319-323), so they only matter as blockers.
//
// Slots: 0 N(+8) 1 NE(+9) 2 E(+1) 3 NW(+7) 4 S 5 SW 6 W 7 SE (slot ^ 4 = opposite ray); 8..15 knight jumps.
// Tables (u64 words, built by spx_luts.cpp:buildDeltaTables, staged in LDS by the update kernel):
//   [0, 1024)     ray[slot][sq]: the squares of the ray from sq (exclusive) to the edge / the one knight target (or 0)
//   [1024, 1408)  pseudo[k][sq]: piecePseudoAttacks, k = 0 black pawn, 1 white pawn, 2 knight, 3 bishop, 4 rook, 5 queen
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kDeltaRayWords = 16 * 64;
constexpr int kDeltaPseudoWords = 6 * 64;
constexpr int kDeltaTabWords = kDeltaRayWords + kDeltaPseudoWords;
constexpr uint32_t kNoDesc = 0xFFFFFFFFu;

SPX_HD int nearestOnSlot(const uint64_t* tab, uint64_t occ, int f, int slot) {  // -1: nothing there
    const uint64_t blockers = occ & tab[slot * 64 + f];
    const uint64_t safe = blockers | (blockers ? 0 : 1);
    const int sq = (slot & 12) == 4 ? 63 - clz64(safe) : ctz64(safe);  // rays 4..7 run towards lower squares
    return blockers ? sq : -1;
}

// does `piece` standing at one end of a ray attack the other end? dir = ray direction FROM the piece, adjacent = the two
// squares touch. Straight-line (this runs divergently per lane): direction sets as byte masks indexed by piece type -
// sliders at any distance (bishop 1,3,5,7; rook 0,2,4,6; queen all), pawns only next to it (white NE, NW; black SW, SE).
SPX_HD bool attacksOnRay(int piece, int dir, bool adjacent) {
    const uint32_t type = uint32_t(piece) >> 1;  // 6 = empty square
    const uint32_t far = uint32_t(0x000000FF55AA0000ull >> (type * 8)) & 0xFFu;
    const uint32_t near = (piece < 2 && adjacent) ? ((0x0AA0u >> (piece * 8)) & 0xFFu) : 0u;
    return ((far | near) >> dir) & 1u;
}

SPX_HD uint32_t packDesc(int attacker, int asq, int victim, int vsq) {  // ThreatDescriptor byte order (psq.h:30-35)
    return uint32_t(attacker) | (uint32_t(asq) << 8) | (uint32_t(victim) << 16) | (uint32_t(vsq) << 24);
}

// One lane of the candidate pass on board B (mail = its 64 piece bytes, occ = its occupancy); f in S.
SPX_HD void deltaCandidates(const uint64_t* tab, const uint8_t* mail, uint64_t occ, uint64_t changed, int f, int slot,
                            uint32_t& d1, uint32_t& d2) {
    const bool isRay = slot < 8;
    const int dir = slot & 7;
    const int xRaw = nearestOnSlot(tab, occ, f, slot), vRaw = nearestOnSlot(tab, occ, f, slot ^ 4);
    const bool hasX = xRaw >= 0, hasV = vRaw >= 0;
    const int x = hasX ? xRaw : 0, v = hasV ? vRaw : 0;
    const int pf = mail[f], px = mail[x], pv = mail[v];
    const int gap = x > f ? x - f : f - x;
    const bool adjacent = gap == ((0x07010908 >> ((dir & 3) * 8)) & 0xFF);  // |step| of N, NE, E, NW (rays only)
    const bool xChanged = (changed >> x) & 1, vChanged = (changed >> v) & 1;
    const bool fOccupied = pf != kNoPiece, fKing = (pf >> 1) == 5;
    // piece on one end attacks the other end: knights on the jump slots, everything else along rays
    const bool fAttacksX = isRay ? attacksOnRay(pf, dir, adjacent) : (pf >> 1) == 1;
    const bool xAttacksF = isRay ? attacksOnRay(px, dir ^ 4, adjacent) : (px >> 1) == 1;
    const uint64_t between = tab[slot * 64 + f] & ~tab[slot * 64 + x] & ~(1ull << x);
    const bool a = hasX & fOccupied & !fKing & ((px >> 1) != 5) & fAttacksX;                                  // (A)
    const bool b = hasX & fOccupied & !fKing & !xChanged & xAttacksF;                                         // (B)
    const bool c = hasX & !fOccupied & isRay & !xChanged & attacksOnRay(px, dir ^ 4, false) & hasV & !vChanged &
                   ((pv >> 1) != 5) & !(between & changed);                                                  // (C)
    d1 = a ? packDesc(pf, f, px, x) : (c ? packDesc(px, x, pv, v) : kNoDesc);
    d2 = b ? packDesc(px, x, pf, f) : kNoDesc;
}

// marlinformat nibble -> piece id (type<<1|colour, white = 1). Nibble: type | colour<<3 with black = 8 and
// type 6 = "unmoved rook" (marlinformat.h:39,52-58).
SPX_HD int nibbleToPiece(int nib) {
    int type = nib & 7;
    if (type == 6) {
        type = 3;
    }
    if (type == 7) {
        type = 0;  // not a marlinformat code: malformed record. Any valid piece keeps every table index in range.
    }
    return (type << 1) | ((nib & 8) ? 0 : 1);
}

// row of a packed threat descriptor for the perspective with transform (x, flipColour); < 0 = not a feature
SPX_HD int32_t descRow(const uint32_t* lut, const uint64_t* tab, uint32_t desc, int x, int flipColour) {
    const int attackerRel = int(desc & 0xFF) ^ flipColour, asqRel = int((desc >> 8) & 0xFF) ^ x;
    const int victimRel = int((desc >> 16) & 0xFF) ^ flipColour, vsqRel = int(desc >> 24) ^ x;
    const int k = attackerRel >= 2 ? (attackerRel >> 1) + 1 : attackerRel;
    return threatRow(lut, attackerRel, asqRel, tab[kDeltaRayWords + k * 64 + asqRel], victimRel, vsqRel);
}

// wdl::normalizeScore<false> (wdl.cpp:28-79; the <true> flavour is identical at the default evalSharpness of 100): a score
// in internal units -> "centipawns" such that 100 = 50 % win probability at this material; zero and decisive scores
// (|score| > kScoreWin, core.h:722-724) pass through. f64 cubic in material / 58, evaluated with fused multiply-adds (the
// reference's x86-64 builds contract the same expression), rounding half away from zero like std::round.
SPX_HD int32_t wdlNormalize(int32_t score, int32_t material) {
    if (score == 0 || score > kScoreWin || score < -kScoreWin) return score;
    const int32_t clamped = material < 17 ? 17 : (material > 78 ? 78 : material);
    const double m = double(clamped) / 58.0;
    const double a = __builtin_fma(__builtin_fma(__builtin_fma(-244.97139595, m, 687.39969858), m, -654.38002091), m, 608.47087786);
    return int32_t(__builtin_round(100.0 * (double(score) / a)));
}

// Position::classicalMaterial (position.h:515-521) from a record's nibble array: pawn 1, knight 3, bishop 3, rook 5
// (code 6 = rook with castling rights), queen 9
SPX_HD int32_t classicalMaterialOfNibble(int nib) {
    return int32_t((0x05095331u >> ((nib & 7) * 4)) & 0xFu);  // types 0..7: 1, 3, 3, 5, 9, 0 (king), 5, 0
}

// 64-bit key of a packed record's position identity (placement incl. castling-right codes, side to move, ep square;
// not the clocks): repetition detection in the self-play driver, computed on the device for every chosen child and on
// the host for start positions. Not a Zobrist key - only equality matters.
SPX_HD uint64_t recordKey(uint64_t occ, uint64_t nibLo, uint64_t nibHi, uint32_t stmEp) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (uint64_t(stmEp & 0xFFu) * 0xD6E8FEB86659FD93ull);
    const uint64_t words[3] = {occ, nibLo, nibHi};
    for (int i = 0; i < 3; ++i) {
        h ^= words[i];
        h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9ull;
        h = (h ^ (h >> 27)) * 0x94D049BB133111EBull;
        h ^= h >> 31;
    }
    return h;
}

// ---- datagen's per-ply bookkeeping (src/datagen/datagen.cpp), shared by the host self-play path and the device step
//      kernel (spx_game_step_kernel) so both adjudicate identically ----
// constants: datagen.cpp:82-91
constexpr int32_t kVerificationScoreLimit = 500, kWinAdjMinScore = 1250, kDrawAdjMaxScore = 10;
constexpr uint32_t kDrawAdjMinPlies = 70, kWinAdjPlyCount = 5, kDrawAdjPlyCount = 10;
constexpr uint32_t kNoOutcome = 255;  // else Outcome: 0 = white loss, 1 = draw, 2 = white win (datagen/common.h:24-28)

// eval::adjustStatic with contempt 0 (eval.cpp:24-27): what a search sees of a leaf is the network output clamped to
// +-(kScoreWin - 1), so a depth-1 score is never "decisive" (core.h:722-724)
SPX_HD int32_t clampStaticEval(int32_t v) {
    return v < -(kScoreWin - 1) ? -(kScoreWin - 1) : (v > kScoreWin - 1 ? kScoreWin - 1 : v);
}

// Position::plyFromStartpos (position.h:511-513) of a record: its fullmove counter and side to move
SPX_HD uint32_t plyFromStartpos(uint32_t fullmove, bool whiteToMove) {
    return fullmove * 2u - (whiteToMove ? 1u : 0u) - 1u;
}

// The win / loss / draw counters of one game (datagen.cpp:197-199) advanced by one searched move (datagen.cpp:224-252):
// `normScore` = wdl::normalizeScore of the white-point-of-view score at the material of the position searched,
// `ply` = plyFromStartpos of that position. Returns the adjudicated Outcome or kNoOutcome.
struct AdjCounters {
    uint32_t win, loss, draw;
};
SPX_HD uint32_t adjudicate(AdjCounters& c, int32_t normScore, uint32_t ply) {
    if (normScore > kWinAdjMinScore) {
        ++c.win;
        c.loss = c.draw = 0;
    } else if (normScore < -kWinAdjMinScore) {
        ++c.loss;
        c.win = c.draw = 0;
    } else if (ply >= kDrawAdjMinPlies && (normScore < 0 ? -normScore : normScore) < kDrawAdjMaxScore) {
        ++c.draw;
        c.win = c.loss = 0;
    } else {
        c.win = c.loss = c.draw = 0;
    }
    if (c.win >= kWinAdjPlyCount) return 2;
    if (c.loss >= kWinAdjPlyCount) return 0;
    if (c.draw >= kDrawAdjPlyCount) return 1;
    return kNoOutcome;
}

// The material part of Position::isDrawn (position.cpp:639-666) on a record: no pawns and no rooks / queens, and then
// KK, KNK / KBK, or one bishop each on squares of opposite colour (the reference's "KBKB OCB" test, restated as written)
SPX_HD bool insufficientMaterial(uint64_t occ, uint64_t nibLo, uint64_t nibHi) {
    uint64_t minors[2] = {0, 0}, bishops[2] = {0, 0};  // [black, white]
    uint32_t idx = 0;
    while (occ) {
        const int sq = ctz64(occ);
        occ &= occ - 1;
        const uint32_t nib = uint32_t(((idx < 16 ? nibLo : nibHi) >> ((idx & 15) * 4)) & 0xF);
        ++idx;
        const uint32_t type = nib & 7u, side = (nib & 8u) ? 0u : 1u;
        if (type == 0 || type == 3 || type == 4 || type == 6) return false;  // pawn, rook, queen, rook with castling right
        if (type == 1 || type == 2) minors[side] |= 1ull << sq;
        if (type == 2) bishops[side] |= 1ull << sq;
    }
    const int nb = popc64(minors[0]), nw = popc64(minors[1]);
    if (nb + nw == 0) return true;                                   // KK
    if ((nb == 0 && nw == 1) || (nw == 0 && nb == 1)) return true;   // KNK, KBK
    constexpr uint64_t kLightSquares = 0x55AA55AA55AA55AAull;
    return nb == 1 && nw == 1 && bishops[0] == minors[0] && bishops[1] == minors[1] &&
           ((bishops[0] & kLightSquares) == 0) != ((bishops[1] & kLightSquares) == 0);
}

}  // namespace spx
