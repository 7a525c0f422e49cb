// Device-side legal move generation + make-move on packed records, for the batched self-play driver.
//
// One wavefront per position, lane = from-square. A lane that holds a piece of the side to move builds its
// pseudo-legal target set from the same per-lane attack math the feature extractor uses (spx_device_math.h), tests
// every candidate the way the host core does - make the move, ask whether the own king is attacked
// (spx_chess.cpp:generateLegal, the reference's Position::isLegal contract, src/position.cpp) - and then writes one
// child RECORD per legal move: the 32-byte marlinformat PackedBoard of the position after the move, byte-identical to
// the host's packBoard(makeMove(..)) (spx_chess.cpp:320-367,663-688; format src/datagen/marlinformat.h:32-84), plus
// the viriformat move word (src/datagen/viriformat.cpp:37-52). Children of one position are contiguous; blocks of
// different positions are placed with one atomic add per position (their relative order is not deterministic, the
// content is). Castling follows Chess960 rules with "king takes rook" encoding, as the reference does.
#include <hip/hip_runtime.h>

#include "spx_arch.h"
#include "spx_device_math.h"
#include "spx_kernels.h"

namespace spx {

namespace {

using u128 = unsigned __int128;

__device__ __forceinline__ uint32_t laneId() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
__device__ __forceinline__ uint64_t kingAttacksBb(uint64_t b) {
    const uint64_t row = b | ((b << 1) & ~kFileA) | ((b >> 1) & ~kFileH);
    return (row | (row << 8) | (row >> 8)) & ~b;
}
__device__ __forceinline__ uint64_t below(int sq) {
    return (1ull << sq) - 1;
}
// squares lo..hi inclusive by index (the host's between(), spx_chess.cpp:299-304)
__device__ __forceinline__ uint64_t spanMask(int a, int c) {
    const int lo = a < c ? a : c, hi = a < c ? c : a;
    const uint64_t upTo = hi == 63 ? ~0ull : ((1ull << (hi + 1)) - 1);
    return upTo & ~below(lo);
}

struct Sets {  // wave-uniform bitboards of the parent position
    uint64_t occ, pawns, knights, bishops, rooks, queens, kings, white;
};

// Board::attacked (spx_chess.cpp:67-77) with `removed` squares taken out of the attackers (a captured piece)
__device__ __forceinline__ bool attackedBy(const Sets& s, int sq, int by, uint64_t occ, uint64_t removed) {
    const uint64_t bit = 1ull << sq;
    const uint64_t side = (by ? s.white : ~s.white) & ~removed;
    if (pawnAttacks(bit, by ^ 1) & s.pawns & side) return true;
    if (knightAttacks(bit) & s.knights & side) return true;
    if (kingAttacksBb(bit) & s.kings & side) return true;
    const uint64_t diag = (s.bishops | s.queens) & side;
    if (diag && ((lineAttacks(occ, bit, diagMask(sq)) | lineAttacks(occ, bit, antiMask(sq))) & diag)) return true;
    const uint64_t orth = (s.rooks | s.queens) & side;
    return orth && ((lineAttacks(occ, bit, fileMask(sq)) | lineAttacks(occ, bit, rankMask(sq))) & orth);
}

// ---- the 32-nibble piece array (one nibble per occupied square, in square order) as a 128-bit integer ----
__device__ __forceinline__ u128 nibMask(int idx) {  // the low idx nibbles
    return idx >= 32 ? ~u128(0) : ((u128(1) << (4 * idx)) - 1);
}
__device__ __forceinline__ u128 deleteNibble(u128 x, int idx) {
    const u128 low = x & nibMask(idx);
    const u128 high = idx >= 31 ? u128(0) : (x >> (4 * (idx + 1)));
    return low | (high << (4 * idx));
}
__device__ __forceinline__ u128 insertNibble(u128 x, int idx, uint32_t nib) {
    const u128 low = x & nibMask(idx);
    const u128 high = idx >= 32 ? u128(0) : (x >> (4 * idx));
    return low | (u128(nib) << (4 * idx)) | (idx >= 31 ? u128(0) : (high << (4 * (idx + 1))));
}
// every nibble equal to `pattern` (6 or 14: an unmoved rook of one colour) loses its castling right: 6 -> 3, 14 -> 11
__device__ __forceinline__ u128 dropCastlingRights(u128 x, uint32_t pattern) {
    const u128 ones = (u128(0x1111111111111111ull) << 64) | 0x1111111111111111ull;
    const u128 y = x ^ (ones * pattern);
    const u128 nonzero = (y | (y >> 1) | (y >> 2) | (y >> 3)) & ones;
    return x - (~nonzero & ones) * 3;
}

struct Parent {
    uint64_t occ;
    u128 nibbles;
    int us;            // side to move, 1 = white
    int ep;            // en-passant target square or 64
    uint32_t halfmove, fullmove;
};

enum ChildKind { kChildNormal = 0, kChildPromotion = 1, kChildCastling = 2, kChildEnPassant = 3 };

// makeMove + packBoard on the record (spx_chess.cpp:320-367,663-688). Castling: from = king square, to = rook square.
__device__ void writeChild(const Parent& p, const Sets& s, int from, int to, int kind, int promoType, uint64_t* out,
                           uint16_t* moveOut) {
    const int us = p.us;
    const uint32_t colourBit = us ? 0u : 8u;
    uint64_t occ = p.occ;
    u128 nib = p.nibbles;
    const int fromIdx = popc64(occ & below(from));
    const uint32_t moverNib = uint32_t(nib >> (4 * fromIdx)) & 0xFu;
    const int moverType = int(moverNib & 7u);  // 6 = rook that still has its castling right
    bool capture = false;
    int epOut = 64;
    if (kind == kChildCastling) {
        const int base = us ? 0 : 56;
        const int side = to > from ? 0 : 1;
        const int kTo = base + (side == 0 ? 6 : 2), rTo = base + (side == 0 ? 5 : 3);
        // remove king and rook (higher index first so the lower one stays valid), then put them back
        const int hiSq = from > to ? from : to, loSq = from > to ? to : from;
        nib = deleteNibble(nib, popc64(occ & below(hiSq)));
        nib = deleteNibble(nib, popc64(occ & below(loSq)));
        occ &= ~((1ull << from) | (1ull << to));
        const int firstSq = kTo < rTo ? kTo : rTo, secondSq = kTo < rTo ? rTo : kTo;
        const uint32_t firstNib = (firstSq == kTo ? 5u : 3u) | colourBit, secondNib = (secondSq == kTo ? 5u : 3u) | colourBit;
        nib = insertNibble(nib, popc64(occ & below(firstSq)), firstNib);
        occ |= 1ull << firstSq;
        nib = insertNibble(nib, popc64(occ & below(secondSq)), secondNib);
        occ |= 1ull << secondSq;
        nib = dropCastlingRights(nib, 6u | colourBit);
    } else {
        int capSq = -1;
        if (kind == kChildEnPassant) {
            capSq = to + (us ? -8 : 8);
        } else if ((occ >> to) & 1) {
            capSq = to;
        }
        if (capSq >= 0) {
            capture = true;
            nib = deleteNibble(nib, popc64(occ & below(capSq)));
            occ &= ~(1ull << capSq);
        }
        nib = deleteNibble(nib, popc64(occ & below(from)));
        occ &= ~(1ull << from);
        uint32_t placed = moverNib;
        if (kind == kChildPromotion) placed = uint32_t(promoType) | colourBit;
        if (moverType == 6) placed = 3u | colourBit;  // a rook that moves loses its right
        nib = insertNibble(nib, popc64(occ & below(to)), placed);
        occ |= 1ull << to;
        if (moverType == 5) nib = dropCastlingRights(nib, 6u | colourBit);
        if (moverType == 0 && (to - from == 16 || from - to == 16)) {
            // the ep square is recorded only if an enemy pawn can LEGALLY capture there (Position::filterEp,
            // position.cpp:1608-1683; spx_chess.cpp:filterEp): make each of the <= 2 candidate captures - capturer to the
            // ep square, the pushed pawn gone - and test the capturing side's king against our pieces (`s` still has the
            // pushed pawn on `from`: taken out of the attackers)
            const int epSq = (from + to) / 2;
            const uint64_t enemy = us ? ~s.white : s.white;
            uint64_t capturers = pawnAttacks(1ull << epSq, us) & s.pawns & enemy;
            const uint64_t enemyKing = s.kings & enemy;
            while (capturers && epOut == 64) {
                const int cf = ctz64(capturers);
                capturers &= capturers - 1;
                const uint64_t occ2 = (occ & ~((1ull << cf) | (1ull << to))) | (1ull << epSq);
                if (enemyKing && !attackedBy(s, ctz64(enemyKing), us, occ2, 1ull << from)) epOut = epSq;
            }
        }
    }
    const bool pawnMove = moverType == 0 && kind != kChildCastling;
    const uint32_t halfmove = (capture || pawnMove) ? 0u : min(p.halfmove + 1u, 255u);
    const uint32_t fullmove = (p.fullmove + (us == 0 ? 1u : 0u)) & 0xFFFFu;
    const uint32_t stmEp = (us ? 0x80u : 0u) | uint32_t(epOut);  // the child's side to move is the other colour
    out[0] = occ;
    out[1] = uint64_t(nib);
    out[2] = uint64_t(nib >> 64);
    out[3] = uint64_t(stmEp) | (uint64_t(halfmove) << 8) | (uint64_t(fullmove) << 16);  // eval, wdl, extra = 0
    static_assert(kChildPromotion == 1 && kChildCastling == 2 && kChildEnPassant == 3, "type bits below");
    const uint32_t typeBits = kind == kChildPromotion ? 0xC000u : kind == kChildCastling ? 0x8000u
                              : kind == kChildEnPassant ? 0x4000u : 0u;
    *moveOut = uint16_t(uint32_t(from) | (uint32_t(to) << 6) |
                        ((kind == kChildPromotion ? uint32_t(promoType - 1) : 0u) << 12) | typeBits);
}

}  // namespace

constexpr int kMaxItems = 256;  // pseudo-legal moves of one position (the legal maximum is 218)

__global__ __launch_bounds__(256) void spx_movegen_kernel(MovegenParams p) {
    __shared__ uint16_t sItems[4][kMaxItems];
    const uint32_t lane = laneId();
    const uint32_t wave = threadIdx.x >> 6;
    for (uint32_t it = blockIdx.x * 4 + wave; it < p.nPositions; it += gridDim.x * 4) {
        const uint64_t* rec = p.positions + size_t(it) * 4;
        Parent par;
        par.occ = rec[0];
        const uint64_t nibLo = rec[1], nibHi = rec[2];
        par.nibbles = (u128(nibHi) << 64) | nibLo;
        const uint32_t tail = uint32_t(rec[3]);
        par.us = (tail & 0x80u) ? 0 : 1;
        par.ep = int(tail & 0x7Fu);
        par.halfmove = (tail >> 8) & 0xFFu;
        par.fullmove = (tail >> 16) & 0xFFFFu;
        const int us = par.us, them = us ^ 1;

        const bool occupied = (par.occ >> lane) & 1;
        const uint32_t idx = min(uint32_t(popc64(par.occ & below(int(lane)))), 31u);
        const uint32_t nibRaw = occupied ? uint32_t(((idx < 16 ? nibLo : nibHi) >> ((idx & 15) * 4)) & 0xF) : 0u;
        int type = int(nibRaw & 7u);
        const bool rights = occupied && type == 6;
        if (type >= 6) type = (type == 6) ? 3 : 0;  // 7 is not a marlinformat code: treated as a pawn, like the evaluator does
        const bool isWhite = occupied && !(nibRaw & 8u);
        Sets s;
        s.occ = par.occ;
        s.pawns = __ballot(occupied && type == 0);
        s.knights = __ballot(occupied && type == 1);
        s.bishops = __ballot(occupied && type == 2);
        s.rooks = __ballot(occupied && type == 3);
        s.queens = __ballot(occupied && type == 4);
        s.kings = __ballot(occupied && type == 5);
        s.white = __ballot(isWhite);
        const uint64_t rightsBb = __ballot(rights);
        const uint64_t own = us ? s.white : (s.occ & ~s.white), enemy = s.occ & ~own;
        const uint64_t ownKing = s.kings & own;
        const int kingSq = ownKing ? ctz64(ownKing) : 0;
        const bool mine = occupied && (isWhite == (us == 1));
        const int from = int(lane);

        // ---- pseudo-legal targets of this lane's piece (generatePseudo, spx_chess.cpp:249-318) ----
        uint64_t targets = 0;
        if (mine) {
            const uint64_t bit = 1ull << from;
            if (type == 0) {
                const int fwd = us ? 8 : -8;
                const int one = from + fwd;
                if (one >= 0 && one < 64) {
                    if (!((s.occ >> one) & 1)) {
                        targets |= 1ull << one;
                        const bool home = us ? ((from >> 3) == 1) : ((from >> 3) == 6);
                        if (home && !((s.occ >> (one + fwd)) & 1)) targets |= 1ull << (one + fwd);
                    }
                }
                const uint64_t att = pawnAttacks(bit, us);
                targets |= att & enemy;
                if (par.ep < 64 && ((att >> par.ep) & 1)) targets |= 1ull << par.ep;
            } else if (type == 5) {
                targets = kingAttacksBb(bit) & ~own;
            } else {
                targets = pieceAttacks((type << 1) | us, from, s.occ) & ~own;
            }
        }
        // ---- flatten: every pseudo-legal (from, to) pair becomes one work item in LDS (lane order, targets ascending),
        // so that the heavy per-move work below runs on up to 64 moves at once instead of lane by lane ----
        uint32_t nItems;
        {
            const uint32_t mine32 = uint32_t(popc64(targets));
            uint32_t incl = mine32;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = __shfl_up(incl, d, 64);
                if (int(lane) >= d) incl += up;
            }
            nItems = min(uint32_t(__shfl(incl, 63, 64)), uint32_t(kMaxItems - 2));  // two slots stay free for castling
            uint32_t slot = incl - mine32;
            uint64_t rest = targets;
            while (rest) {
                const int to = ctz64(rest);
                rest &= rest - 1;
                if (slot < nItems) sItems[wave][slot] = uint16_t(uint32_t(from) | (uint32_t(to) << 6));
                ++slot;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- legality of every item: make the move, the own king must not be attacked (generateLegal,
        // spx_chess.cpp:602-612); flags are stored back into the item (bit 12 legal, 13 promotion, 14 en passant) ----
        uint32_t total = 0;
        for (uint32_t base = 0; base < nItems; base += 64) {
            const uint32_t i = base + lane;
            uint32_t children = 0;
            const uint32_t item = i < nItems ? sItems[wave][i] : 0u;
            const int f = int(item & 63), to = int((item >> 6) & 63);
            const int fType = __shfl(type, f, 64);  // outside the branch: the source lane must be active
            if (i < nItems) {
                const bool isEp = fType == 0 && to == par.ep && !((s.occ >> to) & 1) && (to & 7) != (f & 7);
                const bool isPromo = fType == 0 && (us ? to >= 56 : to < 8);
                const uint64_t capBit = isEp ? (1ull << (to + (us ? -8 : 8))) : (s.occ & (1ull << to));
                const uint64_t occ2 = (s.occ & ~(1ull << f) & ~capBit) | (1ull << to);
                const int ksq2 = fType == 5 ? to : kingSq;
                const bool ok = !attackedBy(s, ksq2, them, occ2, capBit);
                children = ok ? (isPromo ? 4u : 1u) : 0u;
                sItems[wave][i] = uint16_t(item | (ok ? 0x1000u : 0u) | (isPromo ? 0x2000u : 0u) | (isEp ? 0x4000u : 0u));
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) children += __shfl_xor(children, off, 64);
            total += children;
        }
        // ---- castling, on the king's lane (spx_chess.cpp:288-317): appended as ready-made legal items ----
        {
            uint32_t castleOk = 0;
            int castleRook[2] = {-1, -1};
            if (mine && type == 5) {
                const uint64_t myRooks = rightsBb & own;
                const uint64_t kingside = myRooks & ~below(from) & ~(1ull << from), queenside = myRooks & below(from);
                // unpackBoard keeps the LAST (highest) flagged rook per side (spx_chess.cpp:708-717)
                if (kingside) castleRook[0] = 63 - __clzll(kingside);
                if (queenside) castleRook[1] = 63 - __clzll(queenside);
                const int base = us ? 0 : 56;
                for (int side = 0; side < 2; ++side) {
                    const int rsq = castleRook[side];
                    if (rsq < 0) continue;
                    const int kTo = base + (side == 0 ? 6 : 2), rTo = base + (side == 0 ? 5 : 3);
                    const uint64_t span = spanMask(from, kTo) | spanMask(rsq, rTo);
                    const uint64_t others = s.occ & ~(1ull << from) & ~(1ull << rsq);
                    if (span & others) continue;
                    bool safe = true;
                    uint64_t path = spanMask(from, kTo);
                    while (path && safe) {
                        const int sq = ctz64(path);
                        path &= path - 1;
                        safe = !attackedBy(s, sq, them, others | (1ull << rsq), 0);
                    }
                    if (safe && attackedBy(s, kTo, them, others | (1ull << rTo), 0)) safe = false;
                    if (safe) castleOk |= 1u << side;
                }
                uint32_t slot = nItems;
                for (int side = 0; side < 2; ++side) {
                    if (castleOk & (1u << side)) {
                        sItems[wave][slot++] = uint16_t(uint32_t(from) | (uint32_t(castleRook[side]) << 6) | 0x9000u);
                    }
                }
            }
            const uint32_t nCastle = uint32_t(popc64(__ballot(castleOk & 1u)) + popc64(__ballot(castleOk & 2u)));
            nItems += nCastle;
            total += nCastle;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- placement: one atomic per position ----
        uint32_t base = 0;
        if (lane == 0) {
            base = total ? atomicAdd(p.cursor, total) : 0u;
            p.first[it] = base;
            p.count[it] = total;
            p.inCheck[it] = attackedBy(s, kingSq, them, s.occ, 0) ? 1 : 0;
        }
        base = __shfl(base, 0, 64);
        if (uint64_t(base) + total > p.capacity) continue;  // the host sees cursor > capacity and reports the overflow
        const uint32_t parentValue = p.parentValues ? p.parentValues[it] : it;
        // ---- children: one item per lane, promotions write four ----
        uint32_t done = 0;
        for (uint32_t chunk = 0; chunk < nItems; chunk += 64) {
            const uint32_t i = chunk + lane;
            const uint32_t item = i < nItems ? sItems[wave][i] : 0u;
            const bool legal = (item & 0x1000u) != 0;
            const bool isPromo = (item & 0x2000u) != 0;
            const uint32_t mineCount = legal ? (isPromo ? 4u : 1u) : 0u;
            uint32_t incl = mineCount;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = __shfl_up(incl, d, 64);
                if (int(lane) >= d) incl += up;
            }
            const uint32_t chunkTotal = __shfl(incl, 63, 64);
            if (legal) {
                uint32_t k = base + done + incl - mineCount;
                const int f = int(item & 63), to = int((item >> 6) & 63);
                if (isPromo) {
                    for (int pt = 4; pt >= 1; --pt) {
                        writeChild(par, s, f, to, kChildPromotion, pt, p.children + size_t(k) * 4, p.moves + k);
                        p.parents[k] = parentValue;
                        ++k;
                    }
                } else {
                    const int kind = (item & 0x8000u) ? kChildCastling : (item & 0x4000u) ? kChildEnPassant : kChildNormal;
                    writeChild(par, s, f, to, kind, 0, p.children + size_t(k) * 4, p.moves + k);
                    p.parents[k] = parentValue;
                }
            }
            done += chunkTotal;
        }
        __builtin_amdgcn_wave_barrier();  // the item list is reused by this wave's next position
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// A uniformly random legal move for every position (one thread each): the random plies of datagen's openings
// (src/datagen/datagen.cpp:153-171) and of spx_random_positions_gpu's playouts. The position's record is replaced by the
// chosen child; one splitmix64 draw per position and ply.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spx_pick_kernel(PickParams p) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= p.nGames) return;
    const uint32_t count = p.count[g];
    if (count == 0 || (p.enable && !p.enable[g])) return;
    const uint64_t state = p.rng[g] + 0x9E3779B97F4A7C15ull;
    uint64_t z = state;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const uint64_t* child = p.children + size_t(p.first[g] + uint32_t(z >> 32) % count) * 4;
    uint64_t* pos = p.positions + size_t(g) * 4;
    for (int w = 0; w < 4; ++w) pos[w] = child[w];
    p.rng[g] = state;
}

// ---------------------------------------------------------------------------------------------------------------------
// One ply of device-resident self-play for every seat of one half - the body of datagen's game loop
// (src/datagen/datagen.cpp:153-300) with the depth-1 "search" of spx_pick_kernel, one wavefront per seat:
//   * terminal positions: checkmate / stalemate (datagen.cpp:213-221); the 50-move rule of Position::isDrawn
//     (position.cpp:622-633: draw unless the position is checkmate - known one ply later, hence SeatState::pendingFifty);
//   * search: score(move) = -staticEval(child) (network output clamped like eval.cpp:24-27), uniformly among the moves
//     within `temperature` of the best; white-point-of-view score and its wdl::normalizeScore at the material of the
//     position searched (search.cpp:237-238);
//   * opening verification (datagen.cpp:176-190): on a game's first ply the search doubles as the verification search - a
//     normalised best score beyond +-500 discards the opening (not counted, nothing written) and the seat draws another;
//   * adjudication counters (datagen.cpp:224-252, spx_device_math.h:adjudicate), key history push, the move, then
//     Position::isDrawn of the new position - repetition (position.cpp:603-619 with ply 0: two earlier occurrences within
//     the halfmove window), insufficient material (:639-666), plus this driver's own ply cap - which overrides an
//     adjudicated result and records the move with score 0 (datagen.cpp:264-268); otherwise the recorded score is the
//     white-point-of-view score, 0 when |score| <= 2 (datagen.cpp:283-284);
//   * a finished game is written as viriformat (viriformat.cpp:28-63: 32-byte initial record with the outcome, 4 bytes per
//     move, 4 zero bytes) straight into the output ring; its seat takes the next opening from the pool while the run's
//     target allows;
//   * every seat that goes on (move made / new game) appends one record to the half's materialising update: parent slot
//     (the null slot - an empty board - for a new game, which the update kernel therefore rebuilds from scratch), child
//     slot (the seat's other slot), new current record. The ~35 siblings of the chosen move were evaluated without ever
//     being stored (eval-only children, spx_update_kernel with childSlots == nullptr).
// The host sees one SelfplayCounters copy per ply and the ring: O(1) work per ply.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t uniform(uint32_t v) {
    return uint32_t(__builtin_amdgcn_readfirstlane(int(v)));
}

// nWords move words: gm[0 .. nWords - 2] from the seat's buffer, the last one from a register (it may never have been stored)
__device__ void writeGame(const GameStepParams& p, uint32_t g, uint32_t lane, const uint32_t* gm, uint32_t nWords,
                          uint32_t lastWord, uint32_t outcome) {
    unsigned long long base = 0;
    if (lane == 0) {
        SelfplayCounters* c = p.counters;
        base = atomicAdd(p.streamWords, static_cast<unsigned long long>(8u + nWords + 1u));
        atomicAdd(&c->games, 1ull);
        atomicAdd(&c->positions, static_cast<unsigned long long>(nWords));
        atomicAdd(&c->outcomes[outcome], 1ull);
    }
    base = (static_cast<unsigned long long>(uniform(uint32_t(base >> 32))) << 32) | uniform(uint32_t(base));
    const uint32_t at = uint32_t(base % p.ringWords);
    if (lane < 8) {
        uint32_t w = reinterpret_cast<const uint32_t*>(p.initial + size_t(g) * 4)[lane];
        if (lane == 7) w = (w & 0xFF00FFFFu) | (outcome << 16);  // byte 30 = wdl (marlinformat.h:32-84)
        p.ring[(at + lane) % p.ringWords] = w;
    }
    for (uint32_t k = lane; k <= nWords; k += 64) {  // k == nWords: the null terminator
        const uint32_t w = k == nWords ? 0u : (k + 1 == nWords ? lastWord : gm[k]);
        p.ring[(at + 8 + k) % p.ringWords] = w;
    }
}

struct MoveResult {
    uint32_t outcome;        // the current game ended with this outcome (kNoOutcome: it goes on or was discarded)
    bool discard;            // ... or was discarded by the verification filter
    bool moved;              // the move was played and the game goes on
    uint32_t nWords, lastWord;  // the finished game's move words: gm[0 .. nWords - 2], then lastWord
};

// The depth-1 policy: score(move) = -staticEval(child), uniformly among the moves within `temperature` of the best (one
// splitmix64 draw per move played, the new state in rngState). evals = the position's children. -> child index, `best`.
__device__ uint32_t pickDepthOne(const int32_t* evals, uint32_t count, int32_t temperature, uint64_t rngIn, uint32_t lane,
                                 uint64_t& rngState, int32_t& best) {
    best = INT32_MIN;
    for (uint32_t k = lane; k < count; k += 64) best = max(best, clampStaticEval(-evals[k]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = max(best, __shfl_xor(best, off, 64));
    uint32_t nCandidates = 0;
    for (uint32_t base = 0; base < count; base += 64) {
        const uint32_t k = base + lane;
        const bool cand = k < count && clampStaticEval(-evals[k]) >= best - temperature;
        nCandidates += uint32_t(popc64(__ballot(cand)));
    }
    rngState = rngIn + 0x9E3779B97F4A7C15ull;
    uint64_t z = rngState;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    uint32_t target = temperature == 0 ? 0u : uint32_t(z >> 32) % nCandidates;
    uint32_t pick = 0;
    for (uint32_t base = 0; base < count; base += 64) {
        const uint32_t k = base + lane;
        const bool cand = k < count && clampStaticEval(-evals[k]) >= best - temperature;
        const uint64_t mask = __ballot(cand);
        const uint32_t here = uint32_t(popc64(mask));
        if (target < here) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
            pick = base + uint32_t(ctz64(__ballot(cand && rank == target)));
            break;
        }
        target -= here;
    }
    return pick;
}

// One searched move of a game (datagen.cpp:176-190,224-300): verification of the opening on the first ply, a decisive
// score ends the game at once (datagen.cpp:224-226), else the adjudication counters; the key history, Position::isDrawn of
// the new position, the recorded word. (p0..p3) = the position searched, (c0..c3) = the chosen child, `score` / `best` from
// the mover's point of view (best = the search's best score: what the verification filter looks at).
__device__ void playMove(const GameStepParams& p, SeatState& st, uint32_t lane, uint64_t p0, uint64_t p1, uint64_t p2,
                         uint64_t p3, uint64_t c0, uint64_t c1, uint64_t c2, uint64_t c3, uint32_t moveWord, int32_t score,
                         int32_t best, uint32_t* gm, uint64_t* keys, MoveResult& r) {
    const bool whiteToMove = !(p3 & 0x80u);
    // Position::classicalMaterial of the position searched (lane k sums nibble k)
    int32_t material = 0;
    {
        const uint32_t pieces = min(uint32_t(popc64(p0)), 32u);
        if (lane < pieces) material = classicalMaterialOfNibble(int(((lane < 16 ? p1 : p2) >> ((lane & 15) * 4)) & 0xF));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) material += __shfl_xor(material, off, 64);
    }
    const int32_t whiteScore = whiteToMove ? score : -score;
    const int32_t normScore = wdlNormalize(whiteScore, material);
    const int32_t normBest = wdlNormalize(whiteToMove ? best : -best, material);
    if (st.plies == 0 && (normBest > kVerificationScoreLimit || normBest < -kVerificationScoreLimit)) {
        r.discard = true;
        return;
    }
    uint32_t outcome;
    if (whiteScore > kScoreWin || whiteScore < -kScoreWin) {  // isDecisive (core.h:722-724): never true of a clamped static eval
        outcome = whiteScore > 0 ? 2u : 0u;
    } else {
        AdjCounters adj{st.win, st.loss, st.draw};
        outcome = adjudicate(adj, normScore, st.startPly + st.plies);
        st.win = adj.win, st.loss = adj.loss, st.draw = adj.draw;
    }
    const uint64_t newKey = recordKey(c0, c1, c2, uint32_t(c3));
    const uint32_t halfmove = uint32_t((c3 >> 8) & 0xFFu);
    // key history: the position searched is pushed, then the new position is looked for (isDrawnByRepetition, ply 0)
    const uint32_t size = st.plies + 1;
    if (lane == 0) keys[st.plies] = recordKey(p0, p1, p2, uint32_t(p3));
    const int32_t limit = max(0, int32_t(size) - int32_t(halfmove) - 2);
    const int32_t i = int32_t(size) - 4 - 2 * int32_t(lane);
    const bool hit = i >= limit && keys[i] == newKey;
    const bool repetition = popc64(__ballot(hit)) >= 2;
    // halfmove clock at 100: Position::isDrawn looks at nothing else (position.cpp:622-633) and the answer - a
    // draw unless checkmate - needs the next move generation: the seat goes on for one ply (pendingFifty) with
    // the adjudicated result, if any, parked in `reserved`
    const bool fifty = halfmove >= 100;
    const bool capped = st.plies + 1 >= p.maxPlies;
    const bool drawn = capped || (!fifty && (repetition || insufficientMaterial(c0, c1, c2)));
    if (drawn) outcome = 1;
    const int32_t recorded = drawn ? 0 : (whiteScore >= -2 && whiteScore <= 2 ? 0 : whiteScore);
    r.lastWord = moveWord | (uint32_t(uint16_t(int16_t(recorded))) << 16);
    r.nWords = st.plies + 1;
    st.pendingFifty = 0;
    st.reserved = 0;
    if (fifty && !drawn) {
        st.pendingFifty = 1;
        st.reserved = outcome == kNoOutcome ? 0u : outcome + 1;
        outcome = kNoOutcome;
    }
    r.outcome = outcome;
    if (outcome == kNoOutcome) {
        r.moved = true;
        if (lane == 0) gm[st.plies] = r.lastWord;
        st.plies += 1;
    }
}

// What the position a seat is about to search says before any search: the parked 50-move decision and mate / stalemate
// (datagen.cpp:213-221). -> true when the game is over (r filled in).
__device__ bool terminalBeforeSearch(const SeatState& st, uint32_t count, bool inCheck, bool whiteToMove, const uint32_t* gm,
                                     MoveResult& r) {
    if (st.pendingFifty && !(count == 0 && inCheck)) {
        r.outcome = 1;
        r.nWords = st.plies;
        r.lastWord = st.plies ? (gm[st.plies - 1] & 0xFFFFu) : 0u;  // that move led to a drawn position: score 0
        return true;
    }
    if (st.pendingFifty && st.reserved) {
        r.outcome = st.reserved - 1;  // checkmate on the board, so not drawn: the result adjudicated with that move stands
        r.nWords = st.plies;
        r.lastWord = st.plies ? gm[st.plies - 1] : 0u;
        return true;
    }
    if (count == 0) {
        r.outcome = inCheck ? (whiteToMove ? 0u : 2u) : 1u;
        r.nWords = st.plies;
        r.lastWord = st.plies ? gm[st.plies - 1] : 0u;
        return true;
    }
    return false;
}

// A free seat takes the next opening while the run's target allows. A discarded game hands its ticket on; the host keeps
// the pool ahead of every claim a step can make (spx_selfplay.cpp), so a claim beyond it only idles the seat.
__device__ bool claimOpening(const GameStepParams& p, uint32_t lane, bool discard, SeatState& st, uint64_t& r0, uint64_t& r1,
                             uint64_t& r2, uint64_t& r3, uint64_t& seed) {
    uint32_t claim = 0xFFFFFFFFu;
    if (lane == 0) {
        SelfplayCounters* c = p.counters;
        if (discard) atomicAdd(&c->discarded, 1ull);
        // (a plain look first: once the target is reached the drained seats stop adding to the counter every ply - it could
        // wrap on a very long tail - and stop queueing on one global atomic)
        const bool ticket = discard || (*reinterpret_cast<volatile uint32_t*>(&c->started) < p.targetGames &&
                                        atomicAdd(&c->started, 1u) < p.targetGames);
        if (ticket) {
            const uint32_t k = atomicAdd(&c->poolCursor, 1u);
            if (k < *reinterpret_cast<volatile uint32_t*>(&c->poolSize)) claim = k;
        }
    }
    claim = uniform(claim);
    st = SeatState{};
    if (claim == 0xFFFFFFFFu) return false;
    const uint64_t* rec = p.poolRecords + size_t(claim % p.poolCap) * 4;
    r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
    seed = p.poolSeeds[claim % p.poolCap];
    st.active = 1;
    st.startPly = plyFromStartpos(uint32_t((r3 >> 16) & 0xFFFFu), !(r3 & 0x80u));
    return true;
}

__global__ __launch_bounds__(256) void spx_game_step_kernel(GameStepParams p) {
    const uint32_t lane = laneId();
    const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= p.nSeats) return;
    SeatState st = p.state[g];
    uint64_t* pos = p.positions + size_t(g) * 4;
    uint32_t* gm = p.gameMoves + size_t(g) * p.maxPlies;
    uint64_t* keys = p.keys + size_t(g) * p.maxPlies;
    const uint32_t seat = p.seatBase + g;
    const uint32_t oldSlot = uniform(p.slots[g]);
    const uint32_t otherSlot = oldSlot == seat ? p.nSeatsTotal + seat : seat;

    MoveResult mr{kNoOutcome, false, false, 0, 0};
    uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;  // the chosen child
    if (st.active) {
        const uint32_t count = uniform(p.count[g]);
        const bool inCheck = uniform(p.inCheck[g]) != 0;
        const uint64_t p0 = pos[0], p1 = pos[1], p2 = pos[2], p3 = pos[3];
        if (!terminalBeforeSearch(st, count, inCheck, !(p3 & 0x80u), gm, mr)) {
            const uint32_t lo = uniform(p.first[g]);
            int32_t best;
            uint64_t rngState;
            const uint32_t c = lo + pickDepthOne(p.evals + lo, count, p.temperature, p.rng[g], lane, rngState, best);
            const uint64_t* child = p.children + size_t(c) * 4;
            c0 = child[0], c1 = child[1], c2 = child[2], c3 = child[3];
            playMove(p, st, lane, p0, p1, p2, p3, c0, c1, c2, c3, uint32_t(p.moves[c]), clampStaticEval(-p.evals[c]), best, gm,
                     keys, mr);
            if (mr.moved && lane == 0) p.rng[g] = rngState;
        }
    }

    if (mr.outcome != kNoOutcome) writeGame(p, g, lane, gm, mr.nWords, mr.lastWord, mr.outcome);
    bool started = false;
    uint64_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, seed = 0;
    if (!st.active || mr.outcome != kNoOutcome || mr.discard) started = claimOpening(p, lane, mr.discard, st, r0, r1, r2, r3, seed);
    if (lane == 0) p.state[g] = st;
    // The half's materialising update has one record per SEAT, at the seat's own index (no compaction: thousands of atomic
    // adds on one counter per launch cost more than the whole ply - ~116 ns each on this chip): the move played (parent =
    // old slot), a new game (parent = the null slot's empty board: the update kernel rebuilds the child from scratch), or,
    // for an idle seat, empty board -> empty board into the seat's spare slot (no rows, nothing read back).
    const bool live = started || mr.moved;
    const uint64_t n0 = started ? r0 : c0, n1 = started ? r1 : c1, n2 = started ? r2 : c2, n3 = started ? r3 : c3;
    if (lane == 0) {
        p.updParents[g] = mr.moved ? oldSlot : 2u * p.nSeatsTotal;
        p.updChildren[g] = otherSlot;
        if (live) p.slots[g] = otherSlot;
        if (started) p.rng[g] = seed;
    }
    if (lane < 4) {
        const uint64_t w = !live ? 0ull : (lane == 0 ? n0 : (lane == 1 ? n1 : (lane == 2 ? n2 : n3)));
        pos[lane] = w;  // (idle seat: an empty record generates no moves)
        p.updPositions[size_t(g) * 4 + lane] = w;
        if (started) p.initial[size_t(g) * 4 + lane] = w;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// One ROUND of the live fixed-node search for every seat of one half (SearchStepParams, spx_kernels.h, has the rules): this
// round's batch holds the children of every seat's `pending` node with their evaluations. One wavefront per seat consumes
// them - a terminal node or a depth-1 node returns its value to its parent frame, possibly through several levels; a
// deeper node keeps its children (records, move words, values) in the seat's frame storage - and walks on to the next node
// to expand: a materialising update (its parent's accumulator slot -> the slot of its level) and the next move generation's
// input. When an iteration ends at the root and the search is over, the move goes through the same bookkeeping as a depth-1
// move (playMove) and the new position - or a new game's opening - becomes the next pending node.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ long long childKey(int32_t value, uint32_t word) {  // value descending, then move word ascending
    return (static_cast<long long>(value) << 16) | static_cast<long long>(0xFFFFu - word);
}
__device__ __forceinline__ long long waveMax(long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t lo = __shfl_xor(uint32_t(v), off, 64);
        const int32_t hi = __shfl_xor(int32_t(v >> 32), off, 64);
        const long long o = (static_cast<long long>(hi) << 32) | lo;
        v = o > v ? o : v;
    }
    return v;
}

__global__ __launch_bounds__(256) void spx_search_step_kernel(SearchStepParams sp) {
    const GameStepParams& p = sp.game;
    const uint32_t lane = laneId();
    const uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= p.nSeats) return;
    SeatState st = p.state[g];
    SearchSeat ss = sp.seats[g];
    uint64_t* pos = p.positions + size_t(g) * 4;
    uint32_t* gm = p.gameMoves + size_t(g) * p.maxPlies;
    uint64_t* keys = p.keys + size_t(g) * p.maxPlies;
    const uint32_t seat = p.seatBase + g;
    const uint32_t oldSlot = uniform(p.slots[g]);
    const uint32_t otherSlot = oldSlot == seat ? p.nSeatsTotal + seat : seat;
    SearchFrame* frames = sp.frames + size_t(g) * kSearchLevels;
    uint64_t* fRecords = sp.frameRecords + size_t(g) * kSearchLevels * kSearchChildren * 4;
    int32_t* fValues = sp.frameValues + size_t(g) * kSearchLevels * kSearchChildren;
    uint16_t* fWords = sp.frameWords + size_t(g) * kSearchLevels * kSearchChildren;

    MoveResult mr{kNoOutcome, false, false, 0, 0};
    uint64_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;  // the chosen child when a move is played / the next node when the search goes on
    bool descend = false;
    uint32_t nextLevel = 0;
    if (st.active) {
        const uint32_t count = min(uniform(p.count[g]), kSearchChildren);
        const bool inCheck = uniform(p.inCheck[g]) != 0;
        const uint32_t lo = uniform(p.first[g]);
        const uint64_t p0 = pos[0], p1 = pos[1], p2 = pos[2], p3 = pos[3];
        const bool over = ss.top == 0 && terminalBeforeSearch(st, count, inCheck, !(p3 & 0x80u), gm, mr);
        if (!over) {
            ss.nodes += 1;
            if (lane == 0) sp.expansions[g] += 1;
            uint32_t L = ss.top;
            const uint32_t batchLevel = L;  // the frame whose children are this round's batch (its storage is written below:
                                            // reads of it in THIS round go to the batch instead)
            // the running frame lives in registers (wave-uniform); frames are read from memory when a child returns into them
            // and written when the search descends below them
            uint32_t fCount = count, fDepth = frames[L].depth;
            int32_t fAlpha = frames[L].alpha, fBeta = frames[L].beta, fBest = -kSearchInf, fBestIdx = -1, fCur = -1;
            uint64_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;  // visited
            fDepth = uniform(fDepth), fAlpha = int32_t(uniform(uint32_t(fAlpha))), fBeta = int32_t(uniform(uint32_t(fBeta)));
            int32_t res = 0;
            enum { kReturn, kDescend, kPlay } action;
            if (count == 0) {
                res = inCheck ? -(kSearchMate - int32_t(L)) : 0;
                action = kReturn;
            } else {
                if (L == 0 || fDepth >= 2) {  // the children stay: the root's always (the move played comes from them)
                    for (uint32_t k = lane; k < count; k += 64) {
                        const size_t at = size_t(L) * kSearchChildren + k;
                        fValues[at] = clampStaticEval(-p.evals[lo + k]);
                        fWords[at] = p.moves[lo + k];
                        for (int w = 0; w < 4; ++w) fRecords[at * 4 + w] = p.children[size_t(lo + k) * 4 + w];
                    }
                }
                if (fDepth == 1) {
                    long long key = INT64_MIN;
                    for (uint32_t k = lane; k < count; k += 64) {
                        const long long mine = childKey(clampStaticEval(-p.evals[lo + k]), p.moves[lo + k]);
                        key = mine > key ? mine : key;
                    }
                    const long long top = waveMax(key);
                    res = int32_t(top >> 16);
                    if (L == 0) {  // iteration 1 at the root: which child it is
                        uint32_t idx = 0xFFFFFFFFu;
                        for (uint32_t k = lane; k < count; k += 64) {
                            if (childKey(clampStaticEval(-p.evals[lo + k]), p.moves[lo + k]) == top) idx = k;
                        }
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) idx = min(idx, uint32_t(__shfl_xor(idx, off, 64)));
                        fBestIdx = int32_t(idx);
                    }
                    fBest = res;
                    action = kReturn;
                } else {
                    action = kDescend;
                }
            }
            for (;;) {
                if (action == kReturn && L != 0) {  // frame L returns `res` to its parent
                    --L;
                    const SearchFrame& f = frames[L];
                    fCount = uniform(f.count), fDepth = uniform(f.depth);
                    fAlpha = int32_t(uniform(uint32_t(f.alpha))), fBeta = int32_t(uniform(uint32_t(f.beta)));
                    fBest = int32_t(uniform(uint32_t(f.best))), fBestIdx = int32_t(uniform(uint32_t(f.bestIdx)));
                    fCur = int32_t(uniform(uint32_t(f.cur)));
                    v0 = f.visited[0], v1 = f.visited[1], v2 = f.visited[2], v3 = f.visited[3];
                    const int32_t v = -res;
                    if (v > fBest) fBest = v, fBestIdx = fCur;
                    if (v > fAlpha) fAlpha = v;
                    const uint32_t seen = uint32_t(popc64(v0) + popc64(v1) + popc64(v2) + popc64(v3));
                    if (fAlpha >= fBeta || seen >= fCount) {
                        res = fBest;
                        continue;
                    }
                    action = kDescend;
                }
                if (action == kReturn) {  // (L == 0) an iteration is complete
                    ss.prevBest = fBestIdx;
                    const bool decisive = fBest > kScoreWin || fBest < -kScoreWin;
                    if (ss.nodes >= sp.nodeBudget || ss.iter >= kSearchLevels || decisive) {
                        action = kPlay;
                        break;
                    }
                    ss.iter += 1;
                    fDepth = ss.iter;
                    fAlpha = -kSearchInf, fBeta = kSearchInf, fBest = -kSearchInf, fBestIdx = -1;
                    v0 = v1 = v2 = v3 = 0;
                    action = kDescend;
                }
                // search the next child of frame L: the best of the children not yet visited (the previous iteration's choice
                // first at the root)
                const bool fromBatch = L == batchLevel;
                long long key = INT64_MIN;
                for (uint32_t k = lane; k < fCount; k += 64) {
                    const uint64_t word = k < 64 ? v0 : (k < 128 ? v1 : (k < 192 ? v2 : v3));
                    if ((word >> (k & 63)) & 1) continue;
                    const size_t at = size_t(L) * kSearchChildren + k;
                    long long mine = fromBatch ? childKey(clampStaticEval(-p.evals[lo + k]), p.moves[lo + k])
                                               : childKey(fValues[at], fWords[at]);
                    if (L == 0 && int32_t(k) == ss.prevBest) mine = INT64_MAX;
                    key = mine > key ? mine : key;
                }
                const long long top = waveMax(key);
                uint32_t idx = 0xFFFFFFFFu;
                for (uint32_t k = lane; k < fCount; k += 64) {
                    const uint64_t word = k < 64 ? v0 : (k < 128 ? v1 : (k < 192 ? v2 : v3));
                    if ((word >> (k & 63)) & 1) continue;
                    const size_t at = size_t(L) * kSearchChildren + k;
                    long long mine = fromBatch ? childKey(clampStaticEval(-p.evals[lo + k]), p.moves[lo + k])
                                               : childKey(fValues[at], fWords[at]);
                    if (L == 0 && int32_t(k) == ss.prevBest) mine = INT64_MAX;
                    if (mine == top) idx = k;
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) idx = min(idx, uint32_t(__shfl_xor(idx, off, 64)));
                const uint64_t bit = 1ull << (idx & 63);
                v0 |= idx < 64 ? bit : 0ull;
                v1 |= (idx >= 64 && idx < 128) ? bit : 0ull;
                v2 |= (idx >= 128 && idx < 192) ? bit : 0ull;
                v3 |= idx >= 192 ? bit : 0ull;
                fCur = int32_t(idx);
                if (lane == 0) {
                    SearchFrame f{};
                    f.count = fCount, f.depth = fDepth, f.alpha = fAlpha, f.beta = fBeta, f.best = fBest, f.bestIdx = fBestIdx;
                    f.cur = fCur;
                    f.visited[0] = v0, f.visited[1] = v1, f.visited[2] = v2, f.visited[3] = v3;
                    frames[L] = f;
                    SearchFrame below{};
                    below.depth = fDepth - 1, below.alpha = -fBeta, below.beta = -fAlpha, below.best = -kSearchInf;
                    below.bestIdx = below.cur = -1;
                    frames[L + 1] = below;
                }
                const uint64_t* rec = fromBatch ? p.children + size_t(lo + idx) * 4
                                                : fRecords + (size_t(L) * kSearchChildren + idx) * 4;
                c0 = rec[0], c1 = rec[1], c2 = rec[2], c3 = rec[3];
                descend = true;
                nextLevel = L + 1;
                break;
            }
            if (action == kPlay) {
                int32_t score = fBest, best = fBest;
                uint32_t pick = uint32_t(fBestIdx);
                uint64_t rngState = 0;
                const bool policy = sp.nodeBudget <= 1;  // the depth-1 policy, temperature included (the root is this round's batch)
                if (policy) {
                    pick = pickDepthOne(p.evals + lo, count, p.temperature, p.rng[g], lane, rngState, best);
                    score = clampStaticEval(-p.evals[lo + pick]);
                }
                const bool fromBatch = batchLevel == 0;
                const uint64_t* rec = fromBatch ? p.children + size_t(lo + pick) * 4 : fRecords + size_t(pick) * 4;
                c0 = rec[0], c1 = rec[1], c2 = rec[2], c3 = rec[3];
                const uint32_t word = fromBatch ? uint32_t(p.moves[lo + pick]) : uint32_t(fWords[pick]);
                playMove(p, st, lane, p0, p1, p2, p3, c0, c1, c2, c3, word, score, best, gm, keys, mr);
                if (mr.moved && policy && lane == 0) p.rng[g] = rngState;
            }
        }
    }

    if (mr.outcome != kNoOutcome) writeGame(p, g, lane, gm, mr.nWords, mr.lastWord, mr.outcome);
    bool started = false;
    uint64_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, seed = 0;
    if (!st.active || mr.outcome != kNoOutcome || mr.discard) started = claimOpening(p, lane, mr.discard, st, r0, r1, r2, r3, seed);
    if (lane == 0) p.state[g] = st;
    // One materialising update per seat and round, as in the depth-1 driver: the node the search walks on to (its parent's
    // slot -> the slot of its level), the move played (old root slot -> the seat's other root slot), a new game (null slot:
    // rebuilt from scratch) or, for an idle seat, empty board -> empty board.
    const bool newRoot = started || mr.moved;
    const bool live = newRoot || descend;
    const uint64_t n0 = started ? r0 : c0, n1 = started ? r1 : c1, n2 = started ? r2 : c2, n3 = started ? r3 : c3;
    const uint32_t levelSlot = sp.levelSlotBase + (nextLevel - 1) * p.nSeatsTotal + seat;  // (descend only: nextLevel >= 1)
    const uint32_t parentSlot = nextLevel <= 1 ? oldSlot : levelSlot - p.nSeatsTotal;
    if (lane == 0) {
        p.updParents[g] = descend ? parentSlot : (mr.moved ? oldSlot : 2u * p.nSeatsTotal);
        p.updChildren[g] = descend ? levelSlot : otherSlot;
        sp.pendingSlots[g] = descend ? levelSlot : otherSlot;
        if (newRoot) {
            p.slots[g] = otherSlot;
            SearchFrame root{};
            root.depth = 1, root.alpha = -kSearchInf, root.beta = kSearchInf, root.best = -kSearchInf;
            root.bestIdx = root.cur = -1;
            frames[0] = root;
            ss = SearchSeat{0u, 1u, 0u, -1};
        } else if (descend) {
            ss.top = nextLevel;
        }
        if (started) p.rng[g] = seed;
        sp.seats[g] = ss;
    }
    if (lane < 4) {
        const uint64_t w = !live ? 0ull : (lane == 0 ? n0 : (lane == 1 ? n1 : (lane == 2 ? n2 : n3)));
        sp.pending[size_t(g) * 4 + lane] = w;  // (idle seat: an empty record generates no moves)
        p.updPositions[size_t(g) * 4 + lane] = w;
        if (newRoot || !live) pos[lane] = w;
        if (started) p.initial[size_t(g) * 4 + lane] = w;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// viriformat game streams -> one record per played move, on the device (src/datagen/viriformat.cpp:28-63; what
// Marlinformat::push would have stored, marlinformat.cpp:31-36): one THREAD per game replays its moves on the packed
// record itself (writeChild = makeMove + packBoard) and stores the position BEFORE every move with eval = the recorded
// score and wdl = the game's outcome. The stream is trusted like the reference's own reader trusts it; only a
// from-square without a piece of the side to move is counted (badGames) and ends that game's replay.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spx_viri_expand_kernel(ViriExpandParams p) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= p.nGames) return;
    const uint8_t* game = p.data + p.gameOffset[g];
    uint64_t rec[4];
    for (int w = 0; w < 4; ++w) {  // byte-wise: games start at 4-byte, not 8-byte, boundaries
        uint64_t v = 0;
        for (int b = 0; b < 8; ++b) v |= uint64_t(game[8 * w + b]) << (8 * b);
        rec[w] = v;
    }
    const uint64_t wdlBits = rec[3] & (0xFFull << 48);  // byte 30: the game's outcome
    const uint64_t first = p.outOffset[g], count = p.outOffset[g + 1] - first;
    const uint8_t* moves = game + 32;
    for (uint64_t k = 0; k < count; ++k) {
        const uint32_t mv = uint32_t(moves[4 * k]) | (uint32_t(moves[4 * k + 1]) << 8);
        const uint32_t score = uint32_t(moves[4 * k + 2]) | (uint32_t(moves[4 * k + 3]) << 8);
        uint64_t* dst = p.out + (first + k) * 4;
        dst[0] = rec[0];
        dst[1] = rec[1];
        dst[2] = rec[2];
        dst[3] = (rec[3] & 0x00000000FFFFFFFFull) | (uint64_t(score) << 32) | wdlBits;  // eval = score, wdl, extra = 0
        // the position after the move
        Parent par;
        par.occ = rec[0];
        par.nibbles = (u128(rec[2]) << 64) | rec[1];
        const uint32_t tail = uint32_t(rec[3]);
        par.us = (tail & 0x80u) ? 0 : 1;
        par.ep = int(tail & 0x7Fu);
        par.halfmove = (tail >> 8) & 0xFFu;
        par.fullmove = (tail >> 16) & 0xFFFFu;
        Sets s{};
        s.occ = par.occ;
        {
            uint64_t occ = par.occ;
            u128 nib = par.nibbles;
            while (occ) {
                const uint64_t bit = occ & (~occ + 1);
                occ &= occ - 1;
                const uint32_t n = uint32_t(nib) & 0xFu;
                nib >>= 4;
                const uint32_t t = n & 7u;
                if (t == 0u) s.pawns |= bit;
                else if (t == 1u) s.knights |= bit;
                else if (t == 2u) s.bishops |= bit;
                else if (t == 3u || t == 6u) s.rooks |= bit;
                else if (t == 4u) s.queens |= bit;
                else if (t == 5u) s.kings |= bit;
                if (!(n & 8u)) s.white |= bit;
            }
        }
        const int from = int(mv & 63u), to = int((mv >> 6) & 63u);
        const uint32_t type = mv >> 14;  // 0 normal, 1 en passant, 2 castling, 3 promotion (viriformat.cpp:37-52)
        const int kind = type == 0 ? kChildNormal : type == 1 ? kChildEnPassant : type == 2 ? kChildCastling : kChildPromotion;
        if (p.unfiltered) {  // datagen.cpp:254: filtered = pos.isCheck() || pos.isNoisy(move) (position.cpp:683-689)
            const uint64_t ownKing = s.kings & (par.us ? s.white : ~s.white);
            const bool inCheck = ownKing && attackedBy(s, ctz64(ownKing), par.us ^ 1, par.occ, 0);
            const bool noisy = kind != kChildCastling &&
                               (kind == kChildEnPassant || (kind == kChildPromotion && ((mv >> 12) & 3u) == 3u) ||
                                ((par.occ >> to) & 1));
            p.unfiltered[first + k] = (inCheck || noisy) ? 0 : 1;
        }
        const bool ownPiece = ((par.occ >> from) & 1) && (((s.white >> from) & 1) == uint64_t(par.us));
        if (!ownPiece) {
            atomicAdd(p.badGames, 1u);
            for (uint64_t r = k + 1; r < count; ++r) {  // keep the output defined: the rest of the game repeats this position
                uint64_t* rest = p.out + (first + r) * 4;
                rest[0] = dst[0];
                rest[1] = dst[1];
                rest[2] = dst[2];
                rest[3] = dst[3];
                if (p.unfiltered) p.unfiltered[first + r] = 0;
            }
            return;
        }
        uint64_t next[4];
        uint16_t ignored;
        writeChild(par, s, from, to, kind, int((mv >> 12) & 3u) + 1, next, &ignored);
        rec[0] = next[0];
        rec[1] = next[1];
        rec[2] = next[2];
        rec[3] = next[3];
    }
}

hipError_t launchViriExpand(const ViriExpandParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(spx_viri_expand_kernel, dim3((p.nGames + 255) / 256), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// End of a half's ply: the run-wide counters and this ply's child count go to the host (page-locked memory mapped into the
// device: no blit kernels on the stream) and the child cursor is zeroed for the half's next move generation.
// The shared counters are copied in 64-bit words with 64-bit atomic loads (the other half's step kernel may be adding to
// them: a word-by-word copy could tear a counter whose low word wraps); they only grow, the host takes the element-wise maximum.
__global__ __launch_bounds__(64) void spx_game_status_kernel(const SelfplayCounters* counters, uint32_t* total,
                                                             const unsigned long long* streamWords, unsigned long long* hostStatus) {
    constexpr uint32_t kWords = sizeof(SelfplayCounters) / 8;
    const uint32_t t = threadIdx.x;
    if (t < kWords) {
        hostStatus[t] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(counters) + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (t == kWords) hostStatus[kWords] = *streamWords;  // this half's own: its step kernel is done (same stream)
    if (t == kWords + 1) {
        hostStatus[kWords + 1] = *total;
        *total = 0;
    }
}

hipError_t launchGameStatus(const SelfplayCounters* counters, uint32_t* total, const unsigned long long* streamWords,
                            void* hostStatus, hipStream_t stream) {
    hipLaunchKernelGGL(spx_game_status_kernel, dim3(1), dim3(64), 0, stream, counters, total, streamWords,
                       static_cast<unsigned long long*>(hostStatus));
    return hipGetLastError();
}

hipError_t launchGameStep(const GameStepParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(spx_game_step_kernel, dim3((p.nSeats + 3) / 4), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launchSearchStep(const SearchStepParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(spx_search_step_kernel, dim3((p.game.nSeats + 3) / 4), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launchPick(const PickParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(spx_pick_kernel, dim3((p.nGames + 255) / 256), dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launchMovegen(const MovegenParams& p, uint32_t gridBlocks, hipStream_t stream) {
    hipLaunchKernelGGL(spx_movegen_kernel, dim3(gridBlocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace spx
