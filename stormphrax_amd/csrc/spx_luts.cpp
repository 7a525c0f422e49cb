// Host-side construction of the two small threat LUTs the kernels stage in LDS.
//
// Built the way the reference's constexpr generators build kOffsets and kAttackIndices
// (/root/reference/src/eval/nnue/features/threats.cpp:31-167) - by counting pseudo-attacks, not from closed-form
// formulas - using the same SPX_HD attack helpers the kernels use. The third reference table, kPieceIndices
// (48 KB, threats.cpp:74-106), is not materialised: kernels compute popcount(pseudo & below(to)) directly.
#include <climits>
#include <vector>

#include "spx_device_math.h"
#include "spx_internal.h"

namespace spx {

namespace {
// kPieceTargetMapNoPpThreats (threats.cpp:42-51): pawn-pawn threats are replaced by the pawn-pair inputs
constexpr int kTargetMap[6][6] = {
    {-1, 0, -1, 1, -1, -1}, {0, 1, 2, 3, 4, -1}, {0, 1, 2, 3, -1, -1},
    {0, 1, 2, 3, -1, -1},   {0, 1, 2, 3, 4, -1}, {-1, -1, -1, -1, -1, -1},
};

uint64_t pseudoAny(int piece, int sq) {
    if ((piece >> 1) == 5) {  // king: only its popcount matters (kOffsets), never indexed at run time
        const uint64_t b = 1ull << sq;
        const uint64_t row = b | ((b << 1) & ~kFileA) | ((b >> 1) & ~kFileH);
        return (row | (row << 8) | (row >> 8)) & ~b;
    }
    return piecePseudoAttacks(piece, sq);
}
}  // namespace

// Fills lut[kLutWords]; returns the total number of threat features (must be 59808).
int buildThreatLut(uint32_t* lut) {
    int targetCount[6];
    for (int src = 0; src < 6; ++src) {
        int count = 0;
        for (int dst = 0; dst < 6; ++dst) {
            count += kTargetMap[src][dst] >= 0;
        }
        targetCount[src] = 2 * count;  // kPieceTargetCount (threats.cpp:56-72)
    }

    int32_t pieceOffset[12], pieceBase[12];
    int32_t offset = 0;
    for (int colour : {1, 0}) {  // {white, black} (threats.cpp:116)
        for (int pt = 0; pt < 6; ++pt) {
            const int piece = (pt << 1) | colour;
            int32_t running = 0;
            for (int sq = 0; sq < 64; ++sq) {
                lut[piece * 64 + sq] = uint32_t(running);
                const int rank = sq >> 3;
                if (pt != 0 || (rank > 0 && rank < 7)) {
                    running += popc64(pseudoAny(piece ^ 1, sq));  // piece.flipColor() (threats.cpp:124)
                }
            }
            pieceOffset[piece] = running;
            pieceBase[piece] = offset;
            offset += targetCount[pt] * running;
        }
    }

    for (int a = 0; a < 12; ++a) {
        for (int v = 0; v < 12; ++v) {
            const int at = a >> 1, vt = v >> 1;
            const bool enemy = (a & 1) != (v & 1);
            const int map = kTargetMap[at][vt];
            const bool semiExcluded = at == vt && (enemy || at != 0);
            const bool excluded = map < 0;
            const int victimColourFlipped = (v & 1) ^ 1;  // attacked.color().flip().raw()
            const int32_t feature = pieceBase[a] + (victimColourFlipped * (targetCount[at] / 2) + map) * pieceOffset[a];
            lut[kLutOffsetsWords + (a * 12 + v) * 2 + 0] = uint32_t(excluded ? INT_MIN : feature);
            lut[kLutOffsetsWords + (a * 12 + v) * 2 + 1] = uint32_t((excluded || semiExcluded) ? INT_MIN : feature);
        }
    }
    return offset;
}

// Fills tab[kDeltaTabWords]: ray / knight-jump masks per (slot, square) and the pseudo-attack sets per (piece kind,
// square) of the threat-delta derivation (layout in spx_device_math.h).
void buildDeltaTables(uint64_t* tab) {
    static const int kRayStep[8][2] = {{0, 1}, {1, 1}, {1, 0}, {-1, 1}, {0, -1}, {-1, -1}, {-1, 0}, {1, -1}};  // (df, dr)
    static const int kJump[8][2] = {{1, 2}, {2, 1}, {2, -1}, {1, -2}, {-1, -2}, {-2, -1}, {-2, 1}, {-1, 2}};
    for (int sq = 0; sq < 64; ++sq) {
        const int file = sq & 7, rank = sq >> 3;
        for (int d = 0; d < 8; ++d) {
            uint64_t m = 0;
            for (int f = file + kRayStep[d][0], r = rank + kRayStep[d][1]; f >= 0 && f < 8 && r >= 0 && r < 8;
                 f += kRayStep[d][0], r += kRayStep[d][1]) {
                m |= 1ull << (r * 8 + f);
            }
            tab[d * 64 + sq] = m;
        }
        for (int j = 0; j < 8; ++j) {
            const int f = file + kJump[j][0], r = rank + kJump[j][1];
            tab[(8 + j) * 64 + sq] = (f >= 0 && f < 8 && r >= 0 && r < 8) ? 1ull << (r * 8 + f) : 0;
        }
        for (int k = 0; k < 6; ++k) {
            const int piece = k < 2 ? k : ((k - 1) << 1);  // 0 black pawn, 1 white pawn, then knight .. queen
            tab[kDeltaRayWords + k * 64 + sq] = piecePseudoAttacks(piece, sq);
        }
    }
}

}  // namespace spx
