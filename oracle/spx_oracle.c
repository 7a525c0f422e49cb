/*
 * spx_oracle.c - CPU restatement of Stormphrax 8.0.2's NNUE evaluation path (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the parity oracle for the MI355X kernels. It is a scalar, plain-C restatement of the reference's
 * algorithm; every function cites the reference file:line it follows (paths relative to /root/reference).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it. The product library
 * (stormphrax_amd/csrc) never links, imports or falls back to anything in this directory.
 *
 * Pinning: tests/golden/ holds raw evals and per-perspective feature-row lists produced by the *compiled
 * reference itself* (oracle/ref_probe.cpp linked against the reference's own sources, see oracle/Makefile);
 * tests/test_oracle_golden.py checks this restatement against every one of them.
 *
 * Conventions (core.h:336-350,389-415): square a1 = 0 ... h8 = 63 (sq = rank*8 + file); colour black = 0,
 * white = 1; piece = type<<1 | colour with type pawn=0 knight=1 bishop=2 rook=3 queen=4 king=5; 12 = empty.
 * All arithmetic that wraps in the reference is done on unsigned types here (no signed-overflow UB).
 */
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define L1 1024
#define PAIRS 512
#define L2 32
#define L2F 64
#define L3 64
#define NBUCKETS 8
#define PSQ_ROWS 11264
#define PSQ_INPUT 704
#define PP_ROWS 4560
#define THREAT_ROWS 64368
#define NO_PIECE 12

typedef struct {
    const int16_t* psqW;   /* [11264][1024] */
    const int8_t* threatW; /* [64368][1024] */
    const int16_t* ftBias; /* [1024] */
    const int8_t* l1W;     /* [8][256][32][4] */
    const int32_t* l1B;    /* [8][32] */
    const int32_t* l2W;    /* [8][64][64] */
    const int32_t* l2B;    /* [8][64] */
    const int32_t* l3W;    /* [8][64] */
    const int32_t* l3B;    /* [8] */
} Net;

static Net g_net;
static unsigned char* g_blob;

/* ---------------------------------------------------------------------------------------------------------------
 * Bitboard helpers and pseudo-attack tables (attacks/attacks.h:100-170; bitboard.h:300-350).
 * ------------------------------------------------------------------------------------------------------------- */
static uint64_t g_pawnAtt[2][64], g_knightAtt[64], g_kingAtt[64];
static const int kDirs[8][2] = {/* N */ {0, 1},  /* NE */ {1, 1},   /* E */ {1, 0},  /* SE */ {1, -1},
                                /* S */ {0, -1}, /* SW */ {-1, -1}, /* W */ {-1, 0}, /* NW */ {-1, 1}};

static inline int popcnt(uint64_t x) {
    return __builtin_popcountll(x);
}
static inline int lsb(uint64_t x) {
    return __builtin_ctzll(x);
}

static uint64_t rayAttacks(int sq, uint64_t occ, int diag, int orth) {
    uint64_t att = 0;
    for (int d = 0; d < 8; ++d) {
        const int isDiag = d & 1;
        if ((isDiag && !diag) || (!isDiag && !orth)) {
            continue;
        }
        int f = sq & 7, r = sq >> 3;
        for (;;) {
            f += kDirs[d][0];
            r += kDirs[d][1];
            if (f < 0 || f > 7 || r < 0 || r > 7) {
                break;
            }
            const uint64_t b = 1ull << (r * 8 + f);
            att |= b;
            if (occ & b) {
                break;
            }
        }
    }
    return att;
}

/* attacks::getAttacks (attacks/attacks.h:130-150) */
static uint64_t attacksOf(int piece, int sq, uint64_t occ) {
    switch (piece >> 1) {
        case 0: return g_pawnAtt[piece & 1][sq];
        case 1: return g_knightAtt[sq];
        case 2: return rayAttacks(sq, occ, 1, 0);
        case 3: return rayAttacks(sq, occ, 0, 1);
        case 4: return rayAttacks(sq, occ, 1, 1);
        case 5: return g_kingAtt[sq];
        default: return 0;
    }
}

static void initAttackTables(void) {
    static const int kn[8][2] = {{1, 2}, {2, 1}, {2, -1}, {1, -2}, {-1, -2}, {-2, -1}, {-2, 1}, {-1, 2}};
    for (int sq = 0; sq < 64; ++sq) {
        const int f = sq & 7, r = sq >> 3;
        uint64_t k = 0, n = 0, pw = 0, pb = 0;
        for (int i = 0; i < 8; ++i) {
            int nf = f + kn[i][0], nr = r + kn[i][1];
            if (nf >= 0 && nf < 8 && nr >= 0 && nr < 8) {
                n |= 1ull << (nr * 8 + nf);
            }
            nf = f + kDirs[i][0];
            nr = r + kDirs[i][1];
            if (nf >= 0 && nf < 8 && nr >= 0 && nr < 8) {
                k |= 1ull << (nr * 8 + nf);
            }
        }
        /* white pawns attack up-left/up-right, black pawns down-left/down-right (bitboard.h:316-334) */
        if (r < 7) {
            if (f > 0) pw |= 1ull << (sq + 7);
            if (f < 7) pw |= 1ull << (sq + 9);
        }
        if (r > 0) {
            if (f > 0) pb |= 1ull << (sq - 9);
            if (f < 7) pb |= 1ull << (sq - 7);
        }
        g_knightAtt[sq] = n;
        g_kingAtt[sq] = k;
        g_pawnAtt[1][sq] = pw;
        g_pawnAtt[0][sq] = pb;
    }
}

/* ---------------------------------------------------------------------------------------------------------------
 * Threat-index LUTs, built exactly as the reference's constexpr generators do (features/threats.cpp:31-167).
 * ------------------------------------------------------------------------------------------------------------- */
/* kPieceTargetMapNoPpThreats, threats.cpp:42-51 (InputFeatureSet::kPawnPawnInputs == true selects this map) */
static const int kPieceTargetMap[6][6] = {
    {-1, 0, -1, 1, -1, -1}, {0, 1, 2, 3, 4, -1}, {0, 1, 2, 3, -1, -1},
    {0, 1, 2, 3, -1, -1},   {0, 1, 2, 3, 4, -1}, {-1, -1, -1, -1, -1, -1},
};
static int g_targetCount[6];             /* kPieceTargetCount, threats.cpp:56-72 */
static uint8_t g_pieceIdx[12][64][64];   /* kPieceIndices,     threats.cpp:74-106 */
static uint32_t g_offsets[12][64];       /* kOffsets.offsets,  threats.cpp:108-136 */
static int32_t g_pieceOffset[12], g_pieceBase[12]; /* kOffsets.indices {pieceOffset, offset} */
static int32_t g_attackIdx[12][12][2];   /* kAttackIndices,    threats.cpp:138-167 */
static uint64_t g_ppMask[64];            /* kPpMasks,          threats.h:106-123 */
static int32_t g_totalThreatFeatures;

static void initThreatLuts(void) {
    for (int src = 0; src < 6; ++src) {
        int count = 0;
        for (int dst = 0; dst < 6; ++dst) {
            if (kPieceTargetMap[src][dst] >= 0) ++count;
        }
        g_targetCount[src] = 2 * count;
    }

    /* generatePieceIndices: number of pseudo-attacked squares below `to` (threats.cpp:74-90); pawns per colour,
     * other types generated for the black piece and shared (threats.cpp:95-103; colour-independent anyway) */
    for (int piece = 0; piece < 12; ++piece) {
        for (int from = 0; from < 64; ++from) {
            const uint64_t pseudo = attacksOf(piece, from, 0);
            for (int to = 0; to < 64; ++to) {
                g_pieceIdx[piece][from][to] = (uint8_t)popcnt(pseudo & ((1ull << to) - 1));
            }
        }
    }

    /* kOffsets (threats.cpp:108-136): colours in order {white, black}, types pawn..king */
    int32_t offset = 0;
    static const int colourOrder[2] = {1, 0};
    for (int ci = 0; ci < 2; ++ci) {
        const int colour = colourOrder[ci];
        for (int pt = 0; pt < 6; ++pt) {
            const int piece = (pt << 1) | colour;
            int32_t pieceOffset = 0;
            for (int sq = 0; sq < 64; ++sq) {
                g_offsets[piece][sq] = (uint32_t)pieceOffset;
                const int rank = sq >> 3;
                if (pt != 0 || (rank > 0 && rank < 7)) {
                    pieceOffset += popcnt(attacksOf(piece ^ 1, sq, 0)); /* piece.flipColor(), threats.cpp:124 */
                }
            }
            g_pieceOffset[piece] = pieceOffset;
            g_pieceBase[piece] = offset;
            offset += g_targetCount[pt] * pieceOffset;
        }
    }
    g_totalThreatFeatures = offset;

    /* kAttackIndices (threats.cpp:138-167) */
    for (int a = 0; a < 12; ++a) {
        for (int v = 0; v < 12; ++v) {
            const int at = a >> 1, vt = v >> 1;
            const int enemy = (a & 1) != (v & 1);
            const int map = kPieceTargetMap[at][vt];
            const int semiExcluded = at == vt && (enemy || at != 0);
            const int excluded = map < 0;
            const int attackedColourFlipped = (v & 1) ^ 1; /* attacked.color().flip().raw(): white -> 0, black -> 1 */
            const int32_t feature =
                g_pieceBase[a] + (attackedColourFlipped * (g_targetCount[at] / 2) + map) * g_pieceOffset[a];
            g_attackIdx[a][v][0] = excluded ? INT_MIN : feature;
            g_attackIdx[a][v][1] = (excluded || semiExcluded) ? INT_MIN : feature;
        }
    }

    /* kPpMasks (threats.h:106-123): own file plus both neighbours, full files, for squares 8..55 */
    for (int sq = 8; sq < 56; ++sq) {
        const int f = sq & 7;
        uint64_t m = 0;
        for (int ff = (f > 0 ? f - 1 : 0); ff <= (f < 7 ? f + 1 : 7); ++ff) {
            m |= 0x0101010101010101ull << ff;
        }
        g_ppMask[sq] = m;
    }
}

int spxo_total_threat_features(void) {
    return g_totalThreatFeatures;
}

/* ---------------------------------------------------------------------------------------------------------------
 * Feature indexers.
 * ------------------------------------------------------------------------------------------------------------- */
/* arch.h:53-65: half-board bucket layout, "visually flipped upside down, a1 = 0" */
static const uint8_t kHalfBuckets[32] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 8,  9,  10, 11,
                                         12, 12, 13, 13, 12, 12, 13, 13, 14, 14, 15, 15, 14, 14, 15, 15};

/* KingBucketsMirrored::kBuckets expansion (psq.h:209-226) */
static int kingBucketOfSquare(int sq) {
    const int rank = sq >> 3, file = sq & 7;
    const int f = file < 4 ? file : 7 - file;
    return kHalfBuckets[rank * 4 + f];
}

/* psq::featureIndex<InputFeatureSet> (psq.h:338-365) with KingBucketsMergedMirrored<kAbcd> (psq.h:204-284,317-321) */
uint32_t spxo_psq_index(int c, int piece, int sq, int kingSq) {
    const uint32_t type = (uint32_t)(piece >> 1);
    const uint32_t colour = (type == 5) ? 0u : (((piece & 1) == c) ? 0u : 1u); /* merged kings */
    if (c == 0) {
        sq ^= 56; /* flipRank */
    }
    if ((kingSq & 7) > 3) {
        sq ^= 7; /* shouldFlip (kAbcd): king file > d => flipFile */
    }
    int k = kingSq;
    if (c == 0) {
        k ^= 56;
    }
    const uint32_t bucket = (uint32_t)kingBucketOfSquare(k);
    return bucket * PSQ_INPUT + colour * 384u + type * 64u + (uint32_t)sq;
}

/* threats::threatFeatureIndex (threats.cpp:170-198) */
int32_t spxo_threat_index(int c, int kingSq, int attacker, int asq, int attacked, int vsq) {
    if (c == 0) {
        attacker ^= 1;
        attacked ^= 1;
        asq ^= 56;
        vsq ^= 56;
    }
    if ((kingSq & 7) >= 4) {
        asq ^= 7;
        vsq ^= 7;
    }
    const int forwards = asq < vsq;
    /* INT_MIN + small non-negative terms stays negative: same arithmetic as the reference's i32 sum */
    const int64_t sum = (int64_t)PP_ROWS + g_attackIdx[attacker][attacked][forwards] + (int64_t)g_offsets[attacker][asq] +
                        g_pieceIdx[attacker][asq][vsq];
    return (int32_t)sum;
}

/* threats::ppPawnId / ppFeatureIndex (threats.cpp:200-221) */
static uint32_t ppPawnId(int c, int kingSq, int pawnColour, int sq) {
    if (c == 0) sq ^= 56;
    if ((kingSq & 7) >= 4) sq ^= 7;
    return (uint32_t)((c != pawnColour ? 48 : 0) + sq - 8);
}
uint32_t spxo_pp_index(int c, int kingSq, int aColour, int aSq, int bColour, int bSq) {
    const uint32_t a = ppPawnId(c, kingSq, aColour, aSq), b = ppPawnId(c, kingSq, bColour, bSq);
    const uint32_t hi = a > b ? a : b, lo = a > b ? b : a;
    return hi * (hi - 1) / 2 + lo;
}

/* ---------------------------------------------------------------------------------------------------------------
 * Board: 64-byte mailbox + side to move. FEN placement only (eval ignores castling / ep / clocks).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct {
    uint8_t mailbox[64];
    int stm; /* 0 black, 1 white */
} Board;

static int parseFen(const char* fen, Board* b) {
    memset(b->mailbox, NO_PIECE, 64);
    int rank = 7, file = 0;
    const char* p = fen;
    while (*p == ' ') ++p;
    for (; *p && *p != ' '; ++p) {
        const char ch = *p;
        if (ch == '/') {
            --rank;
            file = 0;
            continue;
        }
        if (ch >= '1' && ch <= '8') {
            file += ch - '0';
            continue;
        }
        static const char* kChars = "pPnNbBrRqQkK"; /* index == piece id */
        const char* at = strchr(kChars, ch);
        if (!at || rank < 0 || file > 7) {
            return -1;
        }
        b->mailbox[rank * 8 + file] = (uint8_t)(at - kChars);
        ++file;
    }
    while (*p == ' ') ++p;
    if (*p != 'w' && *p != 'b') {
        return -1;
    }
    b->stm = (*p == 'w') ? 1 : 0;
    return 0;
}

static void boardSets(const Board* b, uint64_t* occ, uint64_t* kings, uint64_t pawns[2], int kingSq[2]) {
    *occ = 0;
    *kings = 0;
    pawns[0] = pawns[1] = 0;
    kingSq[0] = kingSq[1] = -1;
    for (int sq = 0; sq < 64; ++sq) {
        const int pc = b->mailbox[sq];
        if (pc == NO_PIECE) continue;
        *occ |= 1ull << sq;
        if ((pc >> 1) == 5) {
            *kings |= 1ull << sq;
            kingSq[pc & 1] = sq;
        }
        if ((pc >> 1) == 0) pawns[pc & 1] |= 1ull << sq;
    }
}

/* ---------------------------------------------------------------------------------------------------------------
 * Feature lists for one perspective.
 *   psq:    resetPsqAccumulator (nnue_state.cpp:440-449) - every piece incl. both kings, ascending square order
 *   threat: addThreatFeatures   (nnue_state.cpp:309-354) - threats (from asc, to asc), then pawn pairs
 * Returns counts; psq list capacity 32, threat list capacity 256 (StaticVector<u16,256>, nnue_state.cpp:315).
 * ------------------------------------------------------------------------------------------------------------- */
static int psqRows(const Board* b, int c, int kingSq, uint32_t* out) {
    int n = 0;
    for (int sq = 0; sq < 64; ++sq) {
        const int pc = b->mailbox[sq];
        if (pc != NO_PIECE) out[n++] = spxo_psq_index(c, pc, sq, kingSq);
    }
    return n;
}

static int threatRows(const Board* b, int c, int kingSq, uint64_t occ, uint64_t kings, const uint64_t pawns[2],
                      uint32_t* out) {
    int n = 0;
    uint64_t froms = occ & ~kings;
    while (froms) {
        const int from = lsb(froms);
        froms &= froms - 1;
        const int piece = b->mailbox[from];
        uint64_t tos = occ & attacksOf(piece, from, occ) & ~kings;
        while (tos) {
            const int to = lsb(tos);
            tos &= tos - 1;
            const int32_t f = spxo_threat_index(c, kingSq, piece, from, b->mailbox[to], to);
            if (f >= 0) out[n++] = (uint32_t)f;
        }
    }
    /* pawn pairs (nnue_state.cpp:330-351): ours x (ours-after, theirs-in-mask), then theirs x theirs-after */
    const uint64_t ours = pawns[c], theirs = pawns[c ^ 1];
    uint64_t it = ours;
    while (it) {
        const int a = lsb(it);
        it &= it - 1; /* `remaining` = pawns above a */
        const uint64_t mask = g_ppMask[a];
        uint64_t bs = it & mask;
        while (bs) {
            const int bsq = lsb(bs);
            bs &= bs - 1;
            out[n++] = spxo_pp_index(c, kingSq, c, a, c, bsq);
        }
        bs = theirs & mask;
        while (bs) {
            const int bsq = lsb(bs);
            bs &= bs - 1;
            out[n++] = spxo_pp_index(c, kingSq, c, a, c ^ 1, bsq);
        }
    }
    it = theirs;
    while (it) {
        const int a = lsb(it);
        it &= it - 1;
        uint64_t bs = it & g_ppMask[a];
        while (bs) {
            const int bsq = lsb(bs);
            bs &= bs - 1;
            out[n++] = spxo_pp_index(c, kingSq, c ^ 1, a, c ^ 1, bsq);
        }
    }
    return n;
}

/* ---------------------------------------------------------------------------------------------------------------
 * Accumulators. i16 wrapping adds (input.h:283-293 Accumulator::add; nnue_state.cpp:89-145 applyThreatRows with
 * i8 -> i16 widening and no bias on the threat accumulator).
 * ------------------------------------------------------------------------------------------------------------- */
static void accumulatePsq(const uint32_t* rows, int n, uint16_t* acc) {
    for (int j = 0; j < L1; ++j) acc[j] = (uint16_t)g_net.ftBias[j]; /* initBoth, input.h:72-75 */
    for (int i = 0; i < n; ++i) {
        const int16_t* w = g_net.psqW + (size_t)rows[i] * L1;
        for (int j = 0; j < L1; ++j) acc[j] = (uint16_t)(acc[j] + (uint16_t)w[j]);
    }
}
static void accumulateThreat(const uint32_t* rows, int n, uint16_t* acc) {
    memset(acc, 0, L1 * sizeof(uint16_t)); /* kZeroInit */
    for (int i = 0; i < n; ++i) {
        const int8_t* w = g_net.threatW + (size_t)rows[i] * L1;
        for (int j = 0; j < L1; ++j) acc[j] = (uint16_t)(acc[j] + (uint16_t)(int16_t)w[j]);
    }
}

/* ---------------------------------------------------------------------------------------------------------------
 * Forward pass after the accumulators (arch/multilayer.h:92-490).
 * psq/thr are [2][1024] indexed by colour; `stm` selects the perspective order (nnue_state.cpp:396-438).
 * ------------------------------------------------------------------------------------------------------------- */
static inline int32_t wrap32(uint32_t x) {
    return (int32_t)x;
}
/* arithmetic (floor) shift right, written without relying on implementation-defined >> of negatives */
static inline int32_t sra32(int32_t x, int n) {
    return x >= 0 ? (x >> n) : ~((~x) >> n);
}

int32_t spxo_forward(const uint16_t* psq, const uint16_t* thr, int stm, int bucket, uint8_t* ftOutOpt) {
    uint8_t ft[L1];
    const int order[2] = {stm, stm ^ 1};
    /* activateFt (multilayer.h:92-152) */
    for (int k = 0; k < 2; ++k) {
        const uint16_t* p = psq + order[k] * L1;
        const uint16_t* t = thr + order[k] * L1;
        for (int j = 0; j < PAIRS; ++j) {
            const int16_t a = (int16_t)(uint16_t)(p[j] + t[j]);
            const int16_t bb = (int16_t)(uint16_t)(p[j + PAIRS] + t[j + PAIRS]);
            int32_t i1 = a < 255 ? a : 255; /* min(one) */
            i1 = i1 > 0 ? i1 : 0;           /* max(zero) */
            const int32_t i2 = bb < 255 ? bb : 255; /* min only: i2 is not clamped at zero */
            /* shiftLeftMulHi(i1, i2, 7): mulhi_epi16(i1 << 7, i2) = arithmetic (x*y) >> 16 (floor) */
            const int32_t prod = sra32((i1 << 7) * i2, 16);
            /* packUnsigned: saturate to u8 */
            ft[PAIRS * k + j] = (uint8_t)(prod < 0 ? 0 : (prod > 255 ? 255 : prod));
        }
    }
    if (ftOutOpt) memcpy(ftOutOpt, ft, L1);

    /* propagateL1 (multilayer.h:154-257): dense contraction is identical to the sparse one (zeros contribute 0) */
    int32_t l1o[L2F];
    const int8_t* w1 = g_net.l1W + (size_t)bucket * L1 * L2;
    for (int o = 0; o < L2; ++o) {
        int32_t s = 0; /* |s| <= 1024*127*127 < 2^31: exact */
        for (int k = 0; k < L1; ++k) {
            s += (int32_t)ft[k] * (int32_t)w1[(size_t)(k / 4) * (L2 * 4) + (size_t)o * 4 + (k % 4)];
        }
        const int32_t sh = sra32(s, 2); /* shift<i32, kShift = -2>: arithmetic shift right by 2 (floor) */
        const int32_t t = wrap32((uint32_t)sh + (uint32_t)g_net.l1B[bucket * L2 + o]);
        int32_t c0 = t < 0 ? 0 : (t > 4096 ? 4096 : t);
        c0 = wrap32((uint32_t)c0 << 6);
        const int32_t sq = wrap32((uint32_t)t * (uint32_t)t); /* mullo wraps BEFORE the signed min */
        int32_t c1 = sq < (1 << 24) ? sq : (1 << 24);
        c1 = sra32(c1, 6); /* srai */
        l1o[o] = c0;
        l1o[L2 + o] = c1;
    }

    /* propagateL2 (multilayer.h:261-343): wrapping i32 */
    int32_t l2[L3];
    const int32_t* w2 = g_net.l2W + (size_t)bucket * L2F * L3;
    for (int o = 0; o < L3; ++o) {
        uint32_t acc = (uint32_t)g_net.l2B[bucket * L3 + o];
        for (int i = 0; i < L2F; ++i) {
            const int32_t in = sra32(l1o[i], 6);
            acc += (uint32_t)in * (uint32_t)w2[i * L3 + o];
        }
        l2[o] = wrap32(acc);
    }

    /* propagateL3 (multilayer.h:345-447): clamp to [0, Q^3], add skipped (unshifted) L1 output, wrapping */
    uint32_t acc3 = (uint32_t)g_net.l3B[bucket];
    const int32_t* w3 = g_net.l3W + (size_t)bucket * L3;
    for (int i = 0; i < L3; ++i) {
        int32_t v = l2[i] < 0 ? 0 : (l2[i] > 262144 ? 262144 : l2[i]);
        const uint32_t in = (uint32_t)v + (uint32_t)l1o[i];
        acc3 += in * (uint32_t)w3[i];
    }
    const int32_t l3 = wrap32(acc3);

    /* propagate tail (multilayer.h:484-489): i64 multiply, C++ truncating division */
    int64_t out = (int64_t)l3;
    out *= 400;
    out /= (int64_t)64 * 64 * 64 * 64;
    return (int32_t)out;
}

/* ---------------------------------------------------------------------------------------------------------------
 * Public API (ctypes): load net, evaluate FENs / mailboxes, dump feature lists.
 * ------------------------------------------------------------------------------------------------------------- */
int spxo_init(const void* blob, size_t n) {
    const size_t need = 89381984;
    if (!blob || n < need) return -1;
    const unsigned char* b = (const unsigned char*)blob;
    if (memcmp(b, "CBNF", 4) != 0) return -2;
    free(g_blob);
    g_blob = (unsigned char*)malloc(need);
    if (!g_blob) return -3;
    memcpy(g_blob, b, need);
    size_t off = 64;
    g_net.psqW = (const int16_t*)(g_blob + off);
    off += (size_t)PSQ_ROWS * L1 * 2;
    g_net.threatW = (const int8_t*)(g_blob + off);
    off += (size_t)THREAT_ROWS * L1;
    g_net.ftBias = (const int16_t*)(g_blob + off);
    off += L1 * 2;
    g_net.l1W = (const int8_t*)(g_blob + off);
    off += (size_t)NBUCKETS * L1 * L2;
    g_net.l1B = (const int32_t*)(g_blob + off);
    off += NBUCKETS * L2 * 4;
    g_net.l2W = (const int32_t*)(g_blob + off);
    off += (size_t)NBUCKETS * L2F * L3 * 4;
    g_net.l2B = (const int32_t*)(g_blob + off);
    off += NBUCKETS * L3 * 4;
    g_net.l3W = (const int32_t*)(g_blob + off);
    off += NBUCKETS * L3 * 4;
    g_net.l3B = (const int32_t*)(g_blob + off);
    initAttackTables();
    initThreatLuts();
    return g_totalThreatFeatures == 59808 ? 0 : -4;
}

/* Feature rows of one perspective of a mailbox position. Returns 0, fills counts. */
int spxo_features_mailbox(const uint8_t* mailbox, int c, uint32_t* psqOut, int* nPsq, uint32_t* thrOut, int* nThr) {
    Board b;
    memcpy(b.mailbox, mailbox, 64);
    b.stm = 1;
    uint64_t occ, kings, pawns[2];
    int kingSq[2];
    boardSets(&b, &occ, &kings, pawns, kingSq);
    if (kingSq[0] < 0 || kingSq[1] < 0) return -1;
    *nPsq = psqRows(&b, c, kingSq[c], psqOut);
    *nThr = threatRows(&b, c, kingSq[c], occ, kings, pawns, thrOut);
    return 0;
}

/* NnueState::evaluateOnce (nnue_state.cpp:612-634) on a mailbox; bucket per output.h:51-54 */
int spxo_eval_mailbox(const uint8_t* mailbox, int stm, int32_t* out) {
    Board b;
    memcpy(b.mailbox, mailbox, 64);
    b.stm = stm;
    uint64_t occ, kings, pawns[2];
    int kingSq[2];
    boardSets(&b, &occ, &kings, pawns, kingSq);
    if (kingSq[0] < 0 || kingSq[1] < 0) return -1;
    uint16_t psq[2 * L1], thr[2 * L1];
    uint32_t rows[256];
    for (int c = 0; c < 2; ++c) {
        int n = psqRows(&b, c, kingSq[c], rows);
        accumulatePsq(rows, n, psq + c * L1);
        n = threatRows(&b, c, kingSq[c], occ, kings, pawns, rows);
        accumulateThreat(rows, n, thr + c * L1);
    }
    const int bucket = (popcnt(occ) - 2) / 4;
    *out = spxo_forward(psq, thr, stm, bucket, NULL);
    return 0;
}

/* activateFt output (u8[1024], stm half first) of a mailbox position - lets tests localise a GPU mismatch to the
 * feature-transformer kernel or to the MLP kernel. */
int spxo_ft_mailbox(const uint8_t* mailbox, int stm, uint8_t* ftOut) {
    Board b;
    memcpy(b.mailbox, mailbox, 64);
    b.stm = stm;
    uint64_t occ, kings, pawns[2];
    int kingSq[2];
    boardSets(&b, &occ, &kings, pawns, kingSq);
    if (kingSq[0] < 0 || kingSq[1] < 0) return -1;
    uint16_t psq[2 * L1], thr[2 * L1];
    uint32_t rows[256];
    for (int c = 0; c < 2; ++c) {
        int n = psqRows(&b, c, kingSq[c], rows);
        accumulatePsq(rows, n, psq + c * L1);
        n = threatRows(&b, c, kingSq[c], occ, kings, pawns, rows);
        accumulateThreat(rows, n, thr + c * L1);
    }
    (void)spxo_forward(psq, thr, stm, (popcnt(occ) - 2) / 4, ftOut);
    return 0;
}

int spxo_eval_fen(const char* fen, int32_t* out) {
    Board b;
    if (parseFen(fen, &b) != 0) return -1;
    return spxo_eval_mailbox(b.mailbox, b.stm, out);
}

int spxo_fen_to_mailbox(const char* fen, uint8_t* mailbox, int* stm) {
    Board b;
    if (parseFen(fen, &b) != 0) return -1;
    memcpy(mailbox, b.mailbox, 64);
    *stm = b.stm;
    return 0;
}

/* Batch evaluation of n mailboxes (64 B each) - used as the bounded CPU baseline ("port") and by parity tests. */
int spxo_eval_mailboxes(const uint8_t* mailboxes, const uint8_t* stm, size_t n, int32_t* out) {
    for (size_t i = 0; i < n; ++i) {
        if (spxo_eval_mailbox(mailboxes + 64 * i, stm[i], &out[i]) != 0) return -1;
    }
    return 0;
}

/* Evaluate from explicit accumulators (for incremental-path checks): psq/thr are [2][1024] i16 by colour. */
int32_t spxo_eval_accumulators(const int16_t* psq, const int16_t* thr, int stm, int bucket) {
    return spxo_forward((const uint16_t*)psq, (const uint16_t*)thr, stm, bucket, NULL);
}

/* Accumulate explicit row lists (for incremental checks): out = bias(if psq) + sum rows */
void spxo_accumulate_rows(const uint32_t* psqRowsIn, int nPsq, const uint32_t* thrRowsIn, int nThr, int16_t* psqOut,
                          int16_t* thrOut) {
    accumulatePsq(psqRowsIn, nPsq, (uint16_t*)psqOut);
    accumulateThreat(thrRowsIn, nThr, (uint16_t*)thrOut);
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Host-scalar post-processing of a raw eval (eval.cpp:24-67):
 *   adjustStatic  (eval.cpp:24-27)   eval += contempt[stm]; clamp to +-(kScoreWin - 1), kScoreWin = 25000 (core.h:708)
 *   adjustEval    (eval.cpp:30-67)   material scaling + optimism, halfmove damping, optional correction / 2048, clamp
 * params: [0..1] contempt (black, white), [2..3] optimism, [4..8] scalingValue{Pawn,Knight,Bishop,Rook,Queen}
 * (tunable.h:161-165), [9] materialScalingBase, [10] optimismBase, [11] optimismMaterialScale (tunable.h:167-169).
 * stages: bit 0 = adjustStatic, bit 1 = adjustEval. i32 arithmetic as in the reference (done unsigned where it could
 * wrap), C division truncating toward zero.
 * ------------------------------------------------------------------------------------------------------------------- */
static int32_t clampScore(int32_t v) {
    return v < -24999 ? -24999 : (v > 24999 ? 24999 : v);
}
static int32_t wrapMul(int32_t a, int32_t b) {
    return (int32_t)((uint32_t)a * (uint32_t)b);
}
static int32_t wrapAdd(int32_t a, int32_t b) {
    return (int32_t)((uint32_t)a + (uint32_t)b);
}

int32_t spxo_adjust(const uint8_t* mailbox, int stm, int halfmove, const int32_t* params, uint32_t stages,
                    int hasCorrection, int32_t correction, int32_t eval) {
    if (stages & 1u) {
        eval = clampScore(wrapAdd(eval, params[stm]));
    }
    if (stages & 2u) {
        int32_t npMaterial = 0;
        for (int sq = 0; sq < 64; ++sq) {
            const int piece = mailbox[sq];
            if (piece != NO_PIECE && (piece >> 1) < 5) {
                npMaterial += params[4 + (piece >> 1)];
            }
        }
        const int32_t a = wrapMul(eval, wrapAdd(params[9], npMaterial));
        const int32_t b = wrapMul(params[2 + stm], wrapAdd(params[10], wrapMul(npMaterial, params[11]) / 1024));
        eval = wrapAdd(a, b) / 32768;
        eval = wrapMul(eval, 200 - halfmove) / 200;
        if (hasCorrection) {
            eval = wrapAdd(eval, correction / 2048);
        }
        eval = clampScore(eval);
    }
    return eval;
}

/* wdl::normalizeScore<false> (src/wdl.cpp:28-79; wdlParams :29-40): score / a(material) * 100, rounded half away from
 * zero; zero and decisive scores (|score| > kScoreWin = 25000, core.h:708,722-724) pass through. `material` =
 * Position::classicalMaterial (position.h:515-521). Plain double arithmetic in the reference's expression order. */
int32_t spxo_wdl_normalize(int32_t score, int32_t material) {
    if (score == 0 || score > 25000 || score < -25000) {
        return score;
    }
    const double m = (double)(material < 17 ? 17 : (material > 78 ? 78 : material)) / 58.0;
    const double a = ((-244.97139595 * m + 687.39969858) * m + -654.38002091) * m + 608.47087786;
    return (int32_t)round(100.0 * ((double)score / a));
}

/* Position::classicalMaterial (position.h:515-521) of a 64-square mailbox */
int32_t spxo_classical_material(const uint8_t* mailbox) {
    static const int kValue[6] = {1, 3, 3, 5, 9, 0};
    int32_t material = 0;
    for (int sq = 0; sq < 64; ++sq) {
        if (mailbox[sq] != NO_PIECE) {
            material += kValue[mailbox[sq] >> 1];
        }
    }
    return material;
}
