// Host chess core (see spx_chess.h). Plain bitboard + mailbox board with legality by make-and-test; speed is
// irrelevant here (it only feeds position batches to the GPU evaluator and replays recorded traces).
#include "spx_chess.h"

#include <cstdlib>
#include <cstring>

#include "../../include/spx_nnue.h"
#include "../../include/spx_nnue_dev.h"
#include "spx_device_math.h"

namespace spx {

namespace {
uint64_t kingAttacksBb(uint64_t b) {
    const uint64_t row = b | ((b << 1) & ~kFileA) | ((b >> 1) & ~kFileH);
    return (row | (row << 8) | (row >> 8)) & ~b;
}

struct SplitMix64 {
    uint64_t s;
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    uint32_t below(uint32_t n) {
        return uint32_t((next() >> 32) % n);
    }
};

const char kPieceChars[] = "pPnNbBrRqQkK";  // index == piece id
}  // namespace

void Board::clear() {
    std::memset(mailbox, kNoPiece, sizeof(mailbox));
    std::memset(pieces, 0, sizeof(pieces));
    colour[0] = colour[1] = occ = 0;
    kingSq[0] = kingSq[1] = -1;
    castleRook[0][0] = castleRook[0][1] = castleRook[1][0] = castleRook[1][1] = -1;
    ep = -1;
    stm = 1;
    halfmove = 0;
    fullmove = 1;
}

void Board::put(int piece, int sq) {
    const uint64_t b = 1ull << sq;
    mailbox[sq] = uint8_t(piece);
    pieces[piece] |= b;
    colour[piece & 1] |= b;
    occ |= b;
    if ((piece >> 1) == 5) {
        kingSq[piece & 1] = int8_t(sq);
    }
}

void Board::remove(int sq) {
    const int piece = mailbox[sq];
    const uint64_t b = 1ull << sq;
    mailbox[sq] = kNoPiece;
    pieces[piece] &= ~b;
    colour[piece & 1] &= ~b;
    occ &= ~b;
}

bool Board::attacked(int sq, int by, uint64_t occupancy) const {
    const uint64_t bit = 1ull << sq;
    // a pawn of colour `by` attacks sq iff a pawn of the other colour standing on sq would attack it
    if (pawnAttacks(bit, by ^ 1) & pieces[0 | by]) return true;
    if (knightAttacks(bit) & pieces[2 | by]) return true;
    if (kingAttacksBb(bit) & pieces[10 | by]) return true;
    const uint64_t diag = lineAttacks(occupancy, bit, diagMask(sq)) | lineAttacks(occupancy, bit, antiMask(sq));
    if (diag & (pieces[4 | by] | pieces[8 | by])) return true;
    const uint64_t orth = lineAttacks(occupancy, bit, fileMask(sq)) | lineAttacks(occupancy, bit, rankMask(sq));
    return (orth & (pieces[6 | by] | pieces[8 | by])) != 0;
}

// Position::filterEp (position.cpp:1608-1683): the en-passant square survives only while an en passant capture is LEGAL
// (capturer not pinned, no discovered check along the rank, no other checker). Here: make each of the <= 2 candidate
// captures and test the king, the way generateLegal() decides legality.
static void filterEp(Board& b) {
    if (b.ep < 0) return;
    const int us = b.stm, them = us ^ 1;
    uint64_t capturers = pawnAttacks(1ull << b.ep, them) & b.pieces[0 | us];
    bool legal = false;
    while (capturers && !legal) {
        const int from = ctz64(capturers);
        capturers &= capturers - 1;
        Board next = b;
        makeMove(next, Move{uint8_t(from), uint8_t(b.ep), kEnPassant, 0});
        legal = !next.attacked(next.kingSq[us], them, next.occ);
    }
    if (!legal) b.ep = -1;
}

bool boardFromFen(const char* fen, Board& b) {
    b.clear();
    if (!fen) return false;
    const char* p = fen;
    while (*p == ' ') ++p;
    int rank = 7, file = 0;
    for (; *p && *p != ' '; ++p) {
        const char ch = *p;
        if (ch == '/') {
            --rank;
            file = 0;
        } else if (ch >= '1' && ch <= '8') {
            file += ch - '0';
            if (file > 8) return false;
        } else {
            const char* at = std::strchr(kPieceChars, ch);
            if (!at || rank < 0 || file > 7) return false;
            b.put(int(at - kPieceChars), rank * 8 + file);
            ++file;
        }
    }
    // a packed record holds 32 pieces (marlinformat.h:36) and the evaluator expects exactly one king per colour
    if (popc64(b.occ) > 32 || popc64(b.pieces[10]) != 1 || popc64(b.pieces[11]) != 1) return false;
    if (b.kingSq[0] < 0 || b.kingSq[1] < 0) return false;
    while (*p == ' ') ++p;
    if (*p != 'w' && *p != 'b') return false;
    b.stm = (*p == 'w');
    ++p;
    while (*p == ' ') ++p;
    // castling: KQkq, Shredder-FEN file letters (HAha) or '-'
    for (; *p && *p != ' '; ++p) {
        const char ch = *p;
        if (ch == '-') continue;
        const int c = (ch >= 'A' && ch <= 'Z') ? 1 : 0;
        const int ksq = b.kingSq[c];
        const int base = c ? 0 : 56;
        if ((ksq >> 3) != (base >> 3)) continue;
        const char lc = char(ch | 0x20);
        int rookSq = -1;
        if (lc == 'k') {
            for (int s = base + 7; s > ksq; --s)
                if (b.mailbox[s] == (6 | c)) {
                    rookSq = s;
                    break;
                }
        } else if (lc == 'q') {
            for (int s = base; s < ksq; ++s)
                if (b.mailbox[s] == (6 | c)) {
                    rookSq = s;
                    break;
                }
        } else if (lc >= 'a' && lc <= 'h') {
            const int s = base + (lc - 'a');
            if (b.mailbox[s] == (6 | c)) rookSq = s;
        }
        if (rookSq >= 0) b.castleRook[c][rookSq > ksq ? 0 : 1] = int8_t(rookSq);
    }
    while (*p == ' ') ++p;
    if (*p && *p != '-') {
        if (p[0] >= 'a' && p[0] <= 'h' && p[1] >= '1' && p[1] <= '8') {
            b.ep = int8_t((p[1] - '1') * 8 + (p[0] - 'a'));
            // only a double-pushed enemy pawn in front of an empty target square can be captured en passant
            const int rank = b.ep >> 3, pushed = b.ep + (b.stm ? -8 : 8);
            if (rank != (b.stm ? 5 : 2) || b.mailbox[b.ep] != kNoPiece || b.mailbox[pushed] != (0 | (b.stm ^ 1))) b.ep = -1;
            filterEp(b);
        }
    }
    while (*p && *p != ' ') ++p;
    while (*p == ' ') ++p;
    if (*p) {
        b.halfmove = uint16_t(std::strtoul(p, nullptr, 10));
        while (*p && *p != ' ') ++p;
        while (*p == ' ') ++p;
        if (*p) b.fullmove = uint16_t(std::strtoul(p, nullptr, 10));
    }
    return true;
}

std::string boardToFen(const Board& b) {
    std::string s;
    for (int rank = 7; rank >= 0; --rank) {
        int empty = 0;
        for (int file = 0; file < 8; ++file) {
            const int pc = b.mailbox[rank * 8 + file];
            if (pc == kNoPiece) {
                ++empty;
            } else {
                if (empty) s += char('0' + empty);
                empty = 0;
                s += kPieceChars[pc];
            }
        }
        if (empty) s += char('0' + empty);
        if (rank) s += '/';
    }
    s += b.stm ? " w " : " b ";
    std::string rights;
    for (int c : {1, 0}) {
        for (int side = 0; side < 2; ++side) {
            const int r = b.castleRook[c][side];
            if (r >= 0) rights += char((c ? 'A' : 'a') + (r & 7));  // Shredder-FEN: unambiguous for Chess960
        }
    }
    s += rights.empty() ? "-" : rights;
    s += ' ';
    if (b.ep >= 0) {
        s += char('a' + (b.ep & 7));
        s += char('1' + (b.ep >> 3));
    } else {
        s += '-';
    }
    s += ' ' + std::to_string(b.halfmove) + ' ' + std::to_string(b.fullmove);
    return s;
}

Board startpos() {
    Board b;
    boardFromFen("rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1", b);
    return b;
}

// Scharnagl's Chess960 numbering -> back rank piece types
static void backRank960(uint32_t n, int out[8]) {
    for (int i = 0; i < 8; ++i) out[i] = -1;
    const uint32_t n2 = n / 4, b1 = n % 4;
    const uint32_t n3 = n2 / 4, b2 = n2 % 4;
    const uint32_t n4 = n3 / 6, q = n3 % 6;
    out[b1 * 2 + 1] = 2;  // light-squared bishop on b, d, f, h
    out[b2 * 2] = 2;      // dark-squared bishop on a, c, e, g
    auto nthEmpty = [&](uint32_t k) {
        for (int i = 0; i < 8; ++i)
            if (out[i] < 0 && k-- == 0) return i;
        return -1;
    };
    out[nthEmpty(q)] = 4;
    static const int kKnights[10][2] = {{0, 1}, {0, 2}, {0, 3}, {0, 4}, {1, 2}, {1, 3}, {1, 4}, {2, 3}, {2, 4}, {3, 4}};
    const int k0 = nthEmpty(uint32_t(kKnights[n4][0]));
    const int k1 = nthEmpty(uint32_t(kKnights[n4][1]));
    out[k0] = 1;
    out[k1] = 1;
    out[nthEmpty(0)] = 3;
    out[nthEmpty(0)] = 5;
    out[nthEmpty(0)] = 3;
}

Board dfrcStart(uint32_t whiteIdx, uint32_t blackIdx) {
    Board b;
    b.clear();
    int w[8], k[8];
    backRank960(whiteIdx % 960, w);
    backRank960(blackIdx % 960, k);
    for (int f = 0; f < 8; ++f) {
        b.put((w[f] << 1) | 1, f);
        b.put(0 | 1, 8 + f);
        b.put((k[f] << 1) | 0, 56 + f);
        b.put(0 | 0, 48 + f);
    }
    for (int c = 0; c < 2; ++c) {
        const int base = c ? 0 : 56;
        for (int f = 0; f < 8; ++f) {
            if (b.mailbox[base + f] == (6 | c)) {
                b.castleRook[c][(base + f) > b.kingSq[c] ? 0 : 1] = int8_t(base + f);
            }
        }
    }
    return b;
}

// ---- move generation ----
static void addPawnMoves(std::vector<Move>& out, int from, int to, bool promo) {
    if (promo) {
        for (uint8_t pt : {4, 3, 2, 1}) out.push_back({uint8_t(from), uint8_t(to), kPromotion, pt});
    } else {
        out.push_back({uint8_t(from), uint8_t(to), kNormal, 0});
    }
}

static void generatePseudo(const Board& b, std::vector<Move>& out) {
    const int us = b.stm, them = us ^ 1;
    const uint64_t own = b.colour[us], enemy = b.colour[them];
    // pawns
    uint64_t pawns = b.pieces[0 | us];
    while (pawns) {
        const int from = ctz64(pawns);
        pawns &= pawns - 1;
        const int fwd = us ? 8 : -8;
        const int one = from + fwd;
        const bool promo = us ? (one >= 56) : (one < 8);
        if (one >= 0 && one < 64 && !(b.occ >> one & 1)) {
            addPawnMoves(out, from, one, promo);
            const bool home = us ? ((from >> 3) == 1) : ((from >> 3) == 6);
            if (home && !(b.occ >> (one + fwd) & 1)) out.push_back({uint8_t(from), uint8_t(one + fwd), kNormal, 0});
        }
        uint64_t caps = pawnAttacks(1ull << from, us) & enemy;
        while (caps) {
            const int to = ctz64(caps);
            caps &= caps - 1;
            addPawnMoves(out, from, to, promo);
        }
        if (b.ep >= 0 && (pawnAttacks(1ull << from, us) >> b.ep & 1)) {
            out.push_back({uint8_t(from), uint8_t(b.ep), kEnPassant, 0});
        }
    }
    for (int type = 1; type <= 5; ++type) {
        uint64_t bb = b.pieces[(type << 1) | us];
        while (bb) {
            const int from = ctz64(bb);
            bb &= bb - 1;
            uint64_t att = (type == 5) ? kingAttacksBb(1ull << from) : pieceAttacks((type << 1) | us, from, b.occ);
            att &= ~own;
            while (att) {
                const int to = ctz64(att);
                att &= att - 1;
                out.push_back({uint8_t(from), uint8_t(to), kNormal, 0});
            }
        }
    }
    // castling (Chess960 rules): king ends on g/c, rook on f/d; every square either piece crosses or lands on must
    // be empty apart from the two of them; the king may not start on, cross or land on an attacked square.
    const int ksq = b.kingSq[us];
    for (int side = 0; side < 2; ++side) {
        const int rsq = b.castleRook[us][side];
        if (rsq < 0) continue;
        const int base = us ? 0 : 56;
        const int kTo = base + (side == 0 ? 6 : 2), rTo = base + (side == 0 ? 5 : 3);
        auto between = [](int a, int c) {
            uint64_t m = 0;
            const int lo = a < c ? a : c, hi = a < c ? c : a;
            for (int s = lo; s <= hi; ++s) m |= 1ull << s;
            return m;
        };
        const uint64_t span = between(ksq, kTo) | between(rsq, rTo);
        const uint64_t others = b.occ & ~(1ull << ksq) & ~(1ull << rsq);
        if (span & others) continue;
        bool safe = true;
        const uint64_t occNoKR = others;  // attack rays see through the castling king and rook
        uint64_t path = between(ksq, kTo);
        while (path && safe) {
            const int s = ctz64(path);
            path &= path - 1;
            safe = !b.attacked(s, them, occNoKR | (1ull << rsq));
        }
        // the rook itself may have been shielding the king's destination from a slider on the back rank
        if (safe && b.attacked(kTo, them, occNoKR | (1ull << rTo))) safe = false;
        if (safe) out.push_back({uint8_t(ksq), uint8_t(rsq), kCastling, 0});
    }
}

void makeMove(Board& b, const Move& m) {
    const int us = b.stm, them = us ^ 1;
    const int moving = b.mailbox[m.from];
    const int movingType = moving >> 1;
    int captured = kNoPiece;
    const int oldEp = b.ep;
    (void)oldEp;
    b.ep = -1;
    if (m.kind == kCastling) {
        const int base = us ? 0 : 56;
        const int side = m.to > m.from ? 0 : 1;
        const int kTo = base + (side == 0 ? 6 : 2), rTo = base + (side == 0 ? 5 : 3);
        b.remove(m.from);
        b.remove(m.to);
        b.put(10 | us, kTo);
        b.put(6 | us, rTo);
    } else {
        if (m.kind == kEnPassant) {
            const int capSq = m.to + (us ? -8 : 8);
            captured = b.mailbox[capSq];
            b.remove(capSq);
        } else if (b.mailbox[m.to] != kNoPiece) {
            captured = b.mailbox[m.to];
            b.remove(m.to);
        }
        b.remove(m.from);
        b.put(m.kind == kPromotion ? ((m.promo << 1) | us) : moving, m.to);
        if (movingType == 0 && std::abs(int(m.to) - int(m.from)) == 16) {
            // the ep square is kept only if an enemy pawn can LEGALLY capture there (filterEp below, run once the side to
            // move has flipped) - as the reference's Position::filterEp: records and repetition keys then agree with it
            const int epSq = (m.from + m.to) / 2;
            if (pawnAttacks(1ull << epSq, us) & b.pieces[0 | them]) b.ep = int8_t(epSq);
        }
    }
    if (movingType == 5) {
        b.castleRook[us][0] = b.castleRook[us][1] = -1;
    } else if (movingType == 3) {
        for (int side = 0; side < 2; ++side)
            if (b.castleRook[us][side] == m.from) b.castleRook[us][side] = -1;
    }
    if (captured != kNoPiece && (captured >> 1) == 3) {
        for (int side = 0; side < 2; ++side)
            if (b.castleRook[them][side] == m.to) b.castleRook[them][side] = -1;
    }
    b.halfmove = (captured == kNoPiece && movingType != 0) ? uint16_t(b.halfmove + 1) : uint16_t(0);
    if (us == 0) ++b.fullmove;
    b.stm = uint8_t(them);
    filterEp(b);
}

// ---------------------------------------------------------------------------------------------------------------------
// Delta capture (scalar restatement of the reference's SIMD "ray geometry").
// Focus square f: 8 rays (N, NE, E, SE, S, SW, W, NW); `closest` = first occupied square along each ray plus every
// occupied knight-jump square (geometry_avx2.h:134-142). All functions look at `mb` exactly as it is at the moment of
// the observer hook; `ignore` is a square treated as empty (permuteMailbox(..., ignore), geometry_avx2.h:114-132).
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int kRayDf[8] = {0, 1, 1, 1, 0, -1, -1, -1};
constexpr int kRayDr[8] = {1, 1, 0, -1, -1, -1, 0, 1};
constexpr int kJumpDf[8] = {-1, 1, 2, 2, 1, -1, -2, -2};
constexpr int kJumpDr[8] = {2, 2, 1, -1, -2, -2, -1, 1};

struct FocusGeometry {
    int raySq[8], rayDist[8];  // closest occupied square per ray (-1: none) and its distance
    int jumpSq[8];             // occupied knight-jump squares (-1: empty / off board)
};

void focusGeometry(const uint8_t* mb, int f, int ignore, FocusGeometry& g) {
    const int ff = f & 7, fr = f >> 3;
    for (int r = 0; r < 8; ++r) {
        g.raySq[r] = -1;
        g.rayDist[r] = 0;
        int file = ff, rank = fr;
        for (int d = 1; d < 8; ++d) {
            file += kRayDf[r];
            rank += kRayDr[r];
            if (file < 0 || file > 7 || rank < 0 || rank > 7) break;
            const int sq = rank * 8 + file;
            if (sq != ignore && mb[sq] != kNoPiece) {
                g.raySq[r] = sq;
                g.rayDist[r] = d;
                break;
            }
        }
        const int jf = ff + kJumpDf[r], jr = fr + kJumpDr[r];
        g.jumpSq[r] = -1;
        if (jf >= 0 && jf < 8 && jr >= 0 && jr < 8) {
            const int sq = jr * 8 + jf;
            if (sq != ignore && mb[sq] != kNoPiece) g.jumpSq[r] = sq;
        }
    }
}

// outgoingThreats (geometry.h:89-104): squares among `closest` the focus piece attacks
int outgoingSquares(int piece, const FocusGeometry& g, int out[8]) {
    int n = 0;
    const int type = piece >> 1;
    if (type == 0) {
        const int a = (piece & 1) ? 1 : 3, b = (piece & 1) ? 7 : 5;  // white: NE, NW; black: SE, SW
        if (g.raySq[a] >= 0 && g.rayDist[a] == 1) out[n++] = g.raySq[a];
        if (g.raySq[b] >= 0 && g.rayDist[b] == 1) out[n++] = g.raySq[b];
    } else if (type == 1) {
        for (int i = 0; i < 8; ++i)
            if (g.jumpSq[i] >= 0) out[n++] = g.jumpSq[i];
    } else if (type <= 4) {
        for (int r = 0; r < 8; ++r) {
            const bool diag = r & 1;
            if ((diag && type == 3) || (!diag && type == 2)) continue;
            if (g.raySq[r] >= 0) out[n++] = g.raySq[r];
        }
    }
    return n;  // kings: none ("Ignore king threats")
}

// can the piece standing on ray r at distance d attack the focus square? (kIncomingThreatsMask, geometry.h:106-125)
bool rayAttacker(int piece, int r, int d, bool slidersOnly) {
    const int type = piece >> 1;
    const bool diag = r & 1;
    if (type == 4 || (diag && type == 2) || (!diag && type == 3)) return true;
    if (slidersOnly || type != 0 || d != 1 || !diag) return false;
    const bool above = (r == 1 || r == 7);       // NE / NW of the focus square
    return above ? (piece & 1) == 0 : (piece & 1) == 1;  // black pawns attack downwards, white pawns upwards
}

struct Emitter {
    MoveDelta& d;
    void add(int a, int asq, int v, int vsq) {
        d.threatsAdded.push_back({uint8_t(a), uint8_t(asq), uint8_t(v), uint8_t(vsq)});
    }
    void remove(int a, int asq, int v, int vsq) {
        d.threatsRemoved.push_back({uint8_t(a), uint8_t(asq), uint8_t(v), uint8_t(vsq)});
    }
    void emit(bool isAdd, int a, int asq, int v, int vsq) {
        isAdd ? add(a, asq, v, vsq) : remove(a, asq, v, vsq);
    }
};

void focusThreats(Emitter& e, const uint8_t* mb, const FocusGeometry& g, int piece, int f, bool isAdd, bool outgoing,
                  bool incoming) {
    if (outgoing) {
        int out[8];
        const int n = outgoingSquares(piece, g, out);
        for (int i = 0; i < n; ++i) e.emit(isAdd, piece, f, mb[out[i]], out[i]);
    }
    if (incoming) {
        for (int i = 0; i < 8; ++i) {
            if (g.jumpSq[i] >= 0 && (mb[g.jumpSq[i]] >> 1) == 1) e.emit(isAdd, mb[g.jumpSq[i]], g.jumpSq[i], piece, f);
            if (g.raySq[i] >= 0 && rayAttacker(mb[g.raySq[i]], i, g.rayDist[i], false)) {
                e.emit(isAdd, mb[g.raySq[i]], g.raySq[i], piece, f);
            }
        }
    }
}

// x-ray threats through the focus square: slider on ray r, closest piece on the opposite ray (nnue.cpp:380-415,450-483)
// pieceAddedAtFocus = true retracts them (removed), false extends them (added)
void discoveredThreats(Emitter& e, const uint8_t* mb, const FocusGeometry& g, bool pieceAddedAtFocus) {
    for (int r = 0; r < 8; ++r) {
        const int s = g.raySq[r], v = g.raySq[(r + 4) & 7];
        if (s < 0 || v < 0 || !rayAttacker(mb[s], r, g.rayDist[r], true)) continue;
        e.emit(!pieceAddedAtFocus, mb[s], s, mb[v], v);
    }
}

// updatePieceThreatsOnChange<kAdd> (nnue.cpp:490-523)
void threatsOnChange(Emitter& e, const uint8_t* mb, bool isAdd, int piece, int f) {
    FocusGeometry g;
    focusGeometry(mb, f, -1, g);
    focusThreats(e, mb, g, piece, f, isAdd, true, true);
    discoveredThreats(e, mb, g, isAdd);
}
// updatePieceThreatsOnMutate (nnue.cpp:525-549): no x-ray change, the square stays occupied
void threatsOnMutate(Emitter& e, const uint8_t* mb, int oldPiece, int newPiece, int f) {
    FocusGeometry g;
    focusGeometry(mb, f, -1, g);
    focusThreats(e, mb, g, oldPiece, f, false, true, false);
    focusThreats(e, mb, g, newPiece, f, true, true, false);
    focusThreats(e, mb, g, oldPiece, f, false, false, true);
    focusThreats(e, mb, g, newPiece, f, true, false, true);
}
// updatePieceThreatsOnMove (nnue.cpp:551-599): src side sees dst as empty, dst side sees the real board
void threatsOnMove(Emitter& e, const uint8_t* mb, int oldPiece, int src, int newPiece, int dst) {
    FocusGeometry gs, gd;
    focusGeometry(mb, src, dst, gs);
    focusGeometry(mb, dst, -1, gd);
    focusThreats(e, mb, gs, oldPiece, src, false, true, false);
    focusThreats(e, mb, gd, newPiece, dst, true, true, false);
    focusThreats(e, mb, gs, oldPiece, src, false, false, true);
    focusThreats(e, mb, gd, newPiece, dst, true, false, true);
    discoveredThreats(e, mb, gs, false);
    discoveredThreats(e, mb, gd, true);
}

// KingBucketsMirrored::refreshRequired (psq.h:264-283) with the arch.h:53-65 bucket layout
bool psqRefreshRequired(int c, int prevKing, int king) {
    if (((prevKing & 7) > 3) != ((king & 7) > 3)) return true;
    if (c == 0) {
        prevKing ^= 56;
        king ^= 56;
    }
    return kingBucket(prevKing) != kingBucket(king);
}
}  // namespace

void makeMoveObserved(Board& b, const Move& m, MoveDelta& d) {
    d = MoveDelta{};
    Emitter e{d};
    const int us = b.stm;
    const int moving = b.mailbox[m.from];
    d.pawnsBefore[0] = b.pieces[0];
    d.pawnsBefore[1] = b.pieces[1];
    uint8_t mb[64];
    std::memcpy(mb, b.mailbox, 64);
    auto pushSub = [&](int piece, int sq) {
        d.subPiece[d.nSub] = uint8_t(piece);
        d.subSq[d.nSub++] = uint8_t(sq);
    };
    auto pushAdd = [&](int piece, int sq) {
        d.addPiece[d.nAdd] = uint8_t(piece);
        d.addSq[d.nAdd++] = uint8_t(sq);
    };
    auto prepareKingMove = [&](int c, int src, int dst) {  // nnue_state.h:118-128
        if (psqRefreshRequired(c, src, dst)) d.psqRefresh[c] = true;
        if (((src & 7) >= 4) != ((dst & 7) >= 4)) d.threatRefresh[c] = true;
    };

    if (m.kind == kCastling) {  // position.cpp:1396-1437
        const int base = us ? 0 : 56;
        const bool shortSide = (m.from & 7) < (m.to & 7);
        const int kDst = base + (shortSide ? 6 : 2), rDst = base + (shortSide ? 5 : 3);
        const int king = 10 | us, rook = 6 | us;
        prepareKingMove(us, m.from, kDst);
        mb[m.from] = kNoPiece;
        pushSub(king, m.from);
        threatsOnChange(e, mb, false, king, m.from);
        mb[m.to] = kNoPiece;
        pushSub(rook, m.to);
        threatsOnChange(e, mb, false, rook, m.to);
        mb[kDst] = uint8_t(king);
        pushAdd(king, kDst);
        threatsOnChange(e, mb, true, king, kDst);
        mb[rDst] = uint8_t(rook);
        pushAdd(rook, rDst);
        threatsOnChange(e, mb, true, rook, rDst);
    } else if (m.kind == kEnPassant) {  // position.cpp:1439-1466
        const int capSq = m.to ^ 8;  // flipRankParity
        const int enemyPawn = moving ^ 1;
        mb[capSq] = kNoPiece;
        pushSub(enemyPawn, capSq);
        threatsOnChange(e, mb, false, enemyPawn, capSq);
        mb[m.from] = kNoPiece;
        mb[m.to] = uint8_t(moving);
        pushSub(moving, m.from);
        pushAdd(moving, m.to);
        threatsOnMove(e, mb, moving, m.from, moving, m.to);
    } else {  // movePiece / promotePawn (position.cpp:1306-1394)
        if ((moving >> 1) == 5) prepareKingMove(us, b.kingSq[us], m.to);
        const int landed = m.kind == kPromotion ? ((m.promo << 1) | us) : moving;
        const int captured = b.mailbox[m.to];
        if (captured != kNoPiece) {
            mb[m.from] = kNoPiece;
            pushSub(moving, m.from);
            threatsOnChange(e, mb, false, moving, m.from);
            mb[m.to] = uint8_t(landed);
            pushSub(captured, m.to);
            pushAdd(landed, m.to);
            threatsOnMutate(e, mb, captured, landed, m.to);
        } else {
            mb[m.from] = kNoPiece;
            mb[m.to] = uint8_t(landed);
            pushSub(moving, m.from);
            pushAdd(landed, m.to);
            threatsOnMove(e, mb, moving, m.from, landed, m.to);
        }
    }
    makeMove(b, m);
    // finalize (nnue_state.h:174-186)
    d.kings[0] = uint8_t(b.kingSq[0]);
    d.kings[1] = uint8_t(b.kingSq[1]);
    d.pawnsAfter[0] = b.pieces[0];
    d.pawnsAfter[1] = b.pieces[1];
}

void generateLegal(const Board& b, std::vector<Move>& out) {
    std::vector<Move> pseudo;
    pseudo.reserve(64);
    generatePseudo(b, pseudo);
    out.clear();
    for (const Move& m : pseudo) {
        Board next = b;
        makeMove(next, m);
        if (!next.attacked(next.kingSq[b.stm], next.stm, next.occ)) out.push_back(m);
    }
}

uint64_t perft(const Board& b, int depth) {
    if (depth == 0) return 1;
    std::vector<Move> moves;
    generateLegal(b, moves);
    if (depth == 1) return moves.size();
    uint64_t n = 0;
    for (const Move& m : moves) {
        Board next = b;
        makeMove(next, m);
        n += perft(next, depth - 1);
    }
    return n;
}

std::string moveToUci(const Board&, const Move& m) {
    std::string s;
    s += char('a' + (m.from & 7));
    s += char('1' + (m.from >> 3));
    s += char('a' + (m.to & 7));
    s += char('1' + (m.to >> 3));
    if (m.kind == kPromotion) s += "  nbrq"[m.promo + 1];
    return s;
}

bool moveFromUci(const Board& b, const char* uci, Move& out) {
    std::vector<Move> moves;
    generateLegal(b, moves);
    for (const Move& m : moves) {
        if (moveToUci(b, m) == uci) {
            out = m;
            return true;
        }
    }
    // standard-chess castling notation (e1g1) for non-960 callers
    for (const Move& m : moves) {
        if (m.kind != kCastling) continue;
        const int base = b.stm ? 0 : 56;
        Move alt = m;
        alt.to = uint8_t(base + (m.to > m.from ? 6 : 2));
        alt.kind = kNormal;
        if (moveToUci(b, alt) == uci) {
            out = m;
            return true;
        }
    }
    return false;
}

// ---- packed records (marlinformat.h:32-84) ----
void packBoard(const Board& b, spx_packed_pos& out) {
    std::memset(&out, 0, sizeof(out));
    out.occupancy = b.occ;
    uint64_t occ = b.occ;
    int i = 0;
    while (occ && i < 32) {  // never beyond the 16 nibble bytes (boards with more pieces are rejected at parse time)
        const int sq = ctz64(occ);
        occ &= occ - 1;
        const int pc = b.mailbox[sq];
        int type = pc >> 1;
        if (type == 3) {
            for (int c = 0; c < 2; ++c)
                for (int side = 0; side < 2; ++side)
                    if (b.castleRook[c][side] == sq) type = 6;
        }
        const uint8_t nib = uint8_t(type | ((pc & 1) ? 0 : 8));
        out.pieces[i / 2] |= uint8_t(nib << ((i & 1) * 4));
        ++i;
    }
    // relative ep square: rank 3 when black is to move, rank 6 when white is (marlinformat.h:66-70); 64 = none
    int ep = 64;
    if (b.ep >= 0) ep = (b.ep & 7) | ((b.stm ? 5 : 2) << 3);
    out.stm_ep = uint8_t((b.stm ? 0 : 0x80) | ep);
    out.halfmove = uint8_t(b.halfmove > 255 ? 255 : b.halfmove);
    out.fullmove = b.fullmove;
}

bool unpackBoard(const spx_packed_pos& in, Board& b) {
    b.clear();
    uint64_t occ = in.occupancy;
    int i = 0;
    if (popc64(occ) > 32) return false;
    while (occ) {
        const int sq = ctz64(occ);
        occ &= occ - 1;
        const int nib = (in.pieces[i / 2] >> ((i & 1) * 4)) & 0xF;
        ++i;
        if ((nib & 7) == 7) return false;
        b.put(nibbleToPiece(nib), sq);
    }
    if (b.kingSq[0] < 0 || b.kingSq[1] < 0) return false;
    if (popc64(b.pieces[10]) != 1 || popc64(b.pieces[11]) != 1) return false;
    // castling rights from the "unmoved rook" code
    occ = in.occupancy;
    i = 0;
    while (occ) {
        const int sq = ctz64(occ);
        occ &= occ - 1;
        const int nib = (in.pieces[i / 2] >> ((i & 1) * 4)) & 0xF;
        ++i;
        if ((nib & 7) == 6) {
            const int c = (nib & 8) ? 0 : 1;
            b.castleRook[c][sq > b.kingSq[c] ? 0 : 1] = int8_t(sq);
        }
    }
    b.stm = (in.stm_ep & 0x80) ? 0 : 1;
    const int ep = in.stm_ep & 0x7F;
    b.ep = int8_t(ep < 64 ? ep : -1);
    b.halfmove = in.halfmove;
    b.fullmove = in.fullmove;
    return true;
}

void randomPositions(uint64_t seed, size_t count, int minPly, int maxPly, int dfrcEvery, spx_packed_pos* out) {
    SplitMix64 rng{seed};
    std::vector<Move> moves;
    size_t produced = 0;
    uint64_t game = 0;
    if (maxPly < minPly) maxPly = minPly;
    while (produced < count) {
        ++game;
        Board b = (dfrcEvery > 0 && game % uint64_t(dfrcEvery) == 0) ? dfrcStart(rng.below(960), rng.below(960))
                                                                     : startpos();
        const int plies = minPly + int(rng.below(uint32_t(maxPly - minPly + 1)));
        bool dead = false;
        for (int i = 0; i < plies; ++i) {
            generateLegal(b, moves);
            if (moves.empty()) {
                dead = true;
                break;
            }
            makeMove(b, moves[rng.below(uint32_t(moves.size()))]);
        }
        if (dead) continue;
        packBoard(b, out[produced++]);
    }
}

}  // namespace spx
