// Host chess core: just enough board logic to feed the evaluator - FEN <-> board <-> 32-byte packed records,
// legal move generation (standard + Chess960 castling), make-move, perft, and seeded random playouts that produce
// the synthetic position batches bench.py and the parity tests run on.
//
// Reference counterparts (behaviour, not code): src/position.{h,cpp} (copy-make board, FEN), src/movegen.*,
// src/datagen/marlinformat.h:32-84 (PackedBoard), src/datagen/datagen.cpp:146-171 (random openings).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "spx_arch.h"

struct spx_packed_pos;  // include/spx_nnue.h

namespace spx {

enum MoveKind : uint8_t { kNormal = 0, kPromotion = 1, kCastling = 2, kEnPassant = 3 };

struct Move {
    uint8_t from, to;  // castling: to = own rook square ("king takes rook", as the reference encodes it)
    uint8_t kind;      // MoveKind
    uint8_t promo;     // piece TYPE for promotions (1..4)
};

struct Board {
    uint8_t mailbox[64];
    uint64_t pieces[12];  // by piece id (type<<1 | colour, white = 1)
    uint64_t colour[2];
    uint64_t occ;
    int8_t kingSq[2];
    int8_t castleRook[2][2];  // [colour][0 = kingside, 1 = queenside] rook square or -1
    int8_t ep;                // en-passant target square or -1
    uint8_t stm;              // 0 black, 1 white
    uint16_t halfmove, fullmove;

    void clear();
    void put(int piece, int sq);
    void remove(int sq);
    bool attacked(int sq, int byColour, uint64_t occupancy) const;
    bool inCheck() const {
        return attacked(kingSq[stm], stm ^ 1, occ);
    }
};

bool boardFromFen(const char* fen, Board& out);
std::string boardToFen(const Board& b);
Board startpos();
Board dfrcStart(uint32_t whiteIdx, uint32_t blackIdx);  // independent Chess960 back ranks (Scharnagl numbering)

void generateLegal(const Board& b, std::vector<Move>& out);
void makeMove(Board& b, const Move& m);
uint64_t perft(const Board& b, int depth);
std::string moveToUci(const Board& b, const Move& m);  // Chess960-style castling (king takes rook)
bool moveFromUci(const Board& b, const char* uci, Move& out);

// ---- make-move delta capture: what the reference's BoardObserver records while Position::applyMove runs
// (src/eval/nnue_state.h:118-186, src/eval/nnue.cpp:490-599, geometry in nnue/features/threats/geometry.h:44-141) ----
struct ThreatDescriptor {  // psq.h:30-35
    uint8_t attacker, attackerSq, attacked, attackedSq;
};
struct MoveDelta {         // UpdateContext: NnueUpdates + kings (nnue_state.h:28-31, threats.h:41-104)
    uint8_t nSub = 0, nAdd = 0;
    uint8_t subPiece[2], subSq[2], addPiece[2], addSq[2];
    bool psqRefresh[2] = {false, false}, threatRefresh[2] = {false, false};
    uint8_t kings[2] = {0, 0};
    uint64_t pawnsBefore[2] = {0, 0}, pawnsAfter[2] = {0, 0};
    std::vector<ThreatDescriptor> threatsAdded, threatsRemoved;
};
// makeMove + the observer's event stream, in the reference's event order (position.cpp:1306-1471)
void makeMoveObserved(Board& b, const Move& m, MoveDelta& delta);

void packBoard(const Board& b, spx_packed_pos& out);
bool unpackBoard(const spx_packed_pos& in, Board& out);  // placement + stm (+ep); castling rights from code-6 rooks

// Seeded random playouts: positions after [minPly, maxPly] uniformly random legal plies from the standard start
// position or (every dfrcEvery-th game, 0 = never) a double-Chess960 start.
void randomPositions(uint64_t seed, size_t count, int minPly, int maxPly, int dfrcEvery, spx_packed_pos* out);

}  // namespace spx
