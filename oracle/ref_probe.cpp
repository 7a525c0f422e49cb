// ref_probe.cpp - drives the COMPILED REFERENCE (Stormphrax 8.0.2) as the parity oracle's anchor.
//
// TEST INFRASTRUCTURE ONLY. This translation unit is the repo's own code; it is compiled together with the
// reference's sources *where they lie* under /root/reference (see oracle/Makefile, target _ref/sp_ref_probe) and
// calls the reference's public C++ API:
//   eval::init / eval::getNetwork              src/eval/nnue.h:38-44
//   eval::NnueState::{reset,push,pop,evaluate,evaluateOnce}   src/eval/nnue_state.h:85-116
//   nnue::features::psq::featureIndex          src/eval/nnue/features/psq.h:338-365
//   nnue::features::threats::{threatFeatureIndex,ppFeatureIndex,kPpMasks}  src/eval/nnue/features/threats.h:106-136
//   Position::{fromFen,startpos,fromDfrcIndex,applyMove,isLegal,toFen}     src/position.h
//   generateAll                                src/movegen.h:36
// Nothing from the reference is copied into this repository; the binary lives in oracle/_ref/ (git-ignored).
//
// Protocol: one command per stdin line, one or more result lines on stdout, each batch terminated by "OK".
//   eval <fen>                 -> "E <raw evaluateOnce>"
//   feat <fen>                 -> "F <raw> <bucket> <stm>" + 4 lines "R <colour> <psq|thr> <n> ids..."
//   playout <seed> <count> <minPly> <maxPly> <dfrc>  -> count lines "P <raw> <fen>"
//   trace <seed> <maxEvals> <depth> <fen>            -> opcode stream of a DFS make/unmake walk with evaluate() values
//   searchtrace <maxEvals> <depth> <hardNodes> <fen> -> the same opcode stream recorded from the reference's own ALPHA-BETA
//                                 SEARCH (Searcher::runDatagenSearch: PVS + qsearch, null moves, reductions, TT): every
//                                 Position::applyMove<BoardObserver> the search makes (thread.cpp:46-67) is a PUSH, every
//                                 NnueState::pop (thread.h:116-122) a POP, every NnueState::evaluate (search.cpp:782,1520)
//                                 an EVAL. The three functions are reached through link-time interposition (oracle/Makefile
//                                 renames the reference's definitions in copies of two of ITS object files; the wrappers
//                                 below log and forward) - no reference source is touched or copied. PUSH and EVAL carry
//                                 the side to move of the position they act on: after a null move (no NNUE push,
//                                 thread.cpp:28-44) it differs from the stack's, which is how the replayer sees null moves.
//   deltas <seed> <count> <dfrc>                     -> per played move the BoardObserver's UpdateContext
//   adjust <cB> <cW> <oB> <oW> <fen>                 -> "A <staticEvalOnce(contempt)> <adjustEval<false>(optimism, static)>"
//   add <fen> / bench <threads> <seconds>            -> timing of evaluateOnce over the added positions
//   pack <fen>                 -> "K <64 hex digits>": the 32 bytes of datagen::marlinformat::PackedBoard::pack(pos, 0)
//   viri <seed> <plies> <dfrc> -> a random game pushed through datagen::Viriformat: per ply "M <fen before> | <uci> |
//                                 <score> | <filtered> | <pack hex of the position before>", then "V <hex of the stream
//                                 writeAllWithOutcome wrote>" (marlinformat.h:32-84, viriformat.cpp:28-63); then "F <kept> <hex>" =
//                                 datagen::Marlinformat's file bytes and "T <lines joined by ;>" = datagen::Fen's, same game
//   drawn <seed> <plies> <undo permille> <fen> -> "S <fen>", then per move "X <uci> | <Position::isDrawn(0, keyHistory) of the
//                                 new position, as datagen.cpp:258-265 asks it> | <fen after>" (position.cpp:603-667)
//   wdl <score> <fen>          -> "W <classicalMaterial> <wdl::normalizeScore(score, material)>" (wdl.cpp:28-79) at
//                                 datagen's evalSharpness of 100 (datagen.cpp:353)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "attacks/attacks.h"
#include "cuckoo.h"
#include "datagen/marlinformat.h"
#include "datagen/fen.h"
#include "datagen/marlinformat.h"
#include "datagen/viriformat.h"
#include "eval/eval.h"
#include "eval/nnue.h"
#include "eval/nnue_state.h"
#include "movegen.h"
#include "opts.h"
#include "position.h"
#include "search.h"
#include "tunable.h"
#include "util/numa/numa.h"
#include "wdl.h"

using namespace stormphrax;

namespace stormphrax {
    // defined near the end of this file: the logging wrapper that `searchtrace` interposes (declared before any use)
    template <>
    Position Position::applyMove<eval::BoardObserver>(Move move, eval::BoardObserver observer) const;
} // namespace stormphrax

namespace {
    struct SplitMix64 {
        u64 s;
        u64 next() {
            u64 z = (s += 0x9E3779B97F4A7C15ull);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            return z ^ (z >> 31);
        }
        u32 below(u32 n) {
            return static_cast<u32>((next() >> 32) % n);
        }
    };

    std::string hexOf(const void* data, usize n) {
        static const char* kDigits = "0123456789abcdef";
        std::string out;
        const auto* bytes = static_cast<const unsigned char*>(data);
        for (usize i = 0; i < n; ++i) {
            out += kDigits[bytes[i] >> 4];
            out += kDigits[bytes[i] & 15];
        }
        return out;
    }

    std::vector<Move> legalMoves(const Position& pos) {
        ScoredMoveList moves{};
        generateAll(moves, pos);
        std::vector<Move> out;
        for (const auto& [move, score] : moves) {
            if (pos.isLegal(move)) {
                out.push_back(move);
            }
        }
        return out;
    }

    // Row lists of one perspective, enumerated in the same order as the reference does internally
    // (nnue_state.cpp:440-449 for psq, :309-354 for threats + pawn pairs), through the reference's own indexers.
    void featureRows(const Position& pos, Color c, std::vector<u32>& psq, std::vector<u32>& thr) {
        using namespace eval;
        namespace threats = eval::nnue::features::threats;
        const auto kingSq = pos.king(c);
        for (const auto [piece, sq] : pos) {
            psq.push_back(nnue::features::psq::featureIndex<InputFeatureSet>(c, piece, sq, kingSq));
        }
        const auto occ = pos.occ();
        const auto kings = pos.bb(PieceTypes::kKing);
        for (const auto from : occ & ~kings) {
            const auto piece = pos.pieceOn(from);
            for (const auto to : occ & attacks::getAttacks(piece, from, occ) & ~kings) {
                const auto f = threats::threatFeatureIndex(c, kingSq, piece, from, pos.pieceOn(to), to);
                if (f >= 0) {
                    thr.push_back(static_cast<u32>(f));
                }
            }
        }
        const auto ours = pos.bb(PieceTypes::kPawn, c);
        const auto theirs = pos.bb(PieceTypes::kPawn, c.flip());
        for (const auto [a, remaining] : ours.iterWithRemaining()) {
            const auto mask = threats::kPpMasks[a.idx()];
            for (const auto b : remaining & mask) {
                thr.push_back(threats::ppFeatureIndex(c, kingSq, c, a, c, b));
            }
            for (const auto b : theirs & mask) {
                thr.push_back(threats::ppFeatureIndex(c, kingSq, c, a, c.flip(), b));
            }
        }
        for (const auto [a, remaining] : theirs.iterWithRemaining()) {
            for (const auto b : remaining & threats::kPpMasks[a.idx()]) {
                thr.push_back(threats::ppFeatureIndex(c, kingSq, c.flip(), a, c.flip(), b));
            }
        }
    }

    void printRows(int colour, const char* kind, const std::vector<u32>& rows) {
        std::printf("R %d %s %zu", colour, kind, rows.size());
        for (const auto r : rows) {
            std::printf(" %u", r);
        }
        std::printf("\n");
    }

    // DFS make/unmake walk that mimics how search drives NnueState (thread.cpp:46-67, thread.h:116-122):
    // PUSH <uci> applies a move through the observer, EVAL prints the lazily-updated evaluate(), POP unwinds.
    struct Tracer {
        eval::NnueState state;
        SplitMix64 rng;
        u64 evals{};
        u64 maxEvals{};

        void walk(const Position& pos, i32 depth) {
            if (evals >= maxEvals) {
                return;
            }
            // evaluate lazily only at some nodes so multi-ply walk-backs (nnue_state.cpp:636-697) are exercised
            if (depth == 0 || rng.below(3) != 0) {
                const auto v = state.evaluate(pos, pos.stm());
                const auto once = eval::NnueState::evaluateOnce(pos, pos.stm());
                std::printf("EVAL %d %d\n", v, once);
                ++evals;
            }
            if (depth == 0) {
                return;
            }
            auto moves = legalMoves(pos);
            // visit a random subset so deep traces stay bounded
            const u32 branch = 2 + rng.below(3);
            for (u32 i = 0; i < branch && !moves.empty() && evals < maxEvals; ++i) {
                const auto idx = rng.below(static_cast<u32>(moves.size()));
                const auto move = moves[idx];
                moves.erase(moves.begin() + idx);
                std::printf("PUSH %s\n", fmt::format("{}", move).c_str());
                const auto next = pos.applyMove(move, state.push());
                walk(next, depth - 1);
                state.pop();
                std::printf("POP\n");
            }
        }
    };
} // namespace

// ---- link-time interposition for `searchtrace` (see the header comment and oracle/Makefile) ----
namespace {
    bool g_tracing = false;
    u64 g_traceEvals = 0, g_traceMaxEvals = 0;
    bool tracing() {
        return g_tracing && g_traceEvals < g_traceMaxEvals;
    }
} // namespace
// the reference's own definitions, under the names oracle/Makefile gave them in its copies of position.o / nnue_state.o
// (Itanium ABI: a non-static member function is a free function taking `this` first)
Position spxOrigApplyMove(const Position* self, Move move, eval::BoardObserver observer) asm("spx_orig_apply_move_observed");
void spxOrigPop(eval::NnueState* self) asm("spx_orig_nnue_pop");
i32 spxOrigEvaluate(eval::NnueState* self, const Position& pos, Color stm) asm("spx_orig_nnue_evaluate");

namespace stormphrax {
    template <>
    Position Position::applyMove<eval::BoardObserver>(Move move, eval::BoardObserver observer) const {
        if (tracing()) {
            std::printf("PUSH %s %c\n", fmt::format("{}", move).c_str(), stm() == Colors::kWhite ? 'w' : 'b');
        }
        return spxOrigApplyMove(this, move, observer);
    }
    void eval::NnueState::pop() {
        if (tracing()) {
            std::printf("POP\n");
        }
        spxOrigPop(this);
    }
    i32 eval::NnueState::evaluate(const Position& pos, Color stm) {
        const auto v = spxOrigEvaluate(this, pos, stm);
        if (tracing()) {
            std::printf("EVAL %d %d %c\n", v, eval::NnueState::evaluateOnce(pos, stm), pos.stm() == Colors::kWhite ? 'w' : 'b');
            ++g_traceEvals;
        }
        return v;
    }
} // namespace stormphrax

int main() {
    if (!numa::init()) {
        return 1;
    }
    tunable::init();
    cuckoo::init();
    eval::init();
    if (!eval::isNetworkLoaded()) {
        std::fprintf(stderr, "reference failed to load the embedded network\n");
        return 2;
    }
    opts::mutableOpts().chess960 = true; // harmless for standard FENs; required for DFRC castling rights

    std::vector<Position> added;
    std::string line;
    while (std::getline(std::cin, line)) {
        std::istringstream in{line};
        std::string cmd;
        in >> cmd;
        if (cmd == "eval" || cmd == "feat" || cmd == "add") {
            std::string fen;
            std::getline(in, fen);
            const auto pos = Position::fromFen(fen);
            if (!pos) {
                std::printf("ERR bad fen\nOK\n");
                std::fflush(stdout);
                continue;
            }
            if (cmd == "add") {
                added.push_back(*pos);
            } else {
                const auto raw = eval::NnueState::evaluateOnce(*pos, pos->stm());
                if (cmd == "eval") {
                    std::printf("E %d\n", raw);
                } else {
                    std::printf(
                        "F %d %u %d\n",
                        raw,
                        eval::OutputBucketing::getBucket(*pos),
                        pos->stm() == Colors::kWhite ? 1 : 0
                    );
                    for (const auto c : {Colors::kBlack, Colors::kWhite}) {
                        std::vector<u32> psq, thr;
                        featureRows(*pos, c, psq, thr);
                        printRows(c == Colors::kWhite ? 1 : 0, "psq", psq);
                        printRows(c == Colors::kWhite ? 1 : 0, "thr", thr);
                    }
                }
            }
        } else if (cmd == "adjust") {
            // eval::staticEvalOnce (eval.cpp:109-112: contempt + clamp) and eval::adjustEval<false> (eval.cpp:30-67:
            // material scaling, optimism, halfmove damping, clamp; no correction history) on one position
            eval::Contempt contempt{};
            eval::Optimism optimism{};
            in >> contempt[0] >> contempt[1] >> optimism[0] >> optimism[1];
            std::string fen;
            std::getline(in, fen);
            const auto pos = Position::fromFen(fen);
            if (!pos) {
                std::printf("ERR bad fen\nOK\n");
                std::fflush(stdout);
                continue;
            }
            const auto stat = eval::staticEvalOnce(*pos, contempt);
            const auto adjusted = eval::adjustEval<false>(*pos, optimism, {}, nullptr, stat);
            std::printf("A %d %d\n", stat, adjusted);
        } else if (cmd == "pack") {
            std::string fen;
            std::getline(in, fen);
            const auto pos = Position::fromFen(fen);
            if (!pos) {
                std::printf("ERR bad fen\nOK\n");
                std::fflush(stdout);
                continue;
            }
            const auto packed = datagen::marlinformat::PackedBoard::pack(*pos, 0);
            std::printf("K %s\n", hexOf(&packed, sizeof(packed)).c_str());
        } else if (cmd == "wdl") {
            i32 score;
            in >> score;
            std::string fen;
            std::getline(in, fen);
            const auto pos = Position::fromFen(fen);
            if (!pos) {
                std::printf("ERR bad fen\nOK\n");
                std::fflush(stdout);
                continue;
            }
            const auto material = pos->classicalMaterial();
            // datagen runs with evalSharpness = 100 (datagen.cpp:353: no sharpening), which makes the default
            // normalizeScore<true> that runDatagenSearch calls (search.cpp:238) equal to normalizeScore<false>
            opts::mutableOpts().evalSharpness = 100;
            const auto norm = wdl::normalizeScore(score, material);
            if (norm != wdl::normalizeScore<false>(score, material)) {
                std::printf("ERR sharpened and plain normalisation differ at sharpness 100\nOK\n");
                std::fflush(stdout);
                continue;
            }
            std::printf("W %d %d\n", material, norm);
        } else if (cmd == "viri") {
            u64 seed;
            u32 plies, dfrc;
            in >> seed >> plies >> dfrc;
            SplitMix64 rng{seed};
            auto pos = dfrc ? *Position::fromDfrcIndex(rng.below(960 * 960)) : Position::startpos();
            datagen::Viriformat format{};
            datagen::Marlinformat marlin{};  // the other two output formats of datagen.cpp:340-346 see the same pushes
            datagen::Fen fenLines{};
            format.start(pos);
            marlin.start(pos);
            fenLines.start(pos);
            for (u32 ply = 0; ply < plies; ++ply) {
                const auto moves = legalMoves(pos);
                if (moves.empty()) {
                    break;
                }
                std::vector<Move> special;
                for (const auto mv : moves) {
                    if (mv.type() != MoveType::kStandard) {
                        special.push_back(mv);
                    }
                }
                const auto& pool = (!special.empty() && rng.below(2) == 0) ? special : moves;
                const auto move = pool[rng.below(static_cast<u32>(pool.size()))];
                const auto score = static_cast<Score>(rng.below(4001)) - 2000;
                const bool filtered = pos.isCheck() || pos.isNoisy(move);  // datagen.cpp:254
                const auto packed = datagen::marlinformat::PackedBoard::pack(pos, static_cast<i16>(score));
                std::printf("M %s | %s | %d | %d | %s\n", pos.toFen().c_str(), fmt::format("{}", move).c_str(), score,
                            filtered ? 1 : 0, hexOf(&packed, sizeof(packed)).c_str());
                format.push(filtered, move, score);
                marlin.push(filtered, move, score);
                fenLines.push(filtered, move, score);
                pos = pos.applyMove(move);
            }
            std::ostringstream stream, marlinStream, fenStream;
            const auto outcome = static_cast<datagen::Outcome>(rng.below(3));
            format.writeAllWithOutcome(stream, outcome);
            const auto bytes = stream.str();
            std::printf("V %s\n", hexOf(bytes.data(), bytes.size()).c_str());
            // datagen::Marlinformat (marlinformat.cpp:32-57) and datagen::Fen (fen.cpp:32-66) of the same game
            const auto kept = marlin.writeAllWithOutcome(marlinStream, outcome);
            const auto marlinBytes = marlinStream.str();
            std::printf("F %zu %s\n", static_cast<size_t>(kept), hexOf(marlinBytes.data(), marlinBytes.size()).c_str());
            fenLines.writeAllWithOutcome(fenStream, outcome);
            auto text = fenStream.str();
            for (auto& ch : text) {
                if (ch == '\n') ch = ';';
            }
            std::printf("T %s\n", text.c_str());
        } else if (cmd == "drawn") {
            // `drawn <seed> <plies> <undo permille> <fen>`: a game whose movers like to take their last move back (repetitions)
            // and to capture (bare material); after every move what datagen asks (datagen.cpp:258-265): the key of the
            // position moved from is pushed, the move made, then Position::isDrawn(0, keyHistory) of the new position
            // (position.cpp:603-667: 50-move rule unless checkmate, threefold repetition, insufficient material). The game goes
            // on after a "drawn" answer - the point is the flags, not the game.
            u64 seed;
            u32 plies, undo;
            in >> seed >> plies >> undo;
            std::string fen;
            std::getline(in, fen);
            const auto start = Position::fromFen(fen);
            if (!start) {
                std::printf("ERR bad fen\nOK\n");
                std::fflush(stdout);
                continue;
            }
            SplitMix64 rng{seed};
            auto pos = *start;
            std::vector<u64> keyHistory;
            std::array<Move, 2> last{kNullMove, kNullMove};  // by colour of the mover
            std::printf("S %s\n", pos.toFen().c_str());
            for (u32 ply = 0; ply < plies; ++ply) {
                const auto moves = legalMoves(pos);
                if (moves.empty()) {
                    break;
                }
                const auto us = pos.stm().idx();
                Move move = kNullMove;
                if (last[us] != kNullMove && rng.below(1000) < undo) {
                    for (const auto mv : moves) {
                        if (mv.type() == MoveType::kStandard && mv.fromSq() == last[us].toSq() && mv.toSq() == last[us].fromSq()) {
                            move = mv;
                        }
                    }
                }
                if (move == kNullMove && rng.below(10) < 3) {
                    std::vector<Move> captures;
                    for (const auto mv : moves) {
                        if (pos.isNoisy(mv)) {
                            captures.push_back(mv);
                        }
                    }
                    if (!captures.empty()) {
                        move = captures[rng.below(static_cast<u32>(captures.size()))];
                    }
                }
                if (move == kNullMove) {
                    move = moves[rng.below(static_cast<u32>(moves.size()))];
                }
                last[us] = move;
                keyHistory.push_back(pos.key());
                pos = pos.applyMove(move);
                const bool drawn = pos.isDrawn(0, keyHistory);
                std::printf("X %s | %d | %s\n", fmt::format("{}", move).c_str(), drawn ? 1 : 0, pos.toFen().c_str());
            }
        } else if (cmd == "playout") {
            u64 seed;
            u32 count, minPly, maxPly, dfrc;
            in >> seed >> count >> minPly >> maxPly >> dfrc;
            SplitMix64 rng{seed};
            u32 produced = 0;
            while (produced < count) {
                auto pos = dfrc ? *Position::fromDfrcIndex(rng.below(960 * 960)) : Position::startpos();
                const u32 plies = minPly + rng.below(maxPly - minPly + 1);
                bool dead = false;
                for (u32 i = 0; i < plies; ++i) {
                    const auto moves = legalMoves(pos);
                    if (moves.empty()) {
                        dead = true;
                        break;
                    }
                    pos = pos.applyMove(moves[rng.below(static_cast<u32>(moves.size()))]);
                }
                if (dead) {
                    continue;
                }
                const auto raw = eval::NnueState::evaluateOnce(pos, pos.stm());
                std::printf("P %d %s\n", raw, pos.toFen().c_str());
                ++produced;
            }
        } else if (cmd == "deltas") {
            // `deltas <seed> <count> <dfrc>`: random playouts; for every played move print the UpdateContext the
            // reference's BoardObserver captured (nnue_state.h:118-186): piece-square subs/adds, threat descriptors
            // added/removed, refresh flags.
            u64 seed;
            u32 count, dfrc;
            in >> seed >> count >> dfrc;
            SplitMix64 rng{seed};
            u32 produced = 0;
            const auto pc = [](Piece p) { return static_cast<int>(p.idx()); };
            while (produced < count) {
                auto pos = dfrc ? *Position::fromDfrcIndex(rng.below(960 * 960)) : Position::startpos();
                for (u32 ply = 0; ply < 200 && produced < count; ++ply) {
                    const auto moves = legalMoves(pos);
                    if (moves.empty()) {
                        break;
                    }
                    // castling, en passant and promotions are rare in uniform playouts: prefer them half of the time
                    std::vector<Move> special;
                    for (const auto mv : moves) {
                        if (mv.type() != MoveType::kStandard) {
                            special.push_back(mv);
                        }
                    }
                    const auto& pool = (!special.empty() && rng.below(2) == 0) ? special : moves;
                    const auto move = pool[rng.below(static_cast<u32>(pool.size()))];
                    eval::UpdateContext ctx{};
                    const auto next = pos.applyMove(move, eval::BoardObserver{ctx});
                    std::printf("D %s | %s |", pos.toFen().c_str(), fmt::format("{}", move).c_str());
                    for (const auto [p, sq] : ctx.updates.sub) {
                        std::printf(" s%d,%d", pc(p), static_cast<int>(sq.idx()));
                    }
                    for (const auto [p, sq] : ctx.updates.add) {
                        std::printf(" a%d,%d", pc(p), static_cast<int>(sq.idx()));
                    }
                    for (const auto& t : ctx.updates.threatsAdded) {
                        std::printf(" +%d,%d,%d,%d", pc(t.attacker), static_cast<int>(t.attackerSq.idx()), pc(t.attacked), static_cast<int>(t.attackedSq.idx()));
                    }
                    for (const auto& t : ctx.updates.threatsRemoved) {
                        std::printf(" -%d,%d,%d,%d", pc(t.attacker), static_cast<int>(t.attackerSq.idx()), pc(t.attacked), static_cast<int>(t.attackedSq.idx()));
                    }
                    std::printf(
                        " f%d%d%d%d\n",
                        ctx.updates.requiresPsqRefresh(Colors::kBlack),
                        ctx.updates.requiresPsqRefresh(Colors::kWhite),
                        ctx.updates.requiresThreatRefresh(Colors::kBlack),
                        ctx.updates.requiresThreatRefresh(Colors::kWhite)
                    );
                    pos = next;
                    ++produced;
                }
            }
        } else if (cmd == "trace") {
            u64 seed, maxEvals;
            i32 depth;
            in >> seed >> maxEvals >> depth;
            std::string fen;
            std::getline(in, fen);
            const auto pos = Position::fromFen(fen);
            if (!pos) {
                std::printf("ERR bad fen\nOK\n");
                std::fflush(stdout);
                continue;
            }
            Tracer tracer{};
            tracer.state.setNetwork(eval::getNetwork(0));
            tracer.rng = SplitMix64{seed};
            tracer.maxEvals = maxEvals;
            tracer.state.reset(*pos);
            std::printf("ROOT %s\n", pos->toFen().c_str());
            tracer.walk(*pos, depth);
        } else if (cmd == "searchtrace") {
            u64 maxEvals, hardNodes;
            i32 depth;
            in >> maxEvals >> depth >> hardNodes;
            std::string fen;
            std::getline(in, fen);
            const auto pos = Position::fromFen(fen);
            if (!pos) {
                std::printf("ERR bad fen\nOK\n");
                std::fflush(stdout);
                continue;
            }
            // the set-up of datagen's searches (src/datagen/datagen.cpp:112-135,177-184): one searcher, its thread data taken
            // over, a hard node limit, a depth limit
            search::Searcher searcher{16};
            searcher.setSilent(true);
            auto& thread = searcher.take(0);
            thread.datagen = true;
            limit::SearchLimiter limiter{util::Instant::now()};
            limiter.setHardNodes(hardNodes);
            searcher.setLimiter(limiter);
            searcher.setMaxDepth(depth);
            searcher.newGame();
            thread.search = search::SearchData{};
            thread.keyHistory.clear();
            thread.rootPos = *pos;
            thread.nnueState.reset(thread.rootPos);
            std::printf("ROOT %s\n", pos->toFen().c_str());
            g_traceEvals = 0;
            g_traceMaxEvals = maxEvals;
            g_tracing = true;
            const auto [score, norm] = searcher.runDatagenSearch();
            g_tracing = false;
            std::printf("# searched to depth <= %d, %llu nodes, score %d (white point of view), %llu evaluates recorded\n", depth,
                        static_cast<unsigned long long>(thread.search.loadNodes()), score,
                        static_cast<unsigned long long>(g_traceEvals));
        } else if (cmd == "bench") {
            u32 threads;
            double seconds;
            in >> threads >> seconds;
            if (added.empty() || threads == 0) {
                std::printf("ERR nothing to bench\nOK\n");
                std::fflush(stdout);
                continue;
            }
            std::atomic<u64> total{0};
            std::atomic<i64> checksum{0};
            std::atomic<bool> stop{false};
            std::vector<std::thread> pool;
            const auto start = std::chrono::steady_clock::now();
            for (u32 t = 0; t < threads; ++t) {
                pool.emplace_back([&, t] {
                    u64 n = 0;
                    i64 sum = 0;
                    usize i = t % added.size();
                    while (!stop.load(std::memory_order_relaxed)) {
                        for (u32 k = 0; k < 256; ++k) {
                            const auto& pos = added[i];
                            sum += eval::NnueState::evaluateOnce(pos, pos.stm());
                            i = i + 1 == added.size() ? 0 : i + 1;
                        }
                        n += 256;
                    }
                    total += n;
                    checksum += sum;
                });
            }
            std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
            stop = true;
            for (auto& th : pool) {
                th.join();
            }
            const std::chrono::duration<double> dt = std::chrono::steady_clock::now() - start;
            std::printf(
                "B %.1f evals_per_sec %llu evals %.3f s %u threads checksum %lld\n",
                static_cast<double>(total.load()) / dt.count(),
                static_cast<unsigned long long>(total.load()),
                dt.count(),
                threads,
                static_cast<long long>(checksum.load())
            );
        } else if (cmd == "quit") {
            break;
        } else if (!cmd.empty()) {
            std::printf("ERR unknown command\n");
        }
        std::printf("OK\n");
        std::fflush(stdout);
    }
    eval::shutdown();
    return 0;
}
