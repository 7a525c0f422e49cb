// oracle/ref_gpu.cpp - TEST INFRASTRUCTURE (never linked into, imported by or shipped with the product).
//
// The reference ENGINE running on the GPU evaluator (the drop-in proven rather than sketched).
// oracle/Makefile (target `refgpu`) links Stormphrax 8.0.2's own objects - search, move generation, transposition table,
// datagen, everything, compiled from the sources where they lie - against stormphrax_amd/libspx_nnue.so, with ONE seam
// cut open by link-time interposition: in copies of two object files the reference's definitions of
//     eval::NnueState::reset / push / pop / applyImmediately / evaluate / evaluateOnce   (src/eval/nnue_state.h:85-116)
//     Position::applyMove<eval::BoardObserver>                                            (src/position.cpp:1306-1471)
// are renamed (objcopy --redefine-sym), and this file defines them under their real names: each forwards to
// include/spx_nnue.hpp (the C++ face of the C ABI - exactly what INTEGRATION.md tells a maintainer to call) and, so that the
// engine's own bookkeeping stays alive for the CPU runs of the same binary, to the renamed original. No reference source is
// changed, copied or stubbed.
//
//   sp_ref_gpu_<preset> bench <depth> cpu|gpu|both     bench::run (src/bench.cpp:95-150: the reference's only functional
//                                                      test - the node count of fixed-depth searches of 52 positions) with
//                                                      every NNUE evaluation of the search on the CPU path / on the GPU;
//                                                      `both`: GPU values used, every one also checked against the CPU's
//   sp_ref_gpu_<preset> raweval cpu|gpu                FENs on stdin -> eval::staticEvalOnce (what UCI `raweval` prints,
//                                                      src/uci.cpp:797-800)
//   sp_ref_gpu_<preset> game <plies> <seed>            datagen's use (src/datagen/datagen.cpp:257-262): random legal moves
//                                                      made with the observer, NnueState::applyImmediately given the
//                                                      REFERENCE'S OWN UpdateContext (converted field by field to
//                                                      spx_move_delta), and the invariant the reference asserts there -
//                                                      staticEvalOnce(pos) == staticEval(pos, nnueState) - checked on
//                                                      the GPU state after every move, next to the CPU state's value
// The net is the repo's synthetic preset the reference objects embed (the default net cannot be fetched offline).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "bench.h"
#include "cuckoo.h"
#include "datagen/marlinformat.h"
#include "eval/eval.h"
#include "eval/nnue.h"
#include "eval/nnue_state.h"
#include "movegen.h"
#include "opts.h"
#include "position.h"
#include "search.h"
#include "tunable.h"
#include "util/numa/numa.h"

#include "../include/spx_nnue.hpp"

// the net the GPU evaluator loads: the file the reference objects embed (there in the reference's permuted layout, here as the
// net file itself), through spx_net_load like any net a user would hand over
__asm__(".section .rodata\n.balign 64\n.global spx_ref_net_begin\nspx_ref_net_begin:\n.incbin \"" SPX_REF_NET_FILE "\"\n"
        ".global spx_ref_net_end\nspx_ref_net_end:\n.previous\n");
extern "C" const unsigned char spx_ref_net_begin[], spx_ref_net_end[];

using namespace stormphrax;

namespace {
    bool g_gpu = false, g_check = false;
    unsigned long long g_evals = 0, g_mismatches = 0, g_nullMoveEvals = 0, g_pushes = 0, g_pendingSum = 0, g_pendingMax = 0;

    spx_nnue::Network& network() {
        static spx_nnue::Network net(spx_ref_net_begin, size_t(spx_ref_net_end - spx_ref_net_begin));
        return net;
    }

    std::mutex g_mutex;
    std::unordered_map<const eval::NnueState*, std::unique_ptr<spx_nnue::NnueState>> g_states;
    // one device-side accumulator stack per eval::NnueState of the engine (one per search thread, thread.h:147)
    spx_nnue::NnueState& mirror(const eval::NnueState* self) {
        std::lock_guard<std::mutex> lock{g_mutex};
        auto& slot = g_states[self];
        if (!slot) {
            slot = std::make_unique<spx_nnue::NnueState>(network(), 0, 256);
        }
        return *slot;
    }
    spx_nnue::NnueState& shared() {  // evaluateOnce is static in the reference: no state of its own
        return mirror(nullptr);
    }

    // the wire record IS the reference's PackedBoard (include/spx_nnue.h): its own packer makes it
    spx_packed_pos pack(const Position& pos) {
        const auto board = datagen::marlinformat::PackedBoard::pack(pos, 0);
        static_assert(sizeof(board) == sizeof(spx_packed_pos));
        spx_packed_pos rec;
        std::memcpy(&rec, &board, sizeof(rec));
        return rec;
    }
    bool blackToMove(const spx_packed_pos& rec) {
        return (reinterpret_cast<const unsigned char*>(&rec)[24] & 0x80u) != 0;
    }

    // eval::UpdateContext (nnue_state.h:28-31, threats.h:46-103, psq.h:43-70) -> spx_move_delta, field by field
    spx_move_delta convert(const eval::UpdateContext& ctx) {
        spx_move_delta d{};
        const auto& u = ctx.updates;
        for (const auto [piece, sq] : u.sub) {
            d.sub_piece[d.n_sub] = static_cast<uint8_t>(piece.idx());
            d.sub_sq[d.n_sub++] = static_cast<uint8_t>(sq.idx());
        }
        for (const auto [piece, sq] : u.add) {
            d.add_piece[d.n_add] = static_cast<uint8_t>(piece.idx());
            d.add_sq[d.n_add++] = static_cast<uint8_t>(sq.idx());
        }
        for (const auto& t : u.threatsAdded) {
            d.threats_added[d.n_threats_added++] = {static_cast<uint8_t>(t.attacker.idx()), static_cast<uint8_t>(t.attackerSq.idx()),
                                                    static_cast<uint8_t>(t.attacked.idx()), static_cast<uint8_t>(t.attackedSq.idx())};
        }
        for (const auto& t : u.threatsRemoved) {
            d.threats_removed[d.n_threats_removed++] = {static_cast<uint8_t>(t.attacker.idx()), static_cast<uint8_t>(t.attackerSq.idx()),
                                                        static_cast<uint8_t>(t.attacked.idx()), static_cast<uint8_t>(t.attackedSq.idx())};
        }
        for (const auto c : {Colors::kBlack, Colors::kWhite}) {
            d.psq_refresh[c.idx()] = u.requiresPsqRefresh(c);
            d.threat_refresh[c.idx()] = u.requiresThreatRefresh(c);
            d.pawns_before[c.idx()] = static_cast<u64>(u.pawnBbsBefore[c.idx()]);
            d.pawns_after[c.idx()] = static_cast<u64>(u.pawnBbsAfter[c.idx()]);
        }
        d.kings[0] = static_cast<uint8_t>(ctx.kings.black().idx());
        d.kings[1] = static_cast<uint8_t>(ctx.kings.white().idx());
        return d;
    }

    thread_local eval::NnueState* t_pushed = nullptr;  // the state whose push() made the observer of the move being applied
} // namespace

// the reference's own definitions, under the names oracle/Makefile gave them in its copies of position.o / nnue_state.o
// (Itanium ABI: a non-static member function is a free function taking `this` first)
Position spxOrigApplyMove(const Position* self, Move move, eval::BoardObserver observer) asm("spx_orig_apply_move_observed");
void spxOrigReset(eval::NnueState* self, const Position& pos) asm("spx_orig_nnue_reset");
eval::BoardObserver spxOrigPush(eval::NnueState* self) asm("spx_orig_nnue_push");
void spxOrigPop(eval::NnueState* self) asm("spx_orig_nnue_pop");
void spxOrigApplyImmediately(eval::NnueState* self, const eval::UpdateContext& ctx, const Position& pos) asm("spx_orig_nnue_apply_immediately");
i32 spxOrigEvaluate(eval::NnueState* self, const Position& pos, Color stm) asm("spx_orig_nnue_evaluate");
i32 spxOrigEvaluateOnce(const Position& pos, Color stm) asm("spx_orig_nnue_evaluate_once");

namespace stormphrax {
    // search: thread.cpp:64 `pos.applyMove(move, nnueState.push())` - push() tells whose move this is, the position after
    // the move is what the device-side stack needs (it derives the delta from the two boards, spx_acc_update_chain_eval).
    // datagen: `pos.applyMove(move, BoardObserver{ctx})` without a push - applyImmediately below gets the context.
    template <>
    Position Position::applyMove<eval::BoardObserver>(Move move, eval::BoardObserver observer) const {
        auto* const state = t_pushed;
        t_pushed = nullptr;
        Position child = spxOrigApplyMove(this, move, observer);
        if (g_gpu && state) {
            mirror(state).push(pack(child));
            ++g_pushes;
        }
        return child;
    }

    void eval::NnueState::reset(const Position& pos) {
        if (g_gpu) {
            mirror(this).reset(pack(pos));
        }
        spxOrigReset(this, pos);
    }

    eval::BoardObserver eval::NnueState::push() {
        t_pushed = this;
        return spxOrigPush(this);
    }

    void eval::NnueState::pop() {
        if (g_gpu) {
            mirror(this).pop();
        }
        spxOrigPop(this);
    }

    void eval::NnueState::applyImmediately(const UpdateContext& ctx, const Position& pos) {
        if (g_gpu) {
            mirror(this).applyImmediately(convert(ctx), pack(pos));
        }
        spxOrigApplyImmediately(this, ctx, pos);
    }

    i32 eval::NnueState::evaluate(const Position& pos, Color stm) {
        if (!g_gpu) {
            ++g_evals;
            return spxOrigEvaluate(this, pos, stm);
        }
        auto& m = mirror(this);
        const auto rec = pack(pos);
        const auto pending = static_cast<unsigned long long>(m.pending());
        g_pendingSum += pending;
        g_pendingMax = std::max(g_pendingMax, pending);
        i32 value;
        if (blackToMove(rec) != blackToMove(m.position())) {
            // a null move (thread.cpp:28-44: no NNUE push): same board, the other side to move - a child with an empty delta
            m.push(rec);
            value = m.evaluate();
            m.pop();
            ++g_nullMoveEvals;
        } else {
            value = m.evaluate();
        }
        ++g_evals;
        if (g_check && value != spxOrigEvaluate(this, pos, stm)) {
            ++g_mismatches;
        }
        return value;
    }

    i32 eval::NnueState::evaluateOnce(const Position& pos, Color stm) {
        if (!g_gpu) {
            ++g_evals;
            return spxOrigEvaluateOnce(pos, stm);
        }
        auto rec = pack(pos);
        if ((stm == Colors::kBlack) != blackToMove(rec)) {
            reinterpret_cast<unsigned char*>(&rec)[24] ^= 0x80u;
        }
        const i32 value = shared().evaluateOnce(rec);
        ++g_evals;
        if (g_check && value != spxOrigEvaluateOnce(pos, stm)) {
            ++g_mismatches;
        }
        return value;
    }
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

    std::vector<Move> legalMoves(const Position& pos) {
        ScoredMoveList generated{};
        generateAll(generated, pos);
        std::vector<Move> moves;
        for (const auto& [move, score] : generated) {
            if (pos.isLegal(move)) {
                moves.push_back(move);
            }
        }
        return moves;
    }

    void setMode(const std::string& mode) {
        g_gpu = mode != "cpu";
        g_check = mode == "both";
    }

    void report(const char* what) {
        std::printf("# %s: %llu NNUE evaluations through %s", what, g_evals, g_gpu ? "libspx_nnue (GPU)" : "the reference's CPU path");
        if (g_gpu) {
            std::printf(", %llu pushes, %llu after a null move, pending plies per evaluate: mean %.2f max %llu", g_pushes,
                        g_nullMoveEvals, g_evals ? double(g_pendingSum) / double(g_evals) : 0.0, g_pendingMax);
        }
        if (g_check) {
            std::printf(", checked against the CPU value: %llu mismatches", g_mismatches);
        }
        std::printf("\n");
    }
} // namespace

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s bench <depth> cpu|gpu|both | raweval cpu|gpu | game <plies> <seed>\n", argv[0]);
        return 64;
    }
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
    const std::string cmd = argv[1];
    try {
        if (cmd == "bench" && argc >= 4) {
            setMode(argv[3]);
            bench::run(std::atoi(argv[2]), bench::kDefaultBenchTtSize);  // prints "<n> nodes <nps> nps"
            std::fflush(stdout);
            report("bench");
            return g_mismatches ? 3 : 0;
        }
        if (cmd == "raweval") {
            setMode(argv[2]);
            opts::mutableOpts().chess960 = true;  // harmless for standard FENs; required for DFRC castling rights
            std::string fen;
            while (std::getline(std::cin, fen)) {
                const auto pos = Position::fromFen(fen);
                if (!pos) {
                    std::printf("ERR bad fen\n");
                    continue;
                }
                std::printf("%d\n", eval::staticEvalOnce(*pos));  // uci.cpp:797-800
            }
            report("raweval");
            return 0;
        }
        if (cmd == "game" && argc >= 4) {
            g_gpu = true;
            const u32 plies = static_cast<u32>(std::atoi(argv[2]));
            SplitMix64 rng{static_cast<u64>(std::strtoull(argv[3], nullptr, 10))};
            opts::mutableOpts().chess960 = true;
            u32 played = 0, games = 0;
            unsigned long long broken = 0, cpuBroken = 0;
            while (played < plies) {
                auto pos = (games & 1) ? *Position::fromDfrcIndex(rng.below(960 * 960)) : Position::startpos();
                ++games;
                eval::NnueState state{};
                state.setNetwork(eval::getNetwork(0));
                state.reset(pos);
                for (u32 ply = 0; ply < 200 && played < plies; ++ply) {
                    const auto moves = legalMoves(pos);
                    if (moves.empty()) {
                        break;
                    }
                    const auto move = moves[rng.below(static_cast<u32>(moves.size()))];
                    // datagen.cpp:257-262
                    eval::UpdateContext ctx{};
                    pos = pos.applyMove(move, eval::BoardObserver{ctx});
                    state.applyImmediately(ctx, pos);
                    const i32 once = eval::staticEvalOnce(pos);        // GPU: full refresh of this position
                    const i32 viaState = eval::staticEval(pos, state);  // GPU: the incrementally maintained accumulators
                    g_gpu = false;
                    const i32 cpuOnce = eval::staticEvalOnce(pos);
                    g_gpu = true;
                    broken += once != viaState;
                    cpuBroken += once != cpuOnce;
                    ++played;
                }
            }
            std::printf("game: %u plies in %u games: staticEvalOnce(pos) == staticEval(pos, nnueState) broken %llu times on the GPU "
                        "state; GPU value != CPU value %llu times\n", played, games, broken, cpuBroken);
            report("game");
            return (broken || cpuBroken) ? 3 : 0;
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 4;
    }
    std::fprintf(stderr, "unknown command\n");
    return 64;
}
