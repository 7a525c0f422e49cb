// spx_raweval - a native host program on top of include/spx_nnue.hpp (the C++ mirror of Stormphrax's eval interface).
//
//   spx_raweval [--preset tame|wild|extreme|realistic | --net file.nnue]             FENs on stdin -> "<raw> <static>" per line
//   spx_raweval [...] --walk <seed> <nodes> <fen>                          make/unmake walk through NnueState
//   spx_raweval [...] --datagen <seed> <plies> <fen>                       datagen's use: applyImmediately after every move
//
// The first mode is what `position fen ...` + `raweval` do in the reference's UCI loop (src/uci/uci.cpp:774-800: raw =
// NnueState::evaluateOnce, static = eval::staticEvalOnce). The second drives the accumulator STACK the way a search
// does - push / pop / evaluate in depth-first order with lazily pending plies - and checks the reference's own
// invariant at every visited node: evaluate() == evaluateOnce(position) (the assert at src/datagen/datagen.cpp:262); it
// also times evaluate() by the number of plies that were pending (one launch whatever their number). The third plays
// random moves the way the data generator advances its root (src/datagen/datagen.cpp:257-262): the move is made with the
// observer (spx_pos_apply_uci_observed == Position::applyMove(move, BoardObserver{ctx})), NnueState::applyImmediately
// consumes the captured UpdateContext, and evaluate() must equal evaluateOnce() after every move.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <string>
#include <vector>

#include "../include/spx_nnue.hpp"

namespace {

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

struct Walk {
    spx_nnue::NnueState& state;
    SplitMix64 rng;
    uint64_t budget, visited = 0, evaluated = 0, mismatches = 0, maxDepth = 0;
    double evalSeconds[9] = {};   // evaluate() time by pending plies (0 .. 7, 8+)
    uint64_t evalCalls[9] = {};

    void visit(int depthLeft) {
        if (visited >= budget) return;
        ++visited;
        maxDepth = std::max<uint64_t>(maxDepth, state.depth());
        // a search does not evaluate every node it passes through: leave some plies pending (lazy updates)
        if (rng.below(4) != 0) {
            const size_t pending = std::min<size_t>(state.pending(), 8);
            const auto t0 = std::chrono::steady_clock::now();
            const int32_t inc = state.evaluate();
            evalSeconds[pending] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            ++evalCalls[pending];
            const int32_t once = state.evaluateOnce(state.position());
            ++evaluated;
            if (inc != once) {
                ++mismatches;
                char fen[128];
                spx_pos_to_fen(&state.position(), fen, sizeof(fen));
                std::fprintf(stderr, "MISMATCH at depth %zu: evaluate %d, evaluateOnce %d, %s\n", state.depth(), inc, once, fen);
            }
        }
        if (depthLeft == 0) return;
        uint16_t moves[256];
        spx_packed_pos children[256];
        int n = 0, inCheck = 0;
        spx_nnue::check(spx_pos_legal_moves(&state.position(), moves, children, &n, &inCheck));
        if (n == 0) return;
        // (long forced lines now and then: several plies stay pending, as below a search's reductions)
        const int branch = 1 + int(rng.below(3));
        for (int b = 0; b < branch && visited < budget; ++b) {
            state.push(children[rng.below(uint32_t(n))]);
            visit(depthLeft - 1);
            state.pop();
        }
    }
};

}  // namespace

int main(int argc, char** argv) {
    try {
        int preset = 0;
        std::string netPath;
        int i = 1;
        for (; i < argc; ++i) {
            if (!std::strcmp(argv[i], "--preset") && i + 1 < argc) {
                const std::string p = argv[++i];
                preset = p == "wild" ? 1 : p == "extreme" ? 2 : p == "realistic" ? 3 : 0;
            } else if (!std::strcmp(argv[i], "--net") && i + 1 < argc) {
                netPath = argv[++i];
            } else {
                break;
            }
        }
        spx_nnue::Network net = [&] {
            if (netPath.empty()) return spx_nnue::Network::synthetic(preset);
            std::ifstream f(netPath, std::ios::binary);
            std::vector<char> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
            return spx_nnue::Network(blob.data(), blob.size());
        }();
        spx_nnue::NnueState state(net, 0, 4096);

        if (i < argc && !std::strcmp(argv[i], "--walk")) {
            if (i + 3 >= argc) {
                std::fprintf(stderr, "usage: spx_raweval --walk <seed> <nodes> <fen>\n");
                return 2;
            }
            const uint64_t seed = std::strtoull(argv[i + 1], nullptr, 10), nodes = std::strtoull(argv[i + 2], nullptr, 10);
            std::string fen;
            for (int k = i + 3; k < argc; ++k) fen += (k > i + 3 ? " " : "") + std::string(argv[k]);
            spx_packed_pos root;
            spx_nnue::check(spx_pos_from_fen(fen.c_str(), &root));
            state.reset(root);
            Walk walk{state, SplitMix64{seed}, nodes};
            while (walk.visited < nodes) {
                const uint64_t before = walk.visited;
                walk.visit(40);
                if (walk.visited == before) break;
            }
            std::printf("walk: %llu nodes, %llu evaluated, max depth %llu, %llu mismatches (net %s)\n",
                        (unsigned long long)walk.visited, (unsigned long long)walk.evaluated,
                        (unsigned long long)walk.maxDepth, (unsigned long long)walk.mismatches, net.name());
            std::printf("evaluate_us_by_pending_plies:");
            for (int k = 0; k < 9; ++k) {
                if (walk.evalCalls[k]) std::printf(" %d%s=%.1f(%llu)", k, k == 8 ? "+" : "", 1e6 * walk.evalSeconds[k] / double(walk.evalCalls[k]),
                                                   (unsigned long long)walk.evalCalls[k]);
            }
            std::printf("\n");
            return walk.mismatches ? 1 : 0;
        }
        if (i < argc && !std::strcmp(argv[i], "--datagen")) {
            if (i + 3 >= argc) {
                std::fprintf(stderr, "usage: spx_raweval --datagen <seed> <plies> <fen>\n");
                return 2;
            }
            SplitMix64 rng{std::strtoull(argv[i + 1], nullptr, 10)};
            const uint64_t plies = std::strtoull(argv[i + 2], nullptr, 10);
            std::string fen;
            for (int k = i + 3; k < argc; ++k) fen += (k > i + 3 ? " " : "") + std::string(argv[k]);
            spx_packed_pos pos;
            spx_nnue::check(spx_pos_from_fen(fen.c_str(), &pos));
            uint64_t played = 0, mismatches = 0, games = 0;
            double seconds = 0;
            while (played < plies) {
                spx_packed_pos cur = pos;
                state.reset(cur);  // NnueState::reset at the start of every game (datagen.cpp:179)
                ++games;
                for (int ply = 0; ply < 200 && played < plies; ++ply) {
                    uint16_t moves[256];
                    int n = 0, inCheck = 0;
                    spx_nnue::check(spx_pos_legal_moves(&cur, moves, nullptr, &n, &inCheck));
                    if (n == 0) break;
                    const uint16_t w = moves[rng.below(uint32_t(n))];
                    const int from = w & 63, to = (w >> 6) & 63, kind = w >> 14;  // viriformat.cpp:37-52: 3 = promotion
                    char uci[6] = {char('a' + (from & 7)), char('1' + (from >> 3)), char('a' + (to & 7)), char('1' + (to >> 3)), 0, 0};
                    if (kind == 3) uci[4] = "nbrq"[(w >> 12) & 3];
                    spx_packed_pos next;
                    spx_move_delta delta;
                    spx_nnue::check(spx_pos_apply_uci_observed(&cur, uci, &next, &delta));
                    const auto t0 = std::chrono::steady_clock::now();
                    state.applyImmediately(delta, next);
                    seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    cur = next;
                    ++played;
                    if (state.evaluate() != state.evaluateOnce(cur)) {  // the reference's own assert (datagen.cpp:262)
                        ++mismatches;
                        char f[128];
                        spx_pos_to_fen(&cur, f, sizeof(f));
                        std::fprintf(stderr, "MISMATCH after %s: %s\n", uci, f);
                    }
                    if (state.depth() != 0) ++mismatches;  // applyImmediately never grows the stack
                }
            }
            std::printf("datagen: %llu moves in %llu games through applyImmediately, %llu mismatches, %.1f us per applyImmediately "
                        "(net %s)\n", (unsigned long long)played, (unsigned long long)games, (unsigned long long)mismatches,
                        1e6 * seconds / double(played ? played : 1), net.name());
            return mismatches ? 1 : 0;
        }

        std::string line;
        while (std::getline(std::cin, line)) {
            if (line.empty()) continue;
            spx_packed_pos pos;
            if (spx_pos_from_fen(line.c_str(), &pos) != SPX_OK) {
                std::printf("error %s\n", spx_last_error());
                continue;
            }
            std::printf("%d %d\n", state.evaluateOnce(pos), state.staticEvalOnce(pos));
        }
        return 0;
    } catch (const spx_nnue::Error& e) {
        std::fprintf(stderr, "spx_raweval: %s (status %d)\n", e.what(), e.status);
        return 3;
    }
}
