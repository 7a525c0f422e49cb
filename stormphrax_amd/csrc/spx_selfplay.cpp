// Batched self-play driver (BASELINE config 4 shape): many concurrent games, every game's candidate moves evaluated in
// ONE GPU batch per ply through the incremental path (parent slot -> one child slot per legal move, fused update+eval).
//
// Mirrors the control flow of the reference's generator (src/datagen/datagen.cpp:96-318): random 8-9 ply opening
// (:153-171), accumulator reset (:179), per move a "search", win/draw adjudication counters with the reference's
// constants (:74-94,224-252), terminal detection, viriformat game records (src/datagen/viriformat.cpp:28-63). The one
// deliberate difference: the reference runs a ~24 000-node alpha-beta search per move (out of scope); here
// the "search" is depth 1 - score(move) = -eval(child) - which is exactly the part that batches. Host work (move
// generation for every child) is spread over std::threads; games are independent, so multi-GPU = one process per GPU
// with its own slice of games and no communication.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spx_nnue.h"
#include "spx_chess.h"
#include "spx_internal.h"

namespace spx {

namespace {

// adjudication constants of the reference (datagen.cpp:78-88); scores here are raw network outputs
constexpr int kWinAdjMinScore = 1250, kDrawAdjMaxScore = 10;
constexpr uint32_t kDrawAdjMinPlies = 70, kWinAdjPlyCount = 5, kDrawAdjPlyCount = 10;

struct Rng {
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

uint64_t boardHash(const Board& b) {  // repetition detection only; not a Zobrist key
    uint64_t h = 0xcbf29ce484222325ull ^ uint64_t(b.stm);
    for (int i = 0; i < 12; ++i) h = (h ^ b.pieces[i]) * 0x100000001b3ull + (h >> 29);
    return h ^ uint64_t(b.ep + 1) ^ (uint64_t(uint8_t(b.castleRook[0][0] + 1)) << 8) ^
           (uint64_t(uint8_t(b.castleRook[0][1] + 1)) << 16) ^ (uint64_t(uint8_t(b.castleRook[1][0] + 1)) << 24) ^
           (uint64_t(uint8_t(b.castleRook[1][1] + 1)) << 32);
}

uint16_t viriMove(const Move& m) {  // viriformat.cpp:37-52
    static const uint16_t kTypes[4] = {0x0000, 0xC000, 0x8000, 0x4000};
    return uint16_t(m.from | (m.to << 6) | ((m.kind == kPromotion ? m.promo - 1 : 0) << 12) | kTypes[m.kind]);
}

struct Game {
    Board board;
    spx_packed_pos initial;
    std::vector<uint16_t> moves;
    std::vector<int16_t> scores;
    std::vector<uint64_t> history;
    uint32_t slot = 0;
    uint32_t winPlies = 0, lossPlies = 0, drawPlies = 0, plies = 0;
    bool active = false;
    // per-step scratch
    std::vector<Move> legal;
    size_t firstChild = 0;
};

void startGame(Game& g, Rng& rng, bool dfrc, uint32_t baseOpeningPlies) {
    std::vector<Move> moves;
    for (;;) {
        g.board = dfrc ? dfrcStart(rng.below(960), rng.below(960)) : startpos();
        const uint32_t count = baseOpeningPlies + uint32_t(rng.next() >> 63);  // 8 + coin flip (datagen.cpp:153)
        bool dead = false;
        for (uint32_t i = 0; i < count && !dead; ++i) {
            generateLegal(g.board, moves);
            dead = moves.empty();
            if (!dead) makeMove(g.board, moves[rng.below(uint32_t(moves.size()))]);
        }
        generateLegal(g.board, moves);
        if (!dead && !moves.empty()) break;
    }
    packBoard(g.board, g.initial);
    g.moves.clear();
    g.scores.clear();
    g.history.clear();
    g.winPlies = g.lossPlies = g.drawPlies = 0;
    g.plies = 0;
    g.active = true;
}

}  // namespace
}  // namespace spx

using namespace spx;

extern "C" int spx_selfplay_run(spx_ctx* ctx, const spx_selfplay_params* p, const char* out_path,
                                spx_selfplay_stats* stats) {
    if (!ctx || !p || !stats || p->n_games == 0 || p->target_games == 0) {
        setError("spx_selfplay_run: invalid argument");
        return SPX_ERR_INVALID_ARG;
    }
    const uint32_t G = p->n_games;
    // threads are spawned per ply (no pool): beyond ~32 the spawn cost outweighs the move generation they share
    const uint32_t threads =
        std::max(1u, p->host_threads ? p->host_threads : std::min(32u, std::thread::hardware_concurrency()));
    const size_t maxChildren = size_t(G) * 64;  // scratch slots per step parity (more children are processed in chunks)
    int rc = spx_acc_reserve(ctx, size_t(G) + 2 * maxChildren);
    if (rc != SPX_OK) return rc;
    FILE* out = nullptr;
    if (out_path && out_path[0]) {
        out = std::fopen(out_path, "wb");
        if (!out) {
            setError(std::string("spx_selfplay_run: cannot open ") + out_path);
            return SPX_ERR_INVALID_ARG;
        }
    }
    std::memset(stats, 0, sizeof(*stats));
    Rng rng{p->seed};
    std::vector<Game> games(G);
    std::vector<spx_packed_pos> childPos;
    std::vector<uint32_t> parents, children, refreshSlots;
    std::vector<spx_packed_pos> refreshPos;
    std::vector<int32_t> evals;
    const auto t0 = std::chrono::steady_clock::now();
    double gpuSeconds = 0.0;
    uint64_t started = 0;
    uint32_t step = 0;

    auto finishGame = [&](Game& g, uint8_t outcome) {
        g.initial.wdl = outcome;
        if (out) {
            std::fwrite(&g.initial, sizeof(g.initial), 1, out);
            for (size_t i = 0; i < g.moves.size(); ++i) {
                std::fwrite(&g.moves[i], 2, 1, out);
                std::fwrite(&g.scores[i], 2, 1, out);
            }
            const uint32_t zero = 0;
            std::fwrite(&zero, 4, 1, out);
        }
        stats->games += 1;
        stats->positions += g.moves.size();
        stats->outcomes[outcome] += 1;
        g.active = false;
    };

    for (;;) {
        // (re)start games in idle slots and full-refresh their accumulators (NnueState::reset, datagen.cpp:179)
        refreshSlots.clear();
        refreshPos.clear();
        for (uint32_t i = 0; i < G; ++i) {
            Game& g = games[i];
            if (!g.active && started < p->target_games) {
                startGame(g, rng, p->dfrc != 0, p->opening_plies ? p->opening_plies : 8);
                g.slot = i;
                ++started;
                refreshSlots.push_back(i);
                spx_packed_pos rec;
                packBoard(g.board, rec);
                refreshPos.push_back(rec);
            }
        }
        if (!refreshSlots.empty()) {
            const auto g0 = std::chrono::steady_clock::now();
            for (size_t lo = 0; lo < refreshSlots.size() && rc == SPX_OK; lo += ctxMaxBatch(ctx)) {
                const size_t m = std::min(ctxMaxBatch(ctx), refreshSlots.size() - lo);
                rc = spx_acc_refresh(ctx, refreshPos.data() + lo, refreshSlots.data() + lo, m);
            }
            gpuSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - g0).count();
            if (rc != SPX_OK) break;
        }
        bool any = false;
        for (const Game& g : games) any = any || g.active;
        if (!any) break;

        // host: legal moves of every active game (threads own contiguous game ranges)
        {
            std::vector<std::thread> pool;
            const uint32_t per = (G + threads - 1) / threads;
            for (uint32_t t = 0; t < threads; ++t) {
                pool.emplace_back([&, t] {
                    for (uint32_t i = t * per; i < std::min(G, (t + 1) * per); ++i) {
                        if (games[i].active) generateLegal(games[i].board, games[i].legal);
                    }
                });
            }
            for (auto& th : pool) th.join();
        }
        // terminal positions: mate / stalemate (datagen.cpp:213-221)
        size_t total = 0;
        for (Game& g : games) {
            if (!g.active) continue;
            if (g.legal.empty()) {
                const uint8_t outcome = g.board.inCheck() ? (g.board.stm == 0 ? 2 : 0) : 1;
                finishGame(g, outcome);
                continue;
            }
            g.firstChild = total;
            total += g.legal.size();
        }
        if (total == 0) continue;
        childPos.resize(total);
        parents.resize(total);
        children.resize(total);
        evals.resize(total);
        const uint32_t region = uint32_t(G + (step & 1) * maxChildren);
        {
            std::vector<std::thread> pool;
            const uint32_t per = (G + threads - 1) / threads;
            for (uint32_t t = 0; t < threads; ++t) {
                pool.emplace_back([&, t] {
                    for (uint32_t i = t * per; i < std::min(G, (t + 1) * per); ++i) {
                        Game& g = games[i];
                        if (!g.active) continue;
                        for (size_t k = 0; k < g.legal.size(); ++k) {
                            Board next = g.board;
                            makeMove(next, g.legal[k]);
                            packBoard(next, childPos[g.firstChild + k]);
                            parents[g.firstChild + k] = g.slot;
                        }
                    }
                });
            }
            for (auto& th : pool) th.join();
        }
        // device: one update+eval batch per chunk of the context's capacity; child slots cycle inside this step's region
        // (a chunk never exceeds maxChildren, and only the CHOSEN child's slot has to survive until the next step: when
        // slots are recycled within a step the chosen child is re-materialised below)
        std::vector<uint8_t> slotValid(total, 1);
        {
            const auto g0 = std::chrono::steady_clock::now();
            size_t done = 0;
            while (done < total && rc == SPX_OK) {
                const size_t n = std::min(total - done, maxChildren);
                for (size_t k = 0; k < n; ++k) children[done + k] = region + uint32_t(k);
                if (done > 0) std::fill(slotValid.begin(), slotValid.begin() + done, 0);  // earlier chunk overwritten
                size_t sub = 0;
                while (sub < n && rc == SPX_OK) {  // respect the context's batch capacity
                    const size_t m = std::min(n - sub, ctxMaxBatch(ctx));
                    rc = spx_acc_update_eval(ctx, parents.data() + done + sub, children.data() + done + sub,
                                             childPos.data() + done + sub, m, evals.data() + done + sub);
                    sub += m;
                }
                done += n;
            }
            gpuSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - g0).count();
            stats->evals += total;
        }
        if (rc != SPX_OK) break;

        // pick moves, adjudicate (datagen.cpp:224-252), advance
        std::vector<uint32_t> fixParents, fixChildren;
        std::vector<spx_packed_pos> fixPos;
        for (uint32_t i = 0; i < G; ++i) {
            Game& g = games[i];
            if (!g.active || g.legal.empty()) continue;
            int best = INT32_MIN;
            for (size_t k = 0; k < g.legal.size(); ++k) best = std::max(best, -evals[g.firstChild + k]);
            // exploration: uniformly among the moves within temperature_cp of the best (0 = greedy, first best)
            size_t pick = 0, seen = 0;
            for (size_t k = 0; k < g.legal.size(); ++k) {
                if (-evals[g.firstChild + k] >= best - p->temperature_cp) {
                    ++seen;
                    if (rng.below(uint32_t(seen)) == 0) pick = k;
                    if (p->temperature_cp == 0) break;
                }
            }
            const int score = -evals[g.firstChild + pick];
            g.moves.push_back(viriMove(g.legal[pick]));
            g.scores.push_back(int16_t(std::max(-32000, std::min(32000, std::abs(score) <= 2 ? 0 : score))));
            const int whiteScore = g.board.stm ? score : -score;
            uint8_t outcome = 255;
            if (whiteScore > kWinAdjMinScore) {
                ++g.winPlies;
                g.lossPlies = g.drawPlies = 0;
            } else if (whiteScore < -kWinAdjMinScore) {
                ++g.lossPlies;
                g.winPlies = g.drawPlies = 0;
            } else if (g.plies >= kDrawAdjMinPlies && std::abs(score) < kDrawAdjMaxScore) {
                ++g.drawPlies;
                g.winPlies = g.lossPlies = 0;
            } else {
                g.winPlies = g.lossPlies = g.drawPlies = 0;
            }
            if (g.winPlies >= kWinAdjPlyCount) outcome = 2;
            else if (g.lossPlies >= kWinAdjPlyCount) outcome = 0;
            else if (g.drawPlies >= kDrawAdjPlyCount) outcome = 1;

            g.history.push_back(boardHash(g.board));
            makeMove(g.board, g.legal[pick]);
            ++g.plies;
            const size_t idx = g.firstChild + pick;
            if (slotValid[idx]) {
                g.slot = children[idx];
            } else {  // its scratch slot was recycled by a later chunk of this step: materialise it again
                fixParents.push_back(g.slot);
                fixChildren.push_back(i);  // the game's home slot
                fixPos.push_back(childPos[idx]);
                g.slot = i;
            }
            // draws: 50-move rule, threefold repetition, ply cap (Position::isDrawn analogue, datagen.cpp:264-268)
            const uint64_t h = boardHash(g.board);
            const size_t reps = size_t(std::count(g.history.begin(), g.history.end(), h));
            if (outcome == 255 && (g.board.halfmove >= 100 || reps >= 2 || g.plies >= p->max_plies)) outcome = 1;
            if (outcome != 255) finishGame(g, outcome);
        }
        if (!fixParents.empty()) {
            // parent == child home slot is not allowed inside one batch: go through the record alone (full refresh)
            for (size_t lo = 0; lo < fixChildren.size() && rc == SPX_OK; lo += ctxMaxBatch(ctx)) {
                const size_t m = std::min(ctxMaxBatch(ctx), fixChildren.size() - lo);
                rc = spx_acc_refresh(ctx, fixPos.data() + lo, fixChildren.data() + lo, m);
            }
            if (rc != SPX_OK) break;
        }
        ++step;
        stats->steps = step;
    }
    if (out) std::fclose(out);
    stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stats->gpu_seconds = gpuSeconds;
    return rc;
}
