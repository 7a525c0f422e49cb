// Batched self-play driver (BASELINE config 4 shape): many concurrent games, every game's candidate moves evaluated in
// ONE GPU batch per ply through the incremental path (parent slot -> one child slot per legal move, fused update+eval).
//
// Mirrors the control flow of the reference's generator (src/datagen/datagen.cpp:96-318): random 8-9 ply opening
// (:153-171), accumulator reset (:179), per move a "search", win/draw adjudication counters with the reference's
// constants (:74-94,224-252), terminal detection, viriformat game records (src/datagen/viriformat.cpp:28-63). The one
// deliberate difference: the reference runs a ~24 000-node alpha-beta search per move (out of scope); here
// the "search" is depth 1 - score(move) = -eval(child) - which is exactly the part that batches.
//
// Host side: a persistent worker pool runs the per-game phases (legal moves, child records, move choice +
// adjudication) over contiguous game ranges. The games are split into two halves that alternate: while the GPU
// evaluates the children of one half (fused update+eval batch, issued from a helper thread), the pool prepares the
// other half - the evaluator and the move generator overlap. Every game owns its RNG stream (seeded when the game
// starts), so results do not depend on the thread count. Games are independent, so multi-GPU = one process per GPU
// with its own slice of games and no communication.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <future>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/spx_nnue.h"
#include "../../include/spx_nnue_dev.h"
#include "spx_chess.h"
#include "spx_device_math.h"
#include "spx_internal.h"
#include "spx_kernels.h"

namespace spx {

namespace {

// (adjudication constants and the counter ladder of datagen.cpp:78-88,224-252: spx_device_math.h, shared with the device)

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

// CPUs this process may actually use: hardware threads capped by the container's cgroup quota (the MI355X boxes show
// 256 logical CPUs under a quota of 16 - oversubscribing the quota is slower than respecting it)
uint32_t usableCpus() {
    uint32_t n = std::max(1u, std::thread::hardware_concurrency());
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        long period = 0;
        if (std::fscanf(f, "%31s %ld", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0) {
            n = std::min(n, uint32_t(std::max(1l, std::atol(quota) / period)));
        }
        std::fclose(f);
    }
    // one process per GPU (torch.distributed.run exports LOCAL_WORLD_SIZE): the ranks of a node share the CPUs
    if (const char* env = std::getenv("LOCAL_WORLD_SIZE")) {
        const long ranks = std::atol(env);
        if (ranks > 1) n = std::max(1u, n / uint32_t(ranks));
    }
    return n;
}

struct Game {
    Board board;
    spx_packed_pos initial;
    std::vector<uint16_t> moves;
    std::vector<int16_t> scores;
    std::vector<uint64_t> history;
    Rng rng{0};
    uint32_t slot = 0;
    uint32_t winPlies = 0, lossPlies = 0, drawPlies = 0, plies = 0;
    bool active = false;
    // per-step scratch
    std::vector<Move> legal;
    size_t firstChild = 0;
    uint8_t outcome = 255;   // set by a worker when the game ended this step (255 = still running)
    bool discard = false;    // the opening failed the verification filter (datagen.cpp:176-190)
    uint32_t startPly = 0;   // Position::plyFromStartpos of the initial position
    bool needsFix = false;   // the chosen child's scratch slot was recycled: re-materialise into the home slot
    spx_packed_pos fixPos;
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
    g.startPly = plyFromStartpos(g.initial.fullmove, g.board.stm == 1);
    g.discard = false;
    g.moves.clear();
    g.scores.clear();
    g.history.clear();
    g.rng = Rng{rng.next()};
    g.winPlies = g.lossPlies = g.drawPlies = 0;
    g.plies = 0;
    g.outcome = 255;
    g.needsFix = false;
    g.active = true;
}

// Persistent workers; run(fn) executes fn(worker) on every worker and returns when all are done. Phases follow each
// other within microseconds, so workers first spin on the generation counter and only then sleep on the condition
// variable (a mutex hand-off per phase per worker costs more than the phases themselves beyond ~32 threads).
class Pool {
public:
    explicit Pool(uint32_t n) : n_(n) {
        for (uint32_t t = 1; t < n_; ++t) threads_.emplace_back([this, t] { loop(t); });
    }
    ~Pool() {
        stop_.store(true, std::memory_order_release);
        publish();
        for (auto& th : threads_) th.join();
    }
    uint32_t size() const {
        return n_;
    }
    void run(const std::function<void(uint32_t)>& fn) {
        job_ = &fn;
        pending_.store(n_ - 1, std::memory_order_release);
        publish();
        fn(0);  // the caller is worker 0
        while (pending_.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
    }

private:
    void publish() {
        generation_.fetch_add(1, std::memory_order_acq_rel);
        if (sleepers_.load(std::memory_order_acquire) != 0) {
            std::lock_guard<std::mutex> lock(m_);
            wake_.notify_all();
        }
    }
    void loop(uint32_t t) {
        uint64_t seen = 0;
        for (;;) {
            uint32_t spins = 0;
            while (generation_.load(std::memory_order_acquire) == seen) {
                if (++spins < 20000) {  // ~100 us of polling before going to sleep
                    __builtin_ia32_pause();
                    continue;
                }
                std::unique_lock<std::mutex> lock(m_);
                sleepers_.fetch_add(1, std::memory_order_acq_rel);
                wake_.wait(lock, [&] { return generation_.load(std::memory_order_acquire) != seen; });
                sleepers_.fetch_sub(1, std::memory_order_acq_rel);
            }
            seen = generation_.load(std::memory_order_acquire);
            if (stop_.load(std::memory_order_acquire)) return;
            (*job_)(t);
            pending_.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
    uint32_t n_;
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable wake_;
    const std::function<void(uint32_t)>* job_ = nullptr;
    std::atomic<uint32_t> pending_{0}, sleepers_{0};
    std::atomic<uint64_t> generation_{0};
    std::atomic<bool> stop_{false};
};

// Batch buffer in page-locked memory (spx_host_alloc): the GPU batch's copies are plain DMA. Grows geometrically.
template <typename T>
class PinnedBuf {
public:
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf() {
        spx_host_free(data_);
    }
    bool ensure(size_t n) {  // contents are not preserved
        if (n <= cap_) return true;
        spx_host_free(data_);
        cap_ = std::max(n, cap_ * 2);
        data_ = static_cast<T*>(spx_host_alloc(cap_ * sizeof(T)));
        if (!data_) cap_ = 0;
        return data_ != nullptr;
    }
    T* data() {
        return data_;
    }
    T& operator[](size_t i) {
        return data_[i];
    }

private:
    T* data_ = nullptr;
    size_t cap_ = 0;
};

// One half of the games: its own child buffers and its own scratch-slot regions, so that its GPU batch can be in
// flight while the other half is being prepared.
struct Half {
    uint32_t begin = 0, end = 0;   // game range
    uint32_t regionBase = 0;       // first scratch slot (two regions of maxChildren, alternating per step)
    size_t maxChildren = 0;
    uint32_t step = 0;
    size_t total = 0;
    PinnedBuf<spx_packed_pos> childPos;
    PinnedBuf<uint32_t> parents, children;
    PinnedBuf<int32_t> evals;
    std::vector<uint8_t> slotValid;
    std::future<int> pending;      // the in-flight GPU batch
    bool inFlight = false;
};

}  // namespace
}  // namespace spx

using namespace spx;

// ---------------------------------------------------------------------------------------------------------------------
// Host-movegen path (SPX_SELFPLAY_HOST_MOVEGEN): legal moves and child records from the host chess core.
// ---------------------------------------------------------------------------------------------------------------------
static int runHostMovegen(spx_ctx* ctx, const spx_selfplay_params* p, const char* out_path, spx_selfplay_stats* stats) {
    const uint32_t G = p->n_games;
    const uint32_t nThreads = std::max(1u, std::min(p->host_threads ? p->host_threads : std::min(16u, usableCpus()), G));
    const uint32_t nHalves = G >= 2 ? 2 : 1;
    std::vector<Half> halves(nHalves);
    uint32_t nextSlot = G;
    for (uint32_t h = 0; h < nHalves; ++h) {
        halves[h].begin = uint32_t(uint64_t(G) * h / nHalves);
        halves[h].end = uint32_t(uint64_t(G) * (h + 1) / nHalves);
        halves[h].maxChildren = size_t(halves[h].end - halves[h].begin) * 64;  // larger steps are processed in chunks
        halves[h].regionBase = nextSlot;
        nextSlot += uint32_t(2 * halves[h].maxChildren);
    }
    int rc = spx_acc_reserve(ctx, nextSlot);
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
    Pool pool(nThreads);
    std::vector<uint32_t> refreshSlots;
    std::vector<spx_packed_pos> refreshPos;
    const auto t0 = std::chrono::steady_clock::now();
    double gpuSeconds = 0.0;
    uint64_t started = 0;
    std::mutex gpuMutex;  // one context: GPU calls are serialised (the helper thread vs. refreshes on this thread)

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
        g.outcome = 255;
    };
    // workers own contiguous slices of a half's game range
    auto forGames = [&](const Half& h, const std::function<void(Game&, uint32_t)>& fn) {
        pool.run([&](uint32_t t) {
            const uint32_t count = h.end - h.begin, per = (count + pool.size() - 1) / pool.size();
            const uint32_t lo = h.begin + std::min(count, t * per), hi = h.begin + std::min(count, (t + 1) * per);
            for (uint32_t i = lo; i < hi; ++i) fn(games[i], i);
        });
    };
    auto refresh = [&](const std::vector<spx_packed_pos>& pos, const std::vector<uint32_t>& slots) {
        std::lock_guard<std::mutex> lock(gpuMutex);
        const auto g0 = std::chrono::steady_clock::now();
        int r = SPX_OK;
        for (size_t lo = 0; lo < slots.size() && r == SPX_OK; lo += ctxMaxBatch(ctx)) {
            const size_t m = std::min(ctxMaxBatch(ctx), slots.size() - lo);
            r = spx_acc_refresh(ctx, pos.data() + lo, slots.data() + lo, m);
        }
        gpuSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - g0).count();
        return r;
    };

    // ---- phase 1 of a half's step: (re)start games, legal moves, child records; then launch the GPU batch ----
    auto prepare = [&](Half& h) -> int {
        refreshSlots.clear();
        refreshPos.clear();
        for (uint32_t i = h.begin; i < h.end; ++i) {
            Game& g = games[i];
            if (!g.active && started < p->target_games) {
                // (re)start and full-refresh the accumulators (NnueState::reset, datagen.cpp:179)
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
            const int r = refresh(refreshPos, refreshSlots);
            if (r != SPX_OK) return r;
        }
        forGames(h, [](Game& g, uint32_t) {
            if (g.active) generateLegal(g.board, g.legal);
        });
        // terminal positions: mate / stalemate (datagen.cpp:213-221)
        h.total = 0;
        for (uint32_t i = h.begin; i < h.end; ++i) {
            Game& g = games[i];
            if (!g.active) continue;
            if (g.legal.empty()) {
                finishGame(g, g.board.inCheck() ? (g.board.stm == 0 ? 2 : 0) : 1);
                continue;
            }
            g.firstChild = h.total;
            h.total += g.legal.size();
        }
        if (h.total == 0) return SPX_OK;
        if (!h.childPos.ensure(h.total) || !h.parents.ensure(h.total) || !h.children.ensure(h.total) ||
            !h.evals.ensure(h.total)) {
            return SPX_ERR_HIP;
        }
        h.slotValid.assign(h.total, 1);
        forGames(h, [&h](Game& g, uint32_t) {
            if (!g.active) return;
            for (size_t k = 0; k < g.legal.size(); ++k) {
                Board next = g.board;
                makeMove(next, g.legal[k]);
                packBoard(next, h.childPos[g.firstChild + k]);
                h.parents[g.firstChild + k] = g.slot;
            }
        });
        // device: one fused update+eval batch per chunk; child slots cycle inside this step's region (only the CHOSEN
        // child's slot has to survive until the half's next step; a chunk that recycles slots invalidates the earlier
        // ones and a chosen child among them is re-materialised in finish())
        const uint32_t region = h.regionBase + uint32_t((h.step & 1) * h.maxChildren);
        size_t done = 0;
        while (done < h.total) {
            const size_t n = std::min(h.total - done, h.maxChildren);
            for (size_t k = 0; k < n; ++k) h.children[done + k] = region + uint32_t(k);
            if (done > 0) std::fill(h.slotValid.begin(), h.slotValid.begin() + done, 0);
            done += n;
        }
        h.pending = std::async(std::launch::async, [&h, &gpuMutex, &gpuSeconds, ctx] {
            std::lock_guard<std::mutex> lock(gpuMutex);
            const auto g0 = std::chrono::steady_clock::now();
            int r = SPX_OK;
            for (size_t lo = 0; lo < h.total && r == SPX_OK;) {  // chunks of maxChildren, within the batch capacity
                const size_t chunkEnd = std::min(h.total, lo - lo % h.maxChildren + h.maxChildren);
                const size_t m = std::min(chunkEnd - lo, ctxMaxBatch(ctx));
                r = spx_acc_update_eval(ctx, h.parents.data() + lo, h.children.data() + lo, h.childPos.data() + lo, m,
                                        h.evals.data() + lo);
                lo += m;
            }
            gpuSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - g0).count();
            return r;
        });
        h.inFlight = true;
        stats->evals += h.total;
        return SPX_OK;
    };

    // ---- phase 2: wait for the evals, pick moves, adjudicate (datagen.cpp:224-252), advance ----
    auto finish = [&](Half& h) -> int {
        if (!h.inFlight) return SPX_OK;
        h.inFlight = false;
        const int r = h.pending.get();
        if (r != SPX_OK) return r;
        forGames(h, [&](Game& g, uint32_t i) {
            if (!g.active || g.legal.empty()) return;
            const int32_t* evals = h.evals.data() + g.firstChild;
            // the leaves as a search sees them: network output clamped like eval::adjustStatic (eval.cpp:24-27)
            int best = INT32_MIN;
            for (size_t k = 0; k < g.legal.size(); ++k) best = std::max(best, clampStaticEval(-evals[k]));
            // exploration: uniformly among the moves within temperature_cp of the best (0 = greedy, first best)
            size_t pick = 0, seen = 0;
            for (size_t k = 0; k < g.legal.size(); ++k) {
                if (clampStaticEval(-evals[k]) >= best - p->temperature_cp) {
                    ++seen;
                    if (g.rng.below(uint32_t(seen)) == 0) pick = k;
                    if (p->temperature_cp == 0) break;
                }
            }
            const int score = clampStaticEval(-evals[pick]);
            // white-point-of-view score (what the reference records) and its WDL-normalised form (what its adjudication
            // counters compare): search.cpp:237-238, datagen.cpp:224-252,283-284
            const int whiteScore = g.board.stm ? score : -score;
            int material = 0;
            for (int sq = 0; sq < 64; ++sq) {
                static const int kValue[7] = {1, 3, 3, 5, 9, 0, 0};
                if (g.board.mailbox[sq] != kNoPiece) material += kValue[g.board.mailbox[sq] >> 1];
            }
            const int normScore = wdlNormalize(whiteScore, material);
            if (g.plies == 0) {  // opening verification (datagen.cpp:176-190): this search doubles as the verification search
                const int normBest = wdlNormalize(g.board.stm ? best : -best, material);
                if (std::abs(normBest) > kVerificationScoreLimit) {
                    g.discard = true;
                    return;
                }
            }
            AdjCounters adj{g.winPlies, g.lossPlies, g.drawPlies};
            uint32_t outcome = adjudicate(adj, normScore, g.startPly + g.plies);
            g.winPlies = adj.win, g.lossPlies = adj.loss, g.drawPlies = adj.draw;

            g.history.push_back(boardHash(g.board));
            makeMove(g.board, g.legal[pick]);
            ++g.plies;
            // Position::isDrawn of the new position (position.cpp:621-667; datagen.cpp:264-268): it overrides an
            // adjudicated result and the move is recorded with score 0
            bool drawn = g.plies >= p->max_plies;  // (this driver's own ply cap)
            if (g.board.halfmove >= 100) {  // 50-move rule, and nothing else is looked at: a draw unless checkmate
                std::vector<Move> replies;
                generateLegal(g.board, replies);
                drawn = drawn || !(g.board.inCheck() && replies.empty());
            } else if (!drawn) {
                const uint64_t hash = boardHash(g.board);
                drawn = std::count(g.history.begin(), g.history.end(), hash) >= 2;
                if (!drawn) {
                    spx_packed_pos rec;
                    packBoard(g.board, rec);
                    uint64_t lo, hi;
                    std::memcpy(&lo, rec.pieces, 8);
                    std::memcpy(&hi, rec.pieces + 8, 8);
                    drawn = insufficientMaterial(rec.occupancy, lo, hi);
                }
            }
            if (drawn) outcome = 1;
            g.moves.push_back(viriMove(g.legal[pick]));
            g.scores.push_back(int16_t(drawn || std::abs(whiteScore) <= 2 ? 0 : whiteScore));
            const size_t idx = g.firstChild + pick;
            if (h.slotValid[idx]) {
                g.slot = h.children[idx];
            } else {  // its scratch slot was recycled by a later chunk of this step: materialise it again
                g.needsFix = true;
                g.fixPos = h.childPos[idx];
                g.slot = i;  // the game's home slot
            }
            g.outcome = uint8_t(outcome);
        });
        refreshSlots.clear();
        refreshPos.clear();
        for (uint32_t i = h.begin; i < h.end; ++i) {
            Game& g = games[i];
            if (!g.active) continue;
            if (g.discard) {  // the verification filter dropped this opening: not counted, nothing written
                g.active = g.discard = false;
                --started;
            } else if (g.outcome != 255) {
                finishGame(g, g.outcome);
            } else if (g.needsFix) {  // through the record alone (full refresh of the home slot)
                refreshSlots.push_back(i);
                refreshPos.push_back(g.fixPos);
            }
            g.needsFix = false;
        }
        ++h.step;
        stats->steps += 1;
        return refreshSlots.empty() ? SPX_OK : refresh(refreshPos, refreshSlots);
    };

    // alternate the halves: prepare(h) runs on the host while the other half's batch is on the GPU
    for (;;) {
        bool progress = false;
        for (Half& h : halves) {
            if ((rc = finish(h)) != SPX_OK) break;
            if ((rc = prepare(h)) != SPX_OK) break;
            progress = progress || h.inFlight;
        }
        if (rc != SPX_OK) break;
        if (!progress) {  // nothing on the GPU: done unless games remain to be started (all of a pass ended in mates)
            bool idleSlot = false;
            for (const Game& g : games) idleSlot = idleSlot || !g.active;
            if (started >= p->target_games || !idleSlot) break;
        }
    }
    for (Half& h : halves) {  // error exit: do not leave a batch in flight
        if (h.inFlight) {
            h.pending.wait();
            h.inFlight = false;
        }
    }
    if (out) std::fclose(out);
    stats->steps = (stats->steps + nHalves - 1) / nHalves;  // plies played per game slot
    stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stats->gpu_seconds = gpuSeconds;
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------------
// Device-movegen path (default): the games live on the GPU. Per ply, for all games at once:
//   spx_movegen_kernel  legal moves + child records of every game's current position (parents = the games' slots)
//   fused update+eval   one accumulator update and evaluation per child (spx_acc_update_eval_device)
//   spx_pick_kernel     the depth-1 policy; the chosen child's record / slot become the game's current ones
// and 24 bytes per game come back (move, score, key, clocks) for the adjudication counters, repetition detection and
// the viriformat records, which stay on the host together with the random openings of newly started games.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

struct DeviceBuffers {
    std::vector<void*> ptrs;
    ~DeviceBuffers() {
        for (void* q : ptrs) (void)hipFree(q);
    }
    template <typename T>
    T* get(size_t count) {
        void* q = nullptr;
        if (hipMalloc(&q, std::max<size_t>(count * sizeof(T), 16)) != hipSuccess) return nullptr;
        ptrs.push_back(q);
        return static_cast<T*>(q);
    }
};

struct PinnedBuffers {
    std::vector<void*> ptrs;
    ~PinnedBuffers() {
        for (void* q : ptrs) spx_host_free(q);
    }
    template <typename T>
    T* get(size_t count) {
        void* q = spx_host_alloc(std::max<size_t>(count * sizeof(T), 16));
        if (q) ptrs.push_back(q);
        return static_cast<T*>(q);
    }
    template <typename T>
    T* getMapped(size_t count, T** deviceView) {  // page-locked AND mapped into the device: kernels write it directly
        void* q = nullptr;
        if (hipHostMalloc(&q, std::max<size_t>(count * sizeof(T), 16), hipHostMallocMapped) != hipSuccess) return nullptr;
        ptrs.push_back(q);
        if (hipHostGetDevicePointer(reinterpret_cast<void**>(deviceView), q, 0) != hipSuccess) return nullptr;
        return static_cast<T*>(q);
    }
};

#define SPX_SP_HIP(call)                                                                      \
    do {                                                                                      \
        const hipError_t err_ = (call);                                                       \
        if (err_ != hipSuccess) {                                                             \
            setError(std::string("spx_selfplay_run: ") + #call + ": " + hipGetErrorString(err_)); \
            return SPX_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

// Random openings generated in bulk on the device (datagen.cpp:146-171: start position or a double-Chess960 start, then
// 8 random plies plus one more on a coin flip): the host only places the start pieces; the plies are spx_movegen_kernel +
// spx_pick_kernel without evaluations (every legal move equally likely), a last generation drops positions that have
// no legal move. With this the host chess core is out of the device self-play loop altogether - what a rank needs
// when eight of them share a node's CPUs.
class OpeningPool {
public:
    OpeningPool(spx_ctx* ctx, hipStream_t stream, bool dfrc, uint32_t basePlies)
        : ctx_(ctx), stream_(stream), dfrc_(dfrc), basePlies_(basePlies) {}

    size_t available() const {
        return records_.size() - next_;
    }
    void pop(spx_packed_pos& rec, uint64_t& gameSeed) {
        rec = records_[next_];
        gameSeed = seeds_[next_];
        ++next_;
    }
    // appends at least `want` openings (rounds of up to kRound candidates; a few percent die on the way)
    int refill(size_t want, Rng& rng) {
        records_.erase(records_.begin(), records_.begin() + long(next_));
        seeds_.erase(seeds_.begin(), seeds_.begin() + long(next_));
        next_ = 0;
        const size_t target = records_.size() + want;
        while (records_.size() < target) {
            const size_t missing = target - records_.size();
            const int rc = round(std::min(kRound, missing + missing / 8 + 64), rng);
            if (rc != SPX_OK) return rc;
        }
        return SPX_OK;
    }

private:
    static constexpr size_t kRound = 16384, kPerSeat = 96;

    int round(size_t b, Rng& rng) {
        if (!dPositions_) {
            dPositions_ = dev_.get<uint64_t>(kRound * 4);
            dRng_ = dev_.get<uint64_t>(kRound);
            dExtra_ = dev_.get<uint8_t>(kRound);
            dChildren_ = dev_.get<uint64_t>(kRound * kPerSeat * 4);
            dMoves_ = dev_.get<uint16_t>(kRound * kPerSeat);
            dParents_ = dev_.get<uint32_t>(kRound * kPerSeat);
            dFirst_ = dev_.get<uint32_t>(kRound);
            dCount_ = dev_.get<uint32_t>(kRound);
            dInCheck_ = dev_.get<uint8_t>(kRound);
            dTotal_ = dev_.get<uint32_t>(1);
            if (!dPositions_ || !dRng_ || !dExtra_ || !dChildren_ || !dMoves_ || !dParents_ || !dFirst_ || !dCount_ ||
                !dInCheck_ || !dTotal_) {
                setError("spx_selfplay_run: out of device memory (opening pool)");
                return SPX_ERR_HIP;
            }
        }
        std::vector<spx_packed_pos> start(b);
        std::vector<uint64_t> pickSeeds(b), gameSeeds(b);
        std::vector<uint8_t> extra(b);
        spx_packed_pos standard;
        packBoard(startpos(), standard);
        for (size_t i = 0; i < b; ++i) {
            if (dfrc_) {
                const uint32_t w = rng.below(960), k = rng.below(960);
                packBoard(dfrcStart(w, k), start[i]);
            } else {
                start[i] = standard;
            }
            extra[i] = uint8_t(rng.next() >> 63);  // 8 + coin flip (datagen.cpp:153)
            pickSeeds[i] = rng.next();
            gameSeeds[i] = rng.next();
        }
        SPX_SP_HIP(hipMemcpyAsync(dPositions_, start.data(), b * 32, hipMemcpyHostToDevice, stream_));
        SPX_SP_HIP(hipMemcpyAsync(dRng_, pickSeeds.data(), b * 8, hipMemcpyHostToDevice, stream_));
        SPX_SP_HIP(hipMemcpyAsync(dExtra_, extra.data(), b, hipMemcpyHostToDevice, stream_));
        for (uint32_t ply = 0; ply <= basePlies_ + 1; ++ply) {
            const int rc = spx_movegen_device(ctx_, dPositions_, b, nullptr, dChildren_, dMoves_, dParents_, dFirst_, dCount_,
                                              dInCheck_, kRound * kPerSeat, dTotal_, stream_);
            if (rc != SPX_OK) return rc;
            if (ply == basePlies_ + 1) break;  // the last generation only tells which positions still have a move
            PickParams pk{};
            pk.nGames = uint32_t(b);
            pk.first = dFirst_;
            pk.count = dCount_;
            pk.enable = ply == basePlies_ ? dExtra_ : nullptr;
            pk.children = dChildren_;
            pk.positions = dPositions_;
            pk.rng = dRng_;
            SPX_SP_HIP(launchPick(pk, stream_));
        }
        std::vector<spx_packed_pos> done(b);
        std::vector<uint32_t> counts(b);
        uint32_t total = 0;
        SPX_SP_HIP(hipMemcpyAsync(done.data(), dPositions_, b * 32, hipMemcpyDeviceToHost, stream_));
        SPX_SP_HIP(hipMemcpyAsync(counts.data(), dCount_, b * 4, hipMemcpyDeviceToHost, stream_));
        SPX_SP_HIP(hipMemcpyAsync(&total, dTotal_, 4, hipMemcpyDeviceToHost, stream_));
        SPX_SP_HIP(hipStreamSynchronize(stream_));
        if (total > kRound * kPerSeat) {
            setError("spx_selfplay_run: opening generation overflowed its child buffer");
            return SPX_ERR_CAPACITY;
        }
        for (size_t i = 0; i < b; ++i) {
            if (counts[i] != 0) {
                records_.push_back(done[i]);
                seeds_.push_back(gameSeeds[i]);
            }
        }
        return SPX_OK;
    }

    spx_ctx* ctx_;
    hipStream_t stream_;
    bool dfrc_;
    uint32_t basePlies_;
    DeviceBuffers dev_;
    uint64_t* dPositions_ = nullptr;
    uint64_t* dRng_ = nullptr;
    uint8_t* dExtra_ = nullptr;
    uint64_t* dChildren_ = nullptr;
    uint16_t* dMoves_ = nullptr;
    uint32_t* dParents_ = nullptr;
    uint32_t* dFirst_ = nullptr;
    uint32_t* dCount_ = nullptr;
    uint8_t* dInCheck_ = nullptr;
    uint32_t* dTotal_ = nullptr;
    std::vector<spx_packed_pos> records_;
    std::vector<uint64_t> seeds_;
    size_t next_ = 0;
};

}  // namespace

// Seeded random playouts ON THE DEVICE (the bulk form of spx_random_positions: datagen-style random plies from the start
// position or a double-Chess960 start): spx_movegen_kernel + spx_pick_kernel without evaluations, position i playing
// min_ply + (draw mod range) plies; a position that runs out of moves keeps its last (mated / stalemated) position. The
// host only places the start pieces - a rank that shares its node's CPUs with seven others generates its batch here.
extern "C" int spx_random_positions_gpu(spx_ctx* ctx, uint64_t seed, size_t count, int min_ply, int max_ply, int dfrc_every,
                                        void* d_out) {
    if (!ctx || (count && !d_out) || min_ply < 0 || max_ply < min_ply || max_ply > 1000) {
        setError("spx_random_positions_gpu: invalid argument");
        return SPX_ERR_INVALID_ARG;
    }
    if (count == 0) return SPX_OK;
    constexpr size_t kRound = 16384, kPerSeat = 96;
    if (hipSetDevice(ctxDevice(ctx)) != hipSuccess) {
        setError("spx_random_positions_gpu: hipSetDevice failed");
        return SPX_ERR_HIP;
    }
    const size_t round = std::min(kRound, ctxMaxBatch(ctx));
    DeviceBuffers dev;
    uint64_t* dRng = dev.get<uint64_t>(round);
    uint8_t* dEnable = dev.get<uint8_t>(round);
    uint64_t* dChildren = dev.get<uint64_t>(round * kPerSeat * 4);
    uint16_t* dMoves = dev.get<uint16_t>(round * kPerSeat);
    uint32_t* dParents = dev.get<uint32_t>(round * kPerSeat);
    uint32_t* dFirst = dev.get<uint32_t>(round);
    uint32_t* dCount = dev.get<uint32_t>(round);
    uint8_t* dInCheck = dev.get<uint8_t>(round);
    uint32_t* dTotal = dev.get<uint32_t>(1);
    if (!dRng || !dEnable || !dChildren || !dMoves || !dParents || !dFirst || !dCount || !dInCheck || !dTotal) {
        setError("spx_random_positions_gpu: out of device memory");
        return SPX_ERR_HIP;
    }
    hipStream_t stream = static_cast<hipStream_t>(ctxStream(ctx));  // the context's own stream (what NULL means to spx_movegen_device)
    Rng rng{seed ^ 0x5DEECE66Dull};
    spx_packed_pos standard;
    packBoard(startpos(), standard);
    std::vector<spx_packed_pos> start(round);
    std::vector<uint64_t> seeds(round);
    std::vector<uint32_t> plies(round);
    std::vector<uint8_t> enable(round);
    for (size_t lo = 0; lo < count; lo += round) {
        const size_t b = std::min(round, count - lo);
        uint64_t* dPositions = static_cast<uint64_t*>(d_out) + lo * 4;
        uint32_t longest = 0;
        for (size_t i = 0; i < b; ++i) {
            const size_t game = lo + i;
            if (dfrc_every > 0 && game % size_t(dfrc_every) == size_t(dfrc_every) - 1) {
                const uint32_t w = rng.below(960), k = rng.below(960);
                packBoard(dfrcStart(w, k), start[i]);
            } else {
                start[i] = standard;
            }
            plies[i] = uint32_t(min_ply) + rng.below(uint32_t(max_ply - min_ply + 1));
            longest = std::max(longest, plies[i]);
            seeds[i] = rng.next();
        }
        SPX_SP_HIP(hipMemcpyAsync(dPositions, start.data(), b * 32, hipMemcpyHostToDevice, stream));
        SPX_SP_HIP(hipMemcpyAsync(dRng, seeds.data(), b * 8, hipMemcpyHostToDevice, stream));
        for (uint32_t ply = 0; ply < longest; ++ply) {
            for (size_t i = 0; i < b; ++i) enable[i] = ply < plies[i];
            SPX_SP_HIP(hipMemcpyAsync(dEnable, enable.data(), b, hipMemcpyHostToDevice, stream));
            const int rc = spx_movegen_device(ctx, dPositions, b, nullptr, dChildren, dMoves, dParents, dFirst, dCount, dInCheck,
                                              round * kPerSeat, dTotal, nullptr);
            if (rc != SPX_OK) return rc;
            PickParams pk{};
            pk.nGames = uint32_t(b);
            pk.first = dFirst;
            pk.count = dCount;
            pk.enable = dEnable;
            pk.children = dChildren;
            pk.positions = dPositions;
            pk.rng = dRng;
            SPX_SP_HIP(launchPick(pk, stream));
            SPX_SP_HIP(hipStreamSynchronize(stream));  // the host rewrites `enable` for the next ply
        }
        uint32_t total = 0;
        SPX_SP_HIP(hipMemcpyAsync(&total, dTotal, 4, hipMemcpyDeviceToHost, stream));
        SPX_SP_HIP(hipStreamSynchronize(stream));
        if (total > round * kPerSeat) {
            setError("spx_random_positions_gpu: a generation overflowed its child buffer");
            return SPX_ERR_CAPACITY;
        }
    }
    return SPX_OK;
}

namespace {

// One half of the seats: its own child buffers, update list and status mirror. The two halves run on the context's two
// lanes and are both kept in flight: the small kernels of one half's ply overlap the other half's update kernel.
struct HalfStatus {              // what the host reads back per ply (page-locked); 64-bit words, as spx_game_status_kernel writes them
    SelfplayCounters counters;   // run-wide counters as of the end of this half's step kernel
    unsigned long long streamWords;  // words this half has written to ITS ring so far
    unsigned long long total;    // children generated this ply (must fit `cap`)
};

constexpr uint32_t kPliesInFlight = 2;  // direct launches, per half: the host enqueues this far ahead of the results it has seen
constexpr uint32_t kGraphPliesMax = 16; // graph mode: consecutive plies of a half per captured graph (even: see enqueue) ...
constexpr uint32_t kGraphsInFlight = 2; // ... and graph launches per half in flight (each with its own status slots)
constexpr uint32_t kStatusSlotsMax = kGraphPliesMax * kGraphsInFlight;
constexpr int kRetryUngraphed = -1000;  // internal: the graph capture was refused before anything ran

struct DeviceHalf {
    uint32_t begin = 0, end = 0;     // seats
    size_t cap = 0;                  // children per ply
    uint64_t* dChildren = nullptr;
    uint16_t* dMoves = nullptr;
    uint32_t* dParents = nullptr;
    int32_t* dEvals = nullptr;
    uint32_t* dTotal = nullptr;
    uint32_t *dUpdParents = nullptr, *dUpdChildren = nullptr;
    uint64_t* dUpdPositions = nullptr;
    HalfStatus* hStatus = nullptr;   // page-locked and mapped into the device, [kStatusSlots] ...
    HalfStatus* dStatus = nullptr;   // ... and its device view
    hipEvent_t done[kStatusSlotsMax] = {};
    uint64_t enqueued = 0, acked = 0;  // plies enqueued / plies whose results the host has read
    hipGraphExec_t graph[kGraphsInFlight] = {};  // graph mode: kGraphPlies consecutive plies of this half each, captured once
    uint32_t index = 0;              // which lane of the context this half runs on
    // this half's output ring (page-locked host memory the step kernel writes through its device mapping) and position
    uint32_t *hRing = nullptr, *dRing = nullptr;
    uint32_t ringWords = 0;
    unsigned long long* dStreamWords = nullptr;
    uint64_t streamWords = 0, consumedWords = 0;  // newest snapshot seen / words already written to the file
};

// The games live on the device: per ply and half the host enqueues one fixed chain of launches and, kPliesInFlight plies
// later, reads 100 bytes of counters plus whatever finished games the step kernels wrote into the output ring - O(1) host
// work per ply, whatever the number of seats, and never on the GPU's critical path (round 2 kept counters, repetition keys
// and the records on the host: 24 bytes and a few hundred instructions per game and ply, one host round trip per ply, and
// the GPU idled half of the wall time at 4 096 games).
int runDeviceGames(spx_ctx* ctx, const spx_selfplay_params* p, const char* out_path, spx_selfplay_stats* stats) {
    const uint32_t G = p->n_games;
    const uint32_t nHalves = G >= 2 ? 2 : 1;
    const uint32_t maxPlies = std::max(8u, std::min(p->max_plies ? p->max_plies : 300u, 4096u));
    if (uint64_t(G) * 2 + 1 > 0x7FFFFFFFull) {
        setError("spx_selfplay_run: too many games for 32-bit slot ids");
        return SPX_ERR_INVALID_ARG;
    }
    // Live fixed-node search (SPX_SELFPLAY_SEARCH_NODES(k) in flags; SearchStepParams in spx_kernels.h has the rules): the
    // per-ply chain below becomes a per-ROUND chain - every seat expands one node of its own search tree per round - with
    // spx_search_step_kernel in place of spx_game_step_kernel and one more accumulator slot per seat and tree level.
    const uint32_t searchNodes = p->flags >> 8;
    const bool search = searchNodes != 0;
    // two slots per seat (current / next position) + the null slot (+ the search levels below the root)
    int rc = spx_acc_reserve(ctx, size_t(G) * 2 + 1 + (search ? size_t(G) * (kSearchLevels - 1) : 0));
    if (rc != SPX_OK) return rc;
    SPX_SP_HIP(hipSetDevice(ctxDevice(ctx)));
    SPX_SP_HIP(hipMemset(ctxSlotRecords(ctx) + size_t(G) * 2 * 32, 0, 32));  // the null slot holds the empty board
    FILE* out = nullptr;
    if (out_path && out_path[0]) {
        out = std::fopen(out_path, "wb");
        if (!out) {
            setError(std::string("spx_selfplay_run: cannot open ") + out_path);
            return SPX_ERR_INVALID_ARG;
        }
    }
    struct FileCloser {
        FILE* f;
        ~FileCloser() {
            if (f) std::fclose(f);
        }
    } closer{out};
    std::memset(stats, 0, sizeof(*stats));

    const size_t perSeat = 96;  // children per seat and ply (mean ~35; a ply that needs more is reported as an error)
    // Plies per captured graph. Between two graph launches on a stream this runtime leaves ~80-100 us (rocprofv3 kernel trace,
    // profiles/r03_selfplay_lane_chain.txt: no gap between the kernels of a graph), 8-11 % of a lane's time at two plies per
    // graph; more plies per graph halve that but idle longer at the end of a run and want a deeper opening pool. Measured
    // (profiles/r03_ab_selfplay_plies_per_graph.txt; 2 / 4 / 8 / 16 plies): 4 096 seats x 65 536 games 2.42 / 2.46 / 2.46 / 2.41
    // x 10^8, 1 024 x 8 192 1.45 / 1.49 / 1.46 / 1.43, but 4 096 x 8 192 1.85 / 1.81 / 1.71 / 1.53 and 16 384 x 65 536 2.58 /
    // 2.56 / 2.44 / 2.25: four for runs of at least eight games per seat, else two. option selfplay_graph_plies overrides (even, 2..16).
    uint32_t kGraphPlies = p->target_games >= 8ull * G ? 4 : 2;
    {
        const int64_t v = ctxSelfplayOption(ctx, 1);
        if (v >= 2 && v <= int64_t(kGraphPliesMax) && v % 2 == 0) kGraphPlies = uint32_t(v);
    }
    const uint32_t kStatusSlots = kGraphPlies * kGraphsInFlight;  // (>= kPliesInFlight)
    const uint32_t poolCap = (kStatusSlots + 3) * G + 32768;
    // (a seat writes 9 + plies words per finished game; with very short ply caps up to ~10 words per ply in flight)
    const uint32_t ringWords = uint32_t(std::max<uint64_t>(
        1u << 20, uint64_t(G) * std::max<uint64_t>(2 * (maxPlies + 9), 10 * (kStatusSlots + 2))));
    DeviceBuffers dev;
    PinnedBuffers pinned;
    auto* dPositions = dev.get<uint64_t>(size_t(G) * 4);
    auto* dSlots = dev.get<uint32_t>(G);
    auto* dRng = dev.get<uint64_t>(G);
    auto* dFirst = dev.get<uint32_t>(G);
    auto* dCount = dev.get<uint32_t>(G);
    auto* dInCheck = dev.get<uint8_t>(G);
    auto* dState = dev.get<SeatState>(G);
    auto* dInitial = dev.get<uint64_t>(size_t(G) * 4);
    auto* dGameMoves = dev.get<uint32_t>(size_t(G) * maxPlies);
    auto* dKeys = dev.get<uint64_t>(size_t(G) * maxPlies);
    auto* dPoolRecords = dev.get<uint64_t>(size_t(poolCap) * 4);
    auto* dPoolSeeds = dev.get<uint64_t>(poolCap);
    auto* dCounters = dev.get<SelfplayCounters>(1);
    auto* hPoolRecords = pinned.get<spx_packed_pos>(16384);
    auto* hPoolSeeds = pinned.get<uint64_t>(16384);
    auto* hPoolSize = pinned.get<uint32_t>(1);
    // search mode: the seats' stacks (57 KiB of child records per seat and level: 1.8 GiB at 4 096 seats - HBM is not the
    // scarce resource here), the nodes to expand next and their slots
    const size_t frameSlots = search ? size_t(G) * kSearchLevels : 0;
    auto* dSeats = dev.get<SearchSeat>(search ? G : 0);
    auto* dFrames = dev.get<SearchFrame>(frameSlots);
    auto* dFrameRecords = dev.get<uint64_t>(frameSlots * kSearchChildren * 4);
    auto* dFrameValues = dev.get<int32_t>(frameSlots * kSearchChildren);
    auto* dFrameWords = dev.get<uint16_t>(frameSlots * kSearchChildren);
    auto* dPending = dev.get<uint64_t>(search ? size_t(G) * 4 : 0);
    auto* dPendingSlots = dev.get<uint32_t>(search ? G : 0);
    auto* dExpansions = dev.get<unsigned long long>(search ? G : 0);
    std::vector<DeviceHalf> halves(nHalves);
    struct RingCloser {  // the halves' output rings
        std::vector<DeviceHalf>& h;
        ~RingCloser() {
            for (DeviceHalf& hf : h) {
                if (hf.hRing) (void)hipHostFree(hf.hRing);
            }
        }
    } ringCloser{halves};
    bool ok = dPositions && dSlots && dRng && dFirst && dCount && dInCheck && dState && dInitial && dGameMoves && dKeys &&
              dPoolRecords && dPoolSeeds && dCounters && hPoolRecords && hPoolSeeds && hPoolSize && dSeats && dFrames &&
              dFrameRecords && dFrameValues && dFrameWords && dPending && dPendingSlots && dExpansions;
    for (uint32_t h = 0; h < nHalves && ok; ++h) {
        DeviceHalf& hf = halves[h];
        hf.index = h;
        hf.begin = uint32_t(uint64_t(G) * h / nHalves);
        hf.end = uint32_t(uint64_t(G) * (h + 1) / nHalves);
        const uint32_t seats = hf.end - hf.begin;
        hf.cap = std::min(size_t(seats) * perSeat, ctxMaxBatch(ctx));
        hf.dChildren = dev.get<uint64_t>(hf.cap * 4);
        hf.dMoves = dev.get<uint16_t>(hf.cap);
        hf.dParents = dev.get<uint32_t>(hf.cap);
        hf.dEvals = dev.get<int32_t>(hf.cap);
        hf.dTotal = dev.get<uint32_t>(1);
        hf.dUpdParents = dev.get<uint32_t>(seats);
        hf.dUpdChildren = dev.get<uint32_t>(seats);
        hf.dUpdPositions = dev.get<uint64_t>(size_t(seats) * 4);
        hf.hStatus = pinned.getMapped<HalfStatus>(kStatusSlots, &hf.dStatus);
        hf.ringWords = ringWords;
        hf.dStreamWords = dev.get<unsigned long long>(1);
        ok = hf.hStatus && hf.dStatus && hf.dTotal && hf.dStreamWords && hipMemset(hf.dTotal, 0, 4) == hipSuccess &&
             hipMemset(hf.dStreamWords, 0, 8) == hipSuccess &&
             hipHostMalloc(reinterpret_cast<void**>(&hf.hRing), size_t(ringWords) * 4, hipHostMallocMapped) == hipSuccess &&
             hipHostGetDevicePointer(reinterpret_cast<void**>(&hf.dRing), hf.hRing, 0) == hipSuccess;
        ok = ok && hf.dChildren && hf.dMoves && hf.dParents && hf.dEvals && hf.dTotal && hf.dUpdParents && hf.dUpdChildren &&
             hf.dUpdPositions && hf.hStatus && seats <= ctxMaxBatch(ctx);
    }
    if (!ok) {
        setError("spx_selfplay_run: out of device or page-locked memory (or a context smaller than half the seats)");
        return SPX_ERR_HIP;
    }
    hipStream_t stream;
    SPX_SP_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    struct StreamCloser {
        hipStream_t s;
        ~StreamCloser() {
            (void)hipStreamSynchronize(s);
            (void)hipStreamDestroy(s);
        }
    } streamCloser{stream};
    struct EventCloser {
        std::vector<DeviceHalf>& hs;
        ~EventCloser() {
            for (DeviceHalf& hf : hs) {
                for (hipEvent_t e : hf.done) {
                    if (e) (void)hipEventDestroy(e);
                }
                for (hipGraphExec_t g : hf.graph) {
                    if (g) (void)hipGraphExecDestroy(g);
                }
            }
        }
    } eventCloser{halves};
    for (DeviceHalf& hf : halves) {
        for (hipEvent_t& e : hf.done) SPX_SP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    {
        std::vector<uint32_t> iota(G);
        for (uint32_t i = 0; i < G; ++i) iota[i] = i;  // every seat starts on its home slot
        SPX_SP_HIP(hipMemcpy(dSlots, iota.data(), size_t(G) * 4, hipMemcpyHostToDevice));
        if (search) SPX_SP_HIP(hipMemcpy(dPendingSlots, iota.data(), size_t(G) * 4, hipMemcpyHostToDevice));
    }
    if (search) {
        SPX_SP_HIP(hipMemset(dSeats, 0, size_t(G) * sizeof(SearchSeat)));
        SPX_SP_HIP(hipMemset(dFrames, 0, frameSlots * sizeof(SearchFrame)));
        SPX_SP_HIP(hipMemset(dPending, 0, size_t(G) * 32));
        SPX_SP_HIP(hipMemset(dExpansions, 0, size_t(G) * 8));
    }
    SPX_SP_HIP(hipMemset(dPositions, 0, size_t(G) * 32));  // empty records generate no moves
    SPX_SP_HIP(hipMemset(dState, 0, size_t(G) * sizeof(SeatState)));
    SPX_SP_HIP(hipMemset(dRng, 0, size_t(G) * 8));
    SPX_SP_HIP(hipMemset(dCounters, 0, sizeof(SelfplayCounters)));
    SPX_SP_HIP(hipDeviceSynchronize());

    Rng rng{p->seed};
    OpeningPool openings(ctx, stream, p->dfrc != 0, p->opening_plies ? p->opening_plies : 8);
    const auto t0 = std::chrono::steady_clock::now();
    double gpuWait = 0.0, enqueueSeconds = 0.0;
    SelfplayCounters latest{};       // newest counters seen
    uint32_t published = 0;          // openings handed to the device so far
    uint64_t evals = 0, steps = 0;

    // Openings: generated in bulk on the device (OpeningPool), published to the device-side ring ahead of every claim the
    // step kernels in flight can make: a step claims at most one opening per seat, and up to 2 * kStatusSlots half-steps may
    // have run since the counters were last seen.
    auto ensurePool = [&](hipStream_t s) -> int {
        const uint32_t want = (kStatusSlots + 1) * G + 64;
        while (published - latest.poolCursor < want) {
            const uint32_t room = poolCap - (published - latest.poolCursor);
            const uint32_t n = std::min<uint32_t>({16384u, room, want + G - (published - latest.poolCursor)});
            if (n == 0) break;
            if (openings.available() < n) {
                const int r = openings.refill(n, rng);
                if (r != SPX_OK) return r;
            }
            for (uint32_t k = 0; k < n; ++k) openings.pop(hPoolRecords[k], hPoolSeeds[k]);
            for (uint32_t k = 0; k < n;) {  // ring: at most two contiguous pieces
                const uint32_t at = (published + k) % poolCap, m = std::min(n - k, poolCap - at);
                SPX_SP_HIP(hipMemcpyAsync(dPoolRecords + size_t(at) * 4, hPoolRecords + k, size_t(m) * 32, hipMemcpyHostToDevice, s));
                SPX_SP_HIP(hipMemcpyAsync(dPoolSeeds + at, hPoolSeeds + k, size_t(m) * 8, hipMemcpyHostToDevice, s));
                k += m;
            }
            published += n;
            *hPoolSize = published;
            SPX_SP_HIP(hipMemcpyAsync(&dCounters->poolSize, hPoolSize, 4, hipMemcpyHostToDevice, s));
            SPX_SP_HIP(hipStreamSynchronize(s));  // the staging buffers are reused (a refill happens every few dozen plies)
        }
        return SPX_OK;
    };

    // One ply of a half, enqueued back to back on its lane stream: move generation, eval-only update + evaluation of the
    // children, the step kernel (search, bookkeeping, game records, new games), the materialising update of the seats
    // that go on, and the status kernel. Launches only (no copies, no events, no waits): the same sequence is what graph
    // mode captures.
    auto plyBody = [&](DeviceHalf& hf, hipStream_t s, uint32_t statusSlot) -> int {
        int r = SPX_OK;
        const uint32_t seats = hf.end - hf.begin;
        {   // (spx_movegen_device without its cursor memset: the status kernel of the half's previous ply zeroed it)
            MovegenParams mp{};
            mp.positions = (search ? dPending : dPositions) + size_t(hf.begin) * 4;
            mp.nPositions = seats;
            mp.parentValues = (search ? dPendingSlots : dSlots) + hf.begin;
            mp.children = hf.dChildren;
            mp.moves = hf.dMoves;
            mp.parents = hf.dParents;
            mp.first = dFirst + hf.begin;
            mp.count = dCount + hf.begin;
            mp.inCheck = dInCheck + hf.begin;
            mp.cursor = hf.dTotal;
            mp.capacity = uint32_t(hf.cap);
            SPX_SP_HIP(launchMovegen(mp, (seats + 3) / 4, s));
        }
        // eval-only children (child slots NULL): ~35 siblings per seat evaluated, none stored
        r = spx_acc_update_eval_device_counted(ctx, hf.dParents, nullptr, hf.dChildren, hf.dTotal, hf.cap, hf.dEvals, s);
        if (r != SPX_OK) return r;
        GameStepParams gp{};
        gp.nSeats = seats;
        gp.seatBase = hf.begin;
        gp.nSeatsTotal = G;
        gp.first = dFirst + hf.begin;
        gp.count = dCount + hf.begin;
        gp.inCheck = dInCheck + hf.begin;
        gp.evals = hf.dEvals;
        gp.moves = hf.dMoves;
        gp.children = hf.dChildren;
        gp.positions = dPositions + size_t(hf.begin) * 4;
        gp.slots = dSlots + hf.begin;
        gp.rng = dRng + hf.begin;
        gp.state = dState + hf.begin;
        gp.initial = dInitial + size_t(hf.begin) * 4;
        gp.gameMoves = dGameMoves + size_t(hf.begin) * maxPlies;
        gp.keys = dKeys + size_t(hf.begin) * maxPlies;
        gp.maxPlies = maxPlies;
        gp.temperature = p->temperature_cp;
        gp.targetGames = p->target_games;
        gp.poolRecords = dPoolRecords;
        gp.poolSeeds = dPoolSeeds;
        gp.poolCap = poolCap;
        gp.counters = dCounters;
        gp.ring = hf.dRing;
        gp.ringWords = hf.ringWords;
        gp.streamWords = hf.dStreamWords;
        gp.updParents = hf.dUpdParents;
        gp.updChildren = hf.dUpdChildren;
        gp.updPositions = hf.dUpdPositions;
        if (search) {
            SearchStepParams sp{};
            sp.game = gp;
            sp.nodeBudget = searchNodes;
            sp.seats = dSeats + hf.begin;
            sp.frames = dFrames + size_t(hf.begin) * kSearchLevels;
            sp.frameRecords = dFrameRecords + size_t(hf.begin) * kSearchLevels * kSearchChildren * 4;
            sp.frameValues = dFrameValues + size_t(hf.begin) * kSearchLevels * kSearchChildren;
            sp.frameWords = dFrameWords + size_t(hf.begin) * kSearchLevels * kSearchChildren;
            sp.pending = dPending + size_t(hf.begin) * 4;
            sp.pendingSlots = dPendingSlots + hf.begin;
            sp.levelSlotBase = 2 * G + 1;
            sp.expansions = dExpansions + hf.begin;
            SPX_SP_HIP(launchSearchStep(sp, s));
        } else {
            SPX_SP_HIP(launchGameStep(gp, s));
        }
        // the one accumulator per seat that has to exist next ply: the move played (parent -> the seat's other slot) or
        // the new game's opening (null slot -> rebuilt from scratch by the update kernel)
        r = spx_acc_update_device(ctx, hf.dUpdParents, hf.dUpdChildren, hf.dUpdPositions, seats, s);
        if (r != SPX_OK) return r;
        SPX_SP_HIP(launchGameStatus(dCounters, hf.dTotal, hf.dStreamWords, hf.dStatus + statusSlot, s));
        return SPX_OK;
    };
    // The per-ply chain is a fixed sequence of ~9 launches with fixed arguments: a launch-bound inner loop at small seat
    // counts (1 024 seats: the host's enqueue time exceeded the GPU's). GRAPH MODE captures kGraphPlies consecutive plies of
    // a half ONCE per status-slot group (two plies, so that the context's alternating sort / refresh buffers are back in
    // phase at the end of a graph) and relaunches the instantiated graphs in turn, kGraphsInFlight of them ahead: one host
    // call per two plies. The lanes' cross-stream event gates cannot be part of a capture, so graph mode runs the lanes
    // ungated. option selfplay_graph = 0, or a HIP runtime that refuses the capture, means direct launches (one ply at a
    // time, kPliesInFlight plies ahead).
    bool useGraph = ctxSelfplayOption(ctx, 0) != 0;
    auto enqueue = [&](DeviceHalf& hf) -> int {
        void* laneStream = nullptr;
        int lr = ctxLaneBegin(ctx, int(hf.index), &laneStream, /*gates=*/!useGraph);
        if (lr != SPX_OK) return lr;
        struct LaneGuard {
            spx_ctx* c;
            int i;
            ~LaneGuard() {
                ctxLaneEnd(c, i);
            }
        } laneGuard{ctx, int(hf.index)};
        hipStream_t s = static_cast<hipStream_t>(laneStream);
        int r = ensurePool(s);
        if (r != SPX_OK) return r;
        const uint32_t which = uint32_t(hf.enqueued / kGraphPlies) % kGraphsInFlight;  // graph mode: the graph / slot group due
        if (useGraph && !hf.graph[which]) {
            hipGraph_t captured = nullptr;
            const bool began = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
            uint32_t bodies = 0;
            bool ok = began;
            for (; ok && bodies < kGraphPlies; ++bodies) ok = plyBody(hf, s, which * kGraphPlies + bodies) == SPX_OK;
            const bool ended = !began || (hipStreamEndCapture(s, &captured) == hipSuccess && captured != nullptr);
            ok = ok && ended && hipGraphInstantiate(&hf.graph[which], captured, nullptr, nullptr, 0) == hipSuccess;
            if (captured) (void)hipGraphDestroy(captured);
            if (!ok) {  // this HIP runtime cannot capture the chain
                (void)hipGetLastError();
                hf.graph[which] = nullptr;
                // (a ply that failed HALFWAY through its launches has advanced the context's alternating refresh buffers by an
                // odd count without running anything: not a state to go on from)
                if (steps != 0 || (began && bodies != 0 && bodies != kGraphPlies)) {
                    setError("spx_selfplay_run: graph capture failed after the run had started or in the middle of a ply");
                    return SPX_ERR_HIP;
                }
                useGraph = false;  // nothing has run yet: direct launches for the whole run
                return kRetryUngraphed;
            }
        }
        if (useGraph) {
            SPX_SP_HIP(hipGraphLaunch(hf.graph[which], s));
            SPX_SP_HIP(hipEventRecord(hf.done[which], s));
            hf.enqueued += kGraphPlies;
            steps += kGraphPlies;
            return SPX_OK;
        }
        r = plyBody(hf, s, uint32_t(hf.enqueued % kPliesInFlight));
        if (r != SPX_OK) return r;
        SPX_SP_HIP(hipEventRecord(hf.done[hf.enqueued % kPliesInFlight], s));
        ++hf.enqueued;
        ++steps;
        return SPX_OK;
    };
    // wait for a half's oldest unit in flight (one ply; in graph mode the kGraphPlies plies of one graph launch); append
    // what the step kernels finished since the last look to the file
    auto awaitUnit = [&](DeviceHalf& hf) -> int {
        const uint32_t plies = useGraph ? kGraphPlies : 1;
        const uint32_t slots = useGraph ? kStatusSlots : kPliesInFlight;
        const auto g0 = std::chrono::steady_clock::now();
        SPX_SP_HIP(hipEventSynchronize(hf.done[useGraph ? (hf.acked / kGraphPlies) % kGraphsInFlight : hf.acked % kPliesInFlight]));
        gpuWait += std::chrono::duration<double>(std::chrono::steady_clock::now() - g0).count();
        for (uint32_t k = 0; k < plies; ++k) {
            const HalfStatus& st = hf.hStatus[hf.acked % slots];
            ++hf.acked;
            if (st.total > hf.cap) {
                setError("spx_selfplay_run: " + std::to_string(st.total) + " children in one ply exceed the buffer of " +
                         std::to_string(hf.cap) + " (context max_batch too small for this many seats?)");
                return SPX_ERR_CAPACITY;
            }
            evals += st.total;
            // every counter only grows: the newest view is the element-wise maximum of the halves' snapshots
            const SelfplayCounters& c = st.counters;
            hf.streamWords = std::max<uint64_t>(hf.streamWords, st.streamWords);
            latest.games = std::max(latest.games, c.games);
            latest.positions = std::max(latest.positions, c.positions);
            for (int o = 0; o < 3; ++o) latest.outcomes[o] = std::max(latest.outcomes[o], c.outcomes[o]);
            latest.discarded = std::max(latest.discarded, c.discarded);
            latest.started = std::max(latest.started, c.started);
            latest.poolCursor = std::max(latest.poolCursor, c.poolCursor);
        }
        if (latest.poolCursor > published) {
            setError("spx_selfplay_run: the opening pool ran dry (internal: the publishing margin was too small)");
            return SPX_ERR_CAPACITY;
        }
        // only THIS half's ring, up to the position its own status kernel reported behind its own step kernel (the event waited
        // for above): every word below it has been written. (Games of the two halves interleave in the file in any order.)
        if (hf.streamWords - hf.consumedWords > hf.ringWords) {
            setError("spx_selfplay_run: the output ring overflowed");
            return SPX_ERR_CAPACITY;
        }
        while (hf.consumedWords < hf.streamWords) {
            const uint32_t at = uint32_t(hf.consumedWords % hf.ringWords);
            const uint64_t m = std::min<uint64_t>(hf.streamWords - hf.consumedWords, hf.ringWords - at);
            if (out && std::fwrite(hf.hRing + at, 4, size_t(m), out) != size_t(m)) {
                setError("spx_selfplay_run: short write to the output file");
                return SPX_ERR_INVALID_ARG;
            }
            hf.consumedWords += m;
        }
        return SPX_OK;
    };
    // Every game that gets a ticket finishes (a discarded opening hands its ticket on), so the run is over when the target
    // number of games has been written. The host sees that up to kStatusSlots plies late: the extra plies run on seats that
    // have gone idle one after the other (empty records generate no moves).
    for (;;) {
        if (latest.games < p->target_games) {
            // keep every half its quota of plies ahead, the halves taking turns (A0 B0 A1 B1 ... / graph mode: A01 B01 A23 B23)
            const uint32_t unit = useGraph ? kGraphPlies : 1, ahead = useGraph ? kStatusSlots : kPliesInFlight;
            bool retry = false;
            for (uint32_t round = 0; round < ahead && rc == SPX_OK && !retry; round += unit) {
                for (DeviceHalf& hf : halves) {
                    const uint64_t inFlight = hf.enqueued - hf.acked;
                    if (rc != SPX_OK || inFlight != round) continue;
                    const auto a1 = std::chrono::steady_clock::now();
                    rc = enqueue(hf);
                    enqueueSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - a1).count();
                    if (rc == kRetryUngraphed) {  // the capture was refused before anything ran: direct launches from the top
                        rc = SPX_OK;
                        retry = true;
                        break;
                    }
                }
            }
            if (retry) continue;
        }
        if (rc != SPX_OK) break;
        DeviceHalf* oldest = nullptr;  // the half whose oldest unit in flight was enqueued first
        for (DeviceHalf& hf : halves) {
            if (hf.enqueued > hf.acked && (!oldest || hf.acked < oldest->acked)) oldest = &hf;
        }
        if (!oldest) break;
        if ((rc = awaitUnit(*oldest)) != SPX_OK) break;
    }
    (void)spx_ctx_synchronize(ctx);  // nothing may still reference the staging buffers on an error exit
    stats->games = latest.games;
    stats->positions = latest.positions;
    for (int k = 0; k < 3; ++k) stats->outcomes[k] = latest.outcomes[k];
    stats->evals = evals;
    stats->steps = (steps + nHalves - 1) / nHalves;
    if (search) {  // nodes expanded by all searches together
        std::vector<unsigned long long> expanded(G);
        SPX_SP_HIP(hipMemcpy(expanded.data(), dExpansions, size_t(G) * 8, hipMemcpyDeviceToHost));
        stats->steps = 0;
        for (unsigned long long e : expanded) stats->steps += e;
    }
    stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stats->gpu_seconds = gpuWait;
    if (ctxSelfplayOption(ctx, 2)) {
        std::fprintf(stderr, "[spx_selfplay] %.3f s: waiting for the GPU %.3f, enqueue + openings %.3f; %llu openings discarded by "
                     "the verification filter, %u published; %s\n", stats->seconds, gpuWait, enqueueSeconds,
                     static_cast<unsigned long long>(latest.discarded), published,
                     useGraph ? (kGraphPlies == 2 ? "graph mode: two plies per launch, two launches ahead"
                                                  : kGraphPlies == 4 ? "graph mode: four plies per launch, two launches ahead"
                                                                     : "graph mode: option selfplay_graph_plies plies per launch, two launches ahead")
                              : "direct launches");
    }
    return rc;
}

}  // namespace

extern "C" int spx_selfplay_run(spx_ctx* ctx, const spx_selfplay_params* p, const char* out_path,
                                spx_selfplay_stats* stats) {
    if (!ctx || !p || !stats || p->n_games == 0 || p->target_games == 0 ||
        (p->flags & 0xFFu & ~uint32_t(SPX_SELFPLAY_HOST_MOVEGEN)) ||
        ((p->flags & SPX_SELFPLAY_HOST_MOVEGEN) && (p->flags >> 8))) {  // the search lives in the device-resident driver
        setError("spx_selfplay_run: invalid argument");
        return SPX_ERR_INVALID_ARG;
    }
    return (p->flags & SPX_SELFPLAY_HOST_MOVEGEN) ? runHostMovegen(ctx, p, out_path, stats)
                                                  : runDeviceGames(ctx, p, out_path, stats);
}
