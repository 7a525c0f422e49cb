// Multi-device group: one spx_ctx per GPU inside ONE process, behind a single call - the C-ABI face of SURVEY 8(e) for a
// native host (a C++ engine embeds one library, not eight worker processes; the Python harnesses keep the
// one-process-per-GPU layout of stormphrax_amd/distributed.py). Positions are independent, so a batch is cut into
// contiguous shards (sizes differ by at most one - the same rule as distributed.shard_bounds) and every member evaluates
// its shard through the ordinary spx_eval_full on its own host thread: the H2D / kernels / D2H of the members overlap,
// and there is no collective on the data path. The weights are uploaded once per member (the "N H2D copies" variant of
// the reference-side ncclBroadcast in SURVEY 8e); an 89 MB image per 288 GB device.
#include <cstdio>
#include <algorithm>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/spx_nnue.h"
#include "../../include/spx_nnue_dev.h"
#include "spx_internal.h"

struct spx_group {
    std::vector<spx_ctx*> members;
    std::vector<int> devices;
};

namespace {

void shardBounds(size_t n, size_t rank, size_t world, size_t& lo, size_t& hi) {
    const size_t base = n / world, extra = n % world;
    lo = rank * base + std::min(rank, extra);
    hi = lo + base + (rank < extra ? 1 : 0);
}

// fn(member index, lo, hi) on one host thread per member with a non-empty shard; first failure wins
template <typename Fn>
int forEachShard(spx_group* group, size_t n, const char* who, Fn fn) {
    const size_t world = group->members.size();
    std::vector<int> status(world, SPX_OK);
    std::vector<std::string> message(world);
    std::vector<std::thread> workers;
    workers.reserve(world);
    for (size_t r = 0; r < world; ++r) {
        size_t lo, hi;
        shardBounds(n, r, world, lo, hi);
        if (lo == hi) continue;
        workers.emplace_back([&, r, lo, hi] {
            status[r] = fn(r, lo, hi);
            if (status[r] != SPX_OK) message[r] = spx_last_error();  // thread-local: carry it to the caller's thread
        });
    }
    for (auto& w : workers) w.join();
    for (size_t r = 0; r < world; ++r) {
        if (status[r] != SPX_OK) {
            spx::setError(std::string(who) + ": member " + std::to_string(r) + " (device " +
                          std::to_string(group->devices[r]) + "): " + message[r]);
            return status[r];
        }
    }
    return SPX_OK;
}

}  // namespace

extern "C" {

int spx_group_create(const spx_net* net, const int* devices, size_t n_devices, size_t max_batch_per_device,
                     uint32_t flags, spx_group** out) {
    if (!net || !out || (n_devices && !devices)) {
        spx::setError("spx_group_create: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    std::vector<int> ids(devices, devices + n_devices);
    if (ids.empty()) {  // every visible device
        int visible = 0;
        if (spx_device_count(&visible) != SPX_OK || visible <= 0) {
            spx::setError("spx_group_create: no HIP device visible (the library has no CPU path)");
            return SPX_ERR_NO_DEVICE;
        }
        for (int d = 0; d < visible; ++d) ids.push_back(d);
    }
    auto group = std::make_unique<spx_group>();
    for (int d : ids) {
        spx_ctx* ctx = nullptr;
        const int rc = spx_ctx_create_ex(net, d, max_batch_per_device, flags, &ctx);
        if (rc != SPX_OK) {
            const std::string why = spx_last_error();
            for (spx_ctx* m : group->members) spx_ctx_destroy(m);
            spx::setError("spx_group_create: device " + std::to_string(d) + ": " + why);
            return rc;
        }
        group->members.push_back(ctx);
        group->devices.push_back(d);
    }
    *out = group.release();
    return SPX_OK;
}

void spx_group_destroy(spx_group* group) {
    if (!group) return;
    for (spx_ctx* m : group->members) spx_ctx_destroy(m);
    delete group;
}

size_t spx_group_size(const spx_group* group) { return group ? group->members.size() : 0; }

spx_ctx* spx_group_member(spx_group* group, size_t index) {
    return (group && index < group->members.size()) ? group->members[index] : nullptr;
}

int spx_group_shard(const spx_group* group, size_t n, size_t index, size_t* lo, size_t* hi) {
    if (!group || !lo || !hi || index >= group->members.size()) {
        spx::setError("spx_group_shard: invalid argument");
        return SPX_ERR_INVALID_ARG;
    }
    shardBounds(n, index, group->members.size(), *lo, *hi);
    return SPX_OK;
}

int spx_group_eval_full(spx_group* group, const spx_packed_pos* positions, size_t n, int32_t* out) {
    if (!group || (n && (!positions || !out))) {
        spx::setError("spx_group_eval_full: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    if (n == 0) return SPX_OK;
    return forEachShard(group, n, "spx_group_eval_full", [&](size_t r, size_t lo, size_t hi) {
        return spx_eval_full(group->members[r], positions + lo, hi - lo, out + lo);
    });
}

int spx_group_adjust(spx_group* group, const spx_packed_pos* positions, size_t n, const spx_adjust_params* params,
                     const int32_t* corrections, int32_t* evals) {
    if (!group || !params || (n && (!positions || !evals))) {
        spx::setError("spx_group_adjust: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    if (n == 0) return SPX_OK;
    return forEachShard(group, n, "spx_group_adjust", [&](size_t r, size_t lo, size_t hi) {
        return spx_adjust(group->members[r], positions + lo, hi - lo, params, corrections ? corrections + lo : nullptr,
                          evals + lo);
    });
}

// BASELINE config 4 from a native host: the concurrent games are dealt to the members (contiguous shares of n_games and of
// target_games, sizes differing by at most one; member r's RNG stream is seed + r, its games go to <out_path>.<r>.vf),
// every member runs the ordinary spx_selfplay_run on its own host thread against its own device - games never cross
// devices, there is no exchange step - and the statistics are summed (seconds / gpu_seconds: the slowest member's).
int spx_group_selfplay_run(spx_group* group, const spx_selfplay_params* params, const char* out_path, spx_selfplay_stats* stats) {
    if (!group || !params || !stats || params->n_games == 0 || params->target_games == 0) {
        spx::setError("spx_group_selfplay_run: invalid argument");
        return SPX_ERR_INVALID_ARG;
    }
    const size_t world = group->members.size();
    // (fewer seats than members: the first n_games members get one seat each and share the whole target)
    const size_t active = std::min<size_t>(world, params->n_games);
    std::vector<spx_selfplay_stats> part(world);
    const int rc = forEachShard(group, params->n_games, "spx_group_selfplay_run", [&](size_t r, size_t lo, size_t hi) {
        spx_selfplay_params mine = *params;
        mine.n_games = uint32_t(hi - lo);
        // the target itself is shared out: with fewer games wanted than members have seats, the members whose share
        // is empty sit the run out instead of playing one game each beyond the target
        size_t tLo, tHi;
        shardBounds(params->target_games, r, active, tLo, tHi);
        const std::string path = (out_path && out_path[0]) ? std::string(out_path) + "." + std::to_string(r) + ".vf" : std::string();
        if (tHi == tLo) {
            part[r] = spx_selfplay_stats{};
            if (!path.empty()) {  // (one file per member, whatever its share: an empty one here)
                if (std::FILE* f = std::fopen(path.c_str(), "wb")) std::fclose(f);
            }
            return int(SPX_OK);
        }
        mine.target_games = uint32_t(tHi - tLo);
        mine.seed = params->seed + r;
        return spx_selfplay_run(group->members[r], &mine, path.empty() ? nullptr : path.c_str(), &part[r]);
    });
    if (rc != SPX_OK) return rc;
    *stats = spx_selfplay_stats{};
    for (const spx_selfplay_stats& s : part) {
        stats->games += s.games;
        stats->positions += s.positions;
        stats->evals += s.evals;
        stats->steps = std::max(stats->steps, s.steps);
        for (int k = 0; k < 3; ++k) stats->outcomes[k] += s.outcomes[k];
        stats->seconds = std::max(stats->seconds, s.seconds);
        stats->gpu_seconds = std::max(stats->gpu_seconds, s.gpu_seconds);
    }
    return SPX_OK;
}

}  // extern "C"
