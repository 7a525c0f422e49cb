// C-ABI implementation of libspx_nnue (include/spx_nnue.h): network validation, device context, launches, and the
// host helpers. There is deliberately NO CPU evaluation path here: without a HIP device spx_ctx_create fails with
// SPX_ERR_NO_DEVICE (the CPU oracle lives in oracle/ and is test infrastructure only).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/spx_nnue.h"
#include "../../include/spx_nnue_dev.h"
#include "spx_chess.h"
#include "spx_device_math.h"
#include "spx_internal.h"
#include "spx_kernels.h"
#include "spx_ftx.h"

namespace spx {

namespace {
thread_local std::string t_lastError;
}

void setError(const std::string& msg) {
    t_lastError = msg;
}

int buildThreatLut(uint32_t* lut);  // spx_luts.cpp
void buildDeltaTables(uint64_t* tab);  // spx_luts.cpp

}  // namespace spx

using namespace spx;

struct spx_net {
    std::vector<unsigned char> blob;  // full file image, logical layout
    std::string name;
    const int16_t* psqW() const { return reinterpret_cast<const int16_t*>(blob.data() + kOffPsqW); }
    const int8_t* threatW() const { return reinterpret_cast<const int8_t*>(blob.data() + kOffThreatW); }
    const int16_t* ftBias() const { return reinterpret_cast<const int16_t*>(blob.data() + kOffFtBias); }
    const int8_t* l1W() const { return reinterpret_cast<const int8_t*>(blob.data() + kOffL1W); }
};

constexpr size_t kProfEventsPerCall = 5;

// scratch of the column-sliced full refresh (spx_ftx.hip): one set per context and per lane, allocated on first use
struct FtxScratch {
    uint32_t *lists = nullptr, *heads = nullptr, *ranks = nullptr, *hist = nullptr, *binStart = nullptr,
             *sorted = nullptr, *plan = nullptr, *groupHead = nullptr, *stages = nullptr, *outHist = nullptr;
    size_t capacity = 0;  // positions per pass
    void release() {
        for (uint32_t* q : {lists, heads, ranks, hist, binStart, sorted, plan, groupHead, stages, outHist}) {
            if (q) (void)hipFree(q);
        }
        *this = FtxScratch{};
    }
};

struct spx_ctx {
    int device = 0;
    size_t maxBatch = 0;       // scratch capacity: positions one launch sequence can hold intermediates for
    size_t callLimit = 0;      // spx_eval_full_device[_async] accept up to this many positions per call (the max_batch the
                               // context was created with) and walk them in chunks of maxBatch
    hipStream_t stream = nullptr;
    // weights
    int16_t* dPsqW = nullptr;
    uint8_t* dThrW = nullptr;
    uint8_t* dRowS = nullptr;   // the column-sliced row table of spx_ftx.hip (built on first use)
    FtxScratch ftx;             // its scratch (the lanes hold their own)
    // the gather's hot set: the threat / pawn-pair rows it keeps in LDS beside the piece-square slab (spx_ftx.h). Chosen from DATA - a
    // histogram over the first batch that takes the pipeline (calibrateHotRows) or over the batch handed to spx_ctx_calibrate -;
    // results never depend on it. Per context: the ranks of a multi-GPU job calibrate independently.
    uint32_t* dHotHash = nullptr;    // [kFtxHotHashWords] row -> LDS slot as a hash (FtxParams::hotHash)
    uint32_t hotHashMul = 1;
    uint8_t* dHotS = nullptr;        // [8][hotRows][128 B]
    uint32_t* dHotIds = nullptr;     // [kFtxHotRowsMax]
    uint32_t* dHotCounts = nullptr;  // [kThreatRows + 16] the calibration's histogram (+ statistics)
    uint8_t* dHiMask = nullptr;      // [kPsqRows] slices in which a piece-square row's high-byte plane is not all zero (spx_ftx.h)
    std::vector<uint32_t> hotIds;    // host copy of the set, in slot order
    uint32_t hotRowsWanted = kFtxHotRowsDefault;  // option ftx_hot_rows
    uint32_t hotRows = 0, coldShift = 1;
    bool hotCalibrated = false;      // the set was chosen (or given: spx_ctx_set_hot_rows); false: the next big batch chooses it
    bool hotAutoCalibrate = true;    // option ftx_auto_calibrate: 0 = only spx_ctx_calibrate / spx_ctx_set_hot_rows choose the set
    bool ftxEnabled = true;     // big full refreshes take the column-sliced pipeline (spx_ftx.hip); option ftx = 0 / SPX_CTX_ONE_KERNEL_FT: never
    size_t ftxMin = kFtxMinPositions;  // option ftx_min: smallest batch that takes the sliced pipeline
    int ftxFailAfter = -1, ftxScratchSets = 0;  // (option ftx_fail_after: simulated allocation failure)
    int ftxFailLaunch = -1;  // (option ftx_fail_launch: the k-th sliced pass from now reports a launch failure; -1: never)
    bool ftxMinForced = false;    // (ftx_min given: the same threshold for stream-ordered and pipelined calls)
    bool ftxUnavailable = false;  // its table or scratch did not fit the device memory
    // (spx_eval_full_device_async: each lane has its own scratch set - swapLane -, so one lane's preparation runs beside the
    // other lane's gather. A ring of extra streams that prepared up to three batches ahead cost 12 %: 1.60 against 1.81e8
    // evals/s - more streams than hardware queues serialise; profiles/r04_sliced_pipeline_overlap_attempts.txt)
    int16_t* dFtBias = nullptr;
    int8_t* dL1W = nullptr;
    int32_t *dL1B = nullptr, *dL2W = nullptr, *dL2B = nullptr, *dL3W = nullptr, *dL3B = nullptr;
    uint32_t* dLut = nullptr;
    uint64_t* dDeltaTab = nullptr;  // ray / knight masks + pseudo-attack sets of the threat-delta derivation
    uint32_t* dOutlierTab = nullptr;  // remainders of the near-compact piece-square rows (nullptr: the net has none)
    uint32_t nearPsqRows = 0;
    uint32_t nearBits[kLutCompactWords] = {};  // host copy of the near-compact bitmap (spx_ctx_count_rows)
    // scratch
    void* dPositions = nullptr;  // staging for the host-buffer entry point
    int32_t* dOut = nullptr;
    uint8_t* dFtOut = nullptr;
    uint8_t* dKingKeys = nullptr;  // counting-sort scratch
    uint8_t* dOutKeys = nullptr;
    uint32_t* dHist = nullptr;     // 3 sort-histogram buffers of kHistWords + 64 words of counters
    uint32_t* dPerspOrder = nullptr;  // perspective ids grouped by king bucket
    uint32_t* dPosOrder = nullptr;    // position ids grouped by output bucket
    uint32_t* dRefreshList = nullptr; // update kernel: perspectives deferred to the rebuild pass (its own buffer: the king sort
                                      // of a full refresh on another stream must not overwrite a list that is being consumed)
    // accumulator arena (incremental path): nSlots x (4 KiB accumulators + 32 B record)
    uint8_t* dArena = nullptr;
    uint8_t* dSlotRecords = nullptr;
    size_t nSlots = 0;
    uint32_t *dSlotsA = nullptr, *dSlotsB = nullptr;  // staging for the host-buffer entry points [max_batch]
    uint8_t* dStaged = nullptr;                        // [max_batch][32] records of the slots being evaluated
    uint8_t* dDeltas = nullptr;                        // [max_batch] spx_move_delta staging (allocated on first use)
    int histCur = 0;               // dHist: buffers [0],[1] alternate between large sorts (each sort clears the other one
                                   // for its successor), [2] belongs to the single-launch small sort; behind them words
                                   // 0,1: alternating counters of the update kernel's deferred-refresh list
    int refreshCur = 0;            // which of the two counters the next update uses (its refresh pass clears the other)
    uint32_t* histUsed = nullptr;  // the buffer the latest sort wrote (what the MLP kernel reads)
    // spx_eval_full_device_async: two scratch sets ("lanes") with their own streams alternate, so that the sorts and
    // the MLP of one batch run beside the feature-transformer kernel of the next; the FT kernels themselves are
    // chained by events (they never overlap each other - two of them thrash the caches)
    struct EvalLane {
        uint8_t *dFtOut = nullptr, *dKingKeys = nullptr, *dOutKeys = nullptr, *dStaged = nullptr;
        uint32_t *dHist = nullptr, *dPerspOrder = nullptr, *dPosOrder = nullptr, *dRefreshList = nullptr, *histUsed = nullptr;
        int histCur = 0, refreshCur = 0;
        hipStream_t stream = nullptr;
        hipEvent_t ftDone = nullptr, done = nullptr;
        hipEvent_t pace[kProfEventsPerCall] = {};  // see spx_ctx::paceEvents
        bool ftRecorded = false;
        // staging of the chunked host-buffer call (allocated on its first use): device in/out + page-locked mirrors
        void *dIn = nullptr, *hIn = nullptr;
        int32_t *dOutStage = nullptr, *hOut = nullptr;
        FtxScratch ftx;
    } lanes[3];  // (the third one serves spx_eval_full_device_async alone: option eval_lanes)
    unsigned evalLanes = 3;
    bool lanesReady = false;
    bool lanesUnavailable = false;   // the lanes did not fit into the device memory: async calls run stream-ordered
    hipEvent_t fallbackDone = nullptr;
    unsigned laneNext = 0;
    hipEvent_t ftGateWait = nullptr, ftGateRecord = nullptr;  // set around a lane's call: FT waits / signals
    size_t tinyBatchMax = 0;       // spx_eval_full*: batches up to this size skip the sorts (one MLP tile per position)
    void* hTinyIo = nullptr;       // page-locked, device-mapped staging of the tiny-batch host call: records, then scores
    size_t mlpShareMax = 0;        // spx_mlp_kernel: positions up to which four waves share one 16-position tile
    size_t streamAccMin = 0;       // spx_update_kernel: records from which the arena is accessed non-temporally
    size_t updateChainMax = 1024;  // option update_chain_max: fused update batches up to this size take spx_update_chain_kernel on
                                   // unit paths (one launch, rebuilds inline); larger ones spx_update_kernel + the rebuild pass
    int64_t selfplayOptions[3] = {1, 0, 0};  // options selfplay_graph (0: direct launches), selfplay_graph_plies (0: automatic), selfplay_trace
    int replayPaths = -1;          // option replay_paths: spx_acc_replay_tree by heavy paths (1) / by levels (0) / its own choice (-1)
    uint32_t replaySegment = 8;    // option replay_segment: plies per path segment of the replay
    size_t updateSplitMaxV2 = 0;   // second-generation kernel: records up to which the perspectives get separate waves
    size_t refreshWaves = 0;       // option refresh_waves: waves of the rebuild pass (0 = automatic)
    size_t teamMaxPersp = 512;     // option ft_team_max: full refreshes of at most this many perspectives run one workgroup per
                                   // perspective (spx_ft_team_kernel): evaluate_once of 64 / 256 positions 30.8 -> 27.5 / 33.6 -> 28.8 us,
                                   // 1 024 positions 43.3 -> 44.6 (profiles/r03_ab_rebuild_pass_team_kernel.txt); 0 = never
    uint32_t compactPsqRows = 0;   // piece-square rows with an i8 copy in the u8 row table (compact_rows = 0: none)
    uint32_t compactBits[kLutCompactWords] = {};  // host copy of the LUT's compact-row bitmap (spx_ctx_count_rows)
    bool kingSortEnabled = true;   // option king_sort = 0 walks perspectives in input order (A/B of the L2-locality sort)
    bool smallL2Weights = false;   // every |l2W| < 2^23: the MLP tail may use 24-bit multiplies
    // Pipelined calls record an event at the five points of a call where spx_profile_* would (round 6): measured on the sustained
    // loop of tools/probes/sustained_rate.py, the three lanes settle into 2.12e8 evals/s with those records in their streams and into
    // 1.89e8 without - the records hold a lane's next preparation back until what the lane ran before has fully drained. Nobody reads
    // them (option pace_events = 0: none, unless a profile is open; pace_mask / pace_head_extra: which).
    hipEvent_t* paceEvents = nullptr;
    bool paceEnabled = true;
    // which of the five, and extra records at the head of the call (A/B on the sustained loop, x 1e8 evals/s: none 1.89 - all five 2.14 -
    // only the two at the head 2.16 - those + one more 2.18 - + two / four more 2.17 / 2.15; the one behind the gather alone 2.10, with
    // the head's 2.11; the one before the gather alone: nothing): three records in front of the extraction
    uint32_t paceMask = 3, paceHeadExtra = 1;
    bool ftxFoldSort = true;       // option ftx_fold_sort: one-pass batches of the pipeline sort the MLP's order themselves
    uint32_t computeUnits = 0;
    uint32_t ftGridCap = 0;
    uint32_t updateGridCap = 0;  // the update kernels' own cap (heavier workgroups: fewer, longer-lived ones win)
    // optional per-kernel timing (spx_profile_*): event triples recorded around the two kernels of each call
    double profLastPrepareMs = 0.0;
    std::vector<hipEvent_t> profEvents;  // kProfEventsPerCall per recorded call: [0] start, [1] after the sorts, [4] before
                                         // the FT kernel (after a pipelined call's wait), [2] after it, [3] after the MLP
    size_t profUsed = 0;
};

namespace spx {
int ctxDevice(const spx_ctx* ctx) {
    return ctx->device;
}

void* ctxStream(const spx_ctx* ctx) {
    return ctx->stream;
}

size_t ctxMaxBatch(const spx_ctx* ctx) {
    return ctx->maxBatch;
}

uint8_t* ctxSlotRecords(const spx_ctx* ctx) {
    return ctx->dSlotRecords;
}

int64_t ctxSelfplayOption(const spx_ctx* ctx, int which) {
    return ctx->selfplayOptions[which];
}
}  // namespace spx

namespace {

#define SPX_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            setError(std::string(#call) + ": " + hipGetErrorString(e_));                           \
            return SPX_ERR_HIP;                                                                    \
        }                                                                                          \
    } while (0)

// Header checks in the order of the reference's validate() (nnue.cpp:85-185); messages name the same conditions.
int validateHeader(const unsigned char* h, std::string& name) {
    if (std::memcmp(h, "CBNF", 4) != 0) {
        setError("invalid magic bytes in network header");
        return SPX_ERR_BAD_NET;
    }
    uint16_t version, flags, hidden;
    std::memcpy(&version, h + 4, 2);
    std::memcpy(&flags, h + 6, 2);
    std::memcpy(&hidden, h + 11, 2);
    const uint8_t arch = h[9], activation = h[10], inputBuckets = h[13], outputBuckets = h[14], nameLen = h[15];
    if (version != 1) {
        setError("unsupported network format version " + std::to_string(version) + " (expected: 1)");
        return SPX_ERR_BAD_NET;
    }
    if (arch != kArchId) {
        setError("wrong network architecture " + std::to_string(arch) +
                 " (expected: 5, perspective_multilayer_dual_act_skip_l2)");
        return SPX_ERR_BAD_NET;
    }
    if (!(flags & kFlagMirrored)) {
        setError("unmirrored network, expected horizontally mirrored");
        return SPX_ERR_BAD_NET;
    }
    if (!(flags & kFlagMergedKings)) {
        setError("network does not have merged king planes, expected merged");
        return SPX_ERR_BAD_NET;
    }
    if (!(flags & kFlagPairwise)) {
        setError("network L1 does not require pairwise multiplication, expected paired");
        return SPX_ERR_BAD_NET;
    }
    if (activation != kActivationId) {
        setError("wrong l1 activation function " + std::to_string(activation) + " (expected: crelu)");
        return SPX_ERR_BAD_NET;
    }
    if (hidden != kL1) {
        setError("wrong number of l1 neurons " + std::to_string(hidden) + " (expected: 1024)");
        return SPX_ERR_BAD_NET;
    }
    if (!(inputBuckets & 0x80)) {
        setError("network does not have the expected threat inputs");
        return SPX_ERR_BAD_NET;
    }
    if ((inputBuckets & 0x7F) != kInputBuckets) {
        setError("wrong number of input buckets " + std::to_string(inputBuckets) + " (expected: 16)");
        return SPX_ERR_BAD_NET;
    }
    if (outputBuckets != kOutputBuckets) {
        setError("wrong number of output buckets " + std::to_string(outputBuckets) + " (expected: 8)");
        return SPX_ERR_BAD_NET;
    }
    name.assign(reinterpret_cast<const char*>(h + 16), nameLen < 48 ? nameLen : 48);
    return SPX_OK;
}

// zstd-compressed nets (header flag 0x0001, nnue.cpp:213-247): the 64-byte header stays plain, the payload is one zstd
// frame of the logical (unpermuted) arrays. The reference inflates it with its vendored decoder; here the system's
// libzstd is loaded on first need (no link-time dependency, no headers: three functions with a stable C ABI).
struct ZstdApi {
    size_t (*decompress)(void*, size_t, const void*, size_t) = nullptr;
    unsigned (*isError)(size_t) = nullptr;
    const char* (*getErrorName)(size_t) = nullptr;
};

const ZstdApi* zstdApi() {
    static const ZstdApi api = [] {
        ZstdApi z;
        for (const char* lib : {"libzstd.so.1", "libzstd.so"}) {
            if (void* h = dlopen(lib, RTLD_NOW | RTLD_LOCAL)) {
                z.decompress = reinterpret_cast<decltype(z.decompress)>(dlsym(h, "ZSTD_decompress"));
                z.isError = reinterpret_cast<decltype(z.isError)>(dlsym(h, "ZSTD_isError"));
                z.getErrorName = reinterpret_cast<decltype(z.getErrorName)>(dlsym(h, "ZSTD_getErrorName"));
                if (z.decompress && z.isError && z.getErrorName) break;
                z = ZstdApi{};
            }
        }
        return z;
    }();
    return api.decompress ? &api : nullptr;
}

// Threat table relayout: +128 bias (so widening is a zero-extend) and per-lane column interleave: the 16 bytes lane
// l loads at offset 16*l are columns 8l..8l+7 followed by 512+8l..512+8l+7 (pairwise partners share a lane).
void relayoutThreatRow(const int8_t* src, uint8_t* dst) {
    for (uint32_t l = 0; l < 64; ++l) {
        for (uint32_t j = 0; j < 8; ++j) {
            // within a dword: columns (c, c + 2, c + 1, c + 3) - bytes 0, 2 are one accumulator word, bytes 1, 3 the next
            const uint32_t k = (j & 4) | ((j & 1) << 1) | ((j & 2) >> 1);
            dst[16 * l + k] = uint8_t(src[8 * l + j]) ^ 0x80u;
            dst[16 * l + 8 + k] = uint8_t(src[512 + 8 * l + j]) ^ 0x80u;
        }
    }
}

// L1 weights for v_mfma_i32_16x16x64_i8 B fragments: [bucket][kstep][ntile][lane][16]; lane (g = lane>>4,
// col = lane&15) holds k = kstep*64 + g*16 + i, output o = ntile*16 + col. Source layout l1W[b][k/4][o][k%4]
// (multilayer.h:182-196).
void relayoutL1(const int8_t* src, int8_t* dst) {
    for (uint32_t b = 0; b < kOutputBuckets; ++b)
        for (uint32_t ks = 0; ks < 16; ++ks)
            for (uint32_t n = 0; n < 2; ++n)
                for (uint32_t lane = 0; lane < 64; ++lane)
                    for (uint32_t i = 0; i < 16; ++i) {
                        const uint32_t k = ks * 64 + (lane >> 4) * 16 + i;
                        const uint32_t o = n * 16 + (lane & 15);
                        dst[(((size_t(b) * 16 + ks) * 2 + n) * 64 + lane) * 16 + i] =
                            src[size_t(b) * kL1 * kL2 + size_t(k / 4) * (kL2 * 4) + o * 4 + (k % 4)];
                    }
}

// L2 weights as the MLP tail reads them (spx_mlp_kernel): [bucket][i / 4][o][i % 4] - lane o fetches the weights of four inputs
// with ONE 16-byte load, a coalesced 1 KiB per wave (the net file: [bucket][i][o], multilayer.h:261-343)
void relayoutL2(const int32_t* src, int32_t* dst) {
    for (uint32_t b = 0; b < kOutputBuckets; ++b)
        for (uint32_t i = 0; i < kL2Full; ++i)
            for (uint32_t o = 0; o < kL3; ++o)
                dst[((size_t(b) * (kL2Full / 4) + i / 4) * kL3 + o) * 4 + (i % 4)] = src[(size_t(b) * kL2Full + i) * kL3 + o];
}

FtTables tablesOf(const spx_ctx* ctx) {
    FtTables t{};
    t.psqW = ctx->dPsqW;
    t.thrW = ctx->dThrW;
    t.ftBias = ctx->dFtBias;
    t.lut = ctx->dLut;
    t.deltaTab = ctx->dDeltaTab;
    t.outlierTab = ctx->dOutlierTab;
    return t;
}

template <typename T>
int uploadArray(T*& dst, const void* src, size_t bytes, hipStream_t) {
    SPX_HIP(hipMalloc(reinterpret_cast<void**>(&dst), bytes));
    SPX_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return SPX_OK;
}

}  // namespace

extern "C" {

const char* spx_last_error(void) {
    return t_lastError.c_str();
}

int spx_net_load(const void* blob, size_t nbytes, spx_net** out) {
    if (!blob || !out) {
        setError("spx_net_load: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    *out = nullptr;
    if (nbytes < kHeaderBytes) {
        setError("Missing default network?");  // nnue.cpp:201-204
        return SPX_ERR_BAD_NET;
    }
    std::string name;
    const int rc = validateHeader(static_cast<const unsigned char*>(blob), name);
    if (rc != SPX_OK) {
        return rc;
    }
    auto net = std::make_unique<spx_net>();
    const auto* bytes = static_cast<const unsigned char*>(blob);
    uint16_t flags;
    std::memcpy(&flags, bytes + 6, 2);
    if (flags & kFlagZstd) {
        const ZstdApi* z = zstdApi();
        if (!z) {
            setError("zstd-compressed network, and libzstd.so.1 could not be loaded: decompress to the raw CBNF image first");
            return SPX_ERR_BAD_NET;
        }
        net->blob.resize(kNetFileBytes);
        std::memcpy(net->blob.data(), bytes, kHeaderBytes);
        net->blob[6] = static_cast<unsigned char>(flags & ~kFlagZstd);  // the image kept in memory is the plain one
        const size_t got = z->decompress(net->blob.data() + kHeaderBytes, kNetFileBytes - kHeaderBytes,
                                         bytes + kHeaderBytes, nbytes - kHeaderBytes);
        if (z->isError(got)) {
            setError(std::string("Failed to decompress default network: ") + z->getErrorName(got));  // nnue.cpp:236-239
            return SPX_ERR_BAD_NET;
        }
        if (got < kNetFileBytes - kHeaderBytes) {
            setError("Decompressed default network too small? " + std::to_string(got) + " < " +
                     std::to_string(kNetFileBytes - kHeaderBytes));  // nnue.cpp:241-244
            return SPX_ERR_BAD_NET;
        }
    } else {
        if (nbytes < kNetFileBytes) {
            setError("Default network too small? " + std::to_string(nbytes - kHeaderBytes) + " < " +
                     std::to_string(kNetFileBytes - kHeaderBytes));  // nnue.cpp:252-255
            return SPX_ERR_BAD_NET;
        }
        net->blob.assign(bytes, bytes + kNetFileBytes);
    }
    net->name = name;
    *out = net.release();
    return SPX_OK;
}

void spx_net_free(spx_net* net) {
    delete net;
}

const char* spx_net_name(const spx_net* net) {
    return net ? net->name.c_str() : "";
}

uint64_t spx_net_digest(const spx_net* net) {
    return net ? fnv1a64(net->blob.data() + kHeaderBytes, net->blob.size() - kHeaderBytes) : 0;
}

int spx_net_psq_row_classes(const spx_net* net, uint32_t* fit_i8, uint32_t* near_compact, uint32_t* wide) {
    if (!net || !fit_i8 || !near_compact || !wide) {
        setError("spx_net_psq_row_classes: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    // the classification spx_ctx_create applies (without its environment switches): how many of the 11 264 piece-square rows
    // a context will serve as 1 KiB copies (all weights fit i8), as 1 KiB copies + remainders (<= kOutlierCap weights do
    // not), and as 2 KiB i16 rows
    const int16_t* psq = reinterpret_cast<const int16_t*>(net->blob.data() + kOffPsqW);
    uint32_t counts[3] = {0, 0, 0};
    for (uint32_t r = 0; r < kPsqRows; ++r) {
        uint32_t outside = 0;
        for (uint32_t j = 0; j < kL1; ++j) outside += psq[size_t(r) * kL1 + j] < -128 || psq[size_t(r) * kL1 + j] > 127;
        counts[outside == 0 ? 0 : (outside <= uint32_t(kOutlierCap) ? 1 : 2)] += 1;
    }
    *fit_i8 = counts[0];
    *near_compact = counts[1];
    *wide = counts[2];
    return SPX_OK;
}

size_t spx_synth_net_bytes(void) {
    return synthNetBytes();
}

int spx_synth_net(uint64_t seed, int preset, void* buf, size_t nbytes) {
    if (!synthNet(seed, preset, buf, nbytes)) {
        setError("spx_synth_net: bad preset or buffer too small");
        return SPX_ERR_INVALID_ARG;
    }
    return SPX_OK;
}

uint64_t spx_fnv1a64(const void* data, size_t nbytes) {
    return fnv1a64(data, nbytes);
}

constexpr size_t kTinyIoRecords = 8192;
constexpr size_t kTinyIoBytesPerRecord = sizeof(spx_packed_pos) + 3 * sizeof(uint32_t) + 4;  // record, score, two slot ids  // capacity of the zero-copy staging buffer (>= any sensible tiny_batch_max)

int spx_device_count(int* count) {
    if (!count) {
        setError("spx_device_count: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        *count = 0;
        setError("no HIP device visible (libspx_nnue has no CPU fallback)");
        return SPX_ERR_NO_DEVICE;
    }
    *count = n;
    return SPX_OK;
}

// ---- options: the tuning knobs of a context (what the reference keeps in tunable.h / its UCI options) ----
// SPX_OPTIONS="name=value,name=value" applies to every context the process creates (the ONE environment variable this file
// reads); spx_ctx_set_option changes a knob of one context between calls. Unknown names and malformed values are refused.
static int parseOptions(const char* text0, const char* source, std::vector<std::pair<std::string, long long>>& out) {
    if (!text0) return SPX_OK;
    std::string text(text0);
    size_t at = 0;
    while (at < text.size()) {
        size_t end = text.find(',', at);
        if (end == std::string::npos) end = text.size();
        const std::string item = text.substr(at, end - at);
        at = end + 1;
        if (item.empty()) continue;
        const size_t eq = item.find('=');
        char* tail = nullptr;
        const long long v = eq == std::string::npos ? 0 : std::strtoll(item.c_str() + eq + 1, &tail, 10);
        if (eq == std::string::npos || eq == 0 || eq + 1 == item.size() || (tail && *tail)) {
            setError(std::string(source) + ": malformed item '" + item + "' (expected name=integer)");
            return SPX_ERR_INVALID_ARG;
        }
        out.emplace_back(item.substr(0, eq), v);
    }
    return SPX_OK;
}

// Fault-injection hooks (options ftx_fail_after / ftx_fail_launch) exist for the tests of the fall-back paths. They are honoured only
// after spx_debug_enable_test_hooks(1) - an entry point of include/spx_nnue_dev.h, which libspx_nnue.so does not export:
// in the product library the two names are unknown options.
static std::atomic<bool> gTestHooks{false};
int spx_debug_enable_test_hooks(int on) {
    gTestHooks.store(on != 0);
    return SPX_OK;
}

int spx_ctx_set_option(spx_ctx* ctx, const char* name, int64_t value) {
    if (!ctx || !name) {
        setError("spx_ctx_set_option: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    const std::string key(name);
    auto nonNegative = [&](size_t& field) {
        if (value < 0) {
            setError("option " + key + " must not be negative");
            return int(SPX_ERR_INVALID_ARG);
        }
        field = size_t(value);
        return int(SPX_OK);
    };
    if (key == "ftx") {  // big full refreshes through the column-sliced pipeline (1) or spx_ft_kernel (0)
        ctx->ftxEnabled = value != 0;
        return SPX_OK;
    }
    if (key == "ftx_min") {  // smallest batch that takes the pipeline (the same threshold for stream-ordered and pipelined calls)
        if (value < 8) {
            setError("option ftx_min must be at least 8");
            return SPX_ERR_INVALID_ARG;
        }
        ctx->ftxMin = size_t(value);
        ctx->ftxMinForced = true;
        return SPX_OK;
    }
    if (key == "ftx_hot_rows") {  // rows of the gather's hot set; takes effect with the next calibration (the next big batch)
        if (value < 0 || value > int64_t(kFtxHotRowsMax)) {
            setError("option ftx_hot_rows must be in [0, " + std::to_string(kFtxHotRowsMax) + "]");
            return SPX_ERR_INVALID_ARG;
        }
        ctx->hotRowsWanted = uint32_t(value);
        ctx->hotCalibrated = false;
        return SPX_OK;
    }
    if (key == "ftx_auto_calibrate") {  // 0: the library never chooses the hot set by itself (no blocking first call: ADVICE r5)
        ctx->hotAutoCalibrate = value != 0;
        return SPX_OK;
    }
    if (key == "eval_lanes") {  // scratch sets spx_eval_full_device_async rotates its batches over (the preparation of up to N - 1 batches beside a gather)
        if (value < 2 || value > 3) {
            setError("option eval_lanes must be 2 or 3");
            return SPX_ERR_INVALID_ARG;
        }
        const int rc = spx_ctx_synchronize(ctx);
        if (rc != SPX_OK) return rc;
        ctx->evalLanes = unsigned(value);
        ctx->laneNext = 0;
        return SPX_OK;
    }
    if ((key == "ftx_fail_after" || key == "ftx_fail_launch") && !gTestHooks.load()) {
        setError("unknown option '" + key + "'");
        return SPX_ERR_INVALID_ARG;
    }
    if (key == "ftx_fail_after") {  // test hook: the k-th scratch set of the pipeline "does not fit" (-1: never)
        ctx->ftxFailAfter = int(value);
        return SPX_OK;
    }
    if (key == "ftx_fail_launch") {  // test hook: the k-th pass of the pipeline from now on "fails to launch" (-1: never)
        ctx->ftxFailLaunch = int(value);
        return SPX_OK;
    }
    if (key == "pace_head_extra") {
        ctx->paceHeadExtra = uint32_t(value);
        return SPX_OK;
    }
    if (key == "pace_mask") {
        ctx->paceMask = uint32_t(value);
        return SPX_OK;
    }
    if (key == "pace_events") {  // pipelined calls: an event record at the five points of a call where a profile would put one (1) or none (0)
        ctx->paceEnabled = value != 0;
        return SPX_OK;
    }
    if (key == "ftx_fold_sort") {  // one-pass batches of the column-sliced pipeline: the MLP's output-bucket order from the pipeline's own sort (1) or spx_sort_* (0)
        ctx->ftxFoldSort = value != 0;
        return SPX_OK;
    }
    if (key == "king_sort") {  // one-kernel path: walk the perspectives in king-bucket order (1) or as they come (0)
        ctx->kingSortEnabled = value != 0;
        return SPX_OK;
    }
    if (key == "tiny_batch_max") return nonNegative(ctx->tinyBatchMax);
    if (key == "mlp_share_max") return nonNegative(ctx->mlpShareMax);
    if (key == "stream_acc_min") return nonNegative(ctx->streamAccMin);
    if (key == "update_split_max") return nonNegative(ctx->updateSplitMaxV2);
    if (key == "update_chain_max") return nonNegative(ctx->updateChainMax);
    if (key == "refresh_waves") return nonNegative(ctx->refreshWaves);
    if (key == "ft_team_max") return nonNegative(ctx->teamMaxPersp);
    if (key == "ft_blocks_per_cu" || key == "update_blocks_per_cu") {
        if (value < 1 || value > 4096) {
            setError("option " + key + " must be in [1, 4096]");
            return SPX_ERR_INVALID_ARG;
        }
        (key[0] == 'f' ? ctx->ftGridCap : ctx->updateGridCap) = ctx->computeUnits * uint32_t(value);
        return SPX_OK;
    }
    if (key == "selfplay_graph" || key == "selfplay_graph_plies" || key == "selfplay_trace") {
        ctx->selfplayOptions[key == "selfplay_graph" ? 0 : key == "selfplay_graph_plies" ? 1 : 2] = value;
        return SPX_OK;
    }
    if (key == "replay_paths") {
        ctx->replayPaths = value < 0 ? -1 : (value != 0);
        return SPX_OK;
    }
    if (key == "replay_segment") {
        if (value < 1) {
            setError("option replay_segment must be positive");
            return SPX_ERR_INVALID_ARG;
        }
        ctx->replaySegment = uint32_t(std::min<int64_t>(value, 1 << 20));
        return SPX_OK;
    }
    if (key == "scratch_cap" || key == "compact_rows" || key == "near_rows") {
        setError("option " + key + " shapes what a context allocates: give it to spx_ctx_create_opts (or through SPX_OPTIONS) when the context is created");
        return SPX_ERR_INVALID_ARG;
    }
    setError("unknown option '" + key + "'");
    return SPX_ERR_INVALID_ARG;
}

int spx_ctx_create(const spx_net* net, int device, size_t max_batch, spx_ctx** out) {
    return spx_ctx_create_ex(net, device, max_batch, 0u, out);
}

namespace {
struct CtxDeleter {  // a context that fails half-way through its creation releases what it already holds
    void operator()(spx_ctx* c) const { spx_ctx_destroy(c); }
};
}  // namespace

int spx_ctx_create_ex(const spx_net* net, int device, size_t max_batch, uint32_t flags, spx_ctx** out) {
    return spx_ctx_create_opts(net, device, max_batch, flags, nullptr, out);
}

int spx_ctx_create_opts(const spx_net* net, int device, size_t max_batch, uint32_t flags, const char* options, spx_ctx** out) {
    if (!net || !out || max_batch == 0 || max_batch > (1ull << 40) || (flags & ~uint32_t(SPX_CTX_WIDE_PSQ_ROWS | SPX_CTX_SLICED_FT | SPX_CTX_ONE_KERNEL_FT))) {
        setError("spx_ctx_create: invalid argument");
        return SPX_ERR_INVALID_ARG;
    }
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) {
        setError("no HIP device " + std::to_string(device) + " visible (libspx_nnue has no CPU fallback)");
        return SPX_ERR_NO_DEVICE;
    }
    SPX_HIP(hipSetDevice(device));
    std::unique_ptr<spx_ctx, CtxDeleter> ctx(new spx_ctx());
    ctx->device = device;
    // Intermediates (1 KiB of activations + sort scratch per position) are kept for at most scratch_cap positions
    // (default 4 Mi): a context created for an HBM-filling batch (BASELINE config 5: 36 bytes per resident position -
    // record in, score out) walks it in chunks of that size instead of reserving ~1.1 KB of scratch per position
    // SPX_OPTIONS (every context of the process) first, then the caller's own string: a later item overrides an earlier one
    std::vector<std::pair<std::string, long long>> envOptions;
    int rc = parseOptions(std::getenv("SPX_OPTIONS"), "SPX_OPTIONS", envOptions);
    if (rc != SPX_OK) return rc;
    if ((rc = parseOptions(options, "spx_ctx_create_opts", envOptions)) != SPX_OK) return rc;
    size_t scratchCap = size_t(1) << 22;
    bool optCompactRows = true, optNearRows = true;
    for (const auto& [name, v] : envOptions) {  // the options that shape what a context allocates: at creation only
        if (name == "scratch_cap") {
            // launch parameters are 32-bit (2 * n perspective ids, n * 1024 activation offsets are 64-bit): positions per chunk
            // stay at or below 2^30; a non-positive value is refused rather than turned into a huge size_t
            if (v <= 0 || v > (1ll << 30)) {
                setError("option scratch_cap must be in [1, 2^30]");
                return SPX_ERR_INVALID_ARG;
            }
            scratchCap = std::max<size_t>(1024, size_t(v));
        } else if (name == "compact_rows") {
            optCompactRows = v != 0;
        } else if (name == "near_rows") {
            optNearRows = v != 0;
        }
    }
    ctx->callLimit = max_batch;
    ctx->maxBatch = std::min(max_batch, scratchCap);
    max_batch = ctx->maxBatch;
    SPX_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));

    const unsigned char* b = net->blob.data();
    uint32_t compactBits[kLutCompactWords] = {}, nearBits[kLutCompactWords] = {};
    if ((rc = uploadArray(ctx->dPsqW, b + kOffPsqW, kPsqWBytes, ctx->stream)) != SPX_OK) return rc;
    {
        // u8 row table: the threat rows, then one slot per piece-square row holding its i8 copy when every weight of
        // that row fits i8 ("compact" rows: the kernels read 1 KiB instead of the 2 KiB i16 row; bit-identical sums)
        std::vector<uint8_t> thr(kThreatWBytes + size_t(kPsqRows) * kL1, 0x80);
        for (uint32_t r = 0; r < kThreatRows; ++r) {
            relayoutThreatRow(net->threatW() + size_t(r) * kL1, thr.data() + size_t(r) * kL1);
        }
        const bool useCompact = !(flags & SPX_CTX_WIDE_PSQ_ROWS) && optCompactRows;
        const int16_t* psq = reinterpret_cast<const int16_t*>(b + kOffPsqW);
        // near-compact rows: all but <= kOutlierCap weights fit i8 -> the u8 copy holds them clamped, the remainders go to
        // a side table the full-refresh kernel adds in (option near_rows = 0: such rows stay wide)
        const bool useNear = useCompact && optNearRows;
        std::vector<uint32_t> outliers;
        for (uint32_t r = 0; r < kPsqRows && useCompact; ++r) {
            const int16_t* row = psq + size_t(r) * kL1;
            uint32_t wide = 0;
            for (uint32_t j = 0; j < kL1; ++j) wide += row[j] < -128 || row[j] > 127;
            if (wide > (useNear ? uint32_t(kOutlierCap) : 0u)) continue;
            int8_t narrow[kL1];
            uint32_t k = 0;
            for (uint32_t j = 0; j < kL1; ++j) {
                const int clamped = std::max(-128, std::min(127, int(row[j])));
                narrow[j] = int8_t(clamped);
                if (clamped != row[j]) {
                    if (outliers.empty()) outliers.assign(size_t(kPsqRows) * kOutlierCap, 0xFFFFFFFFu);
                    outliers[size_t(r) * kOutlierCap + k++] = j | (uint32_t(uint16_t(int(row[j]) - clamped)) << 16);
                }
            }
            relayoutThreatRow(narrow, thr.data() + (size_t(kThreatRows) + r) * kL1);
            if (wide) {
                nearBits[r >> 5] |= 1u << (r & 31);
                ++ctx->nearPsqRows;
            } else {
                compactBits[r >> 5] |= 1u << (r & 31);
                ++ctx->compactPsqRows;
            }
        }
        if (!outliers.empty() &&
            (rc = uploadArray(ctx->dOutlierTab, outliers.data(), outliers.size() * sizeof(uint32_t), ctx->stream)) != SPX_OK) {
            return rc;
        }
        if ((rc = uploadArray(ctx->dThrW, thr.data(), thr.size(), ctx->stream)) != SPX_OK) return rc;
    }
    if ((rc = uploadArray(ctx->dFtBias, b + kOffFtBias, kFtBiasBytes, ctx->stream)) != SPX_OK) return rc;
    {
        std::vector<int8_t> l1(kL1WBytes);
        relayoutL1(net->l1W(), l1.data());
        if ((rc = uploadArray(ctx->dL1W, l1.data(), l1.size(), ctx->stream)) != SPX_OK) return rc;
    }
    if ((rc = uploadArray(ctx->dL1B, b + kOffL1B, kL1BBytes, ctx->stream)) != SPX_OK) return rc;
    {
        std::vector<int32_t> l2(kL2WBytes / 4);
        relayoutL2(reinterpret_cast<const int32_t*>(b + kOffL2W), l2.data());
        if ((rc = uploadArray(ctx->dL2W, l2.data(), kL2WBytes, ctx->stream)) != SPX_OK) return rc;
    }
    if ((rc = uploadArray(ctx->dL2B, b + kOffL2B, kL2BBytes, ctx->stream)) != SPX_OK) return rc;
    if ((rc = uploadArray(ctx->dL3W, b + kOffL3W, kL3WBytes, ctx->stream)) != SPX_OK) return rc;
    if ((rc = uploadArray(ctx->dL3B, b + kOffL3B, kL3BBytes, ctx->stream)) != SPX_OK) return rc;
    {
        uint32_t lut[kLutWords];
        if (buildThreatLut(lut) != int(kThreatOnlyRows)) {
            setError("internal: threat LUT does not cover 59808 features");
            return SPX_ERR_BAD_NET;
        }
        std::memcpy(lut + kLutCompactBase, compactBits, sizeof(compactBits));
        std::memcpy(lut + kLutNearBase, nearBits, sizeof(nearBits));
        std::memcpy(ctx->compactBits, compactBits, sizeof(compactBits));
        std::memcpy(ctx->nearBits, nearBits, sizeof(nearBits));
        if ((rc = uploadArray(ctx->dLut, lut, sizeof(lut), ctx->stream)) != SPX_OK) return rc;
        std::vector<uint64_t> tab(kDeltaTabWords);
        buildDeltaTables(tab.data());
        if ((rc = uploadArray(ctx->dDeltaTab, tab.data(), tab.size() * sizeof(uint64_t), ctx->stream)) != SPX_OK) return rc;
    }
    SPX_HIP(hipMalloc(&ctx->dPositions, max_batch * sizeof(spx_packed_pos)));
    SPX_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->dOut), max_batch * sizeof(int32_t)));
    SPX_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->dFtOut), max_batch * size_t(kL1)));
    SPX_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->dKingKeys), max_batch * 2));
    SPX_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->dOutKeys), max_batch));
    SPX_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->dHist), (3 * kHistWords + 64) * sizeof(uint32_t)));
    SPX_HIP(hipMemset(ctx->dHist, 0, (3 * kHistWords + 64) * sizeof(uint32_t)));
    SPX_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->dPerspOrder), max_batch * 2 * sizeof(uint32_t)));
    SPX_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->dPosOrder), max_batch * sizeof(uint32_t)));
    SPX_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->dRefreshList), max_batch * 2 * sizeof(uint32_t)));
    SPX_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->dSlotsA), max_batch * sizeof(uint32_t)));
    SPX_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->dSlotsB), max_batch * sizeof(uint32_t)));
    SPX_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->dStaged), max_batch * 32));
    ctx->ftxEnabled = !(flags & (SPX_CTX_ONE_KERNEL_FT | SPX_CTX_WIDE_PSQ_ROWS)) || (flags & SPX_CTX_SLICED_FT);
    // A/B on MI355X (tools/gpu_small_ab.sh, us per incremental ply unsplit/unshared -> split+shared): 1 024 records
    // 52.5 -> 34.0, 4 096: 60.3 -> 53.4, 8 192: 87.4 -> 81.8; split alone 32 768: 263 -> 247, 65 536: 471 -> 455,
    // 131 072: 873 -> 857, 524 288: 3276 -> 3295; sharing tiles costs throughput from 16 384 positions on
    ctx->streamAccMin = 32768;
    ctx->mlpShareMax = 8192;
    ctx->tinyBatchMax = kTinyIoRecords;  // MI355X, us per synchronous host call without -> with: 1 position 48 -> 28, 1 024: 68 -> 37, 2 048: 73 -> 47, 4 096: 93 -> 71, 8 192: 126 -> 117
    SPX_HIP(hipHostMalloc(&ctx->hTinyIo, kTinyIoRecords * kTinyIoBytesPerRecord, hipHostMallocMapped));
    ctx->updateSplitMaxV2 = 16384;
    {
        const int32_t* w = reinterpret_cast<const int32_t*>(b + kOffL2W);
        bool small = true;
        for (size_t i = 0; i < kL2WBytes / 4 && small; ++i) small = w[i] > -(1 << 23) && w[i] < (1 << 23);
        ctx->smallL2Weights = small;
    }

    hipDeviceProp_t prop;
    SPX_HIP(hipGetDeviceProperties(&prop, device));
    // persistent-ish grid: workgroups (4 waves) per CU, grid-stride over perspectives. A/B on MI355X: 2 -> 0.72 ms,
    // 4 -> 0.575, 8 -> 0.565, 16 -> 0.557, 64 -> 0.554 (finer-grained tail balancing)
    // round 2, with the round-robin chunk traversal (tools/gpu_r02_p.sh): 16 -> 0.4427, 32 -> 0.4204, 48 -> 0.4160, 64 -> 0.4203, 96 -> 0.4284
    ctx->computeUnits = uint32_t(prop.multiProcessorCount);
    ctx->ftGridCap = ctx->computeUnits * 48u;
    // the update kernel measured on the same sweep (tools/gpu_r02_r.sh, us per ply at 65 536 records):
    // 16 -> 318.5, 32 -> 318.2, 48 -> 325.2, 64 -> 336.0; finer, on the final build (tools/gpu_r02_ag.sh): 12 -> 317.5,
    // 16 -> 315.8, 20 -> 311.8, 24 -> 309.7, 28 -> 313.3, 32 -> 316.1 (self-play at 4 096 games: +1.6 % at 24 too)
    ctx->updateGridCap = ctx->computeUnits * 24u;
    SPX_HIP(hipDeviceSynchronize());  // the hist memset ran on the null stream, the context's stream does not wait for it
    for (const auto& [name, v] : envOptions) {
        if (name == "scratch_cap" || name == "compact_rows" || name == "near_rows") continue;
        if ((rc = spx_ctx_set_option(ctx.get(), name.c_str(), v)) != SPX_OK) return rc;
    }
    *out = ctx.release();
    return SPX_OK;
}

void spx_ctx_destroy(spx_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    void* ptrs[] = {ctx->dPsqW, ctx->dThrW, ctx->dFtBias, ctx->dL1W, ctx->dL1B, ctx->dL2W,  ctx->dL2B,
                    ctx->dL3W,  ctx->dL3B, ctx->dLut,    ctx->dDeltaTab, ctx->dOutlierTab, ctx->dPositions, ctx->dOut, ctx->dFtOut,
                    ctx->dKingKeys, ctx->dOutKeys, ctx->dHist, ctx->dPerspOrder, ctx->dPosOrder, ctx->dRefreshList,
                    ctx->dArena, ctx->dSlotRecords, ctx->dSlotsA, ctx->dSlotsB, ctx->dStaged, ctx->dDeltas};
    for (void* p : ptrs) {
        if (p) (void)hipFree(p);
    }
    ctx->ftx.release();
    for (void* q : {static_cast<void*>(ctx->dRowS), static_cast<void*>(ctx->dHotHash), static_cast<void*>(ctx->dHotS),
                    static_cast<void*>(ctx->dHotIds), static_cast<void*>(ctx->dHotCounts), static_cast<void*>(ctx->dHiMask)}) {
        if (q) (void)hipFree(q);
    }
    for (hipEvent_t e : ctx->profEvents) (void)hipEventDestroy(e);
    if (ctx->hTinyIo) (void)hipHostFree(ctx->hTinyIo);
    if (ctx->fallbackDone) (void)hipEventDestroy(ctx->fallbackDone);
    for (auto& lane : ctx->lanes) {
        if (lane.stream) (void)hipStreamSynchronize(lane.stream);
        void* lanePtrs[] = {lane.dFtOut, lane.dKingKeys, lane.dOutKeys, lane.dStaged, lane.dHist, lane.dPerspOrder,
                            lane.dPosOrder, lane.dRefreshList, lane.dIn, lane.dOutStage};
        if (lane.hIn) (void)hipHostFree(lane.hIn);
        if (lane.hOut) (void)hipHostFree(lane.hOut);
        for (void* q : lanePtrs) {
            if (q) (void)hipFree(q);
        }
        lane.ftx.release();
        if (lane.ftDone) (void)hipEventDestroy(lane.ftDone);
        if (lane.done) (void)hipEventDestroy(lane.done);
        for (hipEvent_t e : lane.pace) {
            if (e) (void)hipEventDestroy(e);
        }
        if (lane.stream) (void)hipStreamDestroy(lane.stream);
    }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// sort (both keys) on `d_records`, then the MLP over ctx->dFtOut[0..n) -> d_out
static int runSortAndMlp(spx_ctx* ctx, const void* d_records, size_t n, void* d_out, hipStream_t s, bool mlp,
                         const uint32_t* d_count = nullptr, bool outOnly = false) {
    if (!mlp) {
        SortParams sp{};
        sp.positions = static_cast<const uint64_t*>(d_records);
        sp.nPositions = uint32_t(n);
        sp.nPositionsPtr = d_count;
        sp.outOnly = outOnly;
        sp.kingKeys = ctx->dKingKeys;
        sp.outKeys = ctx->dOutKeys;
        if (n <= 1024 && !d_count) {  // single-launch path (kSmallSortMax): its own buffer, never needs clearing
            sp.hist = ctx->dHist + 2 * kHistWords;
            sp.histNext = sp.hist;
        } else {
            sp.hist = ctx->dHist + kHistWords * ctx->histCur;
            sp.histNext = ctx->dHist + kHistWords * (ctx->histCur ^ 1);
            ctx->histCur ^= 1;
        }
        ctx->histUsed = sp.hist;
        sp.perspOrder = ctx->dPerspOrder;
        sp.posOrder = ctx->dPosOrder;
        SPX_HIP(launchSort(sp, s));
        return SPX_OK;
    }
    MlpParams mp{};
    mp.nPositions = uint32_t(n);
    mp.posOrder = ctx->dPosOrder;
    mp.hist = ctx->histUsed;
    mp.ftOut = ctx->dFtOut;
    mp.l1W = ctx->dL1W;
    mp.l1B = ctx->dL1B;
    mp.l2W = ctx->dL2W;
    mp.l2B = ctx->dL2B;
    mp.l3W = ctx->dL3W;
    mp.l3B = ctx->dL3B;
    mp.out = static_cast<int32_t*>(d_out);
    SPX_HIP(launchMlp(mp, ctx->smallL2Weights, (n <= ctx->mlpShareMax && !d_count) ? kMlpTileShared : kMlpTileSorted, s));
    return SPX_OK;
}

static uint32_t cappedGrid(size_t waves, uint32_t cap) {
    const uint32_t wavesPerBlock = ftWavesPerBlock();
    uint32_t blocks = uint32_t((waves + wavesPerBlock - 1) / wavesPerBlock);
    if (blocks > cap) blocks = cap;
    return (blocks + 7u) & ~7u;  // whole multiples of the 8 XCDs
}
static uint32_t ftGrid(const spx_ctx* ctx, size_t waves) { return cappedGrid(waves, ctx->ftGridCap); }
static uint32_t updateGrid(const spx_ctx* ctx, size_t waves) { return cappedGrid(waves, ctx->updateGridCap); }
// full refresh of nPersp perspectives: one wave each, or - launches too small to fill the wave slots, which are bound by the
// latency of a single perspective - one workgroup each (spx_ft_team_kernel)
static hipError_t launchFullFt(const spx_ctx* ctx, const FtParams& fp, size_t nPersp, hipStream_t s) {
    if (nPersp <= ctx->teamMaxPersp) return launchFtTeam(fp, ftGrid(ctx, nPersp * ftWavesPerBlock()), s);
    return launchFt(fp, ftGrid(ctx, nPersp), s);
}

// MLP of a handful of positions without any sort: every position is its own tile and finds its bucket from its record
static int runTinyMlp(spx_ctx* ctx, const void* d_records, size_t n, void* d_out, hipStream_t s) {
    MlpParams mp{};
    mp.nPositions = uint32_t(n);
    mp.records = static_cast<const uint64_t*>(d_records);
    mp.ftOut = ctx->dFtOut;
    mp.l1W = ctx->dL1W;
    mp.l1B = ctx->dL1B;
    mp.l2W = ctx->dL2W;
    mp.l2B = ctx->dL2B;
    mp.l3W = ctx->dL3W;
    mp.l3B = ctx->dL3B;
    mp.out = static_cast<int32_t*>(d_out);
    SPX_HIP(launchMlp(mp, ctx->smallL2Weights, kMlpTilePerPosition, s));
    return SPX_OK;
}

// ---- the gather's hot set ----
// the tables of a chosen set: row -> slot map, the rows' slices in slot order. The caller guarantees that no gather of this context is
// in flight (lists hold LDS offsets of the set they were extracted under).
static int installHotRows(spx_ctx* ctx, const std::vector<uint32_t>& ids, hipStream_t s) {
    const uint32_t n = uint32_t(std::min<size_t>(ids.size(), kFtxHotRowsMax));
    // no gather of this context may be in flight: lists hold LDS offsets of the set they were extracted under
    SPX_HIP(hipStreamSynchronize(ctx->stream));
    for (auto& lane : ctx->lanes) {
        if (lane.stream) SPX_HIP(hipStreamSynchronize(lane.stream));
    }
    // the extraction's view of the set: a hash of 256 buckets x 4 entries (row | slot << 16), the multiplier that leaves the fewest
    // rows without a place (such a row is simply cold: its slot in the gather's LDS goes unused)
    std::vector<uint32_t> best(kFtxHotHashWords, 0xFFFFFFFFu);
    uint32_t bestMul = 1, bestLost = n + 1;
    for (uint32_t trial = 0; trial < 64 && bestLost; ++trial) {
        const uint32_t mul = (0x9E3779u + 0x3C6EF3u * trial) | 1u;  // (24-bit odd multipliers: the kernel uses v_mul_u32_u24)
        std::vector<uint32_t> table(kFtxHotHashWords, 0xFFFFFFFFu);
        uint32_t lost = 0;
        for (uint32_t slot = 0; slot < n; ++slot) {
            const uint32_t bucket = (((ids[slot] & 0xFFFFFFu) * (mul & 0xFFFFFFu)) >> 16) & (kFtxHotHashWords / 4 - 1);
            uint32_t i = 0;
            while (i < 4 && table[4 * bucket + i] != 0xFFFFFFFFu) ++i;
            if (i < 4) table[4 * bucket + i] = ids[slot] | (slot << 16); else ++lost;
        }
        if (lost < bestLost) bestLost = lost, bestMul = mul & 0xFFFFFFu, best.swap(table);
    }
    ctx->hotHashMul = bestMul;
    SPX_HIP(hipMemcpyAsync(ctx->dHotHash, best.data(), kFtxHotHashWords * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    if (n) SPX_HIP(hipMemcpyAsync(ctx->dHotIds, ids.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    SPX_HIP(launchFtxBuildHot(ctx->dRowS, ctx->dHotIds, n, ctx->dHotS, s));
    SPX_HIP(hipStreamSynchronize(s));  // (the host buffer may go; other streams may use the tables next)
    ctx->hotRows = n;
    return SPX_OK;
}

// Chooses the set from the batch `xp` describes: its lists extracted with an EMPTY set, a histogram of their threat / pawn-pair
// rows, the most popular hotRowsWanted of them (ties: the lower row id - the choice is deterministic for a batch).
static int calibrateHotRows(spx_ctx* ctx, FtxParams xp, hipStream_t s) {
    ctx->hotCalibrated = false;  // (true once the set is installed: a failure on the way leaves the empty set and the next batch retries)
    ctx->hotRows = 0;
    ctx->coldShift = 1;
    ctx->hotIds.clear();
    const uint32_t wanted = std::min(ctx->hotRowsWanted, kFtxHotRowsMax);
    if (!wanted || xp.nPositions == 0) {
        ctx->hotCalibrated = true;
        return SPX_OK;
    }
    xp.hotHash = ctx->dHotHash;
    xp.hotHashMul = ctx->hotHashMul;
    xp.hotS = ctx->dHotS;
    xp.hotRows = 0;
    xp.coldShift = 1;
    SPX_HIP(hipMemsetAsync(ctx->dHotCounts, 0, (kThreatRows + 16) * sizeof(uint32_t), s));
    SPX_HIP(launchFtxExtract(xp, s));
    SPX_HIP(launchFtxHistogram(xp, ctx->dHotCounts, ctx->dHotCounts + kThreatRows, s));
    std::vector<uint32_t> counts(kThreatRows + 16);
    SPX_HIP(hipMemcpyAsync(counts.data(), ctx->dHotCounts, counts.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    SPX_HIP(hipStreamSynchronize(s));
    std::vector<uint32_t> order;
    order.reserve(8192);
    uint64_t total = 0;
    for (uint32_t r = 0; r < kThreatRows; ++r) {
        total += counts[r];
        if (counts[r]) order.push_back(r);
    }
    const size_t take = std::min<size_t>(wanted, order.size());
    std::partial_sort(order.begin(), order.begin() + take, order.end(),
                      [&](uint32_t a, uint32_t b) { return counts[a] != counts[b] ? counts[a] > counts[b] : a < b; });
    order.resize(take);
    uint64_t covered = 0;
    for (uint32_t r : order) covered += counts[r];
    ctx->hotIds = order;
    const int rc = installHotRows(ctx, ctx->hotIds, s);
    if (rc != SPX_OK) return rc;
    // sort key: global quartets (high planes + cold rows) in 16 classes; with more than ~9 of them per perspective on average the
    // classes are two quartets wide
    const double perPersp = 2.0 * xp.nPositions;
    const double globalQ = double(total - covered) / perPersp / 4.0 + double(counts[kThreatRows]) / perPersp / 4.0 + 1.0;
    ctx->coldShift = globalQ > 9.0 ? 1u : 0u;
    ctx->hotCalibrated = true;
    return SPX_OK;
}

// the column-sliced pipeline's table (per context) and scratch (per lane; `ctx->ftx` is the set swapped in): allocated on
// first use; a context sized to fill the HBM that has no room for them keeps the one-kernel path (false)
static bool ensureFtx(spx_ctx* ctx, FtxScratch& x, size_t passPositions, hipStream_t s) {
    if (!ctx->ftxEnabled || ctx->ftxUnavailable) return false;
    auto fail = [&]() {
        (void)hipGetLastError();
        x.release();
        ctx->ftxUnavailable = true;
        return false;
    };
    if (!ctx->dRowS) {
        // the gather needs a full-size MI355X: 256 workgroups = (XCD b % 8 = column slice, CU slot b / 8), one per CU, each with
        // ~140 KiB of LDS. On a partitioned / CU-masked device they would still add up the right sums but lose the slicing of the
        // L2s (and may not be co-resident): such devices keep the one-kernel path
        if (ctx->computeUnits != 256 || prepareFtxGather(ctx->device) != hipSuccess) return fail();
        if (hipMalloc(reinterpret_cast<void**>(&ctx->dRowS), kFtxTableBytes) != hipSuccess) return fail();
        if (hipMalloc(reinterpret_cast<void**>(&ctx->dHotHash), kFtxHotHashWords * sizeof(uint32_t)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&ctx->dHotS), size_t(8) * kFtxHotRowsMax * 128) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&ctx->dHotIds), kFtxHotRowsMax * sizeof(uint32_t)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&ctx->dHotCounts), (kThreatRows + 16) * sizeof(uint32_t)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&ctx->dHiMask), kPsqRows) != hipSuccess) {
            return fail();
        }
        if (hipMemsetAsync(ctx->dHotHash, 0xFF, kFtxHotHashWords * sizeof(uint32_t), s) != hipSuccess) return fail();
        if (launchFtxBuildTable(ctx->dThrW, ctx->dPsqW, ctx->dLut, ctx->dRowS, s) != hipSuccess) return fail();
        if (launchFtxBuildHiMask(ctx->dRowS, ctx->dHiMask, s) != hipSuccess) return fail();
        // (other streams may use the tables next: the lanes' streams do not wait for this one)
        if (hipStreamSynchronize(s) != hipSuccess) return fail();
        if (!ctx->hotIds.empty() && installHotRows(ctx, ctx->hotIds, s) != SPX_OK) return fail();  // (a set given before the first batch)
    }
    if (x.capacity >= passPositions) return true;
    x.release();
    // (option ftx_fail_after = k, tests only: the k-th scratch set "does not fit" - what a context sized to fill the HBM runs into)
    if (ctx->ftxFailAfter >= 0 && ctx->ftxScratchSets++ >= ctx->ftxFailAfter) return fail();
    const size_t cap = std::min(ctx->maxBatch, kFtxMaxPositions);
    auto alloc = [&](uint32_t*& ptr, size_t bytes) { return hipMalloc(reinterpret_cast<void**>(&ptr), bytes) == hipSuccess; };
    if (!alloc(x.lists, ftxListBytes(cap)) || !alloc(x.heads, 2 * cap * 16) ||
        !alloc(x.ranks, 2 * cap * 4) || !alloc(x.hist, kFtxBins * 4) || !alloc(x.binStart, (kFtxBins + 17 + 8) * 4) ||
        !alloc(x.outHist, (kHistOut + 16) * 4) ||
        !alloc(x.sorted, (2 * cap + 128) * 16) || !alloc(x.plan, kFtxPlanWords * 4) ||
        !alloc(x.groupHead, ftxGroups(cap) * kFtxGroupHeadWords * 4) || !alloc(x.stages, ftxStageBytes(cap))) {
        return fail();
    }
    if (hipMemsetAsync(x.hist, 0, kFtxBins * 4, s) != hipSuccess) return fail();
    if (hipMemsetAsync(x.outHist, 0, (kHistOut + 16) * 4, s) != hipSuccess) return fail();
    x.capacity = cap;
    return true;
}

int spx_eval_full_device(spx_ctx* ctx, const void* d_positions, size_t n, void* d_out, void* stream) {
    if (!ctx || (n && (!d_positions || !d_out))) {
        setError("spx_eval_full_device: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    if (n > ctx->callLimit) {
        setError("batch of " + std::to_string(n) + " exceeds context capacity " + std::to_string(ctx->callLimit));
        return SPX_ERR_CAPACITY;
    }
    if (n == 0) {
        return SPX_OK;
    }
    if (n > ctx->maxBatch) {  // more positions than the scratch holds: chunk by chunk, stream-ordered
        for (size_t lo = 0; lo < n; lo += ctx->maxBatch) {
            const int rc = spx_eval_full_device(ctx, static_cast<const char*>(d_positions) + lo * sizeof(spx_packed_pos),
                                                std::min(ctx->maxBatch, n - lo), static_cast<int32_t*>(d_out) + lo, stream);
            if (rc != SPX_OK) return rc;
        }
        return SPX_OK;
    }
    SPX_HIP(hipSetDevice(ctx->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    hipEvent_t* ev = nullptr;
    bool profiled = false;
    if (ctx->profUsed + kProfEventsPerCall <= ctx->profEvents.size()) {
        ev = &ctx->profEvents[ctx->profUsed];
        ctx->profUsed += kProfEventsPerCall;
        profiled = true;
    } else if (ctx->paceEvents && ctx->paceEnabled) {
        ev = ctx->paceEvents;
    }
    if (ev && (profiled || (ctx->paceMask >> 0 & 1u))) SPX_HIP(hipEventRecord(ev[0], s));
    if (ev && !profiled) {
        for (uint32_t r = 0; r < ctx->paceHeadExtra; ++r) SPX_HIP(hipEventRecord(ev[0], s));
    }
    // big batches: the column-sliced pipeline (spx_ftx.hip); it orders the perspectives itself, so only the MLP's
    // output-bucket order is sorted here
    FtxScratch& scratch = ctx->ftx;
    // (pipelined calls - a lane's gate is set - gain from 6 Ki positions on, stream-ordered ones from 10 Ki:
    // profiles/r06_sliced_pipeline_small_batches.txt)
    const size_t sliceFrom = (ctx->ftGateRecord && !ctx->ftxMinForced) ? std::min(ctx->ftxMin, kFtxMinPositionsPipelined) : ctx->ftxMin;
    // a handful of positions: no sort launch, every position its own MLP tile (a pipelined call that reaches the pipeline's
    // threshold takes the pipeline, whatever tiny_batch_max says)
    const bool tiny = n <= ctx->tinyBatchMax && !(ctx->ftGateRecord && n >= sliceFrom && ctx->ftxEnabled && !ctx->ftxUnavailable);
    const bool sliced = !tiny && n >= sliceFrom && ensureFtx(ctx, scratch, std::min(n, kFtxMaxPositions), s);
    // a one-pass batch of the pipeline gets the MLP's output-bucket order from the pipeline's own sort (FtxParams::posOrder): no
    // spx_sort_* launches (option ftx_fold_sort = 0: as before)
    const bool foldSort = sliced && ctx->ftxFoldSort && n <= scratch.capacity;
    int rc = (tiny || foldSort) ? SPX_OK : runSortAndMlp(ctx, d_positions, n, nullptr, s, false, nullptr, sliced);
    if (rc != SPX_OK) return rc;
    if (ev && (profiled || (ctx->paceMask >> 1 & 1u))) SPX_HIP(hipEventRecord(ev[1], s));
    if (sliced) {
        // passes of at most the scratch's capacity; a pass's preparation (extraction, sort, plan) comes before the
        // pipelined calls' gate, so that it runs beside another batch's gather
        for (size_t lo = 0; lo < n; lo += scratch.capacity) {
            const size_t m = std::min(scratch.capacity, n - lo);
            FtxParams xp{};
            xp.positions = static_cast<const char*>(d_positions) + lo * sizeof(spx_packed_pos);
            xp.nPositions = uint32_t(m);
            xp.t = tablesOf(ctx);
            xp.rowS = ctx->dRowS;
            xp.lists = scratch.lists;
            xp.heads = scratch.heads;
            xp.ranks = scratch.ranks;
            xp.hist = scratch.hist;
            xp.binStart = scratch.binStart;
            xp.sorted = scratch.sorted;
            xp.plan = scratch.plan;
            xp.groupHead = scratch.groupHead;
            xp.stages = scratch.stages;
            xp.hiMask = ctx->dHiMask;
            xp.ftOut = ctx->dFtOut + lo * size_t(kL1);
            xp.posOrder = foldSort ? ctx->dPosOrder : nullptr;
            xp.outCounts = scratch.outHist + kHistOut + 8;  // (the 8 words behind the counts the MLP reads)
            xp.mlpHist = scratch.outHist;
            if (foldSort) ctx->histUsed = scratch.outHist;
            if (!ctx->hotCalibrated && ctx->hotAutoCalibrate) {
                // the first big batch of this context chooses the hot set: one extra extraction + a histogram, and the host WAITS for
                // them (and for the context's other streams) inside this call. A stream that is being captured into a hipGraph must
                // not be synchronised: such a call runs with the set as it is - empty unless one was given - and the
                // first call outside a capture calibrates; option ftx_auto_calibrate = 0 leaves the choice to spx_ctx_calibrate /
                // spx_ctx_set_hot_rows altogether
                hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
                if (hipStreamIsCapturing(s, &capture) != hipSuccess) (void)hipGetLastError();
                if (capture == hipStreamCaptureStatusNone && (rc = calibrateHotRows(ctx, xp, s)) != SPX_OK) return rc;
            }
            xp.hotHash = ctx->dHotHash;
            xp.hotHashMul = ctx->hotHashMul;
            xp.hotS = ctx->dHotS;
            xp.hotRows = ctx->hotRows;
            xp.coldShift = ctx->coldShift;
            // a pipelined call: the preparation is not gated - it runs beside the other lane's gather and MLP -, the gather is
            // (the two lanes' gathers are chained). Gating the preparation too: 1.58 instead of 1.81e8 evals/s; no gate at all: 1.83e8,
            // but then the gather's event interval includes its wait for free CUs
            // A launch the runtime refuses (the header promises the one-kernel path whenever the pipeline cannot run):
            // the pipeline is given up for this context and the WHOLE batch goes through spx_ft_kernel - what the passes issued
            // so far wrote is overwritten in stream order; the sort histogram a half-run preparation leaves behind is zeroed
            hipError_t launched = (ctx->ftxFailLaunch >= 0 && ctx->ftxFailLaunch-- == 0) ? hipErrorLaunchFailure : launchFtxPrepare(xp, s);
            if (launched == hipSuccess && lo == 0) {
                if (ctx->ftGateWait) SPX_HIP(hipStreamWaitEvent(s, ctx->ftGateWait, 0));
                if (ev && (profiled || (ctx->paceMask >> 4 & 1u))) SPX_HIP(hipEventRecord(ev[4], s));
            }
            if (launched == hipSuccess) launched = launchFtxGather(xp, s);
            if (launched != hipSuccess) {
                (void)hipGetLastError();
                ctx->ftxUnavailable = true;
                SPX_HIP(hipMemsetAsync(scratch.hist, 0, kFtxBins * 4, s));
                SPX_HIP(hipMemsetAsync(scratch.outHist, 0, (kHistOut + 16) * 4, s));
                if (profiled) ctx->profUsed -= kProfEventsPerCall;
                return spx_eval_full_device(ctx, d_positions, n, d_out, stream);
            }
        }
    } else {
        if (ctx->ftGateWait) SPX_HIP(hipStreamWaitEvent(s, ctx->ftGateWait, 0));  // pipelined calls: FT kernels are chained
        if (ev && (profiled || (ctx->paceMask >> 4 & 1u))) SPX_HIP(hipEventRecord(ev[4], s));  // after the wait: the FT interval is the kernel alone
        FtParams fp{};
        fp.positions = d_positions;
        fp.nPositions = uint32_t(n);
        fp.order = (ctx->kingSortEnabled && !tiny) ? ctx->dPerspOrder : nullptr;
        fp.t = tablesOf(ctx);
        fp.ftOut = ctx->dFtOut;
        SPX_HIP(launchFullFt(ctx, fp, 2 * n, s));
    }
    if (ctx->ftGateRecord) SPX_HIP(hipEventRecord(ctx->ftGateRecord, s));
    if (ev && (profiled || (ctx->paceMask >> 2 & 1u))) SPX_HIP(hipEventRecord(ev[2], s));
    rc = tiny ? runTinyMlp(ctx, d_positions, n, d_out, s) : runSortAndMlp(ctx, d_positions, n, d_out, s, true);
    if (rc != SPX_OK) return rc;
    if (ev && (profiled || (ctx->paceMask >> 3 & 1u))) SPX_HIP(hipEventRecord(ev[3], s));
    return SPX_OK;
}

// ---- pipelined full-refresh evaluation ----
static void swapLane(spx_ctx* ctx, spx_ctx::EvalLane& lane) {
    std::swap(ctx->dFtOut, lane.dFtOut);
    std::swap(ctx->dKingKeys, lane.dKingKeys);
    std::swap(ctx->dOutKeys, lane.dOutKeys);
    std::swap(ctx->dStaged, lane.dStaged);
    std::swap(ctx->dHist, lane.dHist);
    std::swap(ctx->dPerspOrder, lane.dPerspOrder);
    std::swap(ctx->dPosOrder, lane.dPosOrder);
    std::swap(ctx->dRefreshList, lane.dRefreshList);
    std::swap(ctx->histUsed, lane.histUsed);
    std::swap(ctx->histCur, lane.histCur);
    std::swap(ctx->refreshCur, lane.refreshCur);
    std::swap(ctx->ftx, lane.ftx);
}

static int ensureLanes(spx_ctx* ctx) {
    if (ctx->lanesReady) return SPX_OK;
    // The two lanes must sit on different hardware queues or their kernels serialise: HIP spreads streams of one
    // priority round-robin over a small pool of queues (two lanes were seen sharing one), but never mixes priorities.
    int leastPriority = 0, greatestPriority = 0;
    SPX_HIP(hipDeviceGetStreamPriorityRange(&leastPriority, &greatestPriority));
    int laneIndex = 0;
    for (auto& lane : ctx->lanes) {
        SPX_HIP(hipMalloc(reinterpret_cast<void**>(&lane.dFtOut), ctx->maxBatch * size_t(kL1)));
        SPX_HIP(hipMalloc(reinterpret_cast<void**>(&lane.dKingKeys), ctx->maxBatch * 2));
        SPX_HIP(hipMalloc(reinterpret_cast<void**>(&lane.dOutKeys), ctx->maxBatch));
        SPX_HIP(hipMalloc(reinterpret_cast<void**>(&lane.dStaged), ctx->maxBatch * 32));
        SPX_HIP(hipMalloc(reinterpret_cast<void**>(&lane.dHist), (3 * kHistWords + 64) * sizeof(uint32_t)));
        SPX_HIP(hipMemset(lane.dHist, 0, (3 * kHistWords + 64) * sizeof(uint32_t)));
        SPX_HIP(hipMalloc(reinterpret_cast<void**>(&lane.dPerspOrder), ctx->maxBatch * 2 * sizeof(uint32_t)));
        SPX_HIP(hipMalloc(reinterpret_cast<void**>(&lane.dPosOrder), ctx->maxBatch * sizeof(uint32_t)));
        SPX_HIP(hipMalloc(reinterpret_cast<void**>(&lane.dRefreshList), ctx->maxBatch * 2 * sizeof(uint32_t)));
        // (lanes 0 / 1: the two ends of the range; lane 2: the level between them, where the device has one)
        const int priority = laneIndex == 0 ? leastPriority : (laneIndex == 1 ? greatestPriority : (leastPriority + greatestPriority) / 2);
        ++laneIndex;
        SPX_HIP(hipStreamCreateWithPriority(&lane.stream, hipStreamNonBlocking, priority));
        SPX_HIP(hipEventCreateWithFlags(&lane.ftDone, hipEventDisableTiming));
        SPX_HIP(hipEventCreateWithFlags(&lane.done, hipEventDisableTiming));
        for (hipEvent_t& e : lane.pace) SPX_HIP(hipEventCreate(&e));
    }
    SPX_HIP(hipDeviceSynchronize());  // the memsets ran on the null stream, the lanes' streams do not wait for it
    ctx->lanesReady = true;
    return SPX_OK;
}

static void releaseLanes(spx_ctx* ctx) {
    for (auto& lane : ctx->lanes) {
        if (lane.stream) (void)hipStreamSynchronize(lane.stream);
        void* lanePtrs[] = {lane.dFtOut, lane.dKingKeys, lane.dOutKeys, lane.dStaged, lane.dHist, lane.dPerspOrder,
                            lane.dPosOrder, lane.dRefreshList, lane.dIn, lane.dOutStage};
        for (void* q : lanePtrs) {
            if (q) (void)hipFree(q);
        }
        lane.ftx.release();
        if (lane.hIn) (void)hipHostFree(lane.hIn);
        if (lane.hOut) (void)hipHostFree(lane.hOut);
        if (lane.ftDone) (void)hipEventDestroy(lane.ftDone);
        if (lane.done) (void)hipEventDestroy(lane.done);
        if (lane.stream) (void)hipStreamDestroy(lane.stream);
        lane = spx_ctx::EvalLane{};
    }
    ctx->lanesReady = false;
}

int spx_eval_full_device_async(spx_ctx* ctx, const void* d_positions, size_t n, void* d_out, void** done_event) {
    if (!ctx) {
        setError("spx_eval_full_device_async: null context");
        return SPX_ERR_INVALID_ARG;
    }
    SPX_HIP(hipSetDevice(ctx->device));
    int rc = ctx->lanesUnavailable ? SPX_ERR_HIP : ensureLanes(ctx);
    if (rc != SPX_OK) {
        // no room for a second scratch set (a context sized to fill the HBM): same results, stream-ordered on the
        // context's own stream
        if (!ctx->lanesUnavailable) {
            (void)hipGetLastError();
            releaseLanes(ctx);
            ctx->lanesUnavailable = true;
            if (!ctx->fallbackDone) SPX_HIP(hipEventCreateWithFlags(&ctx->fallbackDone, hipEventDisableTiming));
        }
        rc = spx_eval_full_device(ctx, d_positions, n, d_out, ctx->stream);
        if (rc != SPX_OK) return rc;
        SPX_HIP(hipEventRecord(ctx->fallbackDone, ctx->stream));
        if (done_event) *done_event = ctx->fallbackDone;
        return SPX_OK;
    }
    if (n > ctx->callLimit) {
        setError("batch of " + std::to_string(n) + " exceeds context capacity " + std::to_string(ctx->callLimit));
        return SPX_ERR_CAPACITY;
    }
    // chunks of the scratch capacity alternate between the two lanes (one chunk for an ordinary batch); where the column-sliced
    // pipeline runs, chunks of one of ITS passes: a lane walks the passes of its chunk one after the other, two lanes overlap
    // them (131 072 positions per call: 1.73 -> 1.8e8 evals/s)
    const size_t chunk = (ctx->ftxEnabled && !ctx->ftxUnavailable && n > kFtxMaxPositions) ? std::min(ctx->maxBatch, kFtxMaxPositions)
                                                                                            : ctx->maxBatch;
    spx_ctx::EvalLane* last = nullptr;
    for (size_t lo = 0; lo < n || lo == 0; lo += chunk) {
        const size_t m = std::min(chunk, n - lo);
        // the batches go round the lanes; a batch's big kernel waits for that of the batch before it (the lane before this one)
        const unsigned nLanes = ctx->evalLanes, li = ctx->laneNext % nLanes;
        spx_ctx::EvalLane& lane = ctx->lanes[li];
        spx_ctx::EvalLane& other = ctx->lanes[(li + nLanes - 1) % nLanes];
        ctx->laneNext = (ctx->laneNext + 1) % 6;
        swapLane(ctx, lane);
        ctx->ftGateWait = other.ftRecorded ? other.ftDone : nullptr;
        ctx->ftGateRecord = lane.ftDone;
        ctx->paceEvents = lane.pace;
        rc = spx_eval_full_device(ctx, static_cast<const char*>(d_positions) + lo * sizeof(spx_packed_pos), m,
                                  static_cast<int32_t*>(d_out) + lo, lane.stream);
        ctx->ftGateWait = ctx->ftGateRecord = nullptr;
        ctx->paceEvents = nullptr;
        swapLane(ctx, lane);
        if (rc != SPX_OK) return rc;
        if (m) lane.ftRecorded = true;
        if (last) SPX_HIP(hipStreamWaitEvent(lane.stream, last->done, 0));  // `done` of the call covers every chunk
        SPX_HIP(hipEventRecord(lane.done, lane.stream));
        last = &lane;
        if (n == 0) break;
    }
    if (done_event) *done_event = last->done;
    return SPX_OK;
}

int spx_ctx_synchronize(spx_ctx* ctx) {
    if (!ctx) {
        setError("spx_ctx_synchronize: null context");
        return SPX_ERR_INVALID_ARG;
    }
    SPX_HIP(hipSetDevice(ctx->device));
    SPX_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->lanesReady) {
        for (auto& lane : ctx->lanes) SPX_HIP(hipStreamSynchronize(lane.stream));
    }
    return SPX_OK;
}

}  // extern "C"

namespace spx {
// Internal (spx_internal.h): run a sequence of *_device calls of this context on one of its two lanes - the lane's
// scratch set and stream - so that two independent sequences (the two halves of the self-play seats) overlap. The
// lane's big kernel (FT / update) first waits for the other lane's latest one.
int ctxLaneBegin(spx_ctx* ctx, int laneIndex, void** stream, bool gates) {
    const int rc = ensureLanes(ctx);
    if (rc != SPX_OK) return rc;
    spx_ctx::EvalLane& lane = ctx->lanes[laneIndex & 1];
    spx_ctx::EvalLane& other = ctx->lanes[(laneIndex & 1) ^ 1];
    swapLane(ctx, lane);
    ctx->ftGateWait = (gates && other.ftRecorded) ? other.ftDone : nullptr;
    ctx->ftGateRecord = gates ? lane.ftDone : nullptr;
    if (gates) lane.ftRecorded = true;  // conservatively: an unrecorded event counts as complete for hipStreamWaitEvent
    // an ungated lane may be under stream capture: spx_profile_* timing events have no place in a graph (a profile that was
    // left open ends here)
    if (!gates) ctx->profUsed = ctx->profEvents.size();
    *stream = lane.stream;
    return SPX_OK;
}

void ctxLaneEnd(spx_ctx* ctx, int laneIndex) {
    ctx->ftGateWait = ctx->ftGateRecord = nullptr;
    swapLane(ctx, ctx->lanes[laneIndex & 1]);
}
}  // namespace spx

extern "C" {

// ---- incremental path: accumulator arena ----
int spx_acc_reserve(spx_ctx* ctx, size_t n_slots) {
    if (!ctx || n_slots == 0 || n_slots > (1ull << 31)) {
        setError("spx_acc_reserve: invalid argument");
        return SPX_ERR_INVALID_ARG;
    }
    SPX_HIP(hipSetDevice(ctx->device));
    if (n_slots <= ctx->nSlots) return SPX_OK;
    SPX_HIP(hipDeviceSynchronize());
    // growing keeps every materialised slot: the old arena and records are copied into the new allocation
    uint8_t *arena = nullptr, *records = nullptr;
    SPX_HIP(hipMalloc(reinterpret_cast<void**>(&arena), n_slots * kAccSlotBytes));
    if (hipMalloc(reinterpret_cast<void**>(&records), n_slots * 32) != hipSuccess) {
        (void)hipFree(arena);
        setError("spx_acc_reserve: out of device memory");
        return SPX_ERR_HIP;
    }
    hipError_t e = hipMemset(records, 0, n_slots * 32);
    if (e == hipSuccess && ctx->nSlots) {
        e = hipMemcpy(arena, ctx->dArena, ctx->nSlots * kAccSlotBytes, hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy(records, ctx->dSlotRecords, ctx->nSlots * 32, hipMemcpyDeviceToDevice);
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();  // null-stream work vs. the non-blocking streams that use the arena next
    if (e != hipSuccess) {
        (void)hipFree(arena);
        (void)hipFree(records);
        setError(std::string("spx_acc_reserve: ") + hipGetErrorString(e));
        return SPX_ERR_HIP;
    }
    if (ctx->dArena) (void)hipFree(ctx->dArena);
    if (ctx->dSlotRecords) (void)hipFree(ctx->dSlotRecords);
    ctx->dArena = arena;
    ctx->dSlotRecords = records;
    ctx->nSlots = n_slots;
    return SPX_OK;
}

static int checkSlots(const spx_ctx* ctx, const uint32_t* slots, size_t n, const char* who);

static int checkAcc(spx_ctx* ctx, size_t n, const char* who) {
    if (!ctx) {
        setError(std::string(who) + ": null context");
        return SPX_ERR_INVALID_ARG;
    }
    if (!ctx->dArena) {
        setError(std::string(who) + ": call spx_acc_reserve first");
        return SPX_ERR_INVALID_ARG;
    }
    if (n > ctx->maxBatch) {
        setError(std::string(who) + ": batch of " + std::to_string(n) + " exceeds what one arena call accepts (" +
                 std::to_string(ctx->maxBatch) + " = min(max_batch, scratch_cap): spx_ctx_scratch_batch); only "
                 "spx_eval_full* walk larger batches in chunks");
        return SPX_ERR_CAPACITY;
    }
    SPX_HIP(hipSetDevice(ctx->device));  // every arena entry point passes through here: contexts on other GPUs stay independent
    return SPX_OK;
}

int spx_acc_refresh_device(spx_ctx* ctx, const void* d_positions, const void* d_slots, size_t n, void* stream) {
    int rc = checkAcc(ctx, n, "spx_acc_refresh_device");
    if (rc != SPX_OK || n == 0) return rc;
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    rc = runSortAndMlp(ctx, d_positions, n, nullptr, s, false);
    if (rc != SPX_OK) return rc;
    FtParams fp{};
    fp.positions = d_positions;
    fp.nPositions = uint32_t(n);
    fp.order = ctx->kingSortEnabled ? ctx->dPerspOrder : nullptr;
    fp.t = tablesOf(ctx);
    fp.ftOut = nullptr;
    fp.accOut = ctx->dArena;
    fp.slots = static_cast<const uint32_t*>(d_slots);
    fp.slotRecords = ctx->dSlotRecords;
    SPX_HIP(launchFullFt(ctx, fp, 2 * n, s));
    return SPX_OK;
}

static int launchUpdateAndRefresh(spx_ctx* ctx, UpdateParams& up, size_t n, hipStream_t s);

int spx_acc_update_device(spx_ctx* ctx, const void* d_parent_slots, const void* d_child_slots,
                          const void* d_child_positions, size_t n, void* stream) {
    int rc = checkAcc(ctx, n, "spx_acc_update_device");
    if (rc != SPX_OK || n == 0) return rc;
    if (!d_child_slots) {  // (only the *_eval entry points have an eval-only mode)
        setError("spx_acc_update_device: null child slots (an update without child slots and without evaluation does nothing)");
        return SPX_ERR_INVALID_ARG;
    }
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    UpdateParams up{};
    up.nRecords = uint32_t(n);
    up.parentSlots = static_cast<const uint32_t*>(d_parent_slots);
    up.childSlots = static_cast<const uint32_t*>(d_child_slots);
    up.childPositions = d_child_positions;
    up.t = tablesOf(ctx);
    up.arena = ctx->dArena;
    up.slotRecords = ctx->dSlotRecords;
    return launchUpdateAndRefresh(ctx, up, n, s);
}

static int updateEvalDevice(spx_ctx* ctx, const void* d_parent_slots, const void* d_child_slots,
                            const void* d_child_positions, size_t n, const uint32_t* d_count, void* d_out, void* stream,
                            const char* who);

// The update kernel proper, followed (second-generation kernel) by the pass that rebuilds the perspectives it deferred:
// the feature-transformer kernel over the refresh list (ids in the context's dRefreshList, the count in one of two alternating
// device words behind the sort histograms; the pass clears the other one for the next update).
static int launchUpdateAndRefresh(spx_ctx* ctx, UpdateParams& up, size_t n, hipStream_t s) {
    const bool split = n <= ctx->updateSplitMaxV2;  // one wave per (record, perspective)
    uint32_t* counters = ctx->dHist + 3 * kHistWords;
    up.refreshList = ctx->dRefreshList;
    up.refreshCount = counters + ctx->refreshCur;
    // streaming (non-temporal) arena accesses only where accumulators are written: eval-only children keep cached parents
    const bool streamAcc = n >= ctx->streamAccMin && up.childSlots != nullptr;
    if (n <= ctx->updateChainMax && !up.nRecordsPtr) {
        // the smallest batches: the chain kernel on unit paths - one wave per (record, perspective), rebuilds inline, ONE launch: a
        // synchronous update + eval of 1 / 1 024 records 28.7 / 39.7 us. From 2 048 records on its 163 VGPRs (3 waves per SIMD)
        // lose to spx_update_kernel + the rebuild pass (profiles/r03_ab_small_update_chain_kernel.txt; rounds 3-4 served 1 025 ..
        // 8 192 records with the round-1 kernel, ~5 % faster there - retired in round 5, experiments/r01_update_kernel_board_diff.hip.txt)
        ChainParams cp{};
        cp.nChains = uint32_t(n);
        cp.parentSlots = up.parentSlots;
        cp.childSlots = up.childSlots;
        cp.childPositions = up.childPositions;
        cp.t = up.t;
        cp.arena = up.arena;
        cp.slotRecords = up.slotRecords;
        cp.ftOut = up.ftOut;
        cp.stagedRecords = up.stagedRecords;
        SPX_HIP(launchUpdateChain(cp, s));
        return SPX_OK;
    }
    SPX_HIP(launchUpdate(up, updateGrid(ctx, split ? 2 * n : n), split, streamAcc, s));
    FtParams fp{};
    fp.positions = up.childPositions;
    fp.nPositions = uint32_t(n);
    fp.order = up.refreshList;
    fp.nPerspPtr = up.refreshCount;
    fp.clearWord = counters + (ctx->refreshCur ^ 1);
    fp.t = up.t;
    fp.ftOut = up.ftOut;
    fp.accOut = up.childSlots ? up.arena : nullptr;  // eval-only children: rebuilt perspectives leave through ftOut alone
    fp.slots = up.childSlots;
    fp.slotRecords = up.slotRecords;
    ctx->refreshCur ^= 1;
    // one wave per deferred perspective: ~n / 15 of them in play, grid-stride beyond. (One WORKGROUP per perspective - the team
    // kernel - was measured here and loses: 4 456 items are more than the resident workgroups, so the pass runs in rounds and
    // pays table staging and list building four times over: update + rebuild 0.291 -> 0.319 ms per 65 536-record ply,
    // profiles/r03_ab_rebuild_pass_team_kernel.txt.)
    // Grid: 4 096 waves at most, grid-stride beyond (option refresh_waves overrides). Round 2 launched n / 4 waves - up to the
    // full 12 288-workgroup grid for self-play's capacity-sized batches; measured now (profiles/r03_ab_rebuild_pass_grid.txt):
    // 16 384 / 8 192 / 4 096 / 2 048 / 1 024 waves = self-play at 4 096 seats 2.39 / 2.39 / 2.41 / 2.43 / 2.37 x 10^8 and the
    // incremental bench's update + rebuild 0.290 / 0.291 / 0.290 / 0.297 / 0.320 ms: 4 096 suits both.
    const size_t waves = ctx->refreshWaves ? ctx->refreshWaves : std::min<size_t>(4096, std::max<size_t>(256, n / 4));
    SPX_HIP(launchFt(fp, ftGrid(ctx, waves), s));
    return SPX_OK;
}

int spx_acc_update_eval_device(spx_ctx* ctx, const void* d_parent_slots, const void* d_child_slots,
                               const void* d_child_positions, size_t n, void* d_out, void* stream) {
    return updateEvalDevice(ctx, d_parent_slots, d_child_slots, d_child_positions, n, nullptr, d_out, stream,
                            "spx_acc_update_eval_device");
}

int spx_acc_update_eval_device_counted(spx_ctx* ctx, const void* d_parent_slots, const void* d_child_slots,
                                       const void* d_child_positions, const void* d_count, size_t capacity, void* d_out,
                                       void* stream) {
    if (!d_count) {
        setError("spx_acc_update_eval_device_counted: null count");
        return SPX_ERR_INVALID_ARG;
    }
    return updateEvalDevice(ctx, d_parent_slots, d_child_slots, d_child_positions, capacity,
                            static_cast<const uint32_t*>(d_count), d_out, stream, "spx_acc_update_eval_device_counted");
}

static int updateEvalDevice(spx_ctx* ctx, const void* d_parent_slots, const void* d_child_slots,
                            const void* d_child_positions, size_t n, const uint32_t* d_count, void* d_out, void* stream,
                            const char* who) {
    int rc = checkAcc(ctx, n, who);
    if (rc != SPX_OK || n == 0) return rc;
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    UpdateParams up{};
    up.nRecords = uint32_t(n);
    up.nRecordsPtr = d_count;
    up.parentSlots = static_cast<const uint32_t*>(d_parent_slots);
    up.childSlots = static_cast<const uint32_t*>(d_child_slots);
    up.childPositions = d_child_positions;
    up.t = tablesOf(ctx);
    up.arena = ctx->dArena;
    up.slotRecords = ctx->dSlotRecords;
    up.ftOut = ctx->dFtOut;          // activations of the children straight from the update kernel's registers
    up.stagedRecords = ctx->dStaged;
    if (ctx->ftGateWait) SPX_HIP(hipStreamWaitEvent(s, ctx->ftGateWait, 0));  // lanes: the big kernels are chained
    // spx_profile_*: the "ft" interval is the update kernel + its rebuild pass, the "mlp" interval the sort + MLP
    hipEvent_t* ev = nullptr;
    if (ctx->profUsed + kProfEventsPerCall <= ctx->profEvents.size()) {
        ev = &ctx->profEvents[ctx->profUsed];
        ctx->profUsed += kProfEventsPerCall;
        SPX_HIP(hipEventRecord(ev[0], s));
        SPX_HIP(hipEventRecord(ev[1], s));
        SPX_HIP(hipEventRecord(ev[4], s));
    }
    rc = launchUpdateAndRefresh(ctx, up, n, s);
    if (rc != SPX_OK) return rc;
    if (ctx->ftGateRecord) SPX_HIP(hipEventRecord(ctx->ftGateRecord, s));
    if (ev) SPX_HIP(hipEventRecord(ev[2], s));
    if (!d_count && n <= ctx->tinyBatchMax) {
        rc = runTinyMlp(ctx, ctx->dStaged, n, d_out, s);
    } else {
        rc = runSortAndMlp(ctx, ctx->dStaged, n, nullptr, s, false, d_count, true);
        if (rc != SPX_OK) return rc;
        rc = runSortAndMlp(ctx, ctx->dStaged, n, d_out, s, true, d_count);
    }
    if (rc != SPX_OK) return rc;
    if (ev) SPX_HIP(hipEventRecord(ev[3], s));
    return SPX_OK;
}

// ---- BASELINE config 3 natively: a recorded make/unmake tree replayed level by level on device-resident buffers ----
int spx_acc_replay_tree(spx_ctx* ctx, const spx_packed_pos* positions, const uint32_t* parents, size_t n_nodes,
                        const uint32_t* eval_nodes, size_t n_evals, int32_t* out, double* gpu_ms) {
    if (!ctx || !positions || !parents || n_nodes == 0 || (n_evals && (!eval_nodes || !out)) || n_nodes > (1ull << 31)) {
        setError("spx_acc_replay_tree: invalid argument");
        return SPX_ERR_INVALID_ARG;
    }
    // levels: nodes are numbered in visiting order (a parent before its children), so depths come in one pass; a
    // counting sort by depth gives the level-ordered (parent slot, child slot, child record) batches
    std::vector<uint32_t> depth(n_nodes, 0), levelStart;
    uint32_t maxDepth = 0;
    for (size_t k = 1; k < n_nodes; ++k) {
        if (parents[k] >= k) {
            setError("spx_acc_replay_tree: node " + std::to_string(k) + " does not come after its parent");
            return SPX_ERR_INVALID_ARG;
        }
        depth[k] = depth[parents[k]] + 1;
        maxDepth = std::max(maxDepth, depth[k]);
    }
    for (size_t k = 0; k < n_evals; ++k) {
        if (eval_nodes[k] >= n_nodes) {
            setError("spx_acc_replay_tree: eval node out of range");
            return SPX_ERR_INVALID_ARG;
        }
    }
    levelStart.assign(maxDepth + 2, 0);
    for (size_t k = 1; k < n_nodes; ++k) ++levelStart[depth[k] + 1];
    for (uint32_t d = 1; d <= maxDepth + 1; ++d) levelStart[d] += levelStart[d - 1];  // levelStart[d] = first index of depth d
    const size_t nUpdates = n_nodes - 1;
    std::vector<uint32_t> hParents(nUpdates), hChildren(nUpdates), cursor(levelStart.begin(), levelStart.end() - 1);
    std::vector<spx_packed_pos> hRecords(nUpdates);
    // DEEP, NARROW trees (the reference's own alpha-beta search on a noisy net dives 249 plies with a few hundred nodes per
    // level) are bound by one dependent launch per level (~15 us each). They are walked by PATHS instead: the tree is cut into
    // heavy paths (every node continues into its largest subtree; the other children start paths of their own), a path is one
    // wavefront pair of spx_update_chain_kernel - the accumulator stays in registers from ply to ply, ~5.5 us per ply - and all
    // paths whose head's parent exists form one launch: 8 launches instead of 249 for that trace. Shallow wide trees (a
    // depth-first walk to depth 12) keep the level batches, whose big launches run at the update kernel's full rate.
    // option replay_paths = 0 / 1 forces the choice.
    bool byPaths = maxDepth >= 32 && nUpdates / std::max<uint32_t>(1, maxDepth) <= 4096;
    if (ctx->replayPaths >= 0) byPaths = ctx->replayPaths != 0;
    std::vector<uint32_t> hChainFirst, hChainCount, roundStart;  // paths mode: chains grouped by round; hParents = per chain
    std::vector<uint32_t> rebuilt;  // paths mode: nodes materialised from scratch beside the root (heads of the later segments of long paths)
    if (!byPaths) {
        for (size_t k = 1; k < n_nodes; ++k) {
            const uint32_t at = cursor[depth[k]]++;  // levelStart[1] == 0: depth-1 nodes come first
            hParents[at] = parents[k];
            hChildren[at] = uint32_t(k);
            hRecords[at] = positions[k];
        }
    } else {
        std::vector<uint32_t> size(n_nodes, 1), heavy(n_nodes, 0), chainOf(n_nodes, 0), roundOf(n_nodes, 0);
        for (size_t k = n_nodes - 1; k >= 1; --k) size[parents[k]] += size[k];
        for (size_t k = 1; k < n_nodes; ++k) {  // heavy[p] = the first child with the largest subtree
            const uint32_t p = parents[k];
            if (heavy[p] == 0 || size[k] > size[heavy[p]]) heavy[p] = uint32_t(k);
        }
        std::vector<uint32_t> chainParent, chainLen, chainRound;
        std::vector<uint8_t> isRebuilt(n_nodes, 0);
        uint32_t maxRound = 0;
        // Long paths are cut into SEGMENTS of `segment` plies that run SIDE BY SIDE in their path's round: the node a later
        // segment starts from is not waited for but REBUILT FROM SCRATCH in the launch that materialises the root (a full
        // refresh gives the same accumulator bit for bit, as it does for the reference on a king-bucket change) - 328 of the
        // 84 067 nodes of the reference's own depth-12 search at 8 plies per segment. A launch lasts as long as its longest
        // path: 249 + 26 + 15 + ... = 336 dependent plies become 8 + 8 + 8 + ... = 54 (round 4 started a spine's segments one
        // round after the other: 311): 1.61 -> 0.80 ms on that trace (tools/gpu_replay_segment_ab.py: 4 / 8 / 16 / 32 plies
        // 0.76 / 0.80 / 0.85 / 0.94 ms; shorter segments rebuild more nodes - 3 022 at 4).
        const uint32_t segment = ctx->replaySegment;
        for (size_t k = 1; k < n_nodes; ++k) {
            const uint32_t p = parents[k];
            const bool continues = p != 0 && heavy[p] == k;
            if (continues && (chainLen[chainOf[p]] < segment || heavy[k] == 0)) {  // goes on along its parent's path (a leaf never starts a segment)
                chainOf[k] = chainOf[p];
                roundOf[k] = roundOf[p];
                ++chainLen[chainOf[k]];
            } else if (continues) {         // a full segment lies behind: k is rebuilt and starts the next one, in the same round
                isRebuilt[k] = 1;
                rebuilt.push_back(uint32_t(k));
                chainOf[k] = uint32_t(chainParent.size());
                roundOf[k] = roundOf[p];
                chainParent.push_back(uint32_t(k));
                chainLen.push_back(0);
                chainRound.push_back(roundOf[k]);
            } else {                        // heads a path of its own, one round after the path its parent is on
                chainOf[k] = uint32_t(chainParent.size());
                roundOf[k] = roundOf[p] + 1;
                chainParent.push_back(p);
                chainLen.push_back(1);
                chainRound.push_back(roundOf[k]);
                maxRound = std::max(maxRound, roundOf[k]);
            }
        }
        const size_t nChains = chainParent.size();
        roundStart.assign(maxRound + 2, 0);  // rounds are 1-based: roundStart[r] = first chain of round r
        for (size_t c = 0; c < nChains; ++c) ++roundStart[chainRound[c] + 1];
        for (uint32_t r = 1; r <= maxRound + 1; ++r) roundStart[r] += roundStart[r - 1];
        std::vector<uint32_t> place(nChains), next(roundStart.begin(), roundStart.end() - 1);
        for (size_t c = 0; c < nChains; ++c) place[c] = next[chainRound[c]]++;
        hChainFirst.assign(nChains, 0);
        hChainCount.assign(nChains, 0);
        hParents.assign(nChains, 0);  // (per chain here: the slot the path starts from)
        {
            std::vector<uint32_t> lenAt(nChains);
            for (size_t c = 0; c < nChains; ++c) lenAt[place[c]] = chainLen[c];
            uint32_t at = 0;
            for (size_t j = 0; j < nChains; ++j) {
                hChainFirst[j] = at;
                at += lenAt[j];
            }
        }
        for (size_t c = 0; c < nChains; ++c) hParents[place[c]] = chainParent[c];
        for (size_t k = 1; k < n_nodes; ++k) {  // visiting order = increasing depth along every path
            if (isRebuilt[k]) continue;
            const uint32_t j = place[chainOf[k]], at = hChainFirst[j] + hChainCount[j]++;
            hChildren[at] = uint32_t(k);
            hRecords[at] = positions[k];
        }
    }
    int rc = spx_acc_reserve(ctx, n_nodes);
    if (rc != SPX_OK) return rc;
    SPX_HIP(hipSetDevice(ctx->device));
    struct Temp {
        std::vector<void*> ptrs;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Temp() {
            for (void* p : ptrs) (void)hipFree(p);
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } temp;
    auto alloc = [&](size_t bytes, void** out) -> hipError_t {
        const hipError_t e = hipMalloc(out, std::max<size_t>(bytes, 16));
        if (e == hipSuccess) temp.ptrs.push_back(*out);
        return e;
    };
    void *dRecords = nullptr, *dParents = nullptr, *dChildren = nullptr, *dEvalNodes = nullptr, *dOut = nullptr, *dRoot = nullptr;
    SPX_HIP(alloc(nUpdates * sizeof(spx_packed_pos), &dRecords));
    SPX_HIP(alloc(nUpdates * 4, &dParents));
    SPX_HIP(alloc(nUpdates * 4, &dChildren));
    void *dChainFirst = nullptr, *dChainCount = nullptr;
    SPX_HIP(alloc(hChainFirst.size() * 4, &dChainFirst));
    SPX_HIP(alloc(hChainCount.size() * 4, &dChainCount));
    SPX_HIP(alloc(n_evals * 4, &dEvalNodes));
    SPX_HIP(alloc(n_evals * 4, &dOut));
    // the nodes built from scratch: the root and the heads of the later segments of long paths (records, then slots = node ids)
    const size_t nScratch = 1 + rebuilt.size();
    if (nScratch > ctx->maxBatch) {
        setError("spx_acc_replay_tree: more path segments than the context's batch capacity (raise option replay_segment)");
        return SPX_ERR_CAPACITY;
    }
    std::vector<spx_packed_pos> hScratchPos(nScratch);
    std::vector<uint32_t> hScratchSlots(nScratch);
    hScratchPos[0] = positions[0];
    hScratchSlots[0] = 0;
    for (size_t i = 0; i < rebuilt.size(); ++i) {
        hScratchPos[1 + i] = positions[rebuilt[i]];
        hScratchSlots[1 + i] = rebuilt[i];
    }
    SPX_HIP(alloc(nScratch * (sizeof(spx_packed_pos) + 4), &dRoot));
    hipStream_t s = ctx->stream;
    char* dScratchSlots = static_cast<char*>(dRoot) + nScratch * sizeof(spx_packed_pos);
    SPX_HIP(hipMemcpyAsync(dRoot, hScratchPos.data(), nScratch * sizeof(spx_packed_pos), hipMemcpyHostToDevice, s));
    SPX_HIP(hipMemcpyAsync(dScratchSlots, hScratchSlots.data(), nScratch * 4, hipMemcpyHostToDevice, s));
    if (nUpdates) {
        SPX_HIP(hipMemcpyAsync(dRecords, hRecords.data(), nUpdates * sizeof(spx_packed_pos), hipMemcpyHostToDevice, s));
        SPX_HIP(hipMemcpyAsync(dParents, hParents.data(), hParents.size() * 4, hipMemcpyHostToDevice, s));
        SPX_HIP(hipMemcpyAsync(dChildren, hChildren.data(), nUpdates * 4, hipMemcpyHostToDevice, s));
        if (byPaths) {
            SPX_HIP(hipMemcpyAsync(dChainFirst, hChainFirst.data(), hChainFirst.size() * 4, hipMemcpyHostToDevice, s));
            SPX_HIP(hipMemcpyAsync(dChainCount, hChainCount.data(), hChainCount.size() * 4, hipMemcpyHostToDevice, s));
        }
    }
    if (n_evals) SPX_HIP(hipMemcpyAsync(dEvalNodes, eval_nodes, n_evals * 4, hipMemcpyHostToDevice, s));
    SPX_HIP(hipEventCreate(&temp.e0));
    SPX_HIP(hipEventCreate(&temp.e1));
    SPX_HIP(hipEventRecord(temp.e0, s));
    rc = spx_acc_refresh_device(ctx, dRoot, dScratchSlots, nScratch, s);
    if (rc != SPX_OK) return rc;
    if (byPaths) {  // one chain launch per round: every path whose head's parent has been materialised
        for (size_t r = 1; r + 1 < roundStart.size(); ++r) {
            const uint32_t lo = roundStart[r], hi = roundStart[r + 1];
            if (hi == lo) continue;
            ChainParams cp{};
            cp.nChains = hi - lo;
            cp.parentSlots = static_cast<const uint32_t*>(dParents) + lo;
            cp.first = static_cast<const uint32_t*>(dChainFirst) + lo;
            cp.count = static_cast<const uint32_t*>(dChainCount) + lo;
            cp.childSlots = static_cast<const uint32_t*>(dChildren);
            cp.childPositions = dRecords;
            cp.t = tablesOf(ctx);
            cp.arena = ctx->dArena;
            cp.slotRecords = ctx->dSlotRecords;
            SPX_HIP(launchUpdateChain(cp, s));
        }
    }
    for (uint32_t d = 1; d <= maxDepth && !byPaths; ++d) {  // one update batch per level (chunked by the context's capacity); no host sync
        for (size_t lo = levelStart[d]; lo < levelStart[d + 1]; lo += ctx->maxBatch) {
            const size_t m = std::min<size_t>(ctx->maxBatch, levelStart[d + 1] - lo);
            rc = spx_acc_update_device(ctx, static_cast<char*>(dParents) + lo * 4, static_cast<char*>(dChildren) + lo * 4,
                                       static_cast<char*>(dRecords) + lo * sizeof(spx_packed_pos), m, s);
            if (rc != SPX_OK) return rc;
        }
    }
    for (size_t lo = 0; lo < n_evals; lo += ctx->maxBatch) {
        const size_t m = std::min<size_t>(ctx->maxBatch, n_evals - lo);
        rc = spx_acc_eval_device(ctx, static_cast<char*>(dEvalNodes) + lo * 4, m, static_cast<char*>(dOut) + lo * 4, s);
        if (rc != SPX_OK) return rc;
    }
    SPX_HIP(hipEventRecord(temp.e1, s));
    if (n_evals) SPX_HIP(hipMemcpyAsync(out, dOut, n_evals * 4, hipMemcpyDeviceToHost, s));
    SPX_HIP(hipStreamSynchronize(s));
    if (gpu_ms) {
        float ms = 0.f;
        SPX_HIP(hipEventElapsedTime(&ms, temp.e0, temp.e1));
        *gpu_ms = ms;
    }
    return SPX_OK;
}

// Pipelined plies: consecutive calls alternate the context's two lanes (scratch sets + streams). The update kernels (and
// their rebuild passes) run in call order, chained by events - a ply's parents are the previous ply's children - while
// the output-bucket sort and the MLP of one ply run beside the update kernel of the next.
int spx_acc_update_eval_device_async(spx_ctx* ctx, const void* d_parent_slots, const void* d_child_slots,
                                     const void* d_child_positions, size_t n, void* d_out, void** done_event) {
    int rc = checkAcc(ctx, n, "spx_acc_update_eval_device_async");
    if (rc != SPX_OK) return rc;
    rc = ctx->lanesUnavailable ? SPX_ERR_HIP : ensureLanes(ctx);
    if (rc != SPX_OK) {  // no room for the second scratch set: same results, stream-ordered on the context's own stream
        if (!ctx->lanesUnavailable) {
            (void)hipGetLastError();
            releaseLanes(ctx);
            ctx->lanesUnavailable = true;
            if (!ctx->fallbackDone) SPX_HIP(hipEventCreateWithFlags(&ctx->fallbackDone, hipEventDisableTiming));
        }
        rc = updateEvalDevice(ctx, d_parent_slots, d_child_slots, d_child_positions, n, nullptr, d_out, ctx->stream,
                              "spx_acc_update_eval_device_async");
        if (rc != SPX_OK) return rc;
        SPX_HIP(hipEventRecord(ctx->fallbackDone, ctx->stream));
        if (done_event) *done_event = ctx->fallbackDone;
        return SPX_OK;
    }
    spx_ctx::EvalLane& lane = ctx->lanes[ctx->laneNext & 1];
    spx_ctx::EvalLane& other = ctx->lanes[(ctx->laneNext & 1) ^ 1];
    ++ctx->laneNext;
    swapLane(ctx, lane);
    ctx->ftGateWait = other.ftRecorded ? other.ftDone : nullptr;
    ctx->ftGateRecord = lane.ftDone;
    rc = updateEvalDevice(ctx, d_parent_slots, d_child_slots, d_child_positions, n, nullptr, d_out, lane.stream,
                          "spx_acc_update_eval_device_async");
    ctx->ftGateWait = ctx->ftGateRecord = nullptr;
    swapLane(ctx, lane);
    if (rc != SPX_OK) return rc;
    if (n) lane.ftRecorded = true;
    SPX_HIP(hipEventRecord(lane.done, lane.stream));
    if (done_event) *done_event = lane.done;
    return SPX_OK;
}

static_assert(sizeof(spx_move_delta) == 1080, "spx_update_observed_kernel hard-codes the spx_move_delta layout");

int spx_acc_update_observed_device(spx_ctx* ctx, const void* d_parent_slots, const void* d_child_slots,
                                   const void* d_child_positions, const void* d_deltas, size_t n, void* d_out,
                                   void* stream) {
    int rc = checkAcc(ctx, n, "spx_acc_update_observed_device");
    if (rc != SPX_OK || n == 0) return rc;
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    UpdateParams up{};
    up.nRecords = uint32_t(n);
    up.parentSlots = static_cast<const uint32_t*>(d_parent_slots);
    up.childSlots = static_cast<const uint32_t*>(d_child_slots);
    up.childPositions = d_child_positions;
    up.t = tablesOf(ctx);
    up.arena = ctx->dArena;
    up.slotRecords = ctx->dSlotRecords;
    up.deltas = static_cast<const uint8_t*>(d_deltas);
    if (d_out) {
        up.ftOut = ctx->dFtOut;
        up.stagedRecords = ctx->dStaged;
    }
    SPX_HIP(launchUpdateObserved(up, ftGrid(ctx, 2 * n), s));  // one wave per (record, perspective)
    if (!d_out) return SPX_OK;
    rc = runSortAndMlp(ctx, ctx->dStaged, n, nullptr, s, false, nullptr, true);
    if (rc != SPX_OK) return rc;
    return runSortAndMlp(ctx, ctx->dStaged, n, d_out, s, true);
}

int spx_acc_update_observed(spx_ctx* ctx, const uint32_t* parent_slots, const uint32_t* child_slots,
                            const spx_packed_pos* child_positions, const spx_move_delta* deltas, size_t n,
                            int32_t* out) {
    int rc = checkAcc(ctx, n, "spx_acc_update_observed");
    if (rc != SPX_OK || n == 0) return rc;
    if (!parent_slots || !child_slots || !child_positions || !deltas) {
        setError("spx_acc_update_observed: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    if ((rc = checkSlots(ctx, parent_slots, n, "spx_acc_update_observed")) != SPX_OK) return rc;
    if ((rc = checkSlots(ctx, child_slots, n, "spx_acc_update_observed")) != SPX_OK) return rc;
    SPX_HIP(hipSetDevice(ctx->device));
    if (!ctx->dDeltas) {
        SPX_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->dDeltas), ctx->maxBatch * sizeof(spx_move_delta)));
    }
    SPX_HIP(hipMemcpyAsync(ctx->dPositions, child_positions, n * sizeof(spx_packed_pos), hipMemcpyHostToDevice,
                           ctx->stream));
    SPX_HIP(hipMemcpyAsync(ctx->dSlotsA, parent_slots, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    SPX_HIP(hipMemcpyAsync(ctx->dSlotsB, child_slots, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    SPX_HIP(hipMemcpyAsync(ctx->dDeltas, deltas, n * sizeof(spx_move_delta), hipMemcpyHostToDevice, ctx->stream));
    rc = spx_acc_update_observed_device(ctx, ctx->dSlotsA, ctx->dSlotsB, ctx->dPositions, ctx->dDeltas, n,
                                        out ? ctx->dOut : nullptr, ctx->stream);
    if (rc != SPX_OK) return rc;
    if (out) SPX_HIP(hipMemcpyAsync(out, ctx->dOut, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SPX_HIP(hipStreamSynchronize(ctx->stream));
    return SPX_OK;
}

int spx_acc_eval_device(spx_ctx* ctx, const void* d_slots, size_t n, void* d_out, void* stream) {
    int rc = checkAcc(ctx, n, "spx_acc_eval_device");
    if (rc != SPX_OK || n == 0) return rc;
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    SlotActParams ap{};
    ap.nSlots = uint32_t(n);
    ap.slots = static_cast<const uint32_t*>(d_slots);
    ap.arena = ctx->dArena;
    ap.slotRecords = ctx->dSlotRecords;
    ap.ftOut = ctx->dFtOut;
    ap.stagedRecords = ctx->dStaged;
    uint32_t blocks = uint32_t((n + 3) / 4);
    if (blocks > ctx->ftGridCap) blocks = ctx->ftGridCap;
    SPX_HIP(launchSlotAct(ap, blocks, s));
    if (n <= ctx->tinyBatchMax) return runTinyMlp(ctx, ctx->dStaged, n, d_out, s);
    rc = runSortAndMlp(ctx, ctx->dStaged, n, nullptr, s, false, nullptr, true);
    if (rc != SPX_OK) return rc;
    return runSortAndMlp(ctx, ctx->dStaged, n, d_out, s, true);
}

static int checkSlots(const spx_ctx* ctx, const uint32_t* slots, size_t n, const char* who) {
    for (size_t i = 0; i < n; ++i) {
        if (slots[i] >= ctx->nSlots) {
            setError(std::string(who) + ": slot " + std::to_string(slots[i]) + " out of range (reserved " +
                     std::to_string(ctx->nSlots) + ")");
            return SPX_ERR_INVALID_ARG;
        }
    }
    return SPX_OK;
}

int spx_acc_refresh(spx_ctx* ctx, const spx_packed_pos* positions, const uint32_t* slots, size_t n) {
    int rc = checkAcc(ctx, n, "spx_acc_refresh");
    if (rc != SPX_OK || n == 0) return rc;
    if (!positions || !slots) {
        setError("spx_acc_refresh: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    if ((rc = checkSlots(ctx, slots, n, "spx_acc_refresh")) != SPX_OK) return rc;
    SPX_HIP(hipSetDevice(ctx->device));
    SPX_HIP(hipMemcpyAsync(ctx->dPositions, positions, n * sizeof(spx_packed_pos), hipMemcpyHostToDevice, ctx->stream));
    SPX_HIP(hipMemcpyAsync(ctx->dSlotsA, slots, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    rc = spx_acc_refresh_device(ctx, ctx->dPositions, ctx->dSlotsA, n, ctx->stream);
    if (rc != SPX_OK) return rc;
    SPX_HIP(hipStreamSynchronize(ctx->stream));
    return SPX_OK;
}

int spx_acc_update(spx_ctx* ctx, const uint32_t* parent_slots, const uint32_t* child_slots,
                   const spx_packed_pos* child_positions, size_t n) {
    int rc = checkAcc(ctx, n, "spx_acc_update");
    if (rc != SPX_OK || n == 0) return rc;
    if (!parent_slots || !child_slots || !child_positions) {
        setError("spx_acc_update: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    if ((rc = checkSlots(ctx, parent_slots, n, "spx_acc_update")) != SPX_OK) return rc;
    if ((rc = checkSlots(ctx, child_slots, n, "spx_acc_update")) != SPX_OK) return rc;
    SPX_HIP(hipSetDevice(ctx->device));
    SPX_HIP(hipMemcpyAsync(ctx->dPositions, child_positions, n * sizeof(spx_packed_pos), hipMemcpyHostToDevice,
                           ctx->stream));
    SPX_HIP(hipMemcpyAsync(ctx->dSlotsA, parent_slots, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    SPX_HIP(hipMemcpyAsync(ctx->dSlotsB, child_slots, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    rc = spx_acc_update_device(ctx, ctx->dSlotsA, ctx->dSlotsB, ctx->dPositions, n, ctx->stream);
    if (rc != SPX_OK) return rc;
    SPX_HIP(hipStreamSynchronize(ctx->stream));
    return SPX_OK;
}

int spx_acc_update_eval(spx_ctx* ctx, const uint32_t* parent_slots, const uint32_t* child_slots,
                        const spx_packed_pos* child_positions, size_t n, int32_t* out) {
    int rc = checkAcc(ctx, n, "spx_acc_update_eval");
    if (rc != SPX_OK || n == 0) return rc;
    if (!parent_slots || !child_positions || !out) {  // child_slots == NULL: eval-only children (nothing is stored)
        setError("spx_acc_update_eval: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    if ((rc = checkSlots(ctx, parent_slots, n, "spx_acc_update_eval")) != SPX_OK) return rc;
    if (child_slots && (rc = checkSlots(ctx, child_slots, n, "spx_acc_update_eval")) != SPX_OK) return rc;
    SPX_HIP(hipSetDevice(ctx->device));
    if (n <= ctx->tinyBatchMax && n <= kTinyIoRecords) {
        // push + evaluate of a few nodes (the search's own step): all operands through device-mapped page-locked memory
        auto* records = static_cast<spx_packed_pos*>(ctx->hTinyIo);
        auto* scores = reinterpret_cast<int32_t*>(records + kTinyIoRecords);
        uint32_t* parents = reinterpret_cast<uint32_t*>(scores + kTinyIoRecords);
        uint32_t* children = parents + kTinyIoRecords;
        std::memcpy(records, child_positions, n * sizeof(spx_packed_pos));
        std::memcpy(parents, parent_slots, n * sizeof(uint32_t));
        if (child_slots) std::memcpy(children, child_slots, n * sizeof(uint32_t));
        char* dBase = nullptr;
        SPX_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&dBase), ctx->hTinyIo, 0));
        const size_t offScores = kTinyIoRecords * sizeof(spx_packed_pos), offParents = offScores + kTinyIoRecords * 4,
                     offChildren = offParents + kTinyIoRecords * 4;
        rc = spx_acc_update_eval_device(ctx, dBase + offParents, child_slots ? dBase + offChildren : nullptr, dBase, n,
                                        dBase + offScores, ctx->stream);
        if (rc != SPX_OK) return rc;
        SPX_HIP(hipStreamSynchronize(ctx->stream));
        std::memcpy(out, scores, n * sizeof(int32_t));
        return SPX_OK;
    }
    SPX_HIP(hipMemcpyAsync(ctx->dPositions, child_positions, n * sizeof(spx_packed_pos), hipMemcpyHostToDevice,
                           ctx->stream));
    SPX_HIP(hipMemcpyAsync(ctx->dSlotsA, parent_slots, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    if (child_slots) SPX_HIP(hipMemcpyAsync(ctx->dSlotsB, child_slots, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    rc = spx_acc_update_eval_device(ctx, ctx->dSlotsA, child_slots ? ctx->dSlotsB : nullptr, ctx->dPositions, n, ctx->dOut,
                                    ctx->stream);
    if (rc != SPX_OK) return rc;
    SPX_HIP(hipMemcpyAsync(out, ctx->dOut, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SPX_HIP(hipStreamSynchronize(ctx->stream));
    return SPX_OK;
}

// NnueState::ensureUpToDate over a whole pending path + evaluate of its last position: ONE launch of the chain kernel (the
// accumulator stays in registers from ply to ply), one tiny MLP launch, one synchronisation.
int spx_acc_update_chain_eval(spx_ctx* ctx, uint32_t parent_slot, const uint32_t* child_slots,
                              const spx_packed_pos* child_positions, size_t n, int32_t* out) {
    int rc = checkAcc(ctx, n, "spx_acc_update_chain_eval");
    if (rc != SPX_OK || n == 0) return rc;
    if (!child_slots || !child_positions || n > kTinyIoRecords) {
        setError("spx_acc_update_chain_eval: null argument or a path longer than 8192 plies");
        return SPX_ERR_INVALID_ARG;
    }
    if ((rc = checkSlots(ctx, &parent_slot, 1, "spx_acc_update_chain_eval")) != SPX_OK) return rc;
    if ((rc = checkSlots(ctx, child_slots, n, "spx_acc_update_chain_eval")) != SPX_OK) return rc;
    // operands through the device-mapped page-locked staging buffer (as the other latency-bound host calls)
    auto* records = static_cast<spx_packed_pos*>(ctx->hTinyIo);
    auto* scores = reinterpret_cast<int32_t*>(records + kTinyIoRecords);
    uint32_t* head = reinterpret_cast<uint32_t*>(scores + kTinyIoRecords);  // [0] parent slot, [1] first, [2] count
    uint32_t* children = head + kTinyIoRecords;
    std::memcpy(records, child_positions, n * sizeof(spx_packed_pos));
    std::memcpy(children, child_slots, n * sizeof(uint32_t));
    head[0] = parent_slot;
    head[1] = 0;
    head[2] = uint32_t(n);
    char* dBase = nullptr;
    SPX_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&dBase), ctx->hTinyIo, 0));
    const size_t offScores = kTinyIoRecords * sizeof(spx_packed_pos), offHead = offScores + kTinyIoRecords * 4,
                 offChildren = offHead + kTinyIoRecords * 4;
    ChainParams cp{};
    cp.nChains = 1;
    cp.parentSlots = reinterpret_cast<const uint32_t*>(dBase + offHead);
    cp.first = cp.parentSlots + 1;
    cp.count = cp.parentSlots + 2;
    cp.childSlots = reinterpret_cast<const uint32_t*>(dBase + offChildren);
    cp.childPositions = dBase;
    cp.t = tablesOf(ctx);
    cp.arena = ctx->dArena;
    cp.slotRecords = ctx->dSlotRecords;
    if (out) {
        cp.ftOut = ctx->dFtOut;
        cp.stagedRecords = ctx->dStaged;
    }
    SPX_HIP(launchUpdateChain(cp, ctx->stream));
    if (out) {
        rc = runTinyMlp(ctx, ctx->dStaged, 1, dBase + offScores, ctx->stream);
        if (rc != SPX_OK) return rc;
    }
    SPX_HIP(hipStreamSynchronize(ctx->stream));
    if (out) *out = scores[0];
    return SPX_OK;
}

int spx_acc_eval(spx_ctx* ctx, const uint32_t* slots, size_t n, int32_t* out) {
    int rc = checkAcc(ctx, n, "spx_acc_eval");
    if (rc != SPX_OK || n == 0) return rc;
    if (!slots || !out) {
        setError("spx_acc_eval: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    if ((rc = checkSlots(ctx, slots, n, "spx_acc_eval")) != SPX_OK) return rc;
    SPX_HIP(hipSetDevice(ctx->device));
    if (n <= ctx->tinyBatchMax && n <= kTinyIoRecords) {  // a few nodes: operands through device-mapped page-locked memory
        auto* scores = reinterpret_cast<int32_t*>(static_cast<spx_packed_pos*>(ctx->hTinyIo) + kTinyIoRecords);
        uint32_t* hSlots = reinterpret_cast<uint32_t*>(scores + kTinyIoRecords);
        std::memcpy(hSlots, slots, n * sizeof(uint32_t));
        char* dBase = nullptr;
        SPX_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&dBase), ctx->hTinyIo, 0));
        const size_t offScores = kTinyIoRecords * sizeof(spx_packed_pos), offSlots = offScores + kTinyIoRecords * 4;
        rc = spx_acc_eval_device(ctx, dBase + offSlots, n, dBase + offScores, ctx->stream);
        if (rc != SPX_OK) return rc;
        SPX_HIP(hipStreamSynchronize(ctx->stream));
        std::memcpy(out, scores, n * sizeof(int32_t));
        return SPX_OK;
    }
    SPX_HIP(hipMemcpyAsync(ctx->dSlotsA, slots, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    rc = spx_acc_eval_device(ctx, ctx->dSlotsA, n, ctx->dOut, ctx->stream);
    if (rc != SPX_OK) return rc;
    SPX_HIP(hipMemcpyAsync(out, ctx->dOut, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SPX_HIP(hipStreamSynchronize(ctx->stream));
    return SPX_OK;
}

int spx_profile_begin(spx_ctx* ctx, size_t max_calls) {
    if (!ctx) {
        setError("spx_profile_begin: null context");
        return SPX_ERR_INVALID_ARG;
    }
    SPX_HIP(hipSetDevice(ctx->device));
    while (ctx->profEvents.size() < max_calls * kProfEventsPerCall) {
        hipEvent_t e;
        SPX_HIP(hipEventCreate(&e));
        ctx->profEvents.push_back(e);
    }
    ctx->profUsed = 0;
    return SPX_OK;
}

int spx_ctx_sliced_ft(const spx_ctx* ctx, size_t n) {
    if (!ctx || !ctx->ftxEnabled || ctx->ftxUnavailable || n <= ctx->tinyBatchMax) return 0;
    // (per chunk of a call; pipelined calls of a context whose lanes did not fit run stream-ordered: the same threshold then)
    const size_t pipelinedFrom = (ctx->ftxMinForced || ctx->lanesUnavailable) ? ctx->ftxMin : std::min(ctx->ftxMin, kFtxMinPositionsPipelined);
    return (n >= ctx->ftxMin ? 1 : 0) | (n >= pipelinedFrom ? 2 : 0);
}

int spx_ctx_calibrate(spx_ctx* ctx, const void* d_positions, size_t n) {
    if (!ctx || (n && !d_positions)) {
        setError("spx_ctx_calibrate: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    SPX_HIP(hipSetDevice(ctx->device));
    const size_t m = std::min({n, ctx->maxBatch, kFtxMaxPositions});
    if (!m || !ensureFtx(ctx, ctx->ftx, m, ctx->stream)) return SPX_OK;  // (no pipeline on this context: nothing to choose)
    FtxParams xp{};
    xp.positions = d_positions;
    xp.nPositions = uint32_t(m);
    xp.t = tablesOf(ctx);
    xp.rowS = ctx->dRowS;
    xp.lists = ctx->ftx.lists;
    xp.heads = ctx->ftx.heads;
    return calibrateHotRows(ctx, xp, ctx->stream);
}

int spx_ctx_set_hot_rows(spx_ctx* ctx, const uint32_t* rows, size_t n) {
    if (!ctx || (n && !rows) || n > kFtxHotRowsMax) {
        setError("spx_ctx_set_hot_rows: null argument or more than " + std::to_string(kFtxHotRowsMax) + " rows");
        return SPX_ERR_INVALID_ARG;
    }
    std::vector<uint32_t> ids(rows, rows + n);
    std::vector<uint32_t> sorted(ids);
    std::sort(sorted.begin(), sorted.end());
    if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end() || (n && sorted.back() >= kThreatRows)) {
        setError("spx_ctx_set_hot_rows: rows must be distinct threat / pawn-pair row ids below " + std::to_string(kThreatRows));
        return SPX_ERR_INVALID_ARG;
    }
    SPX_HIP(hipSetDevice(ctx->device));
    ctx->hotIds = ids;
    ctx->hotCalibrated = true;
    ctx->coldShift = n >= 128 ? 0u : 1u;
    if (!ctx->dRowS) return SPX_OK;  // (installed when the pipeline's tables are built)
    return installHotRows(ctx, ctx->hotIds, ctx->stream);
}

int spx_ctx_get_hot_rows(const spx_ctx* ctx, uint32_t* rows, size_t capacity, size_t* n) {
    if (!ctx || !n || (capacity && !rows)) {
        setError("spx_ctx_get_hot_rows: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    *n = ctx->hotCalibrated ? ctx->hotIds.size() : 0;
    for (size_t i = 0; i < std::min(capacity, *n); ++i) rows[i] = ctx->hotIds[i];
    return SPX_OK;
}

size_t spx_ctx_scratch_batch(const spx_ctx* ctx) {
    return ctx ? ctx->maxBatch : 0;
}

uint32_t spx_ctx_compact_psq_rows(const spx_ctx* ctx) {
    return ctx ? ctx->compactPsqRows : 0;
}

uint32_t spx_ctx_near_psq_rows(const spx_ctx* ctx) {
    return ctx ? ctx->nearPsqRows : 0;
}

int spx_profile_end(spx_ctx* ctx, double* sort_ms, double* ft_ms, double* mlp_ms, size_t* calls) {
    if (!ctx || !sort_ms || !ft_ms || !mlp_ms || !calls) {
        setError("spx_profile_end: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    SPX_HIP(hipSetDevice(ctx->device));
    *sort_ms = *ft_ms = *mlp_ms = 0.0;
    double prepare = 0.0;
    *calls = ctx->profUsed / kProfEventsPerCall;
    for (size_t i = 0; i + kProfEventsPerCall - 1 < ctx->profUsed; i += kProfEventsPerCall) {
        float a = 0.f, b = 0.f, c = 0.f;
        SPX_HIP(hipEventSynchronize(ctx->profEvents[i + 3]));
        SPX_HIP(hipEventElapsedTime(&a, ctx->profEvents[i], ctx->profEvents[i + 1]));
        SPX_HIP(hipEventElapsedTime(&b, ctx->profEvents[i + 4], ctx->profEvents[i + 2]));
        SPX_HIP(hipEventElapsedTime(&c, ctx->profEvents[i + 2], ctx->profEvents[i + 3]));
        *sort_ms += a;
        *ft_ms += b;
        *mlp_ms += c;
        SPX_HIP(hipEventElapsedTime(&a, ctx->profEvents[i + 1], ctx->profEvents[i + 4]));
        prepare += a;
    }
    ctx->profLastPrepareMs = prepare;
    ctx->profUsed = ctx->profEvents.size();  // stop recording until the next spx_profile_begin
    return SPX_OK;
}

int spx_profile_last_prepare_ms(const spx_ctx* ctx, double* prepare_ms) {
    if (!ctx || !prepare_ms) {
        setError("spx_profile_last_prepare_ms: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    *prepare_ms = ctx->profLastPrepareMs;
    return SPX_OK;
}

int spx_count_rows(const spx_packed_pos* positions, size_t n, uint64_t* psq_rows, uint64_t* threat_rows) {
    if ((n && !positions) || !psq_rows || !threat_rows) {
        setError("spx_count_rows: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    uint64_t np = 0, nt = 0;
    uint32_t psq[32], thr[256];
    for (size_t i = 0; i < n; ++i) {
        for (int c = 0; c < 2; ++c) {
            int a = 0, b = 0;
            const int rc = spx_debug_features(&positions[i], c, psq, &a, thr, &b);
            if (rc != SPX_OK) return rc;
            np += uint64_t(a);
            nt += uint64_t(b);
        }
    }
    *psq_rows = np;
    *threat_rows = nt;
    return SPX_OK;
}

int spx_ctx_count_rows(const spx_ctx* ctx, const spx_packed_pos* positions, size_t n, uint64_t* psq_wide_rows,
                       uint64_t* psq_compact_rows, uint64_t* threat_rows) {
    if (!ctx || (n && !positions) || !psq_wide_rows || !psq_compact_rows || !threat_rows) {
        setError("spx_ctx_count_rows: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    uint64_t nw = 0, nc = 0, nt = 0;
    uint32_t psq[32], thr[256];
    for (size_t i = 0; i < n; ++i) {
        for (int c = 0; c < 2; ++c) {
            int a = 0, b = 0;
            const int rc = spx_debug_features(&positions[i], c, psq, &a, thr, &b);
            if (rc != SPX_OK) return rc;
            for (int k = 0; k < a; ++k) {
                const bool compact = ((ctx->compactBits[psq[k] >> 5] | ctx->nearBits[psq[k] >> 5]) >> (psq[k] & 31)) & 1u;
                (compact ? nc : nw) += 1;
            }
            nt += uint64_t(b);
        }
    }
    *psq_wide_rows = nw;
    *psq_compact_rows = nc;
    *threat_rows = nt;
    return SPX_OK;
}

int spx_eval_full(spx_ctx* ctx, const spx_packed_pos* positions, size_t n, int32_t* out) {
    if (!ctx || (n && (!positions || !out))) {
        setError("spx_eval_full: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    if (n == 0) return SPX_OK;
    SPX_HIP(hipSetDevice(ctx->device));
    if (n <= ctx->tinyBatchMax && n <= kTinyIoRecords && n <= ctx->maxBatch) {
        // the one-position drop-in call (NnueState::evaluateOnce): no DMA transfers at all - the kernels read the records
        // from, and write the scores to, page-locked host memory mapped into the device's address space
        auto* records = static_cast<spx_packed_pos*>(ctx->hTinyIo);
        auto* scores = reinterpret_cast<int32_t*>(records + kTinyIoRecords);
        std::memcpy(records, positions, n * sizeof(spx_packed_pos));
        void* dRecords = nullptr;
        SPX_HIP(hipHostGetDevicePointer(&dRecords, records, 0));
        const int rc = spx_eval_full_device(ctx, dRecords, n, static_cast<char*>(dRecords) + kTinyIoRecords * sizeof(spx_packed_pos),
                                            ctx->stream);
        if (rc != SPX_OK) return rc;
        SPX_HIP(hipStreamSynchronize(ctx->stream));
        std::memcpy(out, scores, n * sizeof(int32_t));
        return SPX_OK;
    }
    if (n > ctx->maxBatch && !ctx->lanesUnavailable && ensureLanes(ctx) != SPX_OK) {
        (void)hipGetLastError();  // no room for the lanes: plain chunk loop below
        releaseLanes(ctx);
        ctx->lanesUnavailable = true;
    }
    if (n > ctx->maxBatch && !ctx->lanesUnavailable) {
        // More than one chunk of the context's capacity (rescoring a data set): the chunks alternate between the two
        // lanes - while one chunk is evaluated, the next is copied into page-locked staging and sent over PCIe and the
        // previous one's scores come back; FT kernels chained as in spx_eval_full_device_async.
        int rc = SPX_OK;
        for (auto& lane : ctx->lanes) {
            if (lane.dIn) continue;
            SPX_HIP(hipMalloc(&lane.dIn, ctx->maxBatch * sizeof(spx_packed_pos)));
            SPX_HIP(hipMalloc(reinterpret_cast<void**>(&lane.dOutStage), ctx->maxBatch * sizeof(int32_t)));
            SPX_HIP(hipHostMalloc(&lane.hIn, ctx->maxBatch * sizeof(spx_packed_pos), hipHostMallocDefault));
            SPX_HIP(hipHostMalloc(reinterpret_cast<void**>(&lane.hOut), ctx->maxBatch * sizeof(int32_t), hipHostMallocDefault));
        }
        struct Pending {
            size_t lo = 0, m = 0;
            bool active = false;
        } pending[2];
        auto drain = [&](int li) -> int {
            if (!pending[li].active) return SPX_OK;
            SPX_HIP(hipEventSynchronize(ctx->lanes[li].done));
            std::memcpy(out + pending[li].lo, ctx->lanes[li].hOut, pending[li].m * sizeof(int32_t));
            pending[li].active = false;
            return SPX_OK;
        };
        int li = 0;
        for (size_t lo = 0; lo < n; lo += ctx->maxBatch, li ^= 1) {
            const size_t m = std::min(ctx->maxBatch, n - lo);
            if ((rc = drain(li)) != SPX_OK) return rc;
            spx_ctx::EvalLane& lane = ctx->lanes[li];
            spx_ctx::EvalLane& other = ctx->lanes[li ^ 1];
            std::memcpy(lane.hIn, positions + lo, m * sizeof(spx_packed_pos));
            SPX_HIP(hipMemcpyAsync(lane.dIn, lane.hIn, m * sizeof(spx_packed_pos), hipMemcpyHostToDevice, lane.stream));
            swapLane(ctx, lane);
            ctx->ftGateWait = other.ftRecorded ? other.ftDone : nullptr;
            ctx->ftGateRecord = lane.ftDone;
            rc = spx_eval_full_device(ctx, lane.dIn, m, lane.dOutStage, lane.stream);
            ctx->ftGateWait = ctx->ftGateRecord = nullptr;
            swapLane(ctx, lane);
            if (rc != SPX_OK) return rc;
            lane.ftRecorded = true;
            SPX_HIP(hipMemcpyAsync(lane.hOut, lane.dOutStage, m * sizeof(int32_t), hipMemcpyDeviceToHost, lane.stream));
            SPX_HIP(hipEventRecord(lane.done, lane.stream));
            pending[li].lo = lo;
            pending[li].m = m;
            pending[li].active = true;
        }
        if ((rc = drain(0)) != SPX_OK) return rc;
        return drain(1);
    }
    // host buffers of any length: processed in chunks of the context's capacity (device variants are strict)
    for (size_t lo = 0; lo < n; lo += ctx->maxBatch) {
        const size_t m = std::min(ctx->maxBatch, n - lo);
        SPX_HIP(hipMemcpyAsync(ctx->dPositions, positions + lo, m * sizeof(spx_packed_pos), hipMemcpyHostToDevice,
                               ctx->stream));
        const int rc = spx_eval_full_device(ctx, ctx->dPositions, m, ctx->dOut, ctx->stream);
        if (rc != SPX_OK) return rc;
        SPX_HIP(hipMemcpyAsync(out + lo, ctx->dOut, m * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        SPX_HIP(hipStreamSynchronize(ctx->stream));
    }
    return SPX_OK;
}

void* spx_host_alloc(size_t nbytes) {
    void* ptr = nullptr;
    if (nbytes == 0 || hipHostMalloc(&ptr, nbytes, hipHostMallocDefault) != hipSuccess) {
        setError("spx_host_alloc: cannot allocate " + std::to_string(nbytes) + " bytes of page-locked memory");
        return nullptr;
    }
    return ptr;
}

void spx_host_free(void* ptr) {
    if (ptr) (void)hipHostFree(ptr);
}

void spx_adjust_defaults(spx_adjust_params* params) {
    if (!params) return;
    *params = spx_adjust_params{};
    const int32_t values[5] = {48, 442, 461, 637, 1223};  // tunable.h:161-165
    std::copy(values, values + 5, params->scaling_value);
    params->material_scaling_base = 26000;   // tunable.h:167
    params->optimism_base = 2024;            // tunable.h:168
    params->optimism_material_scale = 1005;  // tunable.h:169
    params->stages = SPX_ADJUST_STATIC | SPX_ADJUST_EVAL;
}

int spx_adjust_device(spx_ctx* ctx, const void* d_positions, size_t n, const spx_adjust_params* params,
                      const void* d_corrections, void* d_evals, void* stream) {
    if (!ctx || !params || (n && (!d_positions || !d_evals)) || n > (1ull << 30)) {
        setError("spx_adjust_device: invalid argument");
        return SPX_ERR_INVALID_ARG;
    }
    if (params->stages == 0 ||
        (params->stages & ~uint32_t(SPX_ADJUST_STATIC | SPX_ADJUST_EVAL | SPX_ADJUST_WHITE_POV | SPX_ADJUST_WDL))) {
        setError("spx_adjust_device: stages must be a combination of the SPX_ADJUST_* flags");
        return SPX_ERR_INVALID_ARG;
    }
    if (n == 0) return SPX_OK;
    SPX_HIP(hipSetDevice(ctx->device));
    AdjustParams ap{};
    std::copy(params->contempt, params->contempt + 2, ap.contempt);
    std::copy(params->optimism, params->optimism + 2, ap.optimism);
    std::copy(params->scaling_value, params->scaling_value + 5, ap.scalingValue);
    ap.materialScalingBase = params->material_scaling_base;
    ap.optimismBase = params->optimism_base;
    ap.optimismMaterialScale = params->optimism_material_scale;
    ap.stages = params->stages;
    ap.nPositions = uint32_t(n);
    ap.positions = static_cast<const uint64_t*>(d_positions);
    ap.corrections = static_cast<const int32_t*>(d_corrections);
    ap.evals = static_cast<int32_t*>(d_evals);
    SPX_HIP(launchAdjust(ap, stream ? static_cast<hipStream_t>(stream) : ctx->stream));
    return SPX_OK;
}

int spx_adjust(spx_ctx* ctx, const spx_packed_pos* positions, size_t n, const spx_adjust_params* params,
               const int32_t* corrections, int32_t* evals) {
    if (!ctx || !params || (n && (!positions || !evals))) {
        setError("spx_adjust: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    if (n == 0) return SPX_OK;
    SPX_HIP(hipSetDevice(ctx->device));
    for (size_t lo = 0; lo < n; lo += ctx->maxBatch) {  // dSlotsA doubles as the corrections staging buffer
        const size_t m = std::min(ctx->maxBatch, n - lo);
        SPX_HIP(hipMemcpyAsync(ctx->dPositions, positions + lo, m * sizeof(spx_packed_pos), hipMemcpyHostToDevice,
                               ctx->stream));
        SPX_HIP(hipMemcpyAsync(ctx->dOut, evals + lo, m * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        if (corrections) {
            SPX_HIP(hipMemcpyAsync(ctx->dSlotsA, corrections + lo, m * sizeof(int32_t), hipMemcpyHostToDevice,
                                   ctx->stream));
        }
        const int rc = spx_adjust_device(ctx, ctx->dPositions, m, params, corrections ? ctx->dSlotsA : nullptr,
                                         ctx->dOut, ctx->stream);
        if (rc != SPX_OK) return rc;
        SPX_HIP(hipMemcpyAsync(evals + lo, ctx->dOut, m * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        SPX_HIP(hipStreamSynchronize(ctx->stream));
    }
    return SPX_OK;
}

int spx_debug_copy_ft(spx_ctx* ctx, size_t n, uint8_t* out) {
    if (!ctx || !out || n > ctx->maxBatch) {
        setError("spx_debug_copy_ft: invalid argument");
        return SPX_ERR_INVALID_ARG;
    }
    SPX_HIP(hipSetDevice(ctx->device));
    SPX_HIP(hipDeviceSynchronize());
    SPX_HIP(hipMemcpy(out, ctx->dFtOut, n * size_t(kL1), hipMemcpyDeviceToHost));
    return SPX_OK;
}

// start / end of every workgroup of the LAST column-sliced gather that used the given scratch set (slot -1: the context's own,
// 0 / 1: the pipelined calls' lanes), on the device's constant 100 MHz clock: out[2 b] = start, out[2 b + 1] = end of workgroup b
int spx_debug_ftx_block_times(spx_ctx* ctx, int slot, uint64_t* out) {
    if (!ctx || !out || slot < -1 || slot > 2) {
        setError("spx_debug_ftx_block_times: invalid argument");
        return SPX_ERR_INVALID_ARG;
    }
    const FtxScratch& x = slot < 0 ? ctx->ftx : ctx->lanes[slot].ftx;
    if (!x.plan) {
        setError("spx_debug_ftx_block_times: that scratch set was never used");
        return SPX_ERR_INVALID_ARG;
    }
    SPX_HIP(hipSetDevice(ctx->device));
    SPX_HIP(hipDeviceSynchronize());
    SPX_HIP(hipMemcpy(out, x.plan + kFtxPlanTimes, 512 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return SPX_OK;
}

// what the LAST packed walk of that scratch set holds (summed over the group heads, spx_ftx.h): out[0] groups, [1] stages, [2] steps of
// the COLD sections and [3] of the LDS sections as walked per column slice, [4] cold rows fetched through the texture path, [5] rows read
// from LDS; the high-byte planes differ per slice - an XCD drops the planes that are all zero in its slice (spx_ftx_gather_kernel) -,
// so they are recounted here from the groups' first stages: [6] steps of the high-byte sections as walked, summed over the 8 slices,
// [7] plane slices (128 B each) fetched, summed over the 8 slices
int spx_debug_ftx_walk(spx_ctx* ctx, int slot, uint32_t* out) {
    if (!ctx || !out || slot < -1 || slot > 2) {
        setError("spx_debug_ftx_walk: invalid argument");
        return SPX_ERR_INVALID_ARG;
    }
    const FtxScratch& x = slot < 0 ? ctx->ftx : ctx->lanes[slot].ftx;
    if (!x.plan) {
        setError("spx_debug_ftx_walk: that scratch set was never used");
        return SPX_ERR_INVALID_ARG;
    }
    SPX_HIP(hipSetDevice(ctx->device));
    SPX_HIP(hipDeviceSynchronize());
    uint32_t nGroups = 0;
    SPX_HIP(hipMemcpy(&nGroups, x.plan + 33, sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::vector<uint32_t> heads(size_t(nGroups) * kFtxGroupHeadWords);
    SPX_HIP(hipMemcpy(heads.data(), x.groupHead, heads.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    uint64_t sums[8] = {nGroups, 0, 0, 0, 0, 0, 0, 0};  // (the pack kernel leaves every group's own figures in its head's spare words)
    bool anyHi = false;
    for (uint32_t g = 0; g < nGroups; ++g) {
        for (int k = 1; k < 6; ++k) sums[k] += heads[size_t(g) * kFtxGroupHeadWords + 8 + k];
        anyHi = anyHi || (heads[size_t(g) * kFtxGroupHeadWords] & 0xFFu);
    }
    if (anyHi) {
        // the high-byte stage of every group (its first: <= 32 planes per perspective are one stage), word 32 k + 4 (2 kb + u) + pr =
        // plane kb of step k of perspective 2 pr + u, bits 24 + x: the plane is not all zero in slice x
        std::vector<uint32_t> stage(size_t(nGroups) * 256);
        SPX_HIP(hipMemcpy2D(stage.data(), 1024, x.stages, size_t(kFtxMaxStages) * 1024, 1024, nGroups, hipMemcpyDeviceToHost));
        for (uint32_t g = 0; g < nGroups; ++g) {
            if (!(heads[size_t(g) * kFtxGroupHeadWords] & 0xFFu)) continue;
            const uint32_t* w = stage.data() + size_t(g) * 256;
            for (uint32_t xs = 0; xs < 8; ++xs) {
                uint32_t kept[8] = {0, 0, 0, 0, 0, 0, 0, 0}, longest = 0;
                for (uint32_t i = 0; i < 256; ++i) {
                    if ((w[i] >> (24 + xs)) & 1u) ++kept[2 * (i & 3u) + ((i >> 2) & 1u)];
                }
                for (uint32_t q = 0; q < 8; ++q) {
                    longest = std::max(longest, kept[q]);
                    sums[7] += kept[q];
                }
                sums[6] += (longest + 3) / 4;
            }
        }
    }
    for (int k = 0; k < 8; ++k) out[k] = uint32_t(std::min<uint64_t>(sums[k], 0xFFFFFFFFull));
    return SPX_OK;
}

// The row lists the extraction pass wrote for the LAST batch of that scratch set, decoded back to the net's row numbering (the
// reference's feature indices: psq.h:338-365, nnue_state.cpp:309-354): per perspective 2 i + c (c = its colour, 1 = white)
// counts[3] = {piece-square rows, threat / pawn-pair rows, high-byte planes} and rows[kFtxListStride] = the piece-square rows
// (bucket * 704 + slab row), then the threat / pawn-pair rows (hot slots through the context's hot set, cold ones from their slice
// offsets), then the piece-square rows whose high-byte plane is listed. tests compare them as multisets with tests/golden/features.jsonl.
int spx_debug_ftx_lists(spx_ctx* ctx, int slot, size_t n, uint32_t* counts, uint32_t* rows) {
    if (!ctx || !counts || !rows || slot < -1 || slot > 2) {
        setError("spx_debug_ftx_lists: invalid argument");
        return SPX_ERR_INVALID_ARG;
    }
    const FtxScratch& x = slot < 0 ? ctx->ftx : ctx->lanes[slot].ftx;
    if (!x.lists || n > x.capacity) {
        setError("spx_debug_ftx_lists: that scratch set was never used (or holds fewer positions)");
        return SPX_ERR_INVALID_ARG;
    }
    SPX_HIP(hipSetDevice(ctx->device));
    SPX_HIP(hipDeviceSynchronize());
    std::vector<uint32_t> lists(2 * n * kFtxListStride), heads(2 * n * 4);
    SPX_HIP(hipMemcpy(lists.data(), x.lists, lists.size() * 4, hipMemcpyDeviceToHost));
    SPX_HIP(hipMemcpy(heads.data(), x.heads, heads.size() * 4, hipMemcpyDeviceToHost));
    for (size_t q = 0; q < 2 * n; ++q) {
        const uint32_t head = heads[4 * q], bucket = heads[4 * q + 2] / kFtxQuartetBins;
        const uint32_t nHi = head & 0x3Fu, nLds = (head >> 6) & 0x1FFu, nCold = (head >> 15) & 0x1FFu;
        const uint32_t* list = lists.data() + q * kFtxListStride;
        uint32_t* out = rows + q * kFtxListStride;
        uint32_t nPsq = 0, nThr = 0, k = 0;
        for (uint32_t i = 0; i < nLds; ++i) {  // piece-square rows first, then the hot rows
            const uint32_t off = list[kFtxListLds + i];
            if (off < kFtxSlabBytes) {
                out[k++] = bucket * kFtxSlabRows + off / 128;
                ++nPsq;
            }
        }
        for (uint32_t i = 0; i < nLds; ++i) {
            const uint32_t off = list[kFtxListLds + i];
            if (off >= kFtxSlabBytes) {
                const uint32_t hot = (off - kFtxSlabBytes) / 128;
                out[k++] = hot < ctx->hotIds.size() ? ctx->hotIds[hot] : 0xFFFFFFFFu;
                ++nThr;
            }
        }
        for (uint32_t i = 0; i < nCold; ++i) {
            out[k++] = list[kFtxListCold + i] / 128;
            ++nThr;
        }
        for (uint32_t i = 0; i < nHi; ++i) out[k++] = list[kFtxListHi + i] / 128 - kFtxPsqHiBase;
        counts[3 * q] = nPsq, counts[3 * q + 1] = nThr, counts[3 * q + 2] = nHi;
    }
    return SPX_OK;
}

// the plan and the bin starts of the LAST column-sliced gather of that scratch set: out[0 .. kFtxPlanTimes) = plan words (CU slot ->
// first segment, segments {bucket, first group, end group}), then kFtxBins + 17 words: first sorted position of every (bucket, length)
// bin, then the buckets' starts. tools/gpu_ftx_block_times.py fits the plan's cost model against the workgroups' times with it.
int spx_debug_ftx_plan(spx_ctx* ctx, int slot, uint32_t* out) {
    if (!ctx || !out || slot < -1 || slot > 2) {
        setError("spx_debug_ftx_plan: invalid argument");
        return SPX_ERR_INVALID_ARG;
    }
    const FtxScratch& x = slot < 0 ? ctx->ftx : ctx->lanes[slot].ftx;
    if (!x.plan) {
        setError("spx_debug_ftx_plan: that scratch set was never used");
        return SPX_ERR_INVALID_ARG;
    }
    SPX_HIP(hipSetDevice(ctx->device));
    SPX_HIP(hipDeviceSynchronize());
    SPX_HIP(hipMemcpy(out, x.plan, kFtxPlanTimes * sizeof(uint32_t), hipMemcpyDeviceToHost));
    SPX_HIP(hipMemcpy(out + kFtxPlanTimes, x.binStart, (kFtxBins + 17) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return SPX_OK;
}


// ---- host helpers ----
int spx_pos_from_fen(const char* fen, spx_packed_pos* out) {
    Board b;
    if (!out || !boardFromFen(fen, b)) {
        setError("spx_pos_from_fen: unparsable FEN");
        return SPX_ERR_BAD_POSITION;
    }
    packBoard(b, *out);
    return SPX_OK;
}

int spx_pos_to_fen(const spx_packed_pos* pos, char* buf, size_t nbytes) {
    Board b;
    if (!pos || !buf || !unpackBoard(*pos, b)) {
        setError("spx_pos_to_fen: bad record");
        return SPX_ERR_BAD_POSITION;
    }
    const std::string fen = boardToFen(b);
    if (fen.size() + 1 > nbytes) {
        setError("spx_pos_to_fen: buffer too small");
        return SPX_ERR_INVALID_ARG;
    }
    std::memcpy(buf, fen.c_str(), fen.size() + 1);
    return SPX_OK;
}

int spx_pos_to_mailbox(const spx_packed_pos* pos, uint8_t mailbox[64], int* stm) {
    Board b;
    if (!pos || !mailbox || !unpackBoard(*pos, b)) {
        setError("spx_pos_to_mailbox: bad record");
        return SPX_ERR_BAD_POSITION;
    }
    std::memcpy(mailbox, b.mailbox, 64);
    if (stm) *stm = b.stm;
    return SPX_OK;
}

int spx_pos_apply_uci(const spx_packed_pos* pos, const char* uci, spx_packed_pos* out) {
    Board b;
    if (!pos || !uci || !out || !unpackBoard(*pos, b)) {
        setError("spx_pos_apply_uci: bad record");
        return SPX_ERR_BAD_POSITION;
    }
    Move m;
    if (!moveFromUci(b, uci, m)) {
        setError(std::string("spx_pos_apply_uci: illegal or unparsable move ") + uci);
        return SPX_ERR_INVALID_ARG;
    }
    makeMove(b, m);
    packBoard(b, *out);
    return SPX_OK;
}

// a recorded tree (or forest) of moves -> the record of every node, in one native call (trace replays: hundreds of
// thousands of nodes; one Python round trip per node would dominate)
int spx_tree_expand_uci(const spx_packed_pos* roots, size_t n_roots, const uint32_t* parents, const char* moves, size_t n,
                        spx_packed_pos* out) {
    if (!parents || !moves || !out || (n_roots && !roots)) {
        setError("spx_tree_expand_uci: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    size_t nextRoot = 0;
    for (size_t k = 0; k < n; ++k) {
        const char* uci = moves + 6 * k;  // 6 bytes per node, NUL padded: "" = a root (the next of `roots`), "0000" = a null move
        if (uci[0] == 0) {
            if (nextRoot >= n_roots) {
                setError("spx_tree_expand_uci: more root nodes than roots");
                return SPX_ERR_INVALID_ARG;
            }
            out[k] = roots[nextRoot++];
            continue;
        }
        if (parents[k] >= k) {
            setError("spx_tree_expand_uci: node " + std::to_string(k) + " does not follow its parent");
            return SPX_ERR_INVALID_ARG;
        }
        if (std::memcmp(uci, "0000", 4) == 0) {  // Position::applyNullMove: same board, other side to move, no en-passant square
            out[k] = out[parents[k]];
            out[k].stm_ep = uint8_t(((out[k].stm_ep & 0x80u) ^ 0x80u) | 64u);
            continue;
        }
        char text[7] = {};
        std::memcpy(text, uci, 6);
        const int rc = spx_pos_apply_uci(&out[parents[k]], text, &out[k]);
        if (rc != SPX_OK) return rc;
    }
    return SPX_OK;
}

int spx_pos_apply_uci_observed(const spx_packed_pos* pos, const char* uci, spx_packed_pos* out, spx_move_delta* delta) {
    Board b;
    if (!pos || !uci || !out || !delta || !unpackBoard(*pos, b)) {
        setError("spx_pos_apply_uci_observed: bad argument");
        return SPX_ERR_BAD_POSITION;
    }
    Move m;
    if (!moveFromUci(b, uci, m)) {
        setError(std::string("spx_pos_apply_uci_observed: illegal or unparsable move ") + uci);
        return SPX_ERR_INVALID_ARG;
    }
    MoveDelta d;
    makeMoveObserved(b, m, d);
    packBoard(b, *out);
    if (d.threatsAdded.size() > 128 || d.threatsRemoved.size() > 128) {  // kMaxThreatsAdded/Removed, threats.h:35-36
        setError("spx_pos_apply_uci_observed: more than 128 threat descriptors");
        return SPX_ERR_CAPACITY;
    }
    std::memset(delta, 0, sizeof(*delta));
    delta->n_sub = d.nSub;
    delta->n_add = d.nAdd;
    for (int i = 0; i < d.nSub; ++i) {
        delta->sub_piece[i] = d.subPiece[i];
        delta->sub_sq[i] = d.subSq[i];
    }
    for (int i = 0; i < d.nAdd; ++i) {
        delta->add_piece[i] = d.addPiece[i];
        delta->add_sq[i] = d.addSq[i];
    }
    for (int c = 0; c < 2; ++c) {
        delta->psq_refresh[c] = d.psqRefresh[c];
        delta->threat_refresh[c] = d.threatRefresh[c];
        delta->kings[c] = d.kings[c];
        delta->pawns_before[c] = d.pawnsBefore[c];
        delta->pawns_after[c] = d.pawnsAfter[c];
    }
    delta->n_threats_added = uint8_t(d.threatsAdded.size());
    delta->n_threats_removed = uint8_t(d.threatsRemoved.size());
    for (size_t i = 0; i < d.threatsAdded.size(); ++i) {
        const ThreatDescriptor& t = d.threatsAdded[i];
        delta->threats_added[i] = {t.attacker, t.attackerSq, t.attacked, t.attackedSq};
    }
    for (size_t i = 0; i < d.threatsRemoved.size(); ++i) {
        const ThreatDescriptor& t = d.threatsRemoved[i];
        delta->threats_removed[i] = {t.attacker, t.attackerSq, t.attacked, t.attackedSq};
    }
    return SPX_OK;
}

int spx_random_positions(uint64_t seed, size_t count, int min_ply, int max_ply, int dfrc_every, spx_packed_pos* out) {
    if (!out || min_ply < 0 || max_ply < min_ply || max_ply > 600) {
        setError("spx_random_positions: invalid argument");
        return SPX_ERR_INVALID_ARG;
    }
    randomPositions(seed, count, min_ply, max_ply, dfrc_every, out);
    return SPX_OK;
}

int spx_random_successors(uint64_t seed, const spx_packed_pos* positions, size_t n, spx_packed_pos* out, uint8_t* moved) {
    if ((n && (!positions || !out))) {
        setError("spx_random_successors: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    uint64_t s = seed;
    auto next = [&s]() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    std::vector<Move> moves;
    for (size_t i = 0; i < n; ++i) {
        Board b;
        if (!unpackBoard(positions[i], b)) {
            setError("spx_random_successors: bad record at index " + std::to_string(i));
            return SPX_ERR_BAD_POSITION;
        }
        generateLegal(b, moves);
        const bool any = !moves.empty();
        if (any) {
            makeMove(b, moves[size_t((next() >> 32) % moves.size())]);
            packBoard(b, out[i]);
        } else {
            out[i] = positions[i];  // checkmate / stalemate: the game stays where it is
        }
        if (moved) moved[i] = any ? 1 : 0;
    }
    return SPX_OK;
}

// ---- move generation ----
static uint16_t viriMoveWord(const Move& m) {  // viriformat.cpp:37-52
    static const uint16_t kTypes[4] = {0x0000, 0xC000, 0x8000, 0x4000};  // by MoveKind: normal, promotion, castling, ep
    return uint16_t(m.from | (m.to << 6) | ((m.kind == kPromotion ? m.promo - 1 : 0) << 12) | kTypes[m.kind]);
}

int spx_pos_legal_moves(const spx_packed_pos* pos, uint16_t* moves, spx_packed_pos* children, int* n, int* in_check) {
    Board b;
    if (!pos || !moves || !n || !unpackBoard(*pos, b)) {
        setError("spx_pos_legal_moves: null argument or bad record");
        return SPX_ERR_BAD_POSITION;
    }
    std::vector<Move> legal;
    generateLegal(b, legal);
    if (legal.size() > 256) {
        setError("spx_pos_legal_moves: more than 256 legal moves");
        return SPX_ERR_CAPACITY;
    }
    *n = int(legal.size());
    if (in_check) *in_check = b.inCheck() ? 1 : 0;
    for (size_t k = 0; k < legal.size(); ++k) {
        moves[k] = viriMoveWord(legal[k]);
        if (children) {
            Board next = b;
            makeMove(next, legal[k]);
            packBoard(next, children[k]);
        }
    }
    return SPX_OK;
}

int spx_movegen_device(spx_ctx* ctx, const void* d_positions, size_t n, const void* d_parent_values, void* d_children,
                       void* d_moves, void* d_parents, void* d_first, void* d_count, void* d_in_check, size_t capacity,
                       void* d_total, void* stream) {
    if (!ctx || (n && (!d_positions || !d_children || !d_moves || !d_parents || !d_first || !d_count || !d_in_check)) ||
        !d_total || n > (1ull << 30) || capacity > 0xFFFFFFFFull) {
        setError("spx_movegen_device: invalid argument");
        return SPX_ERR_INVALID_ARG;
    }
    SPX_HIP(hipSetDevice(ctx->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    SPX_HIP(hipMemsetAsync(d_total, 0, sizeof(uint32_t), s));
    if (n == 0) return SPX_OK;
    MovegenParams mp{};
    mp.positions = static_cast<const uint64_t*>(d_positions);
    mp.nPositions = uint32_t(n);
    mp.parentValues = static_cast<const uint32_t*>(d_parent_values);
    mp.children = static_cast<uint64_t*>(d_children);
    mp.moves = static_cast<uint16_t*>(d_moves);
    mp.parents = static_cast<uint32_t*>(d_parents);
    mp.first = static_cast<uint32_t*>(d_first);
    mp.count = static_cast<uint32_t*>(d_count);
    mp.inCheck = static_cast<uint8_t*>(d_in_check);
    mp.cursor = static_cast<uint32_t*>(d_total);
    mp.capacity = uint32_t(capacity);
    uint32_t blocks = uint32_t((n + 3) / 4);
    if (blocks > ctx->ftGridCap) blocks = ctx->ftGridCap;
    SPX_HIP(launchMovegen(mp, blocks, s));
    return SPX_OK;
}

int spx_movegen(spx_ctx* ctx, const spx_packed_pos* positions, size_t n, const uint32_t* parent_values,
                spx_packed_pos* children, uint16_t* moves, uint32_t* parents, uint32_t* first, uint32_t* count,
                uint8_t* in_check, size_t capacity, size_t* total) {
    if (!ctx || !total || (n && (!positions || !children || !moves || !parents || !first || !count || !in_check))) {
        setError("spx_movegen: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    *total = 0;
    if (n == 0) return SPX_OK;
    SPX_HIP(hipSetDevice(ctx->device));
    // convenience entry point (tests, tools): device scratch lives for the duration of the call
    struct Scratch {
        std::vector<void*> ptrs;
        ~Scratch() {
            for (void* q : ptrs) (void)hipFree(q);
        }
        void* get(size_t bytes) {
            void* q = nullptr;
            if (hipMalloc(&q, std::max<size_t>(bytes, 16)) != hipSuccess) return nullptr;
            ptrs.push_back(q);
            return q;
        }
    } scratch;
    void* dPos = scratch.get(n * 32);
    void* dPv = parent_values ? scratch.get(n * 4) : nullptr;
    void* dChildren = scratch.get(capacity * 32);
    void* dMoves = scratch.get(capacity * 2);
    void* dParents = scratch.get(capacity * 4);
    void* dFirst = scratch.get(n * 4);
    void* dCount = scratch.get(n * 4);
    void* dCheck = scratch.get(n);
    void* dTotal = scratch.get(4);
    if (!dPos || (parent_values && !dPv) || !dChildren || !dMoves || !dParents || !dFirst || !dCount || !dCheck || !dTotal) {
        setError("spx_movegen: out of device memory");
        return SPX_ERR_HIP;
    }
    hipStream_t s = ctx->stream;
    SPX_HIP(hipMemcpyAsync(dPos, positions, n * 32, hipMemcpyHostToDevice, s));
    if (parent_values) SPX_HIP(hipMemcpyAsync(dPv, parent_values, n * 4, hipMemcpyHostToDevice, s));
    const int rc = spx_movegen_device(ctx, dPos, n, dPv, dChildren, dMoves, dParents, dFirst, dCount, dCheck, capacity,
                                      dTotal, s);
    if (rc != SPX_OK) return rc;
    uint32_t produced = 0;
    SPX_HIP(hipMemcpyAsync(&produced, dTotal, 4, hipMemcpyDeviceToHost, s));
    SPX_HIP(hipMemcpyAsync(first, dFirst, n * 4, hipMemcpyDeviceToHost, s));
    SPX_HIP(hipMemcpyAsync(count, dCount, n * 4, hipMemcpyDeviceToHost, s));
    SPX_HIP(hipMemcpyAsync(in_check, dCheck, n, hipMemcpyDeviceToHost, s));
    SPX_HIP(hipStreamSynchronize(s));
    *total = produced;
    if (produced > capacity) {
        setError("spx_movegen: " + std::to_string(produced) + " children exceed the capacity of " + std::to_string(capacity));
        return SPX_ERR_CAPACITY;
    }
    SPX_HIP(hipMemcpy(children, dChildren, size_t(produced) * 32, hipMemcpyDeviceToHost));
    SPX_HIP(hipMemcpy(moves, dMoves, size_t(produced) * 2, hipMemcpyDeviceToHost));
    SPX_HIP(hipMemcpy(parents, dParents, size_t(produced) * 4, hipMemcpyDeviceToHost));
    return SPX_OK;
}

// ---- viriformat game streams (src/datagen/viriformat.cpp:28-63) ----
// game = PackedBoard (32 B) + { u16 move, i16 score }* + 4 zero bytes.
// move: bits 0-5 from, 6-11 to (castling: own rook square), 12-13 promotion (0 = knight .. 3 = queen),
//       bits 14-15 type (0 normal, 1 = 0x4000 en passant, 2 = 0x8000 castling, 3 = 0xC000 promotion).
static bool viriDecodeMove(const Board& b, uint16_t v, Move& out) {
    Move want{};
    want.from = uint8_t(v & 63);
    want.to = uint8_t((v >> 6) & 63);
    const int type = v >> 14;
    want.kind = type == 0 ? kNormal : type == 1 ? kEnPassant : type == 2 ? kCastling : kPromotion;
    want.promo = want.kind == kPromotion ? uint8_t(((v >> 12) & 3) + 1) : 0;
    std::vector<Move> moves;
    generateLegal(b, moves);
    for (const Move& m : moves) {
        if (m.from == want.from && m.to == want.to && m.kind == want.kind && m.promo == want.promo) {
            out = m;
            return true;
        }
    }
    return false;
}

static uint16_t viriEncodeMove(const Move& m) {
    static const uint16_t kTypes[4] = {0x0000, 0xC000, 0x8000, 0x4000};  // by MoveKind (normal, promo, castling, ep)
    return uint16_t(m.from | (m.to << 6) | ((m.kind == kPromotion ? m.promo - 1 : 0) << 12) | kTypes[m.kind]);
}

// The move that ENDS a game through Position::isDrawn is pushed as filtered whatever it is (datagen.cpp:264-268:
// output.push(true, move, 0)). `records` = the positions before each of the game's n moves (n >= 1), `lastMove` the viriformat
// word of the last one: is the position after it drawn (position.cpp:603-667)? The halfmove clock at 100 decides alone (a draw
// unless checkmate); else two earlier occurrences of the same position (only what the key distinguishes: placement with castling
// rights, side to move, en-passant square) or insufficient material. (This library's own ply cap is not one of the reference's
// rules: a game it cut short keeps its last position.)
static bool lastMoveEndsInADraw(const spx_packed_pos* records, size_t n, uint16_t lastMove) {
    Board b;
    Move m;
    if (!unpackBoard(records[n - 1], b) || !viriDecodeMove(b, lastMove, m)) return false;
    makeMove(b, m);
    if (b.halfmove >= 100) {
        std::vector<Move> replies;
        generateLegal(b, replies);
        return !(b.inCheck() && replies.empty());
    }
    spx_packed_pos after;
    packBoard(b, after);
    size_t seen = 0;
    for (size_t k = 0; k < n; ++k) {
        seen += records[k].occupancy == after.occupancy && std::memcmp(records[k].pieces, after.pieces, 16) == 0 &&
                records[k].stm_ep == after.stm_ep;
    }
    if (seen >= 2) return true;
    uint64_t lo, hi;
    std::memcpy(&lo, after.pieces, 8);
    std::memcpy(&hi, after.pieces + 8, 8);
    return insufficientMaterial(after.occupancy, lo, hi);
}

int spx_viri_expand(const void* data, size_t nbytes, spx_packed_pos* out, int16_t* scores, uint8_t* unfiltered,
                    size_t capacity, size_t* n_positions, size_t* n_games) {
    if (!data || !n_positions) {
        setError("spx_viri_expand: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    const auto* p = static_cast<const unsigned char*>(data);
    size_t off = 0, count = 0, games = 0;
    while (off + sizeof(spx_packed_pos) + 4 <= nbytes) {
        spx_packed_pos initial;
        std::memcpy(&initial, p + off, sizeof(initial));
        off += sizeof(initial);
        Board b;
        if (!unpackBoard(initial, b)) {
            setError("spx_viri_expand: bad initial board in game " + std::to_string(games));
            return SPX_ERR_BAD_POSITION;
        }
        const size_t first = count;
        uint16_t lastMove = 0;
        for (;;) {
            if (off + 4 > nbytes) {
                setError("spx_viri_expand: truncated game " + std::to_string(games));
                return SPX_ERR_INVALID_ARG;
            }
            uint16_t mv;
            int16_t score;
            std::memcpy(&mv, p + off, 2);
            std::memcpy(&score, p + off + 2, 2);
            off += 4;
            if (mv == 0 && score == 0) break;  // null terminator
            Move m;
            if (!viriDecodeMove(b, mv, m)) {
                setError("spx_viri_expand: illegal move in game " + std::to_string(games));
                return SPX_ERR_BAD_POSITION;
            }
            if (out) {
                if (count >= capacity) {
                    setError("spx_viri_expand: output capacity exceeded");
                    return SPX_ERR_CAPACITY;
                }
                packBoard(b, out[count]);
                out[count].eval = score;
                out[count].wdl = initial.wdl;
                if (scores) scores[count] = score;
                if (unfiltered) {  // datagen.cpp:254: filtered = pos.isCheck() || pos.isNoisy(move) (position.cpp:683-689)
                    const bool noisy = m.kind != kCastling && (m.kind == kEnPassant || (m.kind == kPromotion && m.promo == 4) ||
                                                                b.mailbox[m.to] != kNoPiece);
                    unfiltered[count] = (b.inCheck() || noisy) ? 0 : 1;
                }
            }
            ++count;
            lastMove = mv;
            makeMove(b, m);
        }
        if (out && unfiltered && count > first && lastMoveEndsInADraw(out + first, count - first, lastMove)) unfiltered[count - 1] = 0;
        ++games;
    }
    *n_positions = count;
    if (n_games) *n_games = games;
    return SPX_OK;
}

// datagen's other output formats from a viriformat stream (include/spx_nnue.h): the expansion above, filtered
static int expandFiltered(const void* data, size_t nbytes, std::vector<spx_packed_pos>& kept, size_t* n_games, const char* who) {
    size_t n = 0, games = 0;
    int rc = spx_viri_expand(data, nbytes, nullptr, nullptr, nullptr, 0, &n, &games);
    if (rc != SPX_OK) return rc;
    std::vector<spx_packed_pos> all(n);
    std::vector<uint8_t> keep(n);
    rc = spx_viri_expand(data, nbytes, all.data(), nullptr, keep.data(), n, &n, &games);
    if (rc != SPX_OK) return rc;
    kept.clear();
    for (size_t i = 0; i < n; ++i) {
        if (keep[i]) kept.push_back(all[i]);
    }
    if (n_games) *n_games = games;
    (void)who;
    return SPX_OK;
}

int spx_viri_to_marlinformat(const void* data, size_t nbytes, spx_packed_pos* out, size_t capacity, size_t* n_records,
                             size_t* n_games) {
    if (!data || !n_records) {
        setError("spx_viri_to_marlinformat: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    std::vector<spx_packed_pos> kept;
    const int rc = expandFiltered(data, nbytes, kept, n_games, "spx_viri_to_marlinformat");
    if (rc != SPX_OK) return rc;
    *n_records = kept.size();
    if (!out) return SPX_OK;
    if (kept.size() > capacity) {
        setError("spx_viri_to_marlinformat: output capacity exceeded");
        return SPX_ERR_CAPACITY;
    }
    if (!kept.empty()) std::memcpy(out, kept.data(), kept.size() * sizeof(spx_packed_pos));
    return SPX_OK;
}

int spx_viri_to_fen(const void* data, size_t nbytes, char* out, size_t capacity, size_t* n_bytes, size_t* n_games) {
    if (!data || !n_bytes) {
        setError("spx_viri_to_fen: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    std::vector<spx_packed_pos> kept;
    int rc = expandFiltered(data, nbytes, kept, n_games, "spx_viri_to_fen");
    if (rc != SPX_OK) return rc;
    std::string text;
    text.reserve(kept.size() * 80);
    static const char* const kOutcome[3] = {"0.0", "0.5", "1.0"};  // Outcome: white loss, draw, white win (fen.cpp:47-59)
    char fen[128];
    for (const spx_packed_pos& rec : kept) {
        if ((rc = spx_pos_to_fen(&rec, fen, sizeof(fen))) != SPX_OK) return rc;
        if (rec.wdl > 2) {
            setError("spx_viri_to_fen: outcome byte " + std::to_string(rec.wdl) + " is not 0, 1 or 2");
            return SPX_ERR_BAD_POSITION;
        }
        text += fen;
        text += " | ";
        text += std::to_string(int(rec.eval));
        text += " | ";
        text += kOutcome[rec.wdl];
        text += '\n';
    }
    *n_bytes = text.size();
    if (!out) return SPX_OK;
    if (text.size() > capacity) {
        setError("spx_viri_to_fen: output capacity exceeded");
        return SPX_ERR_CAPACITY;
    }
    std::memcpy(out, text.data(), text.size());
    return SPX_OK;
}

// viriformat expansion on the device: the host only finds the game boundaries (one linear scan for the 4-byte null
// terminators), a thread per game replays the moves (spx_viri_expand_kernel)
int spx_viri_expand_gpu(spx_ctx* ctx, const void* data, size_t nbytes, spx_packed_pos* out, uint8_t* unfiltered,
                        size_t capacity, size_t* n_positions, size_t* n_games, size_t* bad_games) {
    if (!ctx || !data || !n_positions) {
        setError("spx_viri_expand_gpu: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    const auto* p = static_cast<const unsigned char*>(data);
    std::vector<uint64_t> gameOffset, outOffset;
    size_t off = 0, count = 0;
    while (off + sizeof(spx_packed_pos) + 4 <= nbytes) {
        gameOffset.push_back(off);
        outOffset.push_back(count);
        off += sizeof(spx_packed_pos);
        for (;;) {
            if (off + 4 > nbytes) {
                setError("spx_viri_expand_gpu: truncated game " + std::to_string(gameOffset.size() - 1));
                return SPX_ERR_INVALID_ARG;
            }
            uint32_t word;
            std::memcpy(&word, p + off, 4);
            off += 4;
            if (word == 0) break;  // null move + null score
            ++count;
        }
    }
    outOffset.push_back(count);
    *n_positions = count;
    if (n_games) *n_games = gameOffset.size();
    if (bad_games) *bad_games = 0;
    if (!out || count == 0) return SPX_OK;
    if (count > capacity) {
        setError("spx_viri_expand_gpu: output capacity exceeded");
        return SPX_ERR_CAPACITY;
    }
    if (gameOffset.size() > 0xFFFFFFFFull) {
        setError("spx_viri_expand_gpu: too many games in one call");
        return SPX_ERR_INVALID_ARG;
    }
    SPX_HIP(hipSetDevice(ctx->device));
    struct Scratch {
        std::vector<void*> ptrs;
        ~Scratch() {
            for (void* q : ptrs) (void)hipFree(q);
        }
        void* get(size_t bytes) {
            void* q = nullptr;
            if (hipMalloc(&q, std::max<size_t>(bytes, 16)) != hipSuccess) return nullptr;
            ptrs.push_back(q);
            return q;
        }
    } scratch;
    void* dData = scratch.get(off);
    void* dGameOffset = scratch.get(gameOffset.size() * 8);
    void* dOutOffset = scratch.get(outOffset.size() * 8);
    void* dOut = scratch.get(count * sizeof(spx_packed_pos));
    void* dBad = scratch.get(4);
    void* dKeep = unfiltered ? scratch.get(count) : nullptr;
    if (!dData || !dGameOffset || !dOutOffset || !dOut || !dBad || (unfiltered && !dKeep)) {
        setError("spx_viri_expand_gpu: out of device memory");
        return SPX_ERR_HIP;
    }
    hipStream_t s = ctx->stream;
    SPX_HIP(hipMemcpyAsync(dData, data, off, hipMemcpyHostToDevice, s));
    SPX_HIP(hipMemcpyAsync(dGameOffset, gameOffset.data(), gameOffset.size() * 8, hipMemcpyHostToDevice, s));
    SPX_HIP(hipMemcpyAsync(dOutOffset, outOffset.data(), outOffset.size() * 8, hipMemcpyHostToDevice, s));
    SPX_HIP(hipMemsetAsync(dBad, 0, 4, s));
    ViriExpandParams vp{};
    vp.data = static_cast<const uint8_t*>(dData);
    vp.nGames = uint32_t(gameOffset.size());
    vp.gameOffset = static_cast<const uint64_t*>(dGameOffset);
    vp.outOffset = static_cast<const uint64_t*>(dOutOffset);
    vp.out = static_cast<uint64_t*>(dOut);
    vp.unfiltered = static_cast<uint8_t*>(dKeep);
    vp.badGames = static_cast<uint32_t*>(dBad);
    SPX_HIP(launchViriExpand(vp, s));
    uint32_t bad = 0;
    SPX_HIP(hipMemcpyAsync(out, dOut, count * sizeof(spx_packed_pos), hipMemcpyDeviceToHost, s));
    SPX_HIP(hipMemcpyAsync(&bad, dBad, 4, hipMemcpyDeviceToHost, s));
    if (unfiltered) SPX_HIP(hipMemcpyAsync(unfiltered, dKeep, count, hipMemcpyDeviceToHost, s));
    SPX_HIP(hipStreamSynchronize(s));
    if (bad_games) *bad_games = bad;
    if (unfiltered && bad == 0) {  // the game-level rule on top of the kernel's per-position filter: one position per game on the host
        for (size_t g = 0; g + 1 < outOffset.size(); ++g) {
            const size_t lo = outOffset[g], hi = outOffset[g + 1];
            if (hi == lo) continue;
            uint16_t lastMove;
            std::memcpy(&lastMove, p + gameOffset[g] + sizeof(spx_packed_pos) + (hi - lo - 1) * 4, 2);
            if (lastMoveEndsInADraw(out + lo, hi - lo, lastMove)) unfiltered[hi - 1] = 0;
        }
    }
    return SPX_OK;
}

int spx_viri_random_game(uint64_t seed, int plies, int dfrc, void* buf, size_t capacity, size_t* nbytes) {
    if (!buf || !nbytes || plies < 0) {
        setError("spx_viri_random_game: invalid argument");
        return SPX_ERR_INVALID_ARG;
    }
    uint64_t s = seed;
    auto next = [&s]() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    Board b = dfrc ? dfrcStart(uint32_t(next() % 960), uint32_t(next() % 960)) : startpos();
    auto* p = static_cast<unsigned char*>(buf);
    size_t off = 0;
    if (capacity < sizeof(spx_packed_pos) + 4) {
        setError("spx_viri_random_game: buffer too small");
        return SPX_ERR_CAPACITY;
    }
    spx_packed_pos initial;
    packBoard(b, initial);
    initial.wdl = 1;
    std::memcpy(p, &initial, sizeof(initial));
    off += sizeof(initial);
    std::vector<Move> moves;
    for (int i = 0; i < plies; ++i) {
        generateLegal(b, moves);
        if (moves.empty()) break;
        const Move m = moves[size_t((next() >> 32) % moves.size())];
        if (off + 8 > capacity) {
            setError("spx_viri_random_game: buffer too small");
            return SPX_ERR_CAPACITY;
        }
        uint16_t mv = viriEncodeMove(m);
        const int16_t score = int16_t(int(next() % 2001) - 1000);
        if (mv == 0 && score == 0) mv = 0;  // a1a1 cannot be legal; kept for clarity
        std::memcpy(p + off, &mv, 2);
        std::memcpy(p + off + 2, &score, 2);
        off += 4;
        makeMove(b, m);
    }
    std::memset(p + off, 0, 4);
    off += 4;
    *nbytes = off;
    return SPX_OK;
}

uint64_t spx_perft(const char* fen, int depth) {
    Board b;
    if (!boardFromFen(fen, b) || depth < 0) return 0;
    return perft(b, depth);
}

// Lane-by-lane host emulation of spx_ft_kernel's extraction phase (same SPX_HD helpers, same loop structure).
int spx_debug_features(const spx_packed_pos* pos, int c, uint32_t* psqRows, int* nPsq, uint32_t* thrRows, int* nThr) {
    if (!pos || !psqRows || !nPsq || !thrRows || !nThr || (c != 0 && c != 1)) {
        setError("spx_debug_features: invalid argument");
        return SPX_ERR_INVALID_ARG;
    }
    const uint64_t occ = pos->occupancy;
    int piece[64];
    uint64_t kingsBb = 0, whiteBb = 0, pawnsBb = 0;
    int kingSq = -1;
    for (int lane = 0; lane < 64; ++lane) {
        piece[lane] = kNoPiece;
        if ((occ >> lane) & 1) {
            const int idx = popc64(occ & ((1ull << lane) - 1));
            if (idx >= 32) {
                setError("spx_debug_features: more than 32 pieces");
                return SPX_ERR_BAD_POSITION;
            }
            piece[lane] = nibbleToPiece((pos->pieces[idx >> 1] >> ((idx & 1) * 4)) & 0xF);
            if ((piece[lane] >> 1) == 5) kingsBb |= 1ull << lane;
            if (piece[lane] == (10 | c)) kingSq = lane;
            if (piece[lane] & 1) whiteBb |= 1ull << lane;
            if ((piece[lane] >> 1) == 0 && lane >= 8 && lane < 56) pawnsBb |= 1ull << lane;
        }
    }
    if (kingSq < 0) {
        setError("spx_debug_features: no king");
        return SPX_ERR_BAD_POSITION;
    }
    static const std::vector<uint32_t> lutStorage = [] {
        std::vector<uint32_t> v(kLutWords);
        buildThreatLut(v.data());
        return v;
    }();
    const uint32_t* lut = lutStorage.data();
    const uint64_t ownPawns = pawnsBb & (c ? whiteBb : ~whiteBb), theirPawns = pawnsBb & ~ownPawns;
    const int x = perspXor(c, kingSq), flipColour = c == 0;
    int np = 0, nt = 0;
    for (int lane = 0; lane < 64; ++lane) {
        if (piece[lane] != kNoPiece) psqRows[np++] = psqRow(c, piece[lane], lane, kingSq);
    }
    for (int lane = 0; lane < 64; ++lane) {
        const int pc = piece[lane];
        if (pc == kNoPiece || (pc >> 1) == 5) continue;
        uint64_t targets = pieceAttacks(pc, lane, occ) & occ & ~kingsBb;
        const uint64_t pseudoRel = piecePseudoAttacks(pc ^ flipColour, lane ^ x);
        while (targets) {
            const int to = ctz64(targets);
            targets &= targets - 1;
            const int32_t row = threatRow(lut, pc ^ flipColour, lane ^ x, pseudoRel, piece[to] ^ flipColour, to ^ x);
            if (row >= 0 && nt < 256) thrRows[nt++] = uint32_t(row);
        }
    }
    for (int lane = 0; lane < 64; ++lane) {
        if (piece[lane] == kNoPiece || (piece[lane] >> 1) != 0) continue;
        const bool own = (piece[lane] & 1) == c;
        const uint64_t above = ~((2ull << lane) - 1);
        uint64_t partners = own ? (((ownPawns & above) | theirPawns) & ppMask(lane)) : (theirPawns & above & ppMask(lane));
        const uint32_t idA = ppId(lane ^ x, !own);
        while (partners) {
            const int b = ctz64(partners);
            partners &= partners - 1;
            if (nt < 256) thrRows[nt++] = ppRow(idA, ppId(b ^ x, !((ownPawns >> b) & 1)));
        }
    }
    *nPsq = np;
    *nThr = nt;
    return SPX_OK;
}


// Host emulation of spx_update_kernel's delta derivation: the same SPX_HD per-lane code (deltaCandidates, descRow,
// pawn-pair partner sets), run lane by lane. Capacities: 8 piece-square rows, 288 threat / pawn-pair rows per list.
int spx_debug_delta(const spx_packed_pos* parent, const spx_packed_pos* child, int c, uint32_t* psqSub, int* nPsqSub,
                    uint32_t* psqAdd, int* nPsqAdd, uint32_t* thrSub, int* nThrSub, uint32_t* thrAdd, int* nThrAdd,
                    int* refresh) {
    if (!parent || !child || !psqSub || !nPsqSub || !psqAdd || !nPsqAdd || !thrSub || !nThrSub || !thrAdd || !nThrAdd ||
        !refresh || (c != 0 && c != 1)) {
        setError("spx_debug_delta: invalid argument");
        return SPX_ERR_INVALID_ARG;
    }
    static const std::vector<uint32_t> lutStorage = [] {
        std::vector<uint32_t> v(kLutWords);
        buildThreatLut(v.data());
        return v;
    }();
    static const std::vector<uint64_t> tabStorage = [] {
        std::vector<uint64_t> v(kDeltaTabWords);
        buildDeltaTables(v.data());
        return v;
    }();
    const uint32_t* lut = lutStorage.data();
    const uint64_t* tab = tabStorage.data();
    struct Side {
        uint64_t occ = 0, white = 0, pawns = 0;
        uint8_t mail[64];
        int king[2] = {-1, -1};
    } side[2];
    const spx_packed_pos* recs[2] = {parent, child};
    for (int b = 0; b < 2; ++b) {
        Side& sd = side[b];
        sd.occ = recs[b]->occupancy;
        if (popc64(sd.occ) > 32) {
            setError("spx_debug_delta: more than 32 pieces");
            return SPX_ERR_BAD_POSITION;
        }
        for (int sq = 0; sq < 64; ++sq) {
            sd.mail[sq] = kNoPiece;
            if (!((sd.occ >> sq) & 1)) continue;
            const int idx = popc64(sd.occ & ((1ull << sq) - 1));
            const int pc = nibbleToPiece((recs[b]->pieces[idx >> 1] >> ((idx & 1) * 4)) & 0xF);
            sd.mail[sq] = uint8_t(pc);
            if (pc & 1) sd.white |= 1ull << sq;
            if ((pc >> 1) == 0 && sq >= 8 && sq < 56) sd.pawns |= 1ull << sq;
            if ((pc >> 1) == 5) sd.king[pc & 1] = sq;
        }
        if (sd.king[0] < 0 || sd.king[1] < 0) {
            setError("spx_debug_delta: no king");
            return SPX_ERR_BAD_POSITION;
        }
    }
    uint64_t changed = 0;
    for (int sq = 0; sq < 64; ++sq) {
        if (side[0].mail[sq] != side[1].mail[sq]) changed |= 1ull << sq;
    }
    const int kingP = side[0].king[c], kingC = side[1].king[c];
    const int relP = c == 0 ? (kingP ^ 56) : kingP, relC = c == 0 ? (kingC ^ 56) : kingC;
    *nPsqSub = *nPsqAdd = *nThrSub = *nThrAdd = 0;
    *refresh = kingBucket(relP) != kingBucket(relC) || ((kingP & 7) >= 4) != ((kingC & 7) >= 4) || popc64(changed) > 4;
    if (*refresh) return SPX_OK;
    const int x = perspXor(c, kingC), flipColour = c == 0;
    // piece-square rows of the changed squares
    int squares[4], nS = 0;
    for (uint64_t m = changed; m; m &= m - 1) squares[nS++] = ctz64(m);
    for (int k = 0; k < nS; ++k) {
        const int f = squares[k];
        if (side[0].mail[f] != kNoPiece) psqSub[(*nPsqSub)++] = psqRow(c, side[0].mail[f], f, kingC);
        if (side[1].mail[f] != kNoPiece) psqAdd[(*nPsqAdd)++] = psqRow(c, side[1].mail[f], f, kingC);
    }
    // threat rows: passes of 64 lanes, lane = board << 5 | f index << 4 | slot
    for (int pass = 0; 2 * pass < nS; ++pass) {
        for (int lane = 0; lane < 64; ++lane) {
            const int b = lane >> 5, fi = 2 * pass + ((lane >> 4) & 1), slot = lane & 15;
            if (fi >= nS) continue;
            uint32_t d[2];
            deltaCandidates(tab, side[b].mail, side[b].occ, changed, squares[fi], slot, d[0], d[1]);
            for (uint32_t desc : d) {
                if (desc == kNoDesc) continue;
                const int32_t row = descRow(lut, tab, desc, x, flipColour);
                if (row < 0) continue;
                if (b == 0 && *nThrSub < 288) thrSub[(*nThrSub)++] = uint32_t(row);
                if (b == 1 && *nThrAdd < 288) thrAdd[(*nThrAdd)++] = uint32_t(row);
            }
        }
    }
    // pawn-pair rows: every pair that involves a pawn that left (parent board) or arrived (child board)
    for (int b = 0; b < 2; ++b) {
        const Side &self = side[b], &other = side[b ^ 1];
        const uint64_t own = self.pawns & (c ? self.white : ~self.white);
        const uint64_t otherOwn = other.pawns & (c ? other.white : ~other.white);
        // pawns of this board that the other board does not have (same square AND same colour)
        uint64_t moved = (own & ~otherOwn) | ((self.pawns & ~own) & ~(other.pawns & ~otherOwn));
        uint64_t done = 0;
        for (; moved; moved &= moved - 1) {
            const int a = ctz64(moved);
            done |= 1ull << a;
            const uint32_t idA = ppId(a ^ x, !((own >> a) & 1));
            for (uint64_t partners = self.pawns & ppMask(a) & ~done; partners; partners &= partners - 1) {
                const int q = ctz64(partners);
                const uint32_t row = ppRow(idA, ppId(q ^ x, !((own >> q) & 1)));
                if (b == 0 && *nThrSub < 288) thrSub[(*nThrSub)++] = row;
                if (b == 1 && *nThrAdd < 288) thrAdd[(*nThrAdd)++] = row;
            }
        }
    }
    return SPX_OK;
}


// Host evaluation of the SPX_HD functions behind SPX_ADJUST_WDL (the device runs the same source): Position::classicalMaterial
// of a record and wdl::normalizeScore of a score at that material. Test-only.
int spx_debug_wdl(const spx_packed_pos* pos, int32_t score, int32_t* material, int32_t* normalized) {
    if (!pos || !material || !normalized) {
        setError("spx_debug_wdl: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    const int n = std::min(popc64(pos->occupancy), 32);
    int32_t m = 0;
    for (int k = 0; k < n; ++k) m += classicalMaterialOfNibble((pos->pieces[k >> 1] >> ((k & 1) * 4)) & 0xF);
    *material = m;
    *normalized = wdlNormalize(score, m);
    return SPX_OK;
}

int spx_debug_datagen_rules(uint32_t* counters, int32_t norm_score, uint32_t ply, uint32_t* outcome,
                            const spx_packed_pos* pos, int* insufficient) {
    if (counters && outcome) {
        AdjCounters c{counters[0], counters[1], counters[2]};
        *outcome = adjudicate(c, norm_score, ply);
        counters[0] = c.win, counters[1] = c.loss, counters[2] = c.draw;
    }
    if (pos && insufficient) {
        uint64_t lo, hi;
        std::memcpy(&lo, pos->pieces, 8);
        std::memcpy(&hi, pos->pieces + 8, 8);
        *insufficient = insufficientMaterial(pos->occupancy, lo, hi) ? 1 : 0;
    }
    return SPX_OK;
}

}  // extern "C"
