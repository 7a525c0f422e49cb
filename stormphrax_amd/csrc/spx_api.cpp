// C-ABI implementation of libspx_nnue (include/spx_nnue.h): network validation, device context, launches, and the
// host helpers. There is deliberately NO CPU evaluation path here: without a HIP device spx_ctx_create fails with
// SPX_ERR_NO_DEVICE (the CPU oracle lives in oracle/ and is test infrastructure only).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/spx_nnue.h"
#include "spx_chess.h"
#include "spx_device_math.h"
#include "spx_internal.h"
#include "spx_kernels.h"

namespace spx {

namespace {
thread_local std::string t_lastError;
}

void setError(const std::string& msg) {
    t_lastError = msg;
}

int buildThreatLut(uint32_t* lut);  // spx_luts.cpp

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

struct spx_ctx {
    int device = 0;
    size_t maxBatch = 0;
    hipStream_t stream = nullptr;
    // weights
    int16_t* dPsqW = nullptr;
    uint8_t* dThrW = nullptr;
    int16_t* dFtBias = nullptr;
    int8_t* dL1W = nullptr;
    int32_t *dL1B = nullptr, *dL2W = nullptr, *dL2B = nullptr, *dL3W = nullptr, *dL3B = nullptr;
    uint32_t* dLut = nullptr;
    // scratch
    void* dPositions = nullptr;  // staging for the host-buffer entry point
    int32_t* dOut = nullptr;
    uint8_t* dFtOut = nullptr;
    uint8_t* dKingKeys = nullptr;  // counting-sort scratch
    uint8_t* dOutKeys = nullptr;
    uint32_t* dHist = nullptr;     // 64 words: counts + cursors
    uint32_t* dPerspOrder = nullptr;  // perspective ids grouped by king bucket
    uint32_t* dPosOrder = nullptr;    // position ids grouped by output bucket
    bool kingSortEnabled = true;   // SPX_NO_SORT=1 walks perspectives in input order (A/B of the L2-locality sort)
    bool smallL2Weights = false;   // every |l2W| < 2^23: the MLP tail may use 24-bit multiplies
    uint32_t ftGridCap = 0;
    // optional per-kernel timing (spx_profile_*): event triples recorded around the two kernels of each call
    std::vector<hipEvent_t> profEvents;  // 3 per recorded call: before ft, between, after mlp
    size_t profUsed = 0;
};

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
    if (flags & kFlagZstd) {
        // the reference inflates these with its vendored zstd decoder (nnue.cpp:219-247); no decoder is linked here
        setError("zstd-compressed network: decompress to the raw CBNF image first");
        return SPX_ERR_BAD_NET;
    }
    name.assign(reinterpret_cast<const char*>(h + 16), nameLen < 48 ? nameLen : 48);
    return SPX_OK;
}

// Threat table relayout: +128 bias (so widening is a zero-extend) and per-lane column interleave: the 16 bytes lane
// l loads at offset 16*l are columns 8l..8l+7 followed by 512+8l..512+8l+7 (pairwise partners share a lane).
void relayoutThreatRow(const int8_t* src, uint8_t* dst) {
    for (uint32_t l = 0; l < 64; ++l) {
        for (uint32_t j = 0; j < 8; ++j) {
            dst[16 * l + j] = uint8_t(src[8 * l + j]) ^ 0x80u;
            dst[16 * l + 8 + j] = uint8_t(src[512 + 8 * l + j]) ^ 0x80u;
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
    if (nbytes < kNetFileBytes) {
        setError("Default network too small? " + std::to_string(nbytes - kHeaderBytes) + " < " +
                 std::to_string(kNetFileBytes - kHeaderBytes));  // nnue.cpp:252-255
        return SPX_ERR_BAD_NET;
    }
    auto net = std::make_unique<spx_net>();
    net->blob.assign(static_cast<const unsigned char*>(blob), static_cast<const unsigned char*>(blob) + kNetFileBytes);
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

int spx_ctx_create(const spx_net* net, int device, size_t max_batch, spx_ctx** out) {
    if (!net || !out || max_batch == 0 || max_batch > (1ull << 30)) {
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
    auto ctx = std::make_unique<spx_ctx>();
    ctx->device = device;
    ctx->maxBatch = max_batch;
    SPX_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));

    int rc;
    const unsigned char* b = net->blob.data();
    if ((rc = uploadArray(ctx->dPsqW, b + kOffPsqW, kPsqWBytes, ctx->stream)) != SPX_OK) return rc;
    {
        std::vector<uint8_t> thr(kThreatWBytes);
        for (uint32_t r = 0; r < kThreatRows; ++r) {
            relayoutThreatRow(net->threatW() + size_t(r) * kL1, thr.data() + size_t(r) * kL1);
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
    if ((rc = uploadArray(ctx->dL2W, b + kOffL2W, kL2WBytes, ctx->stream)) != SPX_OK) return rc;
    if ((rc = uploadArray(ctx->dL2B, b + kOffL2B, kL2BBytes, ctx->stream)) != SPX_OK) return rc;
    if ((rc = uploadArray(ctx->dL3W, b + kOffL3W, kL3WBytes, ctx->stream)) != SPX_OK) return rc;
    if ((rc = uploadArray(ctx->dL3B, b + kOffL3B, kL3BBytes, ctx->stream)) != SPX_OK) return rc;
    {
        uint32_t lut[kLutWords];
        if (buildThreatLut(lut) != int(kThreatOnlyRows)) {
            setError("internal: threat LUT does not cover 59808 features");
            return SPX_ERR_BAD_NET;
        }
        if ((rc = uploadArray(ctx->dLut, lut, sizeof(lut), ctx->stream)) != SPX_OK) return rc;
    }
    SPX_HIP(hipMalloc(&ctx->dPositions, max_batch * sizeof(spx_packed_pos)));
    SPX_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->dOut), max_batch * sizeof(int32_t)));
    SPX_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->dFtOut), max_batch * size_t(kL1)));
    SPX_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->dKingKeys), max_batch * 2));
    SPX_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->dOutKeys), max_batch));
    SPX_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->dHist), 64 * sizeof(uint32_t)));
    SPX_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->dPerspOrder), max_batch * 2 * sizeof(uint32_t)));
    SPX_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->dPosOrder), max_batch * sizeof(uint32_t)));
    if (const char* env = std::getenv("SPX_NO_SORT")) ctx->kingSortEnabled = env[0] == '0';
    {
        const int32_t* w = reinterpret_cast<const int32_t*>(b + kOffL2W);
        bool small = true;
        for (size_t i = 0; i < kL2WBytes / 4 && small; ++i) small = w[i] > -(1 << 23) && w[i] < (1 << 23);
        ctx->smallL2Weights = small;
    }

    hipDeviceProp_t prop;
    SPX_HIP(hipGetDeviceProperties(&prop, device));
    // persistent-ish grid: 8 workgroups (of 4 waves) per CU, grid-stride over perspectives
    ctx->ftGridCap = uint32_t(prop.multiProcessorCount) * 8u;
    *out = ctx.release();
    return SPX_OK;
}

void spx_ctx_destroy(spx_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    void* ptrs[] = {ctx->dPsqW, ctx->dThrW, ctx->dFtBias, ctx->dL1W, ctx->dL1B, ctx->dL2W,  ctx->dL2B,
                    ctx->dL3W,  ctx->dL3B, ctx->dLut,    ctx->dPositions, ctx->dOut, ctx->dFtOut,
                    ctx->dKingKeys, ctx->dOutKeys, ctx->dHist, ctx->dPerspOrder, ctx->dPosOrder};
    for (void* p : ptrs) {
        if (p) (void)hipFree(p);
    }
    for (hipEvent_t e : ctx->profEvents) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int spx_eval_full_device(spx_ctx* ctx, const void* d_positions, size_t n, void* d_out, void* stream) {
    if (!ctx || (n && (!d_positions || !d_out))) {
        setError("spx_eval_full_device: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    if (n > ctx->maxBatch) {
        setError("batch of " + std::to_string(n) + " exceeds context capacity " + std::to_string(ctx->maxBatch));
        return SPX_ERR_CAPACITY;
    }
    if (n == 0) {
        return SPX_OK;
    }
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    hipEvent_t* ev = nullptr;
    if (ctx->profUsed + 3 <= ctx->profEvents.size()) {
        ev = &ctx->profEvents[ctx->profUsed];
        ctx->profUsed += 3;
        SPX_HIP(hipEventRecord(ev[0], s));
    }
    {
        SortParams sp{};
        sp.positions = static_cast<const uint64_t*>(d_positions);
        sp.nPositions = uint32_t(n);
        sp.kingKeys = ctx->dKingKeys;
        sp.outKeys = ctx->dOutKeys;
        sp.hist = ctx->dHist;
        sp.perspOrder = ctx->dPerspOrder;
        sp.posOrder = ctx->dPosOrder;
        SPX_HIP(launchSort(sp, s));
    }
    FtParams fp{};
    fp.positions = d_positions;
    fp.nPositions = uint32_t(n);
    fp.order = ctx->kingSortEnabled ? ctx->dPerspOrder : nullptr;
    fp.psqW = ctx->dPsqW;
    fp.thrW = ctx->dThrW;
    fp.ftBias = ctx->dFtBias;
    fp.lut = ctx->dLut;
    fp.ftOut = ctx->dFtOut;
    const uint32_t wavesPerBlock = ftWavesPerBlock();
    uint32_t blocks = uint32_t((2 * n + wavesPerBlock - 1) / wavesPerBlock);
    if (blocks > ctx->ftGridCap) blocks = ctx->ftGridCap;
    blocks = (blocks + 7u) & ~7u;  // whole multiples of the 8 XCDs
    SPX_HIP(launchFt(fp, blocks, s));
    if (ev) SPX_HIP(hipEventRecord(ev[1], s));

    MlpParams mp{};
    mp.nPositions = uint32_t(n);
    mp.posOrder = ctx->dPosOrder;
    mp.hist = ctx->dHist;
    mp.ftOut = ctx->dFtOut;
    mp.l1W = ctx->dL1W;
    mp.l1B = ctx->dL1B;
    mp.l2W = ctx->dL2W;
    mp.l2B = ctx->dL2B;
    mp.l3W = ctx->dL3W;
    mp.l3B = ctx->dL3B;
    mp.out = static_cast<int32_t*>(d_out);
    SPX_HIP(launchMlp(mp, ctx->smallL2Weights, s));
    if (ev) SPX_HIP(hipEventRecord(ev[2], s));
    return SPX_OK;
}

int spx_profile_begin(spx_ctx* ctx, size_t max_calls) {
    if (!ctx) {
        setError("spx_profile_begin: null context");
        return SPX_ERR_INVALID_ARG;
    }
    SPX_HIP(hipSetDevice(ctx->device));
    while (ctx->profEvents.size() < max_calls * 3) {
        hipEvent_t e;
        SPX_HIP(hipEventCreate(&e));
        ctx->profEvents.push_back(e);
    }
    ctx->profUsed = 0;
    return SPX_OK;
}

int spx_profile_end(spx_ctx* ctx, double* ft_ms, double* mlp_ms, size_t* calls) {
    if (!ctx || !ft_ms || !mlp_ms || !calls) {
        setError("spx_profile_end: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    SPX_HIP(hipSetDevice(ctx->device));
    *ft_ms = *mlp_ms = 0.0;
    *calls = ctx->profUsed / 3;
    for (size_t i = 0; i + 2 < ctx->profUsed + 0 && i < ctx->profUsed; i += 3) {
        float a = 0.f, b = 0.f;
        SPX_HIP(hipEventSynchronize(ctx->profEvents[i + 2]));
        SPX_HIP(hipEventElapsedTime(&a, ctx->profEvents[i], ctx->profEvents[i + 1]));
        SPX_HIP(hipEventElapsedTime(&b, ctx->profEvents[i + 1], ctx->profEvents[i + 2]));
        *ft_ms += a;
        *mlp_ms += b;
    }
    ctx->profUsed = ctx->profEvents.size();  // stop recording until the next spx_profile_begin
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

int spx_eval_full(spx_ctx* ctx, const spx_packed_pos* positions, size_t n, int32_t* out) {
    if (!ctx || (n && (!positions || !out))) {
        setError("spx_eval_full: null argument");
        return SPX_ERR_INVALID_ARG;
    }
    if (n > ctx->maxBatch) {
        setError("batch of " + std::to_string(n) + " exceeds context capacity " + std::to_string(ctx->maxBatch));
        return SPX_ERR_CAPACITY;
    }
    if (n == 0) return SPX_OK;
    SPX_HIP(hipSetDevice(ctx->device));
    SPX_HIP(hipMemcpyAsync(ctx->dPositions, positions, n * sizeof(spx_packed_pos), hipMemcpyHostToDevice, ctx->stream));
    const int rc = spx_eval_full_device(ctx, ctx->dPositions, n, ctx->dOut, ctx->stream);
    if (rc != SPX_OK) return rc;
    SPX_HIP(hipMemcpyAsync(out, ctx->dOut, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    SPX_HIP(hipStreamSynchronize(ctx->stream));
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

int spx_random_positions(uint64_t seed, size_t count, int min_ply, int max_ply, int dfrc_every, spx_packed_pos* out) {
    if (!out || min_ply < 0 || max_ply < min_ply || max_ply > 600) {
        setError("spx_random_positions: invalid argument");
        return SPX_ERR_INVALID_ARG;
    }
    randomPositions(seed, count, min_ply, max_ply, dfrc_every, out);
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
            if ((piece[lane] >> 1) == 0) pawnsBb |= 1ull << lane;
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

}  // extern "C"
