// Synthetic CBNF network generator.
//
// The reference's default net (net093_255_128_q6, /root/reference/network.txt:1) is downloaded at build time and is
// not obtainable offline, so parity and benchmarks run on a random-weight net with a valid header
// (/root/reference/src/eval/header.h:38-52) and the exact array order/sizes of preprocess/permute.cpp:33-56.
// The stream comes from the repo's own splitmix64 so the same 89 381 984-byte file is regenerated bit-identically
// on any box (tests pin its FNV-1a digest).
//
// The file produced here is in LOGICAL column order (what a trainer writes); the reference's build permutes FT
// columns per ISA before embedding (permute.cpp:121-126) - the GPU path consumes the logical order.
#include <cstring>

#include "spx_arch.h"
#include "spx_internal.h"

namespace spx {

namespace {
struct SplitMix64 {
    uint64_t s;
    inline uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    // uniform integer in [lo, hi]; modulo bias is irrelevant for synthetic weights
    inline int32_t range(int32_t lo, int32_t hi) {
        const uint32_t span = uint32_t(hi - lo) + 1u;
        return lo + int32_t(uint32_t(next() >> 32) % span);
    }
};

struct Ranges {
    int psq, thr, biasLo, biasHi, l1w, l1b, l2w, l2b, l3w, l3b;
};

// preset 0 "tame": no i32 wrap in L2/L3, raw evals inside +-24999 (oracle `raweval` clamp safe)
// preset 1 "wild": L3 products and t*t wrap in i32
// preset 2 "extreme": additionally the i16 accumulators wrap and |l2W| reaches 2^30 (general 32-bit multiply path)
// l3B is 0 in the wrapping presets: the reference's final `l3Biases[b] + hsum(s)` (multilayer.h:446) is a SCALAR
// signed add - overflow there is UB (the AVX2 build was observed to widen it to i64), unlike every SIMD op on the
// path, which wraps by definition. Keeping that one add overflow-free keeps the reference's result well-defined.
constexpr Ranges kPresets[3] = {
    {48, 12, -64, 191, 127, 4096, 8, 4096, 8, 65536},
    {48, 12, -64, 191, 127, 4096, 64, 1 << 20, 64, 0},
    {6000, 127, -32768, 32767, 127, 1 << 22, 1 << 30, 1 << 30, 1 << 24, 0},
};

// preset 3 "realistic": the SHAPE a trained QA = 255 net has (arch.h:36-50: i16 piece-square weights quantised at 255, i8
// threat weights, i8 L1 weights at 128) instead of uniform noise - two-sided geometric ("Laplace") magnitudes with heavy
// tails, so that piece-square rows split three ways as a real net's would: rows that fit i8 entirely, rows with a handful
// of weights beyond +-127 (the near-compact path), and rows densely beyond it (2 KiB path). Integer-only sampling, so the
// file is bit-identical on every box: magnitude = step * (leading zeros of a random word) + uniform[0, step), i.e. a
// staircase exponential with half-life `step`.
int32_t laplace(SplitMix64& rng, int32_t step, int32_t clip) {
    const uint64_t a = rng.next(), b = rng.next();
    const int32_t zeros = (a >> 1) ? __builtin_clzll(a >> 1) - 1 : 63;  // geometric: P(zeros >= k) = 2^-k
    int32_t mag = step * zeros + int32_t(uint32_t(b >> 32) % uint32_t(step));
    if (mag > clip) mag = clip;
    return (b & 1) ? -mag : mag;
}

template <typename T>
void fill(SplitMix64& rng, unsigned char* dst, size_t count, int32_t lo, int32_t hi) {
    T* p = reinterpret_cast<T*>(dst);
    for (size_t i = 0; i < count; ++i) {
        p[i] = static_cast<T>(rng.range(lo, hi));
    }
}
}  // namespace

size_t synthNetBytes() {
    return kNetFileBytes;
}

bool synthNet(uint64_t seed, int preset, void* buf, size_t n) {
    if (!buf || n < kNetFileBytes || preset < 0 || preset > 3) {
        return false;
    }
    auto* out = static_cast<unsigned char*>(buf);
    const Ranges& r = kPresets[preset == 3 ? 0 : preset];  // "realistic": the tame ranges beyond L1 (no i32 wraps)

    // ---- header ----
    std::memset(out, 0, kHeaderBytes);
    std::memcpy(out, "CBNF", 4);
    const uint16_t version = 1;
    const uint16_t flags = kFlagMirrored | kFlagMergedKings | kFlagPairwise;
    std::memcpy(out + 4, &version, 2);
    std::memcpy(out + 6, &flags, 2);
    out[8] = 0;  // padding
    out[9] = kArchId;
    out[10] = kActivationId;
    const uint16_t hidden = kL1;
    std::memcpy(out + 11, &hidden, 2);
    out[13] = uint8_t(kInputBuckets | 0x80);  // bit 7 = threat inputs (nnue.cpp:153)
    out[14] = uint8_t(kOutputBuckets);
    char name[48] = {};
    static const char* kNames[4] = {"spx_synth_tame", "spx_synth_wild", "spx_synth_extreme", "spx_synth_realistic"};
    std::strncpy(name, kNames[preset], sizeof(name) - 1);
    out[15] = uint8_t(std::strlen(name));
    std::memcpy(out + 16, name, 48);

    SplitMix64 rng{seed ^ (0xC0FFEEull * uint64_t(preset + 1))};
    if (preset == 3) {
        // piece-square rows: 80 % half-life 12 (about 0.7 weights per row beyond +-127: half of these rows fit i8, the
        // rest carry 1-5 outliers), 15 % half-life 28 (~45 per row: wide), 5 % half-life 70 (dense: queen-like features)
        auto* psq = reinterpret_cast<int16_t*>(out + kOffPsqW);
        for (uint32_t row = 0; row < kPsqRows; ++row) {
            const uint32_t kind = uint32_t(rng.next() >> 32) % 100u;
            const int32_t step = kind < 80 ? 12 : (kind < 95 ? 28 : 70);
            for (uint32_t j = 0; j < kL1; ++j) psq[size_t(row) * kL1 + j] = int16_t(laplace(rng, step, 8000));
        }
        auto* thr = reinterpret_cast<int8_t*>(out + kOffThreatW);
        for (size_t i = 0; i < size_t(kThreatRows) * kL1; ++i) thr[i] = int8_t(laplace(rng, 6, 127));
        fill<int16_t>(rng, out + kOffFtBias, kL1, r.biasLo, r.biasHi);
        auto* l1 = reinterpret_cast<int8_t*>(out + kOffL1W);
        for (size_t i = 0; i < kL1WBytes; ++i) l1[i] = int8_t(laplace(rng, 14, 127));
    } else {
        fill<int16_t>(rng, out + kOffPsqW, size_t(kPsqRows) * kL1, -r.psq, r.psq);
        fill<int8_t>(rng, out + kOffThreatW, size_t(kThreatRows) * kL1, -r.thr, r.thr);
        fill<int16_t>(rng, out + kOffFtBias, kL1, r.biasLo, r.biasHi);
        fill<int8_t>(rng, out + kOffL1W, kL1WBytes, -r.l1w, r.l1w);
    }
    fill<int32_t>(rng, out + kOffL1B, kOutputBuckets * kL2, -r.l1b, r.l1b);
    fill<int32_t>(rng, out + kOffL2W, size_t(kOutputBuckets) * kL2Full * kL3, -r.l2w, r.l2w);
    fill<int32_t>(rng, out + kOffL2B, kOutputBuckets * kL3, -r.l2b, r.l2b);
    fill<int32_t>(rng, out + kOffL3W, kOutputBuckets * kL3, -r.l3w, r.l3w);
    fill<int32_t>(rng, out + kOffL3B, kOutputBuckets, -r.l3b, r.l3b);
    return true;
}

uint64_t fnv1a64(const void* data, size_t n) {
    const auto* p = static_cast<const unsigned char*>(data);
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; ++i) {
        h = (h ^ p[i]) * 0x100000001b3ull;
    }
    return h;
}

}  // namespace spx
