// Compile-time shape of the Stormphrax 8.0.2 network, as plain constants shared by host and device code.
//
// Restates /root/reference/src/eval/arch.h:33-82 (L1=1024, L2=32, L3=64, 16 mirrored king buckets with merged
// kings, threat + pawn-pair inputs, 8 material output buckets, scale 400) and the on-disk layout of
// /root/reference/src/eval/header.h:38-52 + preprocess/permute.cpp:33-56.
#pragma once

#include <cstddef>
#include <cstdint>

namespace spx {

constexpr uint32_t kL1 = 1024;           // arch.h:41  kL1Size (accumulator width per perspective)
constexpr uint32_t kPairs = kL1 / 2;     // pairwise activation output per perspective (multilayer.h:103)
constexpr uint32_t kL2 = 32;             // arch.h:42
constexpr uint32_t kL2Full = 64;         // dual activation doubles L1 outputs (multilayer.h:66)
constexpr uint32_t kL3 = 64;             // arch.h:43
constexpr uint32_t kOutputBuckets = 8;   // arch.h:69 MaterialCount<8>
constexpr uint32_t kInputBuckets = 16;   // arch.h:53-65
constexpr uint32_t kPsqInputSize = 704;  // psq.h:319 merged kings: 384 own(+both kings) + 320 enemy
constexpr uint32_t kPsqRows = kInputBuckets * kPsqInputSize;  // 11264
constexpr uint32_t kPpRows = 96 * 95 / 2;                     // threats.h:31  4560 pawn-pair rows
constexpr uint32_t kThreatOnlyRows = 59808;                   // threats.h:32
constexpr uint32_t kThreatRows = kThreatOnlyRows + kPpRows;   // 64368 rows in the threat table
constexpr int32_t kScale = 400;          // arch.h:50
constexpr int32_t kQBits = 6;            // multilayer.h:89 kQuantBits
constexpr int32_t kFtQBits = 8;          // arch.h:36
constexpr int32_t kFtScaleBits = 7;      // arch.h:39
constexpr int32_t kL1Shift = 2;          // multilayer.h:162: kShift = 16+6+6-7-8-8-7 = -2  => arithmetic >> 2

// ---- file layout (little endian) ----
constexpr size_t kHeaderBytes = 64;
constexpr size_t kPsqWBytes = size_t(kPsqRows) * kL1 * 2;         // i16 [11264][1024]
constexpr size_t kThreatWBytes = size_t(kThreatRows) * kL1;       // i8  [64368][1024]
constexpr size_t kFtBiasBytes = kL1 * 2;                          // i16 [1024]
constexpr size_t kL1WBytes = size_t(kOutputBuckets) * kL1 * kL2;  // i8  [8][256][32][4]
constexpr size_t kL1BBytes = kOutputBuckets * kL2 * 4;            // i32 [8][32]
constexpr size_t kL2WBytes = size_t(kOutputBuckets) * kL2Full * kL3 * 4;  // i32 [8][64][64]
constexpr size_t kL2BBytes = kOutputBuckets * kL3 * 4;            // i32 [8][64]
constexpr size_t kL3WBytes = kOutputBuckets * kL3 * 4;            // i32 [8][64]
constexpr size_t kL3BBytes = kOutputBuckets * 4;                  // i32 [8]

constexpr size_t kOffPsqW = kHeaderBytes;
constexpr size_t kOffThreatW = kOffPsqW + kPsqWBytes;
constexpr size_t kOffFtBias = kOffThreatW + kThreatWBytes;
constexpr size_t kOffL1W = kOffFtBias + kFtBiasBytes;
constexpr size_t kOffL1B = kOffL1W + kL1WBytes;
constexpr size_t kOffL2W = kOffL1B + kL1BBytes;
constexpr size_t kOffL2B = kOffL2W + kL2WBytes;
constexpr size_t kOffL3W = kOffL2B + kL2BBytes;
constexpr size_t kOffL3B = kOffL3W + kL3WBytes;
constexpr size_t kNetFileBytes = kOffL3B + kL3BBytes;  // 89 381 984

static_assert(kNetFileBytes == 89381984, "file size must match the reference's default net");

// header flags (header.h:29-35)
constexpr uint16_t kFlagZstd = 0x0001;
constexpr uint16_t kFlagMirrored = 0x0002;
constexpr uint16_t kFlagMergedKings = 0x0004;
constexpr uint16_t kFlagPairwise = 0x0008;
constexpr uint8_t kArchId = 5;        // multilayer.h:51: 2 + dual(1) + 2*skipL2(1)
constexpr uint8_t kActivationId = 0;  // activation.h:99 ClippedReLU

// pieces: type<<1 | colour, black = 0, white = 1 (core.h:336-350); 12 = none
constexpr uint8_t kNoPiece = 12;
constexpr int32_t kScoreWin = 25000;  // core.h:708; static evals are clamped to +-(kScoreWin - 1) (eval.cpp:26,63)

}  // namespace spx
