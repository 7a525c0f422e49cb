// Gather-ceiling probe (measurement infrastructure): what does this chip sustain for the feature
// transformer's row gather when NOTHING but the loads is done?
//
// spx_probe_lists_kernel runs the product kernel's own traversal and list construction (spx_ft_device.h: decodeBoard,
// buildFullLists) on the king-bucket-sorted perspective order and writes every perspective's row-offset lists to HBM.
// spx_probe_gather_kernel<kPath, kRing, kWaves> then replays exactly those lists - same grid, same XCD round-robin
// traversal, same rows in the same order - with loads only: one v_xor_b32 per loaded dword keeps the loads alive, 8 bytes
// per lane are written where the product kernel writes its activations. Two memory paths:
//   kPath 0  global_load_dwordx4 into VGPRs, bursts of 8 x 1 KiB per wave (the product kernel's loadGatherRow form)
//   kPath 1  LDS-DMA: global_load_lds_dwordx4 (1 KiB per wave instruction, no VGPR landing buffers) into a per-wave ring
//            of kRing 1 KiB slots, two halves in flight alternately, consumed with ds_read_b128
// The difference between the product kernel's time and the probe's is what extraction, widening and their scheduling cost
// on top of the memory path; the probe's time is the ceiling a better kernel could approach with the same bytes.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "spx_ft_device.h"
#include "spx_probe.h"

namespace spx {

namespace {

constexpr uint32_t kListStride = 328;  // words per perspective: [0] nThr, [1] nPsq, [2] perspective id, [4 .. 292) u8 rows, [292 .. 324) i16 rows
constexpr uint32_t kListThr = 4, kListPsq = 4 + kU8Cap;

// the product kernel's traversal (spx_ft_kernel): perspective order dealt to the XCDs in round-robin chunks
struct Traversal {
    uint32_t chunkShift, myItems, stride, first, xcd;
    __device__ Traversal(uint32_t nPersp, uint32_t wave) {
        xcd = blockIdx.x & 7;
        const uint32_t blockInXcd = blockIdx.x >> 3, blocksPerXcd = gridDim.x >> 3;
        stride = blocksPerXcd * kWavesPerBlock;
        chunkShift = nPersp >= 64u * SPX_FT_CHUNK ? uint32_t(__builtin_ctz(SPX_FT_CHUNK)) : 0u;
        const uint32_t nChunks = (nPersp + (1u << chunkShift) - 1) >> chunkShift;
        myItems = ((nChunks + 7 - xcd) / 8) << chunkShift;
        first = blockInXcd * kWavesPerBlock + wave;
    }
    __device__ uint32_t item(uint32_t t) const {
        return ((((t >> chunkShift) * 8 + xcd)) << chunkShift) + (t & ((1u << chunkShift) - 1));
    }
};

__device__ __forceinline__ constexpr int vmcntImm(int n) {  // s_waitcnt vmcnt(n) only (gfx9 encoding: vmcnt = [15:14][3:0])
    return (n & 0xF) | ((n >> 4) << 14) | (0x7 << 4) | (0xF << 8);
}

// kN consecutive 1 KiB ring slots -> this lane's 16 bytes of each, one LDS wait for all of them
template <int kN>
__device__ __forceinline__ void readSlots(uint32_t ldsAddr, u32x4 (&w)[kN]) {
    static_assert(kN == 1 || kN == 2 || kN == 4 || kN == 8, "ring halves of 1, 2, 4 or 8 slots");
    if constexpr (kN == 1) {
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w[0]) : "v"(ldsAddr) : "memory");
    } else if constexpr (kN == 2) {
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(w[0]), "=&v"(w[1]) : "v"(ldsAddr) : "memory");
    } else if constexpr (kN == 4) {
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                     "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]) : "v"(ldsAddr) : "memory");
    } else {
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\t"
                     "ds_read_b128 %3, %8 offset:3072\n\tds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\t"
                     "ds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7])
                     : "v"(ldsAddr) : "memory");
    }
}

}  // namespace

__global__ __launch_bounds__(64 * kWavesPerBlock, SPX_FT_WAVES_PER_SIMD) void spx_probe_lists_kernel(ProbeParams p) {
    __shared__ uint32_t sLut[kLutWords];
    __shared__ uint32_t sThr[kWavesPerBlock][kU8Cap];
    __shared__ uint32_t sPsq[kWavesPerBlock][kPsqCap];
    __shared__ uint64_t sPseudo[kDeltaPseudoWords];
    for (int i = threadIdx.x; i < kLutWords; i += blockDim.x) sLut[i] = p.t.lut[i];
    for (int i = threadIdx.x; i < kDeltaPseudoWords; i += blockDim.x) sPseudo[i] = p.t.deltaTab[kDeltaRayWords + i];
    __syncthreads();
    const uint32_t lane = laneId(), wave = threadIdx.x >> 6;
    const uint32_t nPersp = p.nPositions * 2;
    const Traversal tr(nPersp, wave);
    for (uint32_t t = tr.first; t < tr.myItems; t += tr.stride) {
        const uint32_t it = tr.item(t);
        if (it >= nPersp) continue;
        const uint32_t q = __builtin_amdgcn_readfirstlane(p.order ? p.order[it] : it);
        const uint8_t* rec = reinterpret_cast<const uint8_t*>(p.positions) + size_t(q >> 1) * 32;
        const LaneBoard board = decodeBoard(rec, lane);
        uint32_t nPsq, nThr;
        // (near-compact rows are listed as wide rows here: the probe replays row fetches, not remainders)
        buildFullLists<false>(board, int(q & 1), lane, sLut, sPsq[wave], sThr[wave], nPsq, nThr, sPseudo);
        uint32_t* out = p.lists + size_t(it) * kListStride;
        if (lane == 0) {
            out[0] = nThr;
            out[1] = nPsq;
            out[2] = q;  // the sort's order within a bucket varies from run to run: results are stored by perspective id
        }
        for (uint32_t i = lane; i < nThr; i += 64) out[kListThr + i] = sThr[wave][i];
        if (lane < nPsq) out[kListPsq + lane] = sPsq[wave][lane];
        __builtin_amdgcn_wave_barrier();
    }
}

template <int kPath, int kRing, int kWaves>
__global__ __launch_bounds__(64 * kWavesPerBlock, kWaves) void spx_probe_gather_kernel(ProbeParams p) {
    __shared__ uint32_t sThr[kWavesPerBlock][kU8Cap];
    __shared__ uint32_t sPsq[kWavesPerBlock][kPsqCap];
    __shared__ __align__(16) uint8_t sRing[kPath == 1 ? kWavesPerBlock : 1][kPath == 1 ? kRing * 1024 : 16];
    const uint32_t lane = laneId(), wave = threadIdx.x >> 6;
    const uint32_t nPersp = p.nPositions * 2;
    const Traversal tr(nPersp, wave);
    const uint8_t* thrTable = p.t.thrW;
    const uint8_t* psqTable = reinterpret_cast<const uint8_t*>(p.t.psqW);
    const uint32_t laneOff = 16 * lane;
    for (uint32_t t = tr.first; t < tr.myItems; t += tr.stride) {
        const uint32_t it = tr.item(t);
        if (it >= nPersp) continue;
        const uint32_t* in = p.lists + size_t(it) * kListStride;
        const uint32_t nThr = __builtin_amdgcn_readfirstlane(in[0]), nPsq = __builtin_amdgcn_readfirstlane(in[1]);
        const uint32_t q = __builtin_amdgcn_readfirstlane(in[2]);
#pragma unroll 1
        for (uint32_t i = lane; i < nThr; i += 64) sThr[wave][i] = in[kListThr + i];
        if (lane < nPsq) sPsq[wave][lane] = in[kListPsq + lane];
        __builtin_amdgcn_wave_barrier();
        u32x4 x = {0, 0, 0, 0};
        if constexpr (kPath == 0) {
            uint32_t i = 0;
            for (; i + 8 <= nThr; i += 8) {  // 8 x 1 KiB wave loads in flight, as in gatherFull
                u32x4 w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) w[u] = loadGatherRow(thrTable, sThr[wave][i + u], laneOff);
#pragma unroll
                for (int u = 0; u < 8; ++u) x ^= w[u];
            }
            for (; i + 2 <= nThr; i += 2) {
                const u32x4 w0 = loadGatherRow(thrTable, sThr[wave][i], laneOff);
                const u32x4 w1 = loadGatherRow(thrTable, sThr[wave][i + 1], laneOff);
                x ^= w0 ^ w1;
            }
            if (i < nThr) x ^= loadGatherRow(thrTable, sThr[wave][i], laneOff);
#pragma unroll 1
            for (uint32_t k = 0; k < nPsq; ++k) {
                x ^= loadGatherRow(psqTable, sPsq[wave][k], laneOff);
                x ^= loadGatherRow(psqTable, sPsq[wave][k], laneOff + 1024);
            }
        } else {
            // ring of kRing slots in two halves: while half h is consumed, half h ^ 1 is in flight. The DMA writes
            // LDS at M0-base + 16 * lane (lane-linear), i.e. exactly the 16 bytes this lane then reads back.
            constexpr int kHalf = kRing / 2;
            static_assert(kRing >= 2 && kRing % 2 == 0, "ring = two halves");
            uint8_t* ring = sRing[kPath == 1 ? wave : 0];
            const uint32_t nBatches = (nThr + kHalf - 1) / kHalf;
            auto issue = [&](uint32_t batch) {
                const uint32_t h = batch & 1;
#pragma unroll
                for (int u = 0; u < kHalf; ++u) {
                    // rows past the end of the list re-fetch the last row (wave-uniform count, branch-free batches)
                    const uint32_t idx = min(batch * kHalf + u, nThr - 1);
                    const uint8_t* src = thrTable + size_t(sThr[wave][idx] + laneOff);
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)src,
                        (__attribute__((address_space(3))) void*)(ring + (h * kHalf + u) * 1024), 16, 0, 0);
                }
            };
            if (nBatches) issue(0);
            for (uint32_t b = 0; b < nBatches; ++b) {
                if (b + 1 < nBatches) {
                    issue(b + 1);
                    __builtin_amdgcn_s_waitcnt(vmcntImm(kHalf));  // batch b has landed, batch b + 1 may be in flight
                } else {
                    __builtin_amdgcn_s_waitcnt(vmcntImm(0));
                }
                __builtin_amdgcn_wave_barrier();
                // The reads are inline asm on purpose: hipcc orders every ds_read of memory an LDS-DMA may write behind
                // s_waitcnt vmcnt(0), which would drain the half in flight too; the vmcnt above is the real dependency.
                const uint32_t h = b & 1;
                const uint32_t base = uint32_t(reinterpret_cast<uintptr_t>(ring)) + h * kHalf * 1024 + laneOff;
                u32x4 w[kHalf];
                readSlots<kHalf>(base, w);
#pragma unroll
                for (int u = 0; u < kHalf; ++u) {
                    const bool valid = b * kHalf + u < nThr;  // wave-uniform
#pragma unroll
                    for (int d = 0; d < 4; ++d) x[d] ^= valid ? w[u][d] : 0u;
                }
                __builtin_amdgcn_wave_barrier();  // the half is consumed before batch b + 2 overwrites it
            }
#pragma unroll 1
            for (uint32_t k = 0; k < nPsq; ++k) {
                x ^= loadGatherRow(psqTable, sPsq[wave][k], laneOff);
                x ^= loadGatherRow(psqTable, sPsq[wave][k], laneOff + 1024);
            }
        }
        u32x2 o;
        o[0] = x[0] ^ x[2];
        o[1] = x[1] ^ x[3];
        *reinterpret_cast<u32x2*>(p.sink + size_t(q) * 512 + 8 * lane) = o;
        __builtin_amdgcn_wave_barrier();
    }
}

hipError_t launchProbeLists(const ProbeParams& p, uint32_t gridBlocks, hipStream_t stream) {
    hipLaunchKernelGGL(spx_probe_lists_kernel, dim3(gridBlocks), dim3(64 * kWavesPerBlock), 0, stream, p);
    return hipGetLastError();
}

static const ProbeVariant kVariants[] = {
    {"global_load_dwordx4, bursts of 8, 5 waves/SIMD", 0, 0, 5}, {"global_load_dwordx4, bursts of 8, 6 waves/SIMD", 0, 0, 6},
    {"global_load_dwordx4, bursts of 8, 8 waves/SIMD", 0, 0, 8}, {"LDS-DMA ring 4 KiB/wave, 5 waves/SIMD", 1, 4, 5},
    {"LDS-DMA ring 8 KiB/wave, 4 waves/SIMD", 1, 8, 4},         {"LDS-DMA ring 4 KiB/wave, 7 waves/SIMD", 1, 4, 7},
    {"LDS-DMA ring 2 KiB/wave, 8 waves/SIMD", 1, 2, 8},         {"LDS-DMA ring 16 KiB/wave, 2 waves/SIMD", 1, 16, 2},
};

int probeVariantCount() {
    return int(sizeof(kVariants) / sizeof(kVariants[0]));
}
const ProbeVariant& probeVariant(int i) {
    return kVariants[i];
}

hipError_t launchProbeGather(const ProbeParams& p, int variant, uint32_t gridBlocks, hipStream_t stream) {
    const dim3 grid(gridBlocks), block(64 * kWavesPerBlock);
    switch (variant) {
    case 0: hipLaunchKernelGGL((spx_probe_gather_kernel<0, 2, 5>), grid, block, 0, stream, p); break;
    case 1: hipLaunchKernelGGL((spx_probe_gather_kernel<0, 2, 6>), grid, block, 0, stream, p); break;
    case 2: hipLaunchKernelGGL((spx_probe_gather_kernel<0, 2, 8>), grid, block, 0, stream, p); break;
    case 3: hipLaunchKernelGGL((spx_probe_gather_kernel<1, 4, 5>), grid, block, 0, stream, p); break;
    case 4: hipLaunchKernelGGL((spx_probe_gather_kernel<1, 8, 4>), grid, block, 0, stream, p); break;
    case 5: hipLaunchKernelGGL((spx_probe_gather_kernel<1, 4, 7>), grid, block, 0, stream, p); break;
    case 6: hipLaunchKernelGGL((spx_probe_gather_kernel<1, 2, 8>), grid, block, 0, stream, p); break;
    case 7: hipLaunchKernelGGL((spx_probe_gather_kernel<1, 16, 2>), grid, block, 0, stream, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace spx
