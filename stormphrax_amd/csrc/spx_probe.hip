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
    const uint32_t nPersp = p.nItems ? p.nItems : p.nPositions * 2;
    const Traversal tr(nPersp, wave);
    for (uint32_t t = tr.first; t < tr.myItems; t += tr.stride) {
        const uint32_t it = tr.item(t);
        if (it >= nPersp) continue;
        const uint32_t q = __builtin_amdgcn_readfirstlane(p.order ? p.order[it] : it);
        if (q == 0xFFFFFFFFu) {  // a hole of a padded order
            if (lane == 0) {
                uint32_t* hole = p.lists + size_t(it) * kListStride;
                hole[0] = 0;
                hole[1] = 0;
                hole[2] = q;
            }
            continue;
        }
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


// ---------------------------------------------------------------------------------------------------------------------
// Column-sliced replay. Today every XCD fetches whole 1 KiB rows, so the eight private 4 MiB
// L2s cache eight copies of the same hot rows. Here XCD x (workgroup b runs on XCD b % 8) reads ONLY the 128-byte slice x
// of every row - columns {64x .. 64x+63} U {512+64x ..}: exactly the bytes lanes 8x .. 8x+7 of the product kernel hold, so
// the pairwise partners stay lane-local - from a table re-laid as [slice][row][128 B]: each L2 then caches an eighth of the
// table's bytes (9.7 MB against 77 MB). A wave load is still 64 x 16 B = 8 lines, now one line of 8 DIFFERENT PERSPECTIVES
// (lane group g = lane >> 3 walks the list of perspective 8 G + g): no cross-lane reduction is needed, and every XCD sees
// every perspective, so the XCDs are balanced by construction. The lists come from a pre-pass in the interleaved form
// [chunk of 4 rows][perspective g][4] (one 128-byte line feeds four wave loads), padded with an all-zero row to the
// group's longest list.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void spx_probe_slice_table_kernel(const uint8_t* table, uint8_t* sliced, uint32_t nRows) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte piece each
    if (idx >= nRows * 64u) return;
    const uint32_t r = idx >> 6, l = idx & 63u, x = l >> 3, t = l & 7u;
    const u32x4 v = *reinterpret_cast<const u32x4*>(table + size_t(r) * 1024 + 16 * l);
    *reinterpret_cast<u32x4*>(sliced + (size_t(x) * (nRows + 1) + r) * 128 + 16 * t) = v;  // row nRows of a slice stays zero
}

// perspective order of a POSITION order (the MLP's output-bucket order): both perspectives of a position side by side
__global__ void spx_probe_persp_order_kernel(const uint32_t* posOrder, uint32_t* perspOrder, uint32_t nPositions) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nPositions) {
        perspOrder[2 * i] = 2 * posOrder[i];
        perspOrder[2 * i + 1] = 2 * posOrder[i] + 1;
    }
}

// [item][kListStride] lists -> per group of 8 consecutive items: [0] chunks of 4 rows, [8 + g] perspective id (or ~0),
// from word 32: u32x4 [chunk][g] = slice byte offsets (row * 128) of rows 4 chunk .. 4 chunk + 3 of item 8 G + g
__global__ void spx_probe_pack_groups_kernel(ProbeParams p, uint32_t zeroRowOffset) {
    const uint32_t lane = laneId();
    const uint32_t nPersp = p.nPositions * 2, nGroups = (nPersp + 7) / 8;
    const uint32_t G = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (G >= nGroups) return;
    uint32_t* out = p.groupLists + size_t(G) * kProbeGroupWords;
    uint32_t nMax = 0;
    for (uint32_t g = 0; g < 8; ++g) {
        const uint32_t it = 8 * G + g;
        const uint32_t* in = p.lists + size_t(it) * kListStride;
        const uint32_t n = it < nPersp ? in[0] : 0u;
        if (it < nPersp && in[1] != 0 && lane == 0) atomicAdd(p.wideRows, in[1]);  // (the sliced replay covers u8 rows only)
        nMax = max(nMax, n);
        if (lane == 0) {
            out[8 + g] = it < nPersp ? in[2] : 0xFFFFFFFFu;
            out[16 + g] = (n + 3) / 4;  // this perspective's own chunks (the masked variant stops there)
        }
    }
    const uint32_t nChunks = (nMax + 3) / 4;
    if (lane == 0) out[0] = nChunks;
    // (+ 2 chunks of padding: the gather requests its next two chunks of entries before it knows they exist)
    for (uint32_t idx = lane; idx < (nChunks + 2) * 8; idx += 64) {
        const uint32_t k = idx >> 3, g = idx & 7u, it = 8 * G + g;
        const uint32_t* in = p.lists + size_t(it) * kListStride;
        const uint32_t n = it < nPersp ? in[0] : 0u;
        u32x4 e;
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = 4 * k + j < n ? in[kListThr + 4 * k + j] >> 3 : zeroRowOffset;
        *reinterpret_cast<u32x4*>(out + 32 + 4 * idx) = e;
    }
}

// kMasked: a lane group whose own list has ended issues no loads (exec-masked) instead of fetching the all-zero row
template <int kWaves, bool kMasked>
__global__ __launch_bounds__(64 * kWavesPerBlock, kWaves) void spx_probe_sliced_gather_kernel(ProbeParams p) {
    const uint32_t lane = laneId(), wave = threadIdx.x >> 6;
    const uint32_t nGroups = (p.nPositions * 2 + 7) / 8;
    const uint32_t xcd = blockIdx.x & 7, blockInXcd = blockIdx.x >> 3, blocksPerXcd = gridDim.x >> 3;
    const uint32_t g = lane >> 3, t = lane & 7u;
    const uint8_t* slice = p.sliced + size_t(xcd) * p.sliceStride;
    const uint32_t laneOff = 16 * t;
    for (uint32_t G = blockInXcd * kWavesPerBlock + wave; G < nGroups; G += blocksPerXcd * kWavesPerBlock) {
        const uint32_t* in = p.groupLists + size_t(G) * kProbeGroupWords;
        const uint32_t nChunks = __builtin_amdgcn_readfirstlane(in[0]);
        const uint32_t q = in[8 + g];
        const uint32_t myChunks = kMasked ? in[16 + g] : 0xFFFFFFFFu;
        const u32x4* ent = reinterpret_cast<const u32x4*>(in + 32) + g;  // chunk k of this lane's perspective: ent[8 k]
        u32x4 x = {0, 0, 0, 0};
        u32x4 e0 = ent[0], e1 = ent[8];
        uint32_t k = 0;
        for (; k + 2 <= nChunks; k += 2) {  // 8 wave loads of 8 x 128 B in flight
            u32x4 w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = u32x4{0, 0, 0, 0};
            if (k < myChunks) {
#pragma unroll
                for (int u = 0; u < 4; ++u) w[u] = loadGatherRow(slice, e0[u], laneOff);
            }
            if (k + 1 < myChunks) {
#pragma unroll
                for (int u = 0; u < 4; ++u) w[4 + u] = loadGatherRow(slice, e1[u], laneOff);
            }
            e0 = ent[8 * (k + 2)];  // the next entries travel behind this burst
            e1 = ent[8 * (k + 3)];
#pragma unroll
            for (int u = 0; u < 8; ++u) x ^= w[u];
        }
        if (k < nChunks) {
            u32x4 w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = u32x4{0, 0, 0, 0};
            if (k < myChunks) {
#pragma unroll
                for (int u = 0; u < 4; ++u) w[u] = loadGatherRow(slice, e0[u], laneOff);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) x ^= w[u];
        }
        u32x2 o;
        o[0] = x[0] ^ x[2];
        o[1] = x[1] ^ x[3];
        if (q != 0xFFFFFFFFu) *reinterpret_cast<u32x2*>(p.sink + size_t(q) * 512 + 8 * (8 * xcd + t)) = o;
    }
}


// The same replay with the entries STAGED THROUGH LDS: the variant above asks for its entries with one replicated wave load
// per chunk (64 lanes fetch 8 distinct 16-byte pieces) - a quarter more vector-memory instructions, each of which holds the
// CU's texture path for its 16 cycles whatever it coalesces to (measured: 11.4 M instructions x 16 cycles = 88 % of the
// kernel's cycles). Here a wave fetches 16 chunks of entries - 2 KiB, every lane a different 16 bytes - with TWO loads,
// parks them in its half of a 4 KiB LDS ring and reads them back per chunk with a broadcasting ds_read_b128.
template <int kWaves>
__global__ __launch_bounds__(64 * kWavesPerBlock, kWaves) void spx_probe_sliced_lds_gather_kernel(ProbeParams p) {
    __shared__ __align__(16) uint32_t sEnt[kWavesPerBlock][2][16 * 8 * 4];  // per wave: two stages of 16 chunks x 8 perspectives x 4
    const uint32_t lane = laneId(), wave = threadIdx.x >> 6;
    const uint32_t nGroups = (p.nPositions * 2 + 7) / 8;
    const uint32_t xcd = blockIdx.x & 7, blockInXcd = blockIdx.x >> 3, blocksPerXcd = gridDim.x >> 3;
    const uint32_t g = lane >> 3, t = lane & 7u;
    const uint8_t* slice = p.sliced + size_t(xcd) * p.sliceStride;
    const uint32_t laneOff = 16 * t;
    for (uint32_t G = blockInXcd * kWavesPerBlock + wave; G < nGroups; G += blocksPerXcd * kWavesPerBlock) {
        const uint32_t* in = p.groupLists + size_t(G) * kProbeGroupWords;
        const uint32_t nChunks = __builtin_amdgcn_readfirstlane(in[0]);
        const uint32_t q = in[8 + g];
        const u32x4* ent = reinterpret_cast<const u32x4*>(in + 32);
        u32x4 x = {0, 0, 0, 0};
        // stage s holds chunks [16 s, 16 s + 16): lane l fetches pieces l and 64 + l of its 128 (the lists are padded, so a
        // stage may be read past the group's last chunk - capacity kProbeGroupWords covers 72 + 2 chunks; clamp beyond)
        const uint32_t nStages = (nChunks + 15) / 16;
        u32x4 a = ent[lane], b = ent[64 + lane];
        for (uint32_t s = 0; s < nStages; ++s) {
            uint32_t* stage = sEnt[wave][s & 1];
            *reinterpret_cast<u32x4*>(stage + 4 * lane) = a;
            *reinterpret_cast<u32x4*>(stage + 256 + 4 * lane) = b;
            if (s + 1 < nStages) {  // the next stage's entries travel behind this stage's row loads
                const uint32_t base = 128 * (s + 1);
                a = ent[min(base + lane, 74u * 8 - 1)];
                b = ent[min(base + 64 + lane, 74u * 8 - 1)];
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t cEnd = min(16u, nChunks - 16 * s);
            uint32_t k = 0;
            for (; k + 2 <= cEnd; k += 2) {
                const u32x4 e0 = *reinterpret_cast<const u32x4*>(stage + 4 * (8 * k + g));
                const u32x4 e1 = *reinterpret_cast<const u32x4*>(stage + 4 * (8 * (k + 1) + g));
                u32x4 w[8];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    w[u] = loadGatherRow(slice, e0[u], laneOff);
                    w[4 + u] = loadGatherRow(slice, e1[u], laneOff);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) x ^= w[u];
            }
            if (k < cEnd) {
                const u32x4 e0 = *reinterpret_cast<const u32x4*>(stage + 4 * (8 * k + g));
                u32x4 w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) w[u] = loadGatherRow(slice, e0[u], laneOff);
#pragma unroll
                for (int u = 0; u < 4; ++u) x ^= w[u];
            }
        }
        u32x2 o;
        o[0] = x[0] ^ x[2];
        o[1] = x[1] ^ x[3];
        if (q != 0xFFFFFFFFu) *reinterpret_cast<u32x2*>(p.sink + size_t(q) * 512 + 8 * (8 * xcd + t)) = o;
        __builtin_amdgcn_wave_barrier();
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Column-sliced replay with the CURRENT KING BUCKET'S PIECE-SQUARE SLAB IN LDS. A slice of a row is 128 bytes, so the 704
// piece-square rows of one king bucket are 88 KiB per slice - they fit the CU's 160 KiB LDS (whole 1 KiB rows never did:
// 160 rows). The perspectives are ordered by (king bucket, row count) and cut into groups of 8 that never straddle a
// bucket; a host-made plan gives each of the 32 CUs of an XCD (one 16-wave workgroup each) a contiguous, equally heavy
// range of groups as segments of one bucket: at a segment's start the workgroup copies that bucket's slab slice into LDS
// once, then every piece-square row (37 % of the row fetches) is a ds_read_b128 instead of a trip through the texture path.
// ---------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kSlabRows = 704, kSlabWaves = 16;
__global__ void spx_probe_pack_slab_groups_kernel(ProbeParams p, uint32_t zeroRowOffset) {
    const uint32_t lane = laneId();
    const uint32_t nItems = p.nItems ? p.nItems : p.nPositions * 2, nGroups = (nItems + 7) / 8;
    const uint32_t G = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (G >= nGroups) return;
    uint32_t* out = p.groupLists + size_t(G) * kProbeGroupWords;
    uint32_t nPsqMax = 0, nThrMax = 0, bucket = 0, nPsqOf[8], nOf[8];
    for (uint32_t g = 0; g < 8; ++g) {
        const uint32_t it = 8 * G + g;
        const uint32_t* in = p.lists + size_t(it) * kListStride;
        const uint32_t n = it < nItems ? in[0] : 0u;
        if (it < nItems && in[1] != 0 && lane == 0) atomicAdd(p.wideRows, in[1]);
        uint32_t nPsq = 0;  // the compact piece-square rows head the u8 list
        for (uint32_t i = lane; i < min(n, 64u); i += 64) nPsq = in[kListThr + i] >= kThreatRows * kL1 ? 1u : 0u;
        nPsq = uint32_t(popc64(__ballot(nPsq != 0)));
        if (nPsq) bucket = (in[kListThr] / kL1 - kThreatRows) / kSlabRows;
        nPsqOf[g] = nPsq;
        nOf[g] = n;
        nPsqMax = max(nPsqMax, nPsq);
        nThrMax = max(nThrMax, n - nPsq);
        if (lane == 0) out[8 + g] = it < nItems ? in[2] : 0xFFFFFFFFu;
    }
    const uint32_t nPsqChunks = (nPsqMax + 3) / 4, nThrChunks = (nThrMax + 3) / 4;
    if (lane == 0) {
        out[0] = nPsqChunks;
        out[1] = nThrChunks;
        out[2] = bucket;
    }
    for (uint32_t idx = lane; idx < (nPsqChunks + nThrChunks + 1) * 8; idx += 64) {
        const uint32_t k = idx >> 3, g = idx & 7u, it = 8 * G + g;
        const uint32_t* in = p.lists + size_t(it) * kListStride;
        uint32_t nPsq = 0, n = 0;
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
            if (j == g) {
                nPsq = nPsqOf[j];
                n = nOf[j];
            }
        }
        u32x4 e;
        if (k < nPsqChunks) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                e[j] = 4 * k + j < nPsq ? (in[kListThr + 4 * k + j] / kL1 - kThreatRows - bucket * kSlabRows) * 128u : kSlabRows * 128u;
            }
        } else {
            const uint32_t k2 = k - nPsqChunks;
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = nPsq + 4 * k2 + j < n ? in[kListThr + nPsq + 4 * k2 + j] >> 3 : zeroRowOffset;
        }
        *reinterpret_cast<u32x4*>(out + 32 + 4 * idx) = e;
    }
}

template <bool kSlab>
__global__ __launch_bounds__(64 * kSlabWaves, 1) void spx_probe_slab_gather_kernel(ProbeParams p) {
    __shared__ __align__(16) uint8_t sSlab[(kSlabRows + 1) * 128];              // the bucket's slab slice + an all-zero row
    __shared__ __align__(16) uint32_t sEnt[kSlabWaves][2][8 * 8 * 4];           // per wave: two stages of 8 chunks of entries
    const uint32_t lane = laneId(), wave = threadIdx.x >> 6;
    const uint32_t xcd = blockIdx.x & 7, cu = blockIdx.x >> 3;
    const uint32_t g = lane >> 3, t = lane & 7u;
    const uint8_t* slice = p.sliced + size_t(xcd) * p.sliceStride;
    const uint32_t laneOff = 16 * t;
    for (uint32_t i = threadIdx.x; i < 8; i += blockDim.x) reinterpret_cast<u32x4*>(sSlab + kSlabRows * 128)[i] = u32x4{0, 0, 0, 0};
    const uint32_t segFirst = p.plan[cu], segEnd = p.plan[cu + 1];
    for (uint32_t seg = segFirst; seg < segEnd; ++seg) {
        const uint32_t bucket = p.plan[64 + 3 * seg], gFirst = p.plan[64 + 3 * seg + 1], gEnd = p.plan[64 + 3 * seg + 2];
        if constexpr (kSlab) {
            __syncthreads();  // the previous segment's readers are done
            const u32x4* src = reinterpret_cast<const u32x4*>(slice + size_t(kThreatRows + bucket * kSlabRows) * 128);
            for (uint32_t i = threadIdx.x; i < kSlabRows * 8; i += blockDim.x) reinterpret_cast<u32x4*>(sSlab)[i] = src[i];
            __syncthreads();
        }
        for (uint32_t G = gFirst + wave; G < gEnd; G += kSlabWaves) {
            const uint32_t* in = p.groupLists + size_t(G) * kProbeGroupWords;
            const uint32_t nPsqChunks = __builtin_amdgcn_readfirstlane(in[0]);
            const uint32_t nChunks = nPsqChunks + __builtin_amdgcn_readfirstlane(in[1]);
            const uint32_t q = in[8 + g];
            const u32x4* ent = reinterpret_cast<const u32x4*>(in + 32);
            u32x4 x = {0, 0, 0, 0};
            const uint32_t nStages = (nChunks + 7) / 8;
            u32x4 a = ent[lane];
            for (uint32_t s = 0; s < nStages; ++s) {
                uint32_t* stage = sEnt[wave][s & 1];
                *reinterpret_cast<u32x4*>(stage + 4 * lane) = a;
                if (s + 1 < nStages) a = ent[min(64 * (s + 1) + lane, 74u * 8 - 1)];
                __builtin_amdgcn_wave_barrier();
                const uint32_t cEnd = min(8u, nChunks - 8 * s);
                for (uint32_t k = 0; k < cEnd; k += 2) {
                    const uint32_t c = 8 * s + k;
                    const bool second = k + 1 < cEnd;
                    const u32x4 e0 = *reinterpret_cast<const u32x4*>(stage + 4 * (8 * k + g));
                    const u32x4 e1 = *reinterpret_cast<const u32x4*>(stage + 4 * (8 * (k + 1) + g));  // (stale but in range when !second)
                    u32x4 w[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) w[u] = u32x4{0, 0, 0, 0};
                    if (kSlab && c < nPsqChunks) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) w[u] = *reinterpret_cast<const u32x4*>(sSlab + e0[u] + laneOff);
                    } else {
                        const uint8_t* base = (!kSlab && c < nPsqChunks) ? slice + size_t(kThreatRows + in[2] * kSlabRows) * 128 : slice;
#pragma unroll
                        for (int u = 0; u < 4; ++u) w[u] = loadGatherRow(base, e0[u], laneOff);
                    }
                    if (second) {
                        if (kSlab && c + 1 < nPsqChunks) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) w[4 + u] = *reinterpret_cast<const u32x4*>(sSlab + e1[u] + laneOff);
                        } else {
                            const uint8_t* base = (!kSlab && c + 1 < nPsqChunks) ? slice + size_t(kThreatRows + in[2] * kSlabRows) * 128 : slice;
#pragma unroll
                            for (int u = 0; u < 4; ++u) w[4 + u] = loadGatherRow(base, e1[u], laneOff);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) x ^= w[u];
                }
            }
            u32x2 o;
            o[0] = x[0] ^ x[2];
            o[1] = x[1] ^ x[3];
            if (q != 0xFFFFFFFFu) *reinterpret_cast<u32x2*>(p.sink + size_t(q) * 512 + 8 * (8 * xcd + t)) = o;
            __builtin_amdgcn_wave_barrier();
        }
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
    // column-sliced: XCD x reads the 128-byte slice x of every row; 8 perspectives per wave (path 2)
    {"column-sliced, king-bucket order, 5 waves/SIMD", 2, 0, 5, 0}, {"column-sliced, king-bucket order, 8 waves/SIMD", 2, 0, 8, 0},
    {"column-sliced, positions as they come, 5 waves/SIMD", 2, 0, 5, 1}, {"column-sliced, positions as they come, 8 waves/SIMD", 2, 0, 8, 1},
    {"column-sliced, output-bucket order, 5 waves/SIMD", 2, 0, 5, 2}, {"column-sliced, output-bucket order, 8 waves/SIMD", 2, 0, 8, 2},
    {"column-sliced, ordered by row count, 5 waves/SIMD", 2, 0, 5, 3}, {"column-sliced, ordered by row count, 8 waves/SIMD", 2, 0, 8, 3},
    {"column-sliced, exec-masked tails, positions as they come, 5 waves/SIMD", 3, 0, 5, 1},
    {"column-sliced, exec-masked tails, output-bucket order, 5 waves/SIMD", 3, 0, 5, 2},
    {"column-sliced, exec-masked tails, output-bucket order, 8 waves/SIMD", 3, 0, 8, 2},
    {"column-sliced, exec-masked tails, ordered by row count, 5 waves/SIMD", 3, 0, 5, 3},
    {"column-sliced, entries through LDS, output-bucket order, 5 waves/SIMD", 4, 0, 5, 2},
    {"column-sliced, entries through LDS, output-bucket order, 8 waves/SIMD", 4, 0, 8, 2},
    {"column-sliced, entries through LDS, ordered by row count, 5 waves/SIMD", 4, 0, 5, 3},
    {"column-sliced, entries through LDS, ordered by row count, 8 waves/SIMD", 4, 0, 8, 3},
    {"column-sliced, (king bucket, row count) order, one 16-wave workgroup per CU, rows from L2", 5, 0, 4, 4},
    {"column-sliced, (king bucket, row count) order, the bucket's piece-square slab in LDS", 6, 0, 4, 4},
};

hipError_t launchProbeSliceTable(const uint8_t* table, uint8_t* sliced, uint32_t nRows, hipStream_t stream) {
    hipLaunchKernelGGL(spx_probe_slice_table_kernel, dim3((nRows * 64u + 255) / 256), dim3(256), 0, stream, table, sliced, nRows);
    return hipGetLastError();
}
hipError_t launchProbePerspOrder(const uint32_t* posOrder, uint32_t* perspOrder, uint32_t nPositions, hipStream_t stream) {
    hipLaunchKernelGGL(spx_probe_persp_order_kernel, dim3((nPositions + 255) / 256), dim3(256), 0, stream, posOrder, perspOrder, nPositions);
    return hipGetLastError();
}
hipError_t launchProbePackSlabGroups(const ProbeParams& p, uint32_t zeroRowOffset, hipStream_t stream) {
    const uint32_t nGroups = ((p.nItems ? p.nItems : p.nPositions * 2) + 7) / 8;
    hipLaunchKernelGGL(spx_probe_pack_slab_groups_kernel, dim3((nGroups + 3) / 4), dim3(256), 0, stream, p, zeroRowOffset);
    return hipGetLastError();
}
hipError_t launchProbePackGroups(const ProbeParams& p, uint32_t zeroRowOffset, hipStream_t stream) {
    const uint32_t nGroups = (p.nPositions * 2 + 7) / 8;
    hipLaunchKernelGGL(spx_probe_pack_groups_kernel, dim3((nGroups + 3) / 4), dim3(256), 0, stream, p, zeroRowOffset);
    return hipGetLastError();
}

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
    case 8: case 10: case 12: case 14: hipLaunchKernelGGL((spx_probe_sliced_gather_kernel<5, false>), grid, block, 0, stream, p); break;
    case 9: case 11: case 13: case 15: hipLaunchKernelGGL((spx_probe_sliced_gather_kernel<8, false>), grid, block, 0, stream, p); break;
    case 16: case 17: case 19: hipLaunchKernelGGL((spx_probe_sliced_gather_kernel<5, true>), grid, block, 0, stream, p); break;
    case 18: hipLaunchKernelGGL((spx_probe_sliced_gather_kernel<8, true>), grid, block, 0, stream, p); break;
    case 20: case 22: hipLaunchKernelGGL((spx_probe_sliced_lds_gather_kernel<5>), grid, block, 0, stream, p); break;
    case 21: case 23: hipLaunchKernelGGL((spx_probe_sliced_lds_gather_kernel<8>), grid, block, 0, stream, p); break;
    case 24: hipLaunchKernelGGL((spx_probe_slab_gather_kernel<false>), dim3(256), dim3(64 * kSlabWaves), 0, stream, p); break;
    case 25: hipLaunchKernelGGL((spx_probe_slab_gather_kernel<true>), dim3(256), dim3(64 * kSlabWaves), 0, stream, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace spx
