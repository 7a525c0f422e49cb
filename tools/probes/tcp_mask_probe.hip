// What does a partially exec-masked 16-byte-per-lane wave load cost on the CU's texture / L1 path (gfx950)?
// Round 5 question behind the gather's "mixed steps": if a wave load whose lane groups are half masked holds the path
// for half the cycles, a step may fetch every row from where it lives (hot rows: ds_read_b128, cold rows: global load,
// each under its own exec mask) and the lists need no separate hot / cold sections. If the instruction costs its 16
// cycles whatever the mask, sections it is.
//
//   hipcc --offload-arch=gfx950 -O2 -o tcp_mask_probe tools/probes/tcp_mask_probe.hip && ./tcp_mask_probe
//
// Every variant: 256 workgroups x 16 waves (one per CU), each wave issues kIters bursts of 8 loads of 8 different 128-byte
// rows (lane group lane >> 3 = row of the burst entry, lane & 7 = 16-byte chunk) from a table of kRows rows; the rows are
// pseudo-random per (wave, iteration) so nothing coalesces across lane groups. Reported: ns per wave-load instruction and
// CU (elapsed / loads per CU) and the same in cycles of the 2.4 GHz clock.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                    \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kWaves = 16;
constexpr int kIters = 512;

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// kMode: 0 global dwordx4, active lane groups = kActive of 8 (groups 0 .. kActive - 1)
//        1 global dwordx4, alternate QUADS of lanes active (lane >> 2 & 1)
//        2 global dwordx2 (8 bytes per lane, all lanes)
//        3 ds_read_b128 from a 64 KiB LDS table, kActive groups
//        4 mixed: groups < kActive from global, the others from LDS (two instructions per burst entry)
//        5 global dwordx4, all 64 lanes of a burst entry read the SAME 128-byte row (8-fold replicated)
//        7 flat_load_dwordx4: groups < kActive global addresses, the others LDS addresses, one instruction
//        6 global dwordx4, every lane group active but groups >= kActive read a fixed all-zero row (the padding form)
template <int kMode, int kActive>
__global__ __launch_bounds__(64 * kWaves, 1) void probe(const uint8_t* table, uint32_t rowMask, uint32_t* sink) {
    extern __shared__ __align__(16) uint8_t sLds[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, g = lane >> 3, t = lane & 7u;
    if (kMode == 3 || kMode == 4 || kMode == 7) {
        for (uint32_t i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<u32x4*>(sLds)[i] = u32x4{i, i, i, i};
        __syncthreads();
    }
    u32x4 x = {0, 0, 0, 0};
    uint32_t seed = (blockIdx.x * kWaves + wave) * 0x9E3779B9u;
    const bool active = kMode == 1 ? ((lane >> 2) & 1u) : (g < uint32_t(kActive));
    uint32_t walk = mix(seed + (kMode == 5 ? 0u : g * 0xC2B2AE35u));
    const uint32_t inc = (mix(seed ^ (kMode == 5 ? 0u : g * 0x85EBCA6Bu)) | 1u) & 0xFFFFu;
    for (int it = 0; it < kIters; ++it) {
        u32x4 w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            w[u] = u32x4{0, 0, 0, 0};
            // (address generation must stay cheap or the probe measures the VALU: one add + one and per load; odd per-group
            // increments sweep the whole table in a scattered order)
            walk += inc;
            const uint32_t row = walk & rowMask;
            if (kMode == 0 || kMode == 1) {
                if (active) w[u] = *reinterpret_cast<const u32x4*>(table + size_t(row) * 128 + 16 * t);
            } else if (kMode == 2) {
                const u32x2 v = *reinterpret_cast<const u32x2*>(table + size_t(row) * 128 + 8 * (lane & 15u));
                w[u][0] = v[0];
                w[u][1] = v[1];
            } else if (kMode == 3) {
                if (active) w[u] = *reinterpret_cast<const u32x4*>(sLds + (row & 511u) * 128 + 16 * t);
            } else if (kMode == 4) {
                if (active) w[u] = *reinterpret_cast<const u32x4*>(table + size_t(row) * 128 + 16 * t);
                if (!active) w[u] = *reinterpret_cast<const u32x4*>(sLds + (row & 511u) * 128 + 16 * t);
            } else if (kMode == 7) {  // one flat_load per burst entry: the hardware routes every lane to its aperture
                const uint8_t* src = active ? table + size_t(row) * 128 + 16 * t : reinterpret_cast<const uint8_t*>(sLds + (row & 511u) * 128 + 16 * t);
                w[u] = *reinterpret_cast<const u32x4*>(src);
            } else if (kMode == 5) {
                w[u] = *reinterpret_cast<const u32x4*>(table + size_t(row) * 128 + 16 * t);
            } else {
                const uint32_t r2 = active ? row : rowMask + 1;  // (the row behind the table: all zero)
                w[u] = *reinterpret_cast<const u32x4*>(table + size_t(r2) * 128 + 16 * t);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) x ^= w[u];
    }
    if ((x[0] ^ x[1] ^ x[2] ^ x[3]) == 0x12345678u) sink[threadIdx.x] = x[0];
}

template <int kMode, int kActive>
static void run(const char* name, const uint8_t* table, uint32_t rows, uint32_t* sink) {
    const size_t lds = (kMode == 3 || kMode == 4 || kMode == 7) ? 65536 : 0;
    if (lds) CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<kMode, kActive>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((probe<kMode, kActive>), dim3(256), dim3(64 * kWaves), lds, 0, table, rows - 1, sink);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    const double loadsPerCu = double(kWaves) * kIters * 8;
    const double ns = best * 1e6 / loadsPerCu;
    printf("%-86s rows %6u  %8.3f ms  %6.2f ns / wave load / CU = %5.1f cycles @ 2.4 GHz\n", name, rows, best, ns, ns * 2.4);
}

int main() {
    const uint32_t sizes[] = {128, 8192, 65536};  // 16 KiB (L1), 1 MiB (L2), 8 MiB (two L2s' worth: some fabric)
    for (uint32_t rows : sizes) {
        uint8_t* table;
        uint32_t* sink;
        CHECK(hipMalloc(reinterpret_cast<void**>(&table), size_t(rows + 1) * 128));
        CHECK(hipMalloc(reinterpret_cast<void**>(&sink), 4096 * 4));
        std::vector<uint8_t> host(size_t(rows + 1) * 128);
        for (size_t i = 0; i < host.size(); ++i) host[i] = uint8_t(i * 2654435761u >> 13);
        for (size_t i = size_t(rows) * 128; i < host.size(); ++i) host[i] = 0;
        CHECK(hipMemcpy(table, host.data(), host.size(), hipMemcpyHostToDevice));
        printf("---- table of %u rows x 128 B ----\n", rows);
        run<0, 8>("global dwordx4, 8 of 8 lane groups active", table, rows, sink);
        run<0, 6>("global dwordx4, 6 of 8 lane groups active", table, rows, sink);
        run<0, 4>("global dwordx4, 4 of 8 lane groups active", table, rows, sink);
        run<0, 2>("global dwordx4, 2 of 8 lane groups active", table, rows, sink);
        run<0, 1>("global dwordx4, 1 of 8 lane groups active", table, rows, sink);
        run<1, 8>("global dwordx4, alternate quads of lanes active", table, rows, sink);
        run<2, 8>("global dwordx2 (8 B per lane), all lanes, 4 rows per load", table, rows, sink);
        run<5, 8>("global dwordx4, all 8 lane groups read the same row", table, rows, sink);
        run<6, 4>("global dwordx4, 4 groups real rows + 4 groups the all-zero row (padding form)", table, rows, sink);
        run<3, 8>("ds_read_b128, 8 of 8 lane groups", table, rows, sink);
        run<3, 4>("ds_read_b128, 4 of 8 lane groups", table, rows, sink);
        run<4, 4>("mixed: 4 groups global + 4 groups LDS", table, rows, sink);
        run<4, 6>("mixed: 6 groups global + 2 groups LDS", table, rows, sink);
        run<4, 2>("mixed: 2 groups global + 6 groups LDS", table, rows, sink);
        run<7, 4>("flat_load_dwordx4: 4 groups global + 4 groups LDS in ONE instruction", table, rows, sink);
        run<7, 8>("flat_load_dwordx4: 8 groups global", table, rows, sink);
        run<7, 0>("flat_load_dwordx4: 8 groups LDS", table, rows, sink);
        CHECK(hipFree(table));
        CHECK(hipFree(sink));
    }
    return 0;
}
