// Probe: semantics and issue rate of v_mqsad_pk_u16_u8 on gfx950 as a 4-byte -> 4 x u16 widening accumulate.
//   acc64 = mqsad(src0 = {dword, x}, ref = 0x000000FF, acc64): only reference byte 0 is unmasked, so window k adds
//   |src0.byte[k] - 255| = 255 - byte[k] to acc.u16[k] (k = 0..3).
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/mqsad_probe.hip -o /tmp/mqsad_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void semantics(const uint32_t* in, uint64_t* out) {
    uint64_t acc = 0x0001000200030004ull;
    acc = __builtin_amdgcn_mqsad_pk_u16_u8(uint64_t(in[threadIdx.x]) | (uint64_t(in[threadIdx.x + 64]) << 32), 0xFFu, acc);
    out[threadIdx.x] = acc;
}

template <int kMode>
__global__ __launch_bounds__(256) void rate(const uint32_t* in, uint64_t* out, int iters) {
    uint32_t x[8];
    for (int i = 0; i < 8; ++i) x[i] = in[threadIdx.x + 64 * i];
    uint64_t a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (kMode == 0) {  // mqsad: 1 op per dword
                a[i] = __builtin_amdgcn_mqsad_pk_u16_u8(uint64_t(x[i]), 0xFFu, a[i]);
            } else if (kMode == 1) {  // perm + add: 2 perms + 2 adds per dword
                uint32_t lo = uint32_t(a[i]), hi = uint32_t(a[i] >> 32);
                lo += __builtin_amdgcn_perm(0u, x[i], 0x0C010C00u);
                hi += __builtin_amdgcn_perm(0u, x[i], 0x0C030C02u);
                a[i] = uint64_t(lo) | (uint64_t(hi) << 32);
            } else {  // plain 32-bit adds: 2 per dword (full-rate yardstick)
                uint32_t lo = uint32_t(a[i]), hi = uint32_t(a[i] >> 32);
                lo += x[i];
                hi += x[i] ^ lo;
                a[i] = uint64_t(lo) | (uint64_t(hi) << 32);
            }
            x[i] += uint32_t(it);  // keep the inputs live and varying (1 extra VALU op in every mode)
        }
    }
    uint64_t r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
    std::vector<uint32_t> h(1024);
    for (size_t i = 0; i < h.size(); ++i) h[i] = uint32_t(i * 2654435761u) ^ 0x00FF01FEu;
    uint32_t* dIn;
    uint64_t* dOut;
    (void)hipMalloc(&dIn, h.size() * 4);
    (void)hipMalloc(&dOut, 256 * 4096 * 8);
    (void)hipMemcpy(dIn, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    semantics<<<1, 64>>>(dIn, dOut);
    uint64_t out[64];
    (void)hipMemcpy(out, dOut, sizeof(out), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t) {
        const uint32_t lo = h[t];
        uint64_t want = 0;
        const uint16_t base[4] = {4, 3, 2, 1};
        for (int k = 0; k < 4; ++k) want |= uint64_t(uint16_t(base[k] + 255 - ((lo >> (8 * k)) & 0xFF))) << (16 * k);
        if (want != out[t]) {
            if (bad < 4) printf("lane %d: in %08x hi %08x got %016llx want %016llx\n", t, lo, h[t + 64], (unsigned long long)out[t], (unsigned long long)want);
            ++bad;
        }
    }
    printf("semantics: %d of 64 lanes differ from '255 - byte[k] into u16[k]'\n", bad);
    const int iters = 4096, blocks = 256 * 8;
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(a);
            if (mode == 0) rate<0><<<blocks, 256>>>(dIn, dOut, iters);
            if (mode == 1) rate<1><<<blocks, 256>>>(dIn, dOut, iters);
            if (mode == 2) rate<2><<<blocks, 256>>>(dIn, dOut, iters);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
        }
        float ms;
        (void)hipEventElapsedTime(&ms, a, b);
        const double dwords = double(blocks) * 4 /*waves*/ * iters * 8;
        printf("mode %d (%s): %.3f ms, %.2f ns per wave-dword per CU-SIMD -> %.2f cycles @2.4GHz per wave-level dword\n", mode,
               mode == 0 ? "mqsad" : mode == 1 ? "perm+add" : "add", ms, ms * 1e6 / (dwords / 1024.0),
               ms * 1e6 / (dwords / 1024.0) * 2.4);
    }
    return 0;
}
