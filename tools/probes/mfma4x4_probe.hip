// Probe: does v_mfma_i32_4x4x4_16B_i8 with A = per-block identity widen each lane's 4 data bytes into its 4 i32
// accumulators?  Expectation H1: D[lane][r] == (int8) byte r of data[lane].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const uint32_t* data, int32_t* out, int swap) {
    const uint32_t lane = threadIdx.x;
    const int ident = 1 << (8 * (lane & 3));
    const int x = int(data[lane]);
    i32x4 c = {1000, 2000, 3000, 4000};
    i32x4 d = swap ? __builtin_amdgcn_mfma_i32_4x4x4i8(x, ident, c, 0, 0, 0)
                   : __builtin_amdgcn_mfma_i32_4x4x4i8(ident, x, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = d[r];
}
int main() {
    uint32_t h[64]; int32_t o[256];
    for (int l = 0; l < 64; ++l) h[l] = uint32_t((l * 4 + 1) & 0xFF) | (uint32_t((200 + l) & 0xFF) << 8) | (uint32_t((l * 7 + 3) & 0xFF) << 16) | (uint32_t((250 - l) & 0xFF) << 24);
    uint32_t* d; int32_t* od;
    hipMalloc(&d, sizeof(h)); hipMalloc(&od, sizeof(o));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int swap = 0; swap < 2; ++swap) {
        probe<<<1, 64>>>(d, od, swap);
        hipMemcpy(o, od, sizeof(o), hipMemcpyDeviceToHost);
        int okH1 = 1, okT = 1;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            const int want = int(int8_t(h[l] >> (8 * r))) + 1000 * (r + 1);
            if (o[l * 4 + r] != want) okH1 = 0;
            const int wantT = int(int8_t(h[(l & ~3) + r] >> (8 * (l & 3)))) + 1000 * (r + 1);
            if (o[l * 4 + r] != wantT) okT = 0;
        }
        printf("swap=%d  H1(widen own bytes)=%d  transposed=%d  lane5: %d %d %d %d\n", swap, okH1, okT, o[20], o[21], o[22], o[23]);
    }
    return 0;
}
