// Does ONE v_mfma_i32_16x16x64_i8 widen AND add up four gathered i8 rows?  (round 4; hipcc --offload-arch=gfx950 -O2)
//
// B operand = the loaded data: lane (n = lane & 15, kb = lane >> 4) holds 16 consecutive bytes of row kb (B[16 kb + i][n]).
// A operand = a constant selection matrix: lane (m = lane & 15, kb) holds 16 bytes, byte m = sign of row kb (+1 / -1), else 0
// (A[m][16 kb + i] = sign_kb * (i == m)). Then D[m][n] = sum_kb sign_kb * (byte m of lane (n, kb)): the four rows' bytes
// added up per column, widened to i32, with a sign per row - no VALU at all. D lane (n, mb = lane >> 4), register r holds
// m = 4 mb + r, i.e. the four consecutive columns 16 n + 4 mb + r of the 256 the instruction covers.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const int8_t* rows /* [4][256] */, const int* signs /* [4] */, int32_t* out /* [256] */) {
    const uint32_t lane = threadIdx.x, n = lane & 15, kb = lane >> 4;
    const i32x4 b = *reinterpret_cast<const i32x4*>(rows + kb * 256 + 16 * n);
    int8_t sel[16];
    for (int i = 0; i < 16; ++i) sel[i] = int8_t(i == int(n) ? signs[kb] : 0);  // (A's m is this lane's lane & 15 too)
    const i32x4 a = *reinterpret_cast<const i32x4*>(sel);
    i32x4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[16 * n + 4 * kb + r] = d[r];
}

int main() {
    int8_t h[4 * 256];
    int signs[4] = {1, -1, 1, -1};
    srand(7);
    for (int t = 0; t < 2; ++t) {
        for (auto& v : h) v = int8_t(rand() % 256 - 128);
        if (t == 1) signs[1] = signs[3] = 1;
        int8_t* d_rows;
        int* d_signs;
        int32_t* d_out;
        hipMalloc(&d_rows, sizeof(h));
        hipMalloc(&d_signs, sizeof(signs));
        hipMalloc(&d_out, 256 * 4);
        hipMemcpy(d_rows, h, sizeof(h), hipMemcpyHostToDevice);
        hipMemcpy(d_signs, signs, sizeof(signs), hipMemcpyHostToDevice);
        probe<<<1, 64>>>(d_rows, d_signs, d_out);
        int32_t out[256];
        hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int c = 0; c < 256; ++c) {
            int want = 0;
            for (int k = 0; k < 4; ++k) want += signs[k] * h[k * 256 + c];
            bad += out[c] != want;
        }
        printf("signs %d %d %d %d: %d of 256 columns wrong\n", signs[0], signs[1], signs[2], signs[3], bad);
    }
    return 0;
}
