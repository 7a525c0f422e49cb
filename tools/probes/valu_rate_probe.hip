// Issue-rate probe for the integer VALU instructions the feature-transformer kernels are made of (gfx950).
// Question it answers: how many SIMD cycles does one wave64 instruction of each kind cost when W waves share a SIMD?
// (the f32 FMA rate in MI355X_MICROARCH.md is 2 cycles per wave64 instruction; bench.py's `valu_util` needs the figure
// for v_perm_b32 / v_add3_u32 / v_pk_add_u16 / 64-bit integer sequences, which is what the kernels issue.)
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_rate_probe tools/probes/valu_rate_probe.hip && ./valu_rate_probe
//
// Every kernel runs kIters x 64 instructions on 8 independent register chains (no memory traffic), on a grid of
// CUs x 4 SIMDs x W waves; cycles/instruction/SIMD = elapsed x clock / (W x instructions per wave).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                    \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

constexpr int kIters = 4096;

#define BODY8(INSTR)                                                                                                  \
    asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)                               \
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])      \
                 : "v"(k));

#define KERNEL(NAME, INSTR)                                                             \
    __global__ __launch_bounds__(64) void NAME(uint32_t* out, uint32_t k) {             \
        uint32_t r[8];                                                                  \
        for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 2654435761u + i;               \
        for (int it = 0; it < kIters; ++it) {                                           \
            BODY8(INSTR) BODY8(INSTR) BODY8(INSTR) BODY8(INSTR)                         \
            BODY8(INSTR) BODY8(INSTR) BODY8(INSTR) BODY8(INSTR)                         \
        }                                                                               \
        uint32_t s = 0;                                                                 \
        for (int i = 0; i < 8; ++i) s ^= r[i];                                          \
        if (s == 0x12345678u) out[threadIdx.x] = s;                                     \
    }

#define I_ADD(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %8\n"
#define I_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %8\n"
#define I_PKADD(n) "v_pk_add_u16 %" #n ", %" #n ", %8\n"
#define I_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define I_LSHL(n) "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
#define I_BFREV(n) "v_bfrev_b32 %" #n ", %" #n "\n"
#define I_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define I_MAD24(n) "v_mad_i32_i24 %" #n ", %" #n ", %8, %8\n"
#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %8\n"
#define I_BCNT(n) "v_bcnt_u32_b32 %" #n ", %" #n ", %8\n"
#define I_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_MBCNT(n) "v_mbcnt_lo_u32_b32 %" #n ", %" #n ", %8\n"
#define I_CMP(n) "v_cmp_lt_u32 vcc, %" #n ", %8\n"
#define I_LSHL64(n) "v_lshlrev_b64 v[200:201], %" #n ", v[202:203]\n"
#define I_MAX(n) "v_max_i32 %" #n ", %" #n ", %8\n"
#define I_MED3(n) "v_med3_i32 %" #n ", %" #n ", %8, %8\n"
#define I_DOT4(n) "v_dot4_i32_i8 %" #n ", %" #n ", %8, %" #n "\n"

KERNEL(k_add, I_ADD)
KERNEL(k_add3, I_ADD3)
KERNEL(k_perm, I_PERM)
KERNEL(k_pkadd, I_PKADD)
KERNEL(k_and, I_AND)
KERNEL(k_lshl, I_LSHL)
KERNEL(k_bfrev, I_BFREV)
KERNEL(k_mullo, I_MULLO)
KERNEL(k_mad24, I_MAD24)
KERNEL(k_fma, I_FMA)
KERNEL(k_bcnt, I_BCNT)
KERNEL(k_cndmask, I_CNDMASK)
KERNEL(k_mbcnt, I_MBCNT)
KERNEL(k_cmp, I_CMP)
KERNEL(k_max, I_MAX)
KERNEL(k_med3, I_MED3)
KERNEL(k_dot4, I_DOT4)

// 64-bit forms (register pairs): the 64-bit multiply-add the u8 gather accumulates whole dwords with, the 64-bit pointer
// add of a per-lane global address, and v_readfirstlane (VALU slot, scalar destination)
#define KERNEL64(NAME, ASM)                                                            \
    __global__ __launch_bounds__(64) void NAME(uint32_t* out, uint32_t k) {             \
        uint64_t r[8];                                                                  \
        for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 2654435761u + i;               \
        for (int it = 0; it < kIters; ++it) {                                           \
            _Pragma("unroll") for (int rep = 0; rep < 8; ++rep) {                       \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) {                         \
                    asm volatile(ASM : "+v"(r[i]) : "v"(k) : "vcc");                    \
                }                                                                       \
            }                                                                           \
        }                                                                               \
        uint64_t s = 0;                                                                 \
        for (int i = 0; i < 8; ++i) s ^= r[i];                                          \
        if (s == 0x12345678u) out[threadIdx.x] = uint32_t(s);                           \
    }
KERNEL64(k_mad64, "v_mad_u64_u32 %0, vcc, %1, 1, %0")
KERNEL64(k_lshladd64, "v_lshl_add_u64 %0, %0, 0, s[2:3]")
__global__ __launch_bounds__(64) void k_rfl(uint32_t* out, uint32_t k) {
    uint32_t r[8];
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 2654435761u + i;
    uint32_t acc = 0;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
            asm volatile("v_readfirstlane_b32 s20, %0\nv_readfirstlane_b32 s21, %1\nv_readfirstlane_b32 s22, %2\n"
                         "v_readfirstlane_b32 s23, %3\nv_readfirstlane_b32 s24, %4\nv_readfirstlane_b32 s25, %5\n"
                         "v_readfirstlane_b32 s26, %6\nv_readfirstlane_b32 s27, %7\n"
                         :
                         : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7])
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        }
    }
    if (acc == 0x12345678u + k) out[threadIdx.x] = acc;
}

typedef void (*kern_t)(uint32_t*, uint32_t);

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clockHz = prop.clockRate * 1e3;  // kHz -> Hz (nominal peak engine clock)
    printf("device %s, %d CUs, nominal clock %.0f MHz\n", prop.name, cus, clockHz / 1e6);
    uint32_t* d;
    CHECK(hipMalloc(&d, 4096));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    struct {
        const char* name;
        kern_t fn;
    } kernels[] = {{"v_add_u32", k_add},       {"v_add3_u32", k_add3},       {"v_perm_b32", k_perm},
                   {"v_pk_add_u16", k_pkadd},  {"v_and_b32", k_and},         {"v_lshlrev_b32", k_lshl},
                   {"v_bfrev_b32", k_bfrev},   {"v_mul_lo_u32", k_mullo},    {"v_mad_i32_i24", k_mad24},
                   {"v_fma_f32", k_fma},       {"v_bcnt_u32_b32", k_bcnt},   {"v_cndmask_b32", k_cndmask},
                   {"v_mbcnt_lo", k_mbcnt},    {"v_cmp_lt_u32", k_cmp},      {"v_max_i32", k_max},
                   {"v_med3_i32", k_med3},     {"v_dot4_i32_i8", k_dot4},
                   {"v_mad_u64_u32", k_mad64}, {"v_lshl_add_u64", k_lshladd64}, {"v_readfirstlane", k_rfl}};
    printf("%-16s", "instruction");
    const int wavesList[] = {1, 2, 4, 8};
    for (int w : wavesList) printf("  W=%d cyc/instr/SIMD", w);
    printf("   (at the nominal clock; the chip may run lower under load)\n");
    for (auto& k : kernels) {
        printf("%-16s", k.name);
        for (int w : wavesList) {
            const int blocks = cus * 4 * w;  // 64-thread blocks: one wave each, spread over the SIMDs
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(64), 0, 0, d, 3u);
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(64), 0, 0, d, 3u);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double instrPerWave = double(kIters) * 64.0;
            printf("  %18.2f", best * 1e-3 * clockHz / (w * instrPerWave));
        }
        printf("\n");
    }
    return 0;
}
