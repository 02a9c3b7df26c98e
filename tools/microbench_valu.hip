// microbench_valu.hip -- per-opcode VALU issue cost on gfx950 with the SIMDs saturated (8 waves/SIMD).
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench_valu.hip -o tools/microbench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP8(x) x x x x x x x x
#define BODY(ASM) \
    for (int it = 0; it < iters; it++) { \
        REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(sm));) \
    }
// 8 independent instructions per asm block, 8 blocks per loop turn = 64 instructions
#define OP8(op) op " %0, %8, %0\n" op " %1, %8, %1\n" op " %2, %8, %2\n" op " %3, %8, %3\n" op " %4, %8, %4\n" op " %5, %8, %5\n" op " %6, %8, %6\n" op " %7, %8, %7\n"
#define OP8_3(op) op " %0, %8, %0, %9\n" op " %1, %8, %1, %9\n" op " %2, %8, %2, %9\n" op " %3, %8, %3, %9\n" op " %4, %8, %4, %9\n" op " %5, %8, %5, %9\n" op " %6, %8, %6, %9\n" op " %7, %8, %7, %9\n"
#define CMP8(op) op " vcc, %8, %0\n" op " vcc, %8, %1\n" op " vcc, %8, %2\n" op " vcc, %8, %3\n" op " vcc, %8, %4\n" op " vcc, %8, %5\n" op " vcc, %8, %6\n" op " vcc, %8, %7\n"
#define CNDS8 "v_cndmask_b32 %0, %8, %0, s[20:21]\nv_cndmask_b32 %1, %8, %1, s[20:21]\nv_cndmask_b32 %2, %8, %2, s[20:21]\nv_cndmask_b32 %3, %8, %3, s[20:21]\nv_cndmask_b32 %4, %8, %4, s[20:21]\nv_cndmask_b32 %5, %8, %5, s[20:21]\nv_cndmask_b32 %6, %8, %6, s[20:21]\nv_cndmask_b32 %7, %8, %7, s[20:21]\n"
#define CMPCND4 "v_cmp_lt_f32 vcc, %8, %0\nv_cndmask_b32 %1, %9, %1, vcc\nv_cmp_lt_f32 vcc, %8, %2\nv_cndmask_b32 %3, %9, %3, vcc\nv_cmp_lt_f32 vcc, %8, %4\nv_cndmask_b32 %5, %9, %5, vcc\nv_cmp_lt_f32 vcc, %8, %6\nv_cndmask_b32 %7, %9, %7, vcc\n"
#define CMPS8 "v_cmp_lt_f32 s[20:21], %8, %0\nv_cmp_lt_f32 s[22:23], %8, %1\nv_cmp_lt_f32 s[24:25], %8, %2\nv_cmp_lt_f32 s[26:27], %8, %3\nv_cmp_lt_f32 s[20:21], %8, %4\nv_cmp_lt_f32 s[22:23], %8, %5\nv_cmp_lt_f32 s[24:25], %8, %6\nv_cmp_lt_f32 s[26:27], %8, %7\n"
#define BFE8 "v_bfe_i32 %0, %0, 3, 1\nv_bfe_i32 %1, %1, 3, 1\nv_bfe_i32 %2, %2, 3, 1\nv_bfe_i32 %3, %3, 3, 1\nv_bfe_i32 %4, %4, 3, 1\nv_bfe_i32 %5, %5, 3, 1\nv_bfe_i32 %6, %6, 3, 1\nv_bfe_i32 %7, %7, 3, 1\n"
#define CND8 "v_cndmask_b32 %0, %8, %0, vcc\nv_cndmask_b32 %1, %8, %1, vcc\nv_cndmask_b32 %2, %8, %2, vcc\nv_cndmask_b32 %3, %8, %3, vcc\nv_cndmask_b32 %4, %8, %4, vcc\nv_cndmask_b32 %5, %8, %5, vcc\nv_cndmask_b32 %6, %8, %6, vcc\nv_cndmask_b32 %7, %8, %7, vcc\n"

template <int K>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    uint32_t sm = 0x7fffffffu;
    if (K == 0) BODY(OP8("v_add_f32"))
    if (K == 1) BODY(OP8_3("v_fma_f32"))
    if (K == 2) BODY(OP8("v_min_f32"))
    if (K == 3) BODY(OP8("v_sub_f32"))
    if (K == 4) BODY(OP8_3("v_med3_f32"))
    if (K == 5) BODY(OP8("v_and_b32"))
    if (K == 6) BODY(OP8("v_xor_b32"))
    if (K == 7) BODY(OP8("v_add_u32"))
    if (K == 8) BODY(OP8("v_lshlrev_b32"))
    if (K == 9) BODY(OP8_3("v_bfi_b32"))
    if (K == 10) BODY(OP8_3("v_alignbit_b32"))
    if (K == 11) BODY(CMP8("v_cmp_lt_f32"))
    if (K == 12) BODY(CND8)
    if (K == 13) BODY(OP8_3("v_and_or_b32"))
    if (K == 14) BODY(OP8("v_mul_f32"))
    if (K == 15) BODY(OP8("v_max_f32"))
    if (K == 17) BODY(OP8_3("v_add3_u32"))
    if (K == 18) BODY(OP8("v_or_b32"))
    if (K == 20) { asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20", "s21"); BODY(CNDS8) }
    if (K == 21) BODY(CMPCND4)
    if (K == 22) BODY(CMPS8)
    if (K == 23) BODY(BFE8)
    if (K == 24) BODY(OP8("v_ashrrev_i32"))
    if (K == 25) BODY(OP8("v_sub_u32"))
    if (K == 26) BODY(OP8("v_min_u32"))
    if (K == 27) BODY(OP8_3("v_perm_b32"))
    if (K == 29) BODY(OP8_3("v_xad_u32"))
    if (K == 30) BODY(OP8_3("v_lshl_or_b32"))
    if (K == 31) BODY(OP8_3("v_lshl_add_u32"))
    if (K == 32) BODY(OP8("v_mul_lo_u32"))
    if (K == 19) BODY(OP8_3("v_min3_f32"))
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int K>
int run(const char *name, int waves_per_simd) {
    const int iters = 2000;
    int grid = 256 * waves_per_simd;  // 256-thread blocks = 4 waves = 1 per SIMD per block
    float *out;
    CHK(hipMalloc(&out, (size_t)grid * 256 * 4));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<K>, dim3(grid), dim3(256), 0, 0, out, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<K>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(b);
    CHK(hipEventSynchronize(b));
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double instr_per_simd = (double)waves_per_simd * iters * 64;
    printf("%-16s %d waves/SIMD: %7.3f ms  -> %.2f clk per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / instr_per_simd);
    hipFree(out);
    return 0;
}

int main() {
    for (int w : {2, 8}) {
        run<0>("v_add_f32", w); run<1>("v_fma_f32", w); run<14>("v_mul_f32", w); run<3>("v_sub_f32", w); run<2>("v_min_f32", w); run<15>("v_max_f32", w);
        run<4>("v_med3_f32", w); run<19>("v_min3_f32", w); run<5>("v_and_b32", w); run<6>("v_xor_b32", w); run<18>("v_or_b32", w); run<7>("v_add_u32", w); run<8>("v_lshlrev_b32", w);
        run<9>("v_bfi_b32", w); run<10>("v_alignbit_b32", w); run<13>("v_and_or_b32", w); run<17>("v_add3_u32", w); run<11>("v_cmp_lt_f32", w); run<12>("v_cndmask_b32 vcc", w);
        run<20>("v_cndmask sgpr", w); run<21>("cmp+cndmask pair", w); run<22>("v_cmp -> sgpr", w); run<23>("v_bfe_i32", w); run<24>("v_ashrrev_i32", w);
        run<25>("v_sub_u32", w); run<26>("v_min_u32", w); run<27>("v_perm_b32", w); run<29>("v_xad_u32", w); run<30>("v_lshl_or_b32", w); run<31>("v_lshl_add_u32", w); run<32>("v_mul_lo_u32", w);
        printf("\n");
    }
    return 0;
}
