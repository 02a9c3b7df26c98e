// microbench_pk16.hip -- issue cost of the packed 16-bit instructions the two-frames-per-lane fp16 min-sum kernel is made of
// (v_pk_fma_f16 / v_pk_add_f16 / v_pk_mul_f16 / v_pk_min_f16 / v_pk_max_f16 and the integer forms v_pk_min_u16 / v_pk_max_u16 /
// v_pk_add_u16), next to v_and_b32 / v_xor_b32 / v_min_f32 / v_med3_f32 as yardsticks.  Same method as microbench_valu2.hip: SIMDs
// saturated with independent instructions, clk per wave-instruction per SIMD at 2.4 GHz.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench_pk16.hip -o tools/microbench_pk16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define B8(op) op " %0, %8, %0\n" op " %1, %8, %1\n" op " %2, %8, %2\n" op " %3, %8, %3\n" op " %4, %8, %4\n" op " %5, %8, %5\n" op " %6, %8, %6\n" op " %7, %8, %7\n"
#define T8(op) op " %0, %8, %0, %9\n" op " %1, %8, %1, %9\n" op " %2, %8, %2, %9\n" op " %3, %8, %3, %9\n" op " %4, %8, %4, %9\n" op " %5, %8, %5, %9\n" op " %6, %8, %6, %9\n" op " %7, %8, %7, %9\n"
#define R8(x) x x x x x x x x
#define OPS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)

template <int K>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters) {
    uint32_t a0 = 0x3c003c00u + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = 0x3bff3c01u, c = 0x38003400u;
    for (int it = 0; it < iters; it++) {
        if (K == 0) { R8(asm volatile(T8("v_pk_fma_f16") : OPS);) }
        if (K == 1) { R8(asm volatile(B8("v_pk_add_f16") : OPS);) }
        if (K == 2) { R8(asm volatile(B8("v_pk_mul_f16") : OPS);) }
        if (K == 3) { R8(asm volatile(B8("v_pk_min_f16") : OPS);) }
        if (K == 4) { R8(asm volatile(B8("v_pk_max_f16") : OPS);) }
        if (K == 5) { R8(asm volatile(B8("v_pk_min_u16") : OPS);) }
        if (K == 6) { R8(asm volatile(B8("v_pk_max_u16") : OPS);) }
        if (K == 7) { R8(asm volatile(B8("v_pk_add_u16") : OPS);) }
        if (K == 8) { R8(asm volatile(B8("v_and_b32") : OPS);) }
        if (K == 9) { R8(asm volatile(B8("v_xor_b32") : OPS);) }
        if (K == 10) { R8(asm volatile(B8("v_min_f32") : OPS);) }
        if (K == 11) { R8(asm volatile(T8("v_med3_f32") : OPS);) }
        if (K == 12) { R8(asm volatile(T8("v_xad_u32") : OPS);) }
        if (K == 13) { R8(asm volatile(T8("v_and_or_b32") : OPS);) }
        if (K == 14) { R8(asm volatile(B8("v_pk_sub_i16") : OPS);) }
        if (K == 15) { R8(asm volatile(B8("v_pk_min_i16") : OPS);) }
        if (K == 16) { R8(asm volatile(T8("v_pk_minimum3_f16") : OPS);) }
        if (K == 17) { R8(asm volatile(T8("v_pk_maximum3_f16") : OPS);) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <class F>
int run(const char *name, F kern, int waves_per_simd) {
    const int iters = 2000, grid = 256 * waves_per_simd;
    uint32_t *out;
    CHK(hipMalloc(&out, (size_t)grid * 256 * 4));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(b);
    CHK(hipEventSynchronize(b));
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-16s %d waves/SIMD: %7.3f ms  -> %.2f clk per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * iters * 64));
    hipFree(out);
    return 0;
}

int main() {
    for (int w : {2, 8}) {
        run("v_pk_fma_f16", k<0>, w); run("v_pk_add_f16", k<1>, w); run("v_pk_mul_f16", k<2>, w); run("v_pk_min_f16", k<3>, w); run("v_pk_max_f16", k<4>, w);
        run("v_pk_min_u16", k<5>, w); run("v_pk_max_u16", k<6>, w); run("v_pk_add_u16", k<7>, w); run("v_pk_sub_i16", k<14>, w); run("v_pk_min_i16", k<15>, w);
        run("v_pk_minimum3_f16", k<16>, w); run("v_pk_maximum3_f16", k<17>, w);
        run("v_and_b32", k<8>, w); run("v_xor_b32", k<9>, w);
        run("v_min_f32", k<10>, w); run("v_med3_f32", k<11>, w); run("v_xad_u32", k<12>, w); run("v_and_or_b32", k<13>, w);
        printf("\n");
    }
    return 0;
}
