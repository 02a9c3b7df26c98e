// microbench_valu2.hip -- issue cost of the instruction classes tools/microbench_valu.hip left out: transcendentals
// (v_exp_f32 / v_log_f32 / v_rcp_f32), packed f32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two results per lane), v_mov_b32,
// v_bitop3_b32 -- what the tanh-rule kernels are made of (tools/isa_histogram.py prices them).  Same method: SIMDs saturated
// with independent instructions, clk per wave-instruction per SIMD at 2.4 GHz.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench_valu2.hip -o tools/microbench_valu2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

#define U8(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"
#define B8(op) op " %0, %8, %0\n" op " %1, %8, %1\n" op " %2, %8, %2\n" op " %3, %8, %3\n" op " %4, %8, %4\n" op " %5, %8, %5\n" op " %6, %8, %6\n" op " %7, %8, %7\n"
#define T8(op) op " %0, %8, %0, %9\n" op " %1, %8, %1, %9\n" op " %2, %8, %2, %9\n" op " %3, %8, %3, %9\n" op " %4, %8, %4, %9\n" op " %5, %8, %5, %9\n" op " %6, %8, %6, %9\n" op " %7, %8, %7, %9\n"
#define BITOP8 "v_bitop3_b32 %0, %8, %0, %9 bitop3:0x78\nv_bitop3_b32 %1, %8, %1, %9 bitop3:0x78\nv_bitop3_b32 %2, %8, %2, %9 bitop3:0x78\nv_bitop3_b32 %3, %8, %3, %9 bitop3:0x78\nv_bitop3_b32 %4, %8, %4, %9 bitop3:0x78\nv_bitop3_b32 %5, %8, %5, %9 bitop3:0x78\nv_bitop3_b32 %6, %8, %6, %9 bitop3:0x78\nv_bitop3_b32 %7, %8, %7, %9 bitop3:0x78\n"
#define R8(x) x x x x x x x x

template <int K>
__global__ __launch_bounds__(256) void ks(float *out, int iters) {   // scalar-per-lane operands
    float a0 = 1.0f + threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    for (int it = 0; it < iters; it++) {
        if (K == 0) { R8(asm volatile(U8("v_exp_f32") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (K == 1) { R8(asm volatile(U8("v_log_f32") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (K == 2) { R8(asm volatile(U8("v_rcp_f32") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (K == 3) { R8(asm volatile(U8("v_mov_b32") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (K == 4) { R8(asm volatile(BITOP8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (K == 5) { R8(asm volatile(U8("v_sqrt_f32") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (K == 6) { R8(asm volatile(U8("v_cvt_f16_f32") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        // do transcendentals overlap with plain VALU work?  7: inside one wave, 8 v_exp then 24 v_mul (x8 per iteration);
        // 8: between waves of one SIMD: even waves run v_exp only, odd waves v_mul only, 26 v_mul for 8 v_exp (equal time alone)
        if (K == 7) { R8(asm volatile(U8("v_exp_f32") B8("v_mul_f32") B8("v_mul_f32") B8("v_mul_f32") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (K == 8 || K == 9 || K == 10) {
            const bool trans = ((threadIdx.x >> 6) & 1) == 0;
            if (trans) { if (K != 10) { R8(asm volatile(U8("v_exp_f32") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) } }
            else if (K != 9) { R8(asm volatile(B8("v_mul_f32") B8("v_mul_f32") B8("v_mul_f32") "v_mul_f32 %0, %8, %0\nv_mul_f32 %1, %8, %1\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int K>
__global__ __launch_bounds__(256) void kp(float *out, int iters) {   // packed: 64-bit register pairs
    f2 a0 = {1.0f + threadIdx.x * 1e-3f, 2.f}, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    for (int it = 0; it < iters; it++) {
        if (K == 0) { R8(asm volatile(T8("v_pk_fma_f32") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (K == 1) { R8(asm volatile(B8("v_pk_mul_f32") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (K == 2) { R8(asm volatile(B8("v_pk_add_f32") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    }
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

template <class F>
int run(const char *name, F kern, int waves_per_simd) {
    const int iters = 2000, grid = 256 * waves_per_simd;
    float *out;
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
        run("v_exp_f32", ks<0>, w); run("v_log_f32", ks<1>, w); run("v_rcp_f32", ks<2>, w); run("v_sqrt_f32", ks<5>, w); run("v_mov_b32", ks<3>, w);
        run("v_bitop3_b32", ks<4>, w); run("v_cvt_f16_f32", ks<6>, w);
        run("v_pk_fma_f32", kp<0>, w); run("v_pk_mul_f32", kp<1>, w); run("v_pk_add_f32", kp<2>, w);
        printf("\n");
    }
    // overlap of the transcendental unit with plain VALU (clk figures below are per R8 block of the named mix, 8 waves/SIMD)
    run("8exp+24mul/wave", ks<7>, 8);
    run("exp|mul waves", ks<8>, 8); run("exp waves only", ks<9>, 8); run("mul waves only", ks<10>, 8);
    return 0;
}
