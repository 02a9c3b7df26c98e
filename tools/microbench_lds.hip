// microbench_lds.hip -- measures what the fused decoder leans on: LDS float-add forms and
// single-wave workgroup residency.  Build: hipcc -O3 --offload-arch=gfx950 tools/microbench_lds.hip -o tools/microbench_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE, int LDSB>
__global__ __launch_bounds__(64) void k_add(float *out, int iters) {
    __shared__ float lds[LDSB / 4];
    const int lane = threadIdx.x;
    for (int i = lane; i < LDSB / 4; i += 64) lds[i] = 0.f;
    float v = 1.0f + lane;
    uint32_t a = lane;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            uint32_t idx = (a + k * 128 + it) & (LDSB / 4 - 1);
            if (MODE == 0) __hip_atomic_fetch_add(&lds[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            else if (MODE == 1) lds[idx] = lds[idx] + v;
            else if (MODE == 2) lds[idx] = v;
            else v += lds[idx];
        }
    }
    float s = v;
    for (int i = lane; i < LDSB / 4; i += 64) s += lds[i];
    out[blockIdx.x * 64 + lane] = s;
}

__global__ void k_census(int *cu_count, unsigned long long *t) {
    // how many single-wave workgroups are resident at once: every WG records start/end clock
    unsigned long long t0 = wall_clock64();
    __builtin_amdgcn_s_sleep(127);
    for (int i = 0; i < 200; i++) __builtin_amdgcn_s_sleep(127);
    unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = t1; }
}

template <int MODE, int LDSB>
int run(const char *name, int grid, int iters) {
    float *out;
    CHK(hipMalloc(&out, (size_t)grid * 64 * 4));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_add<MODE, LDSB>), dim3(grid), dim3(64), 0, 0, out, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL((k_add<MODE, LDSB>), dim3(grid), dim3(64), 0, 0, out, iters);
    hipEventRecord(b);
    CHK(hipEventSynchronize(b));
    float ms;
    hipEventElapsedTime(&ms, a, b);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k_add<MODE, LDSB>, 64, 0);
    double ops = (double)grid * iters * 16;  // wave-instructions
    printf("%-28s LDS %6d B/WG occ(API) %2d WG/CU  grid %6d: %8.3f ms  -> %.2f wave-ops/clk/CU (2.4 GHz, 256 CU)\n", name, LDSB, occ, grid, ms,
           ops / (ms * 1e-3) / 2.4e9 / 256);
    hipFree(out);
    return 0;
}

int main() {
    const int iters = 4000;
    for (int rep = 0; rep < 2; rep++) {
        run<0, 22528 - 22528 % 4096 + 4096 * 0 + 0>("ds_add_f32 (atomic)", 256 * 7, iters);
    }
    run<0, 16384>("ds_add_f32 (atomic)", 256 * 8, iters);
    run<1, 16384>("read-add-write", 256 * 8, iters);
    run<2, 16384>("ds_write_b32", 256 * 8, iters);
    run<3, 16384>("ds_read_b32", 256 * 8, iters);
    run<0, 16384>("ds_add_f32 1 WG/CU", 256 * 1, iters);
    run<3, 16384>("ds_read_b32 1 WG/CU", 256 * 1, iters);
    run<0, 16384>("ds_add_f32 2 WG/CU", 256 * 2, iters);
    run<0, 16384>("ds_add_f32 4 WG/CU", 256 * 4, iters);
    // residency census of single-wave workgroups with 22.5 KB LDS each is what the fused kernel relies on
    return 0;
}
