// microbench_addtid.hip -- semantics and rate of ds_write_addtid_b32 / ds_read_addtid_b32 on gfx950
// (address = M0[15:0] + offset + lane*4, no address VGPR) against the VGPR-addressed forms.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(128) void k_sem(float *out) {
    __shared__ float lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 128) lds[i] = -1.f;
    __syncthreads();
    float v = 1000.f + threadIdx.x;
    // every wave writes at M0 + 512 B + lane*4  -> both waves hit the SAME 64 dwords if TID is the lane id
    asm volatile("s_mov_b32 m0, 0\n\tds_write_addtid_b32 %0 offset:512\n\ts_waitcnt lgkmcnt(0)" ::"v"(v) : "memory");
    __syncthreads();
    float r;
    asm volatile("s_mov_b32 m0, 16\n\tds_read_addtid_b32 %0 offset:512\n\ts_waitcnt lgkmcnt(0)" : "=v"(r)::"memory");
    for (int i = threadIdx.x; i < 1024; i += 128) out[i] = lds[i];
    out[1024 + threadIdx.x] = r;
}

template <int MODE>
__global__ __launch_bounds__(64) void k_rate(float *out, int iters) {
    __shared__ float lds[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) lds[i] = 0.f;
    float v = 1.0f + lane, acc = 0.f;
    uint32_t a = lane * 4;
    asm volatile("s_mov_b32 m0, 0" ::: "memory");
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
            asm volatile("ds_write_addtid_b32 %0 offset:0\n\tds_write_addtid_b32 %0 offset:256\n\tds_write_addtid_b32 %0 offset:512\n\tds_write_addtid_b32 %0 offset:768\n\t"
                         "ds_write_addtid_b32 %0 offset:1024\n\tds_write_addtid_b32 %0 offset:1280\n\tds_write_addtid_b32 %0 offset:1536\n\tds_write_addtid_b32 %0 offset:1792" ::"v"(v) : "memory");
        } else if (MODE == 1) {
            asm volatile("ds_write_b32 %1, %0 offset:0\n\tds_write_b32 %1, %0 offset:256\n\tds_write_b32 %1, %0 offset:512\n\tds_write_b32 %1, %0 offset:768\n\t"
                         "ds_write_b32 %1, %0 offset:1024\n\tds_write_b32 %1, %0 offset:1280\n\tds_write_b32 %1, %0 offset:1536\n\tds_write_b32 %1, %0 offset:1792" ::"v"(v), "v"(a) : "memory");
        } else if (MODE == 2) {
            float r0, r1, r2, r3, r4, r5, r6, r7;
            asm volatile("ds_read_addtid_b32 %0 offset:0\n\tds_read_addtid_b32 %1 offset:256\n\tds_read_addtid_b32 %2 offset:512\n\tds_read_addtid_b32 %3 offset:768\n\t"
                         "ds_read_addtid_b32 %4 offset:1024\n\tds_read_addtid_b32 %5 offset:1280\n\tds_read_addtid_b32 %6 offset:1536\n\tds_read_addtid_b32 %7 offset:1792\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7)::"memory");
            acc += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
        } else {
            float r0, r1, r2, r3, r4, r5, r6, r7;
            asm volatile("ds_read_b32 %0, %8 offset:0\n\tds_read_b32 %1, %8 offset:256\n\tds_read_b32 %2, %8 offset:512\n\tds_read_b32 %3, %8 offset:768\n\t"
                         "ds_read_b32 %4, %8 offset:1024\n\tds_read_b32 %5, %8 offset:1280\n\tds_read_b32 %6, %8 offset:1536\n\tds_read_b32 %7, %8 offset:1792\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(a) : "memory");
            acc += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * 64 + lane] = acc + lds[lane];
}

template <int MODE>
int run(const char *name) {
    const int iters = 20000, grid = 256 * 8;
    float *out;
    CHK(hipMalloc(&out, (size_t)grid * 64 * 4));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(grid), dim3(64), 0, 0, out, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(k_rate<MODE>, dim3(grid), dim3(64), 0, 0, out, iters);
    hipEventRecord(b);
    CHK(hipEventSynchronize(b));
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-22s %8.3f ms -> %.2f clk per wave-instruction per CU (8 waves/CU, 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / (8.0 * iters * 8));
    hipFree(out);
    return 0;
}

int main() {
    float *out, h[1024 + 128];
    CHK(hipMalloc(&out, sizeof(h)));
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(128), 0, 0, out);
    CHK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < 1024; i++) if (h[i] >= 1000.f) { printf("first written dword: lds[%d] = %.0f, ", i, h[i]); int j = i; while (j < 1024 && h[j] >= 1000.f) j++; printf("run length %d, last = %.0f\n", j - i, h[j - 1]); break; }
    printf("semantics: lds[127]=%.0f lds[128]=%.0f lds[129]=%.0f lds[191]=%.0f lds[192]=%.0f lds[255]=%.0f lds[256]=%.0f\n", h[127], h[128], h[129], h[191], h[192], h[255], h[256]);
    printf("read (m0=16 B): thread0 got %.0f thread1 got %.0f thread64 got %.0f thread127 got %.0f\n", h[1024], h[1025], h[1024 + 64], h[1024 + 127]);
    run<0>("ds_write_addtid_b32"); run<1>("ds_write_b32"); run<2>("ds_read_addtid_b32"); run<3>("ds_read_b32");
    return 0;
}
