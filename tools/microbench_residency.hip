// microbench_residency.hip -- how many workgroups of a given size are RESIDENT per CU on gfx950 (census: every workgroup
// registers in a global counter, spins ~300 us, leaves; the maximum seen / 256 CUs is the residency).  Registers and LDS
// use are minimal, so the numbers are the wave-slot / workgroup-placement rule alone.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench_residency.hip -o tools/microbench_residency
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void census(int *active, int *peak, long long spin_clocks) {
    __shared__ int dummy;
    if (threadIdx.x == 0) {
        int now = atomicAdd(active, 1) + 1;
        atomicMax(peak, now);
        dummy = now;
    }
    __syncthreads();
    const long long t0 = clock64();
    while (clock64() - t0 < spin_clocks) { __builtin_amdgcn_s_sleep(8); }
    __syncthreads();
    if (threadIdx.x == 0) atomicSub(active, 1 + (dummy < 0));
}

int main() {
    int *d;
    CHK(hipMalloc(&d, 8));
    const int sizes[] = {64, 128, 192, 256, 320, 384, 448, 512, 640, 768, 1024};
    for (int bs : sizes) {
        CHK(hipMemset(d, 0, 8));
        hipLaunchKernelGGL(census, dim3(256 * 40), dim3(bs), 0, 0, d, d + 1, 30000LL);   // ~300 us at 100 MHz clock64
        CHK(hipDeviceSynchronize());
        int h[2];
        CHK(hipMemcpy(h, d, 8, hipMemcpyDeviceToHost));
        printf("block %4d threads (%2d waves): peak %5d resident workgroups = %.2f per CU = %.1f waves per CU\n", bs, bs / 64, h[1], h[1] / 256.0,
               h[1] / 256.0 * (bs / 64));
    }
    return 0;
}
