// fused_layered.hip -- built-in instances of the on-chip layered min-sum kernel (fused_layered_body.h): LDPC_SCHED_LAYERED with
// LDPC_PATH_FUSED (what LDPC_PATH_AUTO picks for the shipped AR4JA matrices in f32; any other code or type: layered_qc.hip, HBM).
#include <stdio.h>

#include "fused_layered_body.h"

#ifndef LAYERED_WAVES_PER_EU
#define LAYERED_WAVES_PER_EU 4
#endif

namespace ldpc {

template <class Plan, int SZ, class T>
__global__ __launch_bounds__((SplitGeom<Plan, SZ>::THREADS), LAYERED_WAVES_PER_EU)
void fused_layered_kernel(FusedArgs A) {
    lay::kernel_body<Plan, SZ, T>(A);
}

// the same in packed fp16, two frames per lane (LDPC_F16PK): 80 packed messages, no channel-LLR registers, the transients of the
// packed leave-one-out minimum for half a row (profiles/r03_layered_rs_ab.txt: 4 waves per SIMD 73.7 Gbit/s at 3 dB, 3 waves 66.8)
#ifndef LAYERED_PK16_WAVES_PER_EU
#define LAYERED_PK16_WAVES_PER_EU 4   // (rows split between the groups: 140 registers wanted; at 128 it spills 3-11 per sweep and still gains 10 %)
#endif
template <class Plan, int SZ, class T>
__global__ __launch_bounds__((SplitGeom<Plan, SZ>::THREADS), LAYERED_PK16_WAVES_PER_EU)
void fused_layered_pk16_kernel(FusedArgs A) {
    laypk::kernel_body<Plan, SZ, T>(A);
}

bool fused_layered_has(int variant, int dtype, int sz, int static_id) {
    return variant == LDPC_MINSUM && (dtype == LDPC_F32 || dtype == LDPC_F16 || dtype == LDPC_F16PK) && ((sz == 128 && static_id == 2) || (sz == 32 && static_id == 1));
}

template <int SZ, class T>
static void launch_layered_pk16(hipStream_t st, FusedArgs &a) {
    using G = SplitGeom<PlanAR4JA45, SZ>;
    const int per_wg = 2 * G::CPW;
    hipLaunchKernelGGL((fused_layered_pk16_kernel<PlanAR4JA45, SZ, T>), dim3((a.batch + per_wg - 1) / per_wg), dim3(G::THREADS), 0, st, a);
}
int fused_layered_pk16_launch(int sz, hipStream_t st, FusedArgs &a, KernelTimer *timer, LaunchInfo *info) {
    if (info) {
        snprintf(info->name, sizeof(info->name), "ldpc::fused_layered_pk16_kernel<ldpc::PlanAR4JA45, %d, ", sz);
        info->threads = sz == 128 ? SplitGeom<PlanAR4JA45, 128>::THREADS : SplitGeom<PlanAR4JA45, 32>::THREADS;
        info->frames_per_wg = 2 * (sz == 128 ? SplitGeom<PlanAR4JA45, 128>::CPW : SplitGeom<PlanAR4JA45, 32>::CPW);
    }
    if (timer) timer->begin(st);
    if (sz == 128) launch_layered_pk16<128, TabJpl4096>(st, a);
    else launch_layered_pk16<32, TabJpl1024>(st, a);
    if (timer) timer->end(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(LDPC_EHIP, "fused_layered_pk16 launch: %s", hipGetErrorString(e));
    return LDPC_OK;
}

template <int SZ, class T>
static void launch_layered(hipStream_t st, FusedArgs &a) {
    using G = SplitGeom<PlanAR4JA45, SZ>;
    const int grid = (a.batch + G::CPW - 1) / G::CPW;
    hipLaunchKernelGGL((fused_layered_kernel<PlanAR4JA45, SZ, T>), dim3(grid), dim3(G::THREADS), 0, st, a);
}

int fused_layered_launch(int sz, hipStream_t st, FusedArgs &a, KernelTimer *timer, LaunchInfo *info) {
    if (info) {
        snprintf(info->name, sizeof(info->name), "ldpc::fused_layered_kernel<ldpc::PlanAR4JA45, %d, ", sz);
        info->threads = sz == 128 ? SplitGeom<PlanAR4JA45, 128>::THREADS : SplitGeom<PlanAR4JA45, 32>::THREADS;
        info->frames_per_wg = sz == 128 ? SplitGeom<PlanAR4JA45, 128>::CPW : SplitGeom<PlanAR4JA45, 32>::CPW;
    }
    if (timer) timer->begin(st);
    if (sz == 128) launch_layered<128, TabJpl4096>(st, a);
    else launch_layered<32, TabJpl1024>(st, a);
    if (timer) timer->end(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(LDPC_EHIP, "fused_layered launch: %s", hipGetErrorString(e));
    return LDPC_OK;
}

}  // namespace ldpc
