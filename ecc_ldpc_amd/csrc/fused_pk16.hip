// fused_pk16.hip -- built-in instances of the packed-fp16 min-sum kernel (fused_pk16_body.h: two frames per lane), LDPC_F16PK.
// Same workgroup shape as fused_split.hip (four waves, block rows dealt to two wave groups, lam in LDS, messages in
// registers); a workgroup decodes 2 (sz = 128) or 4 (sz = 32) frames.
#include <stdio.h>

#include "fused_pk16_body.h"

// 78 packed messages + 24 packed channel LLRs per thread as in the f32 split kernel, but the leave-one-out minimum of a
// weight-18 row wants ~36 transient registers (magnitudes, suffix minima) where the f32 (min1, min2) form wants 18: at 4 waves
// per SIMD (128 VGPRs) the compiler spills 150-210 registers, at 3 (168 VGPRs; 160-162 used) none.
// r04, the verdict's route to 4 waves tried at compile time: the 24 round-0 copies of the channel LLRs moved from registers into a
// lane-private area of LDS (read back chunk by chunk in round 0 like the columns of the other rounds).  The kernel then wants 137
// VGPRs -- still 9 above the 128 of four waves ("failed to meet occupancy target") -- and 50 KB of LDS per workgroup (22.5 KB of
// lam + 27 KB of LLR copies), of which only three fit a CU: the LDS left over by four workgroups (17 KB each) holds 17 of the 24
// copies, i.e. 143 registers.  Four waves need the leave-one-out minimum of the weight-18 rows to get by with ~25 fewer transient
// registers as well, which is the diet r03 measured at +13 % instructions.  Not pursued.
#ifndef PK16_WAVES_PER_EU
#define PK16_WAVES_PER_EU 3
#endif

// wave groups a frame pair's block rows are dealt to (2 = as the f32 split kernel).  Measured on jpl.4096, 65 536 frames, 2 dB
// (profiles/r03_pk16_np_ab.txt), ms per launch: 2 groups at 3 waves per SIMD 11.88 (the default); 4 groups (512 threads, 39
// messages per thread, 115 VGPRs, 4 waves) 13.93; 4 groups at 5 waves (spills) 16.85; 3 groups 19.22; 2 groups at 4 waves (spills) 21.99.
#ifndef PK16_NP
#define PK16_NP 2
#endif

namespace ldpc {

struct PlanPk16 {   // the AR4JA rate-4/5 plan of fused_common.h with its own number of wave groups
    static constexpr int NBR = PlanAR4JA45::NBR, NBC = PlanAR4JA45::NBC, NEDGE = PlanAR4JA45::NEDGE, DMAX = PlanAR4JA45::DMAX;
    static constexpr int NP = PK16_NP;
    static constexpr int owner_br(int br) { return br % NP; }
    static constexpr int deg(int br) { return PlanAR4JA45::deg(br); }
    static constexpr int ebeg(int br) { return PlanAR4JA45::ebeg(br); }
};

template <class Plan, int SZ, class T>
__global__ __launch_bounds__((SplitGeom<Plan, SZ>::THREADS), PK16_WAVES_PER_EU)
void fused_pk16_kernel(FusedArgs A) {
    pk::kernel_body<Plan, SZ, T>(A);
}

bool fused_pk16_has(int variant, int sz, int static_id) {
    return variant == LDPC_MINSUM && ((sz == 128 && static_id == 2) || (sz == 32 && static_id == 1));
}

template <int SZ, class T>
static void launch_pk16(hipStream_t st, FusedArgs &a) {
    using G = SplitGeom<PlanPk16, SZ>;
    const int per_wg = 2 * G::CPW;
    const int grid = (a.batch + per_wg - 1) / per_wg;
    hipLaunchKernelGGL((fused_pk16_kernel<PlanPk16, SZ, T>), dim3(grid), dim3(G::THREADS), 0, st, a);
}

int fused_pk16_launch(int sz, hipStream_t st, FusedArgs &a, KernelTimer *timer, LaunchInfo *info) {
    if (info) {
        snprintf(info->name, sizeof(info->name), "ldpc::fused_pk16_kernel<ldpc::PlanPk16, %d, ", sz);
        info->threads = sz == 128 ? SplitGeom<PlanPk16, 128>::THREADS : SplitGeom<PlanPk16, 32>::THREADS;
        info->frames_per_wg = 2 * (sz == 128 ? SplitGeom<PlanPk16, 128>::CPW : SplitGeom<PlanPk16, 32>::CPW);
    }
    if (timer) timer->begin(st);
    if (sz == 128) launch_pk16<128, TabJpl4096>(st, a);
    else launch_pk16<32, TabJpl1024>(st, a);
    if (timer) timer->end(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(LDPC_EHIP, "fused_pk16 launch: %s", hipGetErrorString(e));
    return LDPC_OK;
}

}  // namespace ldpc
