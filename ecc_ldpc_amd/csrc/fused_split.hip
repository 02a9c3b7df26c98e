// fused_split.hip -- per-edge-message fused kernel with a frame's BLOCK ROWS split between two wave pairs.
//
// fused_msg.hip keeps all 156 messages of a thread's circulant row in VGPRs: 256 VGPRs, 2 waves per SIMD,
// and with only two waves per SIMD ~47 % of the VALU issue slots stay empty while waves sit in LDS round
// trips and barriers (profiles/r01_fused_v2_f32_minsum_pmc.json).  Here a frame (sz = 128) is owned by
// FOUR waves: pair 0 (threads 0..127) holds the even block rows, pair 1 (threads 128..255) the odd ones,
// thread (pair, r) = row r of every circulant of its pair's block rows.  78 messages + <= 24 channel-LLR
// registers per thread -> 128 VGPRs -> 4 waves per SIMD, 16 waves per CU (4 frames, as before).
// For sz < 64 a workgroup holds 64/sz frames interleaved lane by lane (sz = 32: wave 0 = pair 0 and wave 1 =
// pair 1 of two frames); loop control then runs on a workgroup-uniform "done" mask, see split_body.
// No cross-lane combine is needed (a check row is still handled by one thread) and the graph stays a
// compile-time table; the two pairs run different straight-line code behind one wave-uniform branch, so
// the total code size is unchanged.
//
// Phase B keeps the column "rounds" of fused_msg.hip (round q = q-th contribution of every block column
// in descending row order = Orig.hs:96 per column): in a round each pair adds the edges it owns; all
// targets of a round are distinct columns; one s_barrier (4 waves) per round.  Even/odd ownership makes
// consecutive contributions of a column alternate between the pairs, which balances the rounds.
#include <stdio.h>

#include "fused_split_body.h"

#ifndef SPLIT_WAVES_PER_EU
#define SPLIT_WAVES_PER_EU 4
#endif
// the tanh rule needs ~3 transient registers per edge of a row (e, suffix A, suffix S).  Measured on jpl.4096,
// 16 384 frames, hyperbolic-recurrence rule: 2 waves per SIMD (209 VGPRs, no spills) 4.92 Gbit/s, 3 waves
// (168 VGPRs, 37 spilled) 5.80; 4 waves (128 VGPRs) spills several hundred registers.
#ifndef SPLIT_TANH_WAVES_PER_EU
#define SPLIT_TANH_WAVES_PER_EU 3
#endif

namespace ldpc {

// ahead-of-time instances: the shipped matrices (any other quasi-cyclic H gets its instance from hiprtc, jit.cc)
template <typename CT, int VARIANT, class Plan, int SZ, class T>
__global__ __launch_bounds__((SplitGeom<Plan, SZ>::THREADS), (VARIANT == LDPC_V_TANH ? SPLIT_TANH_WAVES_PER_EU : SPLIT_WAVES_PER_EU))
void fused_split_kernel(FusedArgs A) {
    split_kernel_body<CT, VARIANT, Plan, SZ, T>(A);
}

bool fused_split_has(int variant, int dtype, int sz, int static_id) {
    if (dtype != LDPC_F32 || !(variant == LDPC_MINSUM || variant == LDPC_TANH)) return false;
    return (sz == 128 && static_id == 2) || (sz == 32 && static_id == 1);
}

template <int VARIANT, int SZ, class T>
static void launch_split(hipStream_t st, FusedArgs &a) {
    using G = SplitGeom<PlanAR4JA45, SZ>;
    const int grid = (a.batch + G::CPW - 1) / G::CPW;
    hipLaunchKernelGGL((fused_split_kernel<float, VARIANT, PlanAR4JA45, SZ, T>), dim3(grid), dim3(G::THREADS), 0, st, a);
}

int fused_split_launch(int variant, int sz, hipStream_t st, FusedArgs &a, KernelTimer *timer, LaunchInfo *info) {
    if (info && !a.step_mode) {
        snprintf(info->name, sizeof(info->name), "ldpc::fused_split_kernel<float, %d, ldpc::PlanAR4JA45, %d, ", variant == LDPC_MINSUM ? LDPC_V_MINSUM : LDPC_V_TANH, sz);
        info->threads = sz == 128 ? SplitGeom<PlanAR4JA45, 128>::THREADS : SplitGeom<PlanAR4JA45, 32>::THREADS;
        info->frames_per_wg = sz == 128 ? SplitGeom<PlanAR4JA45, 128>::CPW : SplitGeom<PlanAR4JA45, 32>::CPW;
    }
    if (timer && !a.step_mode) timer->begin(st);
#ifdef SPLIT_ABLATION_MINSUM128   // timing builds (tools/build_ablation.sh): the headline instance only
    launch_split<LDPC_V_MINSUM, 128, TabJpl4096>(st, a);
#else
    if (sz == 128) {
        if (variant == LDPC_MINSUM) launch_split<LDPC_V_MINSUM, 128, TabJpl4096>(st, a);
        else launch_split<LDPC_V_TANH, 128, TabJpl4096>(st, a);
    } else {
        if (variant == LDPC_MINSUM) launch_split<LDPC_V_MINSUM, 32, TabJpl1024>(st, a);
        else launch_split<LDPC_V_TANH, 32, TabJpl1024>(st, a);
    }
#endif
    if (timer && !a.step_mode) timer->end(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(LDPC_EHIP, "fused_split launch: %s", hipGetErrorString(e));
    return LDPC_OK;
}

}  // namespace ldpc
