// fused_common.h -- pieces shared by the fused on-chip kernels (fused.hip: compressed-record min-sum;
// fused_msg.hip: per-edge messages in VGPRs, min-sum and tanh).
#pragma once
#include "ldpc_math.h"   // first: it decides whether this is a device-only (run-time) compilation
#ifndef LDPC_DEVICE_ONLY
#include <string.h>
#include "fused.h"
#endif
#include <type_traits>
#include <utility>

namespace ldpc {

// ------------------------------------------------------------------ plans
// AR4JA rate-4/5 protograph as shipped in codes/jpl.1024.4.5 and codes/jpl.4096.4.5:
// 12 x 44 blocks, block rows 0-3 of weight 3, 4-11 of weight 18.
// number of wave groups a frame's block rows are dealt to in the split kernel (block row br -> group br % SPLIT_NP);
// 2 = the wave PAIRS of fused_split.hip.  Four groups (39 messages per thread, 5-6 waves/SIMD) measured slower on
// jpl.4096 (13.3 vs 14.5 Gbit/s), about equal on jpl.1024.
#ifndef SPLIT_NP
#define SPLIT_NP 2
#endif
struct PlanAR4JA45 {
    static constexpr int NBR = 12, NBC = 44, NEDGE = 4 * 3 + 8 * 18, DMAX = 18;
    static constexpr int NP = SPLIT_NP;
    static constexpr int owner_br(int br) { return br % NP; }
    static constexpr int deg(int br) { return br < 4 ? 3 : 18; }
    static constexpr int ebeg(int br) { return br < 4 ? 3 * br : 12 + 18 * (br - 4); }
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int I, int N, class F>
__device__ __forceinline__ void static_rfor(F &&f) {  // N-1 down to I
    if constexpr (I < N) {
        f(std::integral_constant<int, N - 1>{});
        static_rfor<I, N - 1>(f);
    }
}

// the graph table is read through the CONSTANT address space: the kernel never writes it, and only
// then may the compiler use scalar loads (s_load) although the kernel also stores to global memory.
typedef const __attribute__((address_space(4))) uint32_t *ctab_t;
struct FusedArgs {
    const uint32_t *tab;
    const void *llr;  // [batch][N], element type llr_fmt (LLR_F32 / LLR_F64 / LLR_F16)
    uint8_t *bits;    // [batch][N]
    int32_t *iters;   // may be null
    uint8_t *conv;    // may be null
    double *final_lam;  // may be null [batch][N]
    double *trace;      // may be null [batch][max_iters+1][N]
    int batch, max_iters, llr_fmt;
    int llr_round16;  // LDPC_F16 context: the LLRs count as stored in fp16 (round on load; a no-op for LLR_F16 input)
    // teacher-forced single step (verification): state in, state out
    int step_mode;
    const double *st_lam;  // [batch][N]
    const void *st_m1, *st_m2;  // [batch][M] CT
    const uint32_t *st_sg;      // [batch][M]
    double *st_ne_out;          // [batch][E] CSR edge order
    const double *st_ne_in;     // [batch][E] (per-edge-message kernels)
    uint8_t *st_syn;            // [batch]
};

// A wave-uniform zero the optimiser cannot see through.  Adding it to the (loop-invariant) graph
// table pointer keeps the table loads and the address arithmetic INSIDE the iteration loop: hoisted,
// 156 x RPL addresses would live in VGPRs for the whole decode and spill.  readfirstlane makes the
// value provably uniform again (inline-asm results count as divergent), so the loads stay s_load.
__device__ __forceinline__ uint32_t opaque_uniform_zero() {
    uint32_t z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return __builtin_amdgcn_readfirstlane(z);
}
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) {  // (a & mask) | (b & ~mask)
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(mask), "v"(a), "v"(b));
    return r;
}

// Geometry of one wave group of the per-edge-message QC kernels.  SZ = circulant size; a block column holds V positions
// in LDS; VT threads serve it.  Power-of-two SZ below 64: CPW = 64/SZ frames interleaved lane by lane fill a wave
// (V = VT = 64).  Any other SZ: one frame, V = SZ positions, VT = SZ rounded up to whole waves (the top lanes idle).
template <int SZ>
struct QcGeom {
    static constexpr bool POW2 = (SZ & (SZ - 1)) == 0;
    static constexpr int CPW = (POW2 && SZ < 64) ? 64 / SZ : 1;
    static constexpr int V = SZ * CPW;
    static constexpr int VT = (V + 63) / 64 * 64;
};
// position wrap inside a block column: a < 2 * (vmask + 1) bytes.  vmask + 1 = V * sizeof(CT); a power of two wraps with
// an AND, anything else with the unsigned-min trick (a - vb wraps to a huge value when a < vb).  vmask is a compile-time
// constant at every call site, so the test folds.
__device__ __forceinline__ uint32_t qc_wrap(uint32_t a, uint32_t vmask) {
    const uint32_t vb = vmask + 1u;
    return (vb & vmask) == 0u ? (a & vmask) : (a < a - vb ? a : a - vb);
}

template <typename CT> struct Bits;
template <> struct Bits<float> { using U = uint32_t; };
template <> struct Bits<double> { using U = uint64_t; };

template <typename CT>
__device__ __forceinline__ CT lds_ld(const char *lds, uint32_t a) { return *reinterpret_cast<const CT *>(lds + a); }
template <typename CT>
__device__ __forceinline__ void lds_st(char *lds, uint32_t a, CT v) { *reinterpret_cast<CT *>(lds + a) = v; }
// SZ = circulant size.  WPF waves per frame, RPL rows per lane and block row, CPW frames per wave.
template <typename CT, class Plan, int SZ>
struct FusedCfg {
    static constexpr int WPF = SZ >= 128 ? 2 : 1;
    static constexpr int THREADS = 64 * WPF;
    static constexpr int RPL = SZ >= THREADS ? SZ / THREADS : 1;
    static constexpr int CPW = SZ >= 64 ? 1 : 64 / SZ;
    static constexpr int V = SZ * CPW;  // "virtual circulant" width in elements (>= 64)
    static constexpr int N = Plan::NBC * SZ;
    static constexpr int M = Plan::NBR * SZ;
    static constexpr int LAM_BYTES = Plan::NBC * V * (int)sizeof(CT);
    static constexpr int LDS_BYTES = LAM_BYTES + (WPF > 1 ? 16 : 0);
    static constexpr int NREC = Plan::NBR * RPL;
    static constexpr int NORIG = Plan::NBC * RPL;
    static constexpr int HSTEP = THREADS * (int)sizeof(CT);  // byte distance between a lane's rows
    // waves per SIMD we ask the register allocator for
    static constexpr int WAVES_PER_EU = (sizeof(CT) == 8) ? (RPL >= 2 ? 1 : 2) : (RPL >= 2 ? 2 : 4);
};


#ifndef LDPC_DEVICE_ONLY
// launcher of the per-edge-message kernels (fused_msg.hip)
bool fused_msg_has(int variant, int dtype, int sz);
int fused_msg_static_id(int sz, const uint16_t *rot, const uint8_t *bc, int nedge);
int fused_msg_launch(int variant, int dtype, int sz, int static_id, hipStream_t st, FusedArgs &a, KernelTimer *timer, LaunchInfo *info);

// four-wave variant with the block rows split between two wave pairs (fused_split.hip)
bool fused_split_has(int variant, int dtype, int sz, int static_id);
int fused_split_launch(int variant, int sz, hipStream_t st, FusedArgs &a, KernelTimer *timer, LaunchInfo *info);
// packed-fp16 min-sum, two frames per lane (fused_pk16.hip): LDPC_F16PK contexts
bool fused_pk16_has(int variant, int sz, int static_id);
int fused_pk16_launch(int sz, hipStream_t st, FusedArgs &a, KernelTimer *timer, LaunchInfo *info);
// row-layered schedule on-chip (fused_layered.hip): min-sum f32, the built-in AR4JA instances
bool fused_layered_has(int variant, int dtype, int sz, int static_id);
int fused_layered_launch(int sz, hipStream_t st, FusedArgs &a, KernelTimer *timer, LaunchInfo *info);
int fused_layered_pk16_launch(int sz, hipStream_t st, FusedArgs &a, KernelTimer *timer, LaunchInfo *info);   // LDPC_F16PK
#endif
constexpr int kSplitMaxIters = 511;  // fused_split_body.h packed result word: bits 23..31 hold the turn a frame converged at

}  // namespace ldpc
