// ldpc_math.h -- device-side check-node arithmetic shared by the flood and fused kernels (gfx950).
//
// Semantics restated from the reference (paths relative to the reference repository):
//   tanh rule   src/ECC/Code/LDPC/Reference/Orig.hs:86-91
//                 ne'[m,n] = -2 * atanh' (product [tanh (-((lam_j - ne[m,j]) / 2)) | j /= n])
//   min-sum     src/ECC/Code/LDPC/Reference/Min.hs:78-86
//                 ne'[m,n] = (-3/4) * foldr1 min' [-(lam_j - ne[m,j]) | j /= n]
//                 min' x y = signum x * signum y * min (abs x) (abs y)
//   clamp       src/ECC/Code/LDPC/Utils.hs:113-117   atanh' -> +-18.714973875118524 when infinite
//   hard        src/ECC/Code/LDPC/GPU/Reference.hs:59-60  x > 0
//
// Two numerics families:
//   * double: the reference's own formulas, same operation order (parity mode; min-sum is
//     bit-exact with the CPU reference, tanh differs only by the device libm's last ulp).
//   * float : min-sum uses the same formulas; the tanh rule is evaluated by a hyperbolic recurrence
//     (cn_tanh_f32 below): the same real-valued function as the tanh product, but it does not
//     saturate where an fp32 tanh rounds to 1.  The +-37.43 clamp is
//     applied where the double reference applies it (its product rounds to exactly +-1 <=> every factor
//     has |t_j|/2 >= 19.0615, i.e. the true value is >= 37.43 anyway).
#pragma once
#if defined(__HIPCC_RTC__) || defined(LDPC_JIT)
// run-time compilation (jit.cc: hiprtc, or `hipcc --genco -DLDPC_JIT`): device code only -- hiprtc has no host headers
#define LDPC_DEVICE_ONLY 1
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>
namespace ldpc { enum { LLR_F32 = 0, LLR_F64 = 1, LLR_F16 = 2 }; }
#ifndef INFINITY
#define INFINITY __builtin_inff()
#endif
#else
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "internal.h"
#endif
#include <type_traits>

// Marks the enclosing basic block as a COLD side path of the iteration loop (syndrome-only last turn, convergence
// snapshot, ...).  Emits only an assembler comment; tools/isa_histogram.py reads it from the compiler's .s output to
// tell the ordinary turn of the loop from the rare ones when it counts instructions.
#define LDPC_COLD_PATH() asm volatile("; ldpc.cold")
// first statement of the body of the BP iteration loop when that loop is nested in another (a persistent workgroup's loop over
// frames): tools/isa_histogram.py then prices the innermost loop around this marker instead of the outermost loop
#define LDPC_TURN_LOOP() asm volatile("; ldpc.turnloop")

#define LDPC_V_TANH 0
#define LDPC_V_MINSUM 1
#define LDPC_V_TANH_CM 2   // the reference's `arraylet-cm` numerics (Fast/CachedMult.hs): f64 only, flood path only
#define LDPC_V_TANH_CUDA32 3   // the reference's `cuda-arraylet2` numerics (cudabits/arraylet2.cu, common.h): f32 only, flood path only

namespace ldpc {

constexpr double kAtanhClamp = 18.714973875118524;   // Utils.hs:115
constexpr double kNeClamp = 2.0 * kAtanhClamp;       // largest |ne'| the tanh rule can return

// ---------------------------------------------------------------- storage <-> compute conversion
template <typename ST> struct Store;
template <> struct Store<float> {
    using CT = float;
    static __device__ __forceinline__ float ld(const float *p) { return *p; }
    static __device__ __forceinline__ void st(float *p, float v) { *p = v; }
};
template <> struct Store<double> {
    using CT = double;
    static __device__ __forceinline__ double ld(const double *p) { return *p; }
    static __device__ __forceinline__ void st(double *p, double v) { *p = v; }
};
#ifndef LDPC_DEVICE_ONLY
template <> struct Store<__half> {
    using CT = float;
    static __device__ __forceinline__ float ld(const __half *p) { return __half2float(*p); }
    // saturate instead of overflowing to inf: an LLR of +-65504 is already "certain"
    static __device__ __forceinline__ void st(__half *p, float v) {
        v = fminf(fmaxf(v, -65504.f), 65504.f);
        *p = __float2half_rn(v);
    }
};
#endif

template <typename CT> __device__ __forceinline__ bool hard(CT x) { return x > CT(0); }

// ---------------------------------------------------------------- channel LLR input formats
// (LLR_F32 / LLR_F64 / LLR_F16: internal.h)
template <typename CT> __device__ __forceinline__ CT load_llr(const void *base, size_t i, int fmt) {
    if (fmt == LLR_F64) return (CT) reinterpret_cast<const double *>(base)[i];
    if (fmt == LLR_F16) return (CT) (float) reinterpret_cast<const _Float16 *>(base)[i];
    return (CT) reinterpret_cast<const float *>(base)[i];
}
// compile-time format: lets a caller dispatch on the format ONCE around a batch of loads, so that the loads
// of one thread issue back to back instead of each sitting behind its own three-way branch
template <typename CT, int FMT> __device__ __forceinline__ CT load_llr_as(const void *base, size_t i) {
    if constexpr (FMT == LLR_F64) return (CT) reinterpret_cast<const double *>(base)[i];
    else if constexpr (FMT == LLR_F16) return (CT) (float) reinterpret_cast<const _Float16 *>(base)[i];
    else return (CT) reinterpret_cast<const float *>(base)[i];
}
template <class F> __device__ __forceinline__ void with_llr_format(int fmt, F &&f) {
    if (fmt == LLR_F64) f(std::integral_constant<int, LLR_F64>{});
    else if (fmt == LLR_F16) f(std::integral_constant<int, LLR_F16>{});
    else f(std::integral_constant<int, LLR_F32>{});
}
// value a float LLR has after being stored as fp16 (LDPC_F16 contexts): saturating round-to-nearest-even
__device__ __forceinline__ float round_f16(float v) {
    v = fminf(fmaxf(v, -65504.f), 65504.f);
    return (float)(_Float16)v;   // IEEE binary16, round to nearest even (== __float2half_rn)
}
template <typename CT> __device__ __forceinline__ CT maybe_round_f16(CT v, int on) {
    if constexpr (sizeof(CT) == 4) return on ? round_f16(v) : v;
    else return v;
}

// ---------------------------------------------------------------- LLR range of the any-H min-sum kernels below f64
// Flooding min-sum is homogeneous: on a graph with heavy columns the LLRs of a frame that does not converge grow by up to
// 3/4 (column weight - 1) per turn -- codes/1920.1280.A (weight 18): x6 per turn measured, 3e9 after 12 turns, past
// FLT_MAX before turn 50.  The reference's Double never gets there (1e39 after 50 turns); a float does, and then
// inf - inf = NaN, hard(NaN) = False on every bit, an all-zero "codeword" whose syndrome is zero: a frame the reference
// reports as failed would come back "converged".  Because the rule is homogeneous -- (lam, ne, orig) -> 2^-k (lam, ne, orig) maps
// a trajectory onto itself, and a multiplication by a power of two is exact -- the kernels that take ANY matrix (fused_csr.hip,
// flood.hip) RESCALE a frame instead of clipping it (r04; r03 clipped at +-2^100, which left the float trajectory where the
// Double one goes on): when a variable-node pass leaves some |lam| above 2^60, the frame's lam and messages are multiplied by
// 2^-40 and its channel LLRs enter later sums with the accumulated factor.  Hard decisions, syndromes and turn counts are those
// of a float with unbounded exponent; LLRs are handed out multiplied back (in double).  Only when the CHANNEL term falls below
// the subnormal range (after ~3 rescalings, LLRs 120 binary orders below the messages they are added to) does anything
// differ, and then by less than an ulp of any sum it enters.  f32 state only: fp16 storage saturates by its own rule, f64 is the
// parity mode (the reference's arithmetic, overflow and all), the tanh rule is bounded (|ne'| <= 37.43, Utils.hs:113-117).
constexpr float kLamBig = 0x1p60f;        // rescale when a column's new LLR passes this ...
constexpr int kRescaleExp = 40;           // ... by 2^-40
constexpr float kRescale = 0x1p-40f;
template <typename CT, int VARIANT> constexpr bool kRescales = (VARIANT == LDPC_V_MINSUM && sizeof(CT) == 4);
// The QUASI-CYCLIC f32 min-sum kernels (split / two-wave / run-time specialised, on-chip and HBM layered, flood_qc_kernel) do not
// rescale -- on the shipped and synthetic QC shapes |lam| stays at 10-17 for hundreds of turns, and their turn loops have no room for
// it -- but the failure it prevents must not pass silently there either: a frame whose LLRs left the float range (inf - inf = NaN,
// hard NaN = False, an all-zero "codeword") is examined when its stop rule fires, once per frame, and any LLR that is not finite turns
// "converged" into "failed" (the channel's hard decisions, the iteration limit as its count): what the reference reports for a frame
// it cannot decode, never a codeword it did not find.
template <typename CT, int VARIANT> constexpr bool kVetoesNonFinite = (VARIANT == LDPC_V_MINSUM && sizeof(CT) == 4);
__device__ __forceinline__ bool not_finite(float v) { return (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u; }
__device__ __forceinline__ bool not_finite(double v) { return !(fabs(v) <= 1.7976931348623157e308); }

// ---------------------------------------------------------------- check-node update, DEG known
// t[k] = lam_k - ne_k (the reference's list element is -(t[k]) for min-sum and
// tanh(-(t[k]/2)) for the tanh rule).  Results overwrite t[k] with ne'_k.

// min-sum, any compute type.  Exact: the only rounding is the final multiply by -3/4, as in Min.hs:78.
template <typename CT, int DEG>
__device__ __forceinline__ void cn_minsum(CT (&t)[DEG]) {
    static_assert(DEG >= 2, "min-sum needs degree >= 2");
    CT m1 = fabs(t[0]), m2 = CT(INFINITY);
    int i1 = 0;
    unsigned par = (t[0] > CT(0)) ? 1u : 0u; // x_j = -t_j is negative <=> t_j > 0
#pragma unroll
    for (int k = 1; k < DEG; k++) {
        CT a = fabs(t[k]);
        par ^= (t[k] > CT(0)) ? 1u : 0u;
        if (a < m1) { m2 = m1; m1 = a; i1 = k; }
        else if (a < m2) { m2 = a; }
    }
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        CT mag = (k == i1) ? m2 : m1;
        unsigned neg = par ^ ((t[k] > CT(0)) ? 1u : 0u); // sign of prod_{j/=k} x_j
        CT acc = neg ? -mag : mag;                       // foldr1 min' ...
        t[k] = CT(-0.75) * acc;                          // (-3/4) * ...
    }
}

// tanh rule, double: the reference formula literally (left-fold product, base-4.9 atanh).
template <int DEG>
__device__ __forceinline__ void cn_tanh_f64(double (&t)[DEG]) {
    double th[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) th[k] = tanh(-(t[k] / 2.0));
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        double prod = 1.0;
#pragma unroll
        for (int j = 0; j < DEG; j++)
            if (j != k) prod = prod * th[j];
        double y = 0.5 * log((1.0 + prod) / (1.0 - prod));
        if (isinf(y)) y = (prod > 0.0 ? 1.0 : -1.0) * kAtanhClamp;
        t[k] = -2.0 * y;
    }
}

// tanh rule with the row product cached as a StableDiv -- the reference's `arraylet-cm` decoder,
// src/ECC/Code/LDPC/Fast/CachedMult.hs:25-56 (StableDiv, lit, smult, sdiv) and :247-259: per row
//   S = foldr1 smult [lit x_j | ascending column],  x_j = tanh(-(t_j/2)),   ne'_k = -2 atanh' (S `sdiv` x_k)
// StableDiv (a, b): a = the factor closest to zero, b = the product of the others.  Same real function as cn_tanh_f64,
// different roundings.
template <int DEG>
__device__ __forceinline__ void cn_tanh_cm_f64(double (&t)[DEG]) {
    double x[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) x[k] = tanh(-(t[k] / 2.0));
    double sa, sb;
    { const double v = x[DEG - 1]; if (v >= 1.0) { sa = 1.0; sb = v; } else { sa = v; sb = 1.0; } }            // lit (:41-44)
#pragma unroll
    for (int k = DEG - 2; k >= 0; k--) {                                                                          // smult (lit x_k) acc (:46-50)
        const double v = x[k];
        const double a = v >= 1.0 ? 1.0 : v, bq = v >= 1.0 ? v : 1.0;
        const bool a_smaller = fabs(a) < fabs(sa);                                                                // absMinMax (:31-34)
        const double mn = a_smaller ? a : sa, mx = a_smaller ? sa : a;
        sb = (bq * mx) * sb;
        sa = mn;
    }
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        const double c = x[k];
        const double q = (sa == c) ? sb : sa * (sb / c);                                                          // sdiv (:52-55)
        double y = 0.5 * log((1.0 + q) / (1.0 - q));
        if (isinf(y)) y = (q > 0.0 ? 1.0 : (q < 0.0 ? -1.0 : q)) * kAtanhClamp;
        t[k] = -2.0 * y;
    }
}

// tanh rule, float: hyperbolic-recurrence form.  With e_j = e^-|t_j|, tanh(|t_j|/2) = (1 - e_j)/(1 + e_j), so
//     prod_{j /= k} tanh(|t_j|/2) = N / D,    D = prod (1 + e_j),  N = prod (1 - e_j)
//     |ne'_k| = 2 atanh(N/D) = ln((D + N) / (D - N)) = ln(A / S),     A = D + N,  S = D - N.
// A and S obey  (A, S) (+) e  =  (A + e S,  S + e A)   starting from (2, 0): only sums of positive terms, so S
// -- the quantity an fp32 tanh product loses when it rounds to 1 -- keeps its
// relative accuracy, and no division is needed per factor.  Two partial results combine like cosh/sinh of a sum:
//     (A1, S1) (+) (A2, S2) = (A1 A2 + S1 S2,  A1 S2 + S1 A2)      (a common factor 1/2 is dropped: only A/S matters)
// which gives leave-one-out from prefix/suffix passes.  Same real function as Orig.hs:86-91; 3 hardware
// transcendentals per edge (v_exp, v_rcp, v_log) -- the phi-domain form used first needed 8, the product/complement
// form after it 4.  A zero t_j gives e = 1, hence A = S EXACTLY from there on (the two updates become the same
// operation; the combine is written symmetrically, products rounded separately, for the same reason), so
// ne' = ln(1 + (A - S)/S) = 0 exactly for the other edges of the row, as a zero factor does in the reference.
// S = 0 (every other |t| enormous) gives ln(inf) -> the +-37.43 clamp, which is where the double reference clamps
// as well (its product rounds to exactly +-1).  Emulated in float32 with +-1 ulp transcendentals against the exact
// value: error <= 7e-7 * max(1, |ne'|) for row weights 2..32 and |t| up to 200.
struct TanhAS {
    float A, S;
    static __device__ __forceinline__ TanhAS one() { return {2.0f, 0.0f}; }
    static __device__ __forceinline__ float e_of(float a) { return __builtin_amdgcn_exp2f(a * -1.44269504088896340736f); }  // a >= 0
    __device__ __forceinline__ TanhAS times(float e) const { return {fmaf(e, S, A), fmaf(e, A, S)}; }
    __device__ __forceinline__ TanhAS times(const TanhAS &o) const {
        return {A * o.A + S * o.S, A * o.S + S * o.A};   // not contracted (-ffp-contract=off): symmetric roundings
    }
    __device__ __forceinline__ float mag() const {
        const float w = (A - S) * __builtin_amdgcn_rcpf(S);
        const float m = __builtin_amdgcn_logf(1.0f + w) * 0.69314718055994530942f;
        return fminf(m, (float)kNeClamp);  // also turns the inf of S == 0 into the clamp value
    }
};

template <int DEG>
__device__ __forceinline__ void cn_tanh_f32(float (&t)[DEG]) {
    static_assert(DEG <= 32, "sign word holds 32 edges");
    uint32_t sg = 0, X = 0;  // sg bit (DEG-1-k) = sign bit of t_k ; X bit 31 = parity of the sign bits
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        uint32_t tb = __float_as_uint(t[k]);
        X ^= tb;
        sg = __builtin_amdgcn_alignbit(sg, tb, 31);
        t[k] = TanhAS::e_of(fabsf(t[k]));
    }
    TanhAS suf[DEG];
    TanhAS run = TanhAS::one();
#pragma unroll
    for (int k = DEG - 1; k >= 0; k--) { suf[k] = run; run = run.times(t[k]); }
    // factor_j = tanh(-(t_j/2)) is negative iff t_j > 0; with zero factors the magnitude is 0 and the sign
    // is irrelevant, so "t_j > 0" may be read off the sign bit: negative factors among j != k are
    // (DEG-1) - sum_{j != k} signbit_j.  ne'_k = -sign(prod) * mag: positive iff that count is odd.
    const uint32_t base = X ^ ((DEG & 1) ? 0u : 0x80000000u);  // bit 31: parity((DEG-1) + all sign bits)
    run = TanhAS::one();
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        const float e = t[k];
        const float mag = run.times(suf[k]).mag();
        run = run.times(e);
        // count parity for edge k = base ^ signbit_k ; ne' > 0 iff odd -> sign bit of ne' = NOT that
        uint32_t sk = (sg << (31 - (DEG - 1 - k)));
        uint32_t neg = ~(base ^ sk) & 0x80000000u;
        t[k] = __uint_as_float(__float_as_uint(mag) | neg);
    }
}

// Rows of weight <= 4 (a code whose heaviest row has weight 4: codes/1920.1280.3.303, BASELINE configs[2]): the four
// leave-one-out (A, S) pairs straight from two PAIR products instead of prefix / suffix chains --
//     P01 = e0 (+) e1 = (1 + e0 e1, e0 + e1),  P23 likewise (the common factor 2 of TanhAS::one() dropped: only A/S matters)
//     edge 0: P23 (+) e1,  edge 1: P23 (+) e0,  edge 2: P01 (+) e3,  edge 3: P01 (+) e2          ((A, S) (+) e = (A + e S, S + e A))
// 4 + 8 fused multiply-adds per row against 40 operations in the chained form (r03: -15 % VALU per turn of the configs[2]
// kernel).  Same real function; the properties the chained form is built around hold here too: a zero t_j (e = 1) gives A = S
// EXACTLY for every other edge (fma(1, e, 1) and 1 + e round alike; fma(1, S, A) and fma(1, A, S) too), hence ne' = 0
// exactly, and a padded slot (t = +inf, e = 0) is the neutral factor, so a row of weight 2 or 3 padded to 4 gets exactly what
// the formula written for its own weight would give.  Every path that can see such a code uses THIS form (the on-chip
// kernels' DMAX = 4 instances, and flood.hip when FloodDev::pairs4 is set), so that they keep agreeing bit for bit.
__device__ __forceinline__ void cn_tanh_f32_pairs4(float (&t)[4], int deg) {
    uint32_t sg[4], X = 0;
    float e[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        sg[k] = __float_as_uint(t[k]) & 0x80000000u;
        X ^= sg[k];
        e[k] = TanhAS::e_of(fabsf(t[k]));   // padding: |t| = +inf -> e = 0
    }
    const TanhAS p01{fmaf(e[0], e[1], 1.0f), e[0] + e[1]}, p23{fmaf(e[2], e[3], 1.0f), e[2] + e[3]};
    const float mag[4] = {p23.times(e[1]).mag(), p23.times(e[0]).mag(), p01.times(e[3]).mag(), p01.times(e[2]).mag()};
    // sign as in cn_tanh_f32: ne'_k > 0 iff the number of negative factors among j /= k is odd, factor_j < 0 iff t_j > 0;
    // padded slots carry sign bit 0 like a positive t and are taken back out through the real degree
    const uint32_t base = X ^ ((deg & 1) ? 0u : 0x80000000u);
#pragma unroll
    for (int k = 0; k < 4; k++) t[k] = __uint_as_float(__float_as_uint(mag[k]) | (~(base ^ sg[k]) & 0x80000000u));
}

// ---------------------------------------------------------------- the reference's CUDA plug-in, operation by operation (parity mode)
// cudabits/arraylet2.cu:43-83 selfProduct + common.h:82-88 atanh_ (float_ty = float): a factor is the DOUBLE tanh of
// -((double) lam - (double) ne) / 2 stored as a float; the leave-one-out product runs in a double register over the row in ascending
// block column; atanh_ takes it as a FLOAT -- the clamp fires when the product ROUNDS to +-1 in float -- and is the float atanh
// otherwise; the message is -2 x that.  l[k], m[k]: the row's LLRs and old messages; m[k] <- the new message.  Slots k >= deg are not
// touched (an absent block holds the factor 1 in the reference).  Specification: oracle/ldpc_oracle.c ORACLE_CUDA32 (device tanh /
// atanhf against the C library's: last ulps, teacher-forced 1e-5).
template <int DMAX>
__device__ __forceinline__ void cn_tanh_cuda32(const float (&l)[DMAX], float (&m)[DMAX], int deg) {
    float th[DMAX];
#pragma unroll
    for (int k = 0; k < DMAX; k++) th[k] = k < deg ? (float)tanh(-(((double)l[k] - (double)m[k]) / 2.0)) : 1.0f;
#pragma unroll
    for (int k = 0; k < DMAX; k++) {
        if (k < deg) {
            double r = 1.0;
#pragma unroll
            for (int j = 0; j < DMAX; j++)
                if (j != k && j < deg) r *= (double)th[j];
            const float x = (float)r;
            const float y = (x == 1.0f || x == -1.0f) ? (float)((x < 0.0f ? -1.0 : 1.0) * kAtanhClamp) : atanhf(x);
            m[k] = -2.0f * y;
        }
    }
}

// ---------------------------------------------------------------- padded rows (generic on-chip kernel)
// Rows of any degree <= DMAX: slots k >= deg hold t = +inf, which is neutral for both rules
// (|t| = inf never wins a min; e^-inf = 0 is the neutral factor; sign bit 0) -- only the "(D odd)" term of the
// sign rule needs the real degree.  Results for the real slots are bit-identical to cn_update<deg>.
template <typename CT, int VARIANT, int DMAX>
__device__ __forceinline__ void cn_update_padded(CT (&t)[DMAX], int deg) {
    if constexpr (VARIANT == LDPC_V_MINSUM) {
        if constexpr (sizeof(CT) == 4) {
            uint32_t X = 0;
            float m1 = INFINITY, m2 = INFINITY;
#pragma unroll
            for (int k = 0; k < DMAX; k++) {
                X ^= __float_as_uint(t[k]);
                float a = fabsf(t[k]);
                m2 = __builtin_amdgcn_fmed3f(m1, m2, a);
                m1 = fminf(m1, a);
            }
            const uint32_t flip = (X ^ ((deg & 1) ? 0x80000000u : 0u)) & 0x80000000u;
            const uint32_t c1 = __float_as_uint(0.75f * m1) ^ flip;
            const uint32_t c2 = __float_as_uint(0.75f * m2) ^ flip;
#pragma unroll
            for (int k = 0; k < DMAX; k++) {
                uint32_t c = (fabsf(t[k]) == m1) ? c2 : c1;
                t[k] = __uint_as_float(__builtin_amdgcn_bitop3_b32(c, __float_as_uint(t[k]), 0x80000000u, 0x78));
            }
        } else {
            CT m1 = CT(INFINITY), m2 = CT(INFINITY);
            unsigned par = 0;
#pragma unroll
            for (int k = 0; k < DMAX; k++) {
                CT a = fabs(t[k]);
                par ^= (k < deg && t[k] > CT(0)) ? 1u : 0u;
                if (a < m1) { m2 = m1; m1 = a; }
                else if (a < m2) { m2 = a; }
            }
#pragma unroll
            for (int k = 0; k < DMAX; k++) {
                CT mag = (fabs(t[k]) == m1) ? m2 : m1;
                unsigned neg = par ^ ((t[k] > CT(0)) ? 1u : 0u);
                t[k] = CT(-0.75) * (neg ? -mag : mag);
            }
        }
    } else if constexpr (sizeof(CT) == 8) {
        double th[DMAX];
#pragma unroll
        for (int k = 0; k < DMAX; k++) th[k] = tanh(-(t[k] / 2.0));
#pragma unroll
        for (int k = 0; k < DMAX; k++) {
            double prod = 1.0;
#pragma unroll
            for (int j = 0; j < DMAX; j++)
                if (j != k && j < deg) prod = prod * th[j];
            double y = 0.5 * log((1.0 + prod) / (1.0 - prod));
            if (isinf(y)) y = (prod > 0.0 ? 1.0 : -1.0) * kAtanhClamp;
            t[k] = -2.0 * y;
        }
    } else if constexpr (DMAX == 4) {
        cn_tanh_f32_pairs4(t, deg);
    } else {
        static_assert(DMAX <= 32, "sign word holds 32 edges");
        uint32_t sg = 0, X = 0;
#pragma unroll
        for (int k = 0; k < DMAX; k++) {
            uint32_t tb = __float_as_uint(t[k]);
            X ^= tb;
            sg = __builtin_amdgcn_alignbit(sg, tb, 31);
            t[k] = TanhAS::e_of(fabsf(t[k]));   // padding: a = +inf -> e = 0: the neutral factor
        }
        TanhAS suf[DMAX];
        TanhAS run = TanhAS::one();
#pragma unroll
        for (int k = DMAX - 1; k >= 0; k--) { suf[k] = run; run = run.times(t[k]); }
        const uint32_t base = X ^ ((deg & 1) ? 0u : 0x80000000u);
        run = TanhAS::one();
#pragma unroll
        for (int k = 0; k < DMAX; k++) {
            const float e = t[k];
            const float mag = run.times(suf[k]).mag();
            run = run.times(e);
            uint32_t sk = (sg << (31 - (DMAX - 1 - k)));
            uint32_t neg = ~(base ^ sk) & 0x80000000u;
            t[k] = __uint_as_float(__float_as_uint(mag) | neg);
        }
    }
}

template <typename CT, int VARIANT, int DEG>
__device__ __forceinline__ void cn_update(CT (&t)[DEG]) {
    if constexpr (VARIANT == LDPC_V_MINSUM) {
        cn_minsum<CT, DEG>(t);
    } else if constexpr (VARIANT == LDPC_V_TANH_CM && sizeof(CT) == 8) {
        cn_tanh_cm_f64<DEG>(t);
    } else if constexpr (sizeof(CT) == 8) {
        cn_tanh_f64<DEG>(t);
    } else {
        cn_tanh_f32<DEG>(t);
    }
}

}  // namespace ldpc
