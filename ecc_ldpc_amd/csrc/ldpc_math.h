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
//   * float : min-sum uses the same formulas; the tanh rule is evaluated in the phi domain
//     (phi(x) = -ln tanh(x/2) = 2 atanh(e^-x)):  |ne'| = phi( sum_{j/=n} phi(|t_j|) ), which is
//     the same real-valued function as the tanh product but does not saturate where fp32 tanh
//     rounds to 1.  The +-37.43 clamp is applied where the double
//     reference applies it (its product rounds to exactly +-1 <=> every factor has
//     |t_j|/2 >= 19.0615, i.e. the true value is >= 37.43 anyway).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#define LDPC_V_TANH 0
#define LDPC_V_MINSUM 1

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
template <> struct Store<__half> {
    using CT = float;
    static __device__ __forceinline__ float ld(const __half *p) { return __half2float(*p); }
    // saturate instead of overflowing to inf: an LLR of +-65504 is already "certain"
    static __device__ __forceinline__ void st(__half *p, float v) {
        v = fminf(fmaxf(v, -65504.f), 65504.f);
        *p = __float2half_rn(v);
    }
};

template <typename CT> __device__ __forceinline__ bool hard(CT x) { return x > CT(0); }

// ---------------------------------------------------------------- phi (float)
// phi(x) = -ln(tanh(x/2)) for x >= 0;  phi(0) = +inf, phi(+inf) = 0.  Relative accuracy ~1e-7:
//   x <  0.5 : tanh(y) = y * P(y^2) (odd Taylor series, y = x/2 <= 0.25), phi = -ln(y P)
//   e <  1/16: phi = 2 atanh(e) = 2e (1 + e^2/3 + e^4/5 + e^6/7),  e = exp(-x)
//   else     : phi = ln((1+e)/(1-e))      (1-e in [0.39, 0.94]: no harmful cancellation)
__device__ __forceinline__ float phi_f32(float x) {
    if (x < 0.5f) {
        float y = 0.5f * x, y2 = y * y;
        // tanh(y)/y = 1 - y^2/3 + 2y^4/15 - 17y^6/315 + 62y^8/2835 - 1382y^10/155925
        float p = -1382.0f / 155925.0f;
        p = fmaf(p, y2, 62.0f / 2835.0f);
        p = fmaf(p, y2, -17.0f / 315.0f);
        p = fmaf(p, y2, 2.0f / 15.0f);
        p = fmaf(p, y2, -1.0f / 3.0f);
        p = fmaf(p, y2, 1.0f);
        return -logf(y * p);
    }
    float e = expf(-x);
    if (e < 0.0625f) {
        float e2 = e * e;
        float p = fmaf(e2, 1.0f / 7.0f, 1.0f / 5.0f);
        p = fmaf(p, e2, 1.0f / 3.0f);
        p = fmaf(p, e2, 1.0f);
        return 2.0f * e * p;
    }
    return logf((1.0f + e) / (1.0f - e));
}

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

// tanh rule, float, phi domain with prefix/suffix leave-one-out sums (no subtraction).
template <int DEG>
__device__ __forceinline__ void cn_tanh_f32(float (&t)[DEG]) {
    float ph[DEG];
    unsigned par = 0; // parity of negative factors: tanh(-(t/2)) < 0 <=> t > 0
    bool clampable = true; // the double reference clamps iff every factor rounds to +-1
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        float a = fabsf(t[k]);
        ph[k] = phi_f32(a);
        par ^= (t[k] > 0.f) ? 1u : 0u;
        (void)clampable;
    }
    float suf[DEG];
    float run = 0.f;
#pragma unroll
    for (int k = DEG - 1; k >= 0; k--) { suf[k] = run; run += ph[k]; }
    float pre = 0.f;
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        float S = pre + suf[k];
        pre += ph[k];
        float mag = fminf(phi_f32(S), (float)kNeClamp);
        unsigned neg = par ^ ((t[k] > 0.f) ? 1u : 0u); // sign of the leave-one-out product
        // ne' = -2 atanh'(sign * P) = -sign * mag
        t[k] = neg ? mag : -mag;
    }
}

template <typename CT, int VARIANT, int DEG>
__device__ __forceinline__ void cn_update(CT (&t)[DEG]) {
    if constexpr (VARIANT == LDPC_V_MINSUM) {
        cn_minsum<CT, DEG>(t);
    } else if constexpr (sizeof(CT) == 8) {
        cn_tanh_f64<DEG>(t);
    } else {
        cn_tanh_f32<DEG>(t);
    }
}

}  // namespace ldpc
