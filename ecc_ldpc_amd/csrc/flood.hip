// flood.hip -- generic flooding BP path: state resident in HBM, two kernels per iteration.
//
// Works for ANY parity-check matrix given as CSR (the `Matrix Bool` decoders of the reference:
// Reference/Orig.hs:30-31, Reference/Min.hs:33-34) and is the fallback for codes the fused
// on-chip kernel (fused.hip) does not cover.
//
// Data layout (batch-major, "one codeword per wavefront lane"):
//   lam  [N][Bp]   a-posteriori LLRs          orig [N][Bp]  channel LLRs
//   msg  [E][Bp]   check->variable messages (ne), CSR edge order
//   Bp = batch padded to a multiple of 64.  Lane l of a wave owns codeword 64*slab + l, so every
//   global access of a wave is one contiguous 256-byte (fp32) segment and every graph index
//   (row_ptr / col_idx / csc tables) is wave-uniform -> scalar loads.
//
// One loop turn n of Reference/Orig.hs:67-71:
//   flood_cn : per (row, 64 codewords): t_j = lam[col_j] - msg[e_j]; row parity of hard(lam)
//              (the syndrome, Orig.hs:73-78) -> unsat stamp; ne' (Orig.hs:81-92 / Min.hs:75-87)
//              written in place.
//   flood_vn : per (column, 64 codewords): if the frame's syndrome was zero it is frozen
//              (done, iters = n: Orig.hs:69 "return lam"); else lam' = foldr (+) orig (col of
//              ne') in the reference's order (Orig.hs:96): descending row.
// Algorithmic HBM bytes per frame per turn: CN reads msg (E) + lam (N, given L2 reuse of the
// slab), writes msg (E); VN reads msg (E) + orig (N), writes lam (N)  => (3E + 3N) * sizeof(ST),
// the B_iter of SURVEY.md section 8d.
#include <stdlib.h>
#include <string.h>
#include "ldpc_math.h"
#include "internal.h"

namespace ldpc {

constexpr int kWave = 64;
constexpr int kCnWaves = 4; // rows per CN block

// ------------------------------------------------------------------ CN + syndrome
template <typename ST, int VARIANT, int DEG>
__device__ __forceinline__ void cn_row_regs(const FloodDev &d, ST *__restrict__ msg,
                                            const ST *__restrict__ lam, int ebeg, size_t b,
                                            int stamp, bool syndrome_only) {
    using CT = typename Store<ST>::CT;
    CT t[DEG];
    unsigned par = 0;
    // column indices first (wave-uniform scalar loads, issued together), then every lam / message load, then the
    // arithmetic: written as one loop the row paid one scalar-load round trip per edge before its vector loads
    int col[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) col[k] = d.col_idx[ebeg + k];
    CT l[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) l[k] = Store<ST>::ld(lam + (size_t)col[k] * d.Bp + b);
    if (!syndrome_only) {
#pragma unroll
        for (int k = 0; k < DEG; k++) t[k] = Store<ST>::ld(msg + (size_t)(ebeg + k) * d.Bp + b);
    }
    if constexpr (VARIANT == LDPC_V_TANH_CUDA32) {   // the CUDA plug-in's own arithmetic: lam - ne is formed in double inside (ldpc_math.h)
#pragma unroll
        for (int k = 0; k < DEG; k++) par ^= hard(l[k]) ? 1u : 0u;
        if (par) d.unsat[b] = stamp;
        if (syndrome_only) return;
        if constexpr (sizeof(CT) == 4) cn_tanh_cuda32<DEG>(l, t, DEG);
#pragma unroll
        for (int k = 0; k < DEG; k++) Store<ST>::st(msg + (size_t)(ebeg + k) * d.Bp + b, t[k]);
        return;
    }
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        par ^= hard(l[k]) ? 1u : 0u;
        if (!syndrome_only) t[k] = l[k] - t[k];
    }
    if (par) d.unsat[b] = stamp; // benign race: every writer stores the same value
    if (syndrome_only) return;
    if constexpr (VARIANT == LDPC_V_TANH && sizeof(CT) == 4 && DEG <= 4) {
        if (d.pairs4) {   // (uniform) a code of row weight <= 4: the pair-product form, bit-identical with the on-chip DMAX = 4 kernels
            float p[4];
#pragma unroll
            for (int k = 0; k < 4; k++) p[k] = k < DEG ? t[k] : INFINITY;
            cn_tanh_f32_pairs4(p, DEG);
#pragma unroll
            for (int k = 0; k < DEG; k++) Store<ST>::st(msg + (size_t)(ebeg + k) * d.Bp + b, p[k]);
            return;
        }
    }
    cn_update<CT, VARIANT, DEG>(t);
#pragma unroll
    for (int k = 0; k < DEG; k++) Store<ST>::st(msg + (size_t)(ebeg + k) * d.Bp + b, t[k]);
}

// row weights 9..32 without an exact-degree instance: the row in DMAX registers, slots k >= deg padded with
// t = +inf (neutral for both rules; cn_update_padded is bit-identical to the exact-degree code, ldpc_math.h).
// Loads and stores of the padding slots are skipped by wave-uniform branches.
template <typename ST, int VARIANT, int DMAX>
__device__ __forceinline__ void cn_row_padded(const FloodDev &d, ST *__restrict__ msg, const ST *__restrict__ lam,
                                              int ebeg, int deg, size_t b, int stamp, bool syndrome_only) {
    using CT = typename Store<ST>::CT;
    int col[DMAX];
#pragma unroll
    for (int k = 0; k < DMAX; k++) col[k] = (k < deg) ? d.col_idx[ebeg + k] : 0;
    CT l[DMAX], t[DMAX];
#pragma unroll
    for (int k = 0; k < DMAX; k++) l[k] = (k < deg) ? Store<ST>::ld(lam + (size_t)col[k] * d.Bp + b) : CT(0);
    if (!syndrome_only) {
#pragma unroll
        for (int k = 0; k < DMAX; k++) t[k] = (k < deg) ? Store<ST>::ld(msg + (size_t)(ebeg + k) * d.Bp + b) : CT(0);
    }
    unsigned par = 0;
    if constexpr (VARIANT == LDPC_V_TANH_CUDA32) {
#pragma unroll
        for (int k = 0; k < DMAX; k++) par ^= (k < deg && hard(l[k])) ? 1u : 0u;
        if (par) d.unsat[b] = stamp;
        if (syndrome_only) return;
        if constexpr (sizeof(CT) == 4) cn_tanh_cuda32<DMAX>(l, t, deg);
#pragma unroll
        for (int k = 0; k < DMAX; k++)
            if (k < deg) Store<ST>::st(msg + (size_t)(ebeg + k) * d.Bp + b, t[k]);
        return;
    }
#pragma unroll
    for (int k = 0; k < DMAX; k++) {
        par ^= (k < deg && hard(l[k])) ? 1u : 0u;
        if (!syndrome_only) t[k] = (k < deg) ? l[k] - t[k] : CT(INFINITY);
    }
    if (par) d.unsat[b] = stamp; // benign race: every writer stores the same value
    if (syndrome_only) return;
    cn_update_padded<CT, VARIANT, DMAX>(t, deg);
#pragma unroll
    for (int k = 0; k < DMAX; k++)
        if (k < deg) Store<ST>::st(msg + (size_t)(ebeg + k) * d.Bp + b, t[k]);
}

// any degree (here: above 32): O(deg^2) re-reads (served by L1/L2); correctness path for unusual row weights.
template <typename ST, int VARIANT>
__device__ void cn_row_generic(const FloodDev &d, ST *__restrict__ msg, ST *__restrict__ scratch,
                               const ST *__restrict__ lam, int ebeg, int deg, size_t b, int stamp,
                               bool syndrome_only) {
    using CT = typename Store<ST>::CT;
    unsigned par = 0;
    for (int k = 0; k < deg; k++) {
        int col = d.col_idx[ebeg + k];
        par ^= hard(Store<ST>::ld(lam + (size_t)col * d.Bp + b)) ? 1u : 0u;
    }
    if (par) d.unsat[b] = stamp;
    if (syndrome_only) return;
    // new messages go to scratch[e] first (old ones are still needed by the other outputs)
    for (int k = 0; k < deg; k++) {
        CT out;
        if constexpr (VARIANT == LDPC_V_MINSUM) {
            CT mag = CT(INFINITY);
            unsigned neg = 0;
            for (int j = 0; j < deg; j++) {
                if (j == k) continue;
                int col = d.col_idx[ebeg + j];
                CT tj = Store<ST>::ld(lam + (size_t)col * d.Bp + b) - Store<ST>::ld(msg + (size_t)(ebeg + j) * d.Bp + b);
                CT a = fabs(tj);
                mag = a < mag ? a : mag;
                neg ^= (tj > CT(0)) ? 1u : 0u;
            }
            out = CT(-0.75) * (neg ? -mag : mag);
        } else if constexpr (VARIANT == LDPC_V_TANH_CM && sizeof(CT) == 8) {
            // arraylet-cm: the whole row's StableDiv (right fold), then `sdiv` by this edge's factor
            double sa = 0, sb = 0, xk = 0;
            for (int j = deg - 1; j >= 0; j--) {
                int col = d.col_idx[ebeg + j];
                double tj = Store<ST>::ld(lam + (size_t)col * d.Bp + b) - Store<ST>::ld(msg + (size_t)(ebeg + j) * d.Bp + b);
                const double v = tanh(-(tj / 2.0));
                if (j == k) xk = v;
                const double a = v >= 1.0 ? 1.0 : v, bq = v >= 1.0 ? v : 1.0;
                if (j == deg - 1) { sa = a; sb = bq; }
                else {
                    const bool a_smaller = fabs(a) < fabs(sa);
                    const double mn = a_smaller ? a : sa, mx = a_smaller ? sa : a;
                    sb = (bq * mx) * sb;
                    sa = mn;
                }
            }
            const double q = (sa == xk) ? sb : sa * (sb / xk);
            double y = 0.5 * log((1.0 + q) / (1.0 - q));
            if (isinf(y)) y = (q > 0.0 ? 1.0 : (q < 0.0 ? -1.0 : q)) * kAtanhClamp;
            out = -2.0 * y;
        } else if constexpr (sizeof(CT) == 8) {
            double prod = 1.0;
            for (int j = 0; j < deg; j++) {
                if (j == k) continue;
                int col = d.col_idx[ebeg + j];
                double tj = Store<ST>::ld(lam + (size_t)col * d.Bp + b) - Store<ST>::ld(msg + (size_t)(ebeg + j) * d.Bp + b);
                prod = prod * tanh(-(tj / 2.0));
            }
            double y = 0.5 * log((1.0 + prod) / (1.0 - prod));
            if (isinf(y)) y = (prod > 0.0 ? 1.0 : -1.0) * kAtanhClamp;
            out = -2.0 * y;
        } else {
            TanhAS acc = TanhAS::one();
            unsigned neg = 0;
            for (int j = 0; j < deg; j++) {
                if (j == k) continue;
                int col = d.col_idx[ebeg + j];
                float tj = Store<ST>::ld(lam + (size_t)col * d.Bp + b) - Store<ST>::ld(msg + (size_t)(ebeg + j) * d.Bp + b);
                acc = acc.times(TanhAS::e_of(fabsf(tj)));
                neg ^= (tj > 0.f) ? 1u : 0u;
            }
            float mag = acc.mag();
            out = neg ? mag : -mag;
        }
        Store<ST>::st(scratch + (size_t)(ebeg + k) * d.Bp + b, out);
    }
    for (int k = 0; k < deg; k++)
        msg[(size_t)(ebeg + k) * d.Bp + b] = scratch[(size_t)(ebeg + k) * d.Bp + b];
}

// grid: 1-D, (Bp/64) slabs x ceil(M/kCnWaves) row groups, remapped so that all row groups of one
// codeword slab run back to back on ONE XCD (blocks are dealt round-robin over the 8 XCDs, so
// block ids congruent mod 8 share an L2): lam[col] of the slab is then re-read from that L2.
// WIDE = false: the kernel every code runs (exact-degree rows 1..8 and 18, O(d^2) fallback above 32); rows of weight
// 9..32 are left to the WIDE = true instance, launched right after it only for codes that have such rows -- kept
// apart because their 12..32-register rows would otherwise set the register count (56 -> 78..107 VGPRs, 8 -> 4..6
// waves/SIMD) of a kernel that lives on occupancy.
template <typename ST, int VARIANT, bool WIDE>
__global__ __launch_bounds__(kWave *kCnWaves) void flood_cn_kernel(FloodDev d, ST *msg, ST *scratch,
                                                                   const ST *lam, int stamp,
                                                                   int syndrome_only, int force) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row_groups = (d.M + kCnWaves - 1) / kCnWaves;
    const int slabs = d.Bp / kWave;
    // XCD-aware remap of the linear block id
    int L = blockIdx.x;
    int slab, rg;
    {
        const int per_xcd_slabs = slabs / 8; // slabs handled "8 at a time" (one per XCD)
        const int full = per_xcd_slabs * 8 * row_groups;
        if (L < full) {
            int xcd = L & 7, j = L >> 3;
            slab = (j / row_groups) * 8 + xcd;
            rg = j % row_groups;
        } else { // tail: slabs that do not fill a group of 8
            int j = L - full;
            slab = per_xcd_slabs * 8 + j / row_groups;
            rg = j % row_groups;
        }
    }
    const int row = rg * kCnWaves + wave;
    if (row >= d.M) return;
    const size_t b = (size_t)slab * kWave + lane;
    // (row extent requested before the frame flag is looked at: two independent round trips instead of a chain)
    const int ebeg = d.row_ptr[row];
    const int deg = d.row_ptr[row + 1] - ebeg;
    const bool active = force ? true : (d.done[b] == 0);
    if (!active) return; // lanes of finished (or padding) frames drop out; an all-done wave exits
    const bool so = syndrome_only != 0;
    constexpr bool kPaddable = !(VARIANT != LDPC_V_MINSUM && sizeof(typename Store<ST>::CT) == 8);  // f64 tanh (both numerics): O(d^2) anyway
    const bool wide_row = kPaddable && d.wide_rows && deg > 8 && deg <= 32 && deg != 18;
    if constexpr (WIDE) {
        if (!wide_row) return;
        if (deg <= 12) cn_row_padded<ST, VARIANT, 12>(d, msg, lam, ebeg, deg, b, stamp, so);
        else if (deg <= 16) cn_row_padded<ST, VARIANT, 16>(d, msg, lam, ebeg, deg, b, stamp, so);
        else if (deg <= 24) cn_row_padded<ST, VARIANT, 24>(d, msg, lam, ebeg, deg, b, stamp, so);
        else cn_row_padded<ST, VARIANT, 32>(d, msg, lam, ebeg, deg, b, stamp, so);
    } else {
        if (wide_row) return;
        switch (deg) {
            case 0: break;
            case 1:
                if constexpr (VARIANT == LDPC_V_TANH) cn_row_regs<ST, VARIANT, 1>(d, msg, lam, ebeg, b, stamp, so);
                else cn_row_generic<ST, VARIANT>(d, msg, scratch, lam, ebeg, deg, b, stamp, true); // rejected at ctx_create
                break;
            case 2: cn_row_regs<ST, VARIANT, 2>(d, msg, lam, ebeg, b, stamp, so); break;
            case 3: cn_row_regs<ST, VARIANT, 3>(d, msg, lam, ebeg, b, stamp, so); break;
            case 4: cn_row_regs<ST, VARIANT, 4>(d, msg, lam, ebeg, b, stamp, so); break;
            case 5: cn_row_regs<ST, VARIANT, 5>(d, msg, lam, ebeg, b, stamp, so); break;
            case 6: cn_row_regs<ST, VARIANT, 6>(d, msg, lam, ebeg, b, stamp, so); break;
            case 7: cn_row_regs<ST, VARIANT, 7>(d, msg, lam, ebeg, b, stamp, so); break;
            case 8: cn_row_regs<ST, VARIANT, 8>(d, msg, lam, ebeg, b, stamp, so); break;
            case 18: cn_row_regs<ST, VARIANT, 18>(d, msg, lam, ebeg, b, stamp, so); break;
            default: cn_row_generic<ST, VARIANT>(d, msg, scratch, lam, ebeg, deg, b, stamp, so); break;
        }
    }
}

// ------------------------------------------------------------------ VN update
// One column of weight DEG for 64 codewords: edge indices (scalar), then all DEG message loads, then the sum in
// the reference's order (Orig.hs:96: foldr => last row first).  A function per weight keeps the loads free of
// per-edge branches, so they are all in flight together.
template <typename ST, int DEG>
__device__ __forceinline__ typename Store<ST>::CT vn_sum(const FloodDev &d, const ST *__restrict__ msg, int qe, size_t b,
                                                         typename Store<ST>::CT acc) {
    using CT = typename Store<ST>::CT;
    int e[DEG];
    CT v[DEG];
#pragma unroll
    for (int j = 0; j < DEG; j++) e[j] = d.csc_edge[qe - 1 - j];
#pragma unroll
    for (int j = 0; j < DEG; j++) v[j] = Store<ST>::ld(msg + (size_t)e[j] * d.Bp + b);
#pragma unroll
    for (int j = 0; j < DEG; j++) acc = v[j] + acc;
    return acc;
}

// the same for any weight <= U: clustered index loads, then message loads predicated on the (wave-uniform) weight
template <typename ST, int U>
__device__ __forceinline__ typename Store<ST>::CT vn_sum_pred(const FloodDev &d, const ST *__restrict__ msg, int qe, int deg,
                                                              size_t b, typename Store<ST>::CT acc) {
    using CT = typename Store<ST>::CT;
    int e[U];
    CT v[U];
#pragma unroll
    for (int j = 0; j < U; j++) e[j] = (j < deg) ? d.csc_edge[qe - 1 - j] : 0;
#pragma unroll
    for (int j = 0; j < U; j++) v[j] = (j < deg) ? Store<ST>::ld(msg + (size_t)e[j] * d.Bp + b) : CT(0);
#pragma unroll
    for (int j = 0; j < U; j++)
        if (j < deg) acc = v[j] + acc;
    return acc;
}

// grid: (Bp/64) x ceil(N/4) blocks of 4 waves; wave = (column, 64 codewords).
template <typename ST>
__global__ __launch_bounds__(256) void flood_vn_kernel(FloodDev d, const ST *__restrict__ msg,
                                                       const ST *__restrict__ orig, ST *__restrict__ lam,
                                                       int n, int force) {
    using CT = typename Store<ST>::CT;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = blockIdx.y * 4 + wave;
    if (col >= d.N) return;
    const size_t b = (size_t)blockIdx.x * kWave + lane;
    if (!force) {
        if (d.done[b]) return;
        if (d.unsat[b] != n + 1) { // syndrome of hard(lam_n) was zero: Orig.hs:69, return lam
            if (col == 0) { d.done[b] = 1; d.iters[b] = n; d.conv[b] = 1; }
            return;
        }
    }
    CT acc = Store<ST>::ld(orig + (size_t)col * d.Bp + b);
    if constexpr (sizeof(ST) == 4) {
        if (d.saturate) acc = ldexpf(acc, -d.kexp[b]);   // min-sum f32: the frame has been rescaled by 2^-kexp (ldpc_math.h); exact
    }
    const int qb = d.col_ptr[col], qe = d.col_ptr[col + 1];
    const int deg = qe - qb;   // wave-uniform
    if (d.cm_order) {   // parity modes of the other registered decoders (uniform branch)
        if (deg > 0 && d.cm_order == 1) {   // arraylet, arraylet-min, arraylet-cm: lam' = orig + foldr1 (+) [ne' of the column, ascending row]
            CT sum = Store<ST>::ld(msg + (size_t)d.csc_edge[qe - 1] * d.Bp + b);   // (Fast/Arraylet.hs:105-109,185-186; CachedMult.hs:190-194,261-262)
            for (int q = qe - 2; q >= qb; q--) sum = Store<ST>::ld(msg + (size_t)d.csc_edge[q] * d.Bp + b) + sum;
            acc = acc + sum;
        } else if (d.cm_order == 3) {       // cuda-arraylet2 (cudabits/common.h:161-171): newLam[i] += ... over ascending block rows, from orig
            for (int q = qb; q < qe; q++) acc = acc + Store<ST>::ld(msg + (size_t)d.csc_edge[q] * d.Bp + b);
        } else if (deg > 0) {               // sparse, sparsemin: lam' = orig + sum (map snd column), a left fold from 0 over ascending rows
            CT sum = CT(0);                 // (Reference/Sparse.hs:112-114, Data/Sparse/Matrix.hs:35-36)
            for (int q = qb; q < qe; q++) sum = sum + Store<ST>::ld(msg + (size_t)d.csc_edge[q] * d.Bp + b);
            acc = acc + sum;
        }
    } else
    // Two ways of getting a column's message loads in flight together (a plain loop pays one memory round trip per
    // edge): a function per weight, or predicated loads after clustered index loads.  Measured on one box, jpl.4096,
    // 16 384 frames, whole flood path: f32 1 050 (predicated) vs 990 Mbit/s (per weight), fp16 storage 1 110 vs
    // 1 250 -- so the choice follows the storage type.
    if (deg > 16) {
        for (int q = qe - 1; q >= qb; q--) {   // Orig.hs:96: foldr => last row first
            int e = d.csc_edge[q];
            acc = Store<ST>::ld(msg + (size_t)e * d.Bp + b) + acc;
        }
    } else if (deg > 8) {
        acc = vn_sum_pred<ST, 16>(d, msg, qe, deg, b, acc);
    } else if constexpr (sizeof(ST) == 2) {
        switch (deg) {
            case 1: acc = vn_sum<ST, 1>(d, msg, qe, b, acc); break;
            case 2: acc = vn_sum<ST, 2>(d, msg, qe, b, acc); break;
            case 3: acc = vn_sum<ST, 3>(d, msg, qe, b, acc); break;
            case 4: acc = vn_sum<ST, 4>(d, msg, qe, b, acc); break;
            case 5: acc = vn_sum<ST, 5>(d, msg, qe, b, acc); break;
            case 6: acc = vn_sum<ST, 6>(d, msg, qe, b, acc); break;
            case 7: acc = vn_sum<ST, 7>(d, msg, qe, b, acc); break;
            case 8: acc = vn_sum<ST, 8>(d, msg, qe, b, acc); break;
            default: break;
        }
    } else {
        acc = vn_sum_pred<ST, 8>(d, msg, qe, deg, b, acc);
    }
    if constexpr (sizeof(ST) == 4) {
        if (d.saturate && fabsf(acc) > kLamBig) d.big[b] = 1;   // (uniform branch; benign race: every writer stores 1) -> flood_rescale_kernel
    }
    Store<ST>::st(lam + (size_t)col * d.Bp + b, acc);
}

// min-sum f32: after a variable-node pass, the frames in which some |lam| passed 2^60 are multiplied by 2^-40 -- lam and
// every message; the channel LLRs keep their values and enter later column sums through kexp (flood_vn_kernel).  One block per
// slab of 64 frames, lane = frame; a slab without such a frame costs one load per lane.
template <typename ST>
__global__ __launch_bounds__(1024) void flood_rescale_kernel(FloodDev d, ST *__restrict__ msg, ST *__restrict__ lam) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const size_t b = (size_t)blockIdx.x * kWave + lane;
    const bool hit = d.big[b] != 0;
    if (!__builtin_amdgcn_ballot_w64(hit)) return;       // (the same answer in every wave of the block)
    if (hit) {
        for (int e = wave; e < d.E; e += nw) msg[(size_t)e * d.Bp + b] *= (ST)kRescale;
        for (int c = wave; c < d.N; c += nw) lam[(size_t)c * d.Bp + b] *= (ST)kRescale;
    }
    __syncthreads();                                     // every wave has read the flags
    if (hit && wave == 0) { d.big[b] = 0; d.kexp[b] += kRescaleExp; }
}

// after the last turn: frames still open get the n = max_iters syndrome verdict (Orig.hs:69-70)
__global__ void flood_finalize_kernel(FloodDev d, int max_iters) {
    size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= (size_t)d.Bp || d.done[b]) return;
    d.done[b] = 1;
    d.iters[b] = max_iters;
    d.conv[b] = (d.unsat[b] != max_iters + 1) ? 1 : 0;
}

__global__ void flood_reset_kernel(FloodDev d, int batch) {
    size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= (size_t)d.Bp) return;
    d.unsat[b] = 0;
    if (d.big) { d.big[b] = 0; d.kexp[b] = 0; }
    d.iters[b] = 0;
    d.conv[b] = 0;
    d.done[b] = (b < (size_t)batch) ? 0 : 1; // padding lanes are born finished
}

// ------------------------------------------------------------------ layout changes
// in [batch][N] (frame-major) -> orig, lam [N][Bp]; 64x64 tiles through LDS.
template <typename IT, typename ST>
__global__ __launch_bounds__(256) void load_llr_kernel(const IT *__restrict__ in, ST *__restrict__ orig,
                                                       ST *__restrict__ lam, int batch, int N, int Bp) {
    __shared__ float tile[64][65];
    __shared__ double tiled[sizeof(IT) == 8 ? 64 : 1][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    for (int r = ty; r < 64; r += 4) {
        int b = b0 + r, n = n0 + tx;
        const bool ok = b < batch && n < N;
        if constexpr (sizeof(IT) == 8) tiled[r][tx] = ok ? in[(size_t)b * N + n] : 0.0;
        else if constexpr (sizeof(IT) == 2) tile[r][tx] = ok ? __half2float(in[(size_t)b * N + n]) : 0.f;
        else tile[r][tx] = ok ? in[(size_t)b * N + n] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        int n = n0 + r, b = b0 + tx;
        if (n < N && b < Bp) {
            typename Store<ST>::CT v;
            if constexpr (sizeof(IT) == 8) v = (typename Store<ST>::CT)tiled[tx][r]; else v = (typename Store<ST>::CT)tile[tx][r];
            Store<ST>::st(orig + (size_t)n * Bp + b, v);
            Store<ST>::st(lam + (size_t)n * Bp + b, v);
        }
    }
}

// result: frames that converged output hard(lam), the others hard(orig) (Orig.hs:59,69-70);
// bits [batch][N] bytes, optional final LLRs [batch][N] double.
template <typename ST>
__global__ __launch_bounds__(256) void store_bits_kernel(FloodDev d, const ST *__restrict__ orig,
                                                         const ST *__restrict__ lam, uint8_t *__restrict__ bits,
                                                         double *__restrict__ final_lam, int batch) {
    __shared__ float tile[64][65];
    __shared__ double tiled[sizeof(ST) == 8 ? 64 : 1][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    for (int r = ty; r < 64; r += 4) {
        int n = n0 + r, b = b0 + tx;
        typename Store<ST>::CT v = 0;
        if (n < d.N && b < d.Bp) v = Store<ST>::ld((d.conv[b] ? lam : orig) + (size_t)n * d.Bp + b);
        if constexpr (sizeof(ST) == 8) tiled[r][tx] = v; else tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        int b = b0 + r, n = n0 + tx;
        if (b < batch && n < d.N) {
            double v;
            if constexpr (sizeof(ST) == 8) v = tiled[tx][r]; else v = tile[tx][r];
            bits[(size_t)b * d.N + n] = v > 0.0 ? 1 : 0;
            if (final_lam) final_lam[(size_t)b * d.N + n] = (d.kexp && d.conv[b]) ? ldexp(v, d.kexp[b]) : v;   // (a rescaled min-sum frame: ldpc_math.h)
        }
    }
}

// trace: lam at the top of loop turn n for frames still open -> trace[b][n][:] (double)
template <typename ST>
__global__ __launch_bounds__(256) void trace_store_kernel(FloodDev d, const ST *__restrict__ lam,
                                                          double *__restrict__ trace, int n, int turns, int batch) {
    __shared__ double tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    for (int r = ty; r < 64; r += 4) {
        int col = n0 + r, b = b0 + tx;
        tile[r][tx] = (col < d.N && b < d.Bp) ? (double)Store<ST>::ld(lam + (size_t)col * d.Bp + b) : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        int b = b0 + r, col = n0 + tx;
        if (b < batch && col < d.N && !d.done[b]) trace[((size_t)b * turns + n) * d.N + col] = d.kexp ? ldexp(tile[tx][r], d.kexp[b]) : tile[tx][r];
    }
}

// generic 2-D transposing copy between frame-major double host images and batch-major device
// arrays (debug entry points only).
template <typename ST>
__global__ void upload_rows_kernel(const double *__restrict__ in, ST *__restrict__ out, int batch, int R, int Bp) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; // over R*Bp, b fastest
    if (i >= (size_t)R * Bp) return;
    int b = (int)(i % Bp);
    size_t r = i / Bp;
    typename Store<ST>::CT v = (b < batch) ? (typename Store<ST>::CT)in[(size_t)b * R + r] : 0;
    Store<ST>::st(out + i, v);
}
template <typename ST>
__global__ void download_rows_kernel(const ST *__restrict__ in, double *__restrict__ out, int batch, int R, int Bp) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)R * Bp) return;
    int b = (int)(i % Bp);
    size_t r = i / Bp;
    if (b < batch) out[(size_t)b * R + r] = (double)Store<ST>::ld(in + i);
}
__global__ void syndrome_flags_kernel(FloodDev d, uint8_t *out, int batch, int stamp) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch) out[b] = d.unsat[b] != stamp ? 1 : 0;
}

// ==================================================================== row-layered schedule (extension)
// BASELINE.json configs[4]: "layered min-sum + early termination" on codes whose frame does not fit on-chip (a
// DVB-S2-class n = 64 800 code keeps 253 KB of LLRs alone).  No counterpart in the reference; the specification is
// oracle_decode_layered (oracle/ldpc_oracle.c) and the arithmetic per row is the flooding rule's.
//
// Same batch-major layout as above (lane = codeword, scalar graph indices), but ONE persistent workgroup of LW waves
// owns a 64-frame slab for the whole decode: layers run one after the other, the rows of a layer (column-disjoint by
// construction: a block row of a single-circulant QC code) are dealt round-robin to the waves, a workgroup barrier
// separates the layers.  No launch per turn, no grid-wide synchronisation: slabs are independent.  Stopping rule of
// the specification, per frame: before the first sweep, syndrome(hard lam) == 0; after a sweep, "no check it saw was
// odd and no hard decision changed".  Finished frames freeze (their lanes stop storing); a workgroup leaves when its
// 64 frames are finished or out of sweeps.
// HBM traffic per frame and sweep: every edge reads and writes its lam cell and its message: 4E * sizeof(ST).
struct LayerDev {
    int n_layers;
    const int32_t *layer_ptr;   // [n_layers + 1] row ranges
};
struct LayerArgs {
    int max_iters, step_mode;
    double *trace;              // may be null: [batch][max_iters + 1][N], lam after sweep n at row n
    int batch;
};
// waves per slab: 16 when there are few slabs (a small batch must still fill the chip), 8 when there are many (more
// independent workgroups per CU hide more memory latency: 74 VGPRs -> 24 waves per CU = three 8-wave workgroups, but
// only one 16-wave workgroup)
constexpr int kLayerWavesMax = 16;

template <typename ST, int VARIANT, int DEG, bool SYNDROME_ONLY>
__device__ __forceinline__ void layer_row_regs(const FloodDev &d, ST *__restrict__ msg, ST *__restrict__ lam, int ebeg, size_t b,
                                               bool active, bool &odd, bool &flip) {
    using CT = typename Store<ST>::CT;
    int col[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) col[k] = d.col_idx[ebeg + k];
    CT l[DEG], t[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) l[k] = Store<ST>::ld(lam + (size_t)col[k] * d.Bp + b);
    if constexpr (!SYNDROME_ONLY) {
#pragma unroll
        for (int k = 0; k < DEG; k++) t[k] = Store<ST>::ld(msg + (size_t)(ebeg + k) * d.Bp + b);
    }
    bool par = false;
#pragma unroll
    for (int k = 0; k < DEG; k++) par ^= hard(l[k]);
    odd |= par && active;
    if constexpr (SYNDROME_ONLY) return;
    CT nm[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) { t[k] = l[k] - t[k]; nm[k] = t[k]; }
    cn_update<CT, VARIANT, DEG>(nm);
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        const CT nw = t[k] + nm[k];
        flip |= active && (hard(nw) != hard(l[k]));
        if (active) {
            Store<ST>::st(lam + (size_t)col[k] * d.Bp + b, nw);
            Store<ST>::st(msg + (size_t)(ebeg + k) * d.Bp + b, nm[k]);
        }
    }
}

// two rows of one layer (same weight, disjoint columns): every load of both rows is issued before either is used
template <typename ST, int VARIANT, int DEG>
__device__ __forceinline__ void layer_row_pair(const FloodDev &d, ST *__restrict__ msg, ST *__restrict__ lam, int ea, int eb, size_t b,
                                               bool active, bool &odd, bool &flip) {
    using CT = typename Store<ST>::CT;
    int ca[DEG], cb[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) { ca[k] = d.col_idx[ea + k]; cb[k] = d.col_idx[eb + k]; }
    CT la[DEG], lb[DEG], ta[DEG], tb[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) { la[k] = Store<ST>::ld(lam + (size_t)ca[k] * d.Bp + b); lb[k] = Store<ST>::ld(lam + (size_t)cb[k] * d.Bp + b); }
#pragma unroll
    for (int k = 0; k < DEG; k++) { ta[k] = Store<ST>::ld(msg + (size_t)(ea + k) * d.Bp + b); tb[k] = Store<ST>::ld(msg + (size_t)(eb + k) * d.Bp + b); }
    bool pa = false, pb = false;
#pragma unroll
    for (int k = 0; k < DEG; k++) { pa ^= hard(la[k]); pb ^= hard(lb[k]); }
    odd |= (pa || pb) && active;
    CT na[DEG], nb[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) { ta[k] = la[k] - ta[k]; na[k] = ta[k]; tb[k] = lb[k] - tb[k]; nb[k] = tb[k]; }
    cn_update<CT, VARIANT, DEG>(na);
    cn_update<CT, VARIANT, DEG>(nb);
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        const CT wa = ta[k] + na[k], wb = tb[k] + nb[k];
        flip |= active && ((hard(wa) != hard(la[k])) || (hard(wb) != hard(lb[k])));
        if (active) {
            Store<ST>::st(lam + (size_t)ca[k] * d.Bp + b, wa);
            Store<ST>::st(msg + (size_t)(ea + k) * d.Bp + b, na[k]);
            Store<ST>::st(lam + (size_t)cb[k] * d.Bp + b, wb);
            Store<ST>::st(msg + (size_t)(eb + k) * d.Bp + b, nb[k]);
        }
    }
}

template <typename ST, int VARIANT, int DMAX, bool SYNDROME_ONLY>
__device__ __forceinline__ void layer_row_padded(const FloodDev &d, ST *__restrict__ msg, ST *__restrict__ lam, int ebeg, int deg,
                                                 size_t b, bool active, bool &odd, bool &flip) {
    using CT = typename Store<ST>::CT;
    int col[DMAX];
#pragma unroll
    for (int k = 0; k < DMAX; k++) col[k] = (k < deg) ? d.col_idx[ebeg + k] : 0;
    CT l[DMAX], t[DMAX];
#pragma unroll
    for (int k = 0; k < DMAX; k++) l[k] = (k < deg) ? Store<ST>::ld(lam + (size_t)col[k] * d.Bp + b) : CT(0);
    if constexpr (!SYNDROME_ONLY) {
#pragma unroll
        for (int k = 0; k < DMAX; k++) t[k] = (k < deg) ? Store<ST>::ld(msg + (size_t)(ebeg + k) * d.Bp + b) : CT(0);
    }
    bool par = false;
#pragma unroll
    for (int k = 0; k < DMAX; k++) par ^= (k < deg) && hard(l[k]);
    odd |= par && active;
    if constexpr (SYNDROME_ONLY) return;
    CT nm[DMAX];
#pragma unroll
    for (int k = 0; k < DMAX; k++) { t[k] = (k < deg) ? l[k] - t[k] : CT(INFINITY); nm[k] = t[k]; }
    cn_update_padded<CT, VARIANT, DMAX>(nm, deg);
#pragma unroll
    for (int k = 0; k < DMAX; k++) {
        if (k < deg) {
            const CT nw = t[k] + nm[k];
            flip |= active && (hard(nw) != hard(l[k]));
            if (active) {
                Store<ST>::st(lam + (size_t)col[k] * d.Bp + b, nw);
                Store<ST>::st(msg + (size_t)(ebeg + k) * d.Bp + b, nm[k]);
            }
        }
    }
}

// DCLASS: the widest row the instance holds in registers (8, 20 or 32): the code's maximum row weight picks it, so a
// code of light rows is not compiled against the register count of 32-edge rows
template <typename ST, int VARIANT, int DCLASS, bool SO>
__device__ __forceinline__ void layer_row(const FloodDev &d, ST *msg, ST *lam, int row, size_t b, bool active, bool &odd, bool &flip) {
    const int ebeg = d.row_ptr[row];
    const int deg = d.row_ptr[row + 1] - ebeg;
    switch (deg) {
        case 0: return;
        case 1:
            if constexpr (VARIANT == LDPC_V_TANH) layer_row_regs<ST, VARIANT, 1, SO>(d, msg, lam, ebeg, b, active, odd, flip);
            else layer_row_regs<ST, LDPC_V_TANH, 1, true>(d, msg, lam, ebeg, b, active, odd, flip);   // min-sum on weight 1: rejected at creation
            return;
        case 2: layer_row_regs<ST, VARIANT, 2, SO>(d, msg, lam, ebeg, b, active, odd, flip); return;
        case 3: layer_row_regs<ST, VARIANT, 3, SO>(d, msg, lam, ebeg, b, active, odd, flip); return;
        case 4: layer_row_regs<ST, VARIANT, 4, SO>(d, msg, lam, ebeg, b, active, odd, flip); return;
        case 5: layer_row_regs<ST, VARIANT, 5, SO>(d, msg, lam, ebeg, b, active, odd, flip); return;
        case 6: layer_row_regs<ST, VARIANT, 6, SO>(d, msg, lam, ebeg, b, active, odd, flip); return;
        case 7: layer_row_regs<ST, VARIANT, 7, SO>(d, msg, lam, ebeg, b, active, odd, flip); return;
        case 8: layer_row_regs<ST, VARIANT, 8, SO>(d, msg, lam, ebeg, b, active, odd, flip); return;
        default: break;
    }
    if constexpr (DCLASS >= 20) {
        if (deg == 18) { layer_row_regs<ST, VARIANT, 18, SO>(d, msg, lam, ebeg, b, active, odd, flip); return; }
        if (deg <= 12) { layer_row_padded<ST, VARIANT, 12, SO>(d, msg, lam, ebeg, deg, b, active, odd, flip); return; }
        if (deg <= 16) { layer_row_padded<ST, VARIANT, 16, SO>(d, msg, lam, ebeg, deg, b, active, odd, flip); return; }
        if (deg <= 20) { layer_row_padded<ST, VARIANT, 20, SO>(d, msg, lam, ebeg, deg, b, active, odd, flip); return; }
    }
    if constexpr (DCLASS >= 32) {
        if (deg <= 24) { layer_row_padded<ST, VARIANT, 24, SO>(d, msg, lam, ebeg, deg, b, active, odd, flip); return; }
        if (deg <= 32) { layer_row_padded<ST, VARIANT, 32, SO>(d, msg, lam, ebeg, deg, b, active, odd, flip); return; }
    }
}

template <typename ST, int VARIANT, int DCLASS, int kLayerWaves>
__global__ __launch_bounds__(kWave *kLayerWaves) void layered_kernel(FloodDev d, LayerDev L, ST *msg, ST *lam, LayerArgs A) {
    using CT = typename Store<ST>::CT;
    __shared__ uint32_t flags[kLayerWaves][kWave];   // bit 0: odd, bit 1: flip -- per wave, per frame of the slab
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t b = (size_t)blockIdx.x * kWave + lane;
    bool done = d.done[b] != 0;        // padding frames are born finished (flood_reset_kernel)
    bool conv = false;
    int iters = 0;
    // OR of a per-lane flag pair over the waves of the workgroup (every wave holds the same 64 frames)
    auto combine = [&](bool odd, bool flip) -> uint32_t {
        flags[wave][lane] = (odd ? 1u : 0u) | (flip ? 2u : 0u);
        __syncthreads();
        uint32_t f = 0;
#pragma unroll
        for (int w = 0; w < kLayerWaves; w++) f |= flags[w][lane];
        __syncthreads();               // the flags are rewritten by the next combine
        return f;
    };
    auto store_trace = [&](int n, bool open) {
        if (!A.trace) return;
        if (open && b < (size_t)A.batch)
            for (int c = wave; c < d.N; c += kLayerWaves)
                A.trace[((size_t)b * (A.max_iters + 1) + n) * d.N + c] = (double)Store<ST>::ld(lam + (size_t)c * d.Bp + b);
        __syncthreads();   // (verification only) nobody starts the next sweep while lam is still being copied out
    };
    if (!A.step_mode) {
        store_trace(0, !done);
        bool odd = false, flip = false;
        for (int r = wave; r < d.M; r += kLayerWaves) layer_row<ST, VARIANT, DCLASS, true>(d, msg, lam, r, b, !done, odd, flip);
        const uint32_t f = combine(odd, false);
        if (!done && !(f & 1u)) { done = true; conv = true; iters = 0; }
    }
    for (int n = 1; n <= A.max_iters; n++) {
        if (__all(done)) break;        // same decision in every wave: `done` derives from the shared flags
        const bool open = !done;
        bool odd = false, flip = false;
        for (int l = 0; l < L.n_layers; l++) {
            const int r0 = L.layer_ptr[l], r1 = L.layer_ptr[l + 1];
            int r = r0 + wave;
            if constexpr (DCLASS <= 8) {
                // two rows of the layer at a time when they have the same weight (always, for a QC block row): both rows'
                // loads are in flight together
                for (; r + kLayerWaves < r1; r += 2 * kLayerWaves) {
                    const int ra = r, rb = r + kLayerWaves;
                    const int ea = d.row_ptr[ra], da = d.row_ptr[ra + 1] - ea, eb = d.row_ptr[rb], db = d.row_ptr[rb + 1] - eb;
                    if (da == db && da >= 2 && da <= 8) {
                        switch (da) {
                            case 2: layer_row_pair<ST, VARIANT, 2>(d, msg, lam, ea, eb, b, open, odd, flip); break;
                            case 3: layer_row_pair<ST, VARIANT, 3>(d, msg, lam, ea, eb, b, open, odd, flip); break;
                            case 4: layer_row_pair<ST, VARIANT, 4>(d, msg, lam, ea, eb, b, open, odd, flip); break;
                            case 5: layer_row_pair<ST, VARIANT, 5>(d, msg, lam, ea, eb, b, open, odd, flip); break;
                            case 6: layer_row_pair<ST, VARIANT, 6>(d, msg, lam, ea, eb, b, open, odd, flip); break;
                            case 7: layer_row_pair<ST, VARIANT, 7>(d, msg, lam, ea, eb, b, open, odd, flip); break;
                            default: layer_row_pair<ST, VARIANT, 8>(d, msg, lam, ea, eb, b, open, odd, flip); break;
                        }
                    } else {
                        layer_row<ST, VARIANT, DCLASS, false>(d, msg, lam, ra, b, open, odd, flip);
                        layer_row<ST, VARIANT, DCLASS, false>(d, msg, lam, rb, b, open, odd, flip);
                    }
                }
            }
            for (; r < r1; r += kLayerWaves) layer_row<ST, VARIANT, DCLASS, false>(d, msg, lam, r, b, open, odd, flip);
            __syncthreads();           // the next layer reads what this one wrote (workgroup-scope release/acquire)
        }
        const uint32_t f = combine(odd, flip);
        store_trace(n, open);
        bool stopped = open && !A.step_mode && f == 0u;
        if constexpr (kVetoesNonFinite<CT, VARIANT> && sizeof(ST) == 4) {
            if (__any(stopped)) {   // (the same in every wave) a frame whose LLRs left the float range is failed, not "converged" (ldpc_math.h)
                bool bad = false;
                if (stopped)
                    for (int c = wave; c < d.N; c += kLayerWaves) bad |= not_finite(Store<ST>::ld(lam + (size_t)c * d.Bp + b));
                if (combine(bad, false) & 1u) { if (stopped) done = true; stopped = false; }
            }
        }
        if (open) {
            iters = n;
            if (stopped) { done = true; conv = true; }
        }
        if (A.step_mode) break;
    }
    if (wave == 0 && b < (size_t)d.Bp) {
        d.iters[b] = conv ? iters : A.max_iters;
        d.conv[b] = conv ? 1 : 0;
        d.done[b] = 1;
        if (A.step_mode) d.unsat[b] = 0;
    }
}

// ------------------------------------------------------------------ host-side launch sequences
#define HIPCHK(x)                                                             \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) return set_error(LDPC_EHIP, "%s: %s", #x, hipGetErrorString(e_)); \
    } while (0)

template <typename ST, int VARIANT>
static void enqueue_turns(FloodState &s, hipStream_t st, int max_iters, int batch, double *d_trace) {
    FloodDev d = s.dev;
    ST *msg = (ST *)s.msg, *scr = (ST *)s.scratch, *lam = (ST *)s.lam, *orig = (ST *)s.orig;
    const int slabs = d.Bp / kWave;
    const int row_groups = (d.M + kCnWaves - 1) / kCnWaves;
    const dim3 cn_grid(slabs * row_groups), cn_block(kWave * kCnWaves);
    const dim3 vn_grid(slabs, (d.N + 3) / 4), vn_block(256);
    const dim3 tr_grid((d.N + 63) / 64, (d.Bp + 63) / 64);
    for (int n = 0; n < max_iters; n++) {
        if (d_trace) hipLaunchKernelGGL((trace_store_kernel<ST>), tr_grid, dim3(256), 0, st, d, lam, d_trace, n, max_iters + 1, batch);
        if (s.timer) s.timer->begin(st);
        hipLaunchKernelGGL((flood_cn_kernel<ST, VARIANT, false>), cn_grid, cn_block, 0, st, d, msg, scr, lam, n + 1, 0, 0);
        if (s.timer) s.timer->end(st);
        if (s.has_wide_rows) hipLaunchKernelGGL((flood_cn_kernel<ST, VARIANT, true>), cn_grid, cn_block, 0, st, d, msg, scr, lam, n + 1, 0, 0);
        hipLaunchKernelGGL((flood_vn_kernel<ST>), vn_grid, vn_block, 0, st, d, msg, orig, lam, n, 0);
        if constexpr (sizeof(ST) == 4) {
            if (d.saturate) hipLaunchKernelGGL((flood_rescale_kernel<ST>), dim3(slabs), dim3(1024), 0, st, d, msg, lam);
        }
    }
    if (d_trace) hipLaunchKernelGGL((trace_store_kernel<ST>), tr_grid, dim3(256), 0, st, d, lam, d_trace, max_iters, max_iters + 1, batch);
    hipLaunchKernelGGL((flood_cn_kernel<ST, VARIANT, false>), cn_grid, cn_block, 0, st, d, msg, scr, lam, max_iters + 1, 1, 0);
    if (s.has_wide_rows) hipLaunchKernelGGL((flood_cn_kernel<ST, VARIANT, true>), cn_grid, cn_block, 0, st, d, msg, scr, lam, max_iters + 1, 1, 0);
    hipLaunchKernelGGL(flood_finalize_kernel, dim3((d.Bp + 255) / 256), dim3(256), 0, st, d, max_iters);
}

// LDPC_FLOOD_GRAPH=0 keeps plain launches (A/B measurements)
static bool graph_wanted() {
    static const bool on = [] { const char *e = getenv("LDPC_FLOOD_GRAPH"); return !(e && !strcmp(e, "0")); }();
    return on;
}

template <typename ST, int VARIANT>
static int run_turns(FloodState &s, hipStream_t st, int max_iters, int batch, double *d_trace) {
    // Graph replay when nothing per-call is inside the loop: no trace buffer, no event timing, a capturable
    // stream (not the legacy default stream), at least one turn.
    const bool timing = s.timer && s.timer->enabled;
    if (graph_wanted() && !d_trace && !timing && st != nullptr && max_iters > 0) {
        if (!s.turn_graph || s.graph_iters != max_iters) {   // (the loop's grids cover the context's Bp frames, whatever `batch` is)
            flood_graph_release(s);
            hipGraph_t g = nullptr;
            hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                KernelTimer *keep = s.timer;
                s.timer = nullptr;                 // no event records inside the captured sequence
                enqueue_turns<ST, VARIANT>(s, st, max_iters, batch, nullptr);
                s.timer = keep;
                e = hipStreamEndCapture(st, &g);
            }
            if (e == hipSuccess && g) e = hipGraphInstantiate(&s.turn_graph, g, nullptr, nullptr, 0);
            if (g) (void)hipGraphDestroy(g);
            if (e != hipSuccess || !s.turn_graph) {   // capture not possible here: plain launches, as before
                (void)hipGetLastError();
                s.turn_graph = nullptr;
                enqueue_turns<ST, VARIANT>(s, st, max_iters, batch, nullptr);
                HIPCHK(hipGetLastError());
                return LDPC_OK;
            }
            s.graph_iters = max_iters;
        }
        HIPCHK(hipGraphLaunch(s.turn_graph, st));
        return LDPC_OK;
    }
    enqueue_turns<ST, VARIANT>(s, st, max_iters, batch, d_trace);
    HIPCHK(hipGetLastError());
    return LDPC_OK;
}

template <typename ST, int VARIANT>
static int decode_impl(FloodState &s, hipStream_t st, int max_iters, int batch, const void *d_llr,
                       int llr_fmt, uint8_t *d_bits, double *d_final, double *d_trace) {
    FloodDev d = s.dev;
    hipLaunchKernelGGL(flood_reset_kernel, dim3((d.Bp + 255) / 256), dim3(256), 0, st, d, batch);
    HIPCHK(hipMemsetAsync(s.msg, 0, (size_t)d.E * d.Bp * sizeof(ST), st)); // Orig.hs:64-65 orig_ne = 0
    const dim3 tgrid((d.N + 63) / 64, (d.Bp + 63) / 64);
    if (llr_fmt == LLR_F64)
        hipLaunchKernelGGL((load_llr_kernel<double, ST>), tgrid, dim3(256), 0, st, (const double *)d_llr, (ST *)s.orig, (ST *)s.lam, batch, d.N, d.Bp);
    else if (llr_fmt == LLR_F16)
        hipLaunchKernelGGL((load_llr_kernel<__half, ST>), tgrid, dim3(256), 0, st, (const __half *)d_llr, (ST *)s.orig, (ST *)s.lam, batch, d.N, d.Bp);
    else
        hipLaunchKernelGGL((load_llr_kernel<float, ST>), tgrid, dim3(256), 0, st, (const float *)d_llr, (ST *)s.orig, (ST *)s.lam, batch, d.N, d.Bp);
    int rc = run_turns<ST, VARIANT>(s, st, max_iters, batch, d_trace);
    if (rc != LDPC_OK) return rc;
    hipLaunchKernelGGL((store_bits_kernel<ST>), tgrid, dim3(256), 0, st, d, (const ST *)s.orig, (const ST *)s.lam, d_bits, d_final, batch);
    HIPCHK(hipGetLastError());
    return LDPC_OK;
}

template <typename ST, int VARIANT>
static int step_impl(FloodState &s, hipStream_t st, int batch, const double *d_orig, const double *d_lam,
                     const double *d_ne, double *d_ne_out, double *d_lam_out, uint8_t *d_syn) {
    FloodDev d = s.dev;
    ST *msg = (ST *)s.msg, *scr = (ST *)s.scratch, *lam = (ST *)s.lam, *orig = (ST *)s.orig;
    hipLaunchKernelGGL(flood_reset_kernel, dim3((d.Bp + 255) / 256), dim3(256), 0, st, d, batch);
    auto blocks = [&](size_t n) { return dim3((unsigned)((n + 255) / 256)); };
    hipLaunchKernelGGL((upload_rows_kernel<ST>), blocks((size_t)d.N * d.Bp), dim3(256), 0, st, d_orig, orig, batch, d.N, d.Bp);
    hipLaunchKernelGGL((upload_rows_kernel<ST>), blocks((size_t)d.N * d.Bp), dim3(256), 0, st, d_lam, lam, batch, d.N, d.Bp);
    hipLaunchKernelGGL((upload_rows_kernel<ST>), blocks((size_t)d.E * d.Bp), dim3(256), 0, st, d_ne, msg, batch, d.E, d.Bp);
    const int slabs = d.Bp / kWave;
    const int row_groups = (d.M + kCnWaves - 1) / kCnWaves;
    hipLaunchKernelGGL((flood_cn_kernel<ST, VARIANT, false>), dim3(slabs * row_groups), dim3(kWave * kCnWaves), 0, st, d, msg, scr, lam, 1, 0, 1);
    if (s.has_wide_rows) hipLaunchKernelGGL((flood_cn_kernel<ST, VARIANT, true>), dim3(slabs * row_groups), dim3(kWave * kCnWaves), 0, st, d, msg, scr, lam, 1, 0, 1);
    hipLaunchKernelGGL((flood_vn_kernel<ST>), dim3(slabs, (d.N + 3) / 4), dim3(256), 0, st, d, msg, orig, lam, 0, 1);
    hipLaunchKernelGGL((download_rows_kernel<ST>), blocks((size_t)d.E * d.Bp), dim3(256), 0, st, msg, d_ne_out, batch, d.E, d.Bp);
    hipLaunchKernelGGL((download_rows_kernel<ST>), blocks((size_t)d.N * d.Bp), dim3(256), 0, st, lam, d_lam_out, batch, d.N, d.Bp);
    hipLaunchKernelGGL(syndrome_flags_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, d, d_syn, batch, 1);
    HIPCHK(hipGetLastError());
    return LDPC_OK;
}

template <typename ST, int VARIANT>
static int layered_launch(FloodState &s, hipStream_t st, const LayerArgs &a) {
    FloodDev d = s.dev;
    LayerDev L{s.n_layers, s.d_layer_ptr};
    const int slabs = d.Bp / kWave;
    // LDPC_LAYER_WAVES=8|16 overrides (A/B measurements)
    static const int forced = [] { const char *e = getenv("LDPC_LAYER_WAVES"); return e ? atoi(e) : 0; }();
    const bool narrow = forced ? forced == 8 : slabs >= 512;
    const dim3 grid(slabs), block(kWave * (narrow ? 8 : kLayerWavesMax));
    if (s.timer && !a.step_mode) s.timer->begin(st);
#define LAUNCH_LAYERED(DC)                                                                                                                       \
    do {                                                                                                                                         \
        if (narrow) hipLaunchKernelGGL((layered_kernel<ST, VARIANT, DC, 8>), grid, block, 0, st, d, L, (ST *)s.msg, (ST *)s.lam, a);             \
        else hipLaunchKernelGGL((layered_kernel<ST, VARIANT, DC, kLayerWavesMax>), grid, block, 0, st, d, L, (ST *)s.msg, (ST *)s.lam, a);       \
    } while (0)
    if (s.max_row_deg <= 8) LAUNCH_LAYERED(8);
    else if (s.max_row_deg <= 20) LAUNCH_LAYERED(20);
    else LAUNCH_LAYERED(32);
#undef LAUNCH_LAYERED
    if (s.timer && !a.step_mode) s.timer->end(st);
    HIPCHK(hipGetLastError());
    return LDPC_OK;
}

template <typename ST, int VARIANT>
static int layered_decode_impl(FloodState &s, hipStream_t st, int max_iters, int batch, const void *d_llr, int llr_fmt, uint8_t *d_bits,
                               double *d_final, double *d_trace) {
    FloodDev d = s.dev;
    hipLaunchKernelGGL(flood_reset_kernel, dim3((d.Bp + 255) / 256), dim3(256), 0, st, d, batch);
    HIPCHK(hipMemsetAsync(s.msg, 0, (size_t)d.E * d.Bp * sizeof(ST), st));
    const dim3 tgrid((d.N + 63) / 64, (d.Bp + 63) / 64);
    if (llr_fmt == LLR_F64)
        hipLaunchKernelGGL((load_llr_kernel<double, ST>), tgrid, dim3(256), 0, st, (const double *)d_llr, (ST *)s.orig, (ST *)s.lam, batch, d.N, d.Bp);
    else if (llr_fmt == LLR_F16)
        hipLaunchKernelGGL((load_llr_kernel<__half, ST>), tgrid, dim3(256), 0, st, (const __half *)d_llr, (ST *)s.orig, (ST *)s.lam, batch, d.N, d.Bp);
    else
        hipLaunchKernelGGL((load_llr_kernel<float, ST>), tgrid, dim3(256), 0, st, (const float *)d_llr, (ST *)s.orig, (ST *)s.lam, batch, d.N, d.Bp);
    LayerArgs a{max_iters, 0, d_trace, batch};
    int rc = layered_launch<ST, VARIANT>(s, st, a);
    if (rc != LDPC_OK) return rc;
    hipLaunchKernelGGL((store_bits_kernel<ST>), tgrid, dim3(256), 0, st, d, (const ST *)s.orig, (const ST *)s.lam, d_bits, d_final, batch);
    HIPCHK(hipGetLastError());
    return LDPC_OK;
}

// teacher-forced SWEEP: from (lam, msg) one full sweep over the layers -> (msg', lam'); d_syn = syndrome of hard(lam) is zero
template <typename ST, int VARIANT>
static int layered_step_impl(FloodState &s, hipStream_t st, int batch, const double *d_orig, const double *d_lam, const double *d_ne,
                             double *d_ne_out, double *d_lam_out, uint8_t *d_syn) {
    FloodDev d = s.dev;
    ST *msg = (ST *)s.msg, *scr = (ST *)s.scratch, *lam = (ST *)s.lam, *orig = (ST *)s.orig;
    hipLaunchKernelGGL(flood_reset_kernel, dim3((d.Bp + 255) / 256), dim3(256), 0, st, d, batch);
    auto blocks = [&](size_t n) { return dim3((unsigned)((n + 255) / 256)); };
    hipLaunchKernelGGL((upload_rows_kernel<ST>), blocks((size_t)d.N * d.Bp), dim3(256), 0, st, d_orig, orig, batch, d.N, d.Bp);
    hipLaunchKernelGGL((upload_rows_kernel<ST>), blocks((size_t)d.N * d.Bp), dim3(256), 0, st, d_lam, lam, batch, d.N, d.Bp);
    hipLaunchKernelGGL((upload_rows_kernel<ST>), blocks((size_t)d.E * d.Bp), dim3(256), 0, st, d_ne, msg, batch, d.E, d.Bp);
    // syndrome of the given lam: the flooding check-node kernel in syndrome-only mode (stamp 1)
    const int slabs = d.Bp / kWave, row_groups = (d.M + kCnWaves - 1) / kCnWaves;
    hipLaunchKernelGGL((flood_cn_kernel<ST, VARIANT, false>), dim3(slabs * row_groups), dim3(kWave * kCnWaves), 0, st, d, msg, scr, lam, 1, 1, 1);
    if (s.has_wide_rows) hipLaunchKernelGGL((flood_cn_kernel<ST, VARIANT, true>), dim3(slabs * row_groups), dim3(kWave * kCnWaves), 0, st, d, msg, scr, lam, 1, 1, 1);
    hipLaunchKernelGGL(syndrome_flags_kernel, dim3((batch + 255) / 256), dim3(256), 0, st, d, d_syn, batch, 1);
    LayerArgs a{1, 1, nullptr, batch};
    int rc = layered_launch<ST, VARIANT>(s, st, a);
    if (rc != LDPC_OK) return rc;
    hipLaunchKernelGGL((download_rows_kernel<ST>), blocks((size_t)d.E * d.Bp), dim3(256), 0, st, msg, d_ne_out, batch, d.E, d.Bp);
    hipLaunchKernelGGL((download_rows_kernel<ST>), blocks((size_t)d.N * d.Bp), dim3(256), 0, st, lam, d_lam_out, batch, d.N, d.Bp);
    HIPCHK(hipGetLastError());
    return LDPC_OK;
}

#define DISPATCH(FN, ...)                                                                   \
    if (s.variant == LDPC_TANH_CUDA32) {                                                    \
        if (s.dtype == LDPC_F32) return FN<float, LDPC_V_TANH_CUDA32>(__VA_ARGS__);         \
        return set_error(LDPC_EUNSUPPORTED, "the cuda-arraylet2 numerics exist in f32 only"); \
    }                                                                                       \
    if (s.variant == LDPC_TANH_CM) {                                                        \
        if (s.dtype == LDPC_F64) return FN<double, LDPC_V_TANH_CM>(__VA_ARGS__);            \
        return set_error(LDPC_EUNSUPPORTED, "the arraylet-cm numerics exist in f64 only");  \
    }                                                                                       \
    switch (s.dtype) {                                                                      \
        case LDPC_F32: return s.variant == LDPC_MINSUM ? FN<float, LDPC_V_MINSUM>(__VA_ARGS__) : FN<float, LDPC_V_TANH>(__VA_ARGS__);   \
        case LDPC_F64: return s.variant == LDPC_MINSUM ? FN<double, LDPC_V_MINSUM>(__VA_ARGS__) : FN<double, LDPC_V_TANH>(__VA_ARGS__); \
        case LDPC_F16: return s.variant == LDPC_MINSUM ? FN<__half, LDPC_V_MINSUM>(__VA_ARGS__) : FN<__half, LDPC_V_TANH>(__VA_ARGS__); \
        default: return set_error(LDPC_EINVAL, "bad dtype %d", s.dtype);                    \
    }

#define DISPATCH_NO_F16(FN, ...)                                                            \
    switch (s.dtype) {                                                                      \
        case LDPC_F32: return s.variant == LDPC_MINSUM ? FN<float, LDPC_V_MINSUM>(__VA_ARGS__) : FN<float, LDPC_V_TANH>(__VA_ARGS__);   \
        case LDPC_F64: return s.variant == LDPC_MINSUM ? FN<double, LDPC_V_MINSUM>(__VA_ARGS__) : FN<double, LDPC_V_TANH>(__VA_ARGS__); \
        default: return set_error(LDPC_EUNSUPPORTED, "the layered schedule exists for f32 and f64");                                    \
    }

int flood_decode(FloodState &s, hipStream_t st, int max_iters, int batch, const void *d_llr, int llr_fmt,
                 uint8_t *d_bits, double *d_final, double *d_trace) {
    if (s.layered) { DISPATCH_NO_F16(layered_decode_impl, s, st, max_iters, batch, d_llr, llr_fmt, d_bits, d_final, d_trace) }
    DISPATCH(decode_impl, s, st, max_iters, batch, d_llr, llr_fmt, d_bits, d_final, d_trace)
}
int flood_step(FloodState &s, hipStream_t st, int batch, const double *d_orig, const double *d_lam,
               const double *d_ne, double *d_ne_out, double *d_lam_out, uint8_t *d_syn) {
    if (s.layered) { DISPATCH_NO_F16(layered_step_impl, s, st, batch, d_orig, d_lam, d_ne, d_ne_out, d_lam_out, d_syn) }
    DISPATCH(step_impl, s, st, batch, d_orig, d_lam, d_ne, d_ne_out, d_lam_out, d_syn)
}

void flood_graph_release(FloodState &s) {
    if (s.turn_graph) { (void)hipGraphExecDestroy(s.turn_graph); s.turn_graph = nullptr; }
    s.graph_iters = -1;
}

size_t flood_elem_size(int dtype) { return dtype == LDPC_F64 ? 8 : (dtype == LDPC_F16 ? 2 : 4); }

}  // namespace ldpc
