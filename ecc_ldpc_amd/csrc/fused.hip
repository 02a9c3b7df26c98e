// fused.hip -- whole-decode-in-one-launch min-sum flooding BP for quasi-cyclic codes (gfx950).
//
// One launch decodes a batch; a frame's entire BP state stays on-chip for all iterations:
//   * lam (a-posteriori LLRs, N values)  -> LDS, 44*V*4 bytes per wave (V = max(sz,64))
//   * check->variable messages           -> VGPRs, compressed per check row to the record
//       {m1s, m2s, idx, signs}  (= the reference's MinSum2 / `omit` two-min semigroup,
//       src/ECC/Code/LDPC/Utils.hs:133-144; leave-one-out min is m2 at the argmin, m1 elsewhere)
//   * channel LLRs (orig)                -> VGPRs
// Mapping: a frame is owned by a workgroup of WPF waves, one circulant row per lane and block row:
//   sz = 128 : WPF = 2 (128 threads, thread t owns row t of every circulant), 7 frames = 14 waves per CU
//   sz = 64  : WPF = 1, one frame per wave;   sz = 32 : one wave holds 2 frames (lane -> frame l&1, row l>>1)
// With WPF = 1 waves never communicate (no barrier, no atomic, no shuffle): the only ordering used is
// a wave's in-order DS queue.  With WPF = 2 the two waves meet at s_barrier between the ordered steps
// of phase B (block rows must be added in descending order and share columns) and to OR their syndromes.
//
// Loop turn n (src/ECC/Code/LDPC/Reference/Min.hs:63-67 == Orig.hs:67-71):
//   phase A  for every row the lane owns: gather lam[col] from LDS (consecutive lanes hit
//            consecutive dwords: conflict-free), row parity of hard(lam) = the syndrome
//            (Min.hs:69-72), rebuild ne from the record, t = lam - ne, reduce to the new record
//            (Min.hs:75-86).  Syndrome zero -> frame done, output hard(lam) (Min.hs:64).
//   phase B  lam <- orig, then lam[col] += ne'[m,col] block-row by block-row in DESCENDING row
//            order = the reference's foldr (+) orig (col of ne') (Min.hs:100-103).  Inside a
//            block-row every lane hits a distinct column; successive block-rows are ordered by
//            the wave's in-order DS queue.
// After max_iters turns one more syndrome pass decides between hard(lam) and hard(orig) (Min.hs:64-65).
//
// Graph description: `tab` holds, block-row-major, one dword per non-empty circulant:
//   lo 16 bits = rotation * cpw * sizeof(CT)   (byte rotation inside a block column)
//   hi 16 bits = bc * V * sizeof(CT)           (byte base of the block column in LDS)
// The degree sequence of the block rows is a compile-time Plan (registers must be named statically).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <utility>
#include <vector>

#include "fused.h"
#include "ldpc_math.h"
#include "fused_common.h"
#include "jit.h"

namespace ldpc {

// rebuild message k of a row from its record.  F bit (D-1-k) = sign of ne_k.
template <int D, int K>
__device__ __forceinline__ float rec_msg(float m1s, float m2s, uint32_t sgi, uint32_t idx) {
    float mag = (idx == (uint32_t)K) ? m2s : m1s;
    uint32_t sgn = sgi << (31 - (D - 1 - K));
    return __uint_as_float(bfi(0x7fffffffu, __float_as_uint(mag), sgn));
}
template <int D, int K>
__device__ __forceinline__ double rec_msg(double m1s, double m2s, uint32_t sgi, uint32_t idx) {
    double mag = (idx == (uint32_t)K) ? m2s : m1s;
    return ((sgi >> (D - 1 - K)) & 1u) ? -mag : mag;
}

// phase A for the RPL rows (r0 + 64h) a lane owns in one block row of degree D: returns whether any
// of them has odd parity of hard(lam) (a non-zero syndrome bit); updates the records.
template <typename CT, int D, int RPL, int HSTEP, bool SYNDROME_ONLY>
__device__ __forceinline__ bool rows_phase_a(const char *lds, ctab_t tabrow, uint32_t p4, uint32_t vmask,
                                             CT *m1s, CT *m2s, uint32_t *sgi) {
    asm volatile("" : "+v"(p4));
    if constexpr (SYNDROME_ONLY) LDPC_COLD_PATH();
    CT l[RPL][D];
#pragma unroll
    for (int k = 0; k < D; k++) {
        uint32_t ent = tabrow[k];  // wave-uniform kernel-argument load -> SGPR
        uint32_t a = ((p4 + (ent & 0xffffu)) & vmask) | (ent >> 16);
#pragma unroll
        for (int h = 0; h < RPL; h++) {
            // row r0 + h*HSTEP/ES sits half a circulant further round: for RPL == 2 that is a0 ^ HSTEP bytes
            uint32_t ah = (RPL == 2) ? (a ^ (uint32_t)(h * HSTEP)) : (h == 0 ? a : ((((p4 + HSTEP * h) + (ent & 0xffffu)) & vmask) | (ent >> 16)));
            l[h][k] = lds_ld<CT>(lds, ah);
        }
    }
    bool any = false;
#pragma unroll
    for (int h = 0; h < RPL; h++) {
        bool par = false;
#pragma unroll
        for (int k = 0; k < D; k++) par ^= (l[h][k] > CT(0));
        any |= par;
    }
    if constexpr (SYNDROME_ONLY) return any;
#pragma unroll
    for (int h = 0; h < RPL; h++) {
        const uint32_t idx = sgi[h] >> 24;
        CT m1 = CT(INFINITY), m2 = CT(INFINITY);
        uint32_t nidx = 0, sg = 0;
        if constexpr (sizeof(CT) == 4) {
            uint32_t X = 0;
            static_for<0, D>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                float ne = rec_msg<D, k>(m1s[h], m2s[h], sgi[h], idx);
                float t = l[h][k] - ne;
                uint32_t tb = __float_as_uint(t);
                X ^= tb;
                sg = __builtin_amdgcn_alignbit(sg, tb, 31);  // sg = (sg << 1) | sign(t)
                float a = fabsf(t);
                nidx = (a < m1) ? (uint32_t)k : nidx;
                m2 = __builtin_amdgcn_fmed3f(m1, m2, a);
                m1 = fminf(m1, a);
            });
            // sign of ne'_k = sign-parity of all t, xor sign(t_k), xor (D odd)   ("min-sum signs")
            uint32_t Xc = X ^ ((D & 1) ? 0x80000000u : 0u);
            uint32_t F = sg ^ (uint32_t)((int32_t)Xc >> 31);
            asm("" : "+v"(nidx));  // keep k small: (k << 24) constants would otherwise occupy 18 VGPRs
            sgi[h] = (nidx << 24) | (F & ((1u << D) - 1u));
        } else {
            unsigned sx = 0;
            static_for<0, D>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                double ne = rec_msg<D, k>(m1s[h], m2s[h], sgi[h], idx);
                double t = l[h][k] - ne;
                unsigned s = (unsigned)(__double_as_longlong(t) >> 63) & 1u;
                sx ^= s;
                sg = (sg << 1) | s;
                double a = fabs(t);
                if (a < m1) { m2 = m1; m1 = a; nidx = k; }
                else if (a < m2) { m2 = a; }
            });
            uint32_t flip = (sx ^ (D & 1)) ? ((1u << D) - 1u) : 0u;
            sgi[h] = (nidx << 24) | ((sg ^ flip) & ((1u << D) - 1u));
        }
        m1s[h] = CT(0.75) * m1;  // |(-3/4) * acc|: the one rounding of Min.hs:78
        m2s[h] = CT(0.75) * m2;
    }
    return any;
}

// phase B for the same rows: lam[col_k] += ne'_k.  All D x RPL targets of one block row are distinct
// columns, so the row is done as: read every target, add, write every target back (three batches; one
// ds_read and one ds_write per edge).  NOT ds_add_f32: LDS float atomics serialise to ~1 lane/clk on
// gfx950 (tools/microbench_lds.hip: 75 clk per wave-instruction vs 11 for read+add+write).
template <typename CT, int D, int RPL, int HSTEP>
__device__ __forceinline__ void rows_phase_b(char *lds, ctab_t tabrow, uint32_t p4, uint32_t vmask,
                                             const CT *m1s, const CT *m2s, const uint32_t *sgi) {
    asm volatile("" : "+v"(p4));
    uint32_t idx[RPL];
#pragma unroll
    for (int h = 0; h < RPL; h++) idx[h] = sgi[h] >> 24;
    CT cur[RPL][D];
    uint32_t adr[D];
#pragma unroll
    for (int k = 0; k < D; k++) {
        uint32_t ent = tabrow[k];
        adr[k] = ((p4 + (ent & 0xffffu)) & vmask) | (ent >> 16);
#pragma unroll
        for (int h = 0; h < RPL; h++) {
            uint32_t ah = (RPL == 2) ? (adr[k] ^ (uint32_t)(h * HSTEP)) : (h == 0 ? adr[k] : ((((p4 + HSTEP * h) + (ent & 0xffffu)) & vmask) | (ent >> 16)));
            cur[h][k] = lds_ld<CT>(lds, ah);
        }
    }
    static_for<0, D>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
#pragma unroll
        for (int h = 0; h < RPL; h++) cur[h][k] = rec_msg<D, k>(m1s[h], m2s[h], sgi[h], idx[h]) + cur[h][k];
    });
#pragma unroll
    for (int k = 0; k < D; k++) {
        uint32_t ent = tabrow[k];
#pragma unroll
        for (int h = 0; h < RPL; h++) {
            uint32_t ah = (RPL == 2) ? (adr[k] ^ (uint32_t)(h * HSTEP)) : (h == 0 ? adr[k] : ((((p4 + HSTEP * h) + (ent & 0xffffu)) & vmask) | (ent >> 16)));
            lds_st<CT>(lds, ah, cur[h][k]);
        }
    }
}

template <typename CT, class Plan, int SZ>
__global__ __launch_bounds__((FusedCfg<CT, Plan, SZ>::THREADS), (FusedCfg<CT, Plan, SZ>::WAVES_PER_EU)) void fused_decode_kernel(FusedArgs A) {
    using Cfg = FusedCfg<CT, Plan, SZ>;
    constexpr int RPL = Cfg::RPL, CPW = Cfg::CPW, V = Cfg::V, N = Cfg::N, WPF = Cfg::WPF, HSTEP = Cfg::HSTEP;
    constexpr int RSTEP = Cfg::THREADS;  // row distance between a lane's rows
    constexpr uint32_t ES = sizeof(CT);
    constexpr uint32_t vmask = V * ES - 1;
    __shared__ __attribute__((aligned(16))) char lds[Cfg::LDS_BYTES];

    const uint32_t lane = threadIdx.x;  // position inside the frame's workgroup (0 .. 64*WPF-1)
    const uint32_t sub = lane % CPW;  // frame inside the wave
    const uint32_t r0 = lane / CPW;   // row / column the lane owns inside a block (h = 0)
    const long long frame = (long long)blockIdx.x * CPW + sub;
    const bool valid = frame < A.batch;
    const uint32_t p4 = lane * ES;    // byte position of the lane inside a block column (h = 0)
    const size_t fN = (size_t)(valid ? frame : 0) * N;

    // ---- channel LLRs -> registers (orig) ; records <- 0 (Min.hs:59-60 orig_ne = 0)
    CT orig[Cfg::NORIG];  // lanes of padding frames read frame 0 (never written back)
#pragma unroll
    for (int i = 0; i < Cfg::NORIG; i++)
        orig[i] = maybe_round_f16<CT>(load_llr<CT>(A.llr, fN + r0 + (i / RPL) * SZ + RSTEP * (i % RPL), A.llr_fmt), A.llr_round16);
    CT m1s[Cfg::NREC], m2s[Cfg::NREC];
    uint32_t sgi[Cfg::NREC];
#pragma unroll
    for (int i = 0; i < Cfg::NREC; i++) { m1s[i] = CT(0); m2s[i] = CT(0); sgi[i] = 0; }

    if (A.step_mode && valid) {  // teacher-forced state: records given
        static_for<0, Plan::NBR>([&](auto brc) {
            constexpr int br = decltype(brc)::value;
#pragma unroll
            for (int h = 0; h < RPL; h++) {
                size_t ri = (size_t)frame * Cfg::M + br * SZ + r0 + RSTEP * h;
                m1s[br * RPL + h] = reinterpret_cast<const CT *>(A.st_m1)[ri];
                m2s[br * RPL + h] = reinterpret_cast<const CT *>(A.st_m2)[ri];
                sgi[br * RPL + h] = A.st_sg[ri];
            }
        });
    }
    // ---- lam <- orig (or the given lam in step mode) in LDS
    static_for<0, Plan::NBC>([&](auto bcc) {
        constexpr int bc = decltype(bcc)::value;
#pragma unroll
        for (int h = 0; h < RPL; h++) {
            CT v = orig[bc * RPL + h];
            if (A.step_mode && valid) v = (CT)A.st_lam[fN + bc * SZ + r0 + RSTEP * h];
            lds_st<CT>(lds, (p4 + HSTEP * h) | (bc * V * ES), v);
        }
    });

    if constexpr (WPF > 1) __syncthreads();  // the other wave's half of lam must be in LDS before turn 0 gathers it

    // lanes that belong to the same frame as this lane
    unsigned long long fmask = ~0ull;
    if constexpr (CPW > 1) {
        unsigned long long m = 0;
#pragma unroll
        for (int i = 0; i < 64; i += CPW) m |= 1ull << i;
        fmask = m << sub;
    }

    bool active = valid;  // frame still iterating
    bool converged = false;
    int n_done = 0;
    const int turns = A.step_mode ? 1 : A.max_iters;

    for (int n = 0;; n++) {
        if (!__any(active)) break;
        if (A.trace && active) {  // lam at the top of loop turn n
            LDPC_COLD_PATH();
            static_for<0, Plan::NBC>([&](auto bcc) {
                constexpr int bc = decltype(bcc)::value;
#pragma unroll
                for (int h = 0; h < RPL; h++)
                    A.trace[((size_t)frame * (A.max_iters + 1) + n) * N + bc * SZ + r0 + RSTEP * h] =
                        (double)lds_ld<CT>(lds, (p4 + HSTEP * h) | (bc * V * ES));
            });
        }
        const bool last = (n >= turns);
        ctab_t tabA = (ctab_t)A.tab + opaque_uniform_zero();
        // ---- phase A: syndrome + new records (records untouched on the last, syndrome-only turn)
        bool unsat = false;
        if (active) {
            if (!last) {
                static_for<0, Plan::NBR>([&](auto brc) {
                    constexpr int br = decltype(brc)::value;
                    constexpr int D = Plan::deg(br);
                    unsat |= rows_phase_a<CT, D, RPL, HSTEP, false>(lds, tabA + Plan::ebeg(br), p4, vmask, &m1s[br * RPL], &m2s[br * RPL], &sgi[br * RPL]);
                });
            } else {
                static_for<0, Plan::NBR>([&](auto brc) {
                    constexpr int br = decltype(brc)::value;
                    constexpr int D = Plan::deg(br);
                    unsat |= rows_phase_a<CT, D, RPL, HSTEP, true>(lds, tabA + Plan::ebeg(br), p4, vmask, (CT *)nullptr, (CT *)nullptr, (uint32_t *)nullptr);
                });
            }
        }
        const unsigned long long ub = __ballot(unsat);  // inactive lanes vote 0
        bool frame_unsat = (ub & fmask) != 0ull;
        if constexpr (WPF > 1) {  // OR over the frame's waves; the barrier also fences phase A reads from phase B writes
            volatile uint32_t *flags = reinterpret_cast<volatile uint32_t *>(lds + Cfg::LAM_BYTES);
            if ((lane & 63) == 0) flags[lane >> 6] = frame_unsat ? 1u : 0u;
            __syncthreads();
            frame_unsat = (flags[0] | flags[1]) != 0u;
        }
        if (A.step_mode) {
            if (valid && r0 == 0) A.st_syn[frame] = frame_unsat ? 0 : 1;
        } else if (active && !frame_unsat) {  // Min.hs:64: syndrome zero -> return lam
            converged = true; active = false; n_done = n;
        }
        if (last) {                            // Min.hs:65: n >= maxIterations -> return orig_lam
            if (active) { active = false; n_done = n; }
            break;
        }
        // ---- phase B: lam <- orig ; lam[col] += ne' in descending row order (Min.hs:100-103)
        if (active) {
            ctab_t tabB = (ctab_t)A.tab + opaque_uniform_zero();
            static_for<0, Plan::NBC>([&](auto bcc) {
                constexpr int bc = decltype(bcc)::value;
#pragma unroll
                for (int h = 0; h < RPL; h++) lds_st<CT>(lds, (p4 + HSTEP * h) | (bc * V * ES), orig[bc * RPL + h]);
            });
            if constexpr (WPF > 1) __syncthreads();
            static_rfor<0, Plan::NBR>([&](auto brc) {
                constexpr int br = decltype(brc)::value;
                constexpr int D = Plan::deg(br);
                rows_phase_b<CT, D, RPL, HSTEP>(lds, tabB + Plan::ebeg(br), p4, vmask, &m1s[br * RPL], &m2s[br * RPL], &sgi[br * RPL]);
                if constexpr (WPF > 1) __syncthreads();  // block rows share columns: next one only after this one landed
            });
        }
        if (A.step_mode) break;
    }

    if (!valid) return;
    if (A.step_mode) {  // state out: lam' and the expanded messages ne' (CSR edge order)
        static_for<0, Plan::NBC>([&](auto bcc) {
            constexpr int bc = decltype(bcc)::value;
#pragma unroll
            for (int h = 0; h < RPL; h++)
                A.final_lam[fN + bc * SZ + r0 + RSTEP * h] = (double)lds_ld<CT>(lds, (p4 + HSTEP * h) | (bc * V * ES));
        });
        static_for<0, Plan::NBR>([&](auto brc) {
            constexpr int br = decltype(brc)::value;
            constexpr int D = Plan::deg(br);
#pragma unroll
            for (int h = 0; h < RPL; h++) {
                const uint32_t idx = sgi[br * RPL + h] >> 24;
                const size_t e0 = (size_t)frame * Plan::NEDGE * SZ + (size_t)SZ * Plan::ebeg(br) + (size_t)D * (r0 + RSTEP * h);
                static_for<0, D>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    A.st_ne_out[e0 + k] = (double)rec_msg<D, k>(m1s[br * RPL + h], m2s[br * RPL + h], sgi[br * RPL + h], idx);
                });
            }
        });
        return;
    }
    // ---- result: hard(lam) for a converged frame, hard(orig) otherwise (Min.hs:55,64-65)
    static_for<0, Plan::NBC>([&](auto bcc) {
        constexpr int bc = decltype(bcc)::value;
#pragma unroll
        for (int h = 0; h < RPL; h++) {
            CT v = converged ? lds_ld<CT>(lds, (p4 + HSTEP * h) | (bc * V * ES)) : orig[bc * RPL + h];
            size_t gi = fN + bc * SZ + r0 + RSTEP * h;
            A.bits[gi] = v > CT(0) ? 1 : 0;
            if (A.final_lam) A.final_lam[gi] = (double)v;
        }
    });
    if (r0 == 0) {
        if (A.iters) A.iters[frame] = n_done;
        if (A.conv) A.conv[frame] = converged ? 1 : 0;
    }
}

// ------------------------------------------------------------------ host side
struct FusedState {
    int variant = 0, dtype = 0, max_batch = 0, sz = 0, M = 0, N = 0, E = 0;  // dtype: the COMPUTE type (f32/f64)
    int round16 = 0;          // LDPC_F16 context: fp16 channel LLRs, f32 state (nothing else of a fused decode lives in HBM)
    CsrState *csr = nullptr;  // set when the code has no QC plan: generic on-chip kernel (fused_csr.hip)
    JitKernel *jit = nullptr; // run-time specialised split kernel (jit.cc): any single-circulant QC code without a built-in instance
    int static_id = 0;    // compiled-in rotation table matching this code (0 = none: table-driven kernel)
    bool use_split = false;  // four waves per frame, block rows split between wave pairs (fused_split.hip)
    bool use_msg = true;  // per-edge-message kernel (fused_msg.hip) vs compressed-record kernel (this file)
    KernelTimer *timer = nullptr;
    LaunchInfo info;
    uint32_t *d_tab = nullptr;
    std::vector<int32_t> row_ptr;  // host copy for the step-mode record conversion
};

static bool plan_matches_ar4ja45(const ldpc_code &c) {
    if (c.sz <= 0 || c.block_rows != PlanAR4JA45::NBR || c.block_cols != PlanAR4JA45::NBC) return false;
    for (int br = 0; br < c.block_rows; br++) {
        int d = 0;
        for (int bc = 0; bc < c.block_cols; bc++) d += c.offsets[(size_t)br * c.block_cols + bc] >= 0;
        if (d != PlanAR4JA45::deg(br)) return false;
    }
    return true;
}

static const char *plan_why_not(const ldpc_code &c, int variant, int dtype) {
    if (variant == LDPC_TANH && dtype != LDPC_F32) return "the fused tanh kernel exists for f32 only (f64 tanh: flood path)";
    if (dtype != LDPC_F32 && dtype != LDPC_F64) return "fused kernels exist for f32 and f64";
    if (c.sz == 0) return "code was not created from a quasi-cyclic description";
    if (!(c.sz == 32 || c.sz == 64 || c.sz == 128)) return "circulant size must be 32, 64 or 128";
    if (!plan_matches_ar4ja45(c)) return "block structure is not the AR4JA rate-4/5 plan (12x44 blocks, row weights 3,3,3,3,18x8)";
    return nullptr;
}
// LDPC_F16 = "fp16 storage in HBM, f32 arithmetic".  The only thing a fused decode keeps in HBM is the channel
// LLRs, so a fused F16 context is the f32 kernel fed fp16-rounded LLRs.
static inline int compute_dtype(int dtype) { return dtype == LDPC_F16 ? LDPC_F32 : dtype; }

// a fused (on-chip) kernel exists if the code matches a compiled QC plan, or failing that if a frame fits in LDS
const char *fused_why_not(const ldpc_code &c, int variant, int dtype) {
    dtype = compute_dtype(dtype);
    const char *p = plan_why_not(c, variant, dtype);
    if (!p) return nullptr;
    const char *j = jit_split_why_not(c, variant, dtype);
    if (!j) return nullptr;
    const char *g = fused_csr_why_not(c, variant, dtype);
    if (!g) return nullptr;
    static thread_local char buf[600];
    snprintf(buf, sizeof(buf), "built-in QC kernel: %s; run-time specialised QC kernel: %s; generic on-chip kernel: %s", p, j, g);
    return buf;
}
bool fused_supported(const ldpc_code &c, int variant, int dtype) { return fused_why_not(c, variant, dtype) == nullptr; }
// measured r01 (jpl.4096, 16384 frames): min-sum fused 10.0 vs flood 0.85 Gbit/s; tanh fused 2.09 vs flood 0.73
// (before the branch-free phi the fused tanh kernel spilled ~560 VGPRs and lost to flood: 0.53 vs 0.68).
bool fused_preferred(const ldpc_code &c, int variant, int dtype) { return fused_supported(c, variant, dtype); }

template <typename CT, int SZ>
static int launch(FusedState &s, hipStream_t st, FusedArgs &a) {
    using Cfg = FusedCfg<CT, PlanAR4JA45, SZ>;
    const int grid = (a.batch + Cfg::CPW - 1) / Cfg::CPW;
    auto kern = fused_decode_kernel<CT, PlanAR4JA45, SZ>;
    if (!a.step_mode) {
        snprintf(s.info.name, sizeof(s.info.name), "ldpc::fused_decode_kernel<%s, ldpc::PlanAR4JA45, %d>", sizeof(CT) == 8 ? "double" : "float", SZ);
        s.info.threads = Cfg::THREADS; s.info.frames_per_wg = Cfg::CPW;
    }
    if (s.timer && !a.step_mode) s.timer->begin(st);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::THREADS), 0, st, a);
    if (s.timer && !a.step_mode) s.timer->end(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(LDPC_EHIP, "fused launch: %s", hipGetErrorString(e));
    return LDPC_OK;
}

static int dispatch(FusedState &s, hipStream_t st, FusedArgs &a) {
    if (s.dtype == LDPC_F32) {
        switch (s.sz) {
            case 32: return launch<float, 32>(s, st, a);
            case 64: return launch<float, 64>(s, st, a);
            case 128: return launch<float, 128>(s, st, a);
        }
    } else if (s.dtype == LDPC_F64) {
        switch (s.sz) {
            case 32: return launch<double, 32>(s, st, a);
            case 64: return launch<double, 64>(s, st, a);
            case 128: return launch<double, 128>(s, st, a);
        }
    }
    return set_error(LDPC_EUNSUPPORTED, "no fused kernel for sz=%d dtype=%d", s.sz, s.dtype);
}

FusedState *fused_create(const ldpc_code &c, int variant, int dtype, int max_batch) {
    const char *why = fused_why_not(c, variant, dtype);
    if (why) { set_error(LDPC_EUNSUPPORTED, "%s", why); return nullptr; }
    FusedState *s = new (std::nothrow) FusedState();
    if (!s) { set_error(LDPC_ENOMEM, "out of host memory"); return nullptr; }
    s->round16 = dtype == LDPC_F16;
    dtype = compute_dtype(dtype);
    s->variant = variant; s->dtype = dtype; s->max_batch = max_batch; s->sz = c.sz; s->M = c.M; s->N = c.N; s->E = c.E;
    s->row_ptr = c.row_ptr;
    // Which kernel.  (1) a built-in instance with compile-time tables (the shipped matrices); (2) any other
    // single-circulant QC code: the same split kernel specialised at run time (jit.cc) -- unless that cannot be built
    // (compiler missing, shape out of range), in which case (3) the table-driven two-wave kernel if the block
    // structure is the AR4JA plan, else (4) the generic on-chip kernel.
    bool builtin = false;
    if (plan_why_not(c, variant, dtype) == nullptr && dtype == LDPC_F32) {
        std::vector<uint16_t> rot; std::vector<uint8_t> bcv;
        for (int br = 0; br < c.block_rows; br++)
            for (int bc = 0; bc < c.block_cols; bc++) {
                int off = c.offsets[(size_t)br * c.block_cols + bc];
                if (off >= 0) { rot.push_back((uint16_t)off); bcv.push_back((uint8_t)bc); }
            }
        const char *d = getenv("LDPC_FUSED_TABLE");
        builtin = !(d && !strcmp(d, "dyn")) && fused_msg_static_id(c.sz, rot.data(), bcv.data(), (int)rot.size()) != 0;
        if (d && !strcmp(d, "dyn")) builtin = true;   // forced table-driven kernel: not the run-time compiler either
    }
    {
        const char *k = getenv("LDPC_FUSED_KERNEL");   // A/B switches name a built-in kernel
        if (k && (!strcmp(k, "rec") || !strcmp(k, "msg"))) builtin = builtin || plan_why_not(c, variant, dtype) == nullptr;
    }
    if (!builtin && jit_split_why_not(c, variant, dtype) == nullptr) {
        s->jit = jit_split_create(c, variant, dtype);
        if (s->jit) {
            snprintf(s->info.name, sizeof(s->info.name), "%s", s->jit->name.c_str());
            s->info.threads = s->jit->threads; s->info.frames_per_wg = s->jit->frames_per_wg;
            return s;
        }
        fprintf(stderr, "[libldpc_hip] run-time specialisation failed (%s); using a table-driven kernel\n", ldpc_last_error());
    }
    if (plan_why_not(c, variant, dtype) != nullptr) {  // no QC plan: generic on-chip kernel
        if (fused_csr_why_not(c, variant, dtype) != nullptr) {
            delete s;
            set_error(LDPC_EUNSUPPORTED, "no fused kernel could be built for this code (%s)", fused_why_not(c, variant, dtype) ? fused_why_not(c, variant, dtype) : "run-time compilation failed");
            return nullptr;
        }
        s->csr = fused_csr_create(c, variant, dtype);
        if (!s->csr) { delete s; return nullptr; }
        fused_csr_set_round16(s->csr, s->round16);
        return s;
    }
    {   // LDPC_FUSED_KERNEL=rec selects the compressed-record kernel (min-sum only) for A/B measurements
        const char *k = getenv("LDPC_FUSED_KERNEL");
        s->use_msg = fused_msg_has(variant, dtype, c.sz) && !(k && !strcmp(k, "rec") && variant == LDPC_MINSUM);
        if (!s->use_msg && variant != LDPC_MINSUM) { delete s; set_error(LDPC_EUNSUPPORTED, "no fused kernel"); return nullptr; }
    }
    const int es = dtype == LDPC_F64 ? 8 : 4;
    const int cpw = c.sz >= 64 ? 1 : 64 / c.sz, V = c.sz * cpw;
    std::vector<uint32_t> tab;
    {
        std::vector<uint16_t> rot; std::vector<uint8_t> bcv;
        for (int br = 0; br < c.block_rows; br++)
            for (int bc = 0; bc < c.block_cols; bc++) {
                int off = c.offsets[(size_t)br * c.block_cols + bc];
                if (off >= 0) { rot.push_back((uint16_t)off); bcv.push_back((uint8_t)bc); }
            }
        const char *d = getenv("LDPC_FUSED_TABLE");  // LDPC_FUSED_TABLE=dyn forces the table-driven kernel
        s->static_id = (d && !strcmp(d, "dyn")) ? 0 : fused_msg_static_id(c.sz, rot.data(), bcv.data(), (int)rot.size());
    }
    {   // LDPC_FUSED_KERNEL=msg keeps the two-wave kernel where the four-wave split kernel exists
        const char *k = getenv("LDPC_FUSED_KERNEL");
        s->use_split = s->use_msg && fused_split_has(variant, dtype, c.sz, s->static_id) && !(k && !strcmp(k, "msg"));
    }
    for (int br = 0; br < c.block_rows; br++)
        for (int bc = 0; bc < c.block_cols; bc++) {
            int off = c.offsets[(size_t)br * c.block_cols + bc];
            if (off < 0) continue;
            uint32_t lo = (uint32_t)(off * cpw * es), hi = (uint32_t)(bc * V * es);
            if (lo > 0xffffu || hi > 0xffffu) { delete s; set_error(LDPC_EUNSUPPORTED, "graph table field overflow"); return nullptr; }
            tab.push_back(lo | (hi << 16));
        }
    hipError_t e = hipMalloc((void **)&s->d_tab, tab.size() * 4);
    if (e == hipSuccess) e = hipMemcpy(s->d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) { set_error(LDPC_EHIP, "fused_create: %s", hipGetErrorString(e)); fused_destroy(s); return nullptr; }
    return s;
}

void fused_destroy(FusedState *s) {
    if (!s) return;
    jit_destroy(s->jit);
    fused_csr_destroy(s->csr);
    (void)hipFree(s->d_tab);
    delete s;
}

void fused_set_timer(FusedState *s, KernelTimer *t) { if (s) { s->timer = t; fused_csr_set_timer(s->csr, t); } }

bool fused_reads_llr_once(const FusedState &s, int max_iters) {
    return s.jit != nullptr || s.csr != nullptr || (s.use_split && max_iters <= kSplitMaxIters);
}

const LaunchInfo &fused_launch_info(const FusedState &s) { return s.csr ? fused_csr_launch_info(*s.csr) : s.info; }

static int launch_jit(FusedState &s, hipStream_t st, FusedArgs &a) {
    const int grid = (a.batch + s.jit->frames_per_wg - 1) / s.jit->frames_per_wg;
    void *params[] = {&a};
    if (s.timer && !a.step_mode) s.timer->begin(st);
    hipError_t e = hipModuleLaunchKernel(s.jit->fn, (unsigned)grid, 1, 1, (unsigned)s.jit->threads, 1, 1, 0, st, params, nullptr);
    if (s.timer && !a.step_mode) s.timer->end(st);
    if (e != hipSuccess) return set_error(LDPC_EHIP, "launch of %s: %s", s.jit->name.c_str(), hipGetErrorString(e));
    return LDPC_OK;
}

const char *fused_kernel_name(const FusedState &s) {
    const LaunchInfo &li = fused_launch_info(s);
    if (li.name[0]) return li.name;
    if (s.csr) return fused_csr_kernel_name(*s.csr);
    if (s.use_split) return "fused_split_kernel";
    if (s.use_msg) return "fused_msg_kernel";
    return "fused_decode_kernel";
}

int fused_decode(FusedState &s, hipStream_t st, int max_iters, int batch, const void *d_llr, int llr_fmt, uint8_t *d_bits,
                 int32_t *d_iters, uint8_t *d_conv, double *d_final, double *d_trace) {
    if (s.csr) return fused_csr_decode(*s.csr, st, max_iters, batch, d_llr, llr_fmt, d_bits, d_iters, d_conv, d_final, d_trace);
    FusedArgs a{};
    a.tab = s.d_tab; a.llr = d_llr; a.llr_fmt = llr_fmt; a.llr_round16 = s.round16; a.bits = d_bits; a.iters = d_iters; a.conv = d_conv;
    a.final_lam = d_final; a.trace = d_trace; a.batch = batch; a.max_iters = max_iters; a.step_mode = 0;
    if (s.jit) return launch_jit(s, st, a);
    // (the split kernel packs a frame's result into one register: 9 bits for the turn it converged at)
    if (s.use_split && max_iters <= kSplitMaxIters) return fused_split_launch(s.variant, s.sz, st, a, s.timer, &s.info);
    if (s.use_msg) return fused_msg_launch(s.variant, s.dtype, s.sz, s.static_id, st, a, s.timer, &s.info);
    return dispatch(s, st, a);
}

// teacher-forced step: the oracle's messages of a row take at most two magnitudes (that is what
// min-sum produces), so they convert losslessly to the kernel's row record on the host.
template <typename CT>
static int step_typed(FusedState &s, hipStream_t st, int batch, const double *d_orig, const double *d_lam, const double *d_ne,
                      double *d_ne_out, double *d_lam_out, uint8_t *d_syn) {
    const size_t B = (size_t)batch, M = (size_t)s.M, E = (size_t)s.E;
    std::vector<double> ne(B * E);
    hipError_t e = hipMemcpyAsync(ne.data(), d_ne, B * E * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return set_error(LDPC_EHIP, "fused_step: %s", hipGetErrorString(e));
    std::vector<CT> m1(B * M), m2(B * M);
    std::vector<uint32_t> sg(B * M);
    for (size_t f = 0; f < B; f++)
        for (size_t m = 0; m < M; m++) {
            const int b = s.row_ptr[m], D = s.row_ptr[m + 1] - b;
            const double *x = &ne[f * E + b];
            double lo = INFINITY, hi = 0;
            for (int k = 0; k < D; k++) { lo = std::min(lo, fabs(x[k])); hi = std::max(hi, fabs(x[k])); }
            int idx = 0, nhi = 0;
            uint32_t F = 0;
            for (int k = 0; k < D; k++) {
                if (fabs(x[k]) == hi && hi != lo) { idx = k; nhi++; }
                else if (fabs(x[k]) != lo) return set_error(LDPC_EINVAL, "fused_step: row %zu holds more than two message magnitudes", m);
                if (std::signbit(x[k])) F |= 1u << (D - 1 - k);
            }
            if (nhi > 1) return set_error(LDPC_EINVAL, "fused_step: row %zu is not a min-sum state", m);
            m1[f * M + m] = (CT)lo; m2[f * M + m] = (CT)(nhi ? hi : lo);
            sg[f * M + m] = ((uint32_t)idx << 24) | F;
        }
    void *dm1 = nullptr, *dm2 = nullptr;
    uint32_t *dsg = nullptr;
    e = hipMalloc(&dm1, B * M * sizeof(CT));
    if (e == hipSuccess) e = hipMalloc(&dm2, B * M * sizeof(CT));
    if (e == hipSuccess) e = hipMalloc((void **)&dsg, B * M * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(dm1, m1.data(), B * M * sizeof(CT), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(dm2, m2.data(), B * M * sizeof(CT), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(dsg, sg.data(), B * M * 4, hipMemcpyHostToDevice, st);
    int rc = LDPC_OK;
    if (e != hipSuccess) rc = set_error(LDPC_EHIP, "fused_step: %s", hipGetErrorString(e));
    if (rc == LDPC_OK) {
        FusedArgs a{};
        a.tab = s.d_tab; a.llr = d_orig; a.llr_fmt = LLR_F64; a.llr_round16 = 0; a.batch = batch; a.max_iters = 1; a.step_mode = 1;
        a.st_lam = d_lam; a.st_m1 = dm1; a.st_m2 = dm2; a.st_sg = dsg; a.st_ne_out = d_ne_out; a.final_lam = d_lam_out; a.st_syn = d_syn;
        rc = dispatch(s, st, a);
    }
    (void)hipStreamSynchronize(st);
    (void)hipFree(dm1); (void)hipFree(dm2); (void)hipFree(dsg);
    return rc;
}

int fused_step(FusedState &s, hipStream_t st, int batch, const double *d_orig, const double *d_lam, const double *d_ne,
               double *d_ne_out, double *d_lam_out, uint8_t *d_syn) {
    if (s.csr) return fused_csr_step(*s.csr, st, batch, d_orig, d_lam, d_ne, d_ne_out, d_lam_out, d_syn);
    if (s.use_msg || s.jit) {  // per-edge messages: the state goes in and out as it is
        FusedArgs a{};
        a.tab = s.d_tab; a.llr = d_orig; a.llr_fmt = LLR_F64; a.llr_round16 = 0; a.batch = batch; a.max_iters = 1; a.step_mode = 1;
        a.st_lam = d_lam; a.st_ne_in = d_ne; a.st_ne_out = d_ne_out; a.final_lam = d_lam_out; a.st_syn = d_syn;
        if (s.jit) return launch_jit(s, st, a);
        if (s.use_split) return fused_split_launch(s.variant, s.sz, st, a, nullptr, nullptr);
        return fused_msg_launch(s.variant, s.dtype, s.sz, s.static_id, st, a, nullptr, nullptr);
    }
    if (s.dtype == LDPC_F64) return step_typed<double>(s, st, batch, d_orig, d_lam, d_ne, d_ne_out, d_lam_out, d_syn);
    return step_typed<float>(s, st, batch, d_orig, d_lam, d_ne, d_ne_out, d_lam_out, d_syn);
}

}  // namespace ldpc
