// fused_msg.hip -- fused on-chip flooding BP with PER-EDGE messages in VGPRs (min-sum and tanh).
//
// Same mapping, LDS layout and loop structure as fused.hip (see its header): a frame is owned by a
// workgroup of WPF waves, lam lives in LDS, one launch decodes the batch.  The difference is the
// message store: every check->variable message ne[m,n] the thread owns stays in its own VGPR
// (156 per thread for the AR4JA plan) instead of the 3-register row record.  On gfx950 only
// f32 add/sub/mul/fma, and/or/xor and add/sub_u32 issue in 2 clk; compares, selects, shifts, bfi,
// min/max/med3 take 4 (tools/microbench_valu.hip).  Rebuilding messages from records twice per turn
// costs ~32 of the record kernel's ~76 VALU-clk per edge; here an edge costs ~44:
//   phase A  t = lam[col] - ne (in place), two-min + sign parity over the row, then
//            ne' = bfi(|.| = (|t| == m1 ? 0.75 m2 : 0.75 m1), sign = signs(all t) ^ sign(t) ^ (D odd))
//            (Reference/Min.hs:75-86); tanh rule: ldpc_math.h cn_update (Reference/Orig.hs:81-92)
//   phase B  lam[col] <- ne' + lam[col], block rows in descending order (Orig.hs:95-98)
// Price: 200+ VGPRs -> 2 waves per SIMD.
#include "fused_common.h"
#include "generated_tables.h"

// LDPC_DBG: timing-only ablation builds (results are WRONG when non-zero; never shipped):
//   1 no barriers between phase-B rounds   2 no phase-B adds   4 no pass 2   8 no lam<-orig init   16 no pass 1
#ifndef LDPC_DBG
#define LDPC_DBG 0
#endif

namespace ldpc {

// graph-table access.  Dyn: encoded dwords in memory (any code with the plan's block structure), read
// with s_load.  Stat<T>: the table is a constexpr array -> rotation becomes a literal operand and the
// block-column base an immediate DS offset; nothing is loaded.
template <typename CT, int SZ>
struct DynRow {
    ctab_t p;
    __device__ __forceinline__ uint32_t lo(int k) const { return p[k] & 0xffffu; }
    __device__ __forceinline__ uint32_t hi(int k) const { return p[k] >> 16; }
};
template <typename CT, int SZ, class T, int EBEG>
struct StatRow {
    static constexpr uint32_t CPW = SZ >= 64 ? 1 : 64 / SZ, V = SZ * CPW, ES = sizeof(CT);
    __device__ __forceinline__ constexpr uint32_t lo(int k) const { return T::rot[EBEG + k] * CPW * ES; }
    __device__ __forceinline__ constexpr uint32_t hi(int k) const { return T::bc[EBEG + k] * V * ES; }
};
struct DynTab {
    ctab_t p;
    template <typename CT, int SZ, int EBEG> __device__ __forceinline__ DynRow<CT, SZ> row() const { return DynRow<CT, SZ>{p + EBEG}; }
    __device__ __forceinline__ DynTab rebase(uint32_t z) const { return DynTab{p + z}; }
};
template <class T>
struct StatTab {
    using Table = T;
    template <typename CT, int SZ, int EBEG> __device__ __forceinline__ StatRow<CT, SZ, T, EBEG> row() const { return {}; }
    __device__ __forceinline__ StatTab rebase(uint32_t) const { return {}; }
};
template <class Tab> struct IsStatic : std::false_type {};
template <class T> struct IsStatic<StatTab<T>> : std::true_type {};

// Phase-B "rounds" for a compile-time table.  A column's contributions must be added in descending
// row order (Orig.hs:96).  round(e) = number of LATER edges (higher block row) in the same block
// column; edges of one round touch every block column at most once, so all targets of a round are
// distinct and a round needs no internal ordering.  Rounds 0,1,2,.. reproduce exactly the per-column
// order of the block-row-by-block-row schedule with max-column-degree barriers instead of NBR.
template <class T>
struct Rounds {
    static constexpr int round_of(int e) {
        int c = 0;
        for (int j = e + 1; j < T::NEDGE; j++) c += (T::bc[j] == T::bc[e]) ? 1 : 0;
        return c;
    }
    static constexpr int num_rounds() {
        int m = 0;
        for (int e = 0; e < T::NEDGE; e++) m = round_of(e) + 1 > m ? round_of(e) + 1 : m;
        return m;
    }
    static constexpr int count(int q) {
        int c = 0;
        for (int e = 0; e < T::NEDGE; e++) c += round_of(e) == q ? 1 : 0;
        return c;
    }
    static constexpr int round0_edge(int bc) {  // the edge that is added first into block column bc
        for (int e = T::NEDGE - 1; e >= 0; e--)
            if (T::bc[e] == bc) return e;
        return -1;
    }
    static constexpr int nth(int q, int i) {  // i-th edge of round q, highest edge index first
        int c = 0;
        for (int e = T::NEDGE - 1; e >= 0; e--)
            if (round_of(e) == q) { if (c == i) return e; c++; }
        return -1;
    }
};

// one chunk [I0, I1) of round Q: read every target, add, write back
template <typename CT, int SZ, class T, int Q, int I0, int I1>
__device__ __forceinline__ void round_chunk_b(char *lds, uint32_t p4, uint32_t vmask, const CT *msg, const CT *orig_rot) {
    constexpr uint32_t CPW = SZ >= 64 ? 1 : 64 / SZ, V = SZ * CPW, ES = sizeof(CT);
    asm volatile("" : "+v"(p4));
    if constexpr (Q == 0) {
        // first contribution of every column: lam = orig + ne' is a plain store -- the thread holds the
        // channel LLR of the column it writes here (orig_rot), so there is no lam <- orig pass at all
        static_for<I0, I1>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int e = Rounds<T>::nth(Q, i);
            lds_st<CT>(lds + T::bc[e] * V * ES, (p4 + T::rot[e] * CPW * ES) & vmask, msg[e] + orig_rot[T::bc[e]]);
        });
        return;
    }
    CT cur[I1 - I0];
    uint32_t adr[I1 - I0];
    static_for<I0, I1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int e = Rounds<T>::nth(Q, i);
        adr[i - I0] = (p4 + T::rot[e] * CPW * ES) & vmask;
        cur[i - I0] = lds_ld<CT>(lds + T::bc[e] * V * ES, adr[i - I0]);
    });
    static_for<I0, I1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int e = Rounds<T>::nth(Q, i);
        lds_st<CT>(lds + T::bc[e] * V * ES, adr[i - I0], msg[e] + cur[i - I0]);
    });
}
template <typename CT, int SZ, class T, int Q, int I0>
__device__ __forceinline__ void round_b(char *lds, uint32_t p4, uint32_t vmask, const CT *msg, const CT *orig_rot) {
    constexpr int CNT = Rounds<T>::count(Q), CH = 8;  // 16 pushes the round-0 LLR registers into scratch
    if constexpr (I0 < CNT) {
        round_chunk_b<CT, SZ, T, Q, I0, (I0 + CH < CNT ? I0 + CH : CNT)>(lds, p4, vmask, msg, orig_rot);
        round_b<CT, SZ, T, Q, I0 + CH>(lds, p4, vmask, msg, orig_rot);
    }
}

template <int RPL, int HSTEP>
__device__ __forceinline__ uint32_t row_addr(uint32_t a0, uint32_t p4, uint32_t lo, uint32_t vmask, int h) {
    if (h == 0) return a0;
    if (RPL == 2) return a0 ^ (uint32_t)HSTEP;
    return ((p4 + HSTEP * h) + lo) & vmask;
}

// phase A for the RPL rows a lane owns in one block row of degree D.  msg: [RPL][D] registers.
template <typename CT, int VARIANT, int D, int RPL, int HSTEP, bool SYNDROME_ONLY, class Row>
__device__ __forceinline__ bool rows_a(const char *lds, Row tabrow, uint32_t p4, uint32_t vmask, CT *msg) {
    asm volatile("" : "+v"(p4));  // keeps the loop-invariant address arithmetic inside the turn loop, row by row
    CT l[RPL][D];
    static_for<0, D>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const uint32_t lo = tabrow.lo(k), hi = tabrow.hi(k);
        uint32_t a0 = (p4 + lo) & vmask;   // position inside the block column; `hi` (its base) is added below
#pragma unroll
        for (int h = 0; h < RPL; h++) l[h][k] = lds_ld<CT>(lds + hi, row_addr<RPL, HSTEP>(a0, p4, lo, vmask, h));
    });
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
        CT *m = msg + h * D;
        if constexpr (VARIANT == LDPC_V_MINSUM && sizeof(CT) == 4) {
            static_assert(D >= 2, "min-sum needs degree >= 2");
            uint32_t X = 0;
            float m1 = INFINITY, m2 = INFINITY;
            static_for<0, D>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                float t = l[h][k] - m[k];
                m[k] = t;
                X ^= __float_as_uint(t);
                float a = fabsf(t);
                m2 = __builtin_amdgcn_fmed3f(m1, m2, a);
                m1 = fminf(m1, a);
            });
            // sign(ne'_k) = signs(all t) ^ sign(t_k) ^ (D odd).  The row part of it is folded into the
            // two candidate magnitudes once per row; per edge one 3-input bit operation then takes the
            // magnitude bits from the candidate and sign = candidate.sign ^ t.sign  (v_bitop3_b32).
            const uint32_t flip = (X ^ ((D & 1) ? 0x80000000u : 0u)) & 0x80000000u;
            const uint32_t c1 = __float_as_uint(0.75f * m1) ^ flip;  // 0.75f*: the one rounding of Min.hs:78
            const uint32_t c2 = __float_as_uint(0.75f * m2) ^ flip;
            static_for<0, D>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                float t = m[k];
                uint32_t c = (fabsf(t) == m1) ? c2 : c1;   // leave-one-out min: m2 at the arg-min (ties: m2 == m1)
                uint32_t tb = __float_as_uint(t);
                // (c & ~S) | ((c ^ tb) & S), S = sign mask: truth table 0x78 with A=0xF0, B=0xCC, C=0xAA
                m[k] = __uint_as_float(__builtin_amdgcn_bitop3_b32(c, tb, 0x80000000u, 0x78));
            });
        } else {
            CT t[D];
#pragma unroll
            for (int k = 0; k < D; k++) t[k] = l[h][k] - m[k];
            cn_update<CT, VARIANT, D>(t);
#pragma unroll
            for (int k = 0; k < D; k++) m[k] = t[k];
        }
    }
    return any;
}

// ---- phase A as three stages, so the kernel can software-pipeline rows: the gathered lam values are
// dead after pass 1, so the NEXT row's gather is issued between pass 1 and pass 2 into the same
// registers and its LDS latency hides behind pass 2.
template <typename CT, int D, int RPL, int HSTEP, int LD, class Row>
__device__ __forceinline__ void row_gather(const char *lds, Row tabrow, uint32_t p4, uint32_t vmask, CT (*l)[LD]) {
    asm volatile("" : "+v"(p4));  // keeps the loop-invariant address arithmetic inside the turn loop
    static_for<0, D>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const uint32_t lo = tabrow.lo(k), hi = tabrow.hi(k);
        uint32_t a0 = (p4 + lo) & vmask;
#pragma unroll
        for (int h = 0; h < RPL; h++) l[h][k] = lds_ld<CT>(lds + hi, row_addr<RPL, HSTEP>(a0, p4, lo, vmask, h));
    });
}
template <typename CT> struct RowRed { CT m1, m2; uint32_t X; };
// pass 1: row parity of hard(lam); t = lam - ne stored in place of ne; two-min + sign parity (min-sum f32)
template <typename CT, int VARIANT, int D, int RPL, int LD>
__device__ __forceinline__ bool row_pass1(const CT (*l)[LD], CT *msg, RowRed<CT> *red) {
    bool any = false;
#pragma unroll
    for (int h = 0; h < RPL; h++) {
        bool par = false;
#pragma unroll
        for (int k = 0; k < D; k++) par ^= (l[h][k] > CT(0));
        any |= par;
        CT *m = msg + h * D;
        if constexpr (VARIANT == LDPC_V_MINSUM && sizeof(CT) == 4) {
            uint32_t X = 0;
            float m1 = INFINITY, m2 = INFINITY;
            static_for<0, D>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                float t = l[h][k] - m[k];
                m[k] = t;
                X ^= __float_as_uint(t);
                float a = fabsf(t);
                m2 = __builtin_amdgcn_fmed3f(m1, m2, a);
                m1 = fminf(m1, a);
            });
            red[h].m1 = m1; red[h].m2 = m2; red[h].X = X;
        } else {
#pragma unroll
            for (int k = 0; k < D; k++) m[k] = l[h][k] - m[k];
        }
    }
    return any;
}
// pass 2: ne' from t (in place)
template <typename CT, int VARIANT, int D, int RPL>
__device__ __forceinline__ void row_pass2(CT *msg, const RowRed<CT> *red) {
#pragma unroll
    for (int h = 0; h < RPL; h++) {
        CT *m = msg + h * D;
        if constexpr (VARIANT == LDPC_V_MINSUM && sizeof(CT) == 4) {
            static_assert(D >= 2, "min-sum needs degree >= 2");
            const float m1 = red[h].m1;
            const uint32_t flip = (red[h].X ^ ((D & 1) ? 0x80000000u : 0u)) & 0x80000000u;
            const uint32_t c1 = __float_as_uint(0.75f * m1) ^ flip;  // 0.75f*: the one rounding of Min.hs:78
            const uint32_t c2 = __float_as_uint(0.75f * red[h].m2) ^ flip;
            static_for<0, D>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                float t = m[k];
                uint32_t c = (fabsf(t) == m1) ? c2 : c1;
                m[k] = __uint_as_float(__builtin_amdgcn_bitop3_b32(c, __float_as_uint(t), 0x80000000u, 0x78));
            });
        } else {
            CT t[D];
#pragma unroll
            for (int k = 0; k < D; k++) t[k] = m[k];
            cn_update<CT, VARIANT, D>(t);
#pragma unroll
            for (int k = 0; k < D; k++) m[k] = t[k];
        }
    }
}

// phase B: lam[col_k] <- ne'_k + lam[col_k]; the D x RPL targets of a block row are distinct columns
template <typename CT, int D, int RPL, int HSTEP, class Row>
__device__ __forceinline__ void rows_b(char *lds, Row tabrow, uint32_t p4, uint32_t vmask, const CT *msg) {
    asm volatile("" : "+v"(p4));
    CT cur[RPL][D];
    uint32_t adr[D];
    static_for<0, D>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const uint32_t lo = tabrow.lo(k), hi = tabrow.hi(k);
        adr[k] = (p4 + lo) & vmask;
#pragma unroll
        for (int h = 0; h < RPL; h++) cur[h][k] = lds_ld<CT>(lds + hi, row_addr<RPL, HSTEP>(adr[k], p4, lo, vmask, h));
    });
    static_for<0, D>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const uint32_t lo = tabrow.lo(k), hi = tabrow.hi(k);
#pragma unroll
        for (int h = 0; h < RPL; h++) lds_st<CT>(lds + hi, row_addr<RPL, HSTEP>(adr[k], p4, lo, vmask, h), msg[h * D + k] + cur[h][k]);
    });
}

template <typename CT, int VARIANT, class Plan, int SZ>
struct MsgCfg : FusedCfg<CT, Plan, SZ> {
    using B = FusedCfg<CT, Plan, SZ>;
    static constexpr int NMSG = Plan::NEDGE * B::RPL;
    // messages + orig + row temporaries: > 168 VGPRs whatever we do -> plan for 2 waves per SIMD
    static constexpr int WAVES_PER_EU = (sizeof(CT) == 8 || B::RPL >= 2) ? 1 : 2;
};

template <typename CT, int VARIANT, class Plan, int SZ, class Tab>
__global__ __launch_bounds__((MsgCfg<CT, VARIANT, Plan, SZ>::THREADS), (MsgCfg<CT, VARIANT, Plan, SZ>::WAVES_PER_EU)) void fused_msg_kernel(FusedArgs A) {
    using Cfg = MsgCfg<CT, VARIANT, Plan, SZ>;
    constexpr int RPL = Cfg::RPL, CPW = Cfg::CPW, V = Cfg::V, N = Cfg::N, WPF = Cfg::WPF, HSTEP = Cfg::HSTEP;
    constexpr int RSTEP = Cfg::THREADS;
    constexpr uint32_t ES = sizeof(CT);
    constexpr uint32_t vmask = V * ES - 1;
    __shared__ __attribute__((aligned(16))) char lds[Cfg::LDS_BYTES];

    const uint32_t lane = threadIdx.x;
    const uint32_t sub = lane % CPW;
    const uint32_t r0 = lane / CPW;
    const long long frame = (long long)blockIdx.x * CPW + sub;
    const bool valid = frame < A.batch;
    const uint32_t p4 = lane * ES;
    const size_t fN = (size_t)(valid ? frame : 0) * N;
    const size_t fE = (size_t)(valid ? frame : 0) * Plan::NEDGE * SZ;

    // Channel LLRs kept in registers for phase B.  Table-driven kernel: the thread's OWN columns
    // (lam <- orig pass).  Compile-time table (kRot): the column the thread WRITES in round 0 of each
    // block column, i.e. own column rotated by that circulant's offset.
    constexpr bool kRot = IsStatic<Tab>::value && RPL == 1;
    auto llr_at = [&](size_t gi) -> CT {
        return A.llr_is_f64 ? (CT) reinterpret_cast<const double *>(A.llr)[gi] : (CT) reinterpret_cast<const float *>(A.llr)[gi];
    };
    CT orig[Cfg::NORIG];
    if constexpr (kRot) {
        // loaded after the LDS fill below (keeps the prologue's register pressure down)
    } else if (A.llr_is_f64) {
        const double *src = reinterpret_cast<const double *>(A.llr) + fN + r0;
#pragma unroll
        for (int i = 0; i < Cfg::NORIG; i++) orig[i] = (CT)src[(i / RPL) * SZ + RSTEP * (i % RPL)];
    } else {
        const float *src = reinterpret_cast<const float *>(A.llr) + fN + r0;
#pragma unroll
        for (int i = 0; i < Cfg::NORIG; i++) orig[i] = (CT)src[(i / RPL) * SZ + RSTEP * (i % RPL)];
    }
    // messages: index ebeg(br)*RPL + h*D + k ; Orig.hs:64-65 orig_ne = 0
    CT msg[Cfg::NMSG];
#pragma unroll
    for (int i = 0; i < Cfg::NMSG; i++) msg[i] = CT(0);
    if (A.step_mode) {  // teacher-forced state: messages given in CSR edge order (row-major, ascending column)
        static_for<0, Plan::NBR>([&](auto brc) {
            constexpr int br = decltype(brc)::value;
            constexpr int D = Plan::deg(br);
#pragma unroll
            for (int h = 0; h < RPL; h++)
#pragma unroll
                for (int k = 0; k < D; k++)
                    msg[Plan::ebeg(br) * RPL + h * D + k] = (CT)A.st_ne_in[fE + (size_t)SZ * Plan::ebeg(br) + (size_t)D * (r0 + RSTEP * h) + k];
        });
    }
    static_for<0, Plan::NBC>([&](auto bcc) {
        constexpr int bc = decltype(bcc)::value;
#pragma unroll
        for (int h = 0; h < RPL; h++) {
            CT v;
            if constexpr (kRot) v = llr_at(fN + bc * SZ + r0); else v = orig[bc * RPL + h];
            if (A.step_mode) v = (CT)A.st_lam[fN + bc * SZ + r0 + RSTEP * h];
            lds_st<CT>(lds, (p4 + HSTEP * h) | (bc * V * ES), v);
        }
    });
    if constexpr (WPF > 1) __syncthreads();
    if constexpr (kRot) {
        asm volatile("" ::: "memory");
        using T = typename Tab::Table;
        static_for<0, Plan::NBC>([&](auto bcc) {
            constexpr int bc = decltype(bcc)::value;
            constexpr int e0 = Rounds<T>::round0_edge(bc);
            orig[bc] = llr_at(fN + bc * SZ + ((r0 + T::rot[e0]) & (SZ - 1)));
        });
    }

    unsigned long long fmask = ~0ull;
    if constexpr (CPW > 1) {
        unsigned long long m = 0;
#pragma unroll
        for (int i = 0; i < 64; i += CPW) m |= 1ull << i;
        fmask = m << sub;
    }

    bool active = valid;
    bool converged = false;
    int n_done = 0;
    const int turns = A.step_mode ? 1 : A.max_iters;

    for (int n = 0;; n++) {
        if (!__any(active)) break;
        if (A.trace && active) {
            static_for<0, Plan::NBC>([&](auto bcc) {
                constexpr int bc = decltype(bcc)::value;
#pragma unroll
                for (int h = 0; h < RPL; h++)
                    A.trace[((size_t)frame * (A.max_iters + 1) + n) * N + bc * SZ + r0 + RSTEP * h] =
                        (double)lds_ld<CT>(lds, (p4 + HSTEP * h) | (bc * V * ES));
            });
        }
        const bool last = (n >= turns);
        Tab tabA;
        if constexpr (std::is_same<Tab, DynTab>::value) tabA = DynTab{(ctab_t)A.tab + opaque_uniform_zero()};
        const uint32_t p4a = p4;
        bool unsat = false;
        if (active) {
            if (!last) {
                // (a software-pipelined variant -- next row's gather issued between pass 1 and pass 2 --
                //  measured no faster and costs 18 VGPRs that push the phase-B registers into scratch)
                static_for<0, Plan::NBR>([&](auto brc) {
                    constexpr int br = decltype(brc)::value;
                    unsat |= rows_a<CT, VARIANT, Plan::deg(br), RPL, HSTEP, false>(lds, tabA.template row<CT, SZ, Plan::ebeg(br)>(), p4a, vmask, &msg[Plan::ebeg(br) * RPL]);
                });
            } else {
                static_for<0, Plan::NBR>([&](auto brc) {
                    constexpr int br = decltype(brc)::value;
                    unsat |= rows_a<CT, VARIANT, Plan::deg(br), RPL, HSTEP, true>(lds, tabA.template row<CT, SZ, Plan::ebeg(br)>(), p4a, vmask, (CT *)nullptr);
                });
            }
        }
        const unsigned long long ub = __ballot(unsat);
        bool frame_unsat = (ub & fmask) != 0ull;
        if constexpr (WPF > 1) {
            volatile uint32_t *flags = reinterpret_cast<volatile uint32_t *>(lds + Cfg::LAM_BYTES);
            if ((lane & 63) == 0) flags[lane >> 6] = frame_unsat ? 1u : 0u;
            __syncthreads();
            frame_unsat = (flags[0] | flags[1]) != 0u;
        }
        if (A.step_mode) {
            if (valid && r0 == 0) A.st_syn[frame] = frame_unsat ? 0 : 1;
        } else if (active && !frame_unsat) {
            converged = true; active = false; n_done = n;
        }
        if (last) {
            if (active) { active = false; n_done = n; }
            break;
        }
        if (active) {
            Tab tabB;
            if constexpr (std::is_same<Tab, DynTab>::value) tabB = DynTab{(ctab_t)A.tab + opaque_uniform_zero()};
            const uint32_t p4b = p4;
            if constexpr (kRot) {
                using T = typename Tab::Table;
                static_for<0, Rounds<T>::num_rounds()>([&](auto qc) {
                    if (!(LDPC_DBG & 2)) round_b<CT, SZ, T, decltype(qc)::value, 0>(lds, p4b, vmask, msg, orig);
                    if constexpr (WPF > 1) if (!(LDPC_DBG & 1)) __syncthreads();  // the next round adds into the same columns
                });
            } else {
                static_for<0, Plan::NBC>([&](auto bcc) {
                    constexpr int bc = decltype(bcc)::value;
#pragma unroll
                    for (int h = 0; h < RPL; h++) if (!(LDPC_DBG & 8)) lds_st<CT>(lds, (p4 + HSTEP * h) | (bc * V * ES), orig[bc * RPL + h]);
                });
                if constexpr (WPF > 1) __syncthreads();
                static_rfor<0, Plan::NBR>([&](auto brc) {
                    constexpr int br = decltype(brc)::value;
                    rows_b<CT, Plan::deg(br), RPL, HSTEP>(lds, tabB.template row<CT, SZ, Plan::ebeg(br)>(), p4b, vmask, &msg[Plan::ebeg(br) * RPL]);
                    if constexpr (WPF > 1) __syncthreads();
                });
            }
        }
        if (A.step_mode) break;
    }

    if (!valid) return;
    if (A.step_mode) {
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
            for (int h = 0; h < RPL; h++)
#pragma unroll
                for (int k = 0; k < D; k++)
                    A.st_ne_out[fE + (size_t)SZ * Plan::ebeg(br) + (size_t)D * (r0 + RSTEP * h) + k] = (double)msg[Plan::ebeg(br) * RPL + h * D + k];
        });
        return;
    }
    static_for<0, Plan::NBC>([&](auto bcc) {
        constexpr int bc = decltype(bcc)::value;
#pragma unroll
        for (int h = 0; h < RPL; h++) {
            size_t gi = fN + bc * SZ + r0 + RSTEP * h;
            CT own;
            if constexpr (kRot) own = llr_at(gi); else own = orig[bc * RPL + h];
            CT v = converged ? lds_ld<CT>(lds, (p4 + HSTEP * h) | (bc * V * ES)) : own;
            A.bits[gi] = v > CT(0) ? 1 : 0;
            if (A.final_lam) A.final_lam[gi] = (double)v;
        }
    });
    if (r0 == 0) {
        if (A.iters) A.iters[frame] = n_done;
        if (A.conv) A.conv[frame] = converged ? 1 : 0;
    }
}

template <typename CT, int VARIANT, int SZ, class Tab>
static int launch_msg(hipStream_t st, FusedArgs &a, KernelTimer *timer) {
    using Cfg = MsgCfg<CT, VARIANT, PlanAR4JA45, SZ>;
    const int grid = (a.batch + Cfg::CPW - 1) / Cfg::CPW;
    if (timer && !a.step_mode) timer->begin(st);
    hipLaunchKernelGGL((fused_msg_kernel<CT, VARIANT, PlanAR4JA45, SZ, Tab>), dim3(grid), dim3(Cfg::THREADS), 0, st, a);
    if (timer && !a.step_mode) timer->end(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(LDPC_EHIP, "fused_msg launch: %s", hipGetErrorString(e));
    return LDPC_OK;
}

bool fused_msg_has(int variant, int dtype, int sz) {
    if (!(sz == 32 || sz == 64 || sz == 128)) return false;
    if (variant == LDPC_MINSUM) return dtype == LDPC_F32 || dtype == LDPC_F64;
    return dtype == LDPC_F32;  // tanh: f32 (phi domain); f64 tanh stays on the flood path
}

// which compiled-in table (if any) equals this code's rotation table: 0 = none, 1 = jpl.1024, 2 = jpl.4096
template <class T>
static bool table_equals(int sz, const uint16_t *rot, const uint8_t *bc, int nedge) {
    if (sz != T::SZ || nedge != T::NEDGE) return false;
    for (int e = 0; e < nedge; e++)
        if (rot[e] != T::rot[e] || bc[e] != T::bc[e]) return false;
    return true;
}
int fused_msg_static_id(int sz, const uint16_t *rot, const uint8_t *bc, int nedge) {
    if (table_equals<TabJpl1024>(sz, rot, bc, nedge)) return 1;
    if (table_equals<TabJpl4096>(sz, rot, bc, nedge)) return 2;
    return 0;
}

int fused_msg_launch(int variant, int dtype, int sz, int static_id, hipStream_t st, FusedArgs &a, KernelTimer *timer) {
    // compile-time tables: f32 kernels of the shipped codes
    if (dtype == LDPC_F32 && static_id == 1 && sz == 32)
        return variant == LDPC_MINSUM ? launch_msg<float, LDPC_V_MINSUM, 32, StatTab<TabJpl1024>>(st, a, timer)
                                      : launch_msg<float, LDPC_V_TANH, 32, StatTab<TabJpl1024>>(st, a, timer);
    if (dtype == LDPC_F32 && static_id == 2 && sz == 128)
        return variant == LDPC_MINSUM ? launch_msg<float, LDPC_V_MINSUM, 128, StatTab<TabJpl4096>>(st, a, timer)
                                      : launch_msg<float, LDPC_V_TANH, 128, StatTab<TabJpl4096>>(st, a, timer);
#define CASE_SZ(CT, V)                                                   \
    switch (sz) {                                                        \
        case 32: return launch_msg<CT, V, 32, DynTab>(st, a, timer);     \
        case 64: return launch_msg<CT, V, 64, DynTab>(st, a, timer);     \
        case 128: return launch_msg<CT, V, 128, DynTab>(st, a, timer);   \
    }
    if (variant == LDPC_MINSUM && dtype == LDPC_F32) { CASE_SZ(float, LDPC_V_MINSUM) }
    else if (variant == LDPC_MINSUM && dtype == LDPC_F64) { CASE_SZ(double, LDPC_V_MINSUM) }
    else if (variant == LDPC_TANH && dtype == LDPC_F32) { CASE_SZ(float, LDPC_V_TANH) }
#undef CASE_SZ
    return set_error(LDPC_EUNSUPPORTED, "no per-edge-message fused kernel for variant=%d dtype=%d sz=%d", variant, dtype, sz);
}

}  // namespace ldpc
