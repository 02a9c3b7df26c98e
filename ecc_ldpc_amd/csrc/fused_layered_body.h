// fused_layered_body.h -- the row-layered schedule ON-CHIP for the codes the split kernel takes (EXTENSION: the reference has
// no layered decoder; specification = oracle/ldpc_oracle.c oracle_decode_layered, HBM implementation = layered_qc.hip).
//
// Same workgroup as the flooding split kernel (fused_split_body.h): lam in LDS, NP wave groups, messages in registers.  A LAYER = a
// block row; layers run in order.  Two ways to give the groups work:
//   * block rows dealt to the groups (plans with NP /= 2, and rows lighter than LAY_SPLIT_MIN_DEG = 8): thread (group, r) = row r of
//     every circulant of its group's block rows; ONE group works per layer (its rows read the lam cells of their columns, apply
//     the check rule, write lam back -- the rows of a block row touch distinct columns), the others wait at the barrier that
//     ends the layer;
//   * rows split between the two groups (NP == 2, the default: Halves below): both groups work on every layer, two barriers.
// No column "rounds", no second pass over LDS, no channel-LLR registers: a sweep is NBR x {gather, rule, scatter, barrier(s)}.
//   t_k = lam[c_k] - msg_k;  odd |= XOR_k hard(lam[c_k]);  msg' = rule(t);  new_k = t_k + msg'_k;  flip |= hard(new_k) != hard(lam[c_k])
// Stopping rule (the specification's): before the first sweep the syndrome of the channel decisions; after a sweep "no check
// it saw was odd and no hard decision changed"; out of sweeps -> the channel's hard decisions, as Orig.hs:70.
// Arithmetic and its order are those of layered_qc_kernel<float> (and of the flooding kernels' check rule): the two layered
// kernels agree bit for bit (tests/test_layered_gpu.py), f32 hard bits / flags / sweep counts against the Double oracle.
#pragma once
#include "fused_pk16_body.h"   // (brings fused_split_body.h; the packed-fp16 primitives serve the second kernel of this file)

namespace ldpc {
namespace lay {

// one layer for the row a lane owns in a block row of degree D (min-sum, f32).  msg: [D] message registers.
// FIRST: the first sweep (messages are zero: t = lam - 0).  -> true if the row's parity was odd or a hard decision flipped
template <int D, bool FIRST, bool SYNDROME_ONLY, class Row>
__device__ __forceinline__ bool layer_row(char *lds, Row tabrow, uint32_t p4, uint32_t vmask, float *m) {
    asm volatile("" : "+v"(p4));   // keeps the loop-invariant address arithmetic inside the sweep loop, row by row
    if constexpr (SYNDROME_ONLY || FIRST) LDPC_COLD_PATH();   // (FIRST: one sweep per frame -- tools/isa_histogram.py prices the ordinary one)
    float l[D];
    uint32_t adr[D];
    static_for<0, D>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        adr[k] = qc_wrap(p4 + tabrow.lo(k), vmask);
        l[k] = lds_ld<float>(lds + tabrow.hi(k), adr[k]);
    });
    bool par = false;
#pragma unroll
    for (int k = 0; k < D; k++) par ^= (l[k] > 0.0f);
    if constexpr (SYNDROME_ONLY) return par;
    static_assert(D >= 2, "min-sum needs degree >= 2");
    // the check rule exactly as fused_rows.h rows_a has it (two-min + sign word, one rounding: the 3/4 of Min.hs:78)
    uint32_t X = 0;
    float m1 = INFINITY, m2 = INFINITY;
    float t[D];
    static_for<0, D>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        t[k] = FIRST ? l[k] - 0.0f : l[k] - m[k];
        X ^= __float_as_uint(t[k]);
        const float a = fabsf(t[k]);
        m2 = __builtin_amdgcn_fmed3f(m1, m2, a);
        m1 = fminf(m1, a);
    });
    const uint32_t flipbit = (X ^ ((D & 1) ? 0x80000000u : 0u)) & 0x80000000u;
    const uint32_t c1 = __float_as_uint(0.75f * m1) ^ flipbit;
    const uint32_t c2 = __float_as_uint(0.75f * m2) ^ flipbit;
    bool flip = false;
    static_for<0, D>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const uint32_t c = (fabsf(t[k]) == m1) ? c2 : c1;
        const float nm = __uint_as_float(__builtin_amdgcn_bitop3_b32(c, __float_as_uint(t[k]), 0x80000000u, 0x78));
        m[k] = nm;
        const float nw = t[k] + nm;
        flip |= (nw > 0.0f) != (l[k] > 0.0f);
        lds_st<float>(lds + tabrow.hi(k), adr[k], nw);
    });
    return par || flip;
}

// ---- rows split between the two wave groups (plans with NP == 2) --------------------------------------------------------
// With block rows dealt to the groups alternately, ONE group works per layer: of the 16 waves a CU holds, 8 compute at any time, and
// the kernel issues at 0.32-0.36 of the VALU peak (profiles/r03_bench_matrix.txt) -- it waits on LDS round trips and barriers.  Here
// thread (g, r) owns row r of EVERY block row, but only HALF of its edges (g = 0: the first D/2, g = 1: the rest): both groups work
// in every layer.  A layer is then
//     gather + t = lam - msg + partial (min1, min2, sign word, parity) over the own edges  ->  three words to LDS  ->  barrier
//     the partner's three words  ->  the row's min1 / min2 / signs  ->  new messages and lam for the own edges  ->  barrier
// Same values in the same operations as layer_row above (the minimum and second minimum of a set do not depend on how it is
// split), so the kernel stays bit for bit the HBM layered kernel.
#ifndef LAY_ROW_SPLIT
#define LAY_ROW_SPLIT 1
#endif
#ifndef LAY_SPLIT_MIN_DEG
#define LAY_SPLIT_MIN_DEG 8    // rows lighter than this stay whole with the group that owns the block row: one barrier instead of two
                               // (jpl.4096, weight-3 rows split / whole: f32 45.9 / 48.1, f16pk 74.6 / 80.0 Gbit/s at 3 dB -- profiles/r03_layered_rs_ab.txt;
                               //  codes with rows of weight 3..13, thresholds 2..33: 8 is the best or within 2 % of it -- profiles/r03_layered_mindeg_jit.txt)
#endif
template <class Plan> struct Halves {
    static constexpr bool split(int br) { return Plan::deg(br) >= LAY_SPLIT_MIN_DEG; }
    static constexpr int cnt(int br, int g) {
        if (!split(br)) return Plan::owner_br(br) == g ? Plan::deg(br) : 0;
        return g == 0 ? Plan::deg(br) / 2 : Plan::deg(br) - Plan::deg(br) / 2;
    }
    static constexpr int k0(int br, int g) { return (split(br) && g == 1) ? Plan::deg(br) / 2 : 0; }
    static constexpr int slot0(int br, int g) { int c = 0; for (int b = 0; b < br; b++) c += cnt(b, g); return c; }
    static constexpr int NMSG = slot0(Plan::NBR, 0) > slot0(Plan::NBR, 1) ? slot0(Plan::NBR, 0) : slot0(Plan::NBR, 1);
    static constexpr bool ok() { for (int b = 0; b < Plan::NBR; b++) if (Plan::deg(b) < 2) return false; return Plan::NP == 2; }
};

template <class Plan, int SZ, class T, int P, bool RS>
__device__ __forceinline__ void body(const FusedArgs &A, char *lds, const uint32_t tid) {
    using S = Split<Plan, T>;
    using H = Halves<Plan>;
    constexpr int CPW = QcGeom<SZ>::CPW, V = QcGeom<SZ>::V, VT = QcGeom<SZ>::VT;
    constexpr int N = Plan::NBC * SZ, THREADS = Plan::NP * VT, NW = THREADS / 64;
    constexpr uint32_t ES = 4, vmask = V * ES - 1;
    constexpr int LAM_BYTES = (Plan::NBC * V * (int)ES + 15) / 16 * 16;
    constexpr int NBCP = (Plan::NBC + Plan::NP - 1) / Plan::NP;
    if constexpr (VT != V) { if ((tid % VT) >= (uint32_t)V) return; }
    const uint32_t p4 = (tid % VT) * ES;
    struct Where {
        uint32_t sub, r0; long long frame; bool valid; size_t fN;
        __device__ __forceinline__ Where(uint32_t p, int batch) {
            asm volatile("" : "+v"(p));
            const uint32_t lane = p / ES;
            sub = lane % CPW;
            r0 = lane / CPW;
            frame = (long long)blockIdx.x * CPW + sub;
            valid = frame < batch;
            fN = (size_t)(valid ? frame : 0) * N;
        }
    };
    constexpr int NM = RS ? H::NMSG : S::NMSG;
    constexpr uint32_t EX0 = LAM_BYTES + 4 * NW + 12;            // RS: exchange area [word 0..2][group][VT] dwords
    constexpr uint32_t EXW = 2 * VT * ES, EXG = VT * ES;
    float msg[NM];
#pragma unroll
    for (int i = 0; i < NM; i++) msg[i] = 0.0f;
    // ---- lam <- channel LLRs: group P fills the block columns bc with bc % NP == P; their hard decisions stay in `obits`
    typename SplitResult<NBCP>::Bits obits{};
    {
        const Where w(p4, A.batch);
        with_llr_format(A.llr_fmt, [&](auto fc) {
            constexpr int FMT = decltype(fc)::value;
            static_for<0, Plan::NBC>([&](auto bcc) {
                constexpr int bc = decltype(bcc)::value;
                if constexpr ((bc % Plan::NP) == P) {
                    const float v = maybe_round_f16<float>(load_llr_as<float, FMT>(A.llr, w.fN + bc * SZ + w.r0), A.llr_round16);
                    obits.set(bc / Plan::NP, v > 0.0f);
                    lds_st<float>(lds, p4 + (bc * V * ES), v);
                }
            });
        });
    }
    __syncthreads();

    volatile uint32_t *flags = reinterpret_cast<volatile uint32_t *>(lds + LAM_BYTES);
    constexpr uint32_t FULL = (1u << CPW) - 1;
    uint32_t done = 0;     // bit s = frame s of this workgroup has finished (workgroup-uniform)
#pragma unroll
    for (int s2 = 0; s2 < CPW; s2++) done |= ((long long)blockIdx.x * CPW + s2 < A.batch) ? 0u : (1u << s2);
    SplitResult<NBCP> res;
    res.bits = obits;
    const uint32_t my_slot = (p4 / ES) % CPW;

    // workgroup-wide OR, per frame, of a lane flag; also the barrier between a sweep's last layer and what follows
    auto frames_with = [&](bool lane_flag) -> uint32_t {
        const unsigned long long ub = __ballot(lane_flag);
        uint32_t wbits = 0;
#pragma unroll
        for (int s2 = 0; s2 < CPW; s2++) {
            unsigned long long mk = 0;
            for (int i = 0; i < 64; i += CPW) mk |= 1ull << i;
            wbits |= ((ub & (mk << s2)) != 0ull) ? (1u << s2) : 0u;
        }
        if ((tid & 63) == 0) flags[tid >> 6] = wbits;
        __syncthreads();
        uint32_t f = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) f |= flags[w];
        f = __builtin_amdgcn_readfirstlane(f);
        __syncthreads();     // (the flags are rewritten by the next call)
        return f;
    };
    auto snapshot = [&](int n, uint32_t newly) {   // frames that stop now: hard(lam) of the lane's own columns
        if ((newly >> my_slot) & 1u) {
            LDPC_COLD_PATH();
            res.converge_at(n);
            static_for<0, Plan::NBC>([&](auto bcc) {
                constexpr int bc = decltype(bcc)::value;
                if constexpr ((bc % Plan::NP) == P) res.bits.set(bc / Plan::NP, lds_ld<float>(lds, p4 + (bc * V * ES)) > 0.0f);
            });
            if (A.final_lam) {
                const Where w(p4, A.batch);
                static_for<0, Plan::NBC>([&](auto bcc) {
                    constexpr int bc = decltype(bcc)::value;
                    if constexpr ((bc % Plan::NP) == P) A.final_lam[w.fN + bc * SZ + w.r0] = (double)lds_ld<float>(lds, p4 + (bc * V * ES));
                });
            }
        }
    };
    auto trace_row = [&](int n) {
        if (A.trace) {
            LDPC_COLD_PATH();
            const Where w(p4, A.batch);
            if (w.valid && !((done >> w.sub) & 1u))
                static_for<0, Plan::NBC>([&](auto bcc) {
                    constexpr int bc = decltype(bcc)::value;
                    if constexpr ((bc % Plan::NP) == P)
                        A.trace[((size_t)w.frame * (A.max_iters + 1) + n) * N + bc * SZ + w.r0] = (double)lds_ld<float>(lds, p4 + (bc * V * ES));
                });
        }
    };

    {   // ---- before the first sweep: the syndrome of the channel's hard decisions
        bool odd = false;
        static_for<0, Plan::NBR>([&](auto brc) {
            constexpr int br = decltype(brc)::value;
            if constexpr (S::owner_br(br) == P) {
                StatRow<float, SZ, T, Plan::ebeg(br)> row;
                odd |= layer_row<Plan::deg(br), false, true>(lds, row, p4, vmask, (float *)nullptr);
            }
        });
        const uint32_t bad = frames_with(odd);
        trace_row(0);
        const uint32_t newly = ~bad & ~done & FULL;
        snapshot(0, newly);
        done |= newly;
        if (A.trace || newly != 0u) __syncthreads();   // the rows copied / snapshotted above are rewritten by layer 0
    }
    // RS: one layer, the own half of the row
    auto half_layer = [&](auto brc, auto firstc, bool &any) {
        constexpr int br = decltype(brc)::value;
        constexpr bool FIRST = decltype(firstc)::value;
        constexpr int D = Plan::deg(br), DH = H::cnt(br, P), K0 = H::k0(br, P), ms0 = H::slot0(br, P);
        if constexpr (FIRST) LDPC_COLD_PATH();
        StatRow<float, SZ, T, Plan::ebeg(br) + K0> row;
        uint32_t pp = p4;
        asm volatile("" : "+v"(pp));
        float l[DH], t[DH];
        uint32_t adr[DH];
        static_for<0, DH>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            adr[k] = qc_wrap(pp + row.lo(k), vmask);
            l[k] = lds_ld<float>(lds + row.hi(k), adr[k]);
        });
        bool par = false;
        uint32_t X = 0;
        float m1 = INFINITY, m2 = INFINITY;
        static_for<0, DH>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            par ^= (l[k] > 0.0f);
            t[k] = FIRST ? l[k] - 0.0f : l[k] - msg[ms0 + k];
            X ^= __float_as_uint(t[k]);
            const float a = fabsf(t[k]);
            m2 = __builtin_amdgcn_fmed3f(m1, m2, a);
            m1 = fminf(m1, a);
        });
        // sign word: bit 31 = parity of the signs of t, bit 0 = parity of the hard decisions
        const uint32_t xw = (X & 0x80000000u) | (par ? 1u : 0u);
        lds_st<float>(lds + EX0 + P * EXG, pp, m1);
        lds_st<float>(lds + EX0 + EXW + P * EXG, pp, m2);
        lds_st<uint32_t>(lds + EX0 + 2 * EXW + P * EXG, pp, xw);
        __syncthreads();
        const float q1 = lds_ld<float>(lds + EX0 + (1 - P) * EXG, pp);
        const float q2 = lds_ld<float>(lds + EX0 + EXW + (1 - P) * EXG, pp);
        const uint32_t xq = lds_ld<uint32_t>(lds + EX0 + 2 * EXW + (1 - P) * EXG, pp);
        const float M1 = fminf(m1, q1);
        const float M2 = fminf(fmaxf(m1, q1), fminf(m2, q2));     // the second smallest of the row
        const uint32_t xa = xw ^ xq;
        const uint32_t flipbit = (xa ^ ((D & 1) ? 0x80000000u : 0u)) & 0x80000000u;
        const uint32_t c1 = __float_as_uint(0.75f * M1) ^ flipbit;
        const uint32_t c2 = __float_as_uint(0.75f * M2) ^ flipbit;
        bool flip = false;
        static_for<0, DH>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const uint32_t c = (fabsf(t[k]) == M1) ? c2 : c1;
            const float nm = __uint_as_float(__builtin_amdgcn_bitop3_b32(c, __float_as_uint(t[k]), 0x80000000u, 0x78));
            msg[ms0 + k] = nm;
            const float nw = t[k] + nm;
            flip |= (nw > 0.0f) != (l[k] > 0.0f);
            lds_st<float>(lds + row.hi(k), adr[k], nw);
        });
        any |= ((xa & 1u) != 0u) || flip;
        __syncthreads();
    };
    for (int n = 1; done != FULL && n <= A.max_iters; n++) {
        bool any = false;
        // ---- one sweep: layers in order.  RS: both groups work on every layer (two barriers per layer); else the group that owns
        // the block row works, everybody meets at the barrier
        static_for<0, Plan::NBR>([&](auto brc) {
            constexpr int br = decltype(brc)::value;
            if constexpr (RS && H::split(br)) {
                if (n == 1) half_layer(brc, std::true_type{}, any);
                else half_layer(brc, std::false_type{}, any);
                return;
            }
            if constexpr (!(RS && H::split(br)) && S::owner_br(br) == P) {
                constexpr int D = Plan::deg(br), ms0 = RS ? H::slot0(br, P) : S::slot(Plan::ebeg(br));
                StatRow<float, SZ, T, Plan::ebeg(br)> row;
                if (n == 1) any |= layer_row<D, true, false>(lds, row, p4, vmask, &msg[ms0]);
                else any |= layer_row<D, false, false>(lds, row, p4, vmask, &msg[ms0]);
            }
            if constexpr (!(RS && H::split(br))) __syncthreads();
        });
        const uint32_t moved = frames_with(any);
        trace_row(n);
        uint32_t newly = ~moved & ~done & FULL;
        if (newly != 0u) {   // (workgroup-uniform, once per frame) LLRs that left the float range: failed, not "converged" (ldpc_math.h)
            LDPC_COLD_PATH();
            bool nf = false;
            if ((newly >> my_slot) & 1u)
                static_for<0, Plan::NBC>([&](auto bcc) {
                    constexpr int bc = decltype(bcc)::value;
                    if constexpr ((bc % Plan::NP) == P) nf |= not_finite(lds_ld<float>(lds, p4 + (bc * V * ES)));
                });
            const uint32_t veto = frames_with(nf) & newly;
            done |= veto;        // stops here as a failure: `res` keeps the channel's hard decisions and a clear flag
            newly &= ~veto;
        }
        snapshot(n, newly);
        done |= newly;
        if (A.trace || (newly != 0u && done != FULL)) __syncthreads();
    }

    const Where w(p4, A.batch);
    if (!w.valid) return;
    const bool converged = res.converged();
    static_for<0, Plan::NBC>([&](auto bcc) {   // hard(lam) of a frame that stopped by the rule, the channel's decisions otherwise
        constexpr int bc = decltype(bcc)::value;
        if constexpr ((bc % Plan::NP) == P) A.bits[w.fN + bc * SZ + w.r0] = res.bits.get(bc / Plan::NP);
    });
    if (!converged && A.final_lam) {
        LDPC_COLD_PATH();
        with_llr_format(A.llr_fmt, [&](auto fc) {
            constexpr int FMT = decltype(fc)::value;
            static_for<0, Plan::NBC>([&](auto bcc) {
                constexpr int bc = decltype(bcc)::value;
                if constexpr ((bc % Plan::NP) == P) {
                    const size_t gi = w.fN + bc * SZ + w.r0;
                    A.final_lam[gi] = (double)maybe_round_f16<float>(load_llr_as<float, FMT>(A.llr, gi), A.llr_round16);
                }
            });
        });
    }
    if (w.r0 == 0 && P == 0) {
        if (A.iters) A.iters[w.frame] = converged ? res.turn() : A.max_iters;
        if (A.conv) A.conv[w.frame] = converged ? 1 : 0;
    }
}

template <class Plan, int SZ, class T>
__device__ __forceinline__ void kernel_body(const FusedArgs &A) {
    using G = SplitGeom<Plan, SZ>;
    constexpr bool RS = LAY_ROW_SPLIT && Halves<Plan>::ok();
    __shared__ __attribute__((aligned(16))) char lds[(Plan::NBC * G::V * 4 + 15) / 16 * 16 + 4 * G::NW + (RS ? 16 + 3 * 2 * G::VT * 4 : 0)];
    const uint32_t tid = threadIdx.x;
    const uint32_t group = __builtin_amdgcn_readfirstlane(tid / G::VT);
    static_for<0, Plan::NP>([&](auto pc) {
        constexpr int P = decltype(pc)::value;
        if (group == (uint32_t)P) body<Plan, SZ, T, P, RS>(A, lds, tid);
    });
}
}  // namespace lay

// ---------------------------------------------------------------------------------------------------------------------
// The same layered kernel in PACKED FP16, two frames per lane: the two r03 kernels composed (LDPC_F16PK + LDPC_SCHED_LAYERED).
// State as in fused_pk16_body.h: L = -lam (so that hard(lam) is the sign bit of L exactly) and u = msg / (3/4).  A row of a layer:
//   tN_k = fma(u_k, 3/4, L_k)                      ( = -(lam - msg); the first sweep has u = 0: tN = L )
//   odd bits ^= L_k                                 ( parity of the hard decisions the row saw: bits 15 / 31 )
//   u'_k     = leave-one-out minimum with sign 1 ^ X ^ sign(tN_k)                (pk::loo_min: no rounding)
//   L'_k     = fma(u'_k, -3/4, tN_k)                ( = -(t + msg') : ONE rounding )
//   flip bits |= L'_k ^ L_k                         ( a hard decision changed: no compare needed )
// Specification: oracle/emulate_f16.py decode_minsum_pk16_layered (bit for bit: tests/test_pk16_gpu.py).
namespace laypk {
using namespace pk;

template <int D>
__device__ __forceinline__ void loo_min_out(const uint32_t (&tn)[D], const uint32_t (&a)[D], uint32_t xf, uint32_t (&out)[D]) {
    uint32_t tmp[D];
#pragma unroll
    for (int k = 0; k < D; k++) tmp[k] = tn[k];
    loo_min_update<D>(tmp, a, xf);     // in place on the copy: tmp[k] = u'_k
#pragma unroll
    for (int k = 0; k < D; k++) out[k] = tmp[k];
}

template <int D, bool FIRST, bool SYNDROME_ONLY, class Row>
__device__ __forceinline__ uint32_t layer_row(char *lds, Row tabrow, uint32_t p4, uint32_t vmask, uint32_t *u) {
    asm volatile("" : "+v"(p4));
    if constexpr (SYNDROME_ONLY || FIRST) LDPC_COLD_PATH();
    uint32_t l[D], adr[D];
    static_for<0, D>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        adr[k] = qc_wrap(p4 + tabrow.lo(k), vmask);
        l[k] = lds_ld<uint32_t>(lds + tabrow.hi(k), adr[k]);
    });
    uint32_t par = 0;
#pragma unroll
    for (int k = 0; k < D; k++) par ^= l[k];
    if constexpr (SYNDROME_ONLY) return par;
    uint32_t X = 0, tn[D], a[D], un[D];
#pragma unroll
    for (int k = 0; k < D; k++) {
        tn[k] = FIRST ? l[k] : fma_k(u[k], K75, l[k]);   // fma(+0, 3/4, L) = L
        X ^= tn[k];
        a[k] = tn[k] & ABS;
    }
    loo_min_out<D>(tn, a, ~X & SGN, un);
    uint32_t fl = 0;
    static_for<0, D>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        u[k] = un[k];
        const uint32_t ln = fma_k(un[k], KN75, tn[k]);
        fl |= ln ^ l[k];
        lds_st<uint32_t>(lds + tabrow.hi(k), adr[k], ln);
    });
    return par | fl;     // bits 15 / 31: the row's parity was odd or a hard decision flipped, low / high frame
}

// leave-one-out minimum over the own D edges of a row whose OTHER edges (the partner thread's half, lay::Halves) have the minimum
// `seed` (saturated at U16_MAX already): pk::loo_min_update with the running prefix started at `seed`.  D >= 1.
template <int D>
__device__ __forceinline__ void loo_min_seeded(const uint32_t (&tn)[D], const uint32_t (&a)[D], const uint32_t (&suf)[(D + 1) / 2], uint32_t xf, uint32_t seed, uint32_t (&out)[D]) {
    auto put = [&](int k, uint32_t loo) { out[k] = xor3(loo, tn[k] ^ a[k], xf); };
    constexpr int NB = (D + 1) / 2;
    uint32_t pre = seed;
    static_for<0, NB>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr bool has_suf = j < NB - 1, has_partner = 2 * j + 1 < D;
        auto others = [&](uint32_t partner) -> uint32_t {
            if constexpr (has_suf && has_partner) return min3(pre, suf[j], partner);
            else if constexpr (has_suf) return min2(pre, suf[j]);
            else if constexpr (has_partner) return min2(pre, partner);
            else return pre;
        };
        if constexpr (has_partner) {
            const uint32_t o0 = others(a[2 * j + 1]), o1 = others(a[2 * j]);
            if constexpr (has_suf) pre = min3(pre, a[2 * j], a[2 * j + 1]);
            put(2 * j, o0); put(2 * j + 1, o1);
        } else {
            put(2 * j, others(0u));
        }
    });
}
// suf[j] = min over the pairs after pair j (suf[NB - 1] unused); -> the minimum of all D magnitudes
template <int D>
__device__ __forceinline__ uint32_t suffix_minima(const uint32_t (&a)[D], uint32_t (&suf)[(D + 1) / 2]) {
    constexpr int NB = (D + 1) / 2;
    auto pair_min = [&](auto jc) -> uint32_t {
        constexpr int j = decltype(jc)::value;
        if constexpr (2 * j + 1 < D) return min2(a[2 * j], a[2 * j + 1]); else return a[2 * j];
    };
    suf[NB - 1] = INF2;
    static_rfor<0, NB - 1>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const uint32_t b = pair_min(std::integral_constant<int, j + 1>{});
        if constexpr (j == NB - 2) suf[j] = b; else suf[j] = min2(suf[j + 1], b);
    });
    const uint32_t first = pair_min(std::integral_constant<int, 0>{});
    if constexpr (NB >= 2) return min2(first, suf[0]); else return first;
}

template <class Plan, int SZ, class T, int P, bool RS>
__device__ __forceinline__ void body(const FusedArgs &A, char *lds, const uint32_t tid) {
    using S = Split<Plan, T>;
    using H = lay::Halves<Plan>;
    constexpr int CPW = QcGeom<SZ>::CPW, V = QcGeom<SZ>::V, VT = QcGeom<SZ>::VT;
    constexpr int N = Plan::NBC * SZ, THREADS = Plan::NP * VT, NW = THREADS / 64;
    constexpr uint32_t ES = 4, vmask = V * ES - 1;
    constexpr int LAM_BYTES = (Plan::NBC * V * (int)ES + 15) / 16 * 16;
    constexpr int NBCP = (Plan::NBC + Plan::NP - 1) / Plan::NP;
    static_assert(CPW <= 8, "the done mask holds 2 * CPW frames");
    if constexpr (VT != V) { if ((tid % VT) >= (uint32_t)V) return; }
    const uint32_t p4 = (tid % VT) * ES;
    struct Where {
        uint32_t sub, r0; long long frame0; bool valid[2]; size_t fN[2];
        __device__ __forceinline__ Where(uint32_t p, int batch) {
            asm volatile("" : "+v"(p));
            const uint32_t lane = p / ES;
            sub = lane % CPW;
            r0 = lane / CPW;
            frame0 = ((long long)blockIdx.x * CPW + sub) * 2;
#pragma unroll
            for (int h = 0; h < 2; h++) { valid[h] = frame0 + h < batch; fN[h] = (size_t)(valid[h] ? frame0 + h : 0) * N; }
        }
    };
    constexpr int NU = RS ? H::NMSG : S::NMSG;
    constexpr uint32_t EX0 = LAM_BYTES + 4 * NW + 12, EXW = 2 * VT * ES, EXG = VT * ES;   // RS: exchange area [word 0..1][group][VT]
    uint32_t u[NU];
#pragma unroll
    for (int i = 0; i < NU; i++) u[i] = 0u;
    typename SplitResult<NBCP>::Bits obits[2];
    {
        const Where w(p4, A.batch);
        with_llr_format(A.llr_fmt, [&](auto fc) {
            constexpr int FMT = decltype(fc)::value;
            float x[NBCP][2];
            static_for<0, Plan::NBC>([&](auto bcc) {
                constexpr int bc = decltype(bcc)::value;
                if constexpr ((bc % Plan::NP) == P) {
#pragma unroll
                    for (int h = 0; h < 2; h++) x[bc / Plan::NP][h] = load_llr_as<float, FMT>(A.llr, w.fN[h] + bc * SZ + w.r0);
                }
            });
            static_for<0, Plan::NBC>([&](auto bcc) {
                constexpr int bc = decltype(bcc)::value;
                if constexpr ((bc % Plan::NP) == P) {
                    uint32_t packed = 0;
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const uint32_t b = neg_llr16(x[bc / Plan::NP][h]);
                        obits[h].set(bc / Plan::NP, (b >> 15) & 1u);
                        packed |= b << (16 * h);
                    }
                    lds_st<uint32_t>(lds, p4 + (bc * V * ES), packed);
                }
            });
        });
    }
    __syncthreads();

    volatile uint32_t *flags = reinterpret_cast<volatile uint32_t *>(lds + LAM_BYTES);
    constexpr uint32_t FULL = (1u << (2 * CPW)) - 1;
    uint32_t done = 0;     // bit 2s + h
#pragma unroll
    for (int s2 = 0; s2 < 2 * CPW; s2++) done |= (((long long)blockIdx.x * CPW + s2 / 2) * 2 + (s2 & 1) < A.batch) ? 0u : (1u << s2);
    SplitResult<NBCP> res[2];
    res[0].bits = obits[0]; res[1].bits = obits[1];
    const uint32_t my_slot = (p4 / ES) % CPW;

    auto frames_with = [&](uint32_t word) -> uint32_t {    // bits 15 / 31 of `word` = the lane's flag for its low / high frame
        uint32_t wbits = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const unsigned long long ub = __ballot((word >> (15 + 16 * h)) & 1u);
#pragma unroll
            for (int s2 = 0; s2 < CPW; s2++) {
                unsigned long long mk = 0;
                for (int i = 0; i < 64; i += CPW) mk |= 1ull << i;
                wbits |= ((ub & (mk << s2)) != 0ull) ? (1u << (2 * s2 + h)) : 0u;
            }
        }
        if ((tid & 63) == 0) flags[tid >> 6] = wbits;
        __syncthreads();
        uint32_t f = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) f |= flags[w];
        f = __builtin_amdgcn_readfirstlane(f);
        __syncthreads();
        return f;
    };
    auto snapshot = [&](int n, uint32_t newly) {
        if ((newly >> (2 * my_slot)) & 3u) {
            LDPC_COLD_PATH();
#pragma unroll
            for (int h = 0; h < 2; h++)
                if ((newly >> (2 * my_slot + h)) & 1u) {
                    res[h].converge_at(n);
                    static_for<0, Plan::NBC>([&](auto bcc) {
                        constexpr int bc = decltype(bcc)::value;
                        if constexpr ((bc % Plan::NP) == P) res[h].bits.set(bc / Plan::NP, (lds_ld<uint32_t>(lds, p4 + (bc * V * ES)) >> (15 + 16 * h)) & 1u);
                    });
                    if (A.final_lam) {
                        const Where w(p4, A.batch);
                        static_for<0, Plan::NBC>([&](auto bcc) {
                            constexpr int bc = decltype(bcc)::value;
                            if constexpr ((bc % Plan::NP) == P) A.final_lam[w.fN[h] + bc * SZ + w.r0] = lam_of(lds_ld<uint32_t>(lds, p4 + (bc * V * ES)), h);
                        });
                    }
                }
        }
    };
    auto trace_row = [&](int n) {
        if (A.trace) {
            LDPC_COLD_PATH();
            const Where w(p4, A.batch);
#pragma unroll
            for (int h = 0; h < 2; h++)
                if (w.valid[h] && !((done >> (2 * w.sub + h)) & 1u))
                    static_for<0, Plan::NBC>([&](auto bcc) {
                        constexpr int bc = decltype(bcc)::value;
                        if constexpr ((bc % Plan::NP) == P)
                            A.trace[((size_t)(w.frame0 + h) * (A.max_iters + 1) + n) * N + bc * SZ + w.r0] = lam_of(lds_ld<uint32_t>(lds, p4 + (bc * V * ES)), h);
                    });
        }
    };

    {
        uint32_t odd = 0;
        static_for<0, Plan::NBR>([&](auto brc) {
            constexpr int br = decltype(brc)::value;
            if constexpr (S::owner_br(br) == P) {
                StatRow<float, SZ, T, Plan::ebeg(br)> row;
                odd |= layer_row<Plan::deg(br), false, true>(lds, row, p4, vmask, (uint32_t *)nullptr);
            }
        });
        const uint32_t bad = frames_with(odd);
        trace_row(0);
        const uint32_t newly = ~bad & ~done & FULL;
        snapshot(0, newly);
        done |= newly;
        if (A.trace || newly != 0u) __syncthreads();
    }
    // RS: one layer, the own half of the row (lay::Halves has the scheme; here the partner's half enters the leave-one-out
    // minimum as ONE value -- the minimum of its magnitudes -- so two words are exchanged: that minimum, and sign / parity bits)
    auto half_layer = [&](auto brc, auto firstc, uint32_t &any) {
        constexpr int br = decltype(brc)::value;
        constexpr bool FIRST = decltype(firstc)::value;
        constexpr int DH = H::cnt(br, P), K0 = H::k0(br, P), ms0 = H::slot0(br, P);
        if constexpr (FIRST) LDPC_COLD_PATH();
        StatRow<float, SZ, T, Plan::ebeg(br) + K0> row;
        uint32_t pp = p4;
        asm volatile("" : "+v"(pp));
        uint32_t l[DH], adr[DH], tn[DH], a[DH], suf[(DH + 1) / 2], un[DH];
        static_for<0, DH>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            adr[k] = qc_wrap(pp + row.lo(k), vmask);
            l[k] = lds_ld<uint32_t>(lds + row.hi(k), adr[k]);
        });
        uint32_t par = 0, X = 0;
#pragma unroll
        for (int k = 0; k < DH; k++) {
            par ^= l[k];
            tn[k] = FIRST ? l[k] : fma_k(u[ms0 + k], K75, l[k]);
            X ^= tn[k];
            a[k] = tn[k] & ABS;
        }
        const uint32_t own_min = suffix_minima<DH>(a, suf);
        const uint32_t xw = (X & SGN) | ((par & SGN) >> 1);      // bits 15 / 31: sign parity of tN; bits 14 / 30: parity of the hard decisions
        lds_st<uint32_t>(lds + EX0 + P * EXG, pp, own_min);
        lds_st<uint32_t>(lds + EX0 + EXW + P * EXG, pp, xw);
        __syncthreads();
        const uint32_t qmin = lds_ld<uint32_t>(lds + EX0 + (1 - P) * EXG, pp);
        const uint32_t xa = xw ^ lds_ld<uint32_t>(lds + EX0 + EXW + (1 - P) * EXG, pp);
        loo_min_seeded<DH>(tn, a, suf, ~xa & SGN, min2_umax(qmin), un);
        uint32_t fl = 0;
        static_for<0, DH>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            u[ms0 + k] = un[k];
            const uint32_t ln = fma_k(un[k], KN75, tn[k]);
            fl |= ln ^ l[k];
            lds_st<uint32_t>(lds + row.hi(k), adr[k], ln);
        });
        any |= ((xa << 1) & SGN) | fl;
        __syncthreads();
    };
    for (int n = 1; done != FULL && n <= A.max_iters; n++) {
        uint32_t any = 0;
        static_for<0, Plan::NBR>([&](auto brc) {
            constexpr int br = decltype(brc)::value;
            if constexpr (RS && H::split(br)) {
                if (n == 1) half_layer(brc, std::true_type{}, any);
                else half_layer(brc, std::false_type{}, any);
                return;
            }
            if constexpr (!(RS && H::split(br)) && S::owner_br(br) == P) {
                constexpr int D = Plan::deg(br), ms0 = RS ? H::slot0(br, P) : S::slot(Plan::ebeg(br));
                StatRow<float, SZ, T, Plan::ebeg(br)> row;
                if (n == 1) any |= layer_row<D, true, false>(lds, row, p4, vmask, &u[ms0]);
                else any |= layer_row<D, false, false>(lds, row, p4, vmask, &u[ms0]);
            }
            if constexpr (!(RS && H::split(br))) __syncthreads();
        });
        const uint32_t moved = frames_with(any);
        trace_row(n);
        const uint32_t newly = ~moved & ~done & FULL;
        snapshot(n, newly);
        done |= newly;
        if (A.trace || (newly != 0u && done != FULL)) __syncthreads();
    }

    const Where w(p4, A.batch);
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (!w.valid[h]) continue;
        const bool converged = res[h].converged();
        static_for<0, Plan::NBC>([&](auto bcc) {
            constexpr int bc = decltype(bcc)::value;
            if constexpr ((bc % Plan::NP) == P) A.bits[w.fN[h] + bc * SZ + w.r0] = res[h].bits.get(bc / Plan::NP);
        });
        if (!converged && A.final_lam) {
            LDPC_COLD_PATH();
            with_llr_format(A.llr_fmt, [&](auto fc) {
                constexpr int FMT = decltype(fc)::value;
                static_for<0, Plan::NBC>([&](auto bcc) {
                    constexpr int bc = decltype(bcc)::value;
                    if constexpr ((bc % Plan::NP) == P) {
                        const size_t gi = w.fN[h] + bc * SZ + w.r0;
                        A.final_lam[gi] = lam_of(neg_llr16(load_llr_as<float, FMT>(A.llr, gi)), 0);
                    }
                });
            });
        }
        if (w.r0 == 0 && P == 0) {
            if (A.iters) A.iters[w.frame0 + h] = converged ? res[h].turn() : A.max_iters;
            if (A.conv) A.conv[w.frame0 + h] = converged ? 1 : 0;
        }
    }
}

template <class Plan, int SZ, class T>
__device__ __forceinline__ void kernel_body(const FusedArgs &A) {
    using G = SplitGeom<Plan, SZ>;
    constexpr bool RS = LAY_ROW_SPLIT && lay::Halves<Plan>::ok();
    __shared__ __attribute__((aligned(16))) char lds[(Plan::NBC * G::V * 4 + 15) / 16 * 16 + 4 * G::NW + (RS ? 16 + 2 * 2 * G::VT * 4 : 0)];
    const uint32_t tid = threadIdx.x;
    const uint32_t group = __builtin_amdgcn_readfirstlane(tid / G::VT);
    static_for<0, Plan::NP>([&](auto pc) {
        constexpr int P = decltype(pc)::value;
        if (group == (uint32_t)P) body<Plan, SZ, T, P, RS>(A, lds, tid);
    });
}
}  // namespace laypk
}  // namespace ldpc
