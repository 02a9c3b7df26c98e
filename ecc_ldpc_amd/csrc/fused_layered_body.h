// fused_layered_body.h -- the row-layered schedule ON-CHIP for the codes the split kernel takes (EXTENSION: the reference has
// no layered decoder; specification = oracle/ldpc_oracle.c oracle_decode_layered, HBM implementation = layered_qc.hip).
//
// Same workgroup as the flooding split kernel (fused_split_body.h): lam in LDS, a frame's block rows dealt to NP wave groups,
// thread (group, r) = row r of every circulant of its group's block rows, its messages in registers.  A LAYER = a block row;
// layers run in order, so at any moment ONE group of the workgroup works (its 128 rows read the lam cells of their columns,
// apply the check rule, and write lam back -- the rows of a block row touch distinct columns) while the other waits at the
// barrier that ends the layer; the SIMDs are kept busy by the other workgroups of the CU.  No column "rounds", no second
// pass over LDS, no channel-LLR registers: a sweep is NBR x {gather, rule, scatter, barrier}.
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
    if constexpr (SYNDROME_ONLY) LDPC_COLD_PATH();
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

template <class Plan, int SZ, class T, int P>
__device__ __forceinline__ void body(const FusedArgs &A, char *lds, const uint32_t tid) {
    using S = Split<Plan, T>;
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
    float msg[S::NMSG];
#pragma unroll
    for (int i = 0; i < S::NMSG; i++) msg[i] = 0.0f;
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
    for (int n = 1; done != FULL && n <= A.max_iters; n++) {
        bool any = false;
        // ---- one sweep: layers in order; the group that owns the block row works, everybody meets at the barrier
        static_for<0, Plan::NBR>([&](auto brc) {
            constexpr int br = decltype(brc)::value;
            if constexpr (S::owner_br(br) == P) {
                constexpr int D = Plan::deg(br), ms0 = S::slot(Plan::ebeg(br));
                StatRow<float, SZ, T, Plan::ebeg(br)> row;
                if (n == 1) any |= layer_row<D, true, false>(lds, row, p4, vmask, &msg[ms0]);
                else any |= layer_row<D, false, false>(lds, row, p4, vmask, &msg[ms0]);
            }
            __syncthreads();
        });
        const uint32_t moved = frames_with(any);
        trace_row(n);
        const uint32_t newly = ~moved & ~done & FULL;
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
    __shared__ __attribute__((aligned(16))) char lds[(Plan::NBC * G::V * 4 + 15) / 16 * 16 + 4 * G::NW];
    const uint32_t tid = threadIdx.x;
    const uint32_t group = __builtin_amdgcn_readfirstlane(tid / G::VT);
    static_for<0, Plan::NP>([&](auto pc) {
        constexpr int P = decltype(pc)::value;
        if (group == (uint32_t)P) body<Plan, SZ, T, P>(A, lds, tid);
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
    if constexpr (SYNDROME_ONLY) LDPC_COLD_PATH();
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

template <class Plan, int SZ, class T, int P>
__device__ __forceinline__ void body(const FusedArgs &A, char *lds, const uint32_t tid) {
    using S = Split<Plan, T>;
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
    uint32_t u[S::NMSG];
#pragma unroll
    for (int i = 0; i < S::NMSG; i++) u[i] = 0u;
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
    for (int n = 1; done != FULL && n <= A.max_iters; n++) {
        uint32_t any = 0;
        static_for<0, Plan::NBR>([&](auto brc) {
            constexpr int br = decltype(brc)::value;
            if constexpr (S::owner_br(br) == P) {
                constexpr int D = Plan::deg(br), ms0 = S::slot(Plan::ebeg(br));
                StatRow<float, SZ, T, Plan::ebeg(br)> row;
                if (n == 1) any |= layer_row<D, true, false>(lds, row, p4, vmask, &u[ms0]);
                else any |= layer_row<D, false, false>(lds, row, p4, vmask, &u[ms0]);
            }
            __syncthreads();
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
    __shared__ __attribute__((aligned(16))) char lds[(Plan::NBC * G::V * 4 + 15) / 16 * 16 + 4 * G::NW];
    const uint32_t tid = threadIdx.x;
    const uint32_t group = __builtin_amdgcn_readfirstlane(tid / G::VT);
    static_for<0, Plan::NP>([&](auto pc) {
        constexpr int P = decltype(pc)::value;
        if (group == (uint32_t)P) body<Plan, SZ, T, P>(A, lds, tid);
    });
}
}  // namespace laypk
}  // namespace ldpc
