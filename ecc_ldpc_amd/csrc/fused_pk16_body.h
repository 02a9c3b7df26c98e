// fused_pk16_body.h -- the four-wave split kernel (fused_split_body.h) with TWO FRAMES PER LANE in packed fp16.
//
// BASELINE configs[3] ("min-sum fp16 LLRs").  The f32 split kernel is bound by VALU issue (12.3 instructions per edge and
// turn, 37 clk); gfx950 issues a packed 16-bit instruction (two results per lane) in the time of one f32 min/max
// (4.2 clk, profiles/r03_microbench_pk16.txt).  So a lane here owns row r of the SAME circulants as in the split kernel, for a
// PAIR of frames: frame 2w in the low half and frame 2w+1 in the high half of every register and of every LDS word.  Graph
// tables, addresses, LDS instructions, barriers and loop control are shared by the two frames; the arithmetic is one packed
// instruction for both.  LDS per pair of frames = what the f32 kernel needs per frame.
//
// Arithmetic (the SPECIFICATION; oracle/emulate_f16.py decode_minsum_pk16 reproduces it bit for bit).  The reference has no
// fp16 decoder: this is the loop of Reference/Min.hs:54-104 with every quantity in IEEE binary16 and the 3/4 of Min.hs:78
// applied where a message is USED, inside a fused multiply-add, instead of where it is made:
//   state   L_j = -lam_j (column LLR, negated) and u_e = ne'_e / (3/4) (check->variable message without its 3/4); u = +0 at start,
//           L = -(channel LLR saturated at +-16384 = LLR16_MAX, rounded to fp16), a zero LLR as +0
//   range   message magnitudes saturate at 2048 = U16_MAX (folded into the leave-one-out minimum: free for rows of degree 3 and 4,
//           one instruction per row above that).  With column degree <= 30 (the host refuses more) every quantity stays finite:
//           |L| <= 16384 + 30 * 1536 (+ roundings) and |tN| <= |L| + 1536 < 65504.  Without it a frame that diverges reaches +-inf,
//           inf - inf = NaN, and a NaN has sign bit 0: the frame would "converge" on the all-zero word (seen on a (3,6)-regular
//           code with a saturated input, tests/test_jit_kinds_gpu.py; the f32 kernels saturate lam for the same reason)
//   hard    hard(lam_j) = lam_j > 0 = sign bit of L_j   (L is never -0: sums that cancel give +0 in round-to-nearest, and -0
//           enters nowhere -- which is why the NEGATED LLR is what is stored: `hard 0 = False` needs no compare)
//   check   tN_k = fma(u_k, 3/4, L_k) = -(lam_k - ne_k);  u'_k = -prod_{j/=k} sgn(tN_j) * min_{j/=k} |tN_j|   (no rounding at all)
//   column  L'_c = fma(u'_e, -3/4, ...fma(u'_e', -3/4, L0_c)), edges of the column in descending row order (Min.hs:101 foldr)
//   stop    as Min.hs:75-76: syndrome of hard(lam) zero -> hard(lam), after `turns` updates -> hard(channel LLR)
// Each fma rounds ONCE (the reference rounds the product and the sum): the fp16 trajectory is slightly closer to the real-valued
// min-sum than a literal fp16 transcription would be.  BER against the f32 decoder: DESIGN.md.
// The two frames of a lane stop independently (a finished frame keeps its answer in a snapshot register and its half keeps
// computing until the partner is done, like the frames of a workgroup in the split kernel for sz < 64).
// Tried in r03 and dropped: every block column of L kept TWICE in LDS, V positions apart, so that a check row's gather at
// p + rotation never wraps (address = the lane's base register + an immediate: 945 -> 855 VALU instructions per wave-turn, LDS
// 45 KB per workgroup, column updates storing both copies).  jpl.4096, 65 536 frames, 2 dB: 13.2 ms per launch against 11.85 --
// the doubled LDS write traffic (both copies on the same bank) costs more than the 10 % of VALU issue it saves.
// Also tried in r03 and dropped (compile-time evidence only): a register diet for 4 waves per SIMD -- heavy rows gathered nine edges
// at a time, tN written over u at once, |tN| recomputed where the leave-one-out minimum uses it instead of being held: 152 VGPRs
// wanted (162 today), at 128 still 12-24 spilled registers per turn, and 1 105 VALU per wave-turn instead of 949 (+13 % issue cost).
// The layered kernel's step from 3 to 4 waves bought 10 % with NO extra instructions; this one would start 13 % behind.
#pragma once
#include "fused_split_body.h"

namespace ldpc {
namespace pk {
constexpr uint32_t K75 = 0x3a003a00u, KN75 = 0xba00ba00u, ABS = 0x7fff7fffu, SGN = 0x80008000u, INF2 = 0x7c007c00u;
constexpr uint32_t UMAX2 = 0x68006800u;    // U16_MAX = 2048 in both halves
constexpr float LLR16_MAX = 16384.0f;
constexpr int PK16_MAX_COLUMN_DEGREE = 30;
// (non-volatile asm: pure functions of their inputs, free to be scheduled; the compiler's own elementwise min/max on half2 adds
//  a canonicalising v_pk_max_f16 x, x, x per operand)
__device__ __forceinline__ uint32_t fma_k(uint32_t a, uint32_t k, uint32_t c) {   // a * k + c, k wave-uniform
    uint32_t r;
    asm("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t min2(uint32_t a, uint32_t b) {   // magnitudes: non-negative fp16 order like their bit patterns
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t min3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t min2_umax(uint32_t a) {               // min(a, U16_MAX)
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(UMAX2));
    return r;
}
__device__ __forceinline__ uint32_t min3_umax(uint32_t a, uint32_t b) {   // min(a, b, U16_MAX)
    uint32_t r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(UMAX2));
    return r;
}
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

// Leave-one-out minimum of D magnitudes a[k] = |tN_k|, each result turned into the new message at once:
//   u'_k = (min over j /= k of a_j) with sign 1 ^ X ^ sign(tN_k)  =  loo_k ^ (tN_k ^ a_k) ^ xf      (tN ^ |tN| = the sign bits of tN)
// Pairs first, suffix minima over the pairs, then a running prefix and one 3-input minimum per element (its partner, everything
// before its pair, everything after): 2.3 packed instructions per element at D = 18 against 4 for the running (min1, min2)
// form plus a select.  In place: u[k] holds tN_k on entry and u'_k on return; a[] is consumed.
template <int D>
__device__ __forceinline__ void loo_min_update(uint32_t *u, const uint32_t (&a)[D], uint32_t xf) {
    static_assert(D >= 2, "min-sum needs degree >= 2");
    auto put = [&](int k, uint32_t loo) { u[k] = xor3(loo, u[k] ^ a[k], xf); };
    if constexpr (D == 2) { put(0, min2_umax(a[1])); put(1, min2_umax(a[0])); }
    else if constexpr (D == 3) {
        const uint32_t m12 = min3_umax(a[1], a[2]), m02 = min3_umax(a[0], a[2]), m01 = min3_umax(a[0], a[1]);
        put(0, m12); put(1, m02); put(2, m01);
    } else {
        constexpr int NB = (D + 1) / 2;
        // U16_MAX rides in the free operand of the first pair's and (two pairs only) the last pair's minimum; with three or more
        // pairs it is folded into the running prefix once, which every later element takes
        constexpr bool FOLD = NB >= 3;
        auto pair_min = [&](auto jc) -> uint32_t {
            constexpr int j = decltype(jc)::value;
            if constexpr (2 * j + 1 < D) return min2(a[2 * j], a[2 * j + 1]); else return a[2 * j];
        };
        uint32_t suf[NB];     // suf[j] = min over the pairs after j
        suf[NB - 1] = INF2;
        static_rfor<0, NB - 1>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const uint32_t b = pair_min(std::integral_constant<int, j + 1>{});
            if constexpr (j == NB - 2) suf[j] = b; else suf[j] = min2(suf[j + 1], b);
        });
        uint32_t pre = INF2;  // min over the pairs before j
        static_for<0, NB>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr bool has_pre = j > 0, has_suf = j < NB - 1, has_partner = 2 * j + 1 < D;
            auto others = [&](uint32_t partner) -> uint32_t {
                if constexpr (has_pre && has_suf && has_partner) return min3(pre, suf[j], partner);
                else if constexpr (has_pre && has_suf) return min2(pre, suf[j]);
                else if constexpr (has_pre && has_partner) { if constexpr (FOLD) return min2(pre, partner); else return min3_umax(pre, partner); }
                else if constexpr (has_suf && has_partner) return min3_umax(suf[j], partner);
                else if constexpr (has_pre) return pre;
                else if constexpr (has_suf) return suf[j];
                else return partner;
            };
            if constexpr (has_partner) {
                const uint32_t o0 = others(a[2 * j + 1]), o1 = others(a[2 * j]);
                if constexpr (has_suf) { const uint32_t b = pair_min(jc); if constexpr (j == 0) pre = FOLD ? min2_umax(b) : b; else pre = min2(pre, b); }
                put(2 * j, o0); put(2 * j + 1, o1);
            } else {
                put(2 * j, others(0u));
            }
        });
    }
}

// phase A for the row a lane owns in one block row of degree D, both frames at once.  u: [D] message registers.
// -> parity word: bit 15 / bit 31 = the row's parity of hard(lam) in the low / high frame
template <int D, bool SYNDROME_ONLY, class Row>
__device__ __forceinline__ uint32_t rows_a(const char *lds, Row tabrow, uint32_t p4, uint32_t vmask, uint32_t *u) {
    asm volatile("" : "+v"(p4));   // keeps the loop-invariant address arithmetic inside the turn loop, row by row
    if constexpr (SYNDROME_ONLY) LDPC_COLD_PATH();
    uint32_t l[D];
    static_for<0, D>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        l[k] = lds_ld<uint32_t>(lds + tabrow.hi(k), qc_wrap(p4 + tabrow.lo(k), vmask));
    });
    uint32_t par = 0;
#pragma unroll
    for (int k = 0; k < D; k++) par ^= l[k];
    if constexpr (SYNDROME_ONLY) return par;
    uint32_t X = 0, a[D];
#pragma unroll
    for (int k = 0; k < D; k++) {
        const uint32_t tn = fma_k(u[k], K75, l[k]);
        u[k] = tn;
        X ^= tn;
        a[k] = tn & ABS;
    }
    loo_min_update<D>(u, a, ~X & SGN);   // sign(u'_k) = 1 ^ parity of the OTHER signs = 1 ^ X ^ sign(tN_k)
    return par;
}

template <int SZ, class Plan, class T, int P, int Q, int I0, int I1>
__device__ __forceinline__ void round_chunk(char *lds, uint32_t p4, uint32_t vmask, const uint32_t *u, const uint32_t *orig_rot) {
    using S = Split<Plan, T>;
    constexpr uint32_t ES = 4, CPW = QcGeom<SZ>::CPW, V = QcGeom<SZ>::V;
    asm volatile("" : "+v"(p4));
    if constexpr (Q == 0) {
        static_for<I0, I1>([&](auto ic) {
            constexpr int e = S::nth(Q, P, decltype(ic)::value);
            constexpr int ms = S::slot(e), os = S::oslot(T::bc[e]);
            lds_st<uint32_t>(lds + T::bc[e] * V * ES, qc_wrap(p4 + T::rot[e] * CPW * ES, vmask), fma_k(u[ms], KN75, orig_rot[os]));
        });
        return;
    }
    uint32_t cur[I1 - I0], adr[I1 - I0];
    static_for<I0, I1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int e = S::nth(Q, P, i);
        adr[i - I0] = qc_wrap(p4 + T::rot[e] * CPW * ES, vmask);
        cur[i - I0] = lds_ld<uint32_t>(lds + T::bc[e] * V * ES, adr[i - I0]);
    });
    static_for<I0, I1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int e = S::nth(Q, P, i);
        constexpr int ms = S::slot(e);
        lds_st<uint32_t>(lds + T::bc[e] * V * ES, adr[i - I0], fma_k(u[ms], KN75, cur[i - I0]));
    });
    asm volatile("" ::: "memory");
}
template <int SZ, class Plan, class T, int P, int Q, int I0>
__device__ __forceinline__ void round(char *lds, uint32_t p4, uint32_t vmask, const uint32_t *u, const uint32_t *orig_rot) {
    constexpr int CNT = Split<Plan, T>::count(Q, P), CH = SZ < 64 ? SPLIT_CH_SMALL : SPLIT_CH;
    if constexpr (I0 < CNT) {
        round_chunk<SZ, Plan, T, P, Q, I0, (I0 + CH < CNT ? I0 + CH : CNT)>(lds, p4, vmask, u, orig_rot);
        round<SZ, Plan, T, P, Q, I0 + CH>(lds, p4, vmask, u, orig_rot);
    }
}

// -(x) of a channel LLR as fp16 bits, saturated at +-LLR16_MAX, a zero as +0
__device__ __forceinline__ uint32_t neg_llr16(float x) {
    const float v = fminf(fmaxf(x, -LLR16_MAX), LLR16_MAX);
    const _Float16 h = (_Float16)(0.0f - v);          // 0 - (+-0) = +0; the f32 negation is exact, the conversion rounds to nearest even
    uint16_t b;
    __builtin_memcpy(&b, &h, 2);
    return b == 0x8000u ? 0u : (uint32_t)b;           // (a value that underflows to -0 in fp16)
}
__device__ __forceinline__ double lam_of(uint32_t packed, int h) {   // lam = -L of half h, as a double
    const uint16_t b = (uint16_t)(h ? packed >> 16 : packed & 0xffffu);
    _Float16 v;
    __builtin_memcpy(&v, &b, 2);
    return -(double)(float)v;
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
    struct Where {   // recomputed where needed, so that it does not occupy registers next to the messages
        uint32_t sub, r0; long long frame0; bool valid[2]; size_t fN[2];
        __device__ __forceinline__ Where(uint32_t p, int batch) {
            asm volatile("" : "+v"(p));
            const uint32_t lane = p / ES;
            sub = lane % CPW;
            r0 = lane / CPW;
            frame0 = ((long long)blockIdx.x * CPW + sub) * 2;      // the lane's low-half frame; the high half is frame0 + 1
#pragma unroll
            for (int h = 0; h < 2; h++) { valid[h] = frame0 + h < batch; fN[h] = (size_t)(valid[h] ? frame0 + h : 0) * N; }
        }
    };
    uint32_t u[S::NMSG], orig[S::NORIG];
#pragma unroll
    for (int i = 0; i < S::NMSG; i++) u[i] = 0u;   // Min.hs:60-61: no messages yet
#pragma unroll
    for (int i = 0; i < S::NORIG; i++) orig[i] = 0u;
    // ---- L <- -(channel LLRs) of both frames: group P fills the block columns bc with bc % NP == P.  Every LLR is read from
    // memory once; the hard decisions of the lane's own columns stay in `obits` (the answer of a frame that runs out of turns).
    typename SplitResult<NBCP>::Bits obits[2];
    {
        const Where w(p4, A.batch);   // (a frame past the batch shadows frame 0: every load below is unconditional and in range)
        with_llr_format(A.llr_fmt, [&](auto fc) {
            constexpr int FMT = decltype(fc)::value;
            float x[NBCP][2];          // all of the thread's loads first, back to back: 2 x 22 of them
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
                        obits[h].set(bc / Plan::NP, (b >> 15) & 1u);      // hard(llr) = llr > 0 = sign of -llr
                        packed |= b << (16 * h);
                    }
                    lds_st<uint32_t>(lds, p4 + (bc * V * ES), packed);
                }
            });
        });
    }
    __syncthreads();
    static_for<0, Plan::NBC>([&](auto bcc) {   // L == -(channel LLRs) right now: the round-0 (rotated) copies are an LDS gather away
        constexpr int bc = decltype(bcc)::value;
        if constexpr (S::oowner(bc) == P) {
            // (constexpr VARIABLES: a constexpr function call in a subscript is not a constant expression and is left as a
            //  run-time loop -- measured here: 58 ms per launch)
            constexpr int e0 = Rounds<T>::round0_edge(bc);
            constexpr int os = S::oslot(bc);
            orig[os] = lds_ld<uint32_t>(lds + bc * V * ES, qc_wrap(p4 + T::rot[e0] * CPW * ES, vmask));
        }
    });

    volatile uint32_t *flags = reinterpret_cast<volatile uint32_t *>(lds + LAM_BYTES);
    // done: bit 2s + h = frame h of slot s of this workgroup has finished (workgroup-uniform: derived from the shared flags)
    constexpr uint32_t FULL = (1u << (2 * CPW)) - 1;
    uint32_t done = 0;
#pragma unroll
    for (int s2 = 0; s2 < 2 * CPW; s2++) done |= (((long long)blockIdx.x * CPW + s2 / 2) * 2 + (s2 & 1) < A.batch) ? 0u : (1u << s2);
    SplitResult<NBCP> res[2];
    res[0].bits = obits[0]; res[1].bits = obits[1];
    const int turns = A.max_iters;
    const uint32_t my_slot = (p4 / ES) % CPW;

    for (int n = 0;; n++) {
        if (done == FULL) break;
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
        const bool last = (n >= turns);
        // ---- phase A over the group's block rows
        uint32_t par = 0;
        static_for<0, Plan::NBR>([&](auto brc) {
            constexpr int br = decltype(brc)::value;
            if constexpr (S::owner_br(br) == P) {
                constexpr int D = Plan::deg(br), ms0 = S::slot(Plan::ebeg(br));
                StatRow<float, SZ, T, Plan::ebeg(br)> row;
                if (last) par |= rows_a<D, true>(lds, row, p4, vmask, (uint32_t *)nullptr);
                else par |= rows_a<D, false>(lds, row, p4, vmask, &u[ms0]);
            }
        });
        // per wave: bit 2s + h = some lane of slot s saw an odd row parity in frame h
        uint32_t wbits = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const unsigned long long ub = __ballot((par >> (15 + 16 * h)) & 1u);
#pragma unroll
            for (int s2 = 0; s2 < CPW; s2++) {
                unsigned long long m = 0;
                for (int i = 0; i < 64; i += CPW) m |= 1ull << i;
                wbits |= ((ub & (m << s2)) != 0ull) ? (1u << (2 * s2 + h)) : 0u;
            }
        }
        if ((tid & 63) == 0) flags[tid >> 6] = wbits;
        __syncthreads();   // syndrome OR over the workgroup's waves; also fences phase A reads from phase B writes
        uint32_t fbits = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) fbits |= flags[w];
        fbits = __builtin_amdgcn_readfirstlane(fbits);
        const uint32_t newly = ~fbits & ~done & FULL;   // Min.hs:75: frames whose syndrome is zero now
        if ((newly >> (2 * my_slot)) & 3u) {
            LDPC_COLD_PATH();   // once per frame
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
        done |= newly;
        // the snapshot read columns that the OTHER group rewrites in round 0 when the workgroup goes on
        if (newly != 0u && done != FULL) __syncthreads();
        if (last) break;   // Min.hs:76
        if (done != FULL) {
            __builtin_amdgcn_s_setprio(SPLIT_PRIO_B);
            static_for<0, Rounds<T>::num_rounds()>([&](auto qc) {
                pk::round<SZ, Plan, T, P, decltype(qc)::value, 0>(lds, p4, vmask, u, orig);
                __syncthreads();   // the next round adds into the same columns
            });
            __builtin_amdgcn_s_setprio(SPLIT_PRIO_A);
        }
    }

    const Where w(p4, A.batch);
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (!w.valid[h]) continue;
        const bool converged = res[h].converged();
        static_for<0, Plan::NBC>([&](auto bcc) {   // hard(lam at convergence), or hard(channel LLR) for a frame out of turns
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
            if (A.iters) A.iters[w.frame0 + h] = converged ? res[h].turn() : turns;
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
}  // namespace pk
}  // namespace ldpc
