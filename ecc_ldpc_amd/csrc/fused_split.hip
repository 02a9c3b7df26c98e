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

#include "fused_rows.h"

#ifndef SPLIT_ORIG_REGS
#define SPLIT_ORIG_REGS 1
#endif
// wave priority while in phase A (check rows: long stretches of independent VALU work) and in phase B (column
// rounds: short, LDS-bound, barrier-separated).  Measured on jpl.4096, 65 536 frames: A=0/B=0 20.72 ms,
// A=0/B=2 20.90, A=2/B=0 20.24 (A = 1, 2 or 3 alike).  The priority is raised after the first phase B only:
// a workgroup that starts (global loads, first syndrome) at high priority costs 0.1-0.2 ms.
#ifndef SPLIT_PRIO_A
#define SPLIT_PRIO_A 2
#endif
#ifndef SPLIT_PRIO_B
#define SPLIT_PRIO_B 0
#endif
#ifndef SPLIT_WAVES_PER_EU
#define SPLIT_WAVES_PER_EU 4
#endif
// edges per read-add-write batch inside a column round (2 registers per edge in flight).  Measured on jpl.4096,
// 65 536 frames: 4: 20.66 ms, 6: 20.15, 8: 20.01, 10: 19.50, 11: 19.36, 12: 19.29, 14: 19.42, 16: 19.57, 24: 19.47.
// number of wave groups a frame's block rows are dealt to (block row br -> group br % SPLIT_NP).  2 = the wave
// PAIRS described above.
#ifndef SPLIT_NP
#define SPLIT_NP 2
#endif
#ifndef SPLIT_CH
#define SPLIT_CH 12
#endif
// sz = 32 (two frames per workgroup, one wave per pair): jpl.1024 4.80 ms at 12, 4.76 at 16, 4.68 at 24
#ifndef SPLIT_CH_SMALL
#define SPLIT_CH_SMALL 24
#endif
// the tanh rule needs ~3 transient registers per edge of a row (e, suffix A, suffix S).  Measured on jpl.4096,
// 16 384 frames, hyperbolic-recurrence rule: 2 waves per SIMD (209 VGPRs, no spills) 4.92 Gbit/s, 3 waves
// (168 VGPRs, 37 spilled) 5.80; 4 waves (128 VGPRs) spills several hundred registers.
#ifndef SPLIT_TANH_WAVES_PER_EU
#define SPLIT_TANH_WAVES_PER_EU 3
#endif

namespace ldpc {

// ownership and per-pair register slots, all compile time
template <class Plan, class T>
struct Split {
    static constexpr int br_of(int e) {
        int br = 0;
        for (int b = 0; b < Plan::NBR; b++) if (Plan::ebeg(b) <= e) br = b;
        return br;
    }
    static constexpr int owner_br(int br) { return br % SPLIT_NP; }
    static constexpr int owner(int e) { return owner_br(br_of(e)); }
    static constexpr int slot(int e) {  // index of e among its owner's edges, plan order
        int c = 0;
        for (int j = 0; j < e; j++) c += owner(j) == owner(e) ? 1 : 0;
        return c;
    }
    static constexpr int nmsg(int p) {
        int c = 0;
        for (int e = 0; e < T::NEDGE; e++) c += owner(e) == p ? 1 : 0;
        return c;
    }
    static constexpr int max_over_groups(int (*f)(int)) { int m = 0; for (int g = 0; g < SPLIT_NP; g++) m = f(g) > m ? f(g) : m; return m; }
    static constexpr int NMSG = max_over_groups(nmsg);
    // channel LLR of the column a thread writes in round 0 of block column bc: held by the owner of that edge
    static constexpr int oowner(int bc) { return owner(Rounds<T>::round0_edge(bc)); }
    static constexpr int oslot(int bc) {
        int c = 0;
        for (int j = 0; j < bc; j++) c += oowner(j) == oowner(bc) ? 1 : 0;
        return c;
    }
    static constexpr int norig(int p) {
        int c = 0;
        for (int bc = 0; bc < T::NBC; bc++) c += oowner(bc) == p ? 1 : 0;
        return c;
    }
    static constexpr int NORIG = max_over_groups(norig);
    // edges of round q owned by pair p, highest edge index first
    static constexpr int count(int q, int p) {
        int c = 0;
        for (int e = 0; e < T::NEDGE; e++) c += (Rounds<T>::round_of(e) == q && owner(e) == p) ? 1 : 0;
        return c;
    }
    static constexpr int nth(int q, int p, int i) {
        int c = 0;
        for (int e = T::NEDGE - 1; e >= 0; e--)
            if (Rounds<T>::round_of(e) == q && owner(e) == p) { if (c == i) return e; c++; }
        return -1;
    }
};

template <typename CT, int SZ, class Plan, class T, int P, int Q, int I0, int I1>
__device__ __forceinline__ void split_round_chunk(char *lds, uint32_t p4, uint32_t vmask, const CT *msg, const CT *orig_rot, const float *gllr, uint32_t r0) {
    using S = Split<Plan, T>;
    constexpr uint32_t ES = sizeof(CT), CPW = SZ >= 64 ? 1 : 64 / SZ, V = SZ * CPW;
    asm volatile("" : "+v"(p4));
    if constexpr (Q == 0) {
        static_for<I0, I1>([&](auto ic) {
            // (constexpr VARIABLES: a constexpr function call in a subscript is not a constant expression and was
            //  left as a run-time loop, which kept msg[] in scratch memory)
            constexpr int e = S::nth(Q, P, decltype(ic)::value);
            constexpr int ms = S::slot(e), os = S::oslot(T::bc[e]);
            CT o;
            if constexpr (SPLIT_ORIG_REGS) o = orig_rot[os]; else o = (CT)gllr[T::bc[e] * SZ + ((r0 + T::rot[e]) & (SZ - 1))];
            lds_st<CT>(lds + T::bc[e] * V * ES, (p4 + T::rot[e] * CPW * ES) & vmask, msg[ms] + o);
        });
        return;
    }
    CT cur[I1 - I0];
    uint32_t adr[I1 - I0];
    static_for<I0, I1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int e = S::nth(Q, P, i);
        adr[i - I0] = (p4 + T::rot[e] * CPW * ES) & vmask;
        cur[i - I0] = lds_ld<CT>(lds + T::bc[e] * V * ES, adr[i - I0]);
    });
    static_for<I0, I1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int e = S::nth(Q, P, i);
        constexpr int ms = S::slot(e);
        lds_st<CT>(lds + T::bc[e] * V * ES, adr[i - I0], msg[ms] + cur[i - I0]);
    });
    asm volatile("" ::: "memory");
}
template <typename CT, int SZ, class Plan, class T, int P, int Q, int I0>
__device__ __forceinline__ void split_round(char *lds, uint32_t p4, uint32_t vmask, const CT *msg, const CT *orig_rot, const float *gllr, uint32_t r0) {
    constexpr int CNT = Split<Plan, T>::count(Q, P), CH = SZ < 64 ? SPLIT_CH_SMALL : SPLIT_CH;
    if constexpr (I0 < CNT) {
        split_round_chunk<CT, SZ, Plan, T, P, Q, I0, (I0 + CH < CNT ? I0 + CH : CNT)>(lds, p4, vmask, msg, orig_rot, gllr, r0);
        split_round<CT, SZ, Plan, T, P, Q, I0 + CH>(lds, p4, vmask, msg, orig_rot, gllr, r0);
    }
}

// The whole decode of one pair: P is a compile-time constant, so every ownership test below is resolved
// at compile time and the two pairs are two independent straight-line programs (one wave-uniform branch
// in the kernel).  Keeping them as separate regions matters for the register allocator: with both pairs'
// code merged in one loop body the 105 loop-carried registers met in phi nodes at every branch merge and
// were spilled wholesale.
template <typename CT, int VARIANT, class Plan, int SZ, class T, int P>
__device__ __forceinline__ void split_body(const FusedArgs &A, char *lds, const uint32_t tid) {
    using S = Split<Plan, T>;
    constexpr int CPW = SZ >= 64 ? 1 : 64 / SZ, V = SZ * CPW;  // frames per workgroup, threads per pair
    constexpr int N = Plan::NBC * SZ, THREADS = SPLIT_NP * V, NW = THREADS / 64;
    constexpr uint32_t ES = sizeof(CT), vmask = V * ES - 1;
    constexpr int LAM_BYTES = Plan::NBC * V * (int)ES;
    // Only p4 (the lane's LDS byte offset inside a block column) lives across the iteration loop; everything else
    // about the lane's place -- frame, row, global offsets -- is recomputed from it where needed (Where), so that
    // it does not occupy registers next to the messages.
    const uint32_t p4 = (tid & (V - 1)) * ES;
    struct Where {
        uint32_t sub, r0; long long frame; bool valid; size_t fN, fE;
        __device__ __forceinline__ Where(uint32_t p, int batch) {
            asm volatile("" : "+v"(p));            // keep the compiler from carrying these over from an earlier Where
            const uint32_t lane = p / ES;          // position inside the pair
            sub = lane % CPW;                      // frame inside the workgroup (frames interleave lane by lane)
            r0 = lane / CPW;                       // circulant row / own column inside a block
            frame = (long long)blockIdx.x * CPW + sub;
            valid = frame < batch;
            fN = (size_t)(valid ? frame : 0) * N;  // lanes of a frame past the batch shadow frame 0 and store nothing
            fE = (size_t)(valid ? frame : 0) * Plan::NEDGE * SZ;
        }
    };
    const Where w0(p4, A.batch);
    const uint32_t r0 = w0.r0;
    const size_t fN = w0.fN, fE = w0.fE;
    // ---- messages (own block rows) and round-0 channel LLRs (own round-0 edges)
    CT msg[S::NMSG];
    CT orig[SPLIT_ORIG_REGS ? S::NORIG : 1];
#pragma unroll
    for (int i = 0; i < S::NMSG; i++) msg[i] = CT(0);  // Orig.hs:64-65
#pragma unroll
    for (int i = 0; i < (SPLIT_ORIG_REGS ? S::NORIG : 1); i++) orig[i] = CT(0);
    // ---- lam <- LLRs (or the given lam): pair P fills the block columns bc with bc % 2 == P.  One dispatch on the
    // LLR element type around ALL of the thread's loads (46 of them): they issue back to back.
    // Every channel LLR is read from global memory ONCE (the input may be page-locked HOST memory read over PCIe,
    // api.cc zero-copy path): the hard decisions of the thread's own columns are kept in `obits` -- the answer of a
    // frame that runs out of turns (Orig.hs:70) -- and the rotated copies phase B wants come out of LDS below.
    uint32_t obits = 0;
    with_llr_format(A.llr_fmt, [&](auto fc) {
        constexpr int FMT = decltype(fc)::value;
        static_for<0, Plan::NBC>([&](auto bcc) {
            constexpr int bc = decltype(bcc)::value;
            if constexpr ((bc % SPLIT_NP) == P) {
                CT v = maybe_round_f16<CT>(load_llr_as<CT, FMT>(A.llr, fN + bc * SZ + r0), A.llr_round16);
                obits |= (v > CT(0) ? 1u : 0u) << (bc / SPLIT_NP);
                if (A.step_mode) v = (CT)A.st_lam[fN + bc * SZ + r0];
                lds_st<CT>(lds, p4 | (bc * V * ES), v);
            }
        });
        if (A.step_mode) {   // teacher-forced step: LDS holds the given lam, the channel LLRs come from memory
            static_for<0, Plan::NBC>([&](auto bcc) {
                constexpr int bc = decltype(bcc)::value;
                if constexpr (SPLIT_ORIG_REGS && S::oowner(bc) == P) {
                    constexpr int e0 = Rounds<T>::round0_edge(bc);
                    constexpr int os = S::oslot(bc);
                    orig[os] = maybe_round_f16<CT>(load_llr_as<CT, FMT>(A.llr, fN + bc * SZ + ((r0 + T::rot[e0]) & (SZ - 1))), A.llr_round16);
                }
            });
        }
    });
    if (A.step_mode) {
        static_for<0, Plan::NBR>([&](auto brc) {
            constexpr int br = decltype(brc)::value;
            if constexpr (S::owner_br(br) == P) {
                constexpr int D = Plan::deg(br);
                static_for<0, D>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    constexpr int ms = S::slot(Plan::ebeg(br) + k);
                    msg[ms] = (CT)A.st_ne_in[fE + (size_t)SZ * Plan::ebeg(br) + (size_t)D * r0 + k];
                });
            }
        });
    }
    __syncthreads();
    if (!A.step_mode) {   // lam == channel LLRs right now: the round-0 (rotated) copies are an LDS gather away
        static_for<0, Plan::NBC>([&](auto bcc) {
            constexpr int bc = decltype(bcc)::value;
            if constexpr (SPLIT_ORIG_REGS && S::oowner(bc) == P) {
                constexpr int e0 = Rounds<T>::round0_edge(bc);
                constexpr int os = S::oslot(bc);
                orig[os] = lds_ld<CT>(lds + bc * V * ES, (p4 + T::rot[e0] * CPW * ES) & vmask);
            }
        });
    }

    volatile uint32_t *flags = reinterpret_cast<volatile uint32_t *>(lds + LAM_BYTES);
    // done: bit s = frame s of this workgroup has finished.  Workgroup-uniform (derived from the shared flags), so
    // loop control and barriers stay uniform with several frames.  A finished frame keeps its answer in `snap`;
    // its lanes then keep computing on their own (disjoint) LDS columns until the workgroup's other frames are
    // done -- masking them off instead makes every message register live across divergent control flow.
    constexpr uint32_t FULL = (1u << CPW) - 1;
    uint32_t done = 0;
#pragma unroll
    for (int s2 = 0; s2 < CPW; s2++) done |= ((long long)blockIdx.x * CPW + s2 < A.batch) ? 0u : (1u << s2);
    // bits 0..21: hard(lam) of this lane's columns at the moment its frame converged; bit 22: converged;
    // bits 23..31: the iteration it converged at (max_iters <= kSplitMaxIters: fused.hip falls back to fused_msg above)
    static_assert(Plan::NBC <= 44, "result word layout");
    uint32_t res = obits;   // until the frame converges: the hard decisions of its channel LLRs (bit 22 clear)
    const int turns = A.step_mode ? 1 : A.max_iters;

    for (int n = 0;; n++) {
        if (done == FULL) break;
        if (A.trace && !((done >> ((p4 / ES) % CPW)) & 1u)) {
            LDPC_COLD_PATH();
            const Where w(p4, A.batch);
            static_for<0, Plan::NBC>([&](auto bcc) {
                constexpr int bc = decltype(bcc)::value;
                if constexpr ((bc % SPLIT_NP) == P)
                    A.trace[((size_t)w.frame * (A.max_iters + 1) + n) * N + bc * SZ + w.r0] = (double)lds_ld<CT>(lds, p4 | (bc * V * ES));
            });
        }
        const bool last = (n >= turns);
        // ---- phase A over the pair's block rows
        bool unsat = false;
        static_for<0, Plan::NBR>([&](auto brc) {
            constexpr int br = decltype(brc)::value;
            if constexpr (S::owner_br(br) == P) {
                constexpr int D = Plan::deg(br), ms0 = S::slot(Plan::ebeg(br));
                StatRow<CT, SZ, T, Plan::ebeg(br)> row;
                if (last) unsat |= rows_a<CT, VARIANT, D, 1, 0, true>(lds, row, p4, vmask, (CT *)nullptr);
                else unsat |= rows_a<CT, VARIANT, D, 1, 0, false>(lds, row, p4, vmask, &msg[ms0]);
            }
        });
        // per wave: bit s = some lane of frame s saw an odd row parity (frames interleave lane by lane)
        const unsigned long long ub = __ballot(unsat);
        uint32_t wbits = 0;
#pragma unroll
        for (int s2 = 0; s2 < CPW; s2++) {
            unsigned long long m = 0;
            for (int i = 0; i < 64; i += CPW) m |= 1ull << i;
            wbits |= ((ub & (m << s2)) != 0ull) ? (1u << s2) : 0u;
        }
        if ((tid & 63) == 0) flags[tid >> 6] = wbits;
        __syncthreads();  // syndrome OR over the workgroup's waves; also fences phase A reads from phase B writes
        uint32_t fbits = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) fbits |= flags[w];
        fbits = __builtin_amdgcn_readfirstlane(fbits);
        if (A.step_mode) {
            const Where w(p4, A.batch);
            if (w.valid && w.r0 == 0 && P == 0) A.st_syn[w.frame] = ((fbits >> w.sub) & 1u) ? 0 : 1;
        } else {
            const uint32_t newly = ~fbits & ~done & FULL;  // Orig.hs:69: frames whose syndrome is zero now
            if ((newly >> ((p4 / ES) % CPW)) & 1u) {
                LDPC_COLD_PATH();   // once per frame
                res = (1u << 22) | ((uint32_t)n << 23);
                static_for<0, Plan::NBC>([&](auto bcc) {
                    constexpr int bc = decltype(bcc)::value;
                    if constexpr ((bc % SPLIT_NP) == P) {
                        CT v = lds_ld<CT>(lds, p4 | (bc * V * ES));
                        res |= (v > CT(0) ? 1u : 0u) << (bc / SPLIT_NP);
                    }
                });
                if (A.final_lam) {
                    const Where w(p4, A.batch);
                    static_for<0, Plan::NBC>([&](auto bcc) {
                        constexpr int bc = decltype(bcc)::value;
                        if constexpr ((bc % SPLIT_NP) == P) A.final_lam[w.fN + bc * SZ + w.r0] = (double)lds_ld<CT>(lds, p4 | (bc * V * ES));
                    });
                }
            }
            done |= newly;
            // the snapshot read columns that the OTHER pair rewrites in round 0 when the workgroup goes on
            if (CPW > 1 && newly != 0u && done != FULL) __syncthreads();
        }
        if (last) break;  // Orig.hs:70
        if (done != FULL) {
            __builtin_amdgcn_s_setprio(SPLIT_PRIO_B);
            static_for<0, Rounds<T>::num_rounds()>([&](auto qc) {
                split_round<CT, SZ, Plan, T, P, decltype(qc)::value, 0>(lds, p4, vmask, msg, orig, reinterpret_cast<const float *>(A.llr) + fN, r0);
                __syncthreads();  // the next round adds into the same columns
            });
            __builtin_amdgcn_s_setprio(SPLIT_PRIO_A);
        }
        if (A.step_mode) break;
    }

    const Where w(p4, A.batch);
    if (!w.valid) return;
    if (A.step_mode) {
        static_for<0, Plan::NBC>([&](auto bcc) {
            constexpr int bc = decltype(bcc)::value;
            if constexpr ((bc % SPLIT_NP) == P) A.final_lam[w.fN + bc * SZ + w.r0] = (double)lds_ld<CT>(lds, p4 | (bc * V * ES));
        });
        static_for<0, Plan::NBR>([&](auto brc) {
            constexpr int br = decltype(brc)::value;
            if constexpr (S::owner_br(br) == P) {
                constexpr int D = Plan::deg(br);
                static_for<0, D>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    constexpr int ms = S::slot(Plan::ebeg(br) + k);
                    A.st_ne_out[w.fE + (size_t)SZ * Plan::ebeg(br) + (size_t)D * w.r0 + k] = (double)msg[ms];
                });
            }
        });
        return;
    }
    // ---- result: hard(lam at convergence) for a converged frame, hard(channel LLR) otherwise (Orig.hs:59,69-70)
    const bool converged = (res >> 22) & 1u;
    if (converged) {
        static_for<0, Plan::NBC>([&](auto bcc) {
            constexpr int bc = decltype(bcc)::value;
            if constexpr ((bc % SPLIT_NP) == P) A.bits[w.fN + bc * SZ + w.r0] = (res >> (bc / SPLIT_NP)) & 1u;
        });
    } else {
        static_for<0, Plan::NBC>([&](auto bcc) {
            constexpr int bc = decltype(bcc)::value;
            if constexpr ((bc % SPLIT_NP) == P) A.bits[w.fN + bc * SZ + w.r0] = (res >> (bc / SPLIT_NP)) & 1u;   // hard(channel LLR)
        });
        if (A.final_lam) {
            with_llr_format(A.llr_fmt, [&](auto fc) {
                constexpr int FMT = decltype(fc)::value;
                static_for<0, Plan::NBC>([&](auto bcc) {
                    constexpr int bc = decltype(bcc)::value;
                    if constexpr ((bc % SPLIT_NP) == P) {
                        const size_t gi = w.fN + bc * SZ + w.r0;
                        A.final_lam[gi] = (double)maybe_round_f16<CT>(load_llr_as<CT, FMT>(A.llr, gi), A.llr_round16);
                    }
                });
            });
        }
    }
    if (w.r0 == 0 && P == 0) {
        if (A.iters) A.iters[w.frame] = converged ? (int)(res >> 23) : turns;
        if (A.conv) A.conv[w.frame] = converged ? 1 : 0;
    }
}

template <int SZ> struct SplitGeom {
    static constexpr int CPW = SZ >= 64 ? 1 : 64 / SZ, V = SZ * CPW, THREADS = SPLIT_NP * V, NW = THREADS / 64;
};

template <typename CT, int VARIANT, class Plan, int SZ, class T>
__global__ __launch_bounds__((SplitGeom<SZ>::THREADS), (VARIANT == LDPC_V_TANH ? SPLIT_TANH_WAVES_PER_EU : SPLIT_WAVES_PER_EU))
void fused_split_kernel(FusedArgs A) {
    using G = SplitGeom<SZ>;
    static_assert((SZ & (SZ - 1)) == 0 && SZ >= 16, "circulant size must be a power of two");
    __shared__ __attribute__((aligned(16))) char lds[Plan::NBC * G::V * (int)sizeof(CT) + 4 * G::NW];
    const uint32_t tid = threadIdx.x;
    const uint32_t pair = __builtin_amdgcn_readfirstlane(tid / G::V);  // wave-uniform (V is a multiple of 64)
    // all programs execute the same number of barriers (same loop structure and round count)
    static_for<0, SPLIT_NP>([&](auto pc) {
        constexpr int P = decltype(pc)::value;
        if (pair == (uint32_t)P) split_body<CT, VARIANT, Plan, SZ, T, P>(A, lds, tid);
    });
}

bool fused_split_has(int variant, int dtype, int sz, int static_id) {
    if (dtype != LDPC_F32 || !(variant == LDPC_MINSUM || variant == LDPC_TANH)) return false;
    return (sz == 128 && static_id == 2) || (sz == 32 && static_id == 1);
}

template <int VARIANT, int SZ, class T>
static void launch_split(hipStream_t st, FusedArgs &a) {
    using G = SplitGeom<SZ>;
    const int grid = (a.batch + G::CPW - 1) / G::CPW;
    hipLaunchKernelGGL((fused_split_kernel<float, VARIANT, PlanAR4JA45, SZ, T>), dim3(grid), dim3(G::THREADS), 0, st, a);
}

int fused_split_launch(int variant, int sz, hipStream_t st, FusedArgs &a, KernelTimer *timer, LaunchInfo *info) {
    if (info && !a.step_mode) {
        snprintf(info->name, sizeof(info->name), "ldpc::fused_split_kernel<float, %d, ldpc::PlanAR4JA45, %d, ", variant == LDPC_MINSUM ? LDPC_V_MINSUM : LDPC_V_TANH, sz);
        info->threads = sz == 128 ? SplitGeom<128>::THREADS : SplitGeom<32>::THREADS;
        info->frames_per_wg = sz == 128 ? SplitGeom<128>::CPW : SplitGeom<32>::CPW;
    }
    if (timer && !a.step_mode) timer->begin(st);
    if (sz == 128) {
        if (variant == LDPC_MINSUM) launch_split<LDPC_V_MINSUM, 128, TabJpl4096>(st, a);
        else launch_split<LDPC_V_TANH, 128, TabJpl4096>(st, a);
    } else {
        if (variant == LDPC_MINSUM) launch_split<LDPC_V_MINSUM, 32, TabJpl1024>(st, a);
        else launch_split<LDPC_V_TANH, 32, TabJpl1024>(st, a);
    }
    if (timer && !a.step_mode) timer->end(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(LDPC_EHIP, "fused_split launch: %s", hipGetErrorString(e));
    return LDPC_OK;
}

}  // namespace ldpc
