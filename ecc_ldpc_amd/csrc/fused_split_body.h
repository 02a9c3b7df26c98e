// fused_split_body.h -- device code of the four-wave "split" fused kernel (see fused_split.hip for the design notes).
// Compiled two ways from this one source: ahead of time for the shipped matrices (fused_split.hip, tables in
// generated_tables.h) and at run time by hiprtc for any other single-circulant quasi-cyclic H (jit.cc: plan and
// rotation table generated as constexpr structs from the code's description) -- so it includes nothing host-side.
#pragma once
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
// edges per read-add-write batch inside a column round (2 registers per edge in flight).  Measured on jpl.4096,
// 65 536 frames: 4: 20.66 ms, 6: 20.15, 8: 20.01, 10: 19.50, 11: 19.36, 12: 19.29, 14: 19.42, 16: 19.57, 24: 19.47.
#ifndef SPLIT_CH
#define SPLIT_CH 12
#endif
// sz = 32 (two frames per workgroup, one wave per pair): jpl.1024 4.80 ms at 12, 4.76 at 16, 4.68 at 24
#ifndef SPLIT_CH_SMALL
#define SPLIT_CH_SMALL 24
#endif

namespace ldpc {

// hard bits of the block columns a lane's group fills, NB of them
#ifndef SPLIT_RESULT_PACKED
#define SPLIT_RESULT_PACKED 1   // run-time compiled instances set 0: no limit on max_iters
#endif
template <int NB, bool PACKED = (SPLIT_RESULT_PACKED && NB <= 22)> struct SplitResult;
template <int NB> struct SplitResult<NB, true> {
    struct Bits {
        uint32_t w = 0;
        __device__ __forceinline__ void set(int i, bool b) { w |= (b ? 1u : 0u) << i; }
        __device__ __forceinline__ uint32_t get(int i) const { return (w >> i) & 1u; }
    } bits;
    __device__ __forceinline__ void converge_at(int n) { bits.w = (1u << 22) | ((uint32_t)n << 23); }
    __device__ __forceinline__ bool converged() const { return (bits.w >> 22) & 1u; }
    __device__ __forceinline__ int turn() const { return (int)(bits.w >> 23); }
};
template <int NB> struct SplitResult<NB, false> {
    struct Bits {
        uint32_t w[(NB + 31) / 32] = {};
        __device__ __forceinline__ void set(int i, bool b) { w[i >> 5] |= (b ? 1u : 0u) << (i & 31); }
        __device__ __forceinline__ uint32_t get(int i) const { return (w[i >> 5] >> (i & 31)) & 1u; }
    } bits;
    uint32_t flag = 0;   // bit 31: converged; low bits: the turn
    __device__ __forceinline__ void converge_at(int n) { bits = Bits{}; flag = 0x80000000u | (uint32_t)n; }
    __device__ __forceinline__ bool converged() const { return flag >> 31; }
    __device__ __forceinline__ int turn() const { return (int)(flag & 0x7fffffffu); }
};

// ownership and per-pair register slots, all compile time
template <class Plan, class T>
struct Split {
    static constexpr int br_of(int e) {
        int br = 0;
        for (int b = 0; b < Plan::NBR; b++) if (Plan::ebeg(b) <= e) br = b;
        return br;
    }
    static constexpr int owner_br(int br) { return Plan::owner_br(br); }
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
    static constexpr int max_over_groups(int (*f)(int)) { int m = 0; for (int g = 0; g < Plan::NP; g++) m = f(g) > m ? f(g) : m; return m; }
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
    constexpr uint32_t ES = sizeof(CT), CPW = QcGeom<SZ>::CPW, V = QcGeom<SZ>::V;
    asm volatile("" : "+v"(p4));
    if constexpr (Q == 0) {
        static_for<I0, I1>([&](auto ic) {
            // (constexpr VARIABLES: a constexpr function call in a subscript is not a constant expression and was
            //  left as a run-time loop, which kept msg[] in scratch memory)
            constexpr int e = S::nth(Q, P, decltype(ic)::value);
            constexpr int ms = S::slot(e), os = S::oslot(T::bc[e]);
            CT o;
            if constexpr (SPLIT_ORIG_REGS) o = orig_rot[os]; else o = (CT)gllr[T::bc[e] * SZ + ((r0 + T::rot[e]) % SZ)];
            lds_st<CT>(lds + T::bc[e] * V * ES, qc_wrap(p4 + T::rot[e] * CPW * ES, vmask), msg[ms] + o);
        });
        return;
    }
    CT cur[I1 - I0];
    uint32_t adr[I1 - I0];
    static_for<I0, I1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int e = S::nth(Q, P, i);
        adr[i - I0] = qc_wrap(p4 + T::rot[e] * CPW * ES, vmask);
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
    constexpr int CPW = QcGeom<SZ>::CPW, V = QcGeom<SZ>::V, VT = QcGeom<SZ>::VT;  // frames per workgroup, positions per block column, threads per group
    constexpr int N = Plan::NBC * SZ, THREADS = Plan::NP * VT, NW = THREADS / 64;
    constexpr uint32_t ES = sizeof(CT), vmask = V * ES - 1;
    constexpr int LAM_BYTES = (Plan::NBC * V * (int)ES + 15) / 16 * 16;
    // Only p4 (the lane's LDS byte offset inside a block column) lives across the iteration loop; everything else
    // about the lane's place -- frame, row, global offsets -- is recomputed from it where needed (Where), so that
    // it does not occupy registers next to the messages.
    if constexpr (VT != V) { if ((tid % VT) >= (uint32_t)V) return; }   // circulant size not a multiple of 64: the top lanes of the group idle
    const uint32_t p4 = (tid % VT) * ES;
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
    typename SplitResult<(Plan::NBC + Plan::NP - 1) / Plan::NP>::Bits obits{};
    with_llr_format(A.llr_fmt, [&](auto fc) {
        constexpr int FMT = decltype(fc)::value;
        static_for<0, Plan::NBC>([&](auto bcc) {
            constexpr int bc = decltype(bcc)::value;
            if constexpr ((bc % Plan::NP) == P) {
                CT v = maybe_round_f16<CT>(load_llr_as<CT, FMT>(A.llr, fN + bc * SZ + r0), A.llr_round16);
                obits.set(bc / Plan::NP, v > CT(0));
                if (A.step_mode) v = (CT)A.st_lam[fN + bc * SZ + r0];
                lds_st<CT>(lds, p4 + (bc * V * ES), v);
            }
        });
        if (A.step_mode) {   // teacher-forced step: LDS holds the given lam, the channel LLRs come from memory
            static_for<0, Plan::NBC>([&](auto bcc) {
                constexpr int bc = decltype(bcc)::value;
                if constexpr (SPLIT_ORIG_REGS && S::oowner(bc) == P) {
                    constexpr int e0 = Rounds<T>::round0_edge(bc);
                    constexpr int os = S::oslot(bc);
                    orig[os] = maybe_round_f16<CT>(load_llr_as<CT, FMT>(A.llr, fN + bc * SZ + ((r0 + T::rot[e0]) % SZ)), A.llr_round16);
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
                orig[os] = lds_ld<CT>(lds + bc * V * ES, qc_wrap(p4 + T::rot[e0] * CPW * ES, vmask));
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
    // Result of the lane's frame: one hard bit per block column the lane's group fills (NBCP of them) + converged flag
    // + the turn it converged at.  Packed into ONE register when it fits (NBCP <= 22: bits 0..21 hard bits, bit 22
    // converged, bits 23..31 the turn -- max_iters <= kSplitMaxIters, the host checks), else bits and flag/turn apart.
    SplitResult<(Plan::NBC + Plan::NP - 1) / Plan::NP> res;
    res.bits = obits;       // until the frame converges: the hard decisions of its channel LLRs (flag clear)
    const int turns = A.step_mode ? 1 : A.max_iters;

    for (int n = 0;; n++) {
        if (done == FULL) break;
        if (A.trace && !((done >> ((p4 / ES) % CPW)) & 1u)) {
            LDPC_COLD_PATH();
            const Where w(p4, A.batch);
            static_for<0, Plan::NBC>([&](auto bcc) {
                constexpr int bc = decltype(bcc)::value;
                if constexpr ((bc % Plan::NP) == P)
                    A.trace[((size_t)w.frame * (A.max_iters + 1) + n) * N + bc * SZ + w.r0] = (double)lds_ld<CT>(lds, p4 + (bc * V * ES));
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
            uint32_t newly = ~fbits & ~done & FULL;  // Orig.hs:69: frames whose syndrome is zero now
            if constexpr (kVetoesNonFinite<CT, VARIANT>) {
                if (newly != 0u) {   // (workgroup-uniform, once per frame) LLRs that left the float range: failed, not "converged" (ldpc_math.h)
                    LDPC_COLD_PATH();
                    bool nf = false;
                    if ((newly >> ((p4 / ES) % CPW)) & 1u)
                        static_for<0, Plan::NBC>([&](auto bcc) {
                            constexpr int bc = decltype(bcc)::value;
                            if constexpr ((bc % Plan::NP) == P) nf |= not_finite(lds_ld<CT>(lds, p4 + (bc * V * ES)));
                        });
                    __syncthreads();   // every wave has read the syndrome flags
                    const unsigned long long vb = __ballot(nf);
                    uint32_t vbits = 0;
#pragma unroll
                    for (int s2 = 0; s2 < CPW; s2++) {
                        unsigned long long m = 0;
                        for (int i = 0; i < 64; i += CPW) m |= 1ull << i;
                        vbits |= ((vb & (m << s2)) != 0ull) ? (1u << s2) : 0u;
                    }
                    if ((tid & 63) == 0) flags[tid >> 6] = vbits;
                    __syncthreads();
                    uint32_t veto = 0;
#pragma unroll
                    for (int w = 0; w < NW; w++) veto |= flags[w];
                    veto = __builtin_amdgcn_readfirstlane(veto) & newly;
                    __syncthreads();   // (the flags are rewritten by the next turn's syndrome)
                    done |= veto;      // stops here as a failure: `res` keeps the channel's hard decisions and a clear flag
                    newly &= ~veto;
                }
            }
            if ((newly >> ((p4 / ES) % CPW)) & 1u) {
                LDPC_COLD_PATH();   // once per frame
                res.converge_at(n);
                static_for<0, Plan::NBC>([&](auto bcc) {
                    constexpr int bc = decltype(bcc)::value;
                    if constexpr ((bc % Plan::NP) == P) {
                        CT v = lds_ld<CT>(lds, p4 + (bc * V * ES));
                        res.bits.set(bc / Plan::NP, v > CT(0));
                    }
                });
                if (A.final_lam) {
                    const Where w(p4, A.batch);
                    static_for<0, Plan::NBC>([&](auto bcc) {
                        constexpr int bc = decltype(bcc)::value;
                        if constexpr ((bc % Plan::NP) == P) A.final_lam[w.fN + bc * SZ + w.r0] = (double)lds_ld<CT>(lds, p4 + (bc * V * ES));
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
            if constexpr ((bc % Plan::NP) == P) A.final_lam[w.fN + bc * SZ + w.r0] = (double)lds_ld<CT>(lds, p4 + (bc * V * ES));
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
    const bool converged = res.converged();
    if (converged) {
        static_for<0, Plan::NBC>([&](auto bcc) {
            constexpr int bc = decltype(bcc)::value;
            if constexpr ((bc % Plan::NP) == P) A.bits[w.fN + bc * SZ + w.r0] = res.bits.get(bc / Plan::NP);
        });
    } else {
        static_for<0, Plan::NBC>([&](auto bcc) {
            constexpr int bc = decltype(bcc)::value;
            if constexpr ((bc % Plan::NP) == P) A.bits[w.fN + bc * SZ + w.r0] = res.bits.get(bc / Plan::NP);   // hard(channel LLR)
        });
        if (A.final_lam) {
            with_llr_format(A.llr_fmt, [&](auto fc) {
                constexpr int FMT = decltype(fc)::value;
                static_for<0, Plan::NBC>([&](auto bcc) {
                    constexpr int bc = decltype(bcc)::value;
                    if constexpr ((bc % Plan::NP) == P) {
                        const size_t gi = w.fN + bc * SZ + w.r0;
                        A.final_lam[gi] = (double)maybe_round_f16<CT>(load_llr_as<CT, FMT>(A.llr, gi), A.llr_round16);
                    }
                });
            });
        }
    }
    if (w.r0 == 0 && P == 0) {
        if (A.iters) A.iters[w.frame] = converged ? res.turn() : turns;
        if (A.conv) A.conv[w.frame] = converged ? 1 : 0;
    }
}

template <class Plan, int SZ> struct SplitGeom {
    static constexpr int CPW = QcGeom<SZ>::CPW, V = QcGeom<SZ>::V, VT = QcGeom<SZ>::VT, THREADS = Plan::NP * VT, NW = THREADS / 64;
    static_assert(THREADS <= 1024, "a frame's wave groups must fit one workgroup");
};

// the kernel proper: every wave group runs its own straight-line program (same loop structure, same barriers)
template <typename CT, int VARIANT, class Plan, int SZ, class T>
__device__ __forceinline__ void split_kernel_body(const FusedArgs &A) {
    using G = SplitGeom<Plan, SZ>;
    static_assert(SZ >= 2, "circulant size");
    __shared__ __attribute__((aligned(16))) char lds[(Plan::NBC * G::V * (int)sizeof(CT) + 15) / 16 * 16 + 4 * G::NW];
    const uint32_t tid = threadIdx.x;
    const uint32_t pair = __builtin_amdgcn_readfirstlane(tid / G::VT);  // wave-uniform (VT is a multiple of 64)
    static_for<0, Plan::NP>([&](auto pc) {
        constexpr int P = decltype(pc)::value;
        if (pair == (uint32_t)P) split_body<CT, VARIANT, Plan, SZ, T, P>(A, lds, tid);
    });
}

}  // namespace ldpc
