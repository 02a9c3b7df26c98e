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
#include <stdio.h>

#include "fused_rows.h"

// LDPC_DBG: timing-only ablation builds (results are WRONG when non-zero; never shipped):
//   1 no barriers between phase-B rounds   2 no phase-B adds   4 no pass 2   8 no lam<-orig init   16 no pass 1
#ifndef LDPC_DBG
#define LDPC_DBG 0
#endif

namespace ldpc {

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
        return maybe_round_f16<CT>(load_llr<CT>(A.llr, gi, A.llr_fmt), A.llr_round16);
    };
    CT orig[Cfg::NORIG];
    if constexpr (kRot) {
        // loaded after the LDS fill below (keeps the prologue's register pressure down)
    } else {
#pragma unroll
        for (int i = 0; i < Cfg::NORIG; i++) orig[i] = llr_at(fN + r0 + (i / RPL) * SZ + RSTEP * (i % RPL));
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
static int launch_msg(hipStream_t st, FusedArgs &a, KernelTimer *timer, LaunchInfo *info) {
    using Cfg = MsgCfg<CT, VARIANT, PlanAR4JA45, SZ>;
    const int grid = (a.batch + Cfg::CPW - 1) / Cfg::CPW;
    if (info && !a.step_mode) {
        snprintf(info->name, sizeof(info->name), "ldpc::fused_msg_kernel<%s, %d, ldpc::PlanAR4JA45, %d, ldpc::%s", sizeof(CT) == 8 ? "double" : "float", VARIANT, SZ,
                 std::is_same<Tab, DynTab>::value ? "DynTab" : "StatTab");
        info->threads = Cfg::THREADS; info->frames_per_wg = Cfg::CPW;
    }
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
    return dtype == LDPC_F32;  // tanh: f32 (ldpc_math.h cn_tanh_f32); f64 tanh stays on the flood path
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

int fused_msg_launch(int variant, int dtype, int sz, int static_id, hipStream_t st, FusedArgs &a, KernelTimer *timer, LaunchInfo *info) {
    // compile-time tables: f32 kernels of the shipped codes
    if (dtype == LDPC_F32 && static_id == 1 && sz == 32)
        return variant == LDPC_MINSUM ? launch_msg<float, LDPC_V_MINSUM, 32, StatTab<TabJpl1024>>(st, a, timer, info)
                                      : launch_msg<float, LDPC_V_TANH, 32, StatTab<TabJpl1024>>(st, a, timer, info);
    if (dtype == LDPC_F32 && static_id == 2 && sz == 128)
        return variant == LDPC_MINSUM ? launch_msg<float, LDPC_V_MINSUM, 128, StatTab<TabJpl4096>>(st, a, timer, info)
                                      : launch_msg<float, LDPC_V_TANH, 128, StatTab<TabJpl4096>>(st, a, timer, info);
#define CASE_SZ(CT, V)                                                   \
    switch (sz) {                                                        \
        case 32: return launch_msg<CT, V, 32, DynTab>(st, a, timer, info);     \
        case 64: return launch_msg<CT, V, 64, DynTab>(st, a, timer, info);     \
        case 128: return launch_msg<CT, V, 128, DynTab>(st, a, timer, info);   \
    }
    if (variant == LDPC_MINSUM && dtype == LDPC_F32) { CASE_SZ(float, LDPC_V_MINSUM) }
    else if (variant == LDPC_MINSUM && dtype == LDPC_F64) { CASE_SZ(double, LDPC_V_MINSUM) }
    else if (variant == LDPC_TANH && dtype == LDPC_F32) { CASE_SZ(float, LDPC_V_TANH) }
#undef CASE_SZ
    return set_error(LDPC_EUNSUPPORTED, "no per-edge-message fused kernel for variant=%d dtype=%d sz=%d", variant, dtype, sz);
}

}  // namespace ldpc
