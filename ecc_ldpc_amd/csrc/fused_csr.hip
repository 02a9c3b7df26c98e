// fused_csr.hip -- on-chip flooding BP for ANY parity-check matrix that fits in LDS (gfx950).
//
// The reference's `Matrix Bool` decoders (Reference/Orig.hs:30-31, Reference/Min.hs:33-34) take any H;
// this is their fused counterpart: one launch decodes the batch, one workgroup (256 threads) owns one
// frame, and lam, the channel LLRs and every message stay in LDS for all iterations.  Used when the
// code has no compiled QC plan (fused_msg.hip) and (2N + DMAX*M) elements fit in 160 KB -- e.g.
// codes/1920.1280.3.303 (BASELINE configs[2]) and codes/moon.7.13; larger codes use flood.hip.
//
// Layout: messages in ELL form, slot(m,k) = k*M + m for the k-th edge of row m (ascending column):
// thread m walks slots m, M+m, 2M+m, .. so a wave touches consecutive dwords (no bank conflicts on the
// message array; the lam gather follows the code's own column pattern and may conflict).
// Graph tables in global memory, read every turn through L1/L2 with coalesced vector loads:
//   ell_col [DMAX][M]  column of slot, -1 = padding
//   csc_slot[CDMAX][N] slots of column n in DESCENDING row order (the reference's foldr, Orig.hs:96), -1 = none
// Turn n (Orig.hs:67-71): rows -> parity of hard(lam) (syndrome) and new messages (ldpc_math.h,
// cn_update_padded: identical arithmetic to the other paths); __syncthreads_or = syndrome verdict;
// columns -> lam = orig + messages.  Two barriers per turn.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "fused.h"
#include "ldpc_math.h"

namespace ldpc {

constexpr int kCsrThreads = 256;

struct CsrArgs {
    const int32_t *ell_col, *csc_slot, *row_ptr;
    const int32_t *row_of_pos, *col_of_pos;   // batched kernel: the row / column stored at an LDS position
    int M, N, E, cdmax;
    const void *llr;
    uint8_t *bits;
    int32_t *iters;
    uint8_t *conv;
    double *final_lam, *trace;
    int batch, max_iters, llr_fmt, llr_round16, step_mode;
    const double *st_lam, *st_ne_in;
    double *st_ne_out;
    uint8_t *st_syn;
    int *work_counter;    // batched kernel, persistent workgroups: next frame to take = gridDim.x + atomicAdd(work_counter, 1)
    const uint32_t *cpack_tab;   // batched kernel, OSH instance: the threads' packed column-slot offsets, [kCpackStride / 4][THREADS][4]
    const uint32_t *rpack_tab;   // ... and packed row-slot offsets, same layout
};

// LDS of the batched kernel (bytes): lam[N] | messages [DMAX][M] | +inf | 0 | next-frame cell | (STAGED instances only:) LLRs of
// the next frame [N] f32, 16-byte aligned | column behind each LDS position [N] u16
__host__ __device__ constexpr uint32_t csrb_off_stage(int N, int M, int dmax) { return (((uint32_t)(N + dmax * M) + 3u) * 4u + 15u) & ~15u; }
__host__ __device__ constexpr uint32_t csrb_lds_bytes(int N, int M, int dmax, bool staged) {
    return staged ? ((csrb_off_stage(N, M, dmax) + (uint32_t)N * 6u + 3u) & ~3u) : ((uint32_t)(N + dmax * M) + 3u) * 4u;
}
#ifdef LDPC_CSR_STAMPS
// debug build only (tools/csr_stamps.py): cycles thread 0 of every workgroup spends in each phase of a frame, summed per WORKGROUP
// in registers and added to these cells once, when the workgroup ends (an atomic per stamp would be the bottleneck it measures)
__device__ unsigned long long g_csr_stamps[8];
#define CSR_STAMP(i) do { if (tid == 0) { const long long now_ = clock64(); acc_[i] += (unsigned long long)(now_ - stamp_); stamp_ = now_; } } while (0)
extern "C" int ldpc_debug_csr_stamps(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_csr_stamps), sizeof(g_csr_stamps)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_csr_stamps), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define CSR_STAMP(i) do {} while (0)
#endif

// RPT / CPT > 0: the thread's column indices (RPT rows x DMAX) and message slots (CPT columns x CDMAX)
// are loaded into registers ONCE before the turn loop; 0: re-read from global memory every turn.
constexpr int kCdMax = 8;  // column degree bound of the register-cached variant
constexpr int kCpackStride = 20;   // dwords per thread in CsrArgs::cpack_tab (2 columns x 9 words, padded to whole 16-byte loads)
// THREADS: 256, or 1024 for frames whose state leaves room for ONE workgroup per CU anyway: 16 waves per CU instead of 4 to cover
// the LDS round trips (codes/1920.1280.A ran here first; it now has a batched instance, below; LDPC_CSR_BATCHED=0 comes back here).
template <typename CT, int VARIANT, int DMAX, int RPT, int CPT, int THREADS = kCsrThreads>
__global__ __launch_bounds__(THREADS) void fused_csr_kernel(CsrArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CT *lam = reinterpret_cast<CT *>(smem);
    CT *orig = lam + A.N;
    CT *msg = orig + A.N;  // [DMAX][M]
    const int tid = threadIdx.x;
    const int frame = blockIdx.x;
    const int M = A.M, N = A.N;
    const size_t fN = (size_t)frame * N, fE = (size_t)frame * A.E;

    for (int n = tid; n < N; n += THREADS) {
        CT v = maybe_round_f16<CT>(load_llr<CT>(A.llr, fN + n, A.llr_fmt), A.llr_round16);
        orig[n] = v;
        lam[n] = A.step_mode ? (CT)A.st_lam[fN + n] : v;
    }
    for (int m = tid; m < M; m += THREADS) {
        const int e0 = A.row_ptr[m], deg = A.row_ptr[m + 1] - e0;
#pragma unroll
        for (int k = 0; k < DMAX; k++) msg[k * M + m] = (A.step_mode && k < deg) ? (CT)A.st_ne_in[fE + e0 + k] : CT(0);  // Orig.hs:64-65
    }
    __syncthreads();

    constexpr bool kCached = RPT > 0;
    int rcol[kCached ? RPT : 1][DMAX];
    int rdeg[kCached ? RPT : 1];
    int cslot[kCached ? CPT : 1][kCdMax];
    if constexpr (kCached) {
#pragma unroll
        for (int i = 0; i < RPT; i++) {
            const int m = tid + i * THREADS;
            rdeg[i] = (m < M) ? A.row_ptr[m + 1] - A.row_ptr[m] : 0;
#pragma unroll
            for (int k = 0; k < DMAX; k++) rcol[i][k] = (m < M) ? A.ell_col[k * M + m] : -1;
        }
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int c = tid + i * THREADS;
#pragma unroll
            for (int j = 0; j < kCdMax; j++) cslot[i][j] = (c < N && j < A.cdmax) ? A.csc_slot[j * N + c] : -1;
        }
    }

    // one row: parity of hard(lam) over its columns (syndrome bit) and, unless `last`, the message update.  All
    // column indices first, then all lam and message reads, then the arithmetic: as one loop per edge the row paid
    // a global-memory round trip (index) plus an LDS round trip per edge.
    auto do_row = [&](int m, int deg, auto colof, bool last) -> int {
        int col[DMAX];
#pragma unroll
        for (int k = 0; k < DMAX; k++) col[k] = colof(k);
        CT l[DMAX], t[DMAX];
#pragma unroll
        for (int k = 0; k < DMAX; k++) l[k] = lam[col[k] < 0 ? 0 : col[k]];
        if (!last) {
#pragma unroll
            for (int k = 0; k < DMAX; k++) t[k] = msg[k * M + m];
        }
        bool par = false;
#pragma unroll
        for (int k = 0; k < DMAX; k++) {
            const bool on = col[k] >= 0;
            par ^= on && (l[k] > CT(0));
            t[k] = on ? (last ? CT(0) : l[k] - t[k]) : CT(INFINITY);
        }
        if (!last) {
            cn_update_padded<CT, VARIANT, DMAX>(t, deg);
#pragma unroll
            for (int k = 0; k < DMAX; k++)
                if (k < deg) msg[k * M + m] = t[k];
        }
        return par ? 1 : 0;
    };

    bool converged = false;
    int n_done = 0;
    const int turns = A.step_mode ? 1 : A.max_iters;
    constexpr bool kResc = kRescales<CT, VARIANT>;   // min-sum in f32: the frame is rescaled by 2^-40 when an LLR passes 2^60 (ldpc_math.h)
    float osc = 1.0f;                                // the factor the channel LLRs enter a column sum with; the frame's LLRs are 2^kexp x lam
    int kexp = 0;
    for (int n = 0;; n++) {
        if (A.trace) {
            LDPC_COLD_PATH();
            for (int c = tid; c < N; c += THREADS) A.trace[((size_t)frame * (A.max_iters + 1) + n) * N + c] = ldexp((double)lam[c], kexp);
        }
        const bool last = n >= turns;
        bool big = false;
        // ---- rows: syndrome + check-node update
        int unsat = 0;
        if constexpr (kCached) {
#pragma unroll
            for (int i = 0; i < RPT; i++) {
                const int m = tid + i * THREADS;
                if (m < M) unsat |= do_row(m, rdeg[i], [&](int k) { return rcol[i][k]; }, last);
            }
        } else {
            for (int m = tid; m < M; m += THREADS)
                unsat |= do_row(m, A.row_ptr[m + 1] - A.row_ptr[m], [&](int k) { return A.ell_col[k * M + m]; }, last);
        }
        const int any_unsat = __syncthreads_or(unsat);  // also: every message written, every lam read
        if (A.step_mode) {
            if (tid == 0) A.st_syn[frame] = any_unsat ? 0 : 1;
        } else if (!any_unsat) {  // Orig.hs:69
            converged = true; n_done = n;
            break;
        }
        if (last) { n_done = n; break; }  // Orig.hs:70
        // ---- columns: lam = foldr (+) orig (column of ne')
        if constexpr (kCached) {
#pragma unroll
            for (int i = 0; i < CPT; i++) {
                const int c = tid + i * THREADS;
                if (c < N) {
                    CT acc = kResc ? orig[c] * (CT)osc : orig[c];
#pragma unroll
                    for (int j = 0; j < kCdMax; j++)
                        if (cslot[i][j] >= 0) acc = msg[cslot[i][j]] + acc;
                    lam[c] = acc;
                    if constexpr (kResc) big |= fabsf((float)acc) > kLamBig;
                }
            }
        } else {
            for (int c = tid; c < N; c += THREADS) {
                CT acc = kResc ? orig[c] * (CT)osc : orig[c];
                if (A.cdmax <= kCdMax) {   // slots first, then the message reads, then the sum (descending rows)
                    int slot[kCdMax];
                    CT v[kCdMax];
#pragma unroll
                    for (int j = 0; j < kCdMax; j++) slot[j] = (j < A.cdmax) ? A.csc_slot[j * N + c] : -1;
#pragma unroll
                    for (int j = 0; j < kCdMax; j++) v[j] = msg[slot[j] < 0 ? 0 : slot[j]];
#pragma unroll
                    for (int j = 0; j < kCdMax; j++)
                        if (slot[j] >= 0) acc = v[j] + acc;
                } else {
                    // heavier columns (codes/1920.1280.A: weight 18), eight edges at a time: slots, then the message reads, then
                    // the additions in the same order as one edge after the other
                    for (int j0 = 0; j0 < A.cdmax; j0 += kCdMax) {
                        int slot[kCdMax];
                        CT v[kCdMax];
#pragma unroll
                        for (int j = 0; j < kCdMax; j++) slot[j] = (j0 + j < A.cdmax) ? A.csc_slot[(j0 + j) * N + c] : -1;
#pragma unroll
                        for (int j = 0; j < kCdMax; j++) v[j] = msg[slot[j] < 0 ? 0 : slot[j]];
#pragma unroll
                        for (int j = 0; j < kCdMax; j++)
                            if (slot[j] >= 0) acc = v[j] + acc;
                    }
                }
                lam[c] = acc;
                if constexpr (kResc) big |= fabsf((float)acc) > kLamBig;
            }
        }
        if constexpr (kResc) {
            if (__syncthreads_or(big ? 1 : 0)) {     // (also the barrier that ends the turn)
                LDPC_COLD_PATH();
                for (int c = tid; c < N; c += THREADS) lam[c] *= (CT)kRescale;
                for (int e = tid; e < DMAX * M; e += THREADS) msg[e] *= (CT)kRescale;
                osc *= kRescale; kexp += kRescaleExp;
                __syncthreads();
            }
        } else __syncthreads();
        if (A.step_mode) break;
    }

    if (A.step_mode) {
        for (int c = tid; c < N; c += THREADS) A.final_lam[fN + c] = ldexp((double)lam[c], kexp);
        for (int m = tid; m < M; m += THREADS) {
            const int e0 = A.row_ptr[m], deg = A.row_ptr[m + 1] - e0;
            for (int k = 0; k < deg; k++) A.st_ne_out[fE + e0 + k] = ldexp((double)msg[k * M + m], kexp);
        }
        return;
    }
    for (int c = tid; c < N; c += THREADS) {
        CT v = converged ? lam[c] : orig[c];
        A.bits[fN + c] = v > CT(0) ? 1 : 0;
        if (A.final_lam) A.final_lam[fN + c] = converged ? ldexp((double)v, kexp) : (double)v;
    }
    if (tid == 0) {
        if (A.iters) A.iters[frame] = n_done;
        if (A.conv) A.conv[frame] = converged ? 1 : 0;
    }
}

// Batched variant for small per-thread shares (RPT rows and CPT columns per thread, column degree <= CD, one
// frame's lam + messages within 64 KB): the thread's graph indices, its rows' messages and its columns' channel
// LLRs live in REGISTERS for the whole decode, and each phase issues ALL of its LDS gathers before it uses any of
// them.  The plain kernel above goes row by row (gather -> wait -> compute -> store) and spends most of its time
// in LDS round trips: r01 profile on codes/1920.1280.3.303 (profiles/r01_csr_*): one VALU instruction per SIMD
// every 7-10 clk, waves waiting 55-64 % of their cycles, LDS only 22-28 % busy.  Here a turn costs three round
// trips (lam gather | message scatter + barrier | message gather) whatever RPT and CPT are.
// Indices are kept as 16-bit LDS BYTE OFFSETS, two per register (that is what keeps the 1920.1280.3.303 instance
// at 128 VGPRs = 4 waves/SIMD), and padding needs no predicate: an absent row slot points at a cell holding +inf
// (t = inf - finite = inf, the neutral element of both rules; its "hard bit" is taken back out of the row parity
// by the known count of padded slots), an absent column slot at a cell holding 0.
// (row weights <= 8: 4 waves/SIMD = 128 VGPRs with a few spilled registers measured 2-4 % faster than 3 waves
//  without; the weight-20 instance needs its 211-256 registers)
// OSH = 2 (r03, codes/1920.1280.A: 146 KB of state, one 1024-thread workgroup per CU): the 16-bit offsets count DWORDS, so that
// they reach the whole 160 KB; its 6 rows x 6 slots and 2 columns x 18 slots per thread are gathered in groups of RG rows / CG
// columns (18 values in flight instead of 36: the instance has to fit 128 VGPRs -- 16 waves on a CU are 4 per SIMD).
template <typename CT, int VARIANT, int DMAX, int RPT, int CPT, int CD, int THREADS, int OSH = 0, bool STAGED = false>
__global__ __launch_bounds__(THREADS, (DMAX <= 8 ? (THREADS > 512 ? 4 : THREADS > 256 ? 6 : 4) : 1)) void fused_csr_batched_kernel(CsrArgs A) {
    static_assert(sizeof(CT) == 4 && DMAX % 2 == 0 && CD % 2 == 0, "pairs of 16-bit offsets");
    constexpr int RG = (RPT * DMAX > 24 && DMAX <= 8) ? (VARIANT == LDPC_V_TANH ? (RPT + 2) / 3 : (RPT + 1) / 2) : RPT;    // rows gathered together
    constexpr int CG = (CPT * CD > 24 && OSH > 0) ? 1 : CPT;                     // columns gathered together
    // the OSH instance does not keep its 18 registers of column-slot offsets across the row phase (with them it spilled 39
    // registers per turn at 128 VGPRs and ran SLOWER than the row-by-row kernel: 44.7 vs 29.9 ms): it re-reads them every turn
    // from a table in memory (72 KB per workgroup, L2-resident), issued before the barrier that ends the row phase
    constexpr bool CP_MEM = OSH > 0;
    // ... and the row offsets likewise (18 registers; tanh still spilled 32 per turn and ran slower than the row-by-row kernel):
    // read for the NEXT turn before the barrier that ends a turn
    constexpr bool RP_MEM = OSH > 0 && VARIANT == LDPC_V_TANH;   // (min-sum: 18.7 ms with the table, 16.6 with 5 spilled registers)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int M = A.M, N = A.N;
    CT *lam = reinterpret_cast<CT *>(smem);
    CT *msg = lam + N;  // [DMAX][M]
    const uint32_t off_inf = (uint32_t)(N + DMAX * M) * 4u, off_zero = off_inf + 4u, off_msg = (uint32_t)N * 4u;
    auto lds_at = [&](uint32_t stored_off) -> CT { return *reinterpret_cast<const CT *>(smem + (stored_off << OSH)); };
    auto load_pack = [&](const uint32_t *tab, auto &dst, auto nwords_c, auto inner_c) {
        constexpr int NW = decltype(nwords_c)::value, INNER = decltype(inner_c)::value;
        // [kCpackStride / 4][THREADS] x 16 bytes: a wave reads 1 KB in one piece; scalar base + one 32-bit lane offset
        // (buffer loads: with flat addresses the compiler spends a 64-bit address register pair per load and spills for it)
        const uint32_t lane_off = (uint32_t)tid * 16u;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(tab), 0, kCpackStride * THREADS * 4, 0x00020000);
#pragma unroll
        for (int w = 0; w < (NW + 3) / 4; w++) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 xv = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, w * THREADS * 16, 0);
            const uint4 x = {xv.x, xv.y, xv.z, xv.w};
            const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int z = 0; z < 4; z++)
                if (4 * w + z < NW) dst[(4 * w + z) / INNER][(4 * w + z) % INNER] = xs[z];
        }
    };

    // ---- this thread's share of the graph: registers, loaded ONCE per workgroup.  The workgroup is persistent: it decodes
    // frames blockIdx.x, blockIdx.x + gridDim.x, ... one after the other (r03: the per-frame prologue -- three dependent
    // global loads deep: position -> row -> row_ptr -> indices -- was ~10 us per frame, 40 % of the launch at 4 dB where a
    // frame takes 4.7 turns; profiles/r03_mackay_f32_tanh_4dB_*).
    uint32_t rpack[RPT][DMAX / 2], cpack[CPT][CD / 2];
    int rdeg_arr[OSH > 0 ? 1 : RPT];
    uint32_t rdeg_pack = 0;                              // OSH instance: the row degrees in one register, 5 bits each
    static_assert(OSH == 0 || (RPT <= 6 && DMAX < 32), "row degrees packed 5 bits each");
    auto rdeg_of = [&](int i) -> int { if constexpr (OSH > 0) return (int)((rdeg_pack >> (5 * i)) & 31u); else return rdeg_arr[i]; };
#pragma unroll
    for (int i = 0; i < RPT; i++) {
        const int m = tid + i * THREADS;                 // a row POSITION; the row behind it:
        const int row = (m < M) ? A.row_of_pos[m] : 0;
        const int e0 = (m < M) ? A.row_ptr[row] : 0;
        const int deg_i = (m < M) ? A.row_ptr[row + 1] - e0 : 0;
        if constexpr (OSH > 0) rdeg_pack |= (uint32_t)deg_i << (5 * i); else rdeg_arr[i] = deg_i;
        if constexpr (!RP_MEM) {
#pragma unroll
        for (int k = 0; k < DMAX; k++) {
            const int col = (m < M) ? A.ell_col[k * M + m] : -1;   // a column POSITION
            const uint32_t off = (col < 0 ? off_inf : (uint32_t)col * 4u) >> OSH;
            if (k & 1) rpack[i][k / 2] |= off << 16; else rpack[i][k / 2] = off;
        }
        }
    }
    if constexpr (!CP_MEM) {
#pragma unroll
    for (int i = 0; i < CPT; i++) {
        const int c = tid + i * THREADS;
#pragma unroll
        for (int j = 0; j < CD; j++) {
            const int slot = (c < N && j < A.cdmax) ? A.csc_slot[j * N + c] : -1;
            const uint32_t off = (slot < 0 ? off_zero : off_msg + (uint32_t)slot * 4u) >> OSH;
            if (j & 1) cpack[i][j / 2] |= off << 16; else cpack[i][j / 2] = off;
        }
    }
    }
    if (tid == 0) {
        *reinterpret_cast<CT *>(smem + off_inf) = CT(INFINITY);
        *reinterpret_cast<CT *>(smem + off_zero) = CT(0);
    }

    // frames are taken from a shared counter, not by a fixed stride: below the waterfall the turns per frame spread from a
    // few to max_iters, and 85 frames per workgroup do not average that out (measured with a fixed stride at 1 dB: +15 % time)
    int *next_frame = reinterpret_cast<int *>(smem + off_zero + 4u);
    // STAGED instances (f32 LLRs of a plain decode, room in LDS): the workgroup holds TWO frames -- the one it decodes and the one
    // it decodes next -- and the next one's LLRs travel HBM -> LDS by LDS-DMA (global_load_lds: no register in between) while the
    // turns of the current one run; the ticket for the frame after that is taken at the start of a frame by the wave with the
    // smallest share of rows and columns; the column permutation stays in LDS.  Nothing of a frame's start then waits for memory.
    // Frames held ahead are frames another workgroup cannot take: over the last kCsrLateZone x gridDim ids of a batch nothing is
    // claimed ahead any more (the ticket is taken when a frame ends and its LLRs are waited for, as the other instances do), or
    // the launch would end with most workgroups idle and a few still owning two frames (measured at 1 dB: +2.7 % time).
    float *stage = reinterpret_cast<float *>(smem + csrb_off_stage(N, M, DMAX));
    uint16_t *cpos = reinterpret_cast<uint16_t *>(stage + N);
    auto stage_frame = [&](int f, int thr) {
        const float *src = reinterpret_cast<const float *>(A.llr) + (size_t)f * N;
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int c = thr + i * THREADS;   // file order: a wave's 64 lanes fetch 256 consecutive bytes, landing at stage[c]
            if (c < N)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + c),
                                                 (__attribute__((address_space(3))) void *)(stage + (c & ~63)), 4, 0, 0);
        }
    };
    constexpr int kCsrLateZone = 3;
    // ONE thread: the id after id `f` for this workgroup, into the next-frame cell; -1 = none yet (`ahead`: asked while f is still to be decoded)
    auto claim = [&](int f, bool ahead) {
        int id = -1;
        if (!A.work_counter) id = f + (int)gridDim.x;
        else if (!ahead || (f >= 0 && f < A.batch - kCsrLateZone * (int)gridDim.x)) id = (int)gridDim.x + atomicAdd(A.work_counter, 1);
        *next_frame = id;
    };
    int frame = blockIdx.x, frame_next = -1;   // frame_next: claimed, its LLRs not asked for yet; -1: nothing held ahead
    if constexpr (STAGED) {
        for (int c = tid; c < N; c += THREADS) cpos[c] = (uint16_t)A.col_of_pos[c];
        if (tid == THREADS - 64) claim(frame, true);
        if (frame < A.batch) stage_frame(frame, tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        frame_next = *next_frame;
    }
#ifdef LDPC_CSR_STAMPS
    long long stamp_ = clock64();
    unsigned long long acc_[4] = {0, 0, 0, 0};
#endif
    while (frame < A.batch) {
    const size_t fN = (size_t)frame * N, fE = (size_t)frame * A.E;
    CT mreg[RPT][DMAX], oreg[CPT];
    int colreg[CPT];   // (re-read per frame, one coalesced load: kept across frames it costs the turn loop four registers)
    if constexpr (!STAGED) {
#pragma unroll
    for (int i = 0; i < CPT; i++) colreg[i] = (tid + i * THREADS < N) ? A.col_of_pos[tid + i * THREADS] : 0;
    }
#pragma unroll
    for (int i = 0; i < RPT; i++)
#pragma unroll
        for (int k = 0; k < DMAX; k++) mreg[i][k] = CT(0);   // Orig.hs:64-65
    if constexpr (STAGED) {
        // (Tried: the thread index made opaque here and in the epilogue, so that the addresses made from it are worked out per frame
        // instead of being hoisted out of the frame loop and spilled across the turn loop: 4 dB +3 % more, but the turn loop the
        // compiler then produced was 2-4 % slower per turn, 1 dB -2.3 %: profiles/r04_csr_stage_ab.txt.  Not kept.)
        const int tf = tid;
        // (the DMA that filled the staging area was waited for before the barrier that ended the previous frame)
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int c = tf + i * THREADS;
            oreg[i] = CT(0);
            if (c < N) {
                oreg[i] = maybe_round_f16<CT>((CT)stage[cpos[c]], A.llr_round16);
                lam[c] = oreg[i];
            }
        }
        __syncthreads();   // the staging area has been read: it is the next frame's now
        if (frame_next >= 0 && frame_next < A.batch) stage_frame(frame_next, tf);
        if (tid == THREADS - 64) claim(frame_next, true);   // the frame after the one held next
    } else {
    if (A.step_mode) {
        LDPC_COLD_PATH();
#pragma unroll
        for (int i = 0; i < RPT; i++) {
            const int m = tid + i * THREADS;
            const int e0 = (m < M) ? A.row_ptr[A.row_of_pos[m]] : 0;
#pragma unroll
            for (int k = 0; k < DMAX; k++)
                if (k < rdeg_of(i)) mreg[i][k] = (CT)A.st_ne_in[fE + e0 + k];
        }
    }
#pragma unroll
    for (int i = 0; i < CPT; i++) {
        const int c = tid + i * THREADS;
        oreg[i] = CT(0);
        if (c < N) {
            oreg[i] = maybe_round_f16<CT>(load_llr<CT>(A.llr, fN + colreg[i], A.llr_fmt), A.llr_round16);
            lam[c] = A.step_mode ? (CT)A.st_lam[fN + colreg[i]] : oreg[i];
        }
    }
    __syncthreads();
    }
    CSR_STAMP(0);

    bool converged = false;
    int n_done = 0;
    const bool step_mode = STAGED ? false : (A.step_mode != 0);   // (STAGED instances: plain decodes only)
    const int turns = step_mode ? 1 : A.max_iters;
    constexpr bool kResc = kRescales<CT, VARIANT>;   // min-sum in f32: the frame is rescaled by 2^-40 when an LLR passes 2^60 (ldpc_math.h)
    float osc = 1.0f;                                // the factor the channel LLRs enter a column sum with; the frame's LLRs are 2^kexp x lam
    int kexp = 0;
    if constexpr (RP_MEM) load_pack(A.rpack_tab, rpack, std::integral_constant<int, RPT * (DMAX / 2)>{}, std::integral_constant<int, DMAX / 2>{});
    for (int n = 0;; n++) {
        LDPC_TURN_LOOP();
        if (A.trace) {   // (never set for a STAGED instance; the test stays: without it the compiler's turn loop came out slower)
            LDPC_COLD_PATH();
            for (int c = tid; c < N; c += THREADS) A.trace[((size_t)frame * (A.max_iters + 1) + n) * N + A.col_of_pos[c]] = ldexp((double)lam[c], kexp);
        }
        const bool last = n >= turns;
        bool big = false;
        // ---- rows: every lam gather of the thread first, then syndrome + check-node updates out of registers
        int unsat = 0;
#pragma unroll
        for (int g = 0; g < RPT; g += RG) {
        CT l[RG][DMAX];
#pragma unroll
        for (int i = g; i < g + RG && i < RPT; i++)
#pragma unroll
            for (int k = 0; k < DMAX; k += 2) {
                l[i - g][k] = lds_at(rpack[i][k / 2] & 0xffffu);
                l[i - g][k + 1] = lds_at(rpack[i][k / 2] >> 16);
            }
#pragma unroll
        for (int i = g; i < g + RG && i < RPT; i++) {
            CT t[DMAX];
            uint32_t par = (uint32_t)(DMAX - rdeg_of(i));   // each padded slot reads +inf: its "hard bit" 1 is taken back out
#pragma unroll
            for (int k = 0; k < DMAX; k++) {
                par ^= (l[i - g][k] > CT(0)) ? 1u : 0u;
                t[k] = l[i - g][k] - mreg[i][k];         // padded slot: inf - finite = inf
            }
            unsat |= (tid + i * THREADS < M) ? (int)(par & 1u) : 0;
            if (!last) {
                cn_update_padded<CT, VARIANT, DMAX>(t, rdeg_of(i));
#pragma unroll
                for (int k = 0; k < DMAX; k++) mreg[i][k] = t[k];   // padded slots: finite, never read by a column
            }
        }
        }
        if (!last) {
            if constexpr (OSH > 0) {
                // slot k of row position tid + i * THREADS: one address register per turn, k * M * 4 from a scalar, i * THREADS * 4 as the
                // instruction's immediate (left to itself the compiler keeps all RPT * DMAX loop-invariant addresses in registers)
                uint32_t mb = off_msg + (uint32_t)tid * 4u;
                asm volatile("" : "+v"(mb));
#pragma unroll
                for (int k = 0; k < DMAX; k++) {
                    const uint32_t kb = mb + (uint32_t)(k * M) * 4u;
#pragma unroll
                    for (int i = 0; i < RPT; i++)
                        if (tid + i * THREADS < M) *reinterpret_cast<CT *>(smem + kb + (uint32_t)(i * THREADS) * 4u) = mreg[i][k];
                }
            } else {
#pragma unroll
            for (int i = 0; i < RPT; i++) {
                const int m = tid + i * THREADS;
                if (m < M) {
#pragma unroll
                    for (int k = 0; k < DMAX; k++) msg[k * M + m] = mreg[i][k];
                }
            }
            }
        }
        if constexpr (CP_MEM) {   // this turn's column-slot offsets: on their way while the workgroup meets at the barrier
#pragma unroll
            for (int i = 0; i < CPT; i++)
#pragma unroll
                for (int jj = 0; jj < CD / 2; jj++) cpack[i][jj] = 0u;
            // 16-byte loads, coalesced across the wave
            static_assert(!CP_MEM || (CPT * (CD / 2) <= kCpackStride && RPT * (DMAX / 2) <= kCpackStride), "table stride");
            load_pack(A.cpack_tab, cpack, std::integral_constant<int, CPT * (CD / 2)>{}, std::integral_constant<int, CD / 2>{});
        }
        const int any_unsat = __syncthreads_or(unsat);  // also: every message written, every lam read
        if (step_mode) {
            if (tid == 0) A.st_syn[frame] = any_unsat ? 0 : 1;
        } else if (!any_unsat) {  // Orig.hs:69
            converged = true; n_done = n;
            break;
        }
        if (last) { n_done = n; break; }  // Orig.hs:70
        // ---- columns: every message gather first, then lam = foldr (+) orig (column of ne'), descending rows
        // (the NEXT turn's row offsets first: they arrive while the columns are summed)
        if constexpr (RP_MEM) load_pack(A.rpack_tab, rpack, std::integral_constant<int, RPT * (DMAX / 2)>{}, std::integral_constant<int, DMAX / 2>{});
#pragma unroll
        for (int g = 0; g < CPT; g += CG) {
        CT v[CG][CD];
#pragma unroll
        for (int i = g; i < g + CG && i < CPT; i++)
#pragma unroll
            for (int j = 0; j < CD; j += 2) {
                v[i - g][j] = lds_at(cpack[i][j / 2] & 0xffffu);
                v[i - g][j + 1] = lds_at(cpack[i][j / 2] >> 16);
            }
#pragma unroll
        for (int i = g; i < g + CG && i < CPT; i++) {
            const int c = tid + i * THREADS;
            CT acc = kResc ? oreg[i] * (CT)osc : oreg[i];
#pragma unroll
            for (int j = 0; j < CD; j++) acc = v[i - g][j] + acc;   // absent slots add 0 (they follow the present ones)
            if (c < N) lam[c] = acc;
            if constexpr (kResc) big |= c < N && fabsf((float)acc) > kLamBig;
        }
        }
        if constexpr (kResc) {
            if (__syncthreads_or(big ? 1 : 0)) {     // (also the barrier that ends the turn)
                LDPC_COLD_PATH();
#pragma unroll
                for (int i = 0; i < CPT; i++)
                    if (tid + i * THREADS < N) lam[tid + i * THREADS] *= (CT)kRescale;       // this thread's own columns
#pragma unroll
                for (int i = 0; i < RPT; i++)
#pragma unroll
                    for (int k = 0; k < DMAX; k++) mreg[i][k] *= (CT)kRescale;               // (their LDS copies are rewritten by the next row phase)
                osc *= kRescale; kexp += kRescaleExp;
                __syncthreads();
            }
        } else __syncthreads();
        if (step_mode) break;
    }
    CSR_STAMP(1);

    if (step_mode) {
        for (int c = tid; c < N; c += THREADS) A.final_lam[fN + A.col_of_pos[c]] = ldexp((double)lam[c], kexp);
#pragma unroll
        for (int i = 0; i < RPT; i++) {
            const int m = tid + i * THREADS;
            if (m < M) {
                const int e0 = A.row_ptr[A.row_of_pos[m]];
#pragma unroll
                for (int k = 0; k < DMAX; k++)
                    if (k < rdeg_of(i)) A.st_ne_out[fE + e0 + k] = ldexp((double)mreg[i][k], kexp);
            }
        }
    } else {
        const int te = tid;
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int c = te + i * THREADS;
            if (c < N) {
                const int col = STAGED ? (int)cpos[c] : A.col_of_pos[c];
                CT vv = converged ? lam[c] : oreg[i];
                A.bits[fN + col] = vv > CT(0) ? 1 : 0;
                if (A.final_lam) A.final_lam[fN + col] = converged ? ldexp((double)vv, kexp) : (double)vv;
            }
        }
        if (tid == 0) {
            if (A.iters) A.iters[frame] = n_done;
            if (A.conv) A.conv[frame] = converged ? 1 : 0;
        }
    }
    CSR_STAMP(2);
    if constexpr (STAGED) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's share of the next frame's LLRs has landed
        __syncthreads();   // ... and everybody's; also: every lam read of this frame is done before the next frame's LLRs are written over it
        if (frame_next >= 0) { frame = frame_next; frame_next = *next_frame; }
        else {             // the end of the batch: nothing was held ahead
            if (tid == THREADS - 64) claim(frame, false);
            __syncthreads();
            frame = *next_frame;
            if (frame < A.batch) stage_frame(frame, tid);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else {
    if (tid == 0) *next_frame = A.work_counter ? (int)gridDim.x + atomicAdd(A.work_counter, 1) : frame + (int)gridDim.x;
    __syncthreads();   // also: every lam read of this frame is done before the next frame's LLRs are written over it
    frame = *next_frame;
    }
    CSR_STAMP(3);
    }
#ifdef LDPC_CSR_STAMPS
    if (tid == 0) for (int i = 0; i < 4; i++) atomicAdd(&g_csr_stamps[i], acc_[i]);
#endif
}

// ------------------------------------------------------------------ host side
struct CsrState {
    const void *attr_kern = nullptr; size_t attr_lds = 0;      // the kernel whose dynamic-LDS limit this context has raised, and to what
    const void *stage_kern = nullptr; bool stage_ok = false;   // batched kernel: whether the LLR staging area fits without costing a resident workgroup
    int variant = 0, dtype = 0, M = 0, N = 0, E = 0, dmax = 0, cdmax = 0, round16 = 0;
    bool want_batched = true, want_cache = true, want_wide = true;   // A/B switches, read from the environment once, at creation
    int32_t *d_ell = nullptr, *d_csc = nullptr, *d_row_ptr = nullptr;
    // batched kernel: the same tables in POSITION space (conflict-aware placement of rows and columns in LDS)
    int32_t *d_ell_b = nullptr, *d_csc_b = nullptr, *d_row_of_pos = nullptr, *d_col_of_pos = nullptr;
    int *d_counter = nullptr;     // work counter of the persistent batched kernel
    int resident = 0, resident_dev = -1;   // workgroups of the batched kernel resident at once on resident_dev with resident_lds bytes of LDS
    size_t resident_lds = 0;
    const void *resident_kern = nullptr;
    uint32_t *d_cpack = nullptr, *d_rpack = nullptr;  // OSH instance: packed column-slot / row-slot offsets per thread
    KernelTimer *timer = nullptr;
    LaunchInfo info;
};

static int pick_dmax(int maxdeg) { return maxdeg <= 4 ? 4 : maxdeg <= 6 ? 6 : maxdeg <= 8 ? 8 : maxdeg <= 20 ? 20 : maxdeg <= 32 ? 32 : 0; }
static size_t csr_lds_bytes(const ldpc_code &c, int dtype) {
    const int dm = pick_dmax(c.max_row_deg);
    return ((size_t)2 * c.N + (size_t)dm * c.M) * (dtype == LDPC_F64 ? 8 : 4);
}

// ---- conflict-aware placement for the batched kernel ---------------------------------------------------------
// ds_read_b32 serves a wave as two 32-lane groups and banks on (dword index) mod 32 (MI355X_MICROARCH.md, LDS).  In
// the batched kernel the lanes of a group are 32 consecutive row POSITIONS reading the lam cells of their k-th
// columns (phase A), or 32 consecutive column POSITIONS reading the message cells of their j-th rows (phase B);
// with rows and columns stored in file order a random code puts ~3.5 addresses on the busiest bank of a group
// (r01 profile: 48 % of the LDS cycles were conflict cycles and LDS was 61 % busy).  Which row or column sits at
// which position is free -- the arithmetic order inside a row or column is fixed by the ORIGINAL indices -- so the
// host searches for positions with few collisions: random swaps of two rows or two columns under a short annealing
// schedule.  Deterministic (fixed seed).  codes/1920.1280.3.303: 679 -> ~330 extra LDS cycles per turn, tanh
// 3.9 -> 4.15 Gbit/s, min-sum 4.4 -> 5.0 (LDPC_CSR_PLACE=0 keeps file order).
struct Placement {
    std::vector<int32_t> row_pos, col_pos, row_of_pos, col_of_pos;
    long cost_before = 0, cost_after = 0;
};

static Placement place_for_banks(const ldpc_code &c, int dmax, bool optimise) {
    const int M = c.M, N = c.N;
    Placement P;
    P.row_pos.resize(M); P.row_of_pos.resize(M); P.col_pos.resize(N); P.col_of_pos.resize(N);
    for (int i = 0; i < M; i++) P.row_pos[i] = P.row_of_pos[i] = i;
    for (int i = 0; i < N; i++) P.col_pos[i] = P.col_of_pos[i] = i;
    // edge lists: row r -> (k, col); column c -> (j, row, k) with j counted in DESCENDING row order
    struct CE { int row, k; };
    std::vector<std::vector<CE>> cedges(N);
    for (int n = 0; n < N; n++)
        for (int q = c.col_ptr[n + 1] - 1; q >= c.col_ptr[n]; q--) {
            const int e = c.csc_edge[q];
            const int r = (int)(std::upper_bound(c.row_ptr.begin(), c.row_ptr.end(), e) - c.row_ptr.begin()) - 1;
            cedges[n].push_back({r, e - c.row_ptr[r]});
        }
    auto costA = [&](int g, int k) {   // row group g, edge slot k
        int cnt[32] = {0}, mx = 0;
        for (int p = 32 * g; p < std::min(32 * g + 32, M); p++) {
            const int r = P.row_of_pos[p];
            if (k < c.row_ptr[r + 1] - c.row_ptr[r]) mx = std::max(mx, ++cnt[P.col_pos[c.col_idx[c.row_ptr[r] + k]] & 31]);
        }
        return mx > 0 ? mx - 1 : 0;
    };
    auto costB = [&](int g, int j) {   // column group g, edge slot j
        int cnt[32] = {0}, mx = 0;
        for (int q = 32 * g; q < std::min(32 * g + 32, N); q++) {
            const auto &ce = cedges[P.col_of_pos[q]];
            if (j < (int)ce.size()) mx = std::max(mx, ++cnt[(N + ce[j].k * M + P.row_pos[ce[j].row]) & 31]);
        }
        return mx > 0 ? mx - 1 : 0;
    };
    auto total = [&]() {
        long t = 0;
        for (int g = 0; g < (M + 31) / 32; g++) for (int k = 0; k < dmax; k++) t += costA(g, k);
        for (int g = 0; g < (N + 31) / 32; g++) for (int j = 0; j < c.max_col_deg; j++) t += costB(g, j);
        return t;
    };
    P.cost_before = P.cost_after = total();
    if (!optimise || M < 64 || N < 64) return P;
    // search objective: colliding PAIRS per group (sum over banks of cnt*(cnt-1)/2) -- it moves with every single
    // collision, unlike the busiest-bank count, and updates in O(1) per element
    const int GA = (M + 31) / 32, GB = (N + 31) / 32, CDm = std::max(c.max_col_deg, 1);
    std::vector<int16_t> cntA((size_t)GA * dmax * 32, 0), cntB((size_t)GB * CDm * 32, 0);
    long pairs = 0;
    // element (row r, slot k) of phase A; element (column n, slot j) of phase B
    auto a_cell = [&](int r, int k) -> int16_t & { return cntA[((size_t)(P.row_pos[r] / 32) * dmax + k) * 32 + (P.col_pos[c.col_idx[c.row_ptr[r] + k]] & 31)]; };
    auto b_cell = [&](int n, int j) -> int16_t & { const CE &ce = cedges[n][j]; return cntB[((size_t)(P.col_pos[n] / 32) * CDm + j) * 32 + ((N + ce.k * M + P.row_pos[ce.row]) & 31)]; };
    auto add_a = [&](int r, int k) { pairs += a_cell(r, k)++; };
    auto del_a = [&](int r, int k) { pairs -= --a_cell(r, k); };
    auto add_b = [&](int n, int j) { pairs += b_cell(n, j)++; };
    auto del_b = [&](int n, int j) { pairs -= --b_cell(n, j); };
    // j index of row r inside column n's list
    auto j_of = [&](int n, int r) { const auto &ce = cedges[n]; for (int j = 0; j < (int)ce.size(); j++) if (ce[j].row == r) return j; return -1; };
    for (int r = 0; r < M; r++) for (int k = 0; k < c.row_ptr[r + 1] - c.row_ptr[r]; k++) add_a(r, k);
    for (int n = 0; n < N; n++) for (int j = 0; j < (int)cedges[n].size(); j++) add_b(n, j);
    auto touch_row = [&](int r, bool add) {   // everything whose cell depends on row r's position
        for (int e = c.row_ptr[r]; e < c.row_ptr[r + 1]; e++) {
            const int k = e - c.row_ptr[r], n = c.col_idx[e], j = j_of(n, r);
            if (add) { add_a(r, k); add_b(n, j); } else { del_a(r, k); del_b(n, j); }
        }
    };
    auto touch_col = [&](int n, bool add) {   // everything whose cell depends on column n's position
        for (int j = 0; j < (int)cedges[n].size(); j++) {
            const CE &ce = cedges[n][j];
            if (add) { add_b(n, j); add_a(ce.row, ce.k); } else { del_b(n, j); del_a(ce.row, ce.k); }
        }
    };
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    auto rnd = [&](int n) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (int)(rng % (uint64_t)n); };
    const char *mm = getenv("LDPC_CSR_PLACE_MOVES");   // search length multiplier (experiments)
    const long moves = 100L * (M + N) * std::max(1, mm ? atoi(mm) : 1);   // 1920.1280.3.303: 679 -> ~330 extra cycles; 4x more moves end at the same cost
    for (long it = 0; it < moves; it++) {
        // annealing: a worsening by d pairs is accepted with probability 2^-(d / T), T falling linearly to 0
        const double T = 0.6 * (1.0 - (double)it / (double)moves);
        const long before = pairs;
        const bool rows = rnd(M + N) < M;
        const int x = rnd(rows ? M : N), y = rnd(rows ? M : N);
        if (x == y) continue;
        // two rows (columns) sharing a column (row) would be touched twice: skip those rare pairs
        bool share = false;
        if (rows) { for (int e = c.row_ptr[x]; e < c.row_ptr[x + 1] && !share; e++) share = j_of(c.col_idx[e], y) >= 0; }
        else { for (auto &ce : cedges[x]) if (j_of(y, ce.row) >= 0) { share = true; break; } }
        if (share) continue;
        auto apply = [&]() {
            if (rows) {
                touch_row(x, false); touch_row(y, false);
                std::swap(P.row_pos[x], P.row_pos[y]);
                touch_row(x, true); touch_row(y, true);
            } else {
                touch_col(x, false); touch_col(y, false);
                std::swap(P.col_pos[x], P.col_pos[y]);
                touch_col(x, true); touch_col(y, true);
            }
        };
        apply();
        const long d = pairs - before;
        if (d > 0) {
            const double u = (double)(rnd(1 << 20) + 1) / (double)(1 << 20);
            if (T <= 0.0 || u > exp2(-(double)d / T)) apply();   // undo (the swap is its own inverse)
        }
    }
    for (int r = 0; r < M; r++) P.row_of_pos[P.row_pos[r]] = r;
    for (int n = 0; n < N; n++) P.col_of_pos[P.col_pos[n]] = n;
    P.cost_after = total();
    return P;
}

const char *fused_csr_why_not(const ldpc_code &c, int variant, int dtype) {
    if (dtype != LDPC_F32 && dtype != LDPC_F64) return "the generic on-chip kernel exists for f32 and f64";
    if (pick_dmax(c.max_row_deg) == 0) return "a check row has more than 32 edges";
    if (variant == LDPC_TANH && dtype == LDPC_F64 && c.max_row_deg > 8) return "f64 tanh rows above degree 8 stay on the flood path";
    if (csr_lds_bytes(c, dtype) > 160 * 1024) return "lam + LLRs + messages of one frame exceed the 160 KB of LDS";
    return nullptr;
}

void fused_csr_destroy(CsrState *s) {
    if (!s) return;
    (void)hipFree(s->d_ell); (void)hipFree(s->d_csc); (void)hipFree(s->d_row_ptr);
    (void)hipFree(s->d_ell_b); (void)hipFree(s->d_csc_b); (void)hipFree(s->d_row_of_pos); (void)hipFree(s->d_col_of_pos); (void)hipFree(s->d_counter); (void)hipFree(s->d_cpack); (void)hipFree(s->d_rpack);
    delete s;
}

// which batched instance (if any) serves this shape: 0 = none (row-by-row kernel)
static int batched_shape(const CsrState &s) {
    const int rpt = (s.M + kCsrThreads - 1) / kCsrThreads, cpt = (s.N + kCsrThreads - 1) / kCsrThreads;
    const size_t cells = (size_t)s.N + (size_t)s.dmax * s.M + 3;
    // r03, codes/1920.1280.A (5760 x 1920, row weights 4 / 6, column weights 14 / 18; 146 KB): one 1024-thread workgroup per
    // frame and CU, 16-bit DWORD offsets
    if (cells * 4 > 65536 && cells * 4 <= 160 * 1024 && cells <= 65536 && s.dmax == 6 && (s.M + 1023) / 1024 <= 6 && (s.N + 1023) / 1024 <= 2 &&
        s.cdmax <= 18 && s.want_wide)
        return 4;
    if (cells * 4 > 65536) return 0;   // 16-bit LDS byte offsets
    if (s.dmax == 4 && rpt <= 6 && cpt <= 8 && s.cdmax <= 4) return 1;
    if (s.dmax == 8 && rpt <= 2 && cpt <= 4 && s.cdmax <= 8) return 2;
    if (s.dmax == 20 && rpt <= 2 && cpt <= 6 && s.cdmax <= 8) return 3;
    return 0;
}

CsrState *fused_csr_create(const ldpc_code &c, int variant, int dtype) {
    const char *why = fused_csr_why_not(c, variant, dtype);
    if (why) { set_error(LDPC_EUNSUPPORTED, "%s", why); return nullptr; }
    CsrState *s = new (std::nothrow) CsrState();
    if (!s) { set_error(LDPC_ENOMEM, "out of host memory"); return nullptr; }
    s->variant = variant; s->dtype = dtype; s->M = c.M; s->N = c.N; s->E = c.E;
    s->dmax = pick_dmax(c.max_row_deg); s->cdmax = c.max_col_deg;
    {   // LDPC_CSR_BATCHED=0: row-by-row kernel; LDPC_CSR_NOCACHE: its indices re-read from memory every turn
        const char *bz = getenv("LDPC_CSR_BATCHED");
        s->want_batched = !(bz && !strcmp(bz, "0"));
        s->want_cache = getenv("LDPC_CSR_NOCACHE") == nullptr;
        const char *wz = getenv("LDPC_CSR_WIDE");   // LDPC_CSR_WIDE=0: 256 threads per frame even where 512 fit
        s->want_wide = !(wz && !strcmp(wz, "0"));
    }
    std::vector<int32_t> ell((size_t)s->dmax * c.M, -1), csc((size_t)std::max(s->cdmax, 1) * c.N, -1), slot_of_edge((size_t)c.E);
    for (int m = 0; m < c.M; m++)
        for (int e = c.row_ptr[m]; e < c.row_ptr[m + 1]; e++) {
            int k = e - c.row_ptr[m];
            ell[(size_t)k * c.M + m] = c.col_idx[e];
            slot_of_edge[e] = k * c.M + m;
        }
    for (int n = 0; n < c.N; n++) {
        int j = 0;
        for (int q = c.col_ptr[n + 1] - 1; q >= c.col_ptr[n]; q--, j++) csc[(size_t)j * c.N + n] = slot_of_edge[c.csc_edge[q]];  // descending row
    }
    auto up = [&](int32_t **dst, const std::vector<int32_t> &v) {
        hipError_t e = hipMalloc((void **)dst, v.size() * 4);
        if (e == hipSuccess) e = hipMemcpy(*dst, v.data(), v.size() * 4, hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up(&s->d_ell, ell);
    if (e == hipSuccess) e = up(&s->d_csc, csc);
    if (e == hipSuccess) e = up(&s->d_row_ptr, c.row_ptr);
    if (e == hipSuccess && dtype == LDPC_F32 && batched_shape(*s)) {   // position-space tables of the batched kernel
        const char *pz = getenv("LDPC_CSR_PLACE");   // LDPC_CSR_PLACE=0: file order (A/B measurements)
        const Placement P = place_for_banks(c, s->dmax, !(pz && !strcmp(pz, "0")));
        if (getenv("LDPC_CSR_PLACE_VERBOSE"))
            fprintf(stderr, "[fused_csr] placement: extra LDS cycles per turn and workgroup %ld -> %ld\n", P.cost_before, P.cost_after);
        std::vector<int32_t> ell_b((size_t)s->dmax * c.M, -1), csc_b((size_t)std::max(s->cdmax, 1) * c.N, -1);
        for (int pp = 0; pp < c.M; pp++) {
            const int r = P.row_of_pos[pp];
            for (int ee = c.row_ptr[r]; ee < c.row_ptr[r + 1]; ee++) ell_b[(size_t)(ee - c.row_ptr[r]) * c.M + pp] = P.col_pos[c.col_idx[ee]];
        }
        for (int q = 0; q < c.N; q++) {
            const int n = P.col_of_pos[q];
            int j = 0;
            for (int qq = c.col_ptr[n + 1] - 1; qq >= c.col_ptr[n]; qq--, j++) {   // descending row
                const int ee = c.csc_edge[qq];
                const int r = (int)(std::upper_bound(c.row_ptr.begin(), c.row_ptr.end(), ee) - c.row_ptr.begin()) - 1;
                csc_b[(size_t)j * c.N + q] = (ee - c.row_ptr[r]) * c.M + P.row_pos[r];
            }
        }
        e = up(&s->d_ell_b, ell_b);
        if (e == hipSuccess) e = up(&s->d_csc_b, csc_b);
        if (e == hipSuccess && batched_shape(*s) == 4) {   // [kCpackStride / 4][1024 threads][4 words]: CPT = 2 columns x CD / 2 = 9 words per thread, offsets in dwords (OSH = 2)
            constexpr int T = 1024, CPT = 2, CD = 18;
            const uint32_t off_msg = (uint32_t)c.N, off_zero = (uint32_t)(c.N + s->dmax * c.M) + 1u;
            std::vector<int32_t> cp((size_t)T * kCpackStride, 0);
            for (int i = 0; i < CPT; i++)
                for (int t = 0; t < T; t++) {
                    const int q = t + i * T;
                    for (int j = 0; j < CD; j++) {
                        const int slot = (q < c.N && j < s->cdmax) ? csc_b[(size_t)j * c.N + q] : -1;
                        const uint32_t off = slot < 0 ? off_zero : off_msg + (uint32_t)slot;
                        const int word = i * (CD / 2) + j / 2;
                        int32_t &w = cp[((size_t)(word / 4) * T + t) * 4 + word % 4];
                        w = (int32_t)((j & 1) ? ((uint32_t)w | (off << 16)) : off);
                    }
                }
            e = up((int32_t **)&s->d_cpack, cp);
            constexpr int RPT = 6;
            const uint32_t off_inf = (uint32_t)(c.N + s->dmax * c.M);
            std::vector<int32_t> rp((size_t)T * kCpackStride, 0);
            for (int i = 0; i < RPT; i++)
                for (int t = 0; t < T; t++) {
                    const int m = t + i * T;
                    for (int k = 0; k < s->dmax; k++) {
                        const int col = m < c.M ? ell_b[(size_t)k * c.M + m] : -1;
                        const uint32_t off = col < 0 ? off_inf : (uint32_t)col;
                        const int word = i * (s->dmax / 2) + k / 2;
                        int32_t &w = rp[((size_t)(word / 4) * T + t) * 4 + word % 4];
                        w = (int32_t)((k & 1) ? ((uint32_t)w | (off << 16)) : off);
                    }
                }
            if (e == hipSuccess) e = up((int32_t **)&s->d_rpack, rp);
        }
        if (e == hipSuccess) e = up(&s->d_row_of_pos, P.row_of_pos);
        if (e == hipSuccess) e = up(&s->d_col_of_pos, P.col_of_pos);
    }
    if (e != hipSuccess) { set_error(LDPC_EHIP, "fused_csr_create: %s", hipGetErrorString(e)); fused_csr_destroy(s); return nullptr; }
    return s;
}

void fused_csr_set_timer(CsrState *s, KernelTimer *t) { if (s) s->timer = t; }
const LaunchInfo &fused_csr_launch_info(const CsrState &s) { return s.info; }
const char *fused_csr_kernel_name(const CsrState &s) {
    return (s.dtype == LDPC_F32 && s.d_ell_b && s.want_batched && batched_shape(s)) ? "fused_csr_batched_kernel" : "fused_csr_kernel";
}
void fused_csr_set_round16(CsrState *s, int on) { if (s) s->round16 = on; }

template <typename CT, int VARIANT, int DMAX, int RPT, int CPT, int THREADS = kCsrThreads>
static int launch_csr(CsrState &s, hipStream_t st, CsrArgs &a) {
    auto kern = fused_csr_kernel<CT, VARIANT, DMAX, RPT, CPT, THREADS>;
    const size_t lds = ((size_t)2 * s.N + (size_t)DMAX * s.M) * sizeof(CT);
    // (remembered per context, not per process: the attribute belongs to the function ON ONE DEVICE, and a process may hold contexts
    //  on several -- ecc-ldpc-hip -d0,1,.. runs one thread per GPU)
    if (lds > 64 * 1024 && (s.attr_kern != (const void *)kern || lds > s.attr_lds)) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return set_error(LDPC_EHIP, "hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
        s.attr_kern = (const void *)kern; s.attr_lds = lds;
    }
    if (!a.step_mode) {
        snprintf(s.info.name, sizeof(s.info.name), "ldpc::fused_csr_kernel<%s, %d, %d, %d, %d, %d>", sizeof(CT) == 8 ? "double" : "float", VARIANT, DMAX, RPT, CPT, THREADS);
        s.info.threads = THREADS; s.info.frames_per_wg = 1;
    }
    if (s.timer && !a.step_mode) s.timer->begin(st);
    hipLaunchKernelGGL(kern, dim3(a.batch), dim3(THREADS), lds, st, a);
    if (s.timer && !a.step_mode) s.timer->end(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(LDPC_EHIP, "fused_csr launch: %s", hipGetErrorString(e));
    return LDPC_OK;
}

template <typename CT, int VARIANT, int DMAX, int RPT, int CPT, int CD, int THREADS, int OSH, bool STAGED>
static int launch_csr_batched_as(CsrState &s, hipStream_t st, CsrArgs &a);

template <typename CT, int VARIANT, int DMAX, int RPT, int CPT, int CD, int THREADS = kCsrThreads, int OSH = 0>
static int launch_csr_batched(CsrState &s, hipStream_t st, CsrArgs &a) {
    // the STAGED instance (the next frame's LLRs on their way into LDS while this one is decoded: see the kernel) for f32 LLRs of a
    // plain decode, where its 6 N bytes of LDS do not cost a resident workgroup (LDPC_CSR_STAGE=0: never, A/B)
    const char *sz = getenv("LDPC_CSR_STAGE");
    bool staged = OSH == 0 && !a.step_mode && !a.trace && a.llr_fmt == LLR_F32 && s.N < 65536 && !(sz && !strcmp(sz, "0")) &&
                  csrb_lds_bytes(s.N, s.M, DMAX, true) <= 160u * 1024u;   // (OSH instances: frames of milliseconds, LDS full)
    if constexpr (OSH > 0) return launch_csr_batched_as<CT, VARIANT, DMAX, RPT, CPT, CD, THREADS, OSH, false>(s, st, a);
    else {
    if (staged) {
        const void *k1 = (const void *)fused_csr_batched_kernel<CT, VARIANT, DMAX, RPT, CPT, CD, THREADS, OSH, true>;
        const void *k0 = (const void *)fused_csr_batched_kernel<CT, VARIANT, DMAX, RPT, CPT, CD, THREADS, OSH, false>;
        if (s.stage_kern != k1) {
            int with = 0, without = 0;
            const size_t l1 = csrb_lds_bytes(s.N, s.M, DMAX, true), l0 = csrb_lds_bytes(s.N, s.M, DMAX, false);
            if (l1 > 64 * 1024) (void)hipFuncSetAttribute(k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l1);
            if (l0 > 64 * 1024) (void)hipFuncSetAttribute(k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l0);
            s.stage_ok = hipOccupancyMaxActiveBlocksPerMultiprocessor(&with, k1, THREADS, l1) == hipSuccess &&
                         hipOccupancyMaxActiveBlocksPerMultiprocessor(&without, k0, THREADS, l0) == hipSuccess && with >= without && with >= 1;
            (void)hipGetLastError();
            s.stage_kern = k1;
        }
        staged = s.stage_ok;
    }
    return staged ? launch_csr_batched_as<CT, VARIANT, DMAX, RPT, CPT, CD, THREADS, OSH, true>(s, st, a)
                  : launch_csr_batched_as<CT, VARIANT, DMAX, RPT, CPT, CD, THREADS, OSH, false>(s, st, a);
    }
}

template <typename CT, int VARIANT, int DMAX, int RPT, int CPT, int CD, int THREADS, int OSH, bool STAGED>
static int launch_csr_batched_as(CsrState &s, hipStream_t st, CsrArgs &a) {
    auto kern = fused_csr_batched_kernel<CT, VARIANT, DMAX, RPT, CPT, CD, THREADS, OSH, STAGED>;
    const size_t lds = csrb_lds_bytes(s.N, s.M, DMAX, STAGED);   // lam, messages, the +inf and 0 cells, the next-frame cell (+ the staging area)
    // (remembered per context, not per process: the attribute belongs to the function ON ONE DEVICE, and a process may hold contexts
    //  on several -- ecc-ldpc-hip -d0,1,.. runs one thread per GPU)
    if (lds > 64 * 1024 && (s.attr_kern != (const void *)kern || lds > s.attr_lds)) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return set_error(LDPC_EHIP, "hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
        s.attr_kern = (const void *)kern; s.attr_lds = lds;
    }
    if (!a.step_mode) {
        // (all eight template arguments, as the assembly and rocprofv3 list the instance: bench.py finds its static instruction count by this name)
        snprintf(s.info.name, sizeof(s.info.name), "ldpc::fused_csr_batched_kernel<%s, %d, %d, %d, %d, %d, %d, %d, %s>", sizeof(CT) == 8 ? "double" : "float", VARIANT, DMAX, RPT, CPT, CD, THREADS, OSH, STAGED ? "true" : "false");
        s.info.threads = THREADS; s.info.frames_per_wg = 1;
    }
    // persistent workgroups: as many as are resident at once (LDPC_CSR_PERSIST=0: one workgroup per frame)
    // (kept in the context's state: it depends on this graph's LDS size and on the device the context lives on)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!s.resident || s.resident_dev != dev || s.resident_lds != lds || s.resident_kern != (const void *)kern) {
        int per_cu = 0;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)kern, THREADS, lds) == hipSuccess && per_cu > 0)
            s.resident = per_cu * prop.multiProcessorCount;
        else { (void)hipGetLastError(); s.resident = 256 * 2; }
        s.resident_dev = dev; s.resident_lds = lds; s.resident_kern = (const void *)kern;
    }
    const char *pz = getenv("LDPC_CSR_PERSIST");
    const int grid = (pz && !strcmp(pz, "0")) ? a.batch : std::min(a.batch, s.resident);
    a.work_counter = nullptr;
    if (grid < a.batch) {
        if (!s.d_counter && hipMalloc((void **)&s.d_counter, sizeof(int)) != hipSuccess) { (void)hipGetLastError(); s.d_counter = nullptr; }
        const char *dz = getenv("LDPC_CSR_DYNAMIC");   // =0: fixed stride (A/B)
        if (s.d_counter && !(dz && !strcmp(dz, "0")) && hipMemsetAsync(s.d_counter, 0, sizeof(int), st) == hipSuccess) a.work_counter = s.d_counter;
    }
    if (s.timer && !a.step_mode) s.timer->begin(st);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds, st, a);
    if (s.timer && !a.step_mode) s.timer->end(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(LDPC_EHIP, "fused_csr (batched) launch: %s", hipGetErrorString(e));
    return LDPC_OK;
}

template <typename CT, int VARIANT>
static int dispatch_dmax(CsrState &s, hipStream_t st, CsrArgs &a) {
    // small per-thread shares (f32): the batched kernel; LDPC_CSR_BATCHED=0 keeps the row-by-row kernel (A/B)
    if constexpr (sizeof(CT) == 4) {
        const int shape = (s.d_ell_b && s.want_batched) ? batched_shape(s) : 0;
        if (shape) {
            a.ell_col = s.d_ell_b; a.csc_slot = s.d_csc_b; a.row_of_pos = s.d_row_of_pos; a.col_of_pos = s.d_col_of_pos; a.cpack_tab = s.d_cpack; a.rpack_tab = s.d_rpack;
            if (shape == 1) {
                // 512 threads per frame when the share then is <= 3 rows / 4 columns per thread: 75-80 VGPRs, 6 waves per
                // SIMD.  Measured on 1920.1280.3.303 against the 256-thread instance: tanh 4.14 -> 4.22 Gbit/s at 1 dB,
                // 12.8 -> 13.6 at 3 dB; min-sum 4.95 -> 5.42 and 17.1 -> 20.7 (1024 threads: slower than either)
                if (s.want_wide && (s.M + 511) / 512 <= 3 && (s.N + 511) / 512 <= 4) return launch_csr_batched<CT, VARIANT, 4, 3, 4, 4, 512>(s, st, a);
                return launch_csr_batched<CT, VARIANT, 4, 6, 8, 4>(s, st, a);
            }
            if (shape == 2) return launch_csr_batched<CT, VARIANT, 8, 2, 4, 8>(s, st, a);
            if (shape == 4) return launch_csr_batched<CT, VARIANT, 6, 6, 2, 18, 1024, 2>(s, st, a);
            // (jpl.1024 given as CSR: min-sum 3.69 -> 4.73 Gbit/s with 512 threads per frame, tanh 2.19 -> 1.71: its
            //  weight-20 rows need the registers)
            if constexpr (VARIANT == LDPC_V_MINSUM)
                if (s.want_wide && (s.M + 511) / 512 <= 1 && (s.N + 511) / 512 <= 3) return launch_csr_batched<CT, VARIANT, 20, 1, 3, 8, 512>(s, st, a);
            return launch_csr_batched<CT, VARIANT, 20, 2, 6, 8>(s, st, a);
        }
    }
    // register-cached graph indices when the per-thread share is small (f32; DMAX 4 or 8):
    //   rows per thread <= 6 or 2, columns per thread <= 8, column degree <= 8
    const int rpt = (s.M + kCsrThreads - 1) / kCsrThreads, cpt = (s.N + kCsrThreads - 1) / kCsrThreads;
    // measured on codes/1920.1280.3.303, 1 dB: min-sum 1 878 vs 1 586 Mbit/s with cached indices, but tanh
    // 1 209 vs 1 665 (its check node needs the registers: 154 VGPRs -> 3 waves/SIMD), so min-sum only
    const bool cache_ok = VARIANT == LDPC_V_MINSUM && sizeof(CT) == 4 && cpt <= 8 && s.cdmax <= kCdMax && s.want_cache;
    switch (s.dmax) {
        case 4:
            if (cache_ok && rpt <= 6) return launch_csr<CT, VARIANT, 4, 6, 8>(s, st, a);
            return launch_csr<CT, VARIANT, 4, 0, 0>(s, st, a);
        case 6:
            // codes/1920.1280.A (5760 x 1920, row weights 4 and 6, column weights 14 and 18): (2N + 6M) * 4 B = 150 KB of LDS per
            // frame, one workgroup per CU -- so a wide one (LDPC_CSR_WIDE=0: 256 threads)
            if constexpr (sizeof(CT) == 4) {
                if (s.want_wide && ((size_t)2 * s.N + (size_t)6 * s.M) * 4 > 80 * 1024) return launch_csr<CT, VARIANT, 6, 0, 0, 1024>(s, st, a);
            }
            return launch_csr<CT, VARIANT, 6, 0, 0>(s, st, a);
        case 8:
            if (cache_ok && rpt <= 2) return launch_csr<CT, VARIANT, 8, 2, 8>(s, st, a);
            return launch_csr<CT, VARIANT, 8, 0, 0>(s, st, a);
        case 20:
            if constexpr (!(VARIANT == LDPC_V_TANH && sizeof(CT) == 8)) {
                if (cache_ok && rpt <= 2) return launch_csr<CT, VARIANT, 20, 2, 8>(s, st, a);
                return launch_csr<CT, VARIANT, 20, 0, 0>(s, st, a);
            }
            break;
        case 32: if constexpr (!(VARIANT == LDPC_V_TANH && sizeof(CT) == 8)) return launch_csr<CT, VARIANT, 32, 0, 0>(s, st, a); break;
    }
    return set_error(LDPC_EUNSUPPORTED, "no generic on-chip kernel for row degree class %d", s.dmax);
}

static int csr_run(CsrState &s, hipStream_t st, CsrArgs &a) {
    a.ell_col = s.d_ell; a.csc_slot = s.d_csc; a.row_ptr = s.d_row_ptr; a.M = s.M; a.N = s.N; a.E = s.E; a.cdmax = s.cdmax;
    if (s.dtype == LDPC_F64) return s.variant == LDPC_MINSUM ? dispatch_dmax<double, LDPC_V_MINSUM>(s, st, a) : dispatch_dmax<double, LDPC_V_TANH>(s, st, a);
    return s.variant == LDPC_MINSUM ? dispatch_dmax<float, LDPC_V_MINSUM>(s, st, a) : dispatch_dmax<float, LDPC_V_TANH>(s, st, a);
}

int fused_csr_decode(CsrState &s, hipStream_t st, int max_iters, int batch, const void *d_llr, int llr_fmt, uint8_t *d_bits,
                     int32_t *d_iters, uint8_t *d_conv, double *d_final, double *d_trace) {
    CsrArgs a{};
    a.llr = d_llr; a.llr_fmt = llr_fmt; a.llr_round16 = s.round16; a.bits = d_bits; a.iters = d_iters; a.conv = d_conv; a.final_lam = d_final; a.trace = d_trace;
    a.batch = batch; a.max_iters = max_iters;
    return csr_run(s, st, a);
}

int fused_csr_step(CsrState &s, hipStream_t st, int batch, const double *d_orig, const double *d_lam, const double *d_ne,
                   double *d_ne_out, double *d_lam_out, uint8_t *d_syn) {
    CsrArgs a{};
    a.llr = d_orig; a.llr_fmt = LLR_F64; a.llr_round16 = 0; a.batch = batch; a.max_iters = 1; a.step_mode = 1;
    a.st_lam = d_lam; a.st_ne_in = d_ne; a.st_ne_out = d_ne_out; a.final_lam = d_lam_out; a.st_syn = d_syn;
    return csr_run(s, st, a);
}

}  // namespace ldpc
