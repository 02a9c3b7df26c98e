// layered_qc.hip -- quasi-cyclic codes of ANY size with the state in HBM, one workgroup per FRAME: the row-layered
// schedule (first part of this file) and the reference's flooding schedule (flood_qc_kernel, second part).
//
// BASELINE.json configs[4]: "DVB-S2 n = 64 800 long code, layered min-sum + early termination".  A frame of that
// size keeps 253 KB of LLRs and 886 KB of messages -- nothing of it fits on-chip -- and early termination makes frames
// of one batch finish at very different sweeps.  The batch-major layered kernel in flood.hip (lane = codeword, any H)
// keeps a 64-frame slab going until its SLOWEST frame is done while the finished lanes still occupy their share of
// every cache line: measured on the DVB-S2-shaped code at 2 dB, 17 sweeps are needed on average but ~50 are streamed.
// Here the mapping is the on-chip kernels' (lane = row of the circulant, a layer = a block row, a workgroup = one
// frame) with the state left in HBM:
//   lam [frame][N]                 a-posteriori LLRs
//   msg [frame][edge block][sz]    check->variable messages, row of the circulant fastest
// so a wave reads 64 consecutive lam cells of a block column (rotated: one wrap) and 64 consecutive messages: every
// access is coalesced, every graph index is wave-uniform (scalar loads of {block-column base, rotation}), and a frame
// leaves the machine the moment ITS stopping rule fires.  Circulant sizes need not be powers of two (DVB-S2: 360);
// lanes beyond sz idle.  Latency is hidden by frames, not by waves per frame: ~40 VGPRs -> 8 waves per SIMD -> five
// 6-wave frames per CU.
// Algorithmic HBM bytes per frame: sweeps * 4E * s (every edge reads and writes its lam cell and its message; the
// first sweep reads no messages) + the LLRs in, lam initialised, the bits out.
// Specification and arithmetic: oracle_decode_layered (oracle/ldpc_oracle.c); check rules of ldpc_math.h.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "layered_qc.h"
#include "ldpc_math.h"

namespace ldpc {

// the graph tables are read through the CONSTANT address space: the kernel never writes them, and only then may the
// compiler use scalar loads although the kernel also stores to global memory (as in the fused kernels, fused_common.h)
typedef const __attribute__((address_space(4))) int32_t *cidx_t;
struct QcLayerDev {
    int sz, nbr, nbc, N, E;         // E = (number of circulants) * sz
    const int32_t *tab;             // per circulant, block-row-major: {block column * sz, rotation} (two words)
    const int32_t *lbeg;            // [nbr + 1] first circulant of each block row
    // flooding schedule only: the circulants of each block column in DESCENDING block-row order (Orig.hs:96 foldr)
    const int32_t *ctab;            // per entry {circulant index, rotation} (two words)
    const int32_t *cbeg;            // [nbc + 1]
};

struct QcLayerArgs {
    const void *llr; int llr_fmt;   // [batch][N]
    uint8_t *bits; int32_t *iters; uint8_t *conv;
    double *final_lam, *trace;      // may be null
    int batch, max_iters, step_mode;
    const double *st_lam, *st_ne_in; double *st_ne_out, *st_lam_out; uint8_t *st_syn;   // teacher-forced sweep (CSR edge order)
};

// lam may be STORED in another type than the arithmetic's (LT = __half with CT = float: LDPC_F16 + LDPC_SCHED_LAYERED, r03 --
// half the bytes of the dominant stream of the record kernel); Store<> converts, saturating on the way to fp16
template <typename CT, typename LT> __device__ __forceinline__ CT ldlam(const LT *lam, int i) { return (CT)Store<LT>::ld(lam + i); }
template <typename CT, typename LT> __device__ __forceinline__ void stlam(LT *lam, int i, CT v) { Store<LT>::st(lam + i, v); }
// the value a lam cell holds after v was stored into it
template <typename LT, typename CT> __device__ __forceinline__ CT lam_round(CT v) {
    if constexpr (std::is_same<LT, CT>::value) return v;
    else { LT tmp; Store<LT>::st(&tmp, v); return (CT)Store<LT>::ld(&tmp); }
}

// MODE 0: layered sweep, 1: first layered sweep (messages are zero), 2: syndrome only,
//      3: flooding check-node pass (new messages written, lam untouched), 4: the same on the first turn (messages are zero)
template <typename CT, int VARIANT, int DEG, int MODE, typename LT>
__device__ __forceinline__ void qc_row(const QcLayerDev &g, LT *__restrict__ lam, CT *__restrict__ msg, int e0, int r, bool live, bool &odd, bool &flip) {
    int idx[DEG];
    CT l[DEG], t[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        const int cb = ((cidx_t)g.tab)[2 * (e0 + k)], rot = ((cidx_t)g.tab)[2 * (e0 + k) + 1];   // wave-uniform -> scalar loads
        int c = r + rot;
        c -= (c >= g.sz) ? g.sz : 0;
        idx[k] = cb + c;
    }
#pragma unroll
    for (int k = 0; k < DEG; k++) l[k] = live ? ldlam<CT>(lam, idx[k]) : CT(0);
    constexpr bool kReadMsg = MODE == 0 || MODE == 3, kFlooding = MODE >= 3;
    if constexpr (kReadMsg) {
#pragma unroll
        for (int k = 0; k < DEG; k++) t[k] = live ? msg[(size_t)(e0 + k) * g.sz + r] : CT(0);
    }
    bool par = false;
#pragma unroll
    for (int k = 0; k < DEG; k++) par ^= hard(l[k]);
    odd |= par && live;
    if constexpr (MODE == 2) return;
    CT nm[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) { t[k] = kReadMsg ? l[k] - t[k] : l[k] - CT(0); nm[k] = t[k]; }
    cn_update<CT, VARIANT, DEG>(nm);
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        if constexpr (kFlooding) {
            if (live) msg[(size_t)(e0 + k) * g.sz + r] = nm[k];
        } else {
            const CT nw = lam_round<LT>(t[k] + nm[k]);
            flip |= live && (hard(nw) != hard(l[k]));
            if (live) { stlam(lam, idx[k], nw); msg[(size_t)(e0 + k) * g.sz + r] = nm[k]; }
        }
    }
}

template <typename CT, int VARIANT, int DMAX, int MODE, typename LT>
__device__ __forceinline__ void qc_row_padded(const QcLayerDev &g, LT *__restrict__ lam, CT *__restrict__ msg, int e0, int deg, int r, bool live,
                                              bool &odd, bool &flip) {
    int idx[DMAX];
    CT l[DMAX], t[DMAX];
#pragma unroll
    for (int k = 0; k < DMAX; k++) {
        const int kk = k < deg ? k : 0;
        const int cb = ((cidx_t)g.tab)[2 * (e0 + kk)], rot = ((cidx_t)g.tab)[2 * (e0 + kk) + 1];
        int c = r + rot;
        c -= (c >= g.sz) ? g.sz : 0;
        idx[k] = cb + c;
    }
#pragma unroll
    for (int k = 0; k < DMAX; k++) l[k] = (live && k < deg) ? ldlam<CT>(lam, idx[k]) : CT(0);
    constexpr bool kReadMsg = MODE == 0 || MODE == 3, kFlooding = MODE >= 3;
    if constexpr (kReadMsg) {
#pragma unroll
        for (int k = 0; k < DMAX; k++) t[k] = (live && k < deg) ? msg[(size_t)(e0 + k) * g.sz + r] : CT(0);
    }
    bool par = false;
#pragma unroll
    for (int k = 0; k < DMAX; k++) par ^= (k < deg) && hard(l[k]);
    odd |= par && live;
    if constexpr (MODE == 2) return;
    CT nm[DMAX];
#pragma unroll
    for (int k = 0; k < DMAX; k++) { t[k] = (k < deg) ? (kReadMsg ? l[k] - t[k] : l[k] - CT(0)) : CT(INFINITY); nm[k] = t[k]; }
    cn_update_padded<CT, VARIANT, DMAX>(nm, deg);
#pragma unroll
    for (int k = 0; k < DMAX; k++) {
        if (k < deg) {
            if constexpr (kFlooding) {
                if (live) msg[(size_t)(e0 + k) * g.sz + r] = nm[k];
            } else {
                const CT nw = lam_round<LT>(t[k] + nm[k]);
                flip |= live && (hard(nw) != hard(l[k]));
                if (live) { stlam(lam, idx[k], nw); msg[(size_t)(e0 + k) * g.sz + r] = nm[k]; }
            }
        }
    }
}

// ---- min-sum with ROW RECORDS instead of per-edge messages (layered schedule) ---------------------------------------
// The messages a min-sum check row sends take two magnitudes: 3/4 min1 everywhere, 3/4 min2 at the arg-min (the
// reference's MinSum2 / `omit` semigroup, Utils.hs:133-144).  Kept in HBM as {c1, c2, meta = signs | idx << 27} -- 12
// bytes per row (f32) instead of 4 per edge: on the DVB-S2-shaped code (weight-7 rows) a sweep moves 0.39 MB of
// records instead of 1.81 MB of messages.  The rebuilt messages are the per-edge kernel's bit for bit (same
// comparisons, first arg-min wins, the one rounding of Min.hs:78 applied once).  Rows up to weight 27.
template <typename CT, int DMAX, bool FIRST, typename LT>
__device__ __forceinline__ void qc_row_rec(const QcLayerDev &g, LT *__restrict__ lam, CT *__restrict__ rc1, CT *__restrict__ rc2, uint32_t *__restrict__ rmeta,
                                           int e0, int deg, int row, int r, bool live, bool &odd, bool &flip) {
    int idx[DMAX];
    CT l[DMAX], t[DMAX];
#pragma unroll
    for (int k = 0; k < DMAX; k++) {
        const int kk = k < deg ? k : 0;
        const int cb = ((cidx_t)g.tab)[2 * (e0 + kk)], rot = ((cidx_t)g.tab)[2 * (e0 + kk) + 1];
        int c = r + rot;
        c -= (c >= g.sz) ? g.sz : 0;
        idx[k] = cb + c;
    }
#pragma unroll
    for (int k = 0; k < DMAX; k++) l[k] = (live && k < deg) ? ldlam<CT>(lam, idx[k]) : CT(0);
    CT c1 = CT(0), c2 = CT(0);
    uint32_t meta = 0;
    if constexpr (!FIRST) {
        if (live) { c1 = rc1[row]; c2 = rc2[row]; meta = rmeta[row]; }
    }
    bool par = false;
#pragma unroll
    for (int k = 0; k < DMAX; k++) par ^= (k < deg) && hard(l[k]);
    odd |= par && live;
    const uint32_t oidx = meta >> 27;
#pragma unroll
    for (int k = 0; k < DMAX; k++) {
        CT old = CT(0);
        if constexpr (!FIRST) { const CT mag = ((uint32_t)k == oidx) ? c2 : c1; old = ((meta >> k) & 1u) ? -mag : mag; }
        t[k] = (k < deg) ? l[k] - old : CT(INFINITY);
    }
    // cn_minsum (ldpc_math.h) with its result captured as a record
    CT m1 = fabs(t[0]), m2 = CT(INFINITY);
    int i1 = 0;
    unsigned parity = (t[0] > CT(0)) ? 1u : 0u;
#pragma unroll
    for (int k = 1; k < DMAX; k++) {
        const CT a = fabs(t[k]);
        parity ^= (k < deg && t[k] > CT(0)) ? 1u : 0u;
        if (a < m1) { m2 = m1; m1 = a; i1 = k; }
        else if (a < m2) { m2 = a; }
    }
    const CT n1 = CT(0.75) * m1, n2 = CT(0.75) * m2;      // |(-3/4) * acc|: the one rounding of Min.hs:78
    uint32_t nsig = 0;
#pragma unroll
    for (int k = 0; k < DMAX; k++) {
        if (k < deg) {
            const unsigned neg = parity ^ ((t[k] > CT(0)) ? 1u : 0u);   // sign of prod_{j/=k} x_j ; message = (-3/4) * (neg ? -mag : mag)
            const CT mag = (k == i1) ? n2 : n1;
            const CT nm = neg ? mag : -mag;
            nsig |= (neg ? 0u : 1u) << k;                               // sign bit of the message
            const CT nw = lam_round<LT>(t[k] + nm);
            flip |= live && (hard(nw) != hard(l[k]));
            if (live) stlam(lam, idx[k], nw);
        }
    }
    if (live) { rc1[row] = n1; rc2[row] = n2; rmeta[row] = nsig | ((uint32_t)i1 << 27); }
}

template <typename CT, int DCLASS, bool FIRST, typename LT>
__device__ __forceinline__ void qc_layer_rec(const QcLayerDev &g, LT *lam, CT *rc1, CT *rc2, uint32_t *rmeta, int layer, int r, bool live, bool &odd, bool &flip) {
    const int e0 = ((cidx_t)g.lbeg)[layer], deg = ((cidx_t)g.lbeg)[layer + 1] - e0;
    const int row = layer * g.sz + r;
    if (deg <= 0) return;
    if (deg <= 8) { qc_row_rec<CT, 8, FIRST>(g, lam, rc1, rc2, rmeta, e0, deg, row, r, live, odd, flip); return; }
    if constexpr (DCLASS >= 20) { if (deg <= 20) { qc_row_rec<CT, 20, FIRST>(g, lam, rc1, rc2, rmeta, e0, deg, row, r, live, odd, flip); return; } }
    if constexpr (DCLASS >= 32) { qc_row_rec<CT, 27, FIRST>(g, lam, rc1, rc2, rmeta, e0, deg, row, r, live, odd, flip); }
}

template <typename CT, int VARIANT, int DCLASS, int MODE, typename LT>
__device__ __forceinline__ void qc_layer(const QcLayerDev &g, LT *lam, CT *msg, int layer, int r, bool live, bool &odd, bool &flip) {
    const int e0 = ((cidx_t)g.lbeg)[layer], deg = ((cidx_t)g.lbeg)[layer + 1] - e0;
    switch (deg) {
        case 0: return;
        case 1:
            if constexpr (VARIANT == LDPC_V_TANH) qc_row<CT, VARIANT, 1, MODE>(g, lam, msg, e0, r, live, odd, flip);
            else qc_row<CT, LDPC_V_TANH, 1, 2>(g, lam, msg, e0, r, live, odd, flip);    // (min-sum on weight 1 is rejected at creation; never reached)
            return;
        case 2: qc_row<CT, VARIANT, 2, MODE>(g, lam, msg, e0, r, live, odd, flip); return;
        case 3: qc_row<CT, VARIANT, 3, MODE>(g, lam, msg, e0, r, live, odd, flip); return;
        case 4: qc_row<CT, VARIANT, 4, MODE>(g, lam, msg, e0, r, live, odd, flip); return;
        case 5: qc_row<CT, VARIANT, 5, MODE>(g, lam, msg, e0, r, live, odd, flip); return;
        case 6: qc_row<CT, VARIANT, 6, MODE>(g, lam, msg, e0, r, live, odd, flip); return;
        case 7: qc_row<CT, VARIANT, 7, MODE>(g, lam, msg, e0, r, live, odd, flip); return;
        case 8: qc_row<CT, VARIANT, 8, MODE>(g, lam, msg, e0, r, live, odd, flip); return;
        default: break;
    }
    if constexpr (DCLASS >= 20) {
        if (deg == 18) { qc_row<CT, VARIANT, 18, MODE>(g, lam, msg, e0, r, live, odd, flip); return; }
        if (deg <= 12) { qc_row_padded<CT, VARIANT, 12, MODE>(g, lam, msg, e0, deg, r, live, odd, flip); return; }
        if (deg <= 16) { qc_row_padded<CT, VARIANT, 16, MODE>(g, lam, msg, e0, deg, r, live, odd, flip); return; }
        if (deg <= 20) { qc_row_padded<CT, VARIANT, 20, MODE>(g, lam, msg, e0, deg, r, live, odd, flip); return; }
    }
    if constexpr (DCLASS >= 32) {
        if (deg <= 24) { qc_row_padded<CT, VARIANT, 24, MODE>(g, lam, msg, e0, deg, r, live, odd, flip); return; }
        if (deg <= 32) { qc_row_padded<CT, VARIANT, 32, MODE>(g, lam, msg, e0, deg, r, live, odd, flip); return; }
    }
}

// block = ceil(sz / 64) waves; thread r = row r of every circulant (threads >= sz idle); grid = frames
// RECORDS: min-sum rows kept as records (qc_row_rec) -- msg_all then holds [frame][3][M] words instead of [frame][E] messages
// Waves per SIMD asked of the register allocator for light rows (weight <= 8, f32).  A workgroup of w waves takes
// ceil(w/4) wave slots on the fuller SIMDs, so DVB-S2's 6-wave frames need 2: with the 86 VGPRs / 104 SGPRs the compiler
// takes when left alone only 2 frames were resident per CU (12.6 waves measured); held to 64 / 80 (SGPRs spill to VGPR
// lanes, no scratch) 3 (17.4 waves): 82.5 -> 67.5 ms for 8 192 frames of the long code (tools/microbench_residency.hip has
// the placement rule: 384-thread blocks reach 4.5 per CU at best).
#ifndef LQC_WAVES_LIGHT
#define LQC_WAVES_LIGHT 8
#endif
template <typename CT, int VARIANT, int DCLASS, bool RECORDS, typename LT = CT>
__global__ __launch_bounds__(1024, (DCLASS <= 8 && sizeof(CT) == 4 ? LQC_WAVES_LIGHT : 4)) void layered_qc_kernel(QcLayerDev g, LT *lam_all, CT *msg_all, QcLayerArgs A) {
    const int r = threadIdx.x;
    const bool live = r < g.sz;
    const size_t frame = blockIdx.x;
    LT *lam = lam_all + frame * (size_t)g.N;
    const int Mrows = g.nbr * g.sz;
    CT *msg = msg_all + frame * (RECORDS ? (size_t)3 * Mrows : (size_t)g.E);
    CT *rc1 = msg, *rc2 = msg + Mrows;
    uint32_t *rmeta = reinterpret_cast<uint32_t *>(msg + 2 * (size_t)Mrows);   // (f64: the upper half of each cell is unused)
    const size_t fN = frame * (size_t)g.N;
    // ---- lam <- channel LLRs (or the given state)
    if (A.step_mode) {
        for (int i = r; i < g.N; i += blockDim.x) stlam(lam, i, (CT)A.st_lam[fN + i]);
        for (int l = 0; l < g.nbr; l++) {       // messages arrive in CSR edge order: row-major, ascending column
            const int e0 = g.lbeg[l], deg = g.lbeg[l + 1] - e0;
            if (live)
                for (int k = 0; k < deg; k++) msg[(size_t)(e0 + k) * g.sz + r] = (CT)A.st_ne_in[frame * (size_t)g.E + (size_t)e0 * g.sz + (size_t)r * deg + k];
        }
    } else {
        for (int i = r; i < g.N; i += blockDim.x) stlam(lam, i, load_llr<CT>(A.llr, fN + i, A.llr_fmt));
    }
    __syncthreads();
    bool conv = false;
    int n = 0;
    {   // syndrome of the hard decisions before the first sweep
        bool odd = false, flip = false;
        for (int l = 0; l < g.nbr; l++) qc_layer<CT, VARIANT, DCLASS, 2>(g, lam, msg, l, r, live, odd, flip);
        const int any = __syncthreads_or(odd ? 1 : 0);
        if (A.step_mode) { if (r == 0) A.st_syn[frame] = any ? 0 : 1; }
        else conv = !any;
    }
    if (A.trace && !A.step_mode) {   // (uniform condition)
        for (int i = r; i < g.N; i += blockDim.x) A.trace[(frame * (A.max_iters + 1)) * (size_t)g.N + i] = (double)ldlam<CT>(lam, i);
        __syncthreads();             // no wave starts layer 0 (which writes lam) while another still copies row 0
    }
    if (!conv) {
        for (n = 1; n <= A.max_iters; n++) {
            bool odd = false, flip = false;
            if constexpr (RECORDS) {
                if (n == 1) { for (int l = 0; l < g.nbr; l++) { qc_layer_rec<CT, DCLASS, true>(g, lam, rc1, rc2, rmeta, l, r, live, odd, flip); __syncthreads(); } }
                else { for (int l = 0; l < g.nbr; l++) { qc_layer_rec<CT, DCLASS, false>(g, lam, rc1, rc2, rmeta, l, r, live, odd, flip); __syncthreads(); } }
            } else if (n == 1 && !A.step_mode) {
                for (int l = 0; l < g.nbr; l++) { qc_layer<CT, VARIANT, DCLASS, 1>(g, lam, msg, l, r, live, odd, flip); __syncthreads(); }
            } else {
                for (int l = 0; l < g.nbr; l++) { qc_layer<CT, VARIANT, DCLASS, 0>(g, lam, msg, l, r, live, odd, flip); __syncthreads(); }
            }
            const int any = __syncthreads_or((odd || flip) ? 1 : 0);
            if (A.trace && !A.step_mode) {
                for (int i = r; i < g.N; i += blockDim.x) A.trace[(frame * (A.max_iters + 1) + n) * (size_t)g.N + i] = (double)ldlam<CT>(lam, i);
                __syncthreads();
            }
            if (A.step_mode) break;
            if (!any) { conv = true; break; }
        }
        if (n > A.max_iters) n = A.max_iters;
    }
    if (A.step_mode) {
        for (int i = r; i < g.N; i += blockDim.x) A.st_lam_out[fN + i] = (double)ldlam<CT>(lam, i);
        for (int l = 0; l < g.nbr; l++) {
            const int e0 = g.lbeg[l], deg = g.lbeg[l + 1] - e0;
            if (live)
                for (int k = 0; k < deg; k++) A.st_ne_out[frame * (size_t)g.E + (size_t)e0 * g.sz + (size_t)r * deg + k] = (double)msg[(size_t)(e0 + k) * g.sz + r];
        }
        return;
    }
    if constexpr (kVetoesNonFinite<CT, VARIANT> && sizeof(LT) == 4) {   // (f32 lam; fp16 lam saturates by its own rule)
        if (conv) {                                       // (uniform) LLRs that left the float range: failed, not "converged" (ldpc_math.h)
            bool bad = false;
            for (int i = r; i < g.N; i += blockDim.x) bad |= not_finite(ldlam<CT>(lam, i));
            if (__syncthreads_or(bad ? 1 : 0)) conv = false;
        }
    }
    // ---- result: hard(lam) of a frame that stopped by the rule, the channel's decisions otherwise (as Orig.hs:69-70)
    for (int i = r; i < g.N; i += blockDim.x) {
        const CT v = conv ? ldlam<CT>(lam, i) : lam_round<LT>(load_llr<CT>(A.llr, fN + i, A.llr_fmt));   // (an LLR counts as stored in LT)
        A.bits[fN + i] = v > CT(0) ? 1 : 0;
        if (A.final_lam) A.final_lam[fN + i] = (double)v;
    }
    if (r == 0) {
        if (A.iters) A.iters[frame] = conv ? n : A.max_iters;
        if (A.conv) A.conv[frame] = conv ? 1 : 0;
    }
}

// ==================================================================== flooding schedule, same mapping
// The reference's own schedule (Orig.hs:67-98) for QC codes whose frame does not fit on-chip: per turn a check-node pass
// over the block rows (syndrome of hard(lam) from the gathered values, new messages in place), then a variable-node
// pass: thread c owns column c of every block column and adds its messages in the reference's order -- foldr (+) orig,
// i.e. descending row (Orig.hs:96); the message of circulant (br, bc) for column c sits at row (c - rot) mod sz, so the
// reads are rotated but contiguous.  One launch is the whole decode; a frame stops the turn ITS syndrome is zero.
// Same arithmetic and orders as flood.hip's two kernels: f64 bit-exact with the oracle, f32 identical to that path.
template <typename CT>
__device__ __forceinline__ CT qc_column(const QcLayerDev &g, const CT *__restrict__ msg, int bc, int c, CT acc) {
    const int q0 = ((cidx_t)g.cbeg)[bc], deg = ((cidx_t)g.cbeg)[bc + 1] - q0;
    for (int base = 0; base < deg; base += 8) {       // eight messages in flight at a time
        CT v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int jj = base + j < deg ? base + j : deg - 1;
            const int e = ((cidx_t)g.ctab)[2 * (q0 + jj)], rot = ((cidx_t)g.ctab)[2 * (q0 + jj) + 1];
            int r = c - rot;
            r += (r < 0) ? g.sz : 0;
            v[j] = (base + j < deg) ? msg[(size_t)e * g.sz + r] : CT(0);
        }
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (base + j < deg) acc = v[j] + acc;
    }
    return acc;
}

// (no register budget here: held to 64 VGPRs / 80 SGPRs the flooding kernel lost 12 % on the long code: 1 663 -> 1 465 Mbit/s)
template <typename CT, int VARIANT, int DCLASS>
__global__ __launch_bounds__(1024) void flood_qc_kernel(QcLayerDev g, CT *lam_all, CT *msg_all, QcLayerArgs A) {
    const int r = threadIdx.x;
    const bool live = r < g.sz;
    const size_t frame = blockIdx.x;
    CT *lam = lam_all + frame * (size_t)g.N;
    CT *msg = msg_all + frame * (size_t)g.E;
    const size_t fN = frame * (size_t)g.N;
    if (A.step_mode) {
        for (int i = r; i < g.N; i += blockDim.x) lam[i] = (CT)A.st_lam[fN + i];
        for (int l = 0; l < g.nbr; l++) {
            const int e0 = g.lbeg[l], deg = g.lbeg[l + 1] - e0;
            if (live)
                for (int k = 0; k < deg; k++) msg[(size_t)(e0 + k) * g.sz + r] = (CT)A.st_ne_in[frame * (size_t)g.E + (size_t)e0 * g.sz + (size_t)r * deg + k];
        }
    } else {
        for (int i = r; i < g.N; i += blockDim.x) lam[i] = load_llr<CT>(A.llr, fN + i, A.llr_fmt);
    }
    __syncthreads();
    bool conv = false;
    int n = 0;
    for (;; n++) {
        if (A.trace && !A.step_mode) {
            for (int i = r; i < g.N; i += blockDim.x) A.trace[(frame * (A.max_iters + 1) + n) * (size_t)g.N + i] = (double)lam[i];
        }
        bool odd = false, flip = false;
        const bool last = !A.step_mode && n >= A.max_iters;     // Orig.hs:70: only the syndrome is still wanted
        if (last) { for (int l = 0; l < g.nbr; l++) qc_layer<CT, VARIANT, DCLASS, 2>(g, lam, msg, l, r, live, odd, flip); }
        else if (n == 0 && !A.step_mode) { for (int l = 0; l < g.nbr; l++) qc_layer<CT, VARIANT, DCLASS, 4>(g, lam, msg, l, r, live, odd, flip); }
        else { for (int l = 0; l < g.nbr; l++) qc_layer<CT, VARIANT, DCLASS, 3>(g, lam, msg, l, r, live, odd, flip); }
        const int any = __syncthreads_or(odd ? 1 : 0);            // also: every new message is written before a column reads it
        if (A.step_mode) { if (r == 0) A.st_syn[frame] = any ? 0 : 1; }
        else {
            if (!any) { conv = true; break; }                      // Orig.hs:69
            if (last) break;                                       // Orig.hs:70
        }
        if (live)
            for (int bc = 0; bc < g.nbc; bc++) {
                const size_t i = (size_t)bc * g.sz + r;
                lam[i] = qc_column<CT>(g, msg, bc, r, load_llr<CT>(A.llr, fN + i, A.llr_fmt));
            }
        __syncthreads();
        if (A.step_mode) break;
    }
    if (A.step_mode) {
        for (int i = r; i < g.N; i += blockDim.x) A.st_lam_out[fN + i] = (double)lam[i];
        for (int l = 0; l < g.nbr; l++) {
            const int e0 = g.lbeg[l], deg = g.lbeg[l + 1] - e0;
            if (live)
                for (int k = 0; k < deg; k++) A.st_ne_out[frame * (size_t)g.E + (size_t)e0 * g.sz + (size_t)r * deg + k] = (double)msg[(size_t)(e0 + k) * g.sz + r];
        }
        return;
    }
    if constexpr (kVetoesNonFinite<CT, VARIANT>) {
        if (conv) {                                       // (uniform) LLRs that left the float range: failed, not "converged" (ldpc_math.h)
            bool bad = false;
            for (int i = r; i < g.N; i += blockDim.x) bad |= not_finite(lam[i]);
            if (__syncthreads_or(bad ? 1 : 0)) conv = false;
        }
    }
    for (int i = r; i < g.N; i += blockDim.x) {
        const CT v = conv ? lam[i] : load_llr<CT>(A.llr, fN + i, A.llr_fmt);
        A.bits[fN + i] = v > CT(0) ? 1 : 0;
        if (A.final_lam) A.final_lam[fN + i] = (double)v;
    }
    if (r == 0) {
        if (A.iters) A.iters[frame] = conv ? n : A.max_iters;
        if (A.conv) A.conv[frame] = conv ? 1 : 0;
    }
}

// ------------------------------------------------------------------ host side
struct LayeredQcState {
    int variant = 0, dtype = 0, max_batch = 0, max_row_deg = 0, threads = 0;
    QcLayerDev g{};
    bool flooding = false;
    bool records = false;     // layered min-sum: row records instead of per-edge messages (LDPC_LAYERED_RECORDS=0: per edge, A/B)
    int32_t *d_tab = nullptr;
    int32_t *d_lbeg = nullptr, *d_ctab = nullptr, *d_cbeg = nullptr;
    void *lam = nullptr, *msg = nullptr;
    LayeredLdsState *lds = nullptr;   // fp16 lam storage and the frame fits LDS: lam on-chip, records streamed (layered_lds.hip); then lam / msg above stay null
    KernelTimer *timer = nullptr;
    LaunchInfo info;
};

const char *layered_qc_why_not(const ldpc_code &c, int variant, int dtype, int flooding) {
    if (c.sz <= 0) return "code was not created from a quasi-cyclic description";
    if (c.sz > 1024) return "circulant size above 1024";
    if (dtype == LDPC_F16) {   // fp16 STORAGE of lam (f32 arithmetic, f32 row records): the layered min-sum record kernel only
        if (flooding || variant != LDPC_MINSUM || c.max_row_deg > 27) return "fp16 lam storage exists for the layered min-sum kernel with row records (rows up to weight 27)";
        const char *re = getenv("LDPC_LAYERED_RECORDS");
        if (re && !strcmp(re, "0")) return "fp16 lam storage needs the row-record kernel (LDPC_LAYERED_RECORDS=0 disables it)";
    } else if (dtype != LDPC_F32 && dtype != LDPC_F64) return "the frame-per-workgroup HBM kernels exist for f32, f64 and (layered min-sum) fp16 lam storage";
    if (variant != LDPC_TANH && variant != LDPC_MINSUM) return "tanh and min-sum rules only";
    if (c.max_row_deg > 32) return "check rows above weight 32";
    if (!flooding) {
        if ((int)c.layer_ptr.size() != c.block_rows + 1) return "layers were replaced: not the block rows";
        for (int br = 0; br <= c.block_rows; br++) if (c.layer_ptr[br] != br * c.sz) return "layers were replaced: not the block rows";
    }
    const char *e = getenv(flooding ? "LDPC_FLOOD_QC" : "LDPC_LAYERED_QC");
    if (e && !strcmp(e, "0")) return flooding ? "disabled (LDPC_FLOOD_QC=0)" : "disabled (LDPC_LAYERED_QC=0)";
    return nullptr;
}

void layered_qc_destroy(LayeredQcState *s) {
    if (!s) return;
    (void)hipFree(s->d_tab); (void)hipFree(s->d_lbeg); (void)hipFree(s->d_ctab); (void)hipFree(s->d_cbeg); (void)hipFree(s->lam); (void)hipFree(s->msg);
    layered_lds_destroy(s->lds);
    delete s;
}

LayeredQcState *layered_qc_create(const ldpc_code &c, int variant, int dtype, int max_batch, int flooding) {
    const char *why = layered_qc_why_not(c, variant, dtype, flooding);
    if (why) { set_error(LDPC_EUNSUPPORTED, "%s", why); return nullptr; }
    LayeredQcState *s = new (std::nothrow) LayeredQcState();
    if (!s) { set_error(LDPC_ENOMEM, "out of host memory"); return nullptr; }
    try {
        s->variant = variant; s->dtype = dtype; s->max_batch = max_batch; s->max_row_deg = c.max_row_deg; s->flooding = flooding != 0;
        if (!flooding && layered_lds_why_not(c, variant, dtype) == nullptr) {
            s->lds = layered_lds_create(c, variant, dtype, max_batch);
            if (!s->lds) { layered_qc_destroy(s); return nullptr; }
            s->info = layered_lds_launch_info(*s->lds);
            return s;
        }
        std::vector<int32_t> tab;
        std::vector<int32_t> lbeg(1, 0);
        std::vector<std::vector<std::pair<int, int>>> cols((size_t)c.block_cols);   // per block column: (circulant index, rotation), ascending block row
        for (int br = 0; br < c.block_rows; br++) {
            for (int bc = 0; bc < c.block_cols; bc++) {
                const int off = c.offsets[(size_t)br * c.block_cols + bc];
                if (off >= 0) { cols[bc].push_back({(int)(tab.size() / 2), off}); tab.push_back(bc * c.sz); tab.push_back(off); }
            }
            lbeg.push_back((int32_t)(tab.size() / 2));
        }
        s->g.sz = c.sz; s->g.nbr = c.block_rows; s->g.nbc = c.block_cols; s->g.N = c.N; s->g.E = c.E;
        s->threads = (c.sz + 63) / 64 * 64;
        const size_t es = dtype == LDPC_F64 ? 8 : 4;
        hipError_t e = hipMalloc((void **)&s->d_tab, sizeof(int32_t) * std::max<size_t>(tab.size(), 2));
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_lbeg, sizeof(int32_t) * lbeg.size());
        if (e == hipSuccess) e = hipMalloc(&s->lam, dtype == LDPC_F16 ? (size_t)max_batch * c.N * 2 : (size_t)max_batch * c.N * es);
        {
            const char *re = getenv("LDPC_LAYERED_RECORDS");
            s->records = !flooding && variant == LDPC_MINSUM && c.max_row_deg <= 27 && !(re && !strcmp(re, "0"));
        }
        // (sized for the per-edge form either way: the teacher-forced step always runs it)
        if (e == hipSuccess) e = hipMalloc(&s->msg, (size_t)max_batch * std::max(std::max(c.E, 3 * c.M), 1) * es);
        std::vector<int32_t> ctab, cbeg(1, 0);
        for (auto &col : cols) {
            for (auto it = col.rbegin(); it != col.rend(); ++it) { ctab.push_back(it->first); ctab.push_back(it->second); }   // descending block row
            cbeg.push_back((int32_t)(ctab.size() / 2));
        }
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_ctab, sizeof(int32_t) * std::max<size_t>(ctab.size(), 2));
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_cbeg, sizeof(int32_t) * cbeg.size());
        if (e == hipSuccess && !ctab.empty()) e = hipMemcpy(s->d_ctab, ctab.data(), sizeof(int32_t) * ctab.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(s->d_cbeg, cbeg.data(), sizeof(int32_t) * cbeg.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess && !tab.empty()) e = hipMemcpy(s->d_tab, tab.data(), sizeof(int32_t) * tab.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(s->d_lbeg, lbeg.data(), sizeof(int32_t) * lbeg.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            set_error(e == hipErrorOutOfMemory ? LDPC_ENOMEM : LDPC_EHIP, "layered_qc_create (%d frames x %zu bytes of state): %s", max_batch,
                      ((size_t)c.N + c.E) * es, hipGetErrorString(e));
            layered_qc_destroy(s);
            return nullptr;
        }
        s->g.tab = s->d_tab; s->g.lbeg = s->d_lbeg; s->g.ctab = s->d_ctab; s->g.cbeg = s->d_cbeg;
        snprintf(s->info.name, sizeof(s->info.name), "ldpc::%s<%s, %d, %d%s%s>", flooding ? "flood_qc_kernel" : "layered_qc_kernel", dtype == LDPC_F64 ? "double" : "float",
                 variant == LDPC_MINSUM ? LDPC_V_MINSUM : LDPC_V_TANH, c.max_row_deg <= 8 ? 8 : (c.max_row_deg <= 20 ? 20 : 32),
                 flooding ? "" : (s->records ? ", true" : ", false"), dtype == LDPC_F16 ? ", __half" : "");
        s->info.threads = s->threads; s->info.frames_per_wg = 1;
        return s;
    } catch (...) { layered_qc_destroy(s); set_error(LDPC_ENOMEM, "out of host memory"); return nullptr; }
}

void layered_qc_set_timer(LayeredQcState *s, KernelTimer *t) { if (s) { s->timer = t; layered_lds_set_timer(s->lds, t); } }
const LaunchInfo &layered_qc_launch_info(const LayeredQcState &s) { return s.info; }

template <typename CT, int VARIANT>
static int launch(LayeredQcState &s, hipStream_t st, QcLayerArgs &a) {
    const dim3 grid(a.batch), block(s.threads);
    if (s.timer && !a.step_mode) s.timer->begin(st);
    if (s.flooding) {
        if (s.max_row_deg <= 8) hipLaunchKernelGGL((flood_qc_kernel<CT, VARIANT, 8>), grid, block, 0, st, s.g, (CT *)s.lam, (CT *)s.msg, a);
        else if (s.max_row_deg <= 20) hipLaunchKernelGGL((flood_qc_kernel<CT, VARIANT, 20>), grid, block, 0, st, s.g, (CT *)s.lam, (CT *)s.msg, a);
        else hipLaunchKernelGGL((flood_qc_kernel<CT, VARIANT, 32>), grid, block, 0, st, s.g, (CT *)s.lam, (CT *)s.msg, a);
    } else if (VARIANT == LDPC_V_MINSUM && s.records && !a.step_mode) {
        if constexpr (VARIANT == LDPC_V_MINSUM) {
            if (s.max_row_deg <= 8) hipLaunchKernelGGL((layered_qc_kernel<CT, VARIANT, 8, true>), grid, block, 0, st, s.g, (CT *)s.lam, (CT *)s.msg, a);
            else if (s.max_row_deg <= 20) hipLaunchKernelGGL((layered_qc_kernel<CT, VARIANT, 20, true>), grid, block, 0, st, s.g, (CT *)s.lam, (CT *)s.msg, a);
            else hipLaunchKernelGGL((layered_qc_kernel<CT, VARIANT, 32, true>), grid, block, 0, st, s.g, (CT *)s.lam, (CT *)s.msg, a);
        }
    } else if (s.max_row_deg <= 8) hipLaunchKernelGGL((layered_qc_kernel<CT, VARIANT, 8, false>), grid, block, 0, st, s.g, (CT *)s.lam, (CT *)s.msg, a);
    else if (s.max_row_deg <= 20) hipLaunchKernelGGL((layered_qc_kernel<CT, VARIANT, 20, false>), grid, block, 0, st, s.g, (CT *)s.lam, (CT *)s.msg, a);
    else hipLaunchKernelGGL((layered_qc_kernel<CT, VARIANT, 32, false>), grid, block, 0, st, s.g, (CT *)s.lam, (CT *)s.msg, a);
    if (s.timer && !a.step_mode) s.timer->end(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(LDPC_EHIP, "layered_qc launch: %s", hipGetErrorString(e));
    return LDPC_OK;
}

static int launch_f16(LayeredQcState &s, hipStream_t st, QcLayerArgs &a) {
    if (a.step_mode) return set_error(LDPC_EUNSUPPORTED, "no teacher-forced step with fp16 lam storage (the record kernel keeps no per-edge messages)");
    const dim3 grid(a.batch), block(s.threads);
    if (s.timer) s.timer->begin(st);
    if (s.max_row_deg <= 8) hipLaunchKernelGGL((layered_qc_kernel<float, LDPC_V_MINSUM, 8, true, __half>), grid, block, 0, st, s.g, (__half *)s.lam, (float *)s.msg, a);
    else if (s.max_row_deg <= 20) hipLaunchKernelGGL((layered_qc_kernel<float, LDPC_V_MINSUM, 20, true, __half>), grid, block, 0, st, s.g, (__half *)s.lam, (float *)s.msg, a);
    else hipLaunchKernelGGL((layered_qc_kernel<float, LDPC_V_MINSUM, 32, true, __half>), grid, block, 0, st, s.g, (__half *)s.lam, (float *)s.msg, a);
    if (s.timer) s.timer->end(st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(LDPC_EHIP, "layered_qc launch: %s", hipGetErrorString(e));
    return LDPC_OK;
}

static int run(LayeredQcState &s, hipStream_t st, QcLayerArgs &a) {
    if (s.dtype == LDPC_F16) return launch_f16(s, st, a);
    if (s.dtype == LDPC_F64) return s.variant == LDPC_MINSUM ? launch<double, LDPC_V_MINSUM>(s, st, a) : launch<double, LDPC_V_TANH>(s, st, a);
    return s.variant == LDPC_MINSUM ? launch<float, LDPC_V_MINSUM>(s, st, a) : launch<float, LDPC_V_TANH>(s, st, a);
}

int layered_qc_decode(LayeredQcState &s, hipStream_t st, int max_iters, int batch, const void *d_llr, int llr_fmt, uint8_t *d_bits, int32_t *d_iters,
                      uint8_t *d_conv, double *d_final, double *d_trace) {
    QcLayerArgs a{};
    a.llr = d_llr; a.llr_fmt = llr_fmt; a.bits = d_bits; a.iters = d_iters; a.conv = d_conv; a.final_lam = d_final; a.trace = d_trace;
    a.batch = batch; a.max_iters = max_iters;
    if (s.lds) return layered_lds_decode(*s.lds, st, max_iters, batch, d_llr, llr_fmt, d_bits, d_iters, d_conv, d_final, d_trace);
    return run(s, st, a);
}

int layered_qc_step(LayeredQcState &s, hipStream_t st, int batch, const double *d_orig, const double *d_lam, const double *d_ne, double *d_ne_out,
                    double *d_lam_out, uint8_t *d_syn) {
    QcLayerArgs a{};
    a.llr = d_orig; a.llr_fmt = LLR_F64; a.batch = batch; a.max_iters = 1; a.step_mode = 1;
    a.st_lam = d_lam; a.st_ne_in = d_ne; a.st_ne_out = d_ne_out; a.st_lam_out = d_lam_out; a.st_syn = d_syn;
    if (s.lds) return set_error(LDPC_EUNSUPPORTED, "no teacher-forced step with fp16 lam storage (the record kernels keep no per-edge messages)");
    return run(s, st, a);
}

}  // namespace ldpc
