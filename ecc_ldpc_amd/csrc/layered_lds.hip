// layered_lds.hip -- row-layered min-sum for LONG quasi-cyclic codes: lam ON-CHIP as fp16, row records streamed from HBM.
//
// BASELINE.json configs[4] ("DVB-S2 n = 64 800 long code, layered min-sum + early termination").  A frame of that size has
// 253 KB of f32 LLRs -- beyond the 160 KB of LDS -- so layered_qc.hip keeps lam AND the check rows' records in HBM and moves, per
// sweep and frame, 2 E s bytes of lam gathers / scatters next to 24 bytes of record per row: 2.59 MB on the DVB-S2-shaped code, of
// which only 0.78 MB are records.  In fp16 the same lam is 130 KB: it fits.  This kernel is layered_qc_kernel<float, 1, D, true,
// __half> (same arithmetic, same roundings: f32 arithmetic and records, every lam write saturated and rounded to binary16 --
// specification oracle/emulate_f16.py decode_minsum_f16_layered, reproduced bit for bit) with
//   * lam in LDS for the whole decode: one workgroup per frame, thread r = row r of every block row (layer), gathers and
//     scatters are 2-byte LDS accesses at (r + rotation) mod sz -- consecutive lanes, consecutive half words;
//   * the row records {3/4 min1, 3/4 min2, signs | arg-min} (the reference's MinSum2 / `omit` semigroup, Utils.hs:133-144) as
//     ONE 12-byte structure per row in a scratch area that belongs to the WORKGROUP, not to the frame: [layer][thread], so a wave
//     moves 768 contiguous bytes per layer and direction, and the records of layer l + P are requested while layer l computes
//     (the barriers between layers wait for LDS only: the loads stay in flight across them);
//   * persistent workgroups -- as many as are resident (one per CU for the long code) -- taking frames from a counter: the scratch
//     area is (workgroups x M x 12) bytes (106 MB for 256 workgroups of the long code: it stays in the 256 MiB Infinity Cache)
//     instead of (frames x M x 12).
// Algorithmic HBM bytes per frame: sweeps * 24 M (the first sweep writes only) + the LLRs in + the bits out.
// Stop rule as Orig.hs:67-71 in its layered form (oracle_decode_layered): a sweep in which every check was satisfied when visited and no
// hard decision changed ends the frame; out of sweeps -> the channel's hard decisions.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "layered_qc.h"
#include "ldpc_math.h"

namespace ldpc {

typedef const __attribute__((address_space(4))) int32_t *ctab_t;   // graph tables: scalar loads (never written by the kernel)

struct LdsRec { float c1, c2; uint32_t meta; };   // meta: bit (deg-1-k) = sign bit of the message on edge k; bits 27..31 = an arg-min edge

struct LdsDev {
    int sz, nbr, nbc, N;
    int two_sz;                     // 2 * sz (bytes of one block column of lam)
    const int32_t *tab;             // per circulant, block-row-major: {A = 2 * (block column * sz + rotation), thr = sz - rotation}:
                                    // row r reads the half word at byte A + 2 r - (r >= thr ? 2 sz : 0)
    const int32_t *lbeg;            // [nbr + 1] first circulant of each block row
    // the pipelined instances: consecutive block rows that share no block column form a GROUP and are run together: their rows touch
    // distinct lam cells, so the result is that of running them one after the other.  A block row is run by ceil(sz / 64) waves (`tl`
    // threads, one "sub"); a sub runs RW block rows of a group one after the other WITHOUT a barrier in between (kernel template
    // argument; the workgroup meets once per group).  ng groups of gsz slots; slot s = group * gsz + sub * RW + w, w < RW.
    int ng, gsz, tl;
    const int32_t *gtab;            // [ng * gsz][kLtab] per slot: {first circulant, weight (0: no block row in this slot), block row, -, the first
                                    // 8 circulants' {A, thr}}; copied to LDS behind lam, read one group ahead of its use
};
constexpr int kLtab = 20;

struct LdsArgs {
    const void *llr; int llr_fmt;   // [batch][N]
    uint8_t *bits; int32_t *iters; uint8_t *conv;
    double *final_lam, *trace;      // may be null
    int batch, max_iters;
    int *work_counter;              // next frame to take = gridDim.x + atomicAdd(work_counter, 1)
};

__device__ __forceinline__ void lds_barrier() {     // LDS traffic only: global loads / stores stay in flight across it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ _Float16 sat16(float v) { return (_Float16)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); }   // Store<__half>::st

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
// eight channel LLRs starting at element i (16-byte aligned), as a decoder with fp16 storage holds them: saturated, rounded to nearest even
template <int FMT> __device__ __forceinline__ void load_llr8(const void *base, size_t i, half8 &h) {
    if constexpr (FMT == LLR_F64) {                                      // (never taken: the wide path is for fp16 and f32 input)
#pragma unroll
        for (int k = 0; k < 8; k++) h[k] = sat16((float)reinterpret_cast<const double *>(base)[i + k]);
    } else if constexpr (FMT == LLR_F16) {
        const half8 v = *reinterpret_cast<const half8 *>(reinterpret_cast<const _Float16 *>(base) + i);
#pragma unroll
        for (int k = 0; k < 8; k++) h[k] = sat16((float)v[k]);          // (an infinity saturates like any other value)
    } else {
        const float4 a = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(base) + i);
        const float4 b = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(base) + i + 4);
        h[0] = sat16(a.x); h[1] = sat16(a.y); h[2] = sat16(a.z); h[3] = sat16(a.w);
        h[4] = sat16(b.x); h[5] = sat16(b.y); h[6] = sat16(b.z); h[7] = sat16(b.w);
    }
}

typedef __attribute__((address_space(3))) _Float16 *lds_half_t;
__device__ __forceinline__ lds_half_t lds_cell(uint32_t byte_addr) { return (lds_half_t)(uintptr_t)byte_addr; }

// LDS byte address of the lam cell row r reads on a circulant {A, thr} (LdsDev::tab): rb = lam + 2 r, rw = rb - 2 sz
__device__ __forceinline__ uint32_t lds_addr(int A, int thr, int r, uint32_t rb, uint32_t rw) { return (uint32_t)A + (r >= thr ? rw : rb); }

// one check row of weight deg <= DMAX (deg wave-uniform; DMAX == deg when EXACT): gather, rebuild the old messages from the record,
// two-min, new record, lam written back.  Arithmetic of layered_qc.hip qc_row_rec; signs by bit operations as ldpc_math.h
// cn_update_padded (a zero t makes every message it could change the sign of a zero).  No lane is masked: the idle lanes of the
// last wave shadow rows of their own wave (same reads in the same instruction, same values written to the same cells).
// tb: the row's first 8 circulants already in (scalar) registers, or null: read from LdsDev::tab
// mid(): called between the gather phase and the write-back phase (the pipelined loop requests the next layer's graph entries there:
// behind the last wait for an LDS read -- scalar loads return out of order, so any wait on the counter they share with LDS becomes a
// wait for them too -- and early enough to have landed when the layer's barrier is reached)
struct NoMid { __device__ __forceinline__ void operator()() const {} };
template <int DMAX, bool EXACT, bool FIRST, class Mid = NoMid>
__device__ __forceinline__ void lds_row(const LdsDev &g, const LdsRec &in, LdsRec &out, int e0, int deg, int r, uint32_t rb, uint32_t rw, bool &odd, bool &flip,
                                        const int *tb = nullptr, Mid &&mid = Mid()) {
    uint32_t ad[DMAX];
    float l[DMAX], t[DMAX];
#pragma unroll
    for (int k = 0; k < DMAX; k++) {
        const int kk = (EXACT || k < deg) ? k : 0;
        if (DMAX <= 8 && tb) ad[k] = lds_addr(tb[2 * kk], tb[2 * kk + 1], r, rb, rw);
        else ad[k] = lds_addr(((ctab_t)g.tab)[2 * (e0 + kk)], ((ctab_t)g.tab)[2 * (e0 + kk) + 1], r, rb, rw);
    }
#pragma unroll
    for (int k = 0; k < DMAX; k++) l[k] = (float)*lds_cell(ad[k]);
    bool par = false;
    uint32_t X = 0;
    float m1 = INFINITY, m2 = INFINITY;
    const uint32_t oidx = in.meta >> 27;
#pragma unroll
    for (int k = 0; k < DMAX; k++) {
        if (EXACT || k < deg) {
            par ^= hard(l[k]);
            float old = 0.f;
            if constexpr (!FIRST) {
                const uint32_t mag = __float_as_uint(((uint32_t)k == oidx) ? in.c2 : in.c1);
                old = __uint_as_float(mag | ((in.meta << (32 - deg + k)) & 0x80000000u));
            }
            t[k] = l[k] - old;
            X ^= __float_as_uint(t[k]);
            const float a = fabsf(t[k]);
            m2 = __builtin_amdgcn_fmed3f(m1, m2, a);
            m1 = fminf(m1, a);
        } else t[k] = INFINITY;
    }
    odd |= par;
    mid();
    const float n1 = 0.75f * m1, n2 = 0.75f * m2;      // |(-3/4) * acc|: the one rounding of Min.hs:78
    // sign bit of message k = (deg odd) ^ (xor of all sign bits of t) ^ (sign bit of t_k)   (cn_update_padded)
    const uint32_t fl = (X ^ ((deg & 1) ? 0x80000000u : 0u)) & 0x80000000u;
    uint32_t c1 = __float_as_uint(n1) ^ fl, c2 = __float_as_uint(n2) ^ fl;
    asm volatile("" : "+v"(c1), "+v"(c2));           // (keeps the compiler from moving the sign flip behind every edge's select)
    uint32_t tsig = 0, nidx = 0;                        // tsig: the sign bits of t, edge k at bit deg - 1 - k
#pragma unroll
    for (int k = 0; k < DMAX; k++) {
        if (EXACT || k < deg) {
            const bool ismin = fabsf(t[k]) == m1;           // ties: n2 == n1, either answer gives the same message
            const uint32_t nmb = __builtin_amdgcn_bitop3_b32(ismin ? c2 : c1, __float_as_uint(t[k]), 0x80000000u, 0x78);   // a ^ (b & c)
            nidx = ismin ? (uint32_t)k : nidx;
            tsig = __builtin_amdgcn_alignbit(tsig, __float_as_uint(t[k]), 31);
            const _Float16 nw = sat16(t[k] + __uint_as_float(nmb));
            flip |= (nw > (_Float16)0) != hard(l[k]);
            *lds_cell(ad[k]) = nw;
        }
    }
    const uint32_t nsig = tsig ^ (fl ? ((1u << deg) - 1u) : 0u);
    out.c1 = n1; out.c2 = n2; out.meta = nsig | (nidx << 27);
}

template <int DCLASS, bool FIRST, class Mid = NoMid>
__device__ __forceinline__ void lds_layer_at(const LdsDev &g, const LdsRec &in, LdsRec &out, int e0, int deg, const int *tb, int r, uint32_t rb, uint32_t rw, bool &odd,
                                             bool &flip, Mid &&mid = Mid()) {
    switch (deg) {
        case 2: lds_row<2, true, FIRST>(g, in, out, e0, 2, r, rb, rw, odd, flip, tb, mid); return;
        case 3: lds_row<3, true, FIRST>(g, in, out, e0, 3, r, rb, rw, odd, flip, tb, mid); return;
        case 4: lds_row<4, true, FIRST>(g, in, out, e0, 4, r, rb, rw, odd, flip, tb, mid); return;
        case 5: lds_row<5, true, FIRST>(g, in, out, e0, 5, r, rb, rw, odd, flip, tb, mid); return;
        case 6: lds_row<6, true, FIRST>(g, in, out, e0, 6, r, rb, rw, odd, flip, tb, mid); return;
        case 7: lds_row<7, true, FIRST>(g, in, out, e0, 7, r, rb, rw, odd, flip, tb, mid); return;
        case 8: lds_row<8, true, FIRST>(g, in, out, e0, 8, r, rb, rw, odd, flip, tb, mid); return;
        default: break;
    }
    if (deg < 2) { out = in; mid(); return; }           // (an empty block row; weight 1 is refused at creation)
    if constexpr (DCLASS >= 20) {
        if (deg <= 12) { lds_row<12, false, FIRST>(g, in, out, e0, deg, r, rb, rw, odd, flip, nullptr, mid); return; }
        if (deg <= 20) { lds_row<20, false, FIRST>(g, in, out, e0, deg, r, rb, rw, odd, flip, nullptr, mid); return; }
    }
    if constexpr (DCLASS >= 32) lds_row<27, false, FIRST>(g, in, out, e0, deg, r, rb, rw, odd, flip, nullptr, mid);
    else mid();
}
template <int DCLASS, bool FIRST>
__device__ __forceinline__ void lds_layer(const LdsDev &g, const LdsRec &in, LdsRec &out, int layer, int r, uint32_t rb, uint32_t rw, bool &odd, bool &flip) {
    const int e0 = ((ctab_t)g.lbeg)[layer], deg = ((ctab_t)g.lbeg)[layer + 1] - e0;
    lds_layer_at<DCLASS, FIRST>(g, in, out, e0, deg, nullptr, r, rb, rw, odd, flip);
}

// ---- the pipelined loop's record traffic: issued and waited for by hand (an s_waitcnt the compiler derives for a loaded value that is
// consumed in the NEXT trip of a loop comes out as vmcnt(0) at the loop head: every record load, the one issued a layer ago included, and
// every record store would be drained each trip).  Each layer issues exactly ONE load (top) and ONE store (end), in that order, and
// nothing else that counts in vmcnt; so when layer l begins, the load of ITS record -- issued at the top of layer l - P -- is followed
// by P stores and P - 1 loads: s_waitcnt vmcnt(2 P - 1).  (The compiler does not know of loads in flight into q[]: nothing but the wait
// statement, which names the registers, may touch them, and they are drained before the loop is left.)
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void rec_load(u32x3 &dst, uint32_t voff, const LdsRec *sbase) {
    asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase));
}
__device__ __forceinline__ void rec_store(const u32x3 &src, uint32_t voff, LdsRec *sbase) {
    asm volatile("global_store_dwordx3 %0, %1, %2\n\ts_nop 1" : : "v"(voff), "v"(src), "s"(sbase));
}
template <int N> __device__ __forceinline__ void rec_wait(u32x3 &q) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(q) : "n"(N)); }

template <class F, int... Js> __device__ __forceinline__ bool first_true(F &f, std::integer_sequence<int, Js...>) {
    return (f(std::integral_constant<int, Js>{}) || ...);     // f<0>() || f<1>() || ...: stops at the first that says so
}

// the pipelined loop keeps LdsDev::ltab in LDS behind lam: every lane reads the same words (broadcast reads, served in order with the
// gathers -- a scalar load would share the LDS counter and return out of order, turning every wait on a gather into a wait for it too --
// and no scalar registers are held: the row code needs two lane masks per edge)
typedef int lds_int4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load_ltab(uint32_t ltab0, int layer, int &e0, int &deg, int (&tb)[16]) {
    const uint32_t a = ltab0 + (uint32_t)layer * (uint32_t)(kLtab * 4);
    typedef __attribute__((address_space(3))) const lds_int4 *p4_t;
    const lds_int4 h = *(p4_t)(uintptr_t)a;
    e0 = h.x; deg = h.y;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const lds_int4 v = *(p4_t)(uintptr_t)(a + 16u + 16u * (uint32_t)i);
        tb[4 * i] = v.x; tb[4 * i + 1] = v.y; tb[4 * i + 2] = v.z; tb[4 * i + 3] = v.w;
    }
}

// block = ceil(sz / 64) waves, thread r = row r of every block row; grid = resident workgroups (persistent)
// P = layers of record prefetch (0: the record is loaded where it is used, waits left to the compiler; this instance also writes traces)
// The pipelined instances must not spill: a scratch access counts in vmcnt like any other and would fall between a record load and its
// hand-counted wait (ecc_ldpc_amd/build.py refuses a build in which one does).  Rows above weight 8 keep up to 27 addresses, LLRs and
// differences per lane: those instances are built for at most 512 threads (256 registers per lane).
constexpr int lds_max_threads(int dclass, int p) { return (p > 0 && dclass > 8) ? 512 : 1024; }
// RW (pipelined instances): block rows of a group per sub.  One barrier per group is the kernel's fixed cost: every wave issues its
// gathers right behind it and the vector units idle until the first answers are back.  With RW = 2 a group holds twice the rows for
// the same number of waves, the second row's gathers are issued while the other waves of the SIMD still compute their first, and the
// barriers per sweep halve.
template <int DCLASS, int P, int RW = 1>
__global__ __launch_bounds__(lds_max_threads(DCLASS, P)) void layered_lds_kernel(LdsDev g, LdsRec *rec_all, LdsArgs A) {
    static_assert(P == 0 ? RW == 1 : P % RW == 0, "a record slot in flight per row of a sub");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16 *lam = reinterpret_cast<_Float16 *>(smem);
    // [0] next frame; [1..3] "sweep n moved" at 1 + n % 3 (cleared by thread 0 two sweeps before its use: every thread has
    // passed a barrier since the slot was last read); [4] "the channel's hard decisions are not a codeword"
    int *ctl = reinterpret_cast<int *>(smem + (((size_t)g.N * 2 + 15) & ~(size_t)15));
    const int T = blockDim.x, tid = threadIdx.x;
    const int nslot = P > 0 ? g.ng * g.gsz : g.nbr, TL = P > 0 ? g.tl : T;
    if constexpr (P > 0) {      // the per-slot graph entries, behind ctl (first read after the first frame's barriers)
        int *lt = ctl + 8;
        for (int i = tid; i < nslot * kLtab; i += T) lt[i] = g.gtab[i];
    }
    // which block row of a group this wave works on, and the thread's place among that block row's threads
    const int sub = P > 0 ? __builtin_amdgcn_readfirstlane(tid / TL) : 0, tin = tid - sub * TL;
    // the idle lanes of a block row's last wave shadow rows of THEIR OWN wave: they read what that row's lane reads in the same
    // instruction, compute and store the same values (and keep a record of their own), so no lane needs masking anywhere
    const int wb = tin & ~63, nlive = min(64, g.sz - wb);
    const int r = wb + (tin - wb) % nlive;
    const uint32_t lam0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;   // LDS byte address of lam
    const uint32_t rb = lam0 + 2u * (uint32_t)r, rw = rb - (uint32_t)g.two_sz;
    const uint32_t ltab0 = lam0 + (uint32_t)((((size_t)g.N * 2 + 15) & ~(size_t)15) + 32);   // (P > 0) the graph entries in LDS
    LdsRec *const wbase = rec_all + (size_t)blockIdx.x * nslot * TL;   // this workgroup's records: [slot or block row][thread of the block row]
    LdsRec *rec = wbase + tid;
    const bool tracing = P == 0 && A.trace != nullptr;
    int frame = blockIdx.x;
    while (frame < A.batch) {
        const size_t fN = (size_t)frame * g.N;
        // ---- lam <- channel LLRs, as stored: saturated, rounded to fp16 (eight per lane and request when the frame is 16-byte aligned)
        const bool wide = (g.N & 7) == 0 && A.llr_fmt != LLR_F64 && !A.final_lam &&
                          (((uintptr_t)A.llr + fN * (A.llr_fmt == LLR_F16 ? 2 : 4)) & 15) == 0 && (((uintptr_t)A.bits + fN) & 7) == 0;   // (uniform)
        if (wide) {
            with_llr_format(A.llr_fmt, [&](auto fmt) {
#pragma unroll 4
                for (int i = tid * 8; i < g.N; i += T * 8) {
                    half8 h;
                    load_llr8<decltype(fmt)::value>(A.llr, fN + i, h);
                    *reinterpret_cast<half8 *>(lam + i) = h;
                }
            });
        } else {
            with_llr_format(A.llr_fmt, [&](auto fmt) {
#pragma unroll 8
                for (int i = tid; i < g.N; i += T) lam[i] = sat16(load_llr_as<float, decltype(fmt)::value>(A.llr, fN + i));
            });
        }
        if (tid == 0) { ctl[1] = 0; ctl[2] = 0; ctl[3] = 0; ctl[4] = 0; }
        lds_barrier();
        bool conv = false;
        int n = 0;
        {   // syndrome of the hard decisions before the first sweep
            bool odd = false;
            for (int l0 = (P > 0 ? sub * RW : 0); l0 < nslot; l0 += (P > 0 ? g.gsz : 1))     // (P > 0: this wave's slots)
            for (int l = l0; l < l0 + RW; l++) {
                bool par = false;
                if constexpr (P > 0) {      // graph entries from the copy in LDS: eight gathers in flight
                    int e0, deg, tb[16];
                    load_ltab(ltab0, l, e0, deg, tb);
                    e0 = __builtin_amdgcn_readfirstlane(e0); deg = __builtin_amdgcn_readfirstlane(deg);
#pragma unroll
                    for (int k = 0; k < 8; k++)
                        if (k < deg) par ^= *lds_cell(lds_addr(tb[2 * k], tb[2 * k + 1], r, rb, rw)) > (_Float16)0;
                    for (int e = e0 + 8; e < e0 + deg; e++)
                        par ^= *lds_cell(lds_addr(((ctab_t)g.tab)[2 * e], ((ctab_t)g.tab)[2 * e + 1], r, rb, rw)) > (_Float16)0;
                } else {
                    const int e0 = ((ctab_t)g.lbeg)[l], e1 = ((ctab_t)g.lbeg)[l + 1];
                    for (int e = e0; e < e1; e++)
                        par ^= *lds_cell(lds_addr(((ctab_t)g.tab)[2 * e], ((ctab_t)g.tab)[2 * e + 1], r, rb, rw)) > (_Float16)0;
                }
                odd |= par;
            }
            if (__builtin_amdgcn_ballot_w64(odd) != 0 && (tid & 63) == 0) ctl[4] = 1;
            lds_barrier();
            conv = ctl[4] == 0;
        }
        if (tracing) {   // (uniform)
            for (int i = tid; i < g.N; i += T) A.trace[((size_t)frame * (A.max_iters + 1)) * g.N + i] = (double)(float)lam[i];
            lds_barrier();   // no wave starts layer 0 (which writes lam) while another still copies row 0
        }
        if constexpr (P == 0) {
            if (!conv) {
                for (n = 1; n <= A.max_iters; n++) {
                    bool odd = false, flip = false;
                    if (tid == 0) ctl[1 + (n + 1) % 3] = 0;                 // the flag of the NEXT sweep (last read two sweeps ago)
                    if (n == 1) {
                        const LdsRec none{0.f, 0.f, 0u};
                        for (int l = 0; l < g.nbr; l++) {
                            LdsRec out;
                            lds_layer<DCLASS, true>(g, none, out, l, r, rb, rw, odd, flip);
                            rec[(size_t)l * T] = out;
                            if (l == g.nbr - 1 && __builtin_amdgcn_ballot_w64(odd || flip) != 0 && (tid & 63) == 0) ctl[1 + n % 3] = 1;
                            lds_barrier();
                        }
                    } else {
                        for (int l = 0; l < g.nbr; l++) {
                            const LdsRec in = rec[(size_t)l * T];
                            LdsRec out;
                            lds_layer<DCLASS, false>(g, in, out, l, r, rb, rw, odd, flip);
                            rec[(size_t)l * T] = out;
                            if (l == g.nbr - 1 && __builtin_amdgcn_ballot_w64(odd || flip) != 0 && (tid & 63) == 0) ctl[1 + n % 3] = 1;
                            lds_barrier();
                        }
                    }
                    const bool any = ctl[1 + n % 3] != 0;
                    if (tracing) {
                        for (int i = tid; i < g.N; i += T) A.trace[((size_t)frame * (A.max_iters + 1) + n) * g.N + i] = (double)(float)lam[i];
                        lds_barrier();
                    }
                    if (!any) { conv = true; break; }
                }
                if (n > A.max_iters) n = A.max_iters;
            }
        } else if (!conv && A.max_iters > 0) {
            // ONE loop over (sweep, group), P groups per trip so that the record in flight for a group has a register slot of its own.  Every
            // group: wait for this group's record -> request the record of the group P ahead (wrapping into the next sweep; in the first sweep
            // the bytes are stale and unused: a zero record stands for "no messages yet") -> the rows, with the NEXT group's graph entries
            // requested half way -> store the new record -> barrier.
            u32x3 q[P];
#pragma unroll
            for (int j = 0; j < P; j++) q[j] = u32x3{0u, 0u, 0u};
            static_assert(P % 2 == 0, "the graph entries alternate between two register sets");
            // byte offsets of this wave's record slots inside the workgroup's area: where group gi stores, where group gi + P loads from
            const uint32_t rstep = (uint32_t)TL * (uint32_t)sizeof(LdsRec);                  // one slot
            const uint32_t gstep = (uint32_t)g.gsz * rstep, gend = gstep * (uint32_t)g.ng;   // one group, one sweep
            const uint32_t voff = (uint32_t)(sub * RW) * rstep + (uint32_t)tin * (uint32_t)sizeof(LdsRec);
            uint32_t soff = 0, loff = gstep * (uint32_t)(P / RW);       // (P / RW < ng)
            // (forming the NEXT group's LDS addresses before the barrier, from these entries, was measured: 59.9 vs 58.3 ms per 16 384 frames --
            //  the arithmetic lengthens the slowest wave's way to the barrier instead of hiding behind the other waves' gathers; not kept)
            // (tried with RW = 2: the three waves of a SIMD at three priorities (s_setprio), so that they drift apart and one wave's gathers
            //  fall behind another's arithmetic -- no difference, 108.7 ms either way: the vector units are busy ~80 % of the launch by the
            //  counters, what is left is not the waves meeting at the barrier)
            int gi = 0, e0v[2], degv[2], tbv[2][16];             // gi: the group being run; this wave's slot is gi * gsz + sub
            bool odd = false, flip = false;
            n = 1;
            load_ltab(ltab0, sub * RW, e0v[0], degv[0], tbv[0]);
            auto group_step = [&](auto J) -> bool {     // one slot of this sub -> the frame is finished
                constexpr int j = decltype(J)::value, cur = j & 1, nxt = cur ^ 1;
                constexpr int w = j % RW;               // which of the sub's block rows in the group (P is a multiple of RW)
                constexpr bool group_done = w == RW - 1;
                LDPC_TURN_LOOP();
                const int gn = (gi + 1 == g.ng) ? 0 : gi + 1;
                rec_wait<2 * P - 1>(q[j]);
                LdsRec in;
                in.c1 = n > 1 ? __uint_as_float(q[j].x) : 0.f; in.c2 = n > 1 ? __uint_as_float(q[j].y) : 0.f; in.meta = n > 1 ? q[j].z : 0u;
                rec_load(q[j], voff + loff, wbase);
                LdsRec out = in;
                const int deg = __builtin_amdgcn_readfirstlane(degv[cur]);
                auto next = [&]() { load_ltab(ltab0, (group_done ? gn : gi) * g.gsz + sub * RW + (group_done ? 0 : w + 1), e0v[nxt], degv[nxt], tbv[nxt]); };
                if (deg > 0) lds_layer_at<DCLASS, false>(g, in, out, __builtin_amdgcn_readfirstlane(e0v[cur]), deg, tbv[cur], r, rb, rw, odd, flip, next);
                else next();                            // (no block row for this wave in this group: it only keeps the counts)
                rec_store(u32x3{__float_as_uint(out.c1), __float_as_uint(out.c2), out.meta}, voff + soff, wbase);
                bool fin = false;
                if constexpr (!group_done) {            // the sub's next block row of the same group: no other wave touches its cells
                    soff += rstep; loff += rstep;
                } else {
                    soff += gstep - (uint32_t)(RW - 1) * rstep; soff -= soff >= gend ? gend : 0u;
                    loff += gstep - (uint32_t)(RW - 1) * rstep; loff -= loff >= gend ? gend : 0u;
                    const bool last = gi == g.ng - 1;
                    if (last && __builtin_amdgcn_ballot_w64(odd || flip) != 0 && (tid & 63) == 0) ctl[1 + n % 3] = 1;
                    lds_barrier();
                    if (last) {
                        if (ctl[1 + n % 3] == 0) { conv = true; fin = true; }
                        else if (n == A.max_iters) fin = true;
                        else { n++; odd = false; flip = false; if (tid == 0) ctl[1 + (n + 1) % 3] = 0; }   // (that flag was last read a sweep ago)
                    }
                    gi = gn;
                }
                return fin;
            };
            for (;;) {
                if (first_true(group_step, std::make_integer_sequence<int, P>{})) break;
            }
#pragma unroll
            for (int j = 0; j < P; j++) rec_wait<0>(q[j]);     // the requests for a sweep that will not run
        }
        // ---- result: hard(lam) of a frame that stopped by the rule, the channel's decisions (as stored: fp16) otherwise (Orig.hs:69-70)
        if (wide) {
            auto put = [&](int i, const half8 &h) {
                uint32_t lo = 0, hi = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) { lo |= (h[k] > (_Float16)0 ? 1u : 0u) << (8 * k); hi |= (h[k + 4] > (_Float16)0 ? 1u : 0u) << (8 * k); }
                *reinterpret_cast<uint2 *>(A.bits + fN + i) = make_uint2(lo, hi);
            };
            if (conv) {
#pragma unroll 4
                for (int i = tid * 8; i < g.N; i += T * 8) put(i, *reinterpret_cast<const half8 *>(lam + i));
            } else {
                with_llr_format(A.llr_fmt, [&](auto fmt) {
#pragma unroll 4
                    for (int i = tid * 8; i < g.N; i += T * 8) {
                        half8 h;
                        load_llr8<decltype(fmt)::value>(A.llr, fN + i, h);
                        put(i, h);
                    }
                });
            }
        } else {
            with_llr_format(A.llr_fmt, [&](auto fmt) {
#pragma unroll 4
                for (int i = tid; i < g.N; i += T) {
                    const float v = conv ? (float)lam[i] : (float)sat16(load_llr_as<float, decltype(fmt)::value>(A.llr, fN + i));
                    A.bits[fN + i] = v > 0.f ? 1 : 0;
                    if (A.final_lam) A.final_lam[fN + i] = (double)v;
                }
            });
        }
        if (tid == 0) {
            if (A.iters) A.iters[frame] = conv ? n : A.max_iters;
            if (A.conv) A.conv[frame] = conv ? 1 : 0;
            ctl[0] = (int)gridDim.x + atomicAdd(A.work_counter, 1);
        }
        lds_barrier();   // also: every lam read of this frame is done before the next frame's LLRs are written over it
        frame = ctl[0];
        lds_barrier();   // (nobody still reads ctl[0] when thread 0 of a fast wave writes the next one)
    }
}

// ------------------------------------------------------------------ host side
struct LayeredLdsState {
    int max_batch = 0, max_row_deg = 0, threads = 0, nbr = 0, grid = 0, prefetch = 0;   // threads: of ONE block row (tl)
    int ng = 0, gsz = 1, nslot = 0, rw = 1;        // groups of block rows run together (pipelined instances); rw: block rows of a group per sub
    size_t lds = 0;
    LdsDev g{};
    int32_t *d_tab = nullptr, *d_lbeg = nullptr, *d_ltab = nullptr;
    int *d_counter = nullptr;
    LdsRec *rec = nullptr;
    KernelTimer *timer = nullptr;
    LaunchInfo info;
};

static size_t lds_bytes_for(const ldpc_code &c) { return (((size_t)c.N * 2 + 15) & ~(size_t)15) + 32; }

const char *layered_lds_why_not(const ldpc_code &c, int variant, int dtype) {
    if (c.sz <= 0) return "code was not created from a quasi-cyclic description";
    if (c.sz > 1024) return "circulant size above 1024";
    if (dtype != LDPC_F16 || variant != LDPC_MINSUM) return "the lam-in-LDS layered kernel is min-sum with fp16 lam storage";
    if (c.max_row_deg > 27) return "check rows above weight 27";
    if (lds_bytes_for(c) > 160 * 1024) return "a frame's fp16 LLRs exceed the 160 KB of LDS";
    if ((int)c.layer_ptr.size() != c.block_rows + 1) return "layers were replaced: not the block rows";
    for (int br = 0; br <= c.block_rows; br++) if (c.layer_ptr[br] != br * c.sz) return "layers were replaced: not the block rows";
    const char *e = getenv("LDPC_LAYERED_LDS");
    if (e && !strcmp(e, "0")) return "disabled (LDPC_LAYERED_LDS=0)";
    return nullptr;
}

void layered_lds_destroy(LayeredLdsState *s) {
    if (!s) return;
    (void)hipFree(s->d_tab); (void)hipFree(s->d_lbeg); (void)hipFree(s->d_ltab); (void)hipFree(s->d_counter); (void)hipFree(s->rec);
    delete s;
}

template <int DCLASS, int P, int RW> static const void *kernel_ptr() { return (const void *)layered_lds_kernel<DCLASS, P, RW>; }
static const void *pick_kernel(int dclass, int p, int rw) {
    if (p == 0) return dclass == 8 ? kernel_ptr<8, 0, 1>() : dclass == 20 ? kernel_ptr<20, 0, 1>() : kernel_ptr<32, 0, 1>();
    if (rw == 2) return dclass == 8 ? kernel_ptr<8, 4, 2>() : dclass == 20 ? kernel_ptr<20, 4, 2>() : kernel_ptr<32, 4, 2>();
    return dclass == 8 ? kernel_ptr<8, 4, 1>() : dclass == 20 ? kernel_ptr<20, 4, 1>() : kernel_ptr<32, 4, 1>();
}

LayeredLdsState *layered_lds_create(const ldpc_code &c, int variant, int dtype, int max_batch) {
    const char *why = layered_lds_why_not(c, variant, dtype);
    if (why) { set_error(LDPC_EUNSUPPORTED, "%s", why); return nullptr; }
    LayeredLdsState *s = new (std::nothrow) LayeredLdsState();
    if (!s) { set_error(LDPC_ENOMEM, "out of host memory"); return nullptr; }
    try {
        s->max_batch = max_batch; s->max_row_deg = c.max_row_deg; s->nbr = c.block_rows;
        std::vector<int32_t> tab, lbeg(1, 0);
        for (int br = 0; br < c.block_rows; br++) {
            for (int bc = 0; bc < c.block_cols; bc++) {
                const int off = c.offsets[(size_t)br * c.block_cols + bc];
                if (off >= 0) { tab.push_back(2 * (bc * c.sz + off)); tab.push_back(c.sz - off); }
            }
            lbeg.push_back((int32_t)(tab.size() / 2));
        }
        s->g.sz = c.sz; s->g.nbr = c.block_rows; s->g.nbc = c.block_cols; s->g.N = c.N; s->g.two_sz = 2 * c.sz;
        s->threads = (c.sz + 63) / 64 * 64;
        // groups: maximal runs of CONSECUTIVE block rows that pairwise share no block column, up to what 1024 threads hold (and four):
        // the rows of a group touch distinct lam cells, so running them together gives what running them in order gives
        const char *ge = getenv("LDPC_LAYERED_LDS_GROUPS");             // =0: one block row at a time (A/B)
        const int dclass = c.max_row_deg <= 8 ? 8 : (c.max_row_deg <= 20 ? 20 : 32);
        const int tmax = lds_max_threads(dclass, 4);                    // what the pipelined instance of this row class is built for
        const bool no_groups = ge && !strcmp(ge, "0");
        auto groups_of = [&](int gcap) {
            std::vector<std::vector<int>> groups;
            std::vector<char> used((size_t)c.block_cols, 0);
            for (int br = 0; br < c.block_rows; br++) {
                bool clash = groups.empty() || (int)groups.back().size() >= gcap;
                for (int bc = 0; bc < c.block_cols && !clash; bc++) clash = c.offsets[(size_t)br * c.block_cols + bc] >= 0 && used[bc];
                if (clash) { groups.emplace_back(); std::fill(used.begin(), used.end(), 0); }
                groups.back().push_back(br);
                for (int bc = 0; bc < c.block_cols; bc++) if (c.offsets[(size_t)br * c.block_cols + bc] >= 0) used[bc] = 1;
            }
            return groups;
        };
        // one block row per sub and as many subs as the threads allow -- or, where that leaves groups of fewer than four, two block
        // rows per sub (RW = 2: twice the rows between two barriers for the same waves), if the matrix's order then yields a quarter
        // fewer groups (LDPC_LAYERED_LDS_RW=1|2 forces one form: A/B, tests)
        const int subs_max = std::max(1, tmax / s->threads);
        std::vector<std::vector<int>> groups = groups_of(no_groups ? 1 : std::min(4, subs_max));
        s->rw = 1;
        {
            const char *re = getenv("LDPC_LAYERED_LDS_RW");
            const int want = re ? atoi(re) : 0;
            if (!no_groups && want != 1 && (subs_max < 4 || want == 2)) {
                std::vector<std::vector<int>> g2 = groups_of(std::min(4, 2 * subs_max));
                if (want == 2 || 4 * g2.size() <= 3 * groups.size()) { groups.swap(g2); s->rw = 2; }
            }
        }
        s->ng = (int)groups.size(); s->gsz = 1;
        for (auto &gr : groups) s->gsz = std::max(s->gsz, (int)gr.size());
        s->gsz = (s->gsz + s->rw - 1) / s->rw * s->rw;       // whole subs
        s->nslot = s->ng * s->gsz;
        std::vector<int32_t> gtab((size_t)s->nslot * kLtab, 0);
        for (int gi = 0; gi < s->ng; gi++)
            for (int k = 0; k < (int)groups[gi].size(); k++) {
                const int br = groups[gi][k];
                int32_t *p = &gtab[((size_t)gi * s->gsz + k) * kLtab];
                p[0] = lbeg[br]; p[1] = lbeg[br + 1] - lbeg[br]; p[2] = br;
                for (int i = 0; i < 16 && 2 * (size_t)lbeg[br] + i < tab.size(); i++) p[4 + i] = tab[2 * (size_t)lbeg[br] + i];
            }
        s->g.ng = s->ng; s->g.gsz = s->gsz; s->g.tl = s->threads;
        s->lds = lds_bytes_for(c);
        const char *pe = getenv("LDPC_LAYERED_LDS_PREFETCH");           // =0: records loaded where they are used, one block row at a time (A/B)
        const size_t lds_pipe = s->lds + (size_t)s->nslot * kLtab * 4;  // + the per-slot graph entries
        // (more groups than records in flight: the record requested for group gi + P -- of the next sweep when that wraps -- must have been
        //  stored already in this sweep, so gi + P - ng < gi)
        s->prefetch = (s->ng > 4 && lds_pipe <= 160 * 1024 && s->threads <= tmax && !(pe && !strcmp(pe, "0"))) ? 4 : 0;
        if (s->prefetch) s->lds = lds_pipe; else { s->gsz = 1; s->rw = 1; s->nslot = c.block_rows; }
        const int subs = s->gsz / s->rw;                                // threads of a workgroup = subs * threads of a block row
        const void *kern = pick_kernel(dclass, s->prefetch, s->rw);
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds);
        if (e == hipSuccess && s->prefetch) e = hipFuncSetAttribute(pick_kernel(dclass, 0, 1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds);   // (traces)
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (e == hipSuccess) e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, s->threads * subs, s->lds);
        if (e == hipSuccess && per_cu <= 0) { set_error(LDPC_EHIP, "layered_lds: no workgroup of %d threads and %zu B of LDS is resident", s->threads * subs, s->lds); layered_lds_destroy(s); return nullptr; }
        if (e == hipSuccess) s->grid = std::min(max_batch, per_cu * prop.multiProcessorCount);
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_tab, sizeof(int32_t) * std::max<size_t>(tab.size(), 2));
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_lbeg, sizeof(int32_t) * lbeg.size());
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_ltab, sizeof(int32_t) * std::max<size_t>(gtab.size(), 1));
        if (e == hipSuccess) e = hipMalloc((void **)&s->d_counter, sizeof(int));
        if (e == hipSuccess) e = hipMalloc((void **)&s->rec, sizeof(LdsRec) * (size_t)s->grid * std::max(s->nslot, c.block_rows) * s->threads);
        if (e == hipSuccess && !tab.empty()) e = hipMemcpy(s->d_tab, tab.data(), sizeof(int32_t) * tab.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(s->d_lbeg, lbeg.data(), sizeof(int32_t) * lbeg.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess && !gtab.empty()) e = hipMemcpy(s->d_ltab, gtab.data(), sizeof(int32_t) * gtab.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            set_error(e == hipErrorOutOfMemory ? LDPC_ENOMEM : LDPC_EHIP, "layered_lds_create (%d workgroups x %zu bytes of records): %s", s->grid,
                      sizeof(LdsRec) * (size_t)c.block_rows * s->threads, hipGetErrorString(e));
            layered_lds_destroy(s);
            return nullptr;
        }
        s->g.tab = s->d_tab; s->g.lbeg = s->d_lbeg; s->g.gtab = s->d_ltab;
        snprintf(s->info.name, sizeof(s->info.name), "ldpc::layered_lds_kernel<%d, %d, %d>", dclass, s->prefetch, s->rw);
        s->info.threads = s->threads * subs; s->info.frames_per_wg = 1;
        return s;
    } catch (...) { layered_lds_destroy(s); set_error(LDPC_ENOMEM, "out of host memory"); return nullptr; }
}

void layered_lds_set_timer(LayeredLdsState *s, KernelTimer *t) { if (s) s->timer = t; }
const LaunchInfo &layered_lds_launch_info(const LayeredLdsState &s) { return s.info; }

int layered_lds_decode(LayeredLdsState &s, hipStream_t st, int max_iters, int batch, const void *d_llr, int llr_fmt, uint8_t *d_bits, int32_t *d_iters,
                       uint8_t *d_conv, double *d_final, double *d_trace) {
    LdsArgs a{};
    a.llr = d_llr; a.llr_fmt = llr_fmt; a.bits = d_bits; a.iters = d_iters; a.conv = d_conv; a.final_lam = d_final; a.trace = d_trace;
    a.batch = batch; a.max_iters = max_iters; a.work_counter = s.d_counter;
    hipError_t e = hipMemsetAsync(s.d_counter, 0, sizeof(int), st);
    if (e != hipSuccess) return set_error(LDPC_EHIP, "layered_lds: %s", hipGetErrorString(e));
    const bool simple = s.prefetch == 0 || d_trace;      // one block row at a time, waits left to the compiler; writes traces
    const dim3 grid(std::min(batch, s.grid)), block(simple ? s.threads : s.threads * (s.gsz / s.rw));
    const int dclass = s.max_row_deg <= 8 ? 8 : (s.max_row_deg <= 20 ? 20 : 32);
    if (s.timer) s.timer->begin(st);
#define LDS_LAUNCH(D, PF, RW) hipLaunchKernelGGL((layered_lds_kernel<D, PF, RW>), grid, block, s.lds, st, s.g, s.rec, a)
    if (simple) { if (dclass == 8) LDS_LAUNCH(8, 0, 1); else if (dclass == 20) LDS_LAUNCH(20, 0, 1); else LDS_LAUNCH(32, 0, 1); }
    else if (s.rw == 2) { if (dclass == 8) LDS_LAUNCH(8, 4, 2); else if (dclass == 20) LDS_LAUNCH(20, 4, 2); else LDS_LAUNCH(32, 4, 2); }
    else { if (dclass == 8) LDS_LAUNCH(8, 4, 1); else if (dclass == 20) LDS_LAUNCH(20, 4, 1); else LDS_LAUNCH(32, 4, 1); }
#undef LDS_LAUNCH
    if (s.timer) s.timer->end(st);
    e = hipGetLastError();
    if (e != hipSuccess) return set_error(LDPC_EHIP, "layered_lds launch: %s", hipGetErrorString(e));
    return LDPC_OK;
}

}  // namespace ldpc
