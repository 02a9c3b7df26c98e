// fused_rows.h -- device building blocks shared by the per-edge-message fused kernels
// (fused_msg.hip: two waves per sz=128 frame; fused_split.hip: four waves, block rows split between pairs).
#pragma once
#include "fused_common.h"
#ifndef LDPC_DEVICE_ONLY
#include "generated_tables.h"
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
    static constexpr uint32_t CPW = QcGeom<SZ>::CPW, V = QcGeom<SZ>::V, ES = sizeof(CT);
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
    return qc_wrap((p4 + HSTEP * h) + lo, vmask);
}

// phase A for the RPL rows a lane owns in one block row of degree D.  msg: [RPL][D] registers.
template <typename CT, int VARIANT, int D, int RPL, int HSTEP, bool SYNDROME_ONLY, class Row>
__device__ __forceinline__ bool rows_a(const char *lds, Row tabrow, uint32_t p4, uint32_t vmask, CT *msg) {
    asm volatile("" : "+v"(p4));  // keeps the loop-invariant address arithmetic inside the turn loop, row by row
    if constexpr (SYNDROME_ONLY) LDPC_COLD_PATH();   // the one extra pass after the last update (Orig.hs:69-70)
    CT l[RPL][D];
    static_for<0, D>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const uint32_t lo = tabrow.lo(k), hi = tabrow.hi(k);
        uint32_t a0 = qc_wrap(p4 + lo, vmask);   // position inside the block column; `hi` (its base) is added below
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

// phase B: lam[col_k] <- ne'_k + lam[col_k]; the D x RPL targets of a block row are distinct columns
template <typename CT, int D, int RPL, int HSTEP, class Row>
__device__ __forceinline__ void rows_b(char *lds, Row tabrow, uint32_t p4, uint32_t vmask, const CT *msg) {
    asm volatile("" : "+v"(p4));
    CT cur[RPL][D];
    uint32_t adr[D];
    static_for<0, D>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const uint32_t lo = tabrow.lo(k), hi = tabrow.hi(k);
        adr[k] = qc_wrap(p4 + lo, vmask);
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

}  // namespace ldpc
