// sim.h -- frame source / tally (sim.hip) interface
#pragma once
#include "internal.h"
namespace ldpc {
struct SimDev {
    int N, k, n_tx, kwords, pp;
    const uint32_t *gt;  // dense generator, [kwords][pp]: word w of column j of G (packed over message bits) at w*pp + j; pp = p
                         // rounded up to a multiple of 4 (16-byte rows); null = no dense table
    // quasi-cyclic generator (Fast/Encoder.hs:26-63): rotation table for sim_parity_qc_kernel, null = none.
    //   qc_rot[cg][r][b][16]: for column group cg (16/W block columns, W = sz/32 words per circulant), block row r and bit
    //   rotation b = 0..31, the W words of rotl(g[r][c], b) for each column c of the group (missing columns: zero)
    const uint32_t *qc_rot;
    int qc_w, qc_brows, qc_bcols, qc_ncg, pwords;   // pwords = qc_bcols * qc_w: packed parity words per frame
};
// parw: scratch for the packed parity words [batch][pwords] (quasi-cyclic encoder only)
int sim_generate(const SimDev &s, uint32_t *msgw, uint32_t *parw, hipStream_t st, uint64_t seed, uint64_t first_frame, int batch,
                 double ebn0_db, void *d_out, int out_fmt, uint8_t *d_msg);   // out_fmt: 0 = f32 LLRs [batch][N], 1 = fp16 LLRs, 2 = codeword bytes [batch][n_tx]
int sim_tally(const SimDev &s, const uint32_t *msgw, hipStream_t st, int batch, const uint8_t *d_bits, const int32_t *d_iters,
              unsigned long long *d_tally);
// hard bits, one byte each [batch][N] -> packed [batch][ceil(N/8)], bit i of a frame in byte i / 8 at bit i % 8
int pack_bits(hipStream_t st, const uint8_t *d_bits, uint8_t *d_packed, int batch, int N);
}  // namespace ldpc
