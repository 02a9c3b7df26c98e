// sim.h -- frame source / tally (sim.hip) interface
#pragma once
#include "internal.h"
namespace ldpc {
struct SimDev {
    int N, k, n_tx, kwords, pp;
    const uint32_t *gt;  // [kwords][pp]: word w of column j of G (packed over message bits) at w*pp + j; pp = p rounded
                         // up to a multiple of 4 (16-byte rows); null = all-zero codewords
};
int sim_generate(const SimDev &s, uint32_t *msgw, hipStream_t st, uint64_t seed, uint64_t first_frame, int batch,
                 double ebn0_db, void *d_llr, int llr_f16, uint8_t *d_msg);
int sim_tally(const SimDev &s, const uint32_t *msgw, hipStream_t st, int batch, const uint8_t *d_bits, const int32_t *d_iters,
              unsigned long long *d_tally);
}  // namespace ldpc
