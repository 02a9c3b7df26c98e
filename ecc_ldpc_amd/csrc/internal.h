// internal.h -- structures shared between the C ABI (api.cc) and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/ldpc_hip.h"

namespace ldpc {

int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// device view of the graph + per-frame bookkeeping of the flood path (passed by value to kernels)
struct FloodDev {
    int M, N, E, Bp;
    const int32_t *row_ptr;  // [M+1]
    const int32_t *col_idx;  // [E]   CSR, ascending column inside a row
    const int32_t *col_ptr;  // [N+1]
    const int32_t *csc_edge; // [E]   edge ids of column j, ascending row
    int32_t *unsat;          // [Bp]  stamp n+1 <=> syndrome of hard(lam_n) is non-zero
    int32_t *iters;          // [Bp]
    uint8_t *conv;           // [Bp]
    uint8_t *done;           // [Bp]
};

struct FloodState {
    FloodDev dev;
    int variant, dtype;
    void *msg = nullptr, *scratch = nullptr, *lam = nullptr, *orig = nullptr;
};

int flood_decode(FloodState &s, hipStream_t st, int max_iters, int batch, const void *d_llr, int llr_is_f64,
                 uint8_t *d_bits, double *d_final, double *d_trace);
int flood_step(FloodState &s, hipStream_t st, int batch, const double *d_orig, const double *d_lam,
               const double *d_ne, double *d_ne_out, double *d_lam_out, uint8_t *d_syn);
size_t flood_elem_size(int dtype);

}  // namespace ldpc

// host-side graph (immutable after creation)
struct ldpc_code {
    int M = 0, N = 0, E = 0;
    std::vector<int32_t> row_ptr, col_idx, col_ptr, csc_edge;
    int max_row_deg = 0, min_row_deg = 0, max_col_deg = 0;
    // quasi-cyclic description when created through ldpc_code_create_qc (sz = 0 otherwise)
    int sz = 0, block_rows = 0, block_cols = 0;
    std::vector<int32_t> offsets;
    // device copies (created lazily by the first context on that device)
    int device = -1;
    int32_t *d_row_ptr = nullptr, *d_col_idx = nullptr, *d_col_ptr = nullptr, *d_csc_edge = nullptr;
};
