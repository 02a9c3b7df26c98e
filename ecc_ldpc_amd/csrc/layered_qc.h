// layered_qc.h -- frame-per-workgroup layered decoder for quasi-cyclic codes with the state in HBM (layered_qc.hip)
#pragma once
#include "internal.h"

namespace ldpc {
struct LayeredQcState;
// flooding = 0: the layered schedule (layers = block rows); 1: the reference's flooding schedule with the same mapping
const char *layered_qc_why_not(const ldpc_code &c, int variant, int dtype, int flooding);
LayeredQcState *layered_qc_create(const ldpc_code &c, int variant, int dtype, int max_batch, int flooding);
void layered_qc_destroy(LayeredQcState *s);
void layered_qc_set_timer(LayeredQcState *s, KernelTimer *t);
const LaunchInfo &layered_qc_launch_info(const LayeredQcState &s);
int layered_qc_decode(LayeredQcState &s, hipStream_t st, int max_iters, int batch, const void *d_llr, int llr_fmt, uint8_t *d_bits, int32_t *d_iters,
                      uint8_t *d_conv, double *d_final, double *d_trace);
// lam on-chip as fp16, row records streamed from HBM (layered_lds.hip): layered min-sum with LDPC_F16 lam storage on QC codes whose
// frame fits LDS in fp16; layered_qc_create() uses it for such contexts
struct LayeredLdsState;
const char *layered_lds_why_not(const ldpc_code &c, int variant, int dtype);
LayeredLdsState *layered_lds_create(const ldpc_code &c, int variant, int dtype, int max_batch);
void layered_lds_destroy(LayeredLdsState *s);
void layered_lds_set_timer(LayeredLdsState *s, KernelTimer *t);
const LaunchInfo &layered_lds_launch_info(const LayeredLdsState &s);
int layered_lds_decode(LayeredLdsState &s, hipStream_t st, int max_iters, int batch, const void *d_llr, int llr_fmt, uint8_t *d_bits, int32_t *d_iters,
                       uint8_t *d_conv, double *d_final, double *d_trace);
int layered_qc_step(LayeredQcState &s, hipStream_t st, int batch, const double *d_orig, const double *d_lam, const double *d_ne, double *d_ne_out,
                    double *d_lam_out, uint8_t *d_syn);
}  // namespace ldpc
