// fused.h -- interface of the fused on-chip decoder (fused.hip): one launch decodes a batch,
// all BP state stays in LDS/registers between iterations.
#pragma once
#include "internal.h"

namespace ldpc {
struct FusedState;
bool fused_supported(const ldpc_code &code, int variant, int dtype);
const char *fused_why_not(const ldpc_code &code, int variant, int dtype);
// whether LDPC_PATH_AUTO should pick the fused kernel (it exists AND is the faster path today)
bool fused_preferred(const ldpc_code &code, int variant, int dtype);
FusedState *fused_create(const ldpc_code &code, int variant, int dtype, int max_batch);
// the row-layered schedule on-chip (an extension; fused_layered.hip): null = no reason, else why not
const char *fused_layered_why_not(const ldpc_code &code, int variant, int dtype);
FusedState *fused_layered_create(const ldpc_code &code, int variant, int dtype, int max_batch);
void fused_destroy(FusedState *s);
void fused_set_timer(FusedState *s, KernelTimer *t);
// d_llr [batch][N] float32/float64; outputs may be null except d_bits
int fused_decode(FusedState &s, hipStream_t st, int max_iters, int batch, const void *d_llr, int llr_fmt,
                 uint8_t *d_bits, int32_t *d_iters, uint8_t *d_conv, double *d_final, double *d_trace);
int fused_step(FusedState &s, hipStream_t st, int batch, const double *d_orig, const double *d_lam,
               const double *d_ne, double *d_ne_out, double *d_lam_out, uint8_t *d_syn);

// name of the kernel fused_decode launches for this state (as rocprofv3 lists it; without template arguments before
// the first launch, with the instance-selecting ones after it) and the geometry of that launch
const char *fused_kernel_name(const FusedState &s);
const LaunchInfo &fused_launch_info(const FusedState &s);
// whether the kernel fused_decode launches reads every channel LLR from memory exactly once (then the LLRs may sit in
// page-locked HOST memory and be read over PCIe by the kernel itself: api.cc zero-copy path)
bool fused_reads_llr_once(const FusedState &s, int max_iters);

// generic on-chip kernel for any H that fits in LDS (fused_csr.hip); reached through the functions above
struct CsrState;
const char *fused_csr_why_not(const ldpc_code &code, int variant, int dtype);
CsrState *fused_csr_create(const ldpc_code &code, int variant, int dtype);
void fused_csr_destroy(CsrState *s);
void fused_csr_set_timer(CsrState *s, KernelTimer *t);
const char *fused_csr_kernel_name(const CsrState &s);
const LaunchInfo &fused_csr_launch_info(const CsrState &s);
void fused_csr_set_round16(CsrState *s, int on);  // LDPC_F16 context: LLRs count as stored in fp16
int fused_csr_decode(CsrState &s, hipStream_t st, int max_iters, int batch, const void *d_llr, int llr_fmt, uint8_t *d_bits,
                     int32_t *d_iters, uint8_t *d_conv, double *d_final, double *d_trace);
int fused_csr_step(CsrState &s, hipStream_t st, int batch, const double *d_orig, const double *d_lam, const double *d_ne,
                   double *d_ne_out, double *d_lam_out, uint8_t *d_syn);
}  // namespace ldpc
