// fused.hip -- host side of the fused on-chip decoders (LDPC_PATH_FUSED): which kernel a context gets and how it is
// launched.  One launch decodes a batch; a frame's entire BP state stays on-chip for all iterations (lam in LDS,
// check->variable messages in VGPRs or LDS).  Kernels, in order of preference for a quasi-cyclic code:
//   (1) fused_split.hip   built-in instances of the four-wave split kernel for the shipped matrices (compile-time tables)
//   (2) jit.cc            the same kernel specialised at run time for any other single-circulant QC code
//   (3) fused_msg.hip     two-wave per-edge-message kernel, table-driven (AR4JA block structure; f64 parity mode;
//                         what runs when run-time compilation is off or unavailable)
//   (4) fused_csr.hip     generic on-chip kernel for any H whose frame fits in LDS
// (Round 1 also kept a compressed-record kernel here -- the reference's MinSum2/`omit` semigroup, Utils.hs:133-144, as
//  three registers per row; rebuilding messages twice per turn cost ~32 of its ~76 VALU clk per edge and it was removed
//  in round 2; DESIGN.md section 3.1 has its numbers.)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <utility>
#include <vector>

#include "fused.h"
#include "ldpc_math.h"
#include "fused_common.h"
#include "jit.h"

namespace ldpc {

// ------------------------------------------------------------------ host side
struct FusedState {
    int variant = 0, dtype = 0, max_batch = 0, sz = 0, M = 0, N = 0, E = 0;  // dtype: the COMPUTE type (f32/f64)
    int round16 = 0;          // LDPC_F16 context: fp16 channel LLRs, f32 state (nothing else of a fused decode lives in HBM)
    CsrState *csr = nullptr;  // set when the code has no QC plan: generic on-chip kernel (fused_csr.hip)
    JitKernel *jit = nullptr; // run-time specialised split kernel (jit.cc): any single-circulant QC code without a built-in instance
    int static_id = 0;    // compiled-in rotation table matching this code (0 = none: table-driven kernel)
    bool use_split = false;  // four waves per frame, block rows split between wave pairs (fused_split.hip)
    bool use_msg = true;  // per-edge-message two-wave kernel (fused_msg.hip)
    bool use_pk16 = false;   // LDPC_F16PK: packed fp16 arithmetic, two frames per lane (fused_pk16.hip)
    bool use_layered = false;   // LDPC_SCHED_LAYERED on-chip (fused_layered.hip)
    KernelTimer *timer = nullptr;
    LaunchInfo info;
    uint32_t *d_tab = nullptr;
    std::vector<int32_t> row_ptr;  // host copy for the step-mode record conversion
};

static bool plan_matches_ar4ja45(const ldpc_code &c) {
    if (c.sz <= 0 || c.block_rows != PlanAR4JA45::NBR || c.block_cols != PlanAR4JA45::NBC) return false;
    for (int br = 0; br < c.block_rows; br++) {
        int d = 0;
        for (int bc = 0; bc < c.block_cols; bc++) d += c.offsets[(size_t)br * c.block_cols + bc] >= 0;
        if (d != PlanAR4JA45::deg(br)) return false;
    }
    return true;
}

static int builtin_static_id(const ldpc_code &c) {
    std::vector<uint16_t> rot; std::vector<uint8_t> bcv;
    for (int br = 0; br < c.block_rows; br++)
        for (int bc = 0; bc < c.block_cols; bc++) {
            int off = c.offsets[(size_t)br * c.block_cols + bc];
            if (off >= 0) { rot.push_back((uint16_t)off); bcv.push_back((uint8_t)bc); }
        }
    return fused_msg_static_id(c.sz, rot.data(), bcv.data(), (int)rot.size());
}
static bool pk16_builtin(const ldpc_code &c, int variant) {
    return c.sz > 0 && plan_matches_ar4ja45(c) && fused_pk16_has(variant, c.sz, builtin_static_id(c));
}
// LDPC_F16PK: the built-in instances (compile-time tables of the shipped AR4JA matrices), any other single-circulant QC code
// through the run-time compiler (jit.cc JIT_PK16)
static const char *pk16_why_not(const ldpc_code &c, int variant) {
    if (variant != LDPC_MINSUM) return "the packed-fp16 kernel implements min-sum";
    if (c.sz == 0) return "the packed-fp16 kernel takes quasi-cyclic codes (built-in instances for codes/jpl.1024.4.5 and codes/jpl.4096.4.5, run-time specialised ones for any other single-circulant .q)";
    if (pk16_builtin(c, variant)) return nullptr;
    return jit_split_why_not(c, variant, LDPC_F16PK, JIT_PK16);
}

const char *fused_layered_why_not(const ldpc_code &c, int variant, int dtype) {
    if (variant != LDPC_MINSUM) return "the on-chip layered kernel implements min-sum";
    if (dtype != LDPC_F32 && dtype != LDPC_F16 && dtype != LDPC_F16PK) return "the on-chip layered kernels compute in f32 or packed fp16";
    if (c.sz == 0) return "the on-chip layered kernels take quasi-cyclic codes";
    if ((int)c.layer_ptr.size() != c.block_rows + 1) return "layers were replaced: not the block rows";
    for (int br = 0; br <= c.block_rows; br++) if (c.layer_ptr[br] != br * c.sz) return "layers were replaced: not the block rows";
    const char *e = getenv("LDPC_LAYERED_FUSED");
    if (e && !strcmp(e, "0")) return "disabled (LDPC_LAYERED_FUSED=0)";
    if (plan_matches_ar4ja45(c) && fused_layered_has(variant, dtype, c.sz, builtin_static_id(c))) return nullptr;
    // any other single-circulant QC code: the same bodies specialised at run time
    return dtype == LDPC_F16PK ? jit_split_why_not(c, variant, LDPC_F16PK, JIT_LAYERED_PK16) : jit_split_why_not(c, variant, LDPC_F32, JIT_LAYERED);
}
FusedState *fused_layered_create(const ldpc_code &c, int variant, int dtype, int max_batch) {
    const char *why = fused_layered_why_not(c, variant, dtype);
    if (why) { set_error(LDPC_EUNSUPPORTED, "%s", why); return nullptr; }
    FusedState *s = new (std::nothrow) FusedState();
    if (!s) { set_error(LDPC_ENOMEM, "out of host memory"); return nullptr; }
    s->use_layered = true; s->use_msg = false; s->round16 = dtype == LDPC_F16; s->use_pk16 = dtype == LDPC_F16PK;
    s->variant = variant; s->dtype = dtype == LDPC_F16PK ? LDPC_F16PK : LDPC_F32; s->max_batch = max_batch; s->sz = c.sz; s->M = c.M; s->N = c.N; s->E = c.E;
    if (!(plan_matches_ar4ja45(c) && fused_layered_has(variant, dtype, c.sz, builtin_static_id(c)))) {
        s->jit = dtype == LDPC_F16PK ? jit_split_create(c, variant, LDPC_F16PK, JIT_LAYERED_PK16) : jit_split_create(c, variant, LDPC_F32, JIT_LAYERED);
        if (!s->jit) { delete s; return nullptr; }
        snprintf(s->info.name, sizeof(s->info.name), "%s", s->jit->name.c_str());
        s->info.threads = s->jit->threads; s->info.frames_per_wg = s->jit->frames_per_wg;
    }
    return s;
}

static const char *plan_why_not(const ldpc_code &c, int variant, int dtype) {
    if (variant == LDPC_TANH && dtype != LDPC_F32) return "the fused tanh kernel exists for f32 only (f64 tanh: flood path)";
    if (dtype != LDPC_F32 && dtype != LDPC_F64) return "fused kernels exist for f32 and f64";
    if (c.sz == 0) return "code was not created from a quasi-cyclic description";
    if (!(c.sz == 32 || c.sz == 64 || c.sz == 128)) return "circulant size must be 32, 64 or 128";
    if (!plan_matches_ar4ja45(c)) return "block structure is not the AR4JA rate-4/5 plan (12x44 blocks, row weights 3,3,3,3,18x8)";
    return nullptr;
}
// LDPC_F16 = "fp16 storage in HBM, f32 arithmetic".  The only thing a fused decode keeps in HBM is the channel
// LLRs, so a fused F16 context is the f32 kernel fed fp16-rounded LLRs.
static inline int compute_dtype(int dtype) { return dtype == LDPC_F16 ? LDPC_F32 : dtype; }

// a fused (on-chip) kernel exists if the code matches a compiled QC plan, or failing that if a frame fits in LDS
const char *fused_why_not(const ldpc_code &c, int variant, int dtype) {
    if (dtype == LDPC_F16PK) return pk16_why_not(c, variant);
    dtype = compute_dtype(dtype);
    const char *p = plan_why_not(c, variant, dtype);
    if (!p) return nullptr;
    const char *j = jit_split_why_not(c, variant, dtype);
    if (!j) return nullptr;
    const char *g = fused_csr_why_not(c, variant, dtype);
    if (!g) return nullptr;
    static thread_local char buf[600];
    snprintf(buf, sizeof(buf), "built-in QC kernel: %s; run-time specialised QC kernel: %s; generic on-chip kernel: %s", p, j, g);
    return buf;
}
bool fused_supported(const ldpc_code &c, int variant, int dtype) { return fused_why_not(c, variant, dtype) == nullptr; }
// measured r01 (jpl.4096, 16384 frames): min-sum fused 10.0 vs flood 0.85 Gbit/s; tanh fused 2.09 vs flood 0.73
// (before the branch-free phi the fused tanh kernel spilled ~560 VGPRs and lost to flood: 0.53 vs 0.68).
bool fused_preferred(const ldpc_code &c, int variant, int dtype) { return fused_supported(c, variant, dtype); }

FusedState *fused_create(const ldpc_code &c, int variant, int dtype, int max_batch) {
    const char *why = fused_why_not(c, variant, dtype);
    if (why) { set_error(LDPC_EUNSUPPORTED, "%s", why); return nullptr; }
    FusedState *s = new (std::nothrow) FusedState();
    if (!s) { set_error(LDPC_ENOMEM, "out of host memory"); return nullptr; }
    if (dtype == LDPC_F16PK) {
        s->use_pk16 = true; s->use_msg = false;
        s->variant = variant; s->dtype = dtype; s->max_batch = max_batch; s->sz = c.sz; s->M = c.M; s->N = c.N; s->E = c.E;
        if (!pk16_builtin(c, variant)) {
            s->jit = jit_split_create(c, variant, LDPC_F16PK, JIT_PK16);
            if (!s->jit) { delete s; return nullptr; }
            snprintf(s->info.name, sizeof(s->info.name), "%s", s->jit->name.c_str());
            s->info.threads = s->jit->threads; s->info.frames_per_wg = s->jit->frames_per_wg;
        }
        return s;
    }
    s->round16 = dtype == LDPC_F16;
    dtype = compute_dtype(dtype);
    s->variant = variant; s->dtype = dtype; s->max_batch = max_batch; s->sz = c.sz; s->M = c.M; s->N = c.N; s->E = c.E;
    s->row_ptr = c.row_ptr;
    // Which kernel.  (1) a built-in instance with compile-time tables (the shipped matrices); (2) any other
    // single-circulant QC code: the same split kernel specialised at run time (jit.cc) -- unless that cannot be built
    // (compiler missing, shape out of range), in which case (3) the table-driven two-wave kernel if the block
    // structure is the AR4JA plan, else (4) the generic on-chip kernel.
    bool builtin = false;
    if (plan_why_not(c, variant, dtype) == nullptr && dtype == LDPC_F32) {
        std::vector<uint16_t> rot; std::vector<uint8_t> bcv;
        for (int br = 0; br < c.block_rows; br++)
            for (int bc = 0; bc < c.block_cols; bc++) {
                int off = c.offsets[(size_t)br * c.block_cols + bc];
                if (off >= 0) { rot.push_back((uint16_t)off); bcv.push_back((uint8_t)bc); }
            }
        const char *d = getenv("LDPC_FUSED_TABLE");
        builtin = !(d && !strcmp(d, "dyn")) && fused_msg_static_id(c.sz, rot.data(), bcv.data(), (int)rot.size()) != 0;
        if (d && !strcmp(d, "dyn")) builtin = true;   // forced table-driven kernel: not the run-time compiler either
    }
    {
        const char *k = getenv("LDPC_FUSED_KERNEL");   // A/B switches name a built-in kernel
        if (k && !strcmp(k, "msg")) builtin = builtin || plan_why_not(c, variant, dtype) == nullptr;
    }
    if (!builtin && jit_split_why_not(c, variant, dtype) == nullptr) {
        s->jit = jit_split_create(c, variant, dtype);
        if (s->jit) {
            snprintf(s->info.name, sizeof(s->info.name), "%s", s->jit->name.c_str());
            s->info.threads = s->jit->threads; s->info.frames_per_wg = s->jit->frames_per_wg;
            return s;
        }
        fprintf(stderr, "[libldpc_hip] run-time specialisation failed (%s); using a table-driven kernel\n", ldpc_last_error());
    }
    if (plan_why_not(c, variant, dtype) != nullptr) {  // no QC plan: generic on-chip kernel
        if (fused_csr_why_not(c, variant, dtype) != nullptr) {
            delete s;
            set_error(LDPC_EUNSUPPORTED, "no fused kernel could be built for this code (%s)", fused_why_not(c, variant, dtype) ? fused_why_not(c, variant, dtype) : "run-time compilation failed");
            return nullptr;
        }
        s->csr = fused_csr_create(c, variant, dtype);
        if (!s->csr) { delete s; return nullptr; }
        fused_csr_set_round16(s->csr, s->round16);
        return s;
    }
    s->use_msg = fused_msg_has(variant, dtype, c.sz);
    if (!s->use_msg) { delete s; set_error(LDPC_EUNSUPPORTED, "no fused kernel"); return nullptr; }
    const int es = dtype == LDPC_F64 ? 8 : 4;
    const int cpw = c.sz >= 64 ? 1 : 64 / c.sz, V = c.sz * cpw;
    std::vector<uint32_t> tab;
    {
        std::vector<uint16_t> rot; std::vector<uint8_t> bcv;
        for (int br = 0; br < c.block_rows; br++)
            for (int bc = 0; bc < c.block_cols; bc++) {
                int off = c.offsets[(size_t)br * c.block_cols + bc];
                if (off >= 0) { rot.push_back((uint16_t)off); bcv.push_back((uint8_t)bc); }
            }
        const char *d = getenv("LDPC_FUSED_TABLE");  // LDPC_FUSED_TABLE=dyn forces the table-driven kernel
        s->static_id = (d && !strcmp(d, "dyn")) ? 0 : fused_msg_static_id(c.sz, rot.data(), bcv.data(), (int)rot.size());
    }
    {   // LDPC_FUSED_KERNEL=msg keeps the two-wave kernel where the four-wave split kernel exists
        const char *k = getenv("LDPC_FUSED_KERNEL");
        s->use_split = s->use_msg && fused_split_has(variant, dtype, c.sz, s->static_id) && !(k && !strcmp(k, "msg"));
    }
    for (int br = 0; br < c.block_rows; br++)
        for (int bc = 0; bc < c.block_cols; bc++) {
            int off = c.offsets[(size_t)br * c.block_cols + bc];
            if (off < 0) continue;
            uint32_t lo = (uint32_t)(off * cpw * es), hi = (uint32_t)(bc * V * es);
            if (lo > 0xffffu || hi > 0xffffu) { delete s; set_error(LDPC_EUNSUPPORTED, "graph table field overflow"); return nullptr; }
            tab.push_back(lo | (hi << 16));
        }
    hipError_t e = hipMalloc((void **)&s->d_tab, tab.size() * 4);
    if (e == hipSuccess) e = hipMemcpy(s->d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) { set_error(LDPC_EHIP, "fused_create: %s", hipGetErrorString(e)); fused_destroy(s); return nullptr; }
    return s;
}

void fused_destroy(FusedState *s) {
    if (!s) return;
    jit_destroy(s->jit);
    fused_csr_destroy(s->csr);
    (void)hipFree(s->d_tab);
    delete s;
}

void fused_set_timer(FusedState *s, KernelTimer *t) { if (s) { s->timer = t; fused_csr_set_timer(s->csr, t); } }

bool fused_reads_llr_once(const FusedState &s, int max_iters) {
    return s.jit != nullptr || s.csr != nullptr || s.use_pk16 || s.use_layered || (s.use_split && max_iters <= kSplitMaxIters);
}

const LaunchInfo &fused_launch_info(const FusedState &s) { return s.csr ? fused_csr_launch_info(*s.csr) : s.info; }

static int launch_jit(FusedState &s, hipStream_t st, FusedArgs &a) {
    const int grid = (a.batch + s.jit->frames_per_wg - 1) / s.jit->frames_per_wg;
    void *params[] = {&a};
    if (s.timer && !a.step_mode) s.timer->begin(st);
    hipError_t e = hipModuleLaunchKernel(s.jit->fn, (unsigned)grid, 1, 1, (unsigned)s.jit->threads, 1, 1, 0, st, params, nullptr);
    if (s.timer && !a.step_mode) s.timer->end(st);
    if (e != hipSuccess) return set_error(LDPC_EHIP, "launch of %s: %s", s.jit->name.c_str(), hipGetErrorString(e));
    return LDPC_OK;
}

const char *fused_kernel_name(const FusedState &s) {
    const LaunchInfo &li = fused_launch_info(s);
    if (li.name[0]) return li.name;
    if (s.csr) return fused_csr_kernel_name(*s.csr);
    if (s.use_layered) return s.use_pk16 ? "fused_layered_pk16_kernel" : "fused_layered_kernel";
    if (s.use_pk16) return "fused_pk16_kernel";
    if (s.use_split) return "fused_split_kernel";
    return "fused_msg_kernel";
}

int fused_decode(FusedState &s, hipStream_t st, int max_iters, int batch, const void *d_llr, int llr_fmt, uint8_t *d_bits,
                 int32_t *d_iters, uint8_t *d_conv, double *d_final, double *d_trace) {
    if (s.csr) return fused_csr_decode(*s.csr, st, max_iters, batch, d_llr, llr_fmt, d_bits, d_iters, d_conv, d_final, d_trace);
    FusedArgs a{};
    a.tab = s.d_tab; a.llr = d_llr; a.llr_fmt = llr_fmt; a.llr_round16 = s.round16; a.bits = d_bits; a.iters = d_iters; a.conv = d_conv;
    a.final_lam = d_final; a.trace = d_trace; a.batch = batch; a.max_iters = max_iters; a.step_mode = 0;
    if (s.jit && (s.use_layered || s.use_pk16)) return launch_jit(s, st, a);   // (run-time instances: no limit on max_iters)
    if (s.use_layered) {
        if (max_iters > kSplitMaxIters) return set_error(LDPC_EUNSUPPORTED, "on-chip layered kernel: at most %d sweeps (a frame's result is packed into one register)", kSplitMaxIters);
        return s.use_pk16 ? fused_layered_pk16_launch(s.sz, st, a, s.timer, &s.info) : fused_layered_launch(s.sz, st, a, s.timer, &s.info);
    }
    if (s.use_pk16) {
        if (max_iters > kSplitMaxIters) return set_error(LDPC_EUNSUPPORTED, "LDPC_F16PK: at most %d iterations (a frame's result is packed into one register)", kSplitMaxIters);
        return fused_pk16_launch(s.sz, st, a, s.timer, &s.info);
    }
    if (s.jit) return launch_jit(s, st, a);
    // (the split kernel packs a frame's result into one register: 9 bits for the turn it converged at)
    if (s.use_split && max_iters <= kSplitMaxIters) return fused_split_launch(s.variant, s.sz, st, a, s.timer, &s.info);
    return fused_msg_launch(s.variant, s.dtype, s.sz, s.static_id, st, a, s.timer, &s.info);
}

int fused_step(FusedState &s, hipStream_t st, int batch, const double *d_orig, const double *d_lam, const double *d_ne,
               double *d_ne_out, double *d_lam_out, uint8_t *d_syn) {
    if (s.csr) return fused_csr_step(*s.csr, st, batch, d_orig, d_lam, d_ne, d_ne_out, d_lam_out, d_syn);
    if (s.use_layered) return set_error(LDPC_EUNSUPPORTED, "the on-chip layered kernel has no teacher-forced step (use path = LDPC_PATH_FLOOD: the same arithmetic, state in HBM)");
    if (s.use_pk16) return set_error(LDPC_EUNSUPPORTED, "LDPC_F16PK has no teacher-forced step (its state is not the reference's: use ldpc_decode_trace)");
    if (s.use_msg || s.jit) {  // per-edge messages: the state goes in and out as it is
        FusedArgs a{};
        a.tab = s.d_tab; a.llr = d_orig; a.llr_fmt = LLR_F64; a.llr_round16 = 0; a.batch = batch; a.max_iters = 1; a.step_mode = 1;
        a.st_lam = d_lam; a.st_ne_in = d_ne; a.st_ne_out = d_ne_out; a.final_lam = d_lam_out; a.st_syn = d_syn;
        if (s.jit) return launch_jit(s, st, a);
        if (s.use_split) return fused_split_launch(s.variant, s.sz, st, a, nullptr, nullptr);
        return fused_msg_launch(s.variant, s.dtype, s.sz, s.static_id, st, a, nullptr, nullptr);
    }
    return set_error(LDPC_EUNSUPPORTED, "no fused kernel for this context");
}

}  // namespace ldpc
