// internal.h -- structures shared between the C ABI (api.cc) and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ldpc_hip.h"

namespace ldpc {
// element type of the [batch][N] channel-LLR array a decode entry point hands to the kernels
enum { LLR_F32 = 0, LLR_F64 = 1, LLR_F16 = 2 };


int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// device view of the graph + per-frame bookkeeping of the flood path (passed by value to kernels)
struct FloodDev {
    int M, N, E, Bp;
    int wide_rows;           // rows of weight 9..32 go to the padded-register CN instance (0: O(d^2) fallback, A/B)
    int cm_order;            // column-sum order (ldpc_sum_order): 0 foldr (+) orig (Orig.hs:96); 1 orig + foldr1 (+) (Fast/Arraylet.hs:185-186,
                             // CachedMult.hs:261-262); 2 orig + sum from 0 (Reference/Sparse.hs:112-114)
    int pairs4;              // tanh rule, f32 arithmetic, a plain-graph code whose heaviest row has weight <= 4: rows use ldpc_math.h
                             // cn_tanh_f32_pairs4, as the on-chip kernels' DMAX = 4 instances do (the paths stay bit-identical)
    int saturate;            // min-sum with f32 state: a frame is rescaled by 2^-40 when an LLR passes 2^60 (ldpc_math.h kRescales; flood_rescale_kernel)
    const int32_t *row_ptr;  // [M+1]
    const int32_t *col_idx;  // [E]   CSR, ascending column inside a row
    const int32_t *col_ptr;  // [N+1]
    const int32_t *csc_edge; // [E]   edge ids of column j, ascending row
    int32_t *unsat;          // [Bp]  stamp n+1 <=> syndrome of hard(lam_n) is non-zero
    int32_t *iters;          // [Bp]
    uint8_t *conv;           // [Bp]
    uint8_t *done;           // [Bp]
    int32_t *big;            // [Bp]  (saturate) set by the variable-node pass of a frame that has to be rescaled
    int32_t *kexp;           // [Bp]  (saturate) the frame's LLRs are 2^kexp x lam
};

// HIP-event bracket around every launch of a context's dominant kernel (bench.py roofline leg)
struct KernelTimer {
    bool enabled = false;
    std::vector<hipEvent_t> ev;  // pairs: start, stop
    size_t used = 0;
    void begin(hipStream_t st) {
        if (!enabled) return;
        if (used + 2 > ev.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { enabled = false; return; }
            ev.push_back(a); ev.push_back(b);
        }
        (void)hipEventRecord(ev[used], st);
    }
    void end(hipStream_t st) {
        if (!enabled) return;
        (void)hipEventRecord(ev[used + 1], st);
        used += 2;
    }
    int drain(int *launches, double *total_ms) {
        double t = 0;
        for (size_t i = 0; i < used; i += 2) {
            if (hipEventSynchronize(ev[i + 1]) != hipSuccess) return -1;
            float ms = 0;
            if (hipEventElapsedTime(&ms, ev[i], ev[i + 1]) != hipSuccess) return -1;
            t += ms;
        }
        if (launches) *launches = (int)(used / 2);
        if (total_ms) *total_ms = t;
        used = 0;
        return 0;
    }
    void destroy() { for (auto e : ev) (void)hipEventDestroy(e); ev.clear(); used = 0; }
};

// what the last decode launch of a context's dominant kernel was: its name as rocprofv3 / the assembly list it
// (demangled, up to the template arguments that select the instance) and its launch geometry.  bench.py uses it to
// look the kernel's instruction histogram up in build/isa_stats.json (tools/isa_histogram.py).
struct LaunchInfo {
    char name[192] = "";
    int threads = 0;         // per workgroup
    int frames_per_wg = 0;   // frames one workgroup decodes (0: not a frame-per-workgroup kernel)
};

struct FloodState {
    FloodDev dev;
    int variant, dtype;
    KernelTimer *timer = nullptr;
    bool has_wide_rows = false;   // some check row has weight 9..32 (not 18): second CN kernel instance, flood.hip
    void *msg = nullptr, *scratch = nullptr, *lam = nullptr, *orig = nullptr;
    // row-layered schedule (extension; flood.hip layered_kernel): layers as row ranges, device copy owned by the context
    bool layered = false;
    int n_layers = 0, max_row_deg = 0;
    int32_t *d_layer_ptr = nullptr;
    // The turn loop (2 launches per turn, no host decision inside: finished frames are frozen on the device)
    // touches only this context's buffers, so it is captured once per max_iters into a hipGraph and
    // replayed: one graph launch instead of 2*max_iters + 2 kernel launches.  flood_graph_release() frees it.
    hipGraphExec_t turn_graph = nullptr;
    int graph_iters = -1;
};
void flood_graph_release(FloodState &s);

int flood_decode(FloodState &s, hipStream_t st, int max_iters, int batch, const void *d_llr, int llr_fmt,
                 uint8_t *d_bits, double *d_final, double *d_trace);
int flood_step(FloodState &s, hipStream_t st, int batch, const double *d_orig, const double *d_lam,
               const double *d_ne, double *d_ne_out, double *d_lam_out, uint8_t *d_syn);
size_t flood_elem_size(int dtype);

}  // namespace ldpc

// host-side graph (immutable after creation, except for the lazily created per-device copies of its tables)
struct ldpc_code_dev {
    int32_t *row_ptr = nullptr, *col_idx = nullptr, *col_ptr = nullptr, *csc_edge = nullptr;
};
struct ldpc_code {
    int M = 0, N = 0, E = 0;
    std::vector<int32_t> row_ptr, col_idx, col_ptr, csc_edge;
    int max_row_deg = 0, min_row_deg = 0, max_col_deg = 0;
    // quasi-cyclic description when created through ldpc_code_create_qc (sz = 0 otherwise)
    int sz = 0, block_rows = 0, block_cols = 0;
    std::vector<int32_t> offsets;
    // layers of the row-layered schedule: row ranges [layer_ptr[l], layer_ptr[l+1]) whose rows share no column.
    // Default: the block rows of a QC code; every row its own layer otherwise (ldpc_code_set_layers replaces it).
    std::vector<int32_t> layer_ptr;
    // device copies, one set per HIP device (created by the first context on that device): a code may be shared by
    // replicas on several GPUs of one process (Utils.hs:53 replicateM maxThreadCount)
    std::mutex dev_mu;
    std::map<int, ldpc_code_dev> dev;
};
