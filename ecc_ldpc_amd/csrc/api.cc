// api.cc -- the C ABI of include/ldpc_hip.h: graph objects, decoder replicas, host<->device plumbing.
// No C++ exception leaves this file; every failure becomes an LDPC_E* code + thread-local message.
#include <stdarg.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <new>

#include "internal.h"
#include "fused.h"
#include "jit.h"
#include "layered_qc.h"
#include "sim.h"

namespace ldpc {
static thread_local char g_err[512] = "";
static thread_local int g_err_code = 0;
int set_error(int code, const char *fmt, ...) {
    g_err_code = code;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    // messages may quote bytes of a damaged input file: keep the text printable ASCII
    for (char *q = g_err; *q; q++)
        if ((unsigned char)*q < 0x20 || (unsigned char)*q > 0x7e) *q = '?';
    return code;
}
}  // namespace ldpc
using ldpc::set_error;

#define HIPCHK(x)                                                                                \
    do {                                                                                         \
        hipError_t e_ = (x);                                                                     \
        if (e_ != hipSuccess) return set_error(LDPC_EHIP, "%s: %s", #x, hipGetErrorString(e_)); \
    } while (0)
#define HIPCHK_NULL(x)                                                               \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            set_error(LDPC_EHIP, "%s: %s", #x, hipGetErrorString(e_));               \
            return nullptr;                                                          \
        }                                                                            \
    } while (0)

struct ldpc_ctx {
    const ldpc_code *code = nullptr;
    int variant = 0, dtype = 0, max_batch = 0, Bp = 0, path = LDPC_PATH_FLOOD, device = 0, schedule = LDPC_SCHED_FLOODING;
    hipStream_t stream = nullptr;
    ldpc::FloodState flood;
    ldpc::FusedState *fused = nullptr;
    ldpc::LayeredQcState *lqc = nullptr;   // layered schedule on a QC code: frame-per-workgroup kernel, state in HBM (layered_qc.hip)
    // staging of the host-pointer entry points, allocated on first use.  kSlots slots, each with its own stream
    // running H2D -> decode -> D2H for one chunk, so that the copies of one chunk overlap the decode of another
    // (fused paths are stateless on the device; the flood path keeps per-context BP state and uses slot 0 with
    // the whole batch).  Measured on MI355X: a third slot changes nothing (65 536 jpl.4096 frames, pinned fp16
    // LLRs: 44.7 ms with 2 and with 3) -- the copies of both directions together run at ~25-32 GB/s, which is the
    // bound, not the pipeline depth.
    static constexpr int kSlots = 2;
    hipStream_t pstream[kSlots] = {};   // [0] aliases `stream`
    int slots = 0, chunk = 0;
    void *d_in[kSlots] = {};            // [chunk][N] float, double or half
    uint8_t *d_bits[kSlots] = {};       // [chunk][N]
    int32_t *d_iters[kSlots] = {};
    uint8_t *d_conv[kSlots] = {};
    double *d_final[kSlots] = {};       // [chunk][N], only when final LLRs are requested
    // latency path of the host-pointer entry points (<= kSmallFrames frames, e.g. ldpc_decode_one): page-locked
    // bounce buffers and ONE device block for bits|iters|converged, so a call costs one H2D copy, one launch and
    // one D2H copy instead of four pageable copies
    static constexpr int kSmallFrames = 16;
    void *h_small_in = nullptr, *d_small_in = nullptr;   // its own 16-frame device input: no full-size staging
    uint8_t *h_small_out = nullptr, *d_small_out = nullptr;
    // zero-copy path: outputs the caller gave as pageable memory while llr/bits are page-locked
    uint8_t *d_unpacked = nullptr;   // packed-result entry points: one byte per bit [max_batch][N], then packed (sim.hip pack_bits)
    uint8_t *d_packed = nullptr;     // ... and the packed image the host-pointer flavour copies back [max_batch][ceil(N/8)]
    int32_t *d_zc_iters = nullptr;
    uint8_t *d_zc_conv = nullptr;
    ldpc::KernelTimer timer;
};

// Device selection.  ldpc_init(d) validates device d and makes it the CALLING THREAD's device (thread-local); the first
// device any thread initialised is the process default for threads that never called ldpc_init.  Objects remember
// the device they were created on, so several threads can drive several GPUs from one process
// (Utils.hs:53,63-69: maxThreadCount replicas, chosen by thread).  ldpc_ctx_create_on names the device explicitly.
static std::mutex g_mu;
static int g_default_device = -1;
static uint64_t g_checked_mask = 0;          // devices that passed check_device (bit d)
static thread_local int t_device = -1;

static int check_device(int device) {
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (device >= 0 && device < 64 && ((g_checked_mask >> device) & 1u)) return LDPC_OK;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return set_error(LDPC_ENODEVICE, "no HIP device visible (%s); the HIP path has no CPU fallback",
                         e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device < 0 || device >= n || device >= 64) return set_error(LDPC_EINVAL, "device %d out of range [0,%d)", device, n);
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, device));
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
        return set_error(LDPC_ENODEVICE, "device %d is %s; libldpc_hip.so carries gfx950 code objects only", device, p.gcnArchName);
    std::lock_guard<std::mutex> lk(g_mu);
    g_checked_mask |= 1ull << device;
    return LDPC_OK;
}
static int current_device() {   // the calling thread's device, else the process default, else -1
    if (t_device >= 0) return t_device;
    std::lock_guard<std::mutex> lk(g_mu);
    return g_default_device;
}

extern "C" {

const char *ldpc_last_error(void) { return ldpc::g_err; }
int ldpc_last_error_code(void) { return ldpc::g_err_code; }
int ldpc_abi_version(void) { return 2; }

int ldpc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ldpc_init(int device) {
    int rc = check_device(device);
    if (rc != LDPC_OK) return rc;
    HIPCHK(hipSetDevice(device));
    t_device = device;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_default_device < 0) g_default_device = device;
    return LDPC_OK;
}

int ldpc_shutdown(void) {
    t_device = -1;
    std::lock_guard<std::mutex> lk(g_mu);
    g_default_device = -1;
    return LDPC_OK;
}

int ldpc_current_device(void) { return current_device(); }

// ------------------------------------------------------------------------------- graph
static int finish_code(ldpc_code *c) {
    c->E = c->row_ptr[c->M];
    c->col_ptr.assign((size_t)c->N + 1, 0);
    c->max_row_deg = 0;
    c->min_row_deg = c->M > 0 ? (1 << 30) : 0;
    for (int m = 0; m < c->M; m++) {
        int b = c->row_ptr[m], e = c->row_ptr[m + 1];
        if (e < b) return set_error(LDPC_EINVAL, "row_ptr not monotone at row %d", m);
        c->max_row_deg = std::max(c->max_row_deg, e - b);
        c->min_row_deg = std::min(c->min_row_deg, e - b);
        for (int q = b; q < e; q++) {
            int col = c->col_idx[q];
            if (col < 0 || col >= c->N) return set_error(LDPC_EINVAL, "column %d out of range in row %d", col, m);
            if (q > b && c->col_idx[q - 1] >= col) return set_error(LDPC_EINVAL, "columns of row %d not strictly ascending", m);
            c->col_ptr[col + 1]++;
        }
    }
    c->max_col_deg = 0;
    for (int j = 0; j < c->N; j++) {
        c->max_col_deg = std::max(c->max_col_deg, c->col_ptr[j + 1]);
        c->col_ptr[j + 1] += c->col_ptr[j];
    }
    // default layers: block rows of a QC code (column-disjoint: one circulant per block), single rows otherwise
    c->layer_ptr.clear();
    if (c->sz > 0) for (int br = 0; br <= c->block_rows; br++) c->layer_ptr.push_back(br * c->sz);
    else for (int m = 0; m <= c->M; m++) c->layer_ptr.push_back(m);
    c->csc_edge.assign((size_t)c->E, 0);
    std::vector<int32_t> fill(c->col_ptr.begin(), c->col_ptr.end() - 1);
    for (int m = 0; m < c->M; m++)
        for (int q = c->row_ptr[m]; q < c->row_ptr[m + 1]; q++) c->csc_edge[fill[c->col_idx[q]]++] = q;
    return LDPC_OK;
}

ldpc_code *ldpc_code_create_csr(int M, int N, const int32_t *row_ptr, const int32_t *col_idx) {
    if (M <= 0 || N <= 0 || M > (1 << 24) || N > (1 << 24) || !row_ptr || !col_idx || row_ptr[0] != 0 || row_ptr[M] < 0 ||
        row_ptr[M] > (1 << 30)) {
        set_error(LDPC_EINVAL, "ldpc_code_create_csr: bad arguments (M=%d N=%d)", M, N);
        return nullptr;
    }
    ldpc_code *c = new (std::nothrow) ldpc_code();
    if (!c) { set_error(LDPC_ENOMEM, "out of host memory"); return nullptr; }
    try {
        c->M = M; c->N = N;
        c->row_ptr.assign(row_ptr, row_ptr + M + 1);
        c->col_idx.assign(col_idx, col_idx + row_ptr[M]);
        if (finish_code(c) != LDPC_OK) { delete c; return nullptr; }
    } catch (...) { delete c; set_error(LDPC_ENOMEM, "out of host memory"); return nullptr; }
    return c;
}

ldpc_code *ldpc_code_create_qc(int sz, int block_rows, int block_cols, const int32_t *offsets) {
    if (sz <= 0 || block_rows <= 0 || block_cols <= 0 || !offsets) {
        set_error(LDPC_EINVAL, "ldpc_code_create_qc: bad arguments (sz=%d %dx%d)", sz, block_rows, block_cols);
        return nullptr;
    }
    // expanded sizes must stay well inside int32 (edge ids are int32): M, N <= 2^24, E <= 2^30
    const long long kMaxDim = 1ll << 24;
    if ((long long)sz * block_rows > kMaxDim || (long long)sz * block_cols > kMaxDim ||
        (long long)block_rows * block_cols > (1ll << 24)) {
        set_error(LDPC_EUNSUPPORTED, "ldpc_code_create_qc: %d x %d blocks of size %d expand beyond 2^24 rows/columns", block_rows, block_cols, sz);
        return nullptr;
    }
    long long nnz_blocks = 0;
    for (int i = 0; i < block_rows * block_cols; i++) nnz_blocks += offsets[i] >= 0;
    if (nnz_blocks * sz > (1ll << 30)) {
        set_error(LDPC_EUNSUPPORTED, "ldpc_code_create_qc: %lld edges exceed 2^30", nnz_blocks * sz);
        return nullptr;
    }
    for (int i = 0; i < block_rows * block_cols; i++)
        if (offsets[i] < -1 || offsets[i] >= sz) {
            set_error(LDPC_EINVAL, "offset %d at block %d outside [-1,%d)", offsets[i], i, sz);
            return nullptr;
        }
    ldpc_code *c = new (std::nothrow) ldpc_code();
    if (!c) { set_error(LDPC_ENOMEM, "out of host memory"); return nullptr; }
    try {
        c->sz = sz; c->block_rows = block_rows; c->block_cols = block_cols;
        c->offsets.assign(offsets, offsets + (size_t)block_rows * block_cols);
        c->M = sz * block_rows; c->N = sz * block_cols;
        c->row_ptr.assign((size_t)c->M + 1, 0);
        // row r of block-row br: one entry per non-empty block, column bc*sz + (r+off) mod sz
        // (QuasiCyclic.hs:19-25); block columns ascend, so columns ascend.
        for (int br = 0; br < block_rows; br++)
            for (int r = 0; r < sz; r++) {
                int m = br * sz + r;
                for (int bc = 0; bc < block_cols; bc++) {
                    int off = offsets[(size_t)br * block_cols + bc];
                    if (off >= 0) c->col_idx.push_back(bc * sz + (r + off) % sz);
                }
                c->row_ptr[m + 1] = (int32_t)c->col_idx.size();
            }
        if (finish_code(c) != LDPC_OK) { delete c; return nullptr; }
    } catch (...) { delete c; set_error(LDPC_ENOMEM, "out of host memory"); return nullptr; }
    return c;
}

static void code_free_device(ldpc_code *c) {
    std::lock_guard<std::mutex> lk(c->dev_mu);
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (auto &kv : c->dev) {
        if (hipSetDevice(kv.first) != hipSuccess) continue;
        (void)hipFree(kv.second.row_ptr); (void)hipFree(kv.second.col_idx); (void)hipFree(kv.second.col_ptr); (void)hipFree(kv.second.csc_edge);
    }
    c->dev.clear();
    if (prev >= 0) (void)hipSetDevice(prev);
}

void ldpc_code_destroy(ldpc_code *code) {
    if (!code) return;
    code_free_device(code);
    delete code;
}

int ldpc_code_dims(const ldpc_code *code, int *M, int *N, int *E) {
    if (!code) return set_error(LDPC_EINVAL, "null code");
    if (M) *M = code->M;
    if (N) *N = code->N;
    if (E) *E = code->E;
    return LDPC_OK;
}

int ldpc_qc_layer_order(int block_rows, int block_cols, const int32_t *offsets, int run, int32_t *perm) {
    if (block_rows <= 0 || block_cols <= 0 || !offsets || !perm || run < 1 || run > 8) return set_error(LDPC_EINVAL, "ldpc_qc_layer_order: bad arguments");
    try {
        const int Q = block_rows;
        auto hit = [&](int br, int bc) { return offsets[(size_t)br * block_cols + bc] >= 0; };
        std::vector<char> ok((size_t)Q * Q, 0);          // ok[i][j]: block rows i and j share no block column
        for (int i = 0; i < Q; i++)
            for (int j = i + 1; j < Q; j++) {
                bool share = false;
                for (int bc = 0; bc < block_cols && !share; bc++) share = hit(i, bc) && hit(j, bc);
                ok[(size_t)i * Q + j] = ok[(size_t)j * Q + i] = share ? 0 : 1;
            }
        std::vector<char> free_((size_t)Q, 1);
        std::vector<std::vector<int>> runs;
        int left = Q;
        while (left > 0) {
            // compatible rows still free, per free row; the run starts with the row that has the fewest, ties to the lower index
            std::vector<int> deg((size_t)Q, 0);
            for (int i = 0; i < Q; i++) if (free_[i]) for (int j = 0; j < Q; j++) if (free_[j] && ok[(size_t)i * Q + j]) deg[i]++;
            int first = -1;
            for (int i = 0; i < Q; i++) if (free_[i] && (first < 0 || deg[i] < deg[first])) first = i;
            std::vector<int> r{first};
            free_[first] = 0; left--;
            while ((int)r.size() < run) {
                int best = -1;
                for (int j = 0; j < Q; j++) {
                    if (!free_[j]) continue;
                    bool all = true;
                    for (int i : r) all = all && ok[(size_t)i * Q + j];
                    if (all && (best < 0 || deg[j] < deg[best])) best = j;
                }
                if (best < 0) break;
                r.push_back(best); free_[best] = 0; left--;
            }
            std::sort(r.begin(), r.end());
            runs.push_back(r);
        }
        std::stable_sort(runs.begin(), runs.end(), [](const std::vector<int> &a, const std::vector<int> &b) {
            return a.size() != b.size() ? a.size() > b.size() : a < b; });
        int n = 0, full = 0;
        for (auto &r : runs) { if ((int)r.size() == run) full++; for (int br : r) perm[n++] = br; }
        return full;
    } catch (...) { return set_error(LDPC_ENOMEM, "out of host memory"); }
}

int ldpc_code_set_layers(ldpc_code *code, int n_layers, const int32_t *layer_ptr) {
    if (!code || n_layers <= 0 || !layer_ptr) return set_error(LDPC_EINVAL, "ldpc_code_set_layers: bad arguments");
    {
        std::lock_guard<std::mutex> lk(code->dev_mu);
        if (!code->dev.empty()) return set_error(LDPC_EINVAL, "ldpc_code_set_layers: the code already has decoder contexts");
    }
    if (layer_ptr[0] != 0 || layer_ptr[n_layers] != code->M) return set_error(LDPC_EINVAL, "layers must cover rows 0..%d", code->M);
    try {
        std::vector<int32_t> seen((size_t)code->N, -1);
        for (int l = 0; l < n_layers; l++) {
            if (layer_ptr[l + 1] <= layer_ptr[l]) return set_error(LDPC_EINVAL, "layer %d is empty or out of order", l);
            for (int m = layer_ptr[l]; m < layer_ptr[l + 1]; m++)
                for (int q = code->row_ptr[m]; q < code->row_ptr[m + 1]; q++) {
                    if (seen[code->col_idx[q]] == l) return set_error(LDPC_EINVAL, "rows of layer %d share column %d", l, code->col_idx[q]);
                    seen[code->col_idx[q]] = l;
                }
        }
        code->layer_ptr.assign(layer_ptr, layer_ptr + n_layers + 1);
    } catch (...) { return set_error(LDPC_ENOMEM, "out of host memory"); }
    return LDPC_OK;
}

int ldpc_code_layers(const ldpc_code *code, int *n_layers, int32_t *layer_ptr) {
    if (!code) return set_error(LDPC_EINVAL, "null code");
    if (n_layers) *n_layers = (int)code->layer_ptr.size() - 1;
    if (layer_ptr) memcpy(layer_ptr, code->layer_ptr.data(), sizeof(int32_t) * code->layer_ptr.size());
    return LDPC_OK;
}

int ldpc_code_csr(const ldpc_code *code, int32_t *row_ptr, int32_t *col_idx) {
    if (!code || !row_ptr || !col_idx) return set_error(LDPC_EINVAL, "null argument");
    memcpy(row_ptr, code->row_ptr.data(), sizeof(int32_t) * ((size_t)code->M + 1));
    memcpy(col_idx, code->col_idx.data(), sizeof(int32_t) * (size_t)code->E);
    return LDPC_OK;
}

// the graph tables on `device` (uploaded by the first caller); the calling thread's current device must be `device`
static int code_upload(ldpc_code *c, int device, ldpc_code_dev *out) {
    std::lock_guard<std::mutex> lk(c->dev_mu);
    auto it = c->dev.find(device);
    if (it != c->dev.end()) { *out = it->second; return LDPC_OK; }
    ldpc_code_dev t;
    auto up = [&](int32_t **dst, const std::vector<int32_t> &v) -> int {
        size_t bytes = sizeof(int32_t) * std::max<size_t>(v.size(), 1);
        HIPCHK(hipMalloc((void **)dst, bytes));
        if (!v.empty()) HIPCHK(hipMemcpy(*dst, v.data(), sizeof(int32_t) * v.size(), hipMemcpyHostToDevice));
        return LDPC_OK;
    };
    int rc;
    if ((rc = up(&t.row_ptr, c->row_ptr)) || (rc = up(&t.col_idx, c->col_idx)) ||
        (rc = up(&t.col_ptr, c->col_ptr)) || (rc = up(&t.csc_edge, c->csc_edge))) {
        (void)hipFree(t.row_ptr); (void)hipFree(t.col_idx); (void)hipFree(t.col_ptr); (void)hipFree(t.csc_edge);   // a half-uploaded graph is not kept
        return rc;
    }
    c->dev[device] = t;
    *out = t;
    return LDPC_OK;
}

// ------------------------------------------------------------------------------- contexts
void ldpc_ctx_destroy(ldpc_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    ldpc::flood_graph_release(ctx->flood);
    hipFree(ctx->flood.msg); hipFree(ctx->flood.scratch); hipFree(ctx->flood.lam); hipFree(ctx->flood.orig);
    hipFree(ctx->flood.dev.unsat); hipFree(ctx->flood.dev.iters); hipFree(ctx->flood.dev.conv); hipFree(ctx->flood.dev.done);
    (void)hipFree(ctx->flood.dev.big); (void)hipFree(ctx->flood.dev.kexp);
    (void)hipFree(ctx->flood.d_layer_ptr);
    for (int i = 1; i < ldpc_ctx::kSlots; i++) if (ctx->pstream[i]) hipStreamSynchronize(ctx->pstream[i]);
    for (int i = 0; i < ldpc_ctx::kSlots; i++) { hipFree(ctx->d_in[i]); hipFree(ctx->d_bits[i]); hipFree(ctx->d_iters[i]); hipFree(ctx->d_conv[i]); hipFree(ctx->d_final[i]); }
    for (int i = 1; i < ldpc_ctx::kSlots; i++) if (ctx->pstream[i]) hipStreamDestroy(ctx->pstream[i]);
    if (ctx->h_small_in) (void)hipHostFree(ctx->h_small_in);
    if (ctx->h_small_out) (void)hipHostFree(ctx->h_small_out);
    (void)hipFree(ctx->d_small_out); (void)hipFree(ctx->d_small_in);
    (void)hipFree(ctx->d_zc_iters); (void)hipFree(ctx->d_zc_conv); (void)hipFree(ctx->d_unpacked); (void)hipFree(ctx->d_packed);
    if (ctx->fused) ldpc::fused_destroy(ctx->fused);
    ldpc::layered_qc_destroy(ctx->lqc);
    ctx->timer.destroy();
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

ldpc_ctx *ldpc_ctx_create_ex(const ldpc_code *code, int variant, int dtype, int max_batch, int path) {
    return ldpc_ctx_create_on(code, -1, variant, dtype, max_batch, path);   // -1: the calling thread's device
}

ldpc_ctx *ldpc_ctx_create_on(const ldpc_code *code, int device, int variant, int dtype, int max_batch, int path) {
    ldpc_ctx_config cfg{};
    cfg.struct_size = sizeof(cfg); cfg.device = device; cfg.variant = variant; cfg.dtype = dtype; cfg.max_batch = max_batch; cfg.path = path;
    cfg.schedule = LDPC_SCHED_FLOODING;
    return ldpc_ctx_create_cfg(code, &cfg);
}

const ldpc_code *ldpc_ctx_code(const ldpc_ctx *ctx) { return ctx ? ctx->code : nullptr; }
int ldpc_ctx_max_batch(const ldpc_ctx *ctx) { return ctx ? ctx->max_batch : set_error(LDPC_EINVAL, "null ctx"); }
int ldpc_ctx_device(const ldpc_ctx *ctx) { return ctx ? ctx->device : set_error(LDPC_EINVAL, "null ctx"); }
int ldpc_ctx_schedule(const ldpc_ctx *ctx) { return ctx ? ctx->schedule : set_error(LDPC_EINVAL, "null ctx"); }

ldpc_ctx *ldpc_ctx_create_cfg(const ldpc_code *code_c, const ldpc_ctx_config *cfg) {
    if (!cfg || cfg->struct_size < offsetof(ldpc_ctx_config, schedule)) { set_error(LDPC_EINVAL, "ldpc_ctx_create_cfg: bad config"); return nullptr; }
    ldpc_code *code = const_cast<ldpc_code *>(code_c);
    const int variant = cfg->variant, dtype = cfg->dtype, max_batch = cfg->max_batch, path = cfg->path;
    // every field added after `path` is read when struct_size covers THAT field (a caller built against an older header passes the
    // size its structure had: 32 bytes with `schedule` as its last member before `sum_order` was added)
    const int schedule = cfg->struct_size >= offsetof(ldpc_ctx_config, schedule) + sizeof(int) ? cfg->schedule : LDPC_SCHED_FLOODING;
    if (schedule != LDPC_SCHED_FLOODING && schedule != LDPC_SCHED_LAYERED) { set_error(LDPC_EINVAL, "unknown schedule %d", schedule); return nullptr; }
    const int sum_order = cfg->struct_size >= offsetof(ldpc_ctx_config, sum_order) + sizeof(int) ? cfg->sum_order : LDPC_SUM_REFERENCE;
    if (sum_order != LDPC_SUM_REFERENCE && sum_order != LDPC_SUM_ARRAYLET && sum_order != LDPC_SUM_SPARSE) { set_error(LDPC_EINVAL, "unknown sum order %d", sum_order); return nullptr; }
    if (sum_order != LDPC_SUM_REFERENCE && (schedule != LDPC_SCHED_FLOODING || path == LDPC_PATH_FUSED || cfg->dtype == LDPC_F16 || cfg->dtype == LDPC_F16PK)) {
        set_error(LDPC_EUNSUPPORTED, "LDPC_SUM_ARRAYLET / LDPC_SUM_SPARSE are parity modes: flooding schedule, flood path, f32 or f64");
        return nullptr;
    }
    if (variant == LDPC_TANH_CM && (dtype != LDPC_F64 || schedule != LDPC_SCHED_FLOODING || path == LDPC_PATH_FUSED)) {
        set_error(LDPC_EUNSUPPORTED, "LDPC_TANH_CM (arraylet-cm numerics) is a parity mode: f64, flooding schedule, flood path (an f32 kernel is 1e-5 away from either tanh flavour)");
        return nullptr;
    }
    if (variant == LDPC_TANH_CUDA32 && (dtype != LDPC_F32 || schedule != LDPC_SCHED_FLOODING || path == LDPC_PATH_FUSED || sum_order != LDPC_SUM_REFERENCE ||
                                        (code && code->max_row_deg > 32))) {
        set_error(LDPC_EUNSUPPORTED, "LDPC_TANH_CUDA32 (cuda-arraylet2 numerics) is a parity mode: f32, flooding schedule, flood path, rows up to weight 32");
        return nullptr;
    }
    if (!code || max_batch <= 0 || (variant != LDPC_TANH && variant != LDPC_MINSUM && variant != LDPC_TANH_CM && variant != LDPC_TANH_CUDA32) ||
        (dtype != LDPC_F32 && dtype != LDPC_F64 && dtype != LDPC_F16 && dtype != LDPC_F16PK) ||
        (path != LDPC_PATH_AUTO && path != LDPC_PATH_FLOOD && path != LDPC_PATH_FUSED)) {
        set_error(LDPC_EINVAL, "ldpc_ctx_create: bad arguments (variant=%d dtype=%d max_batch=%d path=%d)", variant, dtype, max_batch, path);
        return nullptr;
    }
    if (variant == LDPC_MINSUM && code->min_row_deg == 1) {
        set_error(LDPC_EDEGREE, "min-sum on a check row of degree 1 (the reference's foldr1 min' fails on [], Min.hs:79)");
        return nullptr;
    }
    int device = cfg->device;
    if (device < 0) device = current_device();
    if (device < 0) { set_error(LDPC_ENODEVICE, "ldpc_init() has not succeeded: no GPU bound (there is no CPU fallback)"); return nullptr; }
    if (check_device(device) != LDPC_OK) return nullptr;
    HIPCHK_NULL(hipSetDevice(device));
    ldpc_code_dev tabs;
    if (code_upload(code, device, &tabs) != LDPC_OK) return nullptr;

    if (dtype == LDPC_F16PK && (variant != LDPC_MINSUM || path == LDPC_PATH_FLOOD)) {
        set_error(LDPC_EUNSUPPORTED, "LDPC_F16PK (packed fp16 arithmetic, two frames per lane) exists for min-sum on the on-chip path");
        return nullptr;
    }
    const bool layered_fused_ok = schedule == LDPC_SCHED_LAYERED && sum_order == LDPC_SUM_REFERENCE && ldpc::fused_layered_why_not(*code, variant, dtype) == nullptr;
    if (schedule == LDPC_SCHED_LAYERED) {
        if (dtype == LDPC_F16 && (!layered_fused_ok || path == LDPC_PATH_FLOOD)) {
            // from HBM: lam stored in fp16 for the frame-per-workgroup min-sum record kernel of QC codes (r03); nothing else
            const char *why = sum_order == LDPC_SUM_REFERENCE ? ldpc::layered_qc_why_not(*code, variant, dtype, 0) : "parity modes are f64";
            if (why) { set_error(LDPC_EUNSUPPORTED, "the layered schedule from HBM with fp16 storage: %s", why); return nullptr; }
        }
        if (code->max_row_deg > 32) { set_error(LDPC_EUNSUPPORTED, "layered schedule: check rows above weight 32 (this code has %d)", code->max_row_deg); return nullptr; }
    }
    const bool fused_ok = layered_fused_ok ||
                          (schedule == LDPC_SCHED_FLOODING && variant != LDPC_TANH_CM && variant != LDPC_TANH_CUDA32 && sum_order == LDPC_SUM_REFERENCE && ldpc::fused_supported(*code, variant, dtype));
    if ((path == LDPC_PATH_FUSED || dtype == LDPC_F16PK) && !fused_ok) {
        if (schedule == LDPC_SCHED_LAYERED) { set_error(LDPC_EUNSUPPORTED, "no on-chip kernel for the layered schedule on this code / rule / type (%s); LDPC_PATH_FLOOD keeps the state in HBM", ldpc::fused_layered_why_not(*code, variant, dtype)); return nullptr; }
        set_error(LDPC_EUNSUPPORTED, "no fused kernel for this code/variant/dtype (%s)", ldpc::fused_why_not(*code, variant, dtype));
        return nullptr;
    }
    ldpc_ctx *ctx = new (std::nothrow) ldpc_ctx();
    if (!ctx) { set_error(LDPC_ENOMEM, "out of host memory"); return nullptr; }
    ctx->code = code; ctx->variant = variant; ctx->dtype = dtype; ctx->max_batch = max_batch; ctx->device = device; ctx->schedule = schedule;
    ctx->Bp = (max_batch + 63) / 64 * 64;
    ctx->path = (path == LDPC_PATH_AUTO) ? ((fused_ok && (layered_fused_ok || ldpc::fused_preferred(*code, variant, dtype))) ? LDPC_PATH_FUSED : LDPC_PATH_FLOOD) : path;

#define CTX_HIP(x)                                                       \
    do {                                                                 \
        hipError_t e_ = (x);                                             \
        if (e_ != hipSuccess) {                                          \
            set_error(e_ == hipErrorOutOfMemory ? LDPC_ENOMEM : LDPC_EHIP, "%s: %s", #x, hipGetErrorString(e_)); \
            ldpc_ctx_destroy(ctx);                                       \
            return nullptr;                                              \
        }                                                                \
    } while (0)

    CTX_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    const size_t es = ldpc::flood_elem_size(dtype);
    const size_t Bp = (size_t)ctx->Bp;
    ldpc::FloodDev &d = ctx->flood.dev;
    d.M = code->M; d.N = code->N; d.E = code->E; d.Bp = ctx->Bp;
    d.row_ptr = tabs.row_ptr; d.col_idx = tabs.col_idx; d.col_ptr = tabs.col_ptr; d.csc_edge = tabs.csc_edge;
    d.unsat = nullptr; d.iters = nullptr; d.conv = nullptr; d.done = nullptr;
    d.wide_rows = 0;
    d.cm_order = variant == LDPC_TANH_CM ? LDPC_SUM_ARRAYLET : (variant == LDPC_TANH_CUDA32 ? 3 : sum_order);   // (3: ((orig + ne_1) + ne_2) + ..., common.h:161-171)
    d.saturate = (variant == LDPC_MINSUM && dtype == LDPC_F32) ? 1 : 0;   // (fp16 storage saturates at +-65504 by its own rule)
    d.big = nullptr; d.kexp = nullptr;
    // ONE predicate for both paths: rows of weight <= 4 take the pair-product form of the tanh rule exactly when the on-chip path of
    // this code is the generic kernel (whose DMAX = 4 instance is written that way) -- a plain graph, or a QC description the
    // split family does not take (circulant size below 16, LDPC_JIT=0, ...).  Such a QC code then also runs its flood path on the
    // batch-major kernels, which know the form; a QC code of the split family uses the chained form everywhere (flood_qc_kernel too).
    d.pairs4 = (variant == LDPC_TANH && dtype != LDPC_F64 && code->max_row_deg <= 4 &&
                (code->sz == 0 || ldpc::jit_split_why_not(*code, variant, LDPC_F32) != nullptr)) ? 1 : 0;
    ctx->flood.variant = variant; ctx->flood.dtype = dtype; ctx->flood.timer = &ctx->timer;
    // (the staging buffers of the host-pointer entry points are allocated on first use: a context driven
    //  through ldpc_decode_batch_dev with 65 536 frames would otherwise park 3 GB of HBM)
    const int qc_flooding = schedule == LDPC_SCHED_FLOODING ? 1 : 0;
    if (ctx->path == LDPC_PATH_FLOOD && sum_order == LDPC_SUM_REFERENCE && variant != LDPC_TANH_CUDA32 && !(d.pairs4 && qc_flooding) && ldpc::layered_qc_why_not(*code, variant, dtype, qc_flooding) == nullptr) {
        // QC code, either schedule: one workgroup per frame, state in HBM (a frame stops when ITS rule fires);
        // any other H, fp16 storage and the arraylet-cm parity mode: the batch-major kernels below
        ctx->lqc = ldpc::layered_qc_create(*code, variant, dtype, max_batch, qc_flooding);
        if (!ctx->lqc) { ldpc_ctx_destroy(ctx); return nullptr; }
        ldpc::layered_qc_set_timer(ctx->lqc, &ctx->timer);
    } else if (ctx->path == LDPC_PATH_FLOOD) {
        CTX_HIP(hipMalloc(&ctx->flood.msg, std::max<size_t>((size_t)code->E, 1) * Bp * es));
        // scratch is only touched by rows whose degree has no register kernel
        bool need_scratch = false;
        const char *wz = getenv("LDPC_FLOOD_WIDE");   // LDPC_FLOOD_WIDE=0: rows of weight 9..32 through the O(d^2) fallback (A/B)
        const bool paddable = !(variant != LDPC_MINSUM && dtype == LDPC_F64) && !(wz && !strcmp(wz, "0"));
        for (int m = 0; m < code->M; m++) {
            int dg = code->row_ptr[m + 1] - code->row_ptr[m];
            if (dg <= 8 || dg == 18) continue;
            if (paddable && dg <= 32) { ctx->flood.has_wide_rows = true; d.wide_rows = 1; }   // padded register rows (second CN instance)
            else need_scratch = true;                                    // O(d^2) fallback writes through scratch
        }
        if (need_scratch) CTX_HIP(hipMalloc(&ctx->flood.scratch, std::max<size_t>((size_t)code->E, 1) * Bp * es));
        CTX_HIP(hipMalloc(&ctx->flood.lam, (size_t)code->N * Bp * es));
        CTX_HIP(hipMalloc(&ctx->flood.orig, (size_t)code->N * Bp * es));
        CTX_HIP(hipMalloc((void **)&d.unsat, sizeof(int32_t) * Bp));
        CTX_HIP(hipMalloc((void **)&d.iters, sizeof(int32_t) * Bp));
        CTX_HIP(hipMalloc((void **)&d.conv, Bp));
        CTX_HIP(hipMalloc((void **)&d.done, Bp));
        if (d.saturate) { CTX_HIP(hipMalloc((void **)&d.big, sizeof(int32_t) * Bp)); CTX_HIP(hipMalloc((void **)&d.kexp, sizeof(int32_t) * Bp)); }
        if (schedule == LDPC_SCHED_LAYERED) {
            ctx->flood.layered = true;
            ctx->flood.n_layers = (int)code->layer_ptr.size() - 1;
            ctx->flood.max_row_deg = code->max_row_deg;
            CTX_HIP(hipMalloc((void **)&ctx->flood.d_layer_ptr, sizeof(int32_t) * code->layer_ptr.size()));
            CTX_HIP(hipMemcpy(ctx->flood.d_layer_ptr, code->layer_ptr.data(), sizeof(int32_t) * code->layer_ptr.size(), hipMemcpyHostToDevice));
        }
    } else {
        ctx->fused = schedule == LDPC_SCHED_LAYERED ? ldpc::fused_layered_create(*code, variant, dtype, max_batch) : ldpc::fused_create(*code, variant, dtype, max_batch);
        if (!ctx->fused && schedule == LDPC_SCHED_LAYERED && path == LDPC_PATH_AUTO && dtype != LDPC_F16PK &&
            ldpc::layered_qc_why_not(*code, variant, dtype, 0) == nullptr) {
            // the on-chip layered kernel of this code is compiled at run time and that failed (no compiler on this host, or it
            // rejected the instance): under LDPC_PATH_AUTO the context keeps its state in HBM instead, as fused_create() falls back
            // to its table-driven kernels for the flooding schedule
            fprintf(stderr, "libldpc_hip: on-chip layered kernel unavailable (%s); the context runs the layered schedule from HBM\n", ldpc_last_error());
            ctx->path = LDPC_PATH_FLOOD;
            ctx->lqc = ldpc::layered_qc_create(*code, variant, dtype, max_batch, 0);
            if (!ctx->lqc) { ldpc_ctx_destroy(ctx); return nullptr; }
            ldpc::layered_qc_set_timer(ctx->lqc, &ctx->timer);
            return ctx;
        }
        if (!ctx->fused) { ldpc_ctx_destroy(ctx); return nullptr; }
        ldpc::fused_set_timer(ctx->fused, &ctx->timer);
    }
    return ctx;
}

ldpc_ctx *ldpc_ctx_create(const ldpc_code *code, int variant, int dtype, int max_batch) {
    return ldpc_ctx_create_ex(code, variant, dtype, max_batch, LDPC_PATH_AUTO);
}

int ldpc_ctx_path(const ldpc_ctx *ctx) { return ctx ? ctx->path : set_error(LDPC_EINVAL, "null ctx"); }

void *ldpc_host_alloc(size_t bytes) {
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) { set_error(LDPC_ENOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e)); return nullptr; }
    return p;
}
void ldpc_host_free(void *p) { if (p) (void)hipHostFree(p); }

int ldpc_ctx_synchronize(ldpc_ctx *ctx) {
    if (!ctx) return set_error(LDPC_EINVAL, "null ctx");
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return LDPC_OK;
}

// ------------------------------------------------------------------------------- decode
static int check_call(ldpc_ctx *ctx, int max_iters, int batch) {
    if (!ctx) return set_error(LDPC_EINVAL, "null ctx");
    if (max_iters < 0) return set_error(LDPC_EINVAL, "max_iters %d < 0", max_iters);
    if (batch < 0 || batch > ctx->max_batch) return set_error(LDPC_EINVAL, "batch %d outside [0,%d]", batch, ctx->max_batch);
    HIPCHK(hipSetDevice(ctx->device));
    return LDPC_OK;
}

// device-side core: d_llr is [batch][N] of element type fmt (ldpc::LLR_F32 / LLR_F64 / LLR_F16); outputs device pointers (may be null)
static int decode_dev(ldpc_ctx *ctx, hipStream_t st, int max_iters, int batch, const void *d_llr, int fmt,
                      uint8_t *d_bits, int32_t *d_iters, uint8_t *d_conv, double *d_final, double *d_trace) {
    if (batch == 0) return LDPC_OK;
    if (ctx->path == LDPC_PATH_FUSED)
        return ldpc::fused_decode(*ctx->fused, st, max_iters, batch, d_llr, fmt, d_bits, d_iters, d_conv, d_final, d_trace);
    if (ctx->lqc) return ldpc::layered_qc_decode(*ctx->lqc, st, max_iters, batch, d_llr, fmt, d_bits, d_iters, d_conv, d_final, d_trace);
    int rc = ldpc::flood_decode(ctx->flood, st, max_iters, batch, d_llr, fmt, d_bits, d_final, d_trace);
    if (rc != LDPC_OK) return rc;
    if (d_iters) HIPCHK(hipMemcpyAsync(d_iters, ctx->flood.dev.iters, sizeof(int32_t) * (size_t)batch, hipMemcpyDeviceToDevice, st));
    if (d_conv) HIPCHK(hipMemcpyAsync(d_conv, ctx->flood.dev.conv, (size_t)batch, hipMemcpyDeviceToDevice, st));
    return LDPC_OK;
}

// frames per pipelined chunk of the host-pointer entry points (fused paths)
static constexpr int kHostChunk = 8192;

static int ensure_staging(ldpc_ctx *ctx, bool want_final) {
    const size_t N = (size_t)ctx->code->N;
    if (!ctx->d_in[0]) {
        const bool pipelined = ctx->path == LDPC_PATH_FUSED && ctx->max_batch > kHostChunk;
        ctx->chunk = pipelined ? kHostChunk : ctx->max_batch;
        ctx->slots = pipelined ? std::min(ldpc_ctx::kSlots, (ctx->max_batch + kHostChunk - 1) / kHostChunk) : 1;
        ctx->pstream[0] = ctx->stream;
        hipError_t e = hipSuccess;
        for (int i = 1; i < ctx->slots && e == hipSuccess; i++) e = hipStreamCreateWithFlags(&ctx->pstream[i], hipStreamNonBlocking);
        for (int i = 0; i < ctx->slots && e == hipSuccess; i++) {
            const size_t c = (size_t)ctx->chunk;
            e = hipMalloc(&ctx->d_in[i], c * N * sizeof(double));
            if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_bits[i], c * N);
            if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_iters[i], sizeof(int32_t) * c);
            if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_conv[i], c);
        }
        if (e != hipSuccess) {
            for (int i = 0; i < ldpc_ctx::kSlots; i++) {
                (void)hipFree(ctx->d_in[i]); (void)hipFree(ctx->d_bits[i]); (void)hipFree(ctx->d_iters[i]); (void)hipFree(ctx->d_conv[i]);
                ctx->d_in[i] = nullptr; ctx->d_bits[i] = nullptr; ctx->d_iters[i] = nullptr; ctx->d_conv[i] = nullptr;
                if (i > 0 && ctx->pstream[i]) { (void)hipStreamDestroy(ctx->pstream[i]); ctx->pstream[i] = nullptr; }
            }
            ctx->slots = 0;
            return set_error(LDPC_ENOMEM, "staging buffers for %d frames: %s", ctx->chunk, hipGetErrorString(e));
        }
    }
    if (want_final)
        for (int i = 0; i < ctx->slots; i++)
            if (!ctx->d_final[i]) HIPCHK(hipMalloc((void **)&ctx->d_final[i], (size_t)ctx->chunk * N * sizeof(double)));
    return LDPC_OK;
}

// device-visible address of page-locked host memory (hipHostMalloc / hipHostRegister), or nullptr for pageable memory
static void *pinned_device_ptr(const void *p) {
    if (!p) return nullptr;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (at.type != hipMemoryTypeHost || !at.devicePointer) return nullptr;
    return at.devicePointer;
}

static int decode_host(ldpc_ctx *ctx, int max_iters, int batch, const void *llr, int fmt, uint8_t *bits,
                       int32_t *iters, uint8_t *converged, double *final_lam, double *trace_lam) {
    int rc = check_call(ctx, max_iters, batch);
    if (rc != LDPC_OK) return rc;
    if (batch == 0) return LDPC_OK;
    if (!llr || !bits) return set_error(LDPC_EINVAL, "null llr/bits");
    // Zero-copy: with page-locked llr and bits (ldpc_host_alloc) and a kernel that reads every LLR once, the decode
    // kernel itself reads the LLRs and writes the bits over PCIe -- no staging copies, the transfer hides under the
    // compute of the other workgroups.  Measured (65 536 jpl.4096 frames, min-sum): f32 26.9 ms vs 57.9 ms through
    // the chunked copy pipeline, fp16 LLRs 24.6 vs 44.7 ms (copies issued next to the decode kernel did not overlap
    // with it on this platform: pipeline time = copy time + kernel time).
    // (Not flood_qc_kernel: it re-reads the channel LLRs from the input in every variable-node pass, N values per frame
    // and turn, which over PCIe would be ~70 GB for 65 536 jpl.4096 frames.  The batch-major flood kernels copy the
    // input to their own `orig` once, the layered kernel reads it once into lam.)
    const bool reads_once = ctx->path == LDPC_PATH_FLOOD ? (!ctx->lqc || ctx->schedule == LDPC_SCHED_LAYERED)
                                                         : ldpc::fused_reads_llr_once(*ctx->fused, max_iters);
    if (!final_lam && !trace_lam && batch > ldpc_ctx::kSmallFrames && reads_once) {
        void *z_llr = pinned_device_ptr(llr), *z_bits = pinned_device_ptr(bits);
        if (z_llr && z_bits) {
            int32_t *z_it = (int32_t *)pinned_device_ptr(iters);
            uint8_t *z_cv = (uint8_t *)pinned_device_ptr(converged);
            if (iters && !z_it) {
                if (!ctx->d_zc_iters) HIPCHK(hipMalloc((void **)&ctx->d_zc_iters, sizeof(int32_t) * (size_t)ctx->max_batch));
                z_it = ctx->d_zc_iters;
            }
            if (converged && !z_cv) {
                if (!ctx->d_zc_conv) HIPCHK(hipMalloc((void **)&ctx->d_zc_conv, (size_t)ctx->max_batch));
                z_cv = ctx->d_zc_conv;
            }
            hipStream_t st = ctx->stream;
            rc = decode_dev(ctx, st, max_iters, batch, z_llr, fmt, (uint8_t *)z_bits, z_it, z_cv, nullptr, nullptr);
            hipError_t e = hipSuccess;
            if (rc == LDPC_OK && iters && z_it == ctx->d_zc_iters) e = hipMemcpyAsync(iters, z_it, sizeof(int32_t) * (size_t)batch, hipMemcpyDeviceToHost, st);
            if (rc == LDPC_OK && e == hipSuccess && converged && z_cv == ctx->d_zc_conv) e = hipMemcpyAsync(converged, z_cv, (size_t)batch, hipMemcpyDeviceToHost, st);
            hipError_t es2 = hipStreamSynchronize(st);
            if (e == hipSuccess) e = es2;
            if (rc == LDPC_OK && e != hipSuccess) rc = set_error(LDPC_EHIP, "decode (zero-copy): %s", hipGetErrorString(e));
            return rc;
        }
    }
    const size_t N = (size_t)ctx->code->N, es = fmt == ldpc::LLR_F64 ? 8 : (fmt == ldpc::LLR_F16 ? 2 : 4);
    if (batch <= ldpc_ctx::kSmallFrames && !final_lam && !trace_lam) {   // latency path (allocates 16 frames, never the full-size staging)
        const size_t cap = (size_t)ldpc_ctx::kSmallFrames;
        const size_t out_cap = cap * N + cap * sizeof(int32_t) + cap + 16;
        if (!ctx->h_small_in) {
            hipError_t e = hipHostMalloc(&ctx->h_small_in, cap * N * sizeof(double), hipHostMallocDefault);
            if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->h_small_out, out_cap, hipHostMallocDefault);
            if (e == hipSuccess) e = hipMalloc((void **)&ctx->d_small_out, out_cap);
            if (e == hipSuccess) e = hipMalloc(&ctx->d_small_in, cap * N * sizeof(double));
            if (e != hipSuccess) {
                if (ctx->h_small_in) (void)hipHostFree(ctx->h_small_in);
                if (ctx->h_small_out) (void)hipHostFree(ctx->h_small_out);
                (void)hipFree(ctx->d_small_out); (void)hipFree(ctx->d_small_in);
                ctx->h_small_in = nullptr; ctx->h_small_out = nullptr; ctx->d_small_out = nullptr; ctx->d_small_in = nullptr;
                return set_error(LDPC_ENOMEM, "latency-path buffers: %s", hipGetErrorString(e));
            }
        }
        const size_t nb = (size_t)batch, in_bytes = nb * N * es;
        const size_t off_it = (nb * N + 3) / 4 * 4, off_cv = off_it + nb * sizeof(int32_t), out_bytes = off_cv + nb;
        memcpy(ctx->h_small_in, llr, in_bytes);
        hipStream_t st = ctx->stream;
        hipError_t e = hipMemcpyAsync(ctx->d_small_in, ctx->h_small_in, in_bytes, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            rc = decode_dev(ctx, st, max_iters, batch, ctx->d_small_in, fmt, ctx->d_small_out, (int32_t *)(ctx->d_small_out + off_it),
                            ctx->d_small_out + off_cv, nullptr, nullptr);
            if (rc == LDPC_OK) e = hipMemcpyAsync(ctx->h_small_out, ctx->d_small_out, out_bytes, hipMemcpyDeviceToHost, st);
        }
        hipError_t es2 = hipStreamSynchronize(st);   // also after an error: nothing may still be in flight
        if (e == hipSuccess) e = es2;
        if (rc == LDPC_OK && e != hipSuccess) rc = set_error(LDPC_EHIP, "decode: %s", hipGetErrorString(e));
        if (rc != LDPC_OK) return rc;
        memcpy(bits, ctx->h_small_out, nb * N);
        if (iters) memcpy(iters, ctx->h_small_out + off_it, nb * sizeof(int32_t));
        if (converged) memcpy(converged, ctx->h_small_out + off_cv, nb);
        return LDPC_OK;
    }
    if ((rc = ensure_staging(ctx, final_lam != nullptr)) != LDPC_OK) return rc;
    double *d_trace = nullptr;
    const size_t turns = (size_t)max_iters + 1;
    if (trace_lam) {  // verification path: one device buffer for the whole batch
        size_t tb = (size_t)batch * turns * N * sizeof(double);
        hipError_t e = hipMalloc((void **)&d_trace, tb);
        if (e != hipSuccess) return set_error(LDPC_ENOMEM, "trace buffer of %zu bytes: %s", tb, hipGetErrorString(e));
        (void)hipMemsetAsync(d_trace, 0, tb, ctx->stream);
        (void)hipStreamSynchronize(ctx->stream);
    }
    hipError_t e = hipSuccess;
    int slot = 0;
    for (int f0 = 0; f0 < batch && rc == LDPC_OK && e == hipSuccess; f0 += ctx->chunk, slot = (slot + 1) % ctx->slots) {
        const int nb = std::min(ctx->chunk, batch - f0);
        hipStream_t st = ctx->pstream[slot];
        // stream order protects the slot: this copy is queued behind the slot's previous D2H
        e = hipMemcpyAsync(ctx->d_in[slot], (const char *)llr + (size_t)f0 * N * es, (size_t)nb * N * es, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) break;
        rc = decode_dev(ctx, st, max_iters, nb, ctx->d_in[slot], fmt, ctx->d_bits[slot], ctx->d_iters[slot], ctx->d_conv[slot],
                        final_lam ? ctx->d_final[slot] : nullptr, d_trace ? d_trace + (size_t)f0 * turns * N : nullptr);
        if (rc != LDPC_OK) break;
        e = hipMemcpyAsync(bits + (size_t)f0 * N, ctx->d_bits[slot], (size_t)nb * N, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && iters) e = hipMemcpyAsync(iters + f0, ctx->d_iters[slot], sizeof(int32_t) * (size_t)nb, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && converged) e = hipMemcpyAsync(converged + f0, ctx->d_conv[slot], (size_t)nb, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && final_lam) e = hipMemcpyAsync(final_lam + (size_t)f0 * N, ctx->d_final[slot], (size_t)nb * N * sizeof(double), hipMemcpyDeviceToHost, st);
    }
    for (int i = 0; i < ctx->slots; i++) {   // every slot, also after an error: nothing may still be writing to the caller's buffers
        hipError_t es = hipStreamSynchronize(ctx->pstream[i]);
        if (e == hipSuccess) e = es;
    }
    if (rc == LDPC_OK && e == hipSuccess && trace_lam)
        e = hipMemcpy(trace_lam, d_trace, (size_t)batch * turns * N * sizeof(double), hipMemcpyDeviceToHost);
    if (rc == LDPC_OK && e != hipSuccess) rc = set_error(LDPC_EHIP, "decode: %s", hipGetErrorString(e));
    (void)hipFree(d_trace);
    return rc;
}

int ldpc_decode_batch(ldpc_ctx *ctx, int max_iters, int batch, const float *llr, uint8_t *bits, int32_t *iters,
                      uint8_t *converged) {
    return decode_host(ctx, max_iters, batch, llr, ldpc::LLR_F32, bits, iters, converged, nullptr, nullptr);
}

int ldpc_decode_batch_f16(ldpc_ctx *ctx, int max_iters, int batch, const uint16_t *llr, uint8_t *bits, int32_t *iters,
                          uint8_t *converged) {
    return decode_host(ctx, max_iters, batch, llr, ldpc::LLR_F16, bits, iters, converged, nullptr, nullptr);
}

int ldpc_decode_batch_f64(ldpc_ctx *ctx, int max_iters, int batch, const double *llr, uint8_t *bits, int32_t *iters,
                          uint8_t *converged, double *final_lam) {
    return decode_host(ctx, max_iters, batch, llr, ldpc::LLR_F64, bits, iters, converged, final_lam, nullptr);
}

int ldpc_decode_trace(ldpc_ctx *ctx, int max_iters, int batch, const double *llr, uint8_t *bits, int32_t *iters,
                      uint8_t *converged, double *trace_lam) {
    if (!trace_lam) return set_error(LDPC_EINVAL, "null trace_lam");
    return decode_host(ctx, max_iters, batch, llr, ldpc::LLR_F64, bits, iters, converged, nullptr, trace_lam);
}

int ldpc_decode_one(ldpc_ctx *ctx, int max_iters, const double *llr, uint8_t *bits, int *iters, int *converged) {
    int32_t it = 0;
    uint8_t cv = 0;
    int rc = decode_host(ctx, max_iters, 1, llr, ldpc::LLR_F64, bits, &it, &cv, nullptr, nullptr);
    if (rc != LDPC_OK) return rc;
    if (iters) *iters = it;
    if (converged) *converged = cv;
    return LDPC_OK;
}

static int decode_dev_checked(ldpc_ctx *ctx, int max_iters, int batch, const void *d_llr, int fmt, uint8_t *d_bits,
                              int32_t *d_iters, uint8_t *d_converged, void *stream) {
    int rc = check_call(ctx, max_iters, batch);
    if (rc != LDPC_OK) return rc;
    if (batch == 0) return LDPC_OK;
    if (!d_llr || !d_bits) return set_error(LDPC_EINVAL, "null d_llr/d_bits");
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    return decode_dev(ctx, st, max_iters, batch, d_llr, fmt, d_bits, d_iters, d_converged, nullptr, nullptr);
}

int ldpc_decode_batch_dev(ldpc_ctx *ctx, int max_iters, int batch, const float *d_llr, uint8_t *d_bits,
                          int32_t *d_iters, uint8_t *d_converged, void *stream) {
    return decode_dev_checked(ctx, max_iters, batch, d_llr, ldpc::LLR_F32, d_bits, d_iters, d_converged, stream);
}

int ldpc_decode_batch_dev_f16(ldpc_ctx *ctx, int max_iters, int batch, const uint16_t *d_llr, uint8_t *d_bits,
                              int32_t *d_iters, uint8_t *d_converged, void *stream) {
    return decode_dev_checked(ctx, max_iters, batch, d_llr, ldpc::LLR_F16, d_bits, d_iters, d_converged, stream);
}

int ldpc_decode_batch_dev_packed(ldpc_ctx *ctx, int max_iters, int batch, const void *d_llr, int llr_f16, uint8_t *d_packed, int32_t *d_iters,
                                 uint8_t *d_converged, void *stream) {
    int rc = check_call(ctx, max_iters, batch);
    if (rc != LDPC_OK) return rc;
    if (batch == 0) return LDPC_OK;
    if (!d_llr || !d_packed) return set_error(LDPC_EINVAL, "null d_llr/d_packed");
    if (!ctx->d_unpacked) HIPCHK(hipMalloc((void **)&ctx->d_unpacked, (size_t)ctx->max_batch * ctx->code->N));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    rc = decode_dev(ctx, st, max_iters, batch, d_llr, llr_f16 ? ldpc::LLR_F16 : ldpc::LLR_F32, ctx->d_unpacked, d_iters, d_converged, nullptr, nullptr);
    if (rc != LDPC_OK) return rc;
    return ldpc::pack_bits(st, ctx->d_unpacked, d_packed, batch, ctx->code->N);
}

int ldpc_decode_batch_packed(ldpc_ctx *ctx, int max_iters, int batch, const void *llr, int llr_f16, uint8_t *packed, int32_t *iters, uint8_t *converged) {
    int rc = check_call(ctx, max_iters, batch);
    if (rc != LDPC_OK) return rc;
    if (batch == 0) return LDPC_OK;
    if (!llr || !packed) return set_error(LDPC_EINVAL, "null llr/packed");
    const size_t N = (size_t)ctx->code->N, PB = (N + 7) / 8, es = llr_f16 ? 2 : 4;
    if ((rc = ensure_staging(ctx, false)) != LDPC_OK) return rc;       // (d_in / d_iters / d_conv of the chunked pipeline; its d_bits slots are not used here)
    if (!ctx->d_packed) HIPCHK(hipMalloc((void **)&ctx->d_packed, (size_t)ctx->max_batch * PB));
    hipError_t e = hipSuccess;
    int slot = 0;
    // LLRs in chunk by chunk as the byte-per-bit entry point does; an eighth of the bytes back
    for (int f0 = 0; f0 < batch && rc == LDPC_OK && e == hipSuccess; f0 += ctx->chunk, slot = (slot + 1) % ctx->slots) {
        const int nb = std::min(ctx->chunk, batch - f0);
        hipStream_t st = ctx->pstream[slot];
        e = hipMemcpyAsync(ctx->d_in[slot], (const char *)llr + (size_t)f0 * N * es, (size_t)nb * N * es, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) break;
        rc = decode_dev(ctx, st, max_iters, nb, ctx->d_in[slot], llr_f16 ? ldpc::LLR_F16 : ldpc::LLR_F32, ctx->d_bits[slot], ctx->d_iters[slot], ctx->d_conv[slot], nullptr, nullptr);
        if (rc == LDPC_OK) rc = ldpc::pack_bits(st, ctx->d_bits[slot], ctx->d_packed + (size_t)f0 * PB, nb, (int)N);
        if (rc != LDPC_OK) break;
        e = hipMemcpyAsync(packed + (size_t)f0 * PB, ctx->d_packed + (size_t)f0 * PB, (size_t)nb * PB, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && iters) e = hipMemcpyAsync(iters + f0, ctx->d_iters[slot], sizeof(int32_t) * (size_t)nb, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && converged) e = hipMemcpyAsync(converged + f0, ctx->d_conv[slot], (size_t)nb, hipMemcpyDeviceToHost, st);
    }
    for (int i = 0; i < ctx->slots; i++) {
        hipError_t es2 = hipStreamSynchronize(ctx->pstream[i]);
        if (e == hipSuccess) e = es2;
    }
    if (rc == LDPC_OK && e != hipSuccess) rc = set_error(LDPC_EHIP, "decode (packed): %s", hipGetErrorString(e));
    return rc;
}

int ldpc_debug_step(ldpc_ctx *ctx, int batch, const double *orig, const double *lam, const double *ne, double *ne_out,
                    double *lam_out, uint8_t *syndrome_zero) {
    int rc = check_call(ctx, 0, batch);
    if (rc != LDPC_OK) return rc;
    if (batch == 0) return LDPC_OK;
    if (!orig || !lam || !ne || !ne_out || !lam_out) return set_error(LDPC_EINVAL, "null argument");
    const size_t N = (size_t)ctx->code->N, E = (size_t)ctx->code->E, B = (size_t)batch;
    double *buf = nullptr;
    uint8_t *d_syn = nullptr;
    const size_t total = B * (3 * N + 2 * E);
    HIPCHK(hipMalloc((void **)&buf, total * sizeof(double)));
    if (hipMalloc((void **)&d_syn, B) != hipSuccess) { hipFree(buf); return set_error(LDPC_ENOMEM, "hipMalloc"); }
    double *d_orig = buf, *d_lam = buf + B * N, *d_lam_out = buf + 2 * B * N, *d_ne = buf + 3 * B * N, *d_ne_out = d_ne + B * E;
    hipStream_t st = ctx->stream;
    hipError_t ce = hipMemcpyAsync(d_orig, orig, B * N * 8, hipMemcpyHostToDevice, st);
    if (ce == hipSuccess) ce = hipMemcpyAsync(d_lam, lam, B * N * 8, hipMemcpyHostToDevice, st);
    if (ce == hipSuccess) ce = hipMemcpyAsync(d_ne, ne, B * E * 8, hipMemcpyHostToDevice, st);
    if (ce != hipSuccess) rc = set_error(LDPC_EHIP, "debug_step upload: %s", hipGetErrorString(ce));
    else if (ctx->path == LDPC_PATH_FUSED)
        rc = ldpc::fused_step(*ctx->fused, st, batch, d_orig, d_lam, d_ne, d_ne_out, d_lam_out, d_syn);
    else if (ctx->lqc)
        rc = ldpc::layered_qc_step(*ctx->lqc, st, batch, d_orig, d_lam, d_ne, d_ne_out, d_lam_out, d_syn);
    else
        rc = ldpc::flood_step(ctx->flood, st, batch, d_orig, d_lam, d_ne, d_ne_out, d_lam_out, d_syn);
    if (rc == LDPC_OK) {
        ce = hipMemcpyAsync(ne_out, d_ne_out, B * E * 8, hipMemcpyDeviceToHost, st);
        if (ce == hipSuccess) ce = hipMemcpyAsync(lam_out, d_lam_out, B * N * 8, hipMemcpyDeviceToHost, st);
        if (ce == hipSuccess && syndrome_zero) ce = hipMemcpyAsync(syndrome_zero, d_syn, B, hipMemcpyDeviceToHost, st);
        if (ce != hipSuccess) rc = set_error(LDPC_EHIP, "debug_step download: %s", hipGetErrorString(ce));
    }
    hipError_t e = hipStreamSynchronize(st);
    if (rc == LDPC_OK && e != hipSuccess) rc = set_error(LDPC_EHIP, "debug_step: %s", hipGetErrorString(e));
    hipFree(buf);
    hipFree(d_syn);
    return rc;
}


// ------------------------------------------------------------------------------- timing
int ldpc_ctx_set_timing(ldpc_ctx *ctx, int enabled) {
    if (!ctx) return set_error(LDPC_EINVAL, "null ctx");
    ctx->timer.enabled = enabled != 0;
    ctx->timer.used = 0;
    return LDPC_OK;
}

int ldpc_ctx_kernel_time(ldpc_ctx *ctx, int *launches, double *total_ms) {
    if (!ctx) return set_error(LDPC_EINVAL, "null ctx");
    if (ctx->timer.drain(launches, total_ms) != 0) return set_error(LDPC_EHIP, "event query failed");
    return LDPC_OK;
}

const char *ldpc_ctx_kernel_name(const ldpc_ctx *ctx) {
    if (!ctx) return "";
    if (ctx->path == LDPC_PATH_FUSED && ctx->fused) return ldpc::fused_kernel_name(*ctx->fused);
    if (ctx->lqc) return ldpc::layered_qc_launch_info(*ctx->lqc).name;
    return ctx->schedule == LDPC_SCHED_LAYERED ? "layered_kernel" : "flood_cn_kernel";
}

int ldpc_ctx_kernel_geometry(const ldpc_ctx *ctx, int *threads_per_workgroup, int *frames_per_workgroup) {
    if (!ctx) return set_error(LDPC_EINVAL, "null ctx");
    int t = 0, f = 0;
    if (ctx->path == LDPC_PATH_FUSED && ctx->fused) { const ldpc::LaunchInfo &li = ldpc::fused_launch_info(*ctx->fused); t = li.threads; f = li.frames_per_wg; }
    if (ctx->lqc) { const ldpc::LaunchInfo &li = ldpc::layered_qc_launch_info(*ctx->lqc); t = li.threads; f = li.frames_per_wg; }
    if (threads_per_workgroup) *threads_per_workgroup = t;
    if (frames_per_workgroup) *frames_per_workgroup = f;
    return LDPC_OK;
}

// ------------------------------------------------------------------------------- run-time specialised kernels
const char *ldpc_jit_cache_dir(void) { return ldpc::jit_cache_dir(); }

static int jit_kind_of(int dtype, int schedule) {
    if (schedule == LDPC_SCHED_LAYERED) return dtype == LDPC_F16PK ? ldpc::JIT_LAYERED_PK16 : ldpc::JIT_LAYERED;
    return dtype == LDPC_F16PK ? ldpc::JIT_PK16 : ldpc::JIT_SPLIT;
}

long ldpc_jit_source_for(const ldpc_code *code, int variant, int dtype, int schedule, char *buf, size_t cap) {
    if (!code) return set_error(LDPC_EINVAL, "null code");
    if (schedule != LDPC_SCHED_FLOODING && schedule != LDPC_SCHED_LAYERED) return set_error(LDPC_EINVAL, "unknown schedule %d", schedule);
    try {
        const int kind = jit_kind_of(dtype, schedule);
        const char *why = ldpc::jit_split_why_not(*code, variant, dtype, kind);
        if (why) return set_error(LDPC_EUNSUPPORTED, "%s", why);
        const std::string src = ldpc::jit_split_source(*code, variant, dtype, nullptr, kind);
        if (buf && cap) { size_t n = std::min(cap - 1, src.size()); memcpy(buf, src.data(), n); buf[n] = 0; }
        return (long)src.size();
    } catch (...) { return set_error(LDPC_ENOMEM, "out of host memory"); }
}

int ldpc_jit_prepare_for(const ldpc_code *code, int variant, int dtype, int schedule, char *kernel_name, size_t cap, int *from_cache, double *seconds) {
    if (!code) return set_error(LDPC_EINVAL, "null code");
    if (schedule != LDPC_SCHED_FLOODING && schedule != LDPC_SCHED_LAYERED) return set_error(LDPC_EINVAL, "unknown schedule %d", schedule);
    try {
        const int kind = jit_kind_of(dtype, schedule);
        const char *why = ldpc::jit_split_why_not(*code, variant, dtype, kind);
        if (why) return set_error(LDPC_EUNSUPPORTED, "%s", why);
        ldpc::JitKernel g;
        const std::string src = ldpc::jit_split_source(*code, variant, dtype, &g, kind);
        std::vector<char> co;
        bool fc = false;
        double sec = 0;
        int rc = ldpc::jit_compile_cached(src, g.name, co, &fc, &sec);
        if (rc != LDPC_OK) return rc;
        if (kernel_name && cap) snprintf(kernel_name, cap, "%s", g.name.c_str());
        if (from_cache) *from_cache = fc ? 1 : 0;
        if (seconds) *seconds = sec;
        return LDPC_OK;
    } catch (...) { return set_error(LDPC_ENOMEM, "out of host memory"); }
}

long ldpc_jit_source(const ldpc_code *code, int variant, int dtype, char *buf, size_t cap) {
    return ldpc_jit_source_for(code, variant, dtype, LDPC_SCHED_FLOODING, buf, cap);
}
int ldpc_jit_prepare(const ldpc_code *code, int variant, int dtype, char *kernel_name, size_t cap, int *from_cache, double *seconds) {
    return ldpc_jit_prepare_for(code, variant, dtype, LDPC_SCHED_FLOODING, kernel_name, cap, from_cache, seconds);
}

// ------------------------------------------------------------------------------- frame source
}  // extern "C"
struct ldpc_sim {
    ldpc::SimDev dev{};
    int p = 0, max_batch = 0, device = 0;
    std::vector<uint32_t> gt_host;     // dense generator, packed columns (host encode)
    std::vector<uint32_t> qc_host;     // quasi-cyclic generator: [brows][bcols][W] first-row words
    uint32_t *d_gt = nullptr, *d_msgw = nullptr, *d_rot = nullptr, *d_parw = nullptr;
};
extern "C" {

void ldpc_sim_destroy(ldpc_sim *sim) {
    if (!sim) return;
    (void)hipSetDevice(sim->device);
    hipFree(sim->d_gt);
    hipFree(sim->d_rot);
    hipFree(sim->d_parw);
    hipFree(sim->d_msgw);
    delete sim;
}

ldpc_sim *ldpc_sim_create(const ldpc_code *code, int k, int n_tx, int p, const uint8_t *G, int max_batch) {
    const int device = current_device();
    if (device < 0) { set_error(LDPC_ENODEVICE, "ldpc_init() has not succeeded"); return nullptr; }
    return ldpc_sim_create_on(code, device, k, n_tx, p, G, max_batch);
}

static ldpc_sim *sim_new(const ldpc_code *code, int device, int k, int n_tx, int max_batch) {
    if (check_device(device) != LDPC_OK) return nullptr;
    ldpc_sim *s = new (std::nothrow) ldpc_sim();
    if (!s) { set_error(LDPC_ENOMEM, "out of host memory"); return nullptr; }
    s->device = device; s->max_batch = max_batch;
    s->dev.N = code->N; s->dev.k = k; s->dev.n_tx = n_tx; s->dev.kwords = (k + 31) / 32; s->dev.gt = nullptr; s->dev.qc_rot = nullptr;
    return s;
}

ldpc_sim *ldpc_sim_create_on(const ldpc_code *code, int device, int k, int n_tx, int p, const uint8_t *G, int max_batch) {
    if (!code || k <= 0 || n_tx < k || n_tx > code->N || max_batch <= 0 || (G && (p <= 0 || k + p < n_tx))) {
        set_error(LDPC_EINVAL, "ldpc_sim_create: bad arguments (k=%d n_tx=%d p=%d N=%d)", k, n_tx, p, code ? code->N : -1);
        return nullptr;
    }
    ldpc_sim *s = sim_new(code, device, k, n_tx, max_batch);
    if (!s) return nullptr;
    s->p = G ? p : 0;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess && G) {
        s->gt_host.assign((size_t)p * s->dev.kwords, 0u);
        for (int r = 0; r < k; r++)
            for (int j = 0; j < p; j++)
                if (G[(size_t)r * p + j]) s->gt_host[(size_t)j * s->dev.kwords + (r >> 5)] |= 1u << (r & 31);
        // device copy transposed and padded: [kwords][pp], parity position fastest, so that the lanes of a wave
        // (consecutive parity positions, four per lane) read consecutive words
        const int pp = (p + 3) / 4 * 4;
        std::vector<uint32_t> gtt((size_t)s->dev.kwords * pp, 0u);
        for (int j = 0; j < p; j++)
            for (int w = 0; w < s->dev.kwords; w++) gtt[(size_t)w * pp + j] = s->gt_host[(size_t)j * s->dev.kwords + w];
        e = hipMalloc((void **)&s->d_gt, gtt.size() * 4);
        if (e == hipSuccess) e = hipMemcpy(s->d_gt, gtt.data(), gtt.size() * 4, hipMemcpyHostToDevice);
        s->dev.gt = s->d_gt;
        s->dev.pp = pp;
    }
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_msgw, (size_t)max_batch * s->dev.kwords * 4);
    if (e != hipSuccess) { set_error(LDPC_EHIP, "ldpc_sim_create: %s", hipGetErrorString(e)); ldpc_sim_destroy(s); return nullptr; }
    return s;
}

// The generator in the reference's quasi-cyclic form (Fast/Encoder.hs:26-40: sz in {32, 64, 128, 256}, one machine word
// of sz bits per circulant = the integer of the .q file, bit b = first-row entry of column b).
ldpc_sim *ldpc_sim_create_qc_on(const ldpc_code *code, int device, int k, int n_tx, int sz, int block_rows, int block_cols,
                                const uint32_t *circ, int max_batch) {
    if (!code || !circ || sz <= 0 || block_rows <= 0 || block_cols <= 0 || max_batch <= 0 || k != sz * block_rows || n_tx < k || n_tx > code->N ||
        (long)k + (long)sz * block_cols < n_tx) {
        set_error(LDPC_EINVAL, "ldpc_sim_create_qc: bad arguments (k=%d n_tx=%d sz=%d blocks %dx%d N=%d)", k, n_tx, sz, block_rows, block_cols, code ? code->N : -1);
        return nullptr;
    }
    if (sz != 32 && sz != 64 && sz != 128 && sz != 256) {   // Fast/Encoder.hs:33 "unsupported size for fast encoder"
        set_error(LDPC_EUNSUPPORTED, "unsupported size for fast encoder : %d (32, 64, 128, 256; give the dense generator to ldpc_sim_create instead)", sz);
        return nullptr;
    }
    ldpc_sim *s = sim_new(code, device, k, n_tx, max_batch);
    if (!s) return nullptr;
    const int W = sz / 32, CB = 16 / W, ncg = (block_cols + CB - 1) / CB;
    s->p = sz * block_cols;
    s->qc_host.assign(circ, circ + (size_t)block_rows * block_cols * W);
    s->dev.qc_w = W; s->dev.qc_brows = block_rows; s->dev.qc_bcols = block_cols; s->dev.qc_ncg = ncg; s->dev.pwords = block_cols * W;
    // rot[cg][r][b][c in group][w] = word w of rotateL(g[r][c], b)
    std::vector<uint32_t> rot((size_t)ncg * block_rows * 32 * 16, 0u);
    for (int bc = 0; bc < block_cols; bc++)
        for (int r = 0; r < block_rows; r++) {
            const uint32_t *g = &circ[((size_t)r * block_cols + bc) * W];
            for (int b = 0; b < 32; b++)
                for (int w = 0; w < W; w++) {
                    const uint32_t lo = g[w], hi = g[(w + W - 1) % W];
                    rot[(((size_t)(bc / CB) * block_rows + r) * 32 + b) * 16 + (bc % CB) * W + w] = b ? ((lo << b) | (hi >> (32 - b))) : lo;
                }
        }
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_rot, rot.size() * 4);
    if (e == hipSuccess) e = hipMemcpy(s->d_rot, rot.data(), rot.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_parw, (size_t)max_batch * s->dev.pwords * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_msgw, (size_t)max_batch * s->dev.kwords * 4);
    s->dev.qc_rot = s->d_rot;
    if (e != hipSuccess) { set_error(LDPC_EHIP, "ldpc_sim_create_qc: %s", hipGetErrorString(e)); ldpc_sim_destroy(s); return nullptr; }
    return s;
}

int ldpc_sim_encoder(const ldpc_sim *sim) {
    if (!sim) return set_error(LDPC_EINVAL, "null sim");
    return sim->dev.qc_rot ? LDPC_ENCODER_QC : (sim->dev.gt ? LDPC_ENCODER_DENSE : LDPC_ENCODER_NONE);
}

static int sim_generate_any(ldpc_sim *sim, uint64_t seed, uint64_t first_frame, int batch, double ebn0_db, void *d_out,
                            int out_fmt, uint8_t *d_msg, void *stream) {
    if (!sim || !d_out || batch < 0 || batch > sim->max_batch) return set_error(LDPC_EINVAL, "ldpc_sim_generate: bad arguments");
    if (batch == 0) return LDPC_OK;
    HIPCHK(hipSetDevice(sim->device));
    return ldpc::sim_generate(sim->dev, sim->d_msgw, sim->d_parw, (hipStream_t)stream, seed, first_frame, batch, ebn0_db, d_out, out_fmt, d_msg);
}

int ldpc_sim_generate(ldpc_sim *sim, uint64_t seed, uint64_t first_frame, int batch, double ebn0_db, float *d_llr,
                      uint8_t *d_msg, void *stream) {
    return sim_generate_any(sim, seed, first_frame, batch, ebn0_db, d_llr, 0, d_msg, stream);
}

int ldpc_sim_generate_f16(ldpc_sim *sim, uint64_t seed, uint64_t first_frame, int batch, double ebn0_db, uint16_t *d_llr,
                          uint8_t *d_msg, void *stream) {
    return sim_generate_any(sim, seed, first_frame, batch, ebn0_db, d_llr, 1, d_msg, stream);
}

int ldpc_sim_encode_batch(ldpc_sim *sim, uint64_t seed, uint64_t first_frame, int batch, uint8_t *d_codewords, uint8_t *d_msg, void *stream) {
    return sim_generate_any(sim, seed, first_frame, batch, 0.0, d_codewords, 2, d_msg, stream);
}

int ldpc_sim_tally(ldpc_sim *sim, int batch, const uint8_t *d_bits, const int32_t *d_iters, uint64_t *d_tally, void *stream) {
    if (!sim || !d_bits || !d_tally || batch < 0 || batch > sim->max_batch) return set_error(LDPC_EINVAL, "ldpc_sim_tally: bad arguments");
    if (batch == 0) return LDPC_OK;
    HIPCHK(hipSetDevice(sim->device));
    return ldpc::sim_tally(sim->dev, sim->d_msgw, (hipStream_t)stream, batch, d_bits, d_iters, (unsigned long long *)d_tally);
}

int ldpc_sim_encode_host(const ldpc_sim *sim, const uint8_t *msg, uint8_t *parity) {
    if (!sim || !msg || !parity) return set_error(LDPC_EINVAL, "null argument");
    const int kw = sim->dev.kwords;
    std::vector<uint32_t> mw((size_t)kw, 0u);
    for (int r = 0; r < sim->dev.k; r++) if (msg[r]) mw[r >> 5] |= 1u << (r & 31);
    if (!sim->qc_host.empty()) {
        // Fast/Encoder.hs:42-63 word by word: res[col] = XOR_row mulWord(v'[row], g[row][col]), mulWord = rotate-and-xor
        // over the set bits of the message word
        const int W = sim->dev.qc_w, R = sim->dev.qc_brows, Cc = sim->dev.qc_bcols, sz = 32 * W;
        std::vector<uint32_t> res((size_t)W), rotd((size_t)W);
        for (int c = 0; c < Cc; c++) {
            std::fill(res.begin(), res.end(), 0u);
            for (int r = 0; r < R; r++) {
                const uint32_t *g = &sim->qc_host[((size_t)r * Cc + c) * W];
                for (int n = 0; n < sz; n++) {
                    if (!((mw[(size_t)r * W + (n >> 5)] >> (n & 31)) & 1u)) continue;
                    const int a = n >> 5, b = n & 31;
                    for (int w = 0; w < W; w++) {
                        const uint32_t lo = g[(w - a + W) % W], hi = g[(w - a - 1 + 2 * W) % W];
                        res[w] ^= b ? ((lo << b) | (hi >> (32 - b))) : lo;
                    }
                }
            }
            for (int j = 0; j < sz; j++) parity[(size_t)c * sz + j] = (uint8_t)((res[j >> 5] >> (j & 31)) & 1u);
        }
        return LDPC_OK;
    }
    for (int j = 0; j < sim->p; j++) {
        uint32_t acc = 0;
        for (int w = 0; w < kw; w++) acc ^= mw[w] & sim->gt_host[(size_t)j * kw + w];
        parity[j] = (uint8_t)(__builtin_popcount(acc) & 1);
    }
    return LDPC_OK;
}

}  // extern "C"
