/*
 * ldpc_hip.h -- C ABI of libldpc_hip.so: the MI355X (gfx950) LDPC belief-propagation decoder
 * that replaces the decoder slot of ku-fpg/ecc-ldpc's `Code`/`ECC` record.
 *
 * Every entry point names the reference interface it replaces (paths relative to the
 * reference repository).  The Haskell side keeps its contract: a plug-in is a `Code` built by
 *   mkLDPC_CodeIO name maxThreadCount encoder decoder initialize finalize
 *                                                   (src/ECC/Code/LDPC/Utils.hs:91-108)
 * and everything the CUDA plug-ins do behind that record -- loadFile ptx, getFun, mallocArray,
 * launchKernel, peekListArray (src/ECC/Code/LDPC/GPU/CUDA/Arraylet2.hs:88-297) -- lives behind
 * this header instead.  INTEGRATION.md shows the `foreign import ccall` binding.
 *
 * Conventions
 *   - plain C types only; no C++ exception crosses the ABI.
 *   - every `int` function returns LDPC_OK (0) or a negative LDPC_E* code; the message for the
 *     last failure on the calling thread is ldpc_last_error().
 *   - pointer-returning functions return NULL on failure (message in ldpc_last_error()).
 *   - LLR sign convention of the reference: LLR > 0 <=> bit 1 (`hard x = x > 0`,
 *     src/ECC/Code/LDPC/GPU/Reference.hs:59-60, cudabits/common.h:180-182).
 *   - decoder semantics are exactly the loop of src/ECC/Code/LDPC/Reference/Orig.hs:67-71:
 *     (1) syndrome of hard(lam) zero -> return lam; (2) n >= max_iters -> return the CHANNEL
 *     LLRs (orig_lam); (3) otherwise update.  So bits = hard(lam_n) for a frame that converged
 *     after n updates and hard(channel LLR) for one that did not.
 *   - a context is NOT thread-safe: one ldpc_ctx per calling thread (the reference's CUDA
 *     plug-ins keep mutable device buffers in the closure the same way, Arraylet2.hs:61,118-144).
 */
#ifndef LDPC_HIP_H
#define LDPC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDPC_OK 0
#define LDPC_EINVAL (-1)       /* bad argument */
#define LDPC_ENOMEM (-2)       /* host or device allocation failed */
#define LDPC_EHIP (-3)         /* a HIP runtime call failed (message has the HIP error string) */
#define LDPC_ENODEVICE (-4)    /* no usable gfx950 device / ldpc_init not called */
#define LDPC_EUNSUPPORTED (-5) /* valid request this build has no kernel for */
#define LDPC_EDEGREE (-6)      /* min-sum on a degree-1 check row (Haskell: foldr1 on [], Min.hs:79) */
#define LDPC_EFORMAT (-7)      /* matrix file does not parse */
#define LDPC_ENOTFOUND (-8)    /* matrix file / code name not found */

typedef struct ldpc_code ldpc_code; /* immutable parity-check graph, host + device tables */
typedef struct ldpc_ctx ldpc_ctx;   /* one decoder replica: device buffers + stream          */

/* check-node rule.  LDPC_TANH: Reference/Orig.hs:81-92 ; LDPC_MINSUM: Reference/Min.hs:75-87 ;
 * LDPC_TANH_CM: the tanh rule with the roundings of the reference's `arraylet-cm` decoder (Fast/CachedMult.hs:25-56,
 * 233-264: row product cached as a StableDiv, leave-one-out by division, column sum orig + foldr1 (+)) -- the same real
 * function as LDPC_TANH, ~1e-11 apart in double.  Parity mode only: LDPC_F64, flooding schedule, flood path. */
/* LDPC_TANH_CUDA32 (r04): the arithmetic of the reference's LIVE GPU decoder `cuda-arraylet2` (GPU/CUDA/Arraylet2.hs:88-331 driving
 * cudabits/arraylet2.cu:43-83 selfProduct, common.h:82-88 atanh_, :151-178 updateLam; float_ty = float): float state, factors = the
 * double tanh stored as floats, the leave-one-out product in a double register, atanh_ on its FLOAT value (so the +-18.71 clamp
 * fires when the product rounds to +-1 in float, |x| > ~17, where Orig.hs's Double goes on to ~37), column sums
 * ((orig + ne_1) + ne_2) + ... in float over ascending rows.  In the saturation regime a different function from LDPC_TANH.
 * Parity mode only: LDPC_F32, flooding schedule, flood path, rows up to weight 32 (checker: oracle "cuda32"). */
typedef enum { LDPC_TANH = 0, LDPC_MINSUM = 1, LDPC_TANH_CM = 2, LDPC_TANH_CUDA32 = 3 } ldpc_variant;
/* arithmetic / storage type of LLRs and messages on the device.
 * F32: the cudabits kernels' `typedef float float_ty` (cudabits/common.h:1).
 * F64: parity mode, same type as the CPU reference (Double).
 * F16: fp16 storage of whatever the decoder keeps in HBM, fp32 arithmetic (BASELINE.json configs[3]):
 *      every LLR given to an F16 context counts as stored in fp16 (saturating round-to-nearest-even on load);
 *      the flood path also keeps lam and the messages in fp16 between its kernels, the fused paths keep them
 *      on-chip in f32 -- so for F16 the two paths are two different (documented) decoders, each with its own
 *      emulation in oracle/emulate_f16.py, whereas for F32/F64 they agree bit for bit.  With LDPC_SCHED_LAYERED from HBM
 *      (quasi-cyclic codes, min-sum): lam stored in fp16, f32 row records (emulation decode_minsum_f16_layered). */
/* F16PK (extension, BASELINE.json configs[3] "min-sum fp16 LLRs"): ARITHMETIC in IEEE binary16, two frames per lane in
 *      packed instructions (csrc/fused_pk16_body.h holds the specification: the loop of Min.hs:54-104 on fp16 values, the 3/4
 *      applied inside fused multiply-adds; channel LLRs saturate at +-16384 and message magnitudes at 2048, so that no sum
 *      leaves the fp16 range).  Min-sum, on-chip path, either schedule (layered: csrc/fused_layered_body.h), single-circulant
 *      quasi-cyclic codes whose frame fits in LDS, column degree <= 30: built-in instances for the shipped AR4JA matrices,
 *      run-time specialised ones for any other (ldpc_jit_prepare_for); anything else: LDPC_EUNSUPPORTED.  Its checker is the
 *      bit-exact emulation oracle/emulate_f16.py decode_minsum_pk16 / decode_minsum_pk16_layered; BER next to the F32
 *      decoder: DESIGN.md, profiles/r03_ber_pk16_vs_f32.txt. */
typedef enum { LDPC_F32 = 0, LDPC_F64 = 1, LDPC_F16 = 2, LDPC_F16PK = 3 } ldpc_dtype;
/* message-passing schedule.  FLOODING: the reference's (Orig.hs:81-98: all checks, then all variables).
 * LAYERED (extension, no reference counterpart): checks layer by layer, each seeing the LLRs the layers before it
 * updated in the same sweep -- about half the sweeps to converge; stopping rule: before the first sweep the
 * syndrome of the channel decisions, after a sweep "no check it saw was odd and no hard decision changed"
 * (specification: oracle/ldpc_oracle.c oracle_decode_layered); a frame out of sweeps returns the channel decisions,
 * as Orig.hs:70 does.  `iters` counts sweeps. */
typedef enum { LDPC_SCHED_FLOODING = 0, LDPC_SCHED_LAYERED = 1 } ldpc_schedule;
/* order in which a column's messages are added to the channel LLR.  The registered decoders of the reference compute the same
 * check rule but add a column up differently -- the last ulps of a Double, nothing else:
 *   REFERENCE  foldr (+) orig: ne_1 + (ne_2 + (... + (ne_k + orig)))          Reference/Orig.hs:96, Min.hs:101 (`reference`, `min`)
 *   ARRAYLET   orig + (ne_1 + (ne_2 + (... + ne_k)))                          Fast/Arraylet.hs:105-109,185-186, ArrayletMin.hs:191-192
 *                                                                             (`arraylet`, `arraylet-min`; LDPC_TANH_CM implies it)
 *   SPARSE     orig + (((0 + ne_1) + ne_2) + ... + ne_k)                      Reference/Sparse.hs:112-114, SparseMin.hs:117-119,
 *                                                                             Data/Sparse/Matrix.hs:35-36 (`sparse`, `sparsemin`)
 * rows ascending.  ARRAYLET and SPARSE are PARITY MODES (bit-exact f64 against oracle "arraylet" / "sparse"...): flooding
 * schedule, LDPC_PATH_FLOOD batch-major kernels; the on-chip kernels add in REFERENCE order. */
typedef enum { LDPC_SUM_REFERENCE = 0, LDPC_SUM_ARRAYLET = 1, LDPC_SUM_SPARSE = 2 } ldpc_sum_order;
/* which kernel family a context uses */
typedef enum {
    LDPC_PATH_AUTO = 0,  /* on-chip kernel when the code/variant/dtype has one, else the HBM path */
    LDPC_PATH_FLOOD = 1, /* state in HBM, any H: quasi-cyclic codes one workgroup per frame and ONE launch per batch (either
                            schedule), any other H batch-major with two kernels per iteration                              */
    LDPC_PATH_FUSED = 2  /* whole decode in one launch, state in LDS/registers: QC codes (built-in or run-time specialised
                            instances) and any H whose frame fits in 160 KB of LDS; flooding schedule                      */
} ldpc_path;

/* ---- library life-cycle -------------------------------------------------------------------
 * replaces  initialize :: IO CudaAllocations  (GPU/CUDA/Arraylet2.hs:287-293: dummy malloc to
 * create the context + loadFile "cudabits/arraylet2.ptx") and finalize (:295-297).          */
/* ldpc_init: idempotent; checks that `device` is a gfx950 GPU and makes it the CALLING THREAD's device: objects the
 * thread creates afterwards live there.  The first device any thread initialised is the default of threads that
 * never called ldpc_init.  One process can drive several GPUs: one thread per GPU each calling ldpc_init(its device),
 * or one thread using the *_on constructors below (the reference makes maxThreadCount replicas in one process and
 * picks one by thread, Utils.hs:53,63-69). */
int ldpc_init(int device);
int ldpc_shutdown(void);
int ldpc_current_device(void);      /* the calling thread's device (see ldpc_init), -1 if none */
const char *ldpc_last_error(void);  /* thread-local, never NULL */
int ldpc_last_error_code(void);     /* the LDPC_E* code that message belongs to (0 if none yet) */
int ldpc_abi_version(void);         /* bumps when a signature in this header changes */
int ldpc_device_count(void);        /* HIP devices visible; 0 when there is no GPU (no error) */

/* ---- graph ---------------------------------------------------------------------------------
 * replaces initMatrixlet (GPU/CUDA/Arraylet2.hs:299-331): the quasi-cyclic H as circulant size
 * + a block_rows x block_cols table of rotations, -1 = empty block.  Rotation `off` at block
 * (br,bc) means row r of the block has its 1 at column (r+off) mod sz
 * (src/Data/Matrix/QuasiCyclic.hs:19-25, Fast/Arraylet.hs:34-43).  The library copies `offsets`. */
ldpc_code *ldpc_code_create_qc(int sz, int block_rows, int block_cols, const int32_t *offsets);
/* any H as CSR (what `Matrix Bool` decoders take, Reference/Orig.hs:30): row_ptr[M+1],
 * col_idx[E] strictly ascending inside each row.  The library copies both arrays. */
ldpc_code *ldpc_code_create_csr(int M, int N, const int32_t *row_ptr, const int32_t *col_idx);
void ldpc_code_destroy(ldpc_code *code);
/* M = checks, N = unpunctured code length (cols H), E = edges */
int ldpc_code_dims(const ldpc_code *code, int *M, int *N, int *E);
/* the CSR edge order the library uses for every per-edge array it exposes (debug entry points):
 * row-major, ascending column inside a row == the order of Orig.hs:86-91.  Arrays caller-owned. */
int ldpc_code_csr(const ldpc_code *code, int32_t *row_ptr /*M+1*/, int32_t *col_idx /*E*/);

/* ---- layers (row-layered schedule, an EXTENSION: BASELINE.json configs[4]; the reference has flooding only) -------
 * A layer is a range of consecutive rows that share no column.  Default: the block rows of a quasi-cyclic code (one
 * circulant per block => column-disjoint), every row its own layer for a CSR code.  ldpc_code_set_layers replaces
 * the partition (layer_ptr[0] = 0 < ... < layer_ptr[n_layers] = M; LDPC_EINVAL if two rows of a layer share a column
 * or a context already exists on the code). */
int ldpc_code_set_layers(ldpc_code *code, int n_layers, const int32_t *layer_ptr);
/* A row-layered decoder visits the block rows of a quasi-cyclic H in the order they are given; block rows that follow one another
 * and share no block column can be worked on together with unchanged results, and the long-code kernel (csrc/layered_lds.hip) does so
 * for runs of up to four.  The order of the rows of H is the caller's to choose (a permutation of H's rows is the same code; the
 * SCHEDULE, hence the last digits of a BER, changes with it): this helper proposes one -- perm[i] = the block row to put at place i,
 * runs of up to `run` (1..8) pairwise column-disjoint block rows, full runs first (greedy, deterministic; what
 * tools/gen_dvbs2_like.py applies to codes/dvbs2like.64800.1.2).  Pure host code, needs no GPU.  Returns the number of full runs,
 * < 0 on error.  No counterpart in the reference (flooding only). */
int ldpc_qc_layer_order(int block_rows, int block_cols, const int32_t *offsets, int run, int32_t *perm /* block_rows */);
int ldpc_code_layers(const ldpc_code *code, int *n_layers, int32_t *layer_ptr /* may be NULL; n_layers+1 entries */);

/* ---- decoder replica -------------------------------------------------------------------------
 * replaces one `decoder vars h` call (Utils.hs:53 replicateM maxThreadCount; Arraylet2.hs:88-144:
 * getFun x9, mallocArray of mLet/newMLet/lam/orig_lam/done, Stream.create).  Owns its device
 * buffers and one HIP stream; sized for frames <= max_batch per call. */
ldpc_ctx *ldpc_ctx_create(const ldpc_code *code, int variant, int dtype, int max_batch);
ldpc_ctx *ldpc_ctx_create_ex(const ldpc_code *code, int variant, int dtype, int max_batch, int path);
/* the same on an explicitly named device (no ldpc_init needed on this thread; a code may have replicas on several
 * devices, its graph tables are uploaded once per device) */
ldpc_ctx *ldpc_ctx_create_on(const ldpc_code *code, int device, int variant, int dtype, int max_batch, int path);
/* everything above, plus what was added later, in one extensible structure: set struct_size = sizeof(ldpc_ctx_config)
 * and zero the rest before filling in; device = -1 means the calling thread's device */
typedef struct {
    size_t struct_size;
    int device, variant, dtype, max_batch, path, schedule;
    int sum_order;   /* ldpc_sum_order; read when struct_size covers it, else LDPC_SUM_REFERENCE */
} ldpc_ctx_config;
ldpc_ctx *ldpc_ctx_create_cfg(const ldpc_code *code, const ldpc_ctx_config *cfg);
int ldpc_ctx_schedule(const ldpc_ctx *ctx);
const ldpc_code *ldpc_ctx_code(const ldpc_ctx *ctx);
int ldpc_ctx_max_batch(const ldpc_ctx *ctx);
int ldpc_ctx_device(const ldpc_ctx *ctx);
void ldpc_ctx_destroy(ldpc_ctx *ctx);
/* LDPC_PATH_FLOOD or LDPC_PATH_FUSED: what the context resolved to */
int ldpc_ctx_path(const ldpc_ctx *ctx);

/* ---- decode ------------------------------------------------------------------------------------
 * ldpc_decode_one replaces the per-frame closure
 *   Rate -> Int -> U.Vector Double -> IO (Maybe (U.Vector Bool))     (Arraylet2.hs:151-273)
 * llr: N doubles, already un-punctured (zeros appended, Utils.hs:55,69).  bits: N bytes 0/1.
 * A non-zero return is the closure's `Nothing` (Utils.hs:70-71). Blocking. */
int ldpc_decode_one(ldpc_ctx *ctx, int max_iters, const double *llr, uint8_t *bits, int *iters,
                    int *converged);
/* throughput path, host buffers: llr [batch][N] float32 frame-major, bits [batch][N] bytes,
 * iters [batch] (updates run, = max_iters for a non-converged frame), converged [batch].
 * iters / converged may be NULL.  Blocking. */
int ldpc_decode_batch(ldpc_ctx *ctx, int max_iters, int batch, const float *llr, uint8_t *bits,
                      int32_t *iters, uint8_t *converged);
/* same with float64 LLRs in and (optionally) the returned LLR vector out -- `lam` of the frame
 * that converged, the channel LLRs otherwise (what Orig.hs:69-70 returns before `map hard`).
 * final_lam may be NULL. */
int ldpc_decode_batch_f64(ldpc_ctx *ctx, int max_iters, int batch, const double *llr, uint8_t *bits,
                          int32_t *iters, uint8_t *converged, double *final_lam);
/* zero-copy path: every pointer is DEVICE memory on the context's device.  d_llr [batch][N]
 * float32; d_bits [batch][N] bytes; d_iters, d_converged may be NULL.  Work is enqueued on
 * `stream` (a hipStream_t) and NOT synchronised.  NULL = the context's OWN stream, which is
 * non-blocking: it does not order against the default (null) stream, so a caller that produces
 * d_llr on another stream must pass that stream here (or synchronise first). */
int ldpc_decode_batch_dev(ldpc_ctx *ctx, int max_iters, int batch, const float *d_llr,
                          uint8_t *d_bits, int32_t *d_iters, uint8_t *d_converged, void *stream);
/* fp16 channel LLRs (IEEE binary16 bit patterns; BASELINE.json configs[3] "min-sum fp16 LLRs"): the same two
 * throughput entry points with half the bytes per frame over PCIe / out of HBM.  Accepted by a context of any
 * dtype (the values convert exactly to f32/f64).  No counterpart in the reference, whose CUDA path pokes
 * float32 (GPU/CUDA/Arraylet2.hs:151-160). */
int ldpc_decode_batch_f16(ldpc_ctx *ctx, int max_iters, int batch, const uint16_t *llr, uint8_t *bits,
                          int32_t *iters, uint8_t *converged);
int ldpc_decode_batch_dev_f16(ldpc_ctx *ctx, int max_iters, int batch, const uint16_t *d_llr,
                              uint8_t *d_bits, int32_t *d_iters, uint8_t *d_converged, void *stream);
/* PACKED result bits (r04): ceil(N/8) bytes per frame instead of N -- bit i of a frame is bit (i % 8) of byte i / 8 (LSB first; what
 * SURVEY.md section 8d's byte model counts as a frame's output) -- an eighth of the bytes back over PCIe or out to the caller's HBM
 * buffer; the decoded values are those of the unpacked entry points.  llr_f16 != 0: d_llr / llr are IEEE binary16 patterns.  The decode
 * kernels write one byte per bit into the context's own staging buffer (max_batch x N bytes, allocated on first use) and a second
 * kernel packs them.  The reference returns `Vector Bool` (Arraylet2.hs:271-273): one value per bit, no counterpart. */
int ldpc_decode_batch_dev_packed(ldpc_ctx *ctx, int max_iters, int batch, const void *d_llr, int llr_f16, uint8_t *d_packed,
                                 int32_t *d_iters, uint8_t *d_converged, void *stream);
int ldpc_decode_batch_packed(ldpc_ctx *ctx, int max_iters, int batch, const void *llr, int llr_f16, uint8_t *packed,
                             int32_t *iters, uint8_t *converged);
/* page-locked host memory for the host-pointer entry points.  With llr AND bits buffers from ldpc_host_alloc (or
 * registered with hipHostRegister) ldpc_decode_batch runs zero-copy: the decode kernel itself reads the LLRs and
 * writes the bits over PCIe (10-11 Gbit/s PCIe-inclusive on jpl.4096); pageable buffers go through a chunked
 * H2D / decode / D2H pipeline at the pageable-copy rate (2-3.5 Gbit/s); up to 16 frames (ldpc_decode_one) take a
 * latency path through internal page-locked bounce buffers.  (The reference pokes pageable Storable vectors,
 * GPU/CUDA/Arraylet2.hs:158-159.)  NULL on failure. */
void *ldpc_host_alloc(size_t bytes);
void ldpc_host_free(void *p);
/* wait for everything enqueued on the context's stream */
int ldpc_ctx_synchronize(ldpc_ctx *ctx);

/* ---- coalescing of per-frame calls ----------------------------------------------------------------------
 * The reference's harness calls the decoder once per frame from up to maxThreadCount threads (Utils.hs:53,63-69) and
 * its CUDA plug-ins decode one codeword per launch sequence (Arraylet2.hs:151-273).  A batcher sits between such
 * callers and ONE decoder replica: the first caller to arrive waits until max_frames requests are queued or
 * max_wait_us have passed, decodes them with one launch and hands every caller its own result.  Thread-safe; each
 * call blocks like ldpc_decode_one and returns exactly what ldpc_decode_one would.  The context must not be used
 * directly while a batcher owns it; max_frames <= the context's max_batch. */
typedef struct ldpc_batcher ldpc_batcher;
ldpc_batcher *ldpc_batcher_create(ldpc_ctx *ctx, int max_frames, int max_wait_us);
void ldpc_batcher_destroy(ldpc_batcher *b);
int ldpc_batcher_decode_one(ldpc_batcher *b, int max_iters, const double *llr, uint8_t *bits, int *iters, int *converged);
int ldpc_batcher_stats(ldpc_batcher *b, long *calls, long *launches);

/* ---- verification entry points (used by tests/; not needed by a harness) ------------------------
 * One teacher-forced update on the device in the context's dtype: from the state (lam, ne) at
 * the top of loop turn n produce (ne', lam') exactly as Orig.hs:81-98 / Min.hs:75-104 would, and
 * report whether the syndrome of hard(lam) is zero (Orig.hs:73-78).  All arrays host, float64,
 * frame-major: orig/lam/lam_out [batch][N], ne/ne_out [batch][E] in ldpc_code_csr edge order. */
int ldpc_debug_step(ldpc_ctx *ctx, int batch, const double *orig, const double *lam, const double *ne,
                    double *ne_out, double *lam_out, uint8_t *syndrome_zero);
/* free-running decode that also returns lam at the top of every loop turn:
 * trace_lam [batch][max_iters+1][N] float64 (turns after a frame stopped are zero-filled). */
int ldpc_decode_trace(ldpc_ctx *ctx, int max_iters, int batch, const double *llr, uint8_t *bits,
                      int32_t *iters, uint8_t *converged, double *trace_lam);

/* ---- dominant-kernel timing (bench.py's `roofline` object) ---------------------------------------
 * When enabled, the context brackets every launch of its dominant kernel (fused: the one decode
 * kernel; HBM path: the one decode kernel of a QC code, the check-node kernel otherwise) with HIP events on the stream the
 * kernel is launched on.
 * ldpc_ctx_kernel_time drains them: number of launches and their summed duration (ms) since the
 * last call.  Blocks until the recorded launches have finished. */
int ldpc_ctx_set_timing(ldpc_ctx *ctx, int enabled);
int ldpc_ctx_kernel_time(ldpc_ctx *ctx, int *launches, double *total_ms);
/* name of that kernel as it appears in a rocprofv3 kernel trace (substring): the kernel family before the context's
 * first decode, narrowed to the launched template instance after it */
const char *ldpc_ctx_kernel_name(const ldpc_ctx *ctx);
/* launch geometry of that kernel after the first decode: threads per workgroup and frames one workgroup decodes (the
 * on-chip kernels and the frame-per-workgroup HBM kernels of QC codes); 0/0 for the batch-major HBM kernels */
int ldpc_ctx_kernel_geometry(const ldpc_ctx *ctx, int *threads_per_workgroup, int *frames_per_workgroup);

/* ---- run-time specialised kernels ------------------------------------------------------------------
 * The fused kernels take the graph as compile-time constants.  For the shipped matrices those instances are built
 * ahead of time; for any other single-circulant quasi-cyclic H (what the reference's QC decoders accept,
 * Fast/Arraylet.hs:68-79) ldpc_ctx_create compiles one the first time -- with the ROCm tool chain (hipcc --genco in a child
 * process) when it is installed, else in-process with hiprtc; LDPC_JIT_COMPILER=hipcc|hiprtc forces one -- and caches the
 * code object on disk (LDPC_JIT_CACHE, default jit_cache/ next to the library).  These entry points let a host warm that cache
 * ahead of time -- they need no GPU -- and look at what would be compiled.  LDPC_JIT=0 disables the mechanism
 * (table-driven / generic kernels are used instead). */
const char *ldpc_jit_cache_dir(void);
/* the generated translation unit for (code, variant, dtype): copies up to cap-1 bytes, returns its full length,
 * or a negative LDPC_E* (LDPC_EUNSUPPORTED with the reason when this code has no such kernel) */
long ldpc_jit_source(const ldpc_code *code, int variant, int dtype, char *buf, size_t cap);
/* compile into the cache (or find it there); kernel_name receives the symbol rocprofv3 will list */
int ldpc_jit_prepare(const ldpc_code *code, int variant, int dtype, char *kernel_name, size_t cap, int *from_cache,
                     double *seconds);
/* the same for a given schedule (ldpc_schedule).  What is specialised at run time: flooding f32 (min-sum, tanh), flooding
 * LDPC_F16PK (min-sum), and the on-chip layered kernels (min-sum; LDPC_F32 or LDPC_F16PK, layers = the block rows).  The two
 * functions above are these with LDPC_SCHED_FLOODING. */
long ldpc_jit_source_for(const ldpc_code *code, int variant, int dtype, int schedule, char *buf, size_t cap);
int ldpc_jit_prepare_for(const ldpc_code *code, int variant, int dtype, int schedule, char *kernel_name, size_t cap,
                         int *from_cache, double *seconds);

/* ---- frame source and error tally for a BER / throughput harness -----------------------------------
 * The reference leaves message generation, BPSK + AWGN and BER statistics to the external tester
 * (ecc-manifold `eccMain`, main/Main.hs:41-48); a harness built on this library can keep frames
 * on the device instead.  Encoding is the reference's systematic rule
 *   codeword = msg ++ take (c_length - k) (msg * G)   (Utils.hs:61, Reference/Orig.hs:25-26,
 * Fast/Encoder.hs:26-63).  G is given dense, row-major bytes [k][p] (0/1), or NULL for all-zero
 * codewords (codes shipped without a generator).  n_tx = transmitted length (c_length, Utils.hs:50);
 * positions n_tx..N-1 are punctured: LLR 0 (Utils.hs:55). */
typedef struct ldpc_sim ldpc_sim;
ldpc_sim *ldpc_sim_create(const ldpc_code *code, int k, int n_tx, int p, const uint8_t *G, int max_batch);
ldpc_sim *ldpc_sim_create_on(const ldpc_code *code, int device, int k, int n_tx, int p, const uint8_t *G, int max_batch);
void ldpc_sim_destroy(ldpc_sim *sim);
/* The generator in QUASI-CYCLIC form, encoded the way the reference's fast encoder does it (Fast/Encoder.hs:26-63:
 * the message cut into sz-bit words, parity word of block column c = XOR over block rows of mulWord = rotate-and-xor
 * over the set message bits).  circ [block_rows][block_cols][sz/32] = the integers of G.q as little-endian 32-bit words
 * (bit b = first-row entry of column b, QuasiCyclic.hs:19-25; ldpc_matrix_qc_words); k = sz * block_rows, parity length
 * sz * block_cols >= n_tx - k.  sz must be 32, 64, 128 or 256 as in the reference (Encoder.hs:28-33: LDPC_EUNSUPPORTED
 * otherwise -- use the dense form).  The device table is the 32 bit-rotations of each circulant
 * (block_rows * block_cols * sz words), never the expanded k x p matrix. */
ldpc_sim *ldpc_sim_create_qc_on(const ldpc_code *code, int device, int k, int n_tx, int sz, int block_rows, int block_cols,
                                const uint32_t *circ, int max_batch);
enum { LDPC_ENCODER_NONE = 0, LDPC_ENCODER_DENSE = 1, LDPC_ENCODER_QC = 2 };
int ldpc_sim_encoder(const ldpc_sim *sim);    /* which of the three this frame source encodes with */
/* the encoder alone: codewords [batch][n_tx] bytes (device) of the same messages ldpc_sim_generate would use (message
 * bits of frame f depend on (seed, f) only); d_msg [batch][k] may be NULL.  Enqueued on `stream`. */
int ldpc_sim_encode_batch(ldpc_sim *sim, uint64_t seed, uint64_t first_frame, int batch, uint8_t *d_codewords, uint8_t *d_msg, void *stream);
/* frames [first_frame, first_frame+batch) of the stream identified by `seed`, at Eb/N0 (dB):
 * d_llr [batch][N] float32 (device), d_msg [batch][k] bytes (device, may be NULL).  Enqueued on
 * `stream` (NULL = the HIP default stream -- NOT a context's stream), not synchronised: pass the same
 * explicit stream to generate, decode and tally.  The message words stay inside `sim` for
 * the next ldpc_sim_tally call. */
int ldpc_sim_generate(ldpc_sim *sim, uint64_t seed, uint64_t first_frame, int batch, double ebn0_db,
                      float *d_llr, uint8_t *d_msg, void *stream);
/* same frames, LLRs written as fp16 (= the float32 values of ldpc_sim_generate, saturated to +-65504 and rounded
 * to nearest even) for ldpc_decode_batch_dev_f16 */
int ldpc_sim_generate_f16(ldpc_sim *sim, uint64_t seed, uint64_t first_frame, int batch, double ebn0_db,
                          uint16_t *d_llr, uint8_t *d_msg, void *stream);
/* d_tally[4] (device, uint64) += {frames, frame errors, message-bit errors, sum of iterations}
 * for the frames of the last ldpc_sim_generate call; d_iters may be NULL. */
int ldpc_sim_tally(ldpc_sim *sim, int batch, const uint8_t *d_bits, const int32_t *d_iters,
                   uint64_t *d_tally, void *stream);
/* host-side encode of one message with the same rule (parity only, p bytes) -- used by tests */
int ldpc_sim_encode_host(const ldpc_sim *sim, const uint8_t *msg, uint8_t *parity);

/* ---- matrix ingest --------------------------------------------------------------------------------
 * replaces loadMatrix (src/Data/BitMatrix/Loader.hs:58-81): `name` is "<matrix>/H" or "<matrix>/G";
 * for each loader in the reference's order (.q, .alist, .m; Loader.hs:53-57) the first existing
 * file <codes_dir>/<name>.<suffix> is parsed: .q = QuasiCyclic.hs:52-56 (integers of any size),
 * .alist = the reference's reader, Alist.hs:30-46 (zeros dropped, ROWS first, row lists only),
 * .m = Data/BitMatrix/Matlab.hs:20-26. */
typedef struct ldpc_matrix ldpc_matrix;
ldpc_matrix *ldpc_matrix_load(const char *codes_dir, const char *name);
/* one file in MacKay's published alist order (N M first, zero-padded lists), e.g.
 * codes/1920.1280.3.303, which the reference's reader cannot load. */
ldpc_matrix *ldpc_matrix_load_mackay(const char *path);
void ldpc_matrix_destroy(ldpc_matrix *m);
/* rows/cols of the EXPANDED matrix (getNRows/getNCols, Loader.hs:31-46); qc_sz = 0 if not .q */
int ldpc_matrix_info(const ldpc_matrix *m, int *rows, int *cols, int *qc_sz, int *block_rows, int *block_cols);
int ldpc_matrix_dense(const ldpc_matrix *m, uint8_t *out /* rows*cols bytes 0/1 */);  /* QuasiCyclic.hs:19-25 */
/* rank over GF(2) of the expanded matrix (a stand-alone parity-check matrix defines cols - rank message bits:
 * codes/1920.1280.A holds 5760 checks of rank 1280); negative = LDPC_E* */
int ldpc_matrix_rank(const ldpc_matrix *m);
/* a QC source's first-row patterns as little-endian 32-bit words: out [block_rows][block_cols][ceil(sz/32)]
 * (what Fast/Encoder.hs:38-39 converts the Integers of G.q to) */
int ldpc_matrix_qc_words(const ldpc_matrix *m, uint32_t *out);
/* rotation table of a .q matrix, -1 = empty block (Fast/Arraylet.hs:68-79); LDPC_EUNSUPPORTED for
 * a block holding more than one circulant (the reference errors there too). */
int ldpc_matrix_qc_offsets(const ldpc_matrix *m, int32_t *offsets /* block_rows*block_cols */);
/* QC graph when every block is a single circulant, generic CSR otherwise */
ldpc_code *ldpc_code_from_matrix(const ldpc_matrix *m);

/* ---- the plug-in record ----------------------------------------------------------------------------
 * C mirror of what mkLDPC returns (src/ECC/Code/LDPC/Utils.hs:35-75): ECC{name, encode, decode,
 * message_length, codeword_length}, selected by the reference's code-name grammar
 *   ldpc/<decoder>/<matrix-name>/<max-rounds>[/<x>/<y>]        (Utils.hs:82-88,100-108; rate x%y)
 * with <decoder> in {hip-tanh, hip-minsum}[-layered][-bool][-f32|-f64|-f16] or hip-tanh-cm-f64 (-bool: H taken as a plain
 * Boolean matrix, the `Matrix Bool` decoders' input).  The reference's own decoder names are accepted as aliases, so a
 * command line written for it runs unchanged: reference, sparse -> hip-tanh-bool; min, sparsemin -> hip-minsum-bool;
 * arraylet, cuda-arraylet1, cuda-arraylet2, two-arrays, cuda-arraylet-cm -> hip-tanh; arraylet-min -> hip-minsum;
 * arraylet-cm -> hip-tanh-cm-f64 (a dtype suffix may follow: ldpc/reference-f64/...).  The ECC keeps the name it was
 * asked for.  NULL + LDPC_ENOTFOUND for any other name (the factory's `_ -> return []`). One decoder replica is created (maxThreadCount = 1, like
 * the CUDA plug-ins, GPU/CUDA/Arraylet2.hs:61). */
typedef struct ldpc_ecc ldpc_ecc;
ldpc_ecc *ldpc_ecc_create(const char *codes_dir, const char *code_name, int max_batch);
/* the reference's maxThreadCount (Utils.hs:53 replicateM maxThreadCount $ decoder0 h): n_replicas decoder replicas in
 * ONE process, replica i on HIP device devices[i] (devices = NULL: all on the calling thread's device) -- one host
 * process can drive every GPU of a node.  ldpc_ecc_decode picks the replica by calling thread, as Utils.hs:63-69
 * does (thread ordinal mod n_replicas); ldpc_ecc_decode_on names it. */
ldpc_ecc *ldpc_ecc_create_replicas(const char *codes_dir, const char *code_name, int max_batch, int n_replicas, const int *devices);
int ldpc_ecc_replicas(const ldpc_ecc *ecc);
ldpc_ctx *ldpc_ecc_ctx_at(ldpc_ecc *ecc, int replica);
ldpc_sim *ldpc_ecc_sim_at(ldpc_ecc *ecc, int replica);
int ldpc_ecc_decode_on(ldpc_ecc *ecc, int replica, const double *llr, uint8_t *msg_bits, int *ok);
/* route ldpc_ecc_decode through a batcher per replica (see above): max_frames = 0 switches it off again */
int ldpc_ecc_set_coalescing(ldpc_ecc *ecc, int max_frames, int max_wait_us);
int ldpc_ecc_coalescing_stats(ldpc_ecc *ecc, long *calls, long *launches);   /* summed over the replicas */
void ldpc_ecc_destroy(ldpc_ecc *ecc);
const char *ldpc_ecc_name(const ldpc_ecc *ecc);            /* Utils.hs:60 */
int ldpc_ecc_message_length(const ldpc_ecc *ecc);          /* Utils.hs:73 */
int ldpc_ecc_codeword_length(const ldpc_ecc *ecc);         /* Utils.hs:74 (after puncturing) */
int ldpc_ecc_unpunctured_length(const ldpc_ecc *ecc);      /* cols H */
int ldpc_ecc_max_iters(const ldpc_ecc *ecc);
/* encode: msg [message_length] bytes -> codeword [codeword_length] bytes (Utils.hs:61) */
int ldpc_ecc_encode(const ldpc_ecc *ecc, const uint8_t *msg, uint8_t *codeword);
/* decode: llr [codeword_length] doubles -> msg_bits [message_length]; *ok = the record's Bool
 * (Utils.hs:62-72: un-puncture with zeros, decode, take message_length) */
int ldpc_ecc_decode(ldpc_ecc *ecc, const double *llr, uint8_t *msg_bits, int *ok);
/* the pieces, for batched use (owned by the record) */
ldpc_ctx *ldpc_ecc_ctx(ldpc_ecc *ecc);
ldpc_sim *ldpc_ecc_sim(ldpc_ecc *ecc);
const ldpc_code *ldpc_ecc_code(const ldpc_ecc *ecc);

#ifdef __cplusplus
}
#endif
#endif /* LDPC_HIP_H */
