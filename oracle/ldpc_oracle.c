/*
 * ldpc_oracle.c -- CPU restatement (double precision) of the reference's flooding
 * belief-propagation LDPC decoder.  TEST INFRASTRUCTURE ONLY.
 *
 *   Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 *   library.  The product (ecc_ldpc_amd/, libldpc_hip.so) never links, imports or calls it.
 *
 * PARITY UNPINNED: the reference (ku-fpg/ecc-ldpc) ships no tests, golden vectors or
 * known-answer values for this path, and it is Haskell (no GHC in
 * this image), so this restatement cannot be checked against reference output.  It is
 * pinned only by (a) two independent restatements of the same source lines -- the literal
 * dense one and the sparse one below, plus a NumPy transliteration in oracle/literal.py --
 * that must agree bit-for-bit, and (b) invariants read off the source (tests/).
 * Two things the reference DOES hold pin the conventions around the arithmetic (not the arithmetic): NOTES.txt's three
 * bit-error counts (the channel's sigma^2, tests/test_notes_pin_gpu.py + tests/test_oracle_ext.py) and
 * codes/Gmat.m.gz + codes/X.gz (the .q parse and the rotation direction of the quasi-cyclic expansion,
 * tests/test_formats.py).
 *
 * What is restated (paths relative to /root/reference):
 *   src/ECC/Code/LDPC/Reference/Orig.hs:58-98   ldpc / loop / ans / ne' / lam'  (tanh rule)
 *   src/ECC/Code/LDPC/Reference/Min.hs:54-104   same loop, (-3/4) * foldr1 min' (min-sum)
 *   src/ECC/Code/LDPC/Utils.hs:113-117          atanh' clamp 18.714973875118524
 *   src/ECC/Code/LDPC/Fast/CachedMult.hs:25-56,233-264   StableDiv / lit / smult / sdiv and the `arraylet-cm` loop
 *       (variant ORACLE_TANH_CM: same real function as the tanh rule, different roundings -- the row product is
 *        kept as (factor closest to zero, product of the others) and the leave-one-out value comes from a division;
 *        the column sum is orig + foldr1 (+) instead of foldr (+) orig)
 *   ecc-manifold ECC.Types.hard (absent); restated in-tree at
 *   src/ECC/Code/LDPC/GPU/Reference.hs:59-60    hard x = x > 0
 * Extension WITHOUT a reference counterpart (BASELINE.json configs[4]): oracle_decode_layered below -- a
 * row-layered schedule of the same check rules with its own stopping rule; it is its own specification.
 * Third-party arithmetic the Haskell relies on (not under /root/reference):
 *   base-4.9.1.0 (GHC 8.0.2; stack.yaml:1 resolver lts-8.14), GHC.Float instance Floating Double:
 *     tanh  = C libm tanh;   atanh x = 0.5 * log ((1.0+x) / (1.0-x));   product = left fold from 1
 *   bitvec-0.1.0.2 Data.Bit Num instance = GF(2) (stack.yaml:55): the syndrome is an XOR.
 *
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction: every + and * below must round once,
 * like the Haskell Double primops).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_TANH 0
#define ORACLE_MINSUM 1
#define ORACLE_TANH_CM 2 /* the reference's `arraylet-cm` numerics (Fast/CachedMult.hs), sparse form only */
/* The arithmetic of the reference's LIVE GPU decoder, `cuda-arraylet2` (GPU/CUDA/Arraylet2.hs:88-331 driving
 * cudabits/arraylet2.cu:43-83 selfProduct and cudabits/common.h:82-88 atanh_, :151-178 updateLam; float_ty = float, common.h:1) --
 * not the Double arithmetic of Orig.hs, in the saturation regime a different function (r04, sparse form only):
 *   channel LLRs, lam and messages are FLOATS (Arraylet2.hs:153 double2Float);
 *   a factor is  (float) tanh(-((double) lam - (double) ne) / 2)   (arraylet2.cu:51,56: v is a double, so the tanh is the double one,
 *       its result stored into the float array smem);
 *   the leave-one-out product runs in a DOUBLE register over the row's blocks in ascending block column (:73-79; an absent block
 *       holds 1);
 *   atanh_ takes it as a FLOAT (common.h:82): the clamp +-18.714973875118524 (rounded to float) fires when the product ROUNDS to +-1
 *       in float -- far earlier than a Double product does -- and otherwise it is the float atanh (CUDA's atanhf there, the C
 *       library's here: last ulps); the message is -2 * that, a float;
 *   a column is  ((orig + ne_1) + ne_2) + ...  in float, ascending block row (common.h:161-171, after copyArray orig -> lam,
 *       Arraylet2.hs:239);  hard and the loop are the reference's (parity check each turn: :176-196, out of turns -> orig: :165). */
#define ORACLE_CUDA32 3
/* The other registered decoders compute the SAME check rule as Orig.hs / Min.hs (product / foldr1 min' over the row in ascending
 * column) but add a column up in their own order -- only the last ulps of a Double differ:
 *   arraylet, arraylet-min (Fast/Arraylet.hs:185-186, Fast/ArrayletMin.hs:191-192):
 *       lam' = zipWith (+) orig_lam (foldRowsMatrixlet (+) ne')  =  orig + (ne_1 + (ne_2 + (... + ne_k))), rows ascending
 *       (Arraylet.hs:105-109: foldr1 over the column's blocks, ascending block row)
 *   sparse, sparsemin (Reference/Sparse.hs:112-114, SparseMin.hs:117-119; Data/Sparse/Matrix.hs:35-36 colSum = sum . map snd):
 *       lam' = orig + (((0 + ne_1) + ne_2) + ... + ne_k), rows ascending (the column's assoc list, built row by row)
 * encoded on top of the rule: variant + ORACLE_SUM_ARRAYLET / ORACLE_SUM_SPARSE (sparse form only). */
#define ORACLE_SUM_ARRAYLET 16
#define ORACLE_SUM_SPARSE 32

#define ORACLE_OK 0
#define ORACLE_EARG (-1)
#define ORACLE_EDEGREE (-2) /* min-sum on a degree-1 row: Haskell foldr1 on [] is a runtime error */
#define ORACLE_ENOMEM (-3)

/* ECC.Types.hard, restated GPU/Reference.hs:59-60 : x > 0  (0 and -0 map to False) */
static inline int hard(double x) { return x > 0.0; }

/* GHC base-4.9 Floating Double atanh, then Utils.hs:113-117 atanh' */
static inline double atanh_ghc(double x) { return 0.5 * log((1.0 + x) / (1.0 - x)); }
static inline double signum(double x) { return x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : x /* 0, -0, NaN */); }
static inline double atanh_clamped(double x) {
    double y = atanh_ghc(x);
    if (isinf(y)) return signum(x) * 18.714973875118524;
    return y;
}
/* Min.hs:82  min' x y = signum x * signum y * min (abs x) (abs y) */
static inline double min_prime(double x, double y) {
    double ax = fabs(x), ay = fabs(y);
    double m = (ax <= ay) ? ax : ay; /* Haskell min: if x <= y then x else y */
    return signum(x) * signum(y) * m;
}

/* ------------------------------------------------------------------------------------------
 * Literal dense form.  H is M x N bytes (0/1), ne is an M x N double matrix, every loop scans
 * all N columns exactly as the list comprehensions of Orig.hs:81-92 / Min.hs:75-85 do.
 * trace_lam: NULL or (max_iters+1) x N  -- lam at the top of loop n (n = 0 .. iterations run)
 * ---------------------------------------------------------------------------------------- */
int oracle_decode_dense(int M, int N, const uint8_t *H, int variant, int max_iters,
                        const double *orig_lam, uint8_t *bits, int *iters_out, int *converged_out,
                        double *trace_lam) {
    if (M <= 0 || N <= 0 || !H || !orig_lam || !bits) return ORACLE_EARG;
    double *ne = calloc((size_t)M * N, sizeof(double));   /* Orig.hs:64-65 orig_ne = 0 */
    double *ne2 = calloc((size_t)M * N, sizeof(double));
    double *lam = malloc(sizeof(double) * N), *lam2 = malloc(sizeof(double) * N);
    if (!ne || !ne2 || !lam || !lam2) { free(ne); free(ne2); free(lam); free(lam2); return ORACLE_ENOMEM; }
    memcpy(lam, orig_lam, sizeof(double) * N);
    int n = 0, conv = 0, rc = ORACLE_OK;
    const double *result = NULL;
    for (;;) {
        if (trace_lam) memcpy(trace_lam + (size_t)n * N, lam, sizeof(double) * N);
        /* Orig.hs:73-78  ans = H * hard(lam) over GF(2) */
        int all_zero = 1;
        for (int m = 0; m < M; m++) {
            int p = 0;
            for (int j = 0; j < N; j++) if (H[(size_t)m * N + j]) p ^= hard(lam[j]);
            if (p) { all_zero = 0; }
        }
        if (all_zero) { result = lam; conv = 1; break; }          /* Orig.hs:69 */
        if (n >= max_iters) { result = orig_lam; conv = 0; break; } /* Orig.hs:70 */
        /* Orig.hs:81-92 / Min.hs:75-85 */
        for (int m = 0; m < M && rc == ORACLE_OK; m++) {
            for (int c = 0; c < N; c++) {
                if (!H[(size_t)m * N + c]) { ne2[(size_t)m * N + c] = 0.0; continue; }
                if (variant == ORACLE_TANH) {
                    double prod = 1.0; /* product = foldl (*) 1, ascending j */
                    for (int j = 0; j < N; j++)
                        if (j != c && H[(size_t)m * N + j])
                            prod = prod * tanh(-((lam[j] - ne[(size_t)m * N + j]) / 2.0));
                    ne2[(size_t)m * N + c] = -2.0 * atanh_clamped(prod);
                } else {
                    /* foldr1 min' [x1..xk] = min' x1 (min' x2 (... xk)) : fold from the right */
                    int have = 0; double acc = 0.0;
                    for (int j = N - 1; j >= 0; j--)
                        if (j != c && H[(size_t)m * N + j]) {
                            double x = -(lam[j] - ne[(size_t)m * N + j]);
                            acc = have ? min_prime(x, acc) : x;
                            have = 1;
                        }
                    if (!have) { rc = ORACLE_EDEGREE; break; }
                    ne2[(size_t)m * N + c] = (-3.0 / 4.0) * acc;
                }
            }
        }
        if (rc != ORACLE_OK) break;
        /* Orig.hs:95-98  lam'[j] = foldr (+) orig[j] (col j of ne') = ne'[1,j] + (ne'[2,j] + (... + orig)) */
        for (int j = 0; j < N; j++) {
            double acc = orig_lam[j];
            for (int m = M - 1; m >= 0; m--) acc = ne2[(size_t)m * N + j] + acc;
            lam2[j] = acc;
        }
        { double *t = ne; ne = ne2; ne2 = t; }
        { double *t = lam; lam = lam2; lam2 = t; }
        n++;
    }
    if (rc == ORACLE_OK) {
        for (int j = 0; j < N; j++) bits[j] = (uint8_t)hard(result[j]); /* Orig.hs:59 U.map hard */
        if (iters_out) *iters_out = n;
        if (converged_out) *converged_out = conv;
    }
    free(ne); free(ne2); free(lam); free(lam2);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Sparse form: same arithmetic, same operation order, over a CSR edge list.
 * row_ptr[M+1], col_idx[E] strictly ascending inside each row (== ascending j of Orig.hs:88).
 * Column sums walk a CSC view in DESCENDING row order (== the foldr of Orig.hs:96; the zeros
 * the dense form adds are exact no-ops).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int M, N, E;
    const int32_t *row_ptr, *col_idx;
    int32_t *col_ptr, *csc_edge; /* edges of column j, ascending row */
} graph_t;

static int graph_build(graph_t *g, int M, int N, const int32_t *row_ptr, const int32_t *col_idx) {
    g->M = M; g->N = N; g->E = row_ptr[M]; g->row_ptr = row_ptr; g->col_idx = col_idx;
    g->col_ptr = calloc((size_t)N + 1, sizeof(int32_t));
    g->csc_edge = malloc(sizeof(int32_t) * (size_t)(g->E > 0 ? g->E : 1));
    if (!g->col_ptr || !g->csc_edge) return ORACLE_ENOMEM;
    for (int m = 0; m < M; m++)
        for (int e = row_ptr[m]; e < row_ptr[m + 1]; e++) {
            int c = col_idx[e];
            if (c < 0 || c >= N) return ORACLE_EARG;
            if (e > row_ptr[m] && col_idx[e - 1] >= c) return ORACLE_EARG;
            g->col_ptr[c + 1]++;
        }
    for (int j = 0; j < N; j++) g->col_ptr[j + 1] += g->col_ptr[j];
    int32_t *fill = malloc(sizeof(int32_t) * (size_t)N);
    if (!fill) return ORACLE_ENOMEM;
    memcpy(fill, g->col_ptr, sizeof(int32_t) * (size_t)N);
    for (int m = 0; m < M; m++)
        for (int e = row_ptr[m]; e < row_ptr[m + 1]; e++) g->csc_edge[fill[col_idx[e]]++] = e;
    free(fill);
    return ORACLE_OK;
}
static void graph_free(graph_t *g) { free(g->col_ptr); free(g->csc_edge); }

static int syndrome_zero(const graph_t *g, const double *lam) {
    for (int m = 0; m < g->M; m++) {
        int p = 0;
        for (int e = g->row_ptr[m]; e < g->row_ptr[m + 1]; e++) p ^= hard(lam[g->col_idx[e]]);
        if (p) return 0;
    }
    return 1;
}

/* one update: (lam, ne) -> (ne2, lam2).  Orig.hs:81-98 / Min.hs:75-104 */
static int step_sparse(const graph_t *g, int variant_and_order, const double *orig, const double *lam,
                       const double *ne, double *ne2, double *lam2) {
    double tbuf[4096];
    const int variant = variant_and_order & 15, sum_order = variant_and_order & ~15;
    for (int m = 0; m < g->M; m++) {
        int b = g->row_ptr[m], d = g->row_ptr[m + 1] - b;
        if (d > 4096) return ORACLE_EARG;
        if (variant == ORACLE_TANH) {
            for (int k = 0; k < d; k++) tbuf[k] = tanh(-((lam[g->col_idx[b + k]] - ne[b + k]) / 2.0));
            for (int k = 0; k < d; k++) {
                double prod = 1.0;
                for (int j = 0; j < d; j++) if (j != k) prod = prod * tbuf[j];
                ne2[b + k] = -2.0 * atanh_clamped(prod);
            }
        } else if (variant == ORACLE_CUDA32) {
            float fbuf[4096];
            for (int k = 0; k < d; k++) {
                const double v = (double)(float)ne[b + k];                                         /* arraylet2.cu:51 double v = mLet[...] */
                fbuf[k] = (float)tanh(-(((double)(float)lam[g->col_idx[b + k]] - v) / 2));          /* :56 */
            }
            for (int k = 0; k < d; k++) {
                double r = 1;                                                                      /* :49 */
                for (int j = 0; j < k; j++) r *= fbuf[j];                                          /* :73-75 */
                for (int j = k + 1; j < d; j++) r *= fbuf[j];                                      /* :77-79 */
                const float x = (float)r;                                                          /* atanh_(float_ty x), common.h:82 */
                float y;
                if (x == 1 || x == -1) y = (float)((x < 0 ? -1.0f : (x > 0 ? 1.0f : 0.0f)) * 18.714973875118524);   /* :83-85 */
                else y = atanhf(x);                                                                /* :87 */
                ne2[b + k] = (double)(-2 * y);                                                     /* arraylet2.cu:81 */
            }
        } else if (variant == ORACLE_TANH_CM) {
            /* CachedMult.hs:247-259: ne_tanh'mat, then per row  foldr1 smult [lit x | ascending block column]
             * (foldColsMatrixletU, :184-188), then ne' = -2 * atanh' (S `sdiv` x) */
            if (d == 0) continue;
            for (int k = 0; k < d; k++) tbuf[k] = tanh(-((lam[g->col_idx[b + k]] - ne[b + k]) / 2.0));
            double sa, sb;   /* StableDiv (a, b): a = factor closest to zero, b = product of the rest (:25-29) */
            {   /* lit (:41-44): x >= 1 -> (1, x) else (x, 1) */
                double x = tbuf[d - 1];
                if (x >= 1.0) { sa = 1.0; sb = x; } else { sa = x; sb = 1.0; }
            }
            for (int k = d - 2; k >= 0; k--) {   /* smult (lit x_k) acc  (:46-50): (a,b) = lit x_k, (c,d) = acc */
                double x = tbuf[k], a, bq;
                if (x >= 1.0) { a = 1.0; bq = x; } else { a = x; bq = 1.0; }
                double mn, mx;                      /* absMinMax a c (:31-34): abs a < abs c -> (a, c) else (c, a) */
                if (fabs(a) < fabs(sa)) { mn = a; mx = sa; } else { mn = sa; mx = a; }
                sb = (bq * mx) * sb;                /* b * maxOfMins * d, left-associated */
                sa = mn;
            }
            for (int k = 0; k < d; k++) {           /* sdiv (:52-55) */
                double c = tbuf[k];
                double q = (sa == c) ? sb : sa * (sb / c);
                ne2[b + k] = -2.0 * atanh_clamped(q);
            }
        } else {
            if (d < 2) { if (d == 1) return ORACLE_EDEGREE; continue; }
            for (int k = 0; k < d; k++) tbuf[k] = -(lam[g->col_idx[b + k]] - ne[b + k]);
            for (int k = 0; k < d; k++) {
                int have = 0; double acc = 0.0;
                for (int j = d - 1; j >= 0; j--) if (j != k) { acc = have ? min_prime(tbuf[j], acc) : tbuf[j]; have = 1; }
                ne2[b + k] = (-3.0 / 4.0) * acc;
            }
        }
    }
    for (int j = 0; j < g->N; j++) {
        const int q0 = g->col_ptr[j], q1 = g->col_ptr[j + 1];
        if (variant == ORACLE_TANH_CM) {
            /* CachedMult.hs:261-262  lam' = zipWith (+) orig_lam (foldRowsMatrixlet (+) ne'): per column
             * foldr1 (+) over ascending block rows (:190-194), THEN orig + that ("assumes one value on every column") */
            if (q1 == q0) { lam2[j] = orig[j]; continue; }
            double acc = ne2[g->csc_edge[q1 - 1]];
            for (int q = q1 - 2; q >= q0; q--) acc = ne2[g->csc_edge[q]] + acc;
            lam2[j] = orig[j] + acc;
            continue;
        }
        if (variant == ORACLE_CUDA32) {            /* common.h:161-171: newLam[i] += newMLet[...] over ascending block rows, in float */
            float acc = (float)orig[j];
            for (int q = q0; q < q1; q++) acc = acc + (float)ne2[g->csc_edge[q]];
            lam2[j] = (double)acc;
            continue;
        }
        if (sum_order == ORACLE_SUM_ARRAYLET) {   /* orig + foldr1 (+): as the arraylet-cm decoder above */
            if (q1 == q0) { lam2[j] = orig[j]; continue; }
            double acc = ne2[g->csc_edge[q1 - 1]];
            for (int q = q1 - 2; q >= q0; q--) acc = ne2[g->csc_edge[q]] + acc;
            lam2[j] = orig[j] + acc;
            continue;
        }
        if (sum_order == ORACLE_SUM_SPARSE) {     /* orig + sum: a left fold from 0 over the rows in ascending order */
            double acc = 0.0;
            for (int q = q0; q < q1; q++) acc = acc + ne2[g->csc_edge[q]];
            lam2[j] = orig[j] + acc;
            continue;
        }
        double acc = orig[j];
        for (int q = q1 - 1; q >= q0; q--) acc = ne2[g->csc_edge[q]] + acc;
        lam2[j] = acc;
    }
    return ORACLE_OK;
}

/* trace_lam: NULL or (max_iters+1) x N ; trace_ne: NULL or max_iters x E (ne' produced by update n) */
static int decode_sparse(const graph_t *g, int variant, int max_iters, const double *orig_lam,
                         uint8_t *bits, int *iters_out, int *converged_out, double *final_lam,
                         double *trace_lam, double *trace_ne, double *work /* 2E + 2N */) {
    const int N = g->N, E = g->E;
    double *ne = work, *ne2 = work + E, *lam = work + 2 * (size_t)E, *lam2 = lam + N;
    memset(ne, 0, sizeof(double) * (size_t)E);
    memcpy(lam, orig_lam, sizeof(double) * (size_t)N);
    int n = 0, conv = 0;
    const double *result;
    for (;;) {
        if (trace_lam) memcpy(trace_lam + (size_t)n * N, lam, sizeof(double) * (size_t)N);
        if (syndrome_zero(g, lam)) { result = lam; conv = 1; break; }
        if (n >= max_iters) { result = orig_lam; conv = 0; break; }
        int rc = step_sparse(g, variant, orig_lam, lam, ne, ne2, lam2);
        if (rc != ORACLE_OK) return rc;
        if (trace_ne) memcpy(trace_ne + (size_t)n * E, ne2, sizeof(double) * (size_t)E);
        { double *t = ne; ne = ne2; ne2 = t; }
        { double *t = lam; lam = lam2; lam2 = t; }
        n++;
    }
    for (int j = 0; j < N; j++) bits[j] = (uint8_t)hard(result[j]);
    if (final_lam) memcpy(final_lam, result, sizeof(double) * (size_t)N);
    if (iters_out) *iters_out = n;
    if (converged_out) *converged_out = conv;
    return ORACLE_OK;
}

int oracle_decode(int M, int N, const int32_t *row_ptr, const int32_t *col_idx, int variant,
                  int max_iters, const double *orig_lam, uint8_t *bits, int *iters_out,
                  int *converged_out, double *final_lam, double *trace_lam, double *trace_ne) {
    if (M <= 0 || N <= 0 || !row_ptr || !col_idx || !orig_lam || !bits || max_iters < 0) return ORACLE_EARG;
    graph_t g;
    int rc = graph_build(&g, M, N, row_ptr, col_idx);
    if (rc == ORACLE_OK) {
        double *work = malloc(sizeof(double) * (2 * (size_t)g.E + 2 * (size_t)N + 1));
        if (!work) rc = ORACLE_ENOMEM;
        else {
            rc = decode_sparse(&g, variant, max_iters, orig_lam, bits, iters_out, converged_out,
                               final_lam, trace_lam, trace_ne, work);
            free(work);
        }
    }
    graph_free(&g);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * EXTENSION (no reference counterpart; BASELINE.json configs[4] "layered min-sum + early termination").
 * Row-layered schedule of the same check rules.  Rows are partitioned into layers [layer_ptr[l], layer_ptr[l+1]);
 * the rows of one layer must not share a column (for a quasi-cyclic H with at most one circulant per block: a layer
 * = one block row), so the order inside a layer is immaterial.  Specification:
 *     lam <- orig ; msg <- 0
 *     if syndrome(hard lam) == 0: return lam, 0 sweeps                        (as Orig.hs:69 at n = 0)
 *     sweep n = 1, 2, ...:  if n > max_iters: return ORIG (not converged)       (as Orig.hs:70)
 *        for every layer, for every row m of it, with c_k its columns in ascending order:
 *           t_k    = lam[c_k] - msg[m,k]
 *           odd   |= XOR_k hard(lam[c_k])                 (parity of the decisions this row saw)
 *           msg'   = the flooding check rule applied to t  ((-3/4) foldr1 min' / -2 atanh' prod: same formulas)
 *           new_k  = t_k + msg'[m,k] ;  flip |= hard(new_k) /= hard(lam[c_k]) ;  lam[c_k] <- new_k ; msg[m,k] <- msg'
 *        if not odd and not flip: return lam, n sweeps     (no decision changed during the sweep and every check it
 *                                                           saw was satisfied => hard lam is a codeword)
 * trace_lam: NULL or (max_iters+1) x N: lam after sweep n at row n (row 0 = orig).
 * ---------------------------------------------------------------------------------------- */
/* one sweep over all rows in ascending order (layers are contiguous ascending row ranges whose rows are column-disjoint,
 * so this IS the layer-by-layer order); lam and msg are updated in place */
static int layered_sweep(int M, const int32_t *row_ptr, const int32_t *col_idx, int variant, double *lam, double *msg,
                         int *odd_out, int *flip_out) {
    double t[4096], x[4096];
    int odd = 0, flip = 0;
    for (int m = 0; m < M; m++) {
        const int b = row_ptr[m], d = row_ptr[m + 1] - b;
        if (d > 4096) return ORACLE_EARG;
        int par = 0;
        for (int k = 0; k < d; k++) {
            const double l = lam[col_idx[b + k]];
            par ^= hard(l);
            t[k] = l - msg[b + k];
        }
        odd |= par;
        if (variant == ORACLE_TANH) {
            for (int k = 0; k < d; k++) x[k] = tanh(-(t[k] / 2.0));
            for (int k = 0; k < d; k++) {
                double prod = 1.0;
                for (int j = 0; j < d; j++) if (j != k) prod = prod * x[j];
                msg[b + k] = -2.0 * atanh_clamped(prod);
            }
        } else {
            if (d == 1) return ORACLE_EDEGREE;
            for (int k = 0; k < d; k++) x[k] = -t[k];
            for (int k = 0; k < d; k++) {
                int have = 0; double acc = 0.0;
                for (int j = d - 1; j >= 0; j--) if (j != k) { acc = have ? min_prime(x[j], acc) : x[j]; have = 1; }
                msg[b + k] = (-3.0 / 4.0) * acc;
            }
        }
        for (int k = 0; k < d; k++) {
            const int c = col_idx[b + k];
            const double nw = t[k] + msg[b + k];
            flip |= hard(nw) != hard(lam[c]);
            lam[c] = nw;
        }
    }
    *odd_out = odd; *flip_out = flip;
    return ORACLE_OK;
}

/* teacher-forced sweep: (lam, msg) -> (lam', msg') */
int oracle_layered_step(int M, int N, const int32_t *row_ptr, const int32_t *col_idx, int variant, const double *lam,
                        const double *msg, double *lam_out, double *msg_out, int *odd, int *flip) {
    if (M <= 0 || N <= 0 || !row_ptr || !col_idx || !lam || !msg || !lam_out || !msg_out) return ORACLE_EARG;
    memcpy(lam_out, lam, sizeof(double) * (size_t)N);
    memcpy(msg_out, msg, sizeof(double) * (size_t)row_ptr[M]);
    int o = 0, f = 0;
    int rc = layered_sweep(M, row_ptr, col_idx, variant, lam_out, msg_out, &o, &f);
    if (odd) *odd = o;
    if (flip) *flip = f;
    return rc;
}

int oracle_decode_layered(int M, int N, const int32_t *row_ptr, const int32_t *col_idx, int n_layers,
                          const int32_t *layer_ptr, int variant, int max_iters, const double *orig_lam, uint8_t *bits,
                          int *iters_out, int *converged_out, double *final_lam, double *trace_lam) {
    if (M <= 0 || N <= 0 || !row_ptr || !col_idx || !orig_lam || !bits || max_iters < 0 || n_layers <= 0 || !layer_ptr ||
        layer_ptr[0] != 0 || layer_ptr[n_layers] != M || !(variant == ORACLE_TANH || variant == ORACLE_MINSUM))
        return ORACLE_EARG;
    graph_t g;
    int rc = graph_build(&g, M, N, row_ptr, col_idx);
    if (rc != ORACLE_OK) { graph_free(&g); return rc; }
    /* layers must be column-disjoint */
    int32_t *seen = malloc(sizeof(int32_t) * (size_t)N);
    double *msg = calloc((size_t)(g.E > 0 ? g.E : 1), sizeof(double)), *lam = malloc(sizeof(double) * (size_t)N);
    if (!seen || !msg || !lam) { free(seen); free(msg); free(lam); graph_free(&g); return ORACLE_ENOMEM; }
    for (int j = 0; j < N; j++) seen[j] = -1;
    for (int l = 0; l < n_layers && rc == ORACLE_OK; l++) {
        if (layer_ptr[l + 1] < layer_ptr[l]) { rc = ORACLE_EARG; break; }
        for (int m = layer_ptr[l]; m < layer_ptr[l + 1]; m++)
            for (int e = row_ptr[m]; e < row_ptr[m + 1]; e++) {
                if (seen[col_idx[e]] == l) { rc = ORACLE_EARG; break; }
                seen[col_idx[e]] = l;
            }
    }
    free(seen);
    int n = 0, conv = 0;
    const double *result = orig_lam;
    if (rc == ORACLE_OK) {
        memcpy(lam, orig_lam, sizeof(double) * (size_t)N);
        if (trace_lam) memcpy(trace_lam, lam, sizeof(double) * (size_t)N);
        if (syndrome_zero(&g, lam)) { conv = 1; result = lam; }
        else for (;;) {
            if (n >= max_iters) { conv = 0; result = orig_lam; break; }
            int odd = 0, flip = 0;
            rc = layered_sweep(M, row_ptr, col_idx, variant, lam, msg, &odd, &flip);
            if (rc != ORACLE_OK) break;
            n++;
            if (trace_lam) memcpy(trace_lam + (size_t)n * N, lam, sizeof(double) * (size_t)N);
            if (!odd && !flip) { conv = 1; result = lam; break; }
        }
    }
    if (rc == ORACLE_OK) {
        for (int j = 0; j < N; j++) bits[j] = (uint8_t)hard(result[j]);
        if (final_lam) memcpy(final_lam, result, sizeof(double) * (size_t)N);
        if (iters_out) *iters_out = n;
        if (converged_out) *converged_out = conv;
    }
    free(msg); free(lam);
    graph_free(&g);
    return rc;
}

/* batch driver of the layered extension (bench.py cpu_baseline leg for --schedule layered) */
typedef struct {
    int M, N; const int32_t *row_ptr, *col_idx; int n_layers; const int32_t *layer_ptr; int variant, max_iters;
    const double *llr; uint8_t *bits; int32_t *iters; uint8_t *conv; int f0, f1, rc;
} ljob_t;
static void *ljob_run(void *p) {
    ljob_t *j = p;
    for (int f = j->f0; f < j->f1; f++) {
        int it = 0, cv = 0;
        int rc = oracle_decode_layered(j->M, j->N, j->row_ptr, j->col_idx, j->n_layers, j->layer_ptr, j->variant, j->max_iters,
                                       j->llr + (size_t)f * j->N, j->bits + (size_t)f * j->N, &it, &cv, NULL, NULL);
        if (rc != ORACLE_OK) { j->rc = rc; break; }
        if (j->iters) j->iters[f] = it;
        if (j->conv) j->conv[f] = (uint8_t)cv;
    }
    return NULL;
}
int oracle_decode_layered_batch(int M, int N, const int32_t *row_ptr, const int32_t *col_idx, int n_layers, const int32_t *layer_ptr,
                                int variant, int max_iters, int frames, const double *llr, uint8_t *bits, int32_t *iters,
                                uint8_t *converged, int nthreads) {
    if (frames < 0 || nthreads < 1 || !llr || !bits) return ORACLE_EARG;
    if (nthreads > frames) nthreads = frames > 0 ? frames : 1;
    ljob_t *jobs = calloc((size_t)nthreads, sizeof(ljob_t));
    pthread_t *th = calloc((size_t)nthreads, sizeof(pthread_t));
    int rc = ORACLE_OK;
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = (ljob_t){M, N, row_ptr, col_idx, n_layers, layer_ptr, variant, max_iters, llr, bits, iters, converged,
                           (int)((long long)frames * t / nthreads), (int)((long long)frames * (t + 1) / nthreads), ORACLE_OK};
        if (t > 0) pthread_create(&th[t], NULL, ljob_run, &jobs[t]);
    }
    ljob_run(&jobs[0]);
    for (int t = 1; t < nthreads; t++) pthread_join(th[t], NULL);
    for (int t = 0; t < nthreads; t++) if (jobs[t].rc != ORACLE_OK) rc = jobs[t].rc;
    free(jobs); free(th);
    return rc;
}

/* Teacher-forcing helper: one update from a given (lam, ne) state; also reports the syndrome of lam. */
int oracle_step(int M, int N, const int32_t *row_ptr, const int32_t *col_idx, int variant,
                const double *orig_lam, const double *lam, const double *ne, double *ne_out,
                double *lam_out, int *syndrome_is_zero) {
    if (M <= 0 || N <= 0 || !row_ptr || !col_idx) return ORACLE_EARG;
    graph_t g;
    int rc = graph_build(&g, M, N, row_ptr, col_idx);
    if (rc == ORACLE_OK) {
        if (syndrome_is_zero) *syndrome_is_zero = syndrome_zero(&g, lam);
        rc = step_sparse(&g, variant, orig_lam, lam, ne, ne_out, lam_out);
    }
    graph_free(&g);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Batch driver for the cpu_baseline leg of bench.py: frames are independent, so they are
 * split over `nthreads` POSIX threads (the reference itself runs one decoder replica per
 * Haskell thread, Utils.hs:53,63-69).  llr is [frames][N] double.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const graph_t *g; int variant, max_iters; const double *llr; uint8_t *bits; int32_t *iters;
    uint8_t *conv; int f0, f1; int rc;
} job_t;

static void *job_run(void *p) {
    job_t *j = p;
    const int N = j->g->N;
    double *work = malloc(sizeof(double) * (2 * (size_t)j->g->E + 2 * (size_t)N + 1));
    if (!work) { j->rc = ORACLE_ENOMEM; return NULL; }
    for (int f = j->f0; f < j->f1; f++) {
        int it = 0, cv = 0;
        int rc = decode_sparse(j->g, j->variant, j->max_iters, j->llr + (size_t)f * N,
                               j->bits + (size_t)f * N, &it, &cv, NULL, NULL, NULL, work);
        if (rc != ORACLE_OK) { j->rc = rc; break; }
        if (j->iters) j->iters[f] = it;
        if (j->conv) j->conv[f] = (uint8_t)cv;
    }
    free(work);
    return NULL;
}

int oracle_decode_batch(int M, int N, const int32_t *row_ptr, const int32_t *col_idx, int variant,
                        int max_iters, int frames, const double *llr, uint8_t *bits,
                        int32_t *iters, uint8_t *converged, int nthreads) {
    if (frames < 0 || nthreads < 1 || !llr || !bits) return ORACLE_EARG;
    graph_t g;
    int rc = graph_build(&g, M, N, row_ptr, col_idx);
    if (rc != ORACLE_OK) { graph_free(&g); return rc; }
    if (nthreads > frames) nthreads = frames > 0 ? frames : 1;
    job_t *jobs = calloc((size_t)nthreads, sizeof(job_t));
    pthread_t *th = calloc((size_t)nthreads, sizeof(pthread_t));
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = (job_t){&g, variant, max_iters, llr, bits, iters, converged,
                          (int)((long long)frames * t / nthreads), (int)((long long)frames * (t + 1) / nthreads), ORACLE_OK};
        if (t > 0) pthread_create(&th[t], NULL, job_run, &jobs[t]);
    }
    job_run(&jobs[0]);
    for (int t = 1; t < nthreads; t++) pthread_join(th[t], NULL);
    for (int t = 0; t < nthreads; t++) if (jobs[t].rc != ORACLE_OK) rc = jobs[t].rc;
    free(jobs); free(th);
    graph_free(&g);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Encoders.
 *  oracle_encode_dense: Orig.hs:25-26  parity = v (1 x k) * G (k x p) over GF(2) (Data.Bit)
 *  oracle_encode_qc   : Fast/Encoder.hs:26-63  word-packed circulant multiply; entry bit b of
 *                       block (row,col) set => message bit i of that block row adds (xor) a 1 at
 *                       parity position (i + b) mod sz of that block column (mulWord: rotateL by i).
 *                       gq holds each block as sz bytes (bit b -> gq[...][b]).
 * ---------------------------------------------------------------------------------------- */
int oracle_encode_dense(int k, int p, const uint8_t *G, const uint8_t *msg, uint8_t *parity) {
    if (k <= 0 || p <= 0 || !G || !msg || !parity) return ORACLE_EARG;
    for (int c = 0; c < p; c++) {
        int acc = 0;
        for (int r = 0; r < k; r++) acc ^= (msg[r] & G[(size_t)r * p + c]);
        parity[c] = (uint8_t)(acc & 1);
    }
    return ORACLE_OK;
}

int oracle_encode_qc(int sz, int brows, int bcols, const uint8_t *gq /*[brows][bcols][sz]*/,
                     const uint8_t *msg /*brows*sz*/, uint8_t *parity /*bcols*sz*/) {
    if (sz <= 0 || brows <= 0 || bcols <= 0 || !gq || !msg || !parity) return ORACLE_EARG;
    memset(parity, 0, (size_t)bcols * sz);
    for (int c = 0; c < bcols; c++)
        for (int r = 0; r < brows; r++) {
            const uint8_t *w2 = gq + ((size_t)r * bcols + c) * sz;
            for (int i = 0; i < sz; i++) {
                if (!msg[(size_t)r * sz + i]) continue;   /* w1 `testBit` i */
                for (int b = 0; b < sz; b++)               /* w2 rotateL i: bit b -> (b+i) mod sz */
                    if (w2[b]) parity[(size_t)c * sz + (b + i) % sz] ^= 1;
            }
        }
    return ORACLE_OK;
}
