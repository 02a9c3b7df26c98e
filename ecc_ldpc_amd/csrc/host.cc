// host.cc -- C++ mirror of the reference's host side of the decode path:
//   * matrix ingest  (src/Data/BitMatrix/Loader.hs:53-81, Alist.hs:30-46, Matlab.hs:20-26,
//                     src/Data/Matrix/QuasiCyclic.hs:19-25,52-56, Fast/Arraylet.hs:68-79)
//   * the plug-in record `mkLDPC` builds (src/ECC/Code/LDPC/Utils.hs:35-75): name, encode, decode,
//     message_length, codeword_length, with the same code-name grammar (Utils.hs:82-88,100-108):
//       ldpc/<decoder>/<matrix>/<max-rounds>[/<x>/<y>]      rate = x % y
//     decoders registered here: hip-tanh, hip-minsum, optionally -layered (row-layered schedule, an extension),
//     then an optional dtype suffix -f32 (default), -f64, -f16 (new tokens next to the reference's registry,
//     main/Main.hs:34-36).
// GHC is not available in this image, so this layer is C++ behind the same C ABI; INTEGRATION.md
// has the Haskell module that calls the ABI from the reference itself.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <fstream>
#include <memory>
#include <mutex>
#include <new>
#include <sstream>
#include <string>
#include <vector>

#include "internal.h"

using ldpc::set_error;

// ------------------------------------------------------------------ arbitrary-size bit rows
// a non-negative integer read from decimal text, as little-endian 32-bit limbs
static bool parse_decimal(const std::string &tok, std::vector<uint32_t> &limbs) {
    limbs.assign(1, 0u);
    if (tok.empty()) return false;
    for (char ch : tok) {
        if (ch < '0' || ch > '9') return false;
        uint64_t carry = (uint64_t)(ch - '0');
        for (size_t i = 0; i < limbs.size(); i++) {
            uint64_t v = (uint64_t)limbs[i] * 10u + carry;
            limbs[i] = (uint32_t)v;
            carry = v >> 32;
        }
        if (carry) limbs.push_back((uint32_t)carry);
    }
    return true;
}
static inline bool limb_bit(const std::vector<uint32_t> &l, int b) {
    size_t w = (size_t)b >> 5;
    return w < l.size() && ((l[w] >> (b & 31)) & 1u);
}
static int limb_top(const std::vector<uint32_t> &l) {  // index of highest set bit, -1 if zero
    for (int w = (int)l.size() - 1; w >= 0; w--)
        if (l[w]) return w * 32 + 31 - __builtin_clz(l[w]);
    return -1;
}

// largest matrix the loaders expand to bytes (the reference's Matrix Bool is dense too); bigger graphs come in as CSR
static constexpr size_t kMaxDenseBytes = (size_t)1 << 31;
static constexpr size_t kMaxDim = (size_t)1 << 24;   // rows / columns of an expanded matrix (int32 edge ids stay safe)

struct ldpc_matrix {
    int rows = 0, cols = 0;                       // EXPANDED size (getNRows/getNCols, Loader.hs:31-46)
    int sz = 0, brows = 0, bcols = 0;             // quasi-cyclic description when sz > 0
    std::vector<std::vector<uint32_t>> blocks;    // [brows*bcols] first-row bit patterns (QC)
    std::vector<uint8_t> dense;                   // rows*cols bytes when not QC
    std::string source;
};

static bool read_file(const std::string &path, std::string &out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::stringstream ss;
    ss << f.rdbuf();
    out = ss.str();
    return true;
}
static bool file_exists(const std::string &p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}
static std::vector<std::string> split_ws(const std::string &s) {
    std::vector<std::string> out;
    std::istringstream is(s);
    std::string t;
    while (is >> t) out.push_back(t);
    return out;
}

// QuasiCyclic.hs:52-56: cycle size, then a Matlab-style matrix of integers (one row per line)
static int parse_q(const std::string &text, ldpc_matrix *m) {
    std::istringstream is(text);
    std::string line;
    std::vector<std::vector<std::string>> rows;
    bool have_sz = false;
    while (std::getline(is, line)) {
        auto toks = split_ws(line);
        if (toks.empty()) continue;
        if (!have_sz) {
            char *end = nullptr;
            long v = strtol(toks[0].c_str(), &end, 10);
            if (*end || v <= 0 || v > (1 << 20)) return set_error(LDPC_EFORMAT, ".q: bad cycle size '%s'", toks[0].c_str());
            m->sz = (int)v;
            have_sz = true;
            toks.erase(toks.begin());
            if (toks.empty()) continue;
        }
        rows.push_back(toks);
    }
    if (!have_sz || rows.empty()) return set_error(LDPC_EFORMAT, ".q: empty matrix");
    m->brows = (int)rows.size();
    m->bcols = (int)rows[0].size();
    if ((size_t)m->brows * m->sz > kMaxDim || (size_t)m->bcols * m->sz > kMaxDim)
        return set_error(LDPC_EUNSUPPORTED, ".q: %d x %d blocks of size %d expand beyond %zu rows/columns", m->brows, m->bcols, m->sz, kMaxDim);
    m->blocks.resize((size_t)m->brows * m->bcols);
    for (int r = 0; r < m->brows; r++) {
        if ((int)rows[r].size() != m->bcols) return set_error(LDPC_EFORMAT, ".q: ragged row %d", r);
        for (int c = 0; c < m->bcols; c++) {
            auto &l = m->blocks[(size_t)r * m->bcols + c];
            if (!parse_decimal(rows[r][c], l)) return set_error(LDPC_EFORMAT, ".q: bad integer '%s'", rows[r][c].c_str());
            if (limb_top(l) >= m->sz) return set_error(LDPC_EFORMAT, ".q: entry (%d,%d) has bits above the cycle size %d", r, c, m->sz);
        }
    }
    m->rows = m->brows * m->sz;
    m->cols = m->bcols * m->sz;
    return LDPC_OK;
}

// Alist.hs:30-46: zeros are filtered out first; rows, cols, 2 ignored, row counts, col counts, row lists
static int parse_alist_reference(const std::string &text, ldpc_matrix *m) {
    std::vector<long> t;
    for (auto &s : split_ws(text)) {
        char *end = nullptr;
        long v = strtol(s.c_str(), &end, 10);
        if (*end) return set_error(LDPC_EFORMAT, "alist: bad token '%s'", s.c_str());
        if (v != 0) t.push_back(v);
    }
    size_t pos = 0;
    auto item = [&](long &v) { if (pos >= t.size()) return false; v = t[pos++]; return true; };
    long n, mm, ign;
    if (!item(n) || !item(mm) || !item(ign) || !item(ign) || n <= 0 || mm <= 0) return set_error(LDPC_EFORMAT, "alist: truncated header");
    // the counts alone need n + mm tokens: checked BEFORE anything is sized by the header (a damaged header must
    // not turn into a multi-gigabyte allocation)
    if ((unsigned long)n > t.size() || (unsigned long)mm > t.size() || (size_t)n + (size_t)mm > t.size() - pos)
        return set_error(LDPC_EFORMAT, "alist: header says %ld x %ld but the file holds %zu numbers", n, mm, t.size());
    if ((size_t)n * (size_t)mm > kMaxDenseBytes)
        return set_error(LDPC_EUNSUPPORTED, "alist: %ld x %ld is beyond the dense loader's limit; pass the graph as CSR", n, mm);
    std::vector<long> num_n((size_t)n), num_m((size_t)mm);
    for (auto &v : num_n) if (!item(v)) return set_error(LDPC_EFORMAT, "alist: truncated row counts");
    for (auto &v : num_m) if (!item(v)) return set_error(LDPC_EFORMAT, "alist: truncated column counts");
    m->rows = (int)n; m->cols = (int)mm;
    m->dense.assign((size_t)n * mm, 0);
    for (long r = 0; r < n; r++)
        for (long c = 0; c < num_n[(size_t)r]; c++) {
            long v;
            if (!item(v) || v < 1 || v > mm) return set_error(LDPC_EFORMAT, "alist: bad entry in row %ld", r + 1);
            m->dense[(size_t)r * mm + (v - 1)] = 1;
        }
    return LDPC_OK;
}

// MacKay's published order: N M / max col wt, max row wt / col weights / row weights / col lists / row lists
static int parse_alist_mackay(const std::string &text, ldpc_matrix *m) {
    std::vector<long> t;
    for (auto &s : split_ws(text)) {
        char *end = nullptr;
        long v = strtol(s.c_str(), &end, 10);
        if (*end) return set_error(LDPC_EFORMAT, "alist: bad token '%s'", s.c_str());
        t.push_back(v);
    }
    if (t.size() < 4) return set_error(LDPC_EFORMAT, "alist: truncated header");
    long N = t[0], M = t[1], maxc = t[2], maxr = t[3];
    if (N <= 0 || M <= 0 || maxc <= 0 || maxr <= 0) return set_error(LDPC_EFORMAT, "alist: bad header");
    if ((unsigned long)N > t.size() || (unsigned long)M > t.size() || (unsigned long)maxc > t.size() || (unsigned long)maxr > t.size())
        return set_error(LDPC_EFORMAT, "alist: header %ld %ld %ld %ld is larger than the file (%zu numbers)", N, M, maxc, maxr, t.size());
    if ((size_t)N * (size_t)M > kMaxDenseBytes)
        return set_error(LDPC_EUNSUPPORTED, "alist: %ld x %ld is beyond the dense loader's limit; pass the graph as CSR", M, N);
    size_t need = 4 + (size_t)N + M + (size_t)N * maxc + (size_t)M * maxr;
    if (t.size() < need) return set_error(LDPC_EFORMAT, "alist: %zu tokens, MacKay layout needs %zu", t.size(), need);
    size_t pos = 4;
    std::vector<long> colw(t.begin() + pos, t.begin() + pos + N); pos += N;
    std::vector<long> roww(t.begin() + pos, t.begin() + pos + M); pos += M;
    // a weight beyond the header's maxima would walk past the column's / row's slot (and past the token vector)
    for (long c = 0; c < N; c++)
        if (colw[(size_t)c] < 0 || colw[(size_t)c] > maxc) return set_error(LDPC_EFORMAT, "alist: weight %ld of column %ld outside [0,%ld]", colw[(size_t)c], c + 1, maxc);
    for (long r = 0; r < M; r++)
        if (roww[(size_t)r] < 0 || roww[(size_t)r] > maxr) return set_error(LDPC_EFORMAT, "alist: weight %ld of row %ld outside [0,%ld]", roww[(size_t)r], r + 1, maxr);
    m->rows = (int)M; m->cols = (int)N;
    m->dense.assign((size_t)M * N, 0);
    for (long c = 0; c < N; c++, pos += maxc)
        for (long q = 0; q < colw[(size_t)c]; q++) {
            long r = t[pos + q];
            if (r < 1 || r > M) return set_error(LDPC_EFORMAT, "alist: bad row index %ld in column %ld", r, c + 1);
            m->dense[(size_t)(r - 1) * N + c] = 1;
        }
    for (long r = 0; r < M; r++, pos += maxr)
        for (long q = 0; q < roww[(size_t)r]; q++) {
            long c = t[pos + q];
            if (c < 1 || c > N || !m->dense[(size_t)r * N + (c - 1)])
                return set_error(LDPC_EFORMAT, "alist: row list of row %ld disagrees with the column lists", r + 1);
        }
    return LDPC_OK;
}

// Data/BitMatrix/Matlab.hs:20-26: lines of "0"/"1" words
static int parse_m(const std::string &text, ldpc_matrix *m) {
    std::istringstream is(text);
    std::string line;
    int cols = -1, rows = 0;
    while (std::getline(is, line)) {
        auto toks = split_ws(line);
        if (toks.empty()) continue;
        if (cols < 0) cols = (int)toks.size();
        if ((int)toks.size() != cols) return set_error(LDPC_EFORMAT, ".m: ragged row %d", rows);
        for (auto &w : toks) {
            if (w == "0") m->dense.push_back(0);
            else if (w == "1") m->dense.push_back(1);
            else return set_error(LDPC_EFORMAT, "readBit: no parse '%s'", w.c_str());
        }
        rows++;
    }
    if (rows == 0) return set_error(LDPC_EFORMAT, ".m: empty matrix");
    m->rows = rows; m->cols = cols;
    return LDPC_OK;
}

extern "C" {

void ldpc_matrix_destroy(ldpc_matrix *m) { delete m; }

// Loader.hs:58-81 loadMatrix: for each loader (q, alist, m) x each prefix, first file that exists wins
ldpc_matrix *ldpc_matrix_load(const char *codes_dir, const char *name) {
    if (!codes_dir || !name) { set_error(LDPC_EINVAL, "null argument"); return nullptr; }
    ldpc_matrix *m = new (std::nothrow) ldpc_matrix();
    if (!m) { set_error(LDPC_ENOMEM, "out of host memory"); return nullptr; }
    try {
        const char *suffixes[3] = {"q", "alist", "m"};
        for (int s = 0; s < 3; s++) {
            std::string path = std::string(codes_dir) + "/" + name + "." + suffixes[s];
            if (!file_exists(path)) continue;
            std::string text;
            if (!read_file(path, text)) { set_error(LDPC_ENOTFOUND, "cannot read %s", path.c_str()); delete m; return nullptr; }
            int rc = s == 0 ? parse_q(text, m) : (s == 1 ? parse_alist_reference(text, m) : parse_m(text, m));
            if (rc != LDPC_OK) { delete m; return nullptr; }
            m->source = path;
            return m;
        }
        set_error(LDPC_ENOTFOUND, "can not find any matrix files in \"%s\" under %s", name, codes_dir);
    } catch (...) { set_error(LDPC_ENOMEM, "out of host memory"); }
    delete m;
    return nullptr;
}

// a single file in MacKay's alist order (codes/1920.1280.3.303; the reference's reader would
// transpose it, SURVEY.md section 0 note ii)
ldpc_matrix *ldpc_matrix_load_mackay(const char *path) {
    if (!path) { set_error(LDPC_EINVAL, "null argument"); return nullptr; }
    ldpc_matrix *m = new (std::nothrow) ldpc_matrix();
    if (!m) { set_error(LDPC_ENOMEM, "out of host memory"); return nullptr; }
    try {
        std::string text;
        if (!read_file(path, text)) { set_error(LDPC_ENOTFOUND, "cannot read %s", path); delete m; return nullptr; }
        if (parse_alist_mackay(text, m) != LDPC_OK) { delete m; return nullptr; }
        m->source = path;
        return m;
    } catch (...) { set_error(LDPC_ENOMEM, "out of host memory"); }
    delete m;
    return nullptr;
}

int ldpc_matrix_info(const ldpc_matrix *m, int *rows, int *cols, int *qc_sz, int *block_rows, int *block_cols) {
    if (!m) return set_error(LDPC_EINVAL, "null matrix");
    if (rows) *rows = m->rows;
    if (cols) *cols = m->cols;
    if (qc_sz) *qc_sz = m->sz;
    if (block_rows) *block_rows = m->brows;
    if (block_cols) *block_cols = m->bcols;
    return LDPC_OK;
}

// QuasiCyclic.hs:19-25 toBitMatrix for QC sources; a copy otherwise.  out: rows*cols bytes
int ldpc_matrix_dense(const ldpc_matrix *m, uint8_t *out) {
    if (!m || !out) return set_error(LDPC_EINVAL, "null argument");
    if (m->sz == 0) { memcpy(out, m->dense.data(), m->dense.size()); return LDPC_OK; }
    memset(out, 0, (size_t)m->rows * m->cols);
    for (int br = 0; br < m->brows; br++)
        for (int bc = 0; bc < m->bcols; bc++) {
            const auto &l = m->blocks[(size_t)br * m->bcols + bc];
            for (int k = 0; k < m->sz; k++) {
                if (!limb_bit(l, k)) continue;
                for (int i = 0; i < m->sz; i++) out[(size_t)(br * m->sz + i) * m->cols + bc * m->sz + (i + k) % m->sz] = 1;
            }
        }
    return LDPC_OK;
}

// rank over GF(2) of the expanded matrix (Gaussian elimination on 64-bit packed rows).  A parity-check matrix given
// alone defines a code of cols - rank message bits: codes/1920.1280.A lists 5760 checks of rank 1280 for 1920 bits.
int ldpc_matrix_rank(const ldpc_matrix *m) {
    if (!m) return set_error(LDPC_EINVAL, "null matrix");
    try {
        std::vector<uint8_t> d((size_t)m->rows * m->cols);
        if (ldpc_matrix_dense(m, d.data()) != LDPC_OK) return ldpc_last_error_code();
        const size_t W = ((size_t)m->cols + 63) / 64;
        std::vector<uint64_t> a((size_t)m->rows * W, 0);
        for (int r = 0; r < m->rows; r++)
            for (int c = 0; c < m->cols; c++)
                if (d[(size_t)r * m->cols + c]) a[(size_t)r * W + (c >> 6)] |= 1ull << (c & 63);
        int rank = 0;
        for (int c = 0; c < m->cols && rank < m->rows; c++) {
            const size_t w = (size_t)c >> 6;
            const uint64_t bit = 1ull << (c & 63);
            int piv = -1;
            for (int r = rank; r < m->rows; r++) if (a[(size_t)r * W + w] & bit) { piv = r; break; }
            if (piv < 0) continue;
            if (piv != rank) for (size_t i = 0; i < W; i++) std::swap(a[(size_t)piv * W + i], a[(size_t)rank * W + i]);
            for (int r = rank + 1; r < m->rows; r++)
                if (a[(size_t)r * W + w] & bit) for (size_t i = w; i < W; i++) a[(size_t)r * W + i] ^= a[(size_t)rank * W + i];
            rank++;
        }
        return rank;
    } catch (...) { return set_error(LDPC_ENOMEM, "out of host memory"); }
}

// Fast/Encoder.hs:38-39 `fmap fromIntegral m`: the Integer of each block as sz-bit machine word(s)
int ldpc_matrix_qc_words(const ldpc_matrix *m, uint32_t *out) {
    if (!m || !out) return set_error(LDPC_EINVAL, "null argument");
    if (m->sz == 0) return set_error(LDPC_EUNSUPPORTED, "can not load %s as QuasiCyclic", m->source.c_str());
    const int W = (m->sz + 31) / 32;
    for (size_t i = 0; i < m->blocks.size(); i++)
        for (int w = 0; w < W; w++) {
            uint32_t v = (size_t)w < m->blocks[i].size() ? m->blocks[i][w] : 0u;
            const int rem = m->sz - 32 * w;               // fromIntegral truncates to the word size
            if (rem < 32) v &= (1u << rem) - 1u;
            out[i * W + w] = v;
        }
    return LDPC_OK;
}

// Fast/Arraylet.hs:68-79 initMatrixlet / GPU/CUDA/Arraylet2.hs:299-331: rotation table, -1 = empty;
// a block with more than one circulant is an error there and LDPC_EUNSUPPORTED here.
int ldpc_matrix_qc_offsets(const ldpc_matrix *m, int32_t *offsets) {
    if (!m || !offsets) return set_error(LDPC_EINVAL, "null argument");
    if (m->sz == 0) return set_error(LDPC_EUNSUPPORTED, "can not load %s as QuasiCyclic", m->source.c_str());
    for (size_t i = 0; i < m->blocks.size(); i++) {
        const auto &l = m->blocks[i];
        int top = limb_top(l);
        if (top < 0) { offsets[i] = -1; continue; }
        int pop = 0;
        for (uint32_t w : l) pop += __builtin_popcount(w);
        if (pop != 1) return set_error(LDPC_EUNSUPPORTED, "QuasiCyclic matrix has non-powers of two initial value at block %zu", i);
        offsets[i] = top;
    }
    return LDPC_OK;
}

// the parity-check graph of a loaded matrix: QC table when every block is a single circulant
// (what the QuasiCyclic decoders take), generic CSR otherwise (what the Matrix Bool decoders take)
// the expanded matrix as a generic CSR graph: what a `Matrix Bool` decoder of the reference is given (Orig.hs:30-31)
static ldpc_code *code_from_dense(const ldpc_matrix *m);

ldpc_code *ldpc_code_from_matrix(const ldpc_matrix *m) {
    if (!m) { set_error(LDPC_EINVAL, "null matrix"); return nullptr; }
    try {
        if (m->sz > 0) {
            std::vector<int32_t> off(m->blocks.size());
            if (ldpc_matrix_qc_offsets(m, off.data()) == LDPC_OK) return ldpc_code_create_qc(m->sz, m->brows, m->bcols, off.data());
        }
        return code_from_dense(m);
    } catch (...) { set_error(LDPC_ENOMEM, "out of host memory"); return nullptr; }
}

static ldpc_code *code_from_dense(const ldpc_matrix *m) {
    try {
        std::vector<uint8_t> d((size_t)m->rows * m->cols);
        if (ldpc_matrix_dense(m, d.data()) != LDPC_OK) return nullptr;
        std::vector<int32_t> rp((size_t)m->rows + 1, 0), ci;
        for (int r = 0; r < m->rows; r++) {
            for (int c = 0; c < m->cols; c++)
                if (d[(size_t)r * m->cols + c]) ci.push_back(c);
            rp[(size_t)r + 1] = (int32_t)ci.size();
        }
        if (ci.empty()) ci.push_back(0);
        return ldpc_code_create_csr(m->rows, m->cols, rp.data(), ci.data());
    } catch (...) { set_error(LDPC_ENOMEM, "out of host memory"); return nullptr; }
}

}  // extern "C"

// ------------------------------------------------------------------ the ECC record (mkLDPC)
// one decoder replica of the record: what one `decoder0 h` call of mkLDPC returns (Utils.hs:53 replicateM maxThreadCount)
struct ldpc_ecc_replica {
    int device = -1;
    ldpc_ctx *ctx = nullptr;
    ldpc_sim *sim = nullptr;
    ldpc_batcher *batcher = nullptr;   // set by ldpc_ecc_set_coalescing
    std::mutex mu;                     // two caller threads may map to one replica (Utils.hs:63-69: tid `rem` maxThreadCount)
    std::vector<double> llr_buf;
    std::vector<uint8_t> bits_buf;
};
struct ldpc_ecc {
    std::string name;            // Utils.hs:60
    int message_length = 0;      // Utils.hs:73   m_length = rows G
    int codeword_length = 0;     // Utils.hs:74   c_length = m_length * den / num
    int unpunctured_length = 0;  // cols H
    int max_iters = 0, variant = 0, dtype = 0, rate_num = 0, rate_den = 0;
    ldpc_code *code = nullptr;
    std::vector<std::unique_ptr<ldpc_ecc_replica>> reps;
    int parity_len = 0;
};
// the calling thread's ordinal (first use): stands in for the Haskell ThreadId of Utils.hs:64-65
static int thread_ordinal() {
    static std::atomic<int> next{0};
    static thread_local int mine = -1;
    if (mine < 0) mine = next.fetch_add(1);
    return mine;
}

static std::vector<std::string> split_slash(const std::string &s) {
    std::vector<std::string> out;
    size_t b = 0;
    while (true) {
        size_t e = s.find('/', b);
        out.push_back(s.substr(b, e == std::string::npos ? e : e - b));
        if (e == std::string::npos) break;
        b = e + 1;
    }
    return out;
}
static bool all_digits(const std::string &s) {
    return !s.empty() && std::all_of(s.begin(), s.end(), [](char c) { return c >= '0' && c <= '9'; });
}

extern "C" {

void ldpc_ecc_destroy(ldpc_ecc *e) {
    if (!e) return;
    for (auto &r : e->reps) {
        if (r->batcher) ldpc_batcher_destroy(r->batcher);
        if (r->sim) ldpc_sim_destroy(r->sim);
        if (r->ctx) ldpc_ctx_destroy(r->ctx);
    }
    if (e->code) ldpc_code_destroy(e->code);
    delete e;
}

ldpc_ecc *ldpc_ecc_create(const char *codes_dir, const char *code_name, int max_batch) {
    return ldpc_ecc_create_replicas(codes_dir, code_name, max_batch, 1, nullptr);
}

// Utils.hs:100-108 (the Code factory) + Utils.hs:35-75 (mkLDPC).  Returns NULL with
// LDPC_ENOTFOUND when the name is not one of this library's decoders (the factory's `_ -> return []`).
// n_replicas decoder replicas (Utils.hs:53), replica i on devices[i] (NULL: all on the calling thread's device)
ldpc_ecc *ldpc_ecc_create_replicas(const char *codes_dir, const char *code_name, int max_batch, int n_replicas, const int *devices) {
    if (!codes_dir || !code_name || max_batch <= 0 || n_replicas <= 0 || n_replicas > 1024) { set_error(LDPC_EINVAL, "bad argument"); return nullptr; }
    ldpc_ecc *e = nullptr;
    ldpc_matrix *g = nullptr, *h = nullptr;
    try {
        auto xs = split_slash(code_name);
        // ["ldpc",nm,m,n] | ["ldpc",nm,m,n,x,y]
        if (!((xs.size() == 4 || xs.size() == 6) && xs[0] == "ldpc" && all_digits(xs[3]) &&
              (xs.size() == 4 || (all_digits(xs[4]) && all_digits(xs[5]))))) {
            set_error(LDPC_ENOTFOUND, "'%s' does not match ldpc/<decoder>/<matrix-name>/<max-rounds>[/x/y]", code_name);
            return nullptr;
        }
        std::string dec = xs[1];
        int dtype = LDPC_F32;
        auto ends = [&](const char *suf) { size_t n = strlen(suf); return dec.size() > n && dec.compare(dec.size() - n, n, suf) == 0; };
        if (ends("-f16pk")) { dtype = LDPC_F16PK; dec.resize(dec.size() - 6); }   // packed fp16 arithmetic, two frames per lane (min-sum)
        else if (ends("-f64")) { dtype = LDPC_F64; dec.resize(dec.size() - 4); }
        else if (ends("-f16")) { dtype = LDPC_F16; dec.resize(dec.size() - 4); }
        else if (ends("-f32")) { dec.resize(dec.size() - 4); }
        int variant, schedule = LDPC_SCHED_FLOODING;
        bool as_bool = false;   // "-bool": take H as a plain Boolean matrix (the Haskell binding's `Matrix Bool` flavour, haskell/.../HIP.hs)
        if (ends("-bool")) { as_bool = true; dec.resize(dec.size() - 5); }
        if (ends("-layered")) { schedule = LDPC_SCHED_LAYERED; dec.resize(dec.size() - 8); }   // extension: row-layered schedule
        // The reference's own decoder names are accepted as aliases of the decoder that computes the same thing here, so
        // that a command line written for the reference runs unchanged (the ECC keeps the name it was asked for):
        //   reference, sparse (Reference/Orig.hs:21, Sparse.hs:39) and min, sparsemin (Min.hs:24, SparseMin.hs:42) take H
        //   as a Boolean matrix; arraylet, arraylet-min, arraylet-cm (Fast/Arraylet.hs:138, ArrayletMin.hs:136,
        //   CachedMult.hs:207) and the CUDA plug-ins (GPU/CUDA/Arraylet1.hs:60, Arraylet2.hs:61, TwoArrays.hs:61,
        //   CachedMult.hs:66) take it quasi-cyclic.  arraylet-cm's last-ulp numerics exist in f64 only (LDPC_TANH_CM).
        // In f64 (the reference's own type: `ldpc/arraylet-f64/...`) an alias also selects ITS decoder's column-sum order
        // (ldpc_sum_order), so that the trajectory is that decoder's bit for bit; in f32 the on-chip kernels run (reference order,
        // within 1e-5 of any of them).
        static const struct { const char *ref, *hip; bool as_bool; int dtype; int sum_order; } kAliases[] = {
            {"reference", "hip-tanh", true, -1, LDPC_SUM_REFERENCE},      {"sparse", "hip-tanh", true, -1, LDPC_SUM_SPARSE},
            {"min", "hip-minsum", true, -1, LDPC_SUM_REFERENCE},          {"sparsemin", "hip-minsum", true, -1, LDPC_SUM_SPARSE},
            {"arraylet", "hip-tanh", false, -1, LDPC_SUM_ARRAYLET},      {"arraylet-min", "hip-minsum", false, -1, LDPC_SUM_ARRAYLET},
            {"arraylet-cm", "hip-tanh-cm", false, LDPC_F64, LDPC_SUM_REFERENCE},
            // cuda-arraylet2 (the live GPU decoder) selects ITS arithmetic -- float tanh factors, double product, float atanh_ and clamp,
            // ascending float column sums (LDPC_TANH_CUDA32, a parity mode on the flood path; `hip-tanh` is the fast decoder)
            {"cuda-arraylet1", "hip-tanh", false, -1, LDPC_SUM_REFERENCE}, {"cuda-arraylet2", "hip-tanh-cuda32", false, LDPC_F32, LDPC_SUM_REFERENCE},
            {"two-arrays", "hip-tanh", false, -1, LDPC_SUM_REFERENCE},     {"cuda-arraylet-cm", "hip-tanh", false, -1, LDPC_SUM_REFERENCE},
        };
        int sum_order = LDPC_SUM_REFERENCE;
        for (const auto &a : kAliases)
            if (dec == a.ref) { dec = a.hip; as_bool = as_bool || a.as_bool; if (a.dtype >= 0) dtype = a.dtype; if (dtype == LDPC_F64) sum_order = a.sum_order; break; }
        if (dec == "hip-tanh") variant = LDPC_TANH;
        else if (dec == "hip-tanh-cm") variant = LDPC_TANH_CM;   // the reference's `arraylet-cm` numerics (f64 parity mode)
        else if (dec == "hip-tanh-cuda32") variant = LDPC_TANH_CUDA32;   // the reference's `cuda-arraylet2` numerics (f32 parity mode)
        else if (dec == "hip-minsum") variant = LDPC_MINSUM;
        else { set_error(LDPC_ENOTFOUND, "decoder '%s' is not provided by libldpc_hip (hip-tanh, hip-minsum [-layered][-bool][-f32|-f64|-f16], or a reference name: reference, min, sparse, sparsemin, arraylet, arraylet-min, arraylet-cm, cuda-arraylet1/2)", xs[1].c_str()); return nullptr; }

        // A matrix name that is a plain FILE under codes_dir is a stand-alone parity-check matrix in
        // MacKay's alist order with no generator (codes/1920.1280.3.303; the reference cannot load it,
        // SURVEY.md section 0 note ii): message length = cols - rows (H full rank), frames are the
        // all-zero codeword.  Otherwise the reference's <matrix>/G + <matrix>/H pair.
        bool standalone = file_exists(std::string(codes_dir) + "/" + xs[2]);
        bool have_g = false;
        for (const char *suf : {"q", "alist", "m"}) have_g = have_g || file_exists(std::string(codes_dir) + "/" + xs[2] + "/G." + suf);
        if (standalone) {
            h = ldpc_matrix_load_mackay((std::string(codes_dir) + "/" + xs[2]).c_str());
            if (!h) return nullptr;
        } else if (!have_g) {
            // a matrix directory that ships H only (codes/dvbs2like.64800.1.2: a dense generator would be 131 MB):
            // same convention as the stand-alone file -- k = cols - rows, all-zero codewords, no encoder
            h = ldpc_matrix_load(codes_dir, (xs[2] + "/H").c_str());
            if (!h) return nullptr;
            standalone = true;
        } else {
            g = ldpc_matrix_load(codes_dir, (xs[2] + "/G").c_str());  // Utils.hs:36
            if (!g) return nullptr;
            h = ldpc_matrix_load(codes_dir, (xs[2] + "/H").c_str());  // Utils.hs:40
            if (!h) { ldpc_matrix_destroy(g); return nullptr; }
            if (g->rows + g->cols != h->cols) {                      // Utils.hs:43
                set_error(LDPC_EFORMAT, "bad code size match (%d,%d)", g->rows + g->cols, h->cols);
                ldpc_matrix_destroy(g); ldpc_matrix_destroy(h);
                return nullptr;
            }
        }
        e = new ldpc_ecc();
        // no generator: k = cols - rank(H); rows < cols is taken as full rank (every shipped H of that shape is), a matrix with
        // at least as many rows as columns (codes/1920.1280.A: redundant checks) has its rank computed
        int k_alone = standalone ? h->cols - h->rows : 0;
        if (standalone && h->rows >= h->cols) {
            const int rk = ldpc_matrix_rank(h);
            if (rk < 0) goto fail;
            k_alone = h->cols - rk;
            if (k_alone <= 0) { set_error(LDPC_EFORMAT, "%s has rank %d over GF(2): no message bits", xs[2].c_str(), rk); goto fail; }
        }
        e->message_length = standalone ? k_alone : g->rows;
        e->unpunctured_length = h->cols;
        e->max_iters = atoi(xs[3].c_str());
        e->variant = variant; e->dtype = dtype;
        if (xs.size() == 6) { e->rate_num = atoi(xs[4].c_str()); e->rate_den = atoi(xs[5].c_str()); }
        else { e->rate_num = e->message_length; e->rate_den = e->unpunctured_length; }  // Utils.hs:46-48
        if (e->rate_num <= 0 || e->rate_den <= 0) { set_error(LDPC_EINVAL, "bad rate %d/%d", e->rate_num, e->rate_den); goto fail; }
        {   // Ratio Int normalises: numerator/denominator of the reduced fraction (Utils.hs:50,60)
            int a = e->rate_num, b = e->rate_den;
            while (b) { int t = a % b; a = b; b = t; }
            e->rate_num /= a; e->rate_den /= a;
        }
        e->codeword_length = (int)(((long long)e->message_length * e->rate_den) / e->rate_num);  // Utils.hs:50
        if (e->codeword_length < e->message_length || e->codeword_length > e->unpunctured_length) {
            set_error(LDPC_EINVAL, "rate %d/%d gives codeword length %d outside [%d,%d]", e->rate_num, e->rate_den,
                      e->codeword_length, e->message_length, e->unpunctured_length);
            goto fail;
        }
        e->name = "ldpc/" + xs[1] + "/" + xs[2] + "/" + std::to_string(e->max_iters) + "/" + std::to_string(e->rate_num) + "/" +
                  std::to_string(e->rate_den);  // Utils.hs:60
        e->code = as_bool ? code_from_dense(h) : ldpc_code_from_matrix(h);
        if (!e->code) goto fail;
        {   // Utils.hs:53 replicateM maxThreadCount.  LDPC_HIP_PATH=flood|fused overrides the automatic kernel choice.
            int path = LDPC_PATH_AUTO;
            const char *pe = getenv("LDPC_HIP_PATH");
            if (pe && !strcmp(pe, "flood")) path = LDPC_PATH_FLOOD;
            else if (pe && !strcmp(pe, "fused")) path = LDPC_PATH_FUSED;
            // encoder: a quasi-cyclic G of a word size the reference's fast encoder takes (Fast/Encoder.hs:28-33) is encoded
            // from its circulants (rotate-and-xor, sim.hip); anything else from the expanded matrix (Orig.hs:25-26).
            // LDPC_SIM_ENCODER=dense forces the expanded form (A/B measurements, tests).
            std::vector<uint8_t> gd;
            std::vector<uint32_t> gq;
            const char *enc_env = getenv("LDPC_SIM_ENCODER");
            const bool g_qc = g && g->sz > 0 && (g->sz == 32 || g->sz == 64 || g->sz == 128 || g->sz == 256) && g->rows == e->message_length &&
                              !(enc_env && !strcmp(enc_env, "dense"));
            if (g) {
                e->parity_len = g->cols;
                if (g_qc) {
                    gq.resize((size_t)g->brows * g->bcols * (g->sz / 32));
                    if (ldpc_matrix_qc_words(g, gq.data()) != LDPC_OK) goto fail;
                } else {
                    gd.resize((size_t)g->rows * g->cols);
                    if (ldpc_matrix_dense(g, gd.data()) != LDPC_OK) goto fail;
                }
            }
            for (int i = 0; i < n_replicas; i++) {
                std::unique_ptr<ldpc_ecc_replica> r(new ldpc_ecc_replica());
                r->device = devices ? devices[i] : ldpc_current_device();
                ldpc_ctx_config cfg{};
                cfg.struct_size = sizeof(cfg); cfg.device = devices ? devices[i] : -1; cfg.variant = variant; cfg.dtype = dtype; cfg.max_batch = max_batch;
                cfg.path = (sum_order != LDPC_SUM_REFERENCE) ? LDPC_PATH_FLOOD : path; cfg.schedule = schedule; cfg.sum_order = sum_order;
                r->ctx = ldpc_ctx_create_cfg(e->code, &cfg);
                ldpc_ecc_replica *rp = r.get();
                e->reps.push_back(std::move(r));      // owned by the record from here on (ldpc_ecc_destroy frees it)
                if (!rp->ctx) goto fail;
                rp->device = ldpc_ctx_device(rp->ctx);
                rp->sim = g_qc ? ldpc_sim_create_qc_on(e->code, rp->device, e->message_length, e->codeword_length, g->sz, g->brows, g->bcols, gq.data(), max_batch)
                               : ldpc_sim_create_on(e->code, rp->device, e->message_length, e->codeword_length, g ? g->cols : 0, g ? gd.data() : nullptr, max_batch);
                if (!rp->sim) goto fail;
                rp->llr_buf.assign((size_t)e->unpunctured_length, 0.0);
                rp->bits_buf.assign((size_t)e->unpunctured_length, 0);
            }
        }
        if (g) ldpc_matrix_destroy(g);
        ldpc_matrix_destroy(h);
        return e;
    } catch (...) { set_error(LDPC_ENOMEM, "out of host memory"); }
fail:
    if (g) ldpc_matrix_destroy(g);
    if (h) ldpc_matrix_destroy(h);
    ldpc_ecc_destroy(e);
    return nullptr;
}

const char *ldpc_ecc_name(const ldpc_ecc *e) { return e ? e->name.c_str() : ""; }
int ldpc_ecc_message_length(const ldpc_ecc *e) { return e ? e->message_length : set_error(LDPC_EINVAL, "null ecc"); }
int ldpc_ecc_codeword_length(const ldpc_ecc *e) { return e ? e->codeword_length : set_error(LDPC_EINVAL, "null ecc"); }
int ldpc_ecc_unpunctured_length(const ldpc_ecc *e) { return e ? e->unpunctured_length : set_error(LDPC_EINVAL, "null ecc"); }
int ldpc_ecc_max_iters(const ldpc_ecc *e) { return e ? e->max_iters : set_error(LDPC_EINVAL, "null ecc"); }
ldpc_ctx *ldpc_ecc_ctx(ldpc_ecc *e) { return (e && !e->reps.empty()) ? e->reps[0]->ctx : nullptr; }
ldpc_sim *ldpc_ecc_sim(ldpc_ecc *e) { return (e && !e->reps.empty()) ? e->reps[0]->sim : nullptr; }
int ldpc_ecc_replicas(const ldpc_ecc *e) { return e ? (int)e->reps.size() : set_error(LDPC_EINVAL, "null ecc"); }
ldpc_ctx *ldpc_ecc_ctx_at(ldpc_ecc *e, int i) { return (e && i >= 0 && i < (int)e->reps.size()) ? e->reps[i]->ctx : nullptr; }
ldpc_sim *ldpc_ecc_sim_at(ldpc_ecc *e, int i) { return (e && i >= 0 && i < (int)e->reps.size()) ? e->reps[i]->sim : nullptr; }

// Coalescing of the per-frame calls (batcher.cc): every replica gets a batcher; ldpc_ecc_decode then queues its frame
// there and up to max_frames concurrent callers share one launch.
int ldpc_ecc_coalescing_stats(ldpc_ecc *e, long *calls, long *launches) {
    if (!e) return set_error(LDPC_EINVAL, "null ecc");
    long c = 0, l = 0;
    for (auto &r : e->reps)
        if (r->batcher) { long cc = 0, ll = 0; (void)ldpc_batcher_stats(r->batcher, &cc, &ll); c += cc; l += ll; }
    if (calls) *calls = c;
    if (launches) *launches = l;
    return LDPC_OK;
}

int ldpc_ecc_set_coalescing(ldpc_ecc *e, int max_frames, int max_wait_us) {
    if (!e) return set_error(LDPC_EINVAL, "null ecc");
    for (auto &r : e->reps) {
        std::lock_guard<std::mutex> lk(r->mu);
        if (r->batcher) { ldpc_batcher_destroy(r->batcher); r->batcher = nullptr; }
        if (max_frames > 0) {
            r->batcher = ldpc_batcher_create(r->ctx, max_frames, max_wait_us);
            if (!r->batcher) return ldpc_last_error_code();
        }
    }
    return LDPC_OK;
}
const ldpc_code *ldpc_ecc_code(const ldpc_ecc *e) { return e ? e->code : nullptr; }

// Utils.hs:61  encode = \inp -> inp ++ take (c_length - m_length) (encoder' inp)
int ldpc_ecc_encode(const ldpc_ecc *e, const uint8_t *msg, uint8_t *codeword) {
    if (!e || !msg || !codeword) return set_error(LDPC_EINVAL, "null argument");
    try {
        if (e->parity_len == 0 && e->codeword_length > e->message_length)
            return set_error(LDPC_EUNSUPPORTED, "this code was loaded without a generator matrix: no encoder");
        std::vector<uint8_t> par((size_t)e->parity_len + 1);
        int rc = ldpc_sim_encode_host(e->reps[0]->sim, msg, par.data());
        if (rc != LDPC_OK) return rc;
        memcpy(codeword, msg, (size_t)e->message_length);
        memcpy(codeword + e->message_length, par.data(), (size_t)(e->codeword_length - e->message_length));
        return LDPC_OK;
    } catch (...) { return set_error(LDPC_ENOMEM, "out of host memory"); }
}

// Utils.hs:62-72  decode: unpuncture (take c_length ++ zeros), run the replica, take m_length bits;
// `Nothing` (a failing replica) -> hard decisions of the input, flag False.
int ldpc_ecc_decode(ldpc_ecc *e, const double *llr, uint8_t *msg_bits, int *ok) {
    if (!e || e->reps.empty()) return set_error(LDPC_EINVAL, "null argument");
    // Utils.hs:63-69: the replica is picked by the calling thread (tid `rem` maxThreadCount)
    return ldpc_ecc_decode_on(e, thread_ordinal() % (int)e->reps.size(), llr, msg_bits, ok);
}

int ldpc_ecc_decode_on(ldpc_ecc *e, int replica, const double *llr, uint8_t *msg_bits, int *ok) {
    if (!e || !llr || !msg_bits || replica < 0 || replica >= (int)e->reps.size()) return set_error(LDPC_EINVAL, "bad argument");
    ldpc_ecc_replica &r = *e->reps[replica];
    int iters = 0, conv = 0, rc;
    if (r.batcher) {   // coalesced: the frame is queued; up to max_frames callers share the launch (no replica lock held)
        static thread_local std::vector<double> tl_llr;
        static thread_local std::vector<uint8_t> tl_bits;
        tl_llr.assign((size_t)e->unpunctured_length, 0.0);                       // Utils.hs:55 unpuncture: zeros appended
        std::copy(llr, llr + e->codeword_length, tl_llr.begin());
        tl_bits.resize((size_t)e->unpunctured_length);
        rc = ldpc_batcher_decode_one(r.batcher, e->max_iters, tl_llr.data(), tl_bits.data(), &iters, &conv);
        if (rc == LDPC_OK) memcpy(msg_bits, tl_bits.data(), (size_t)e->message_length);   // Utils.hs:72
    } else {
        std::lock_guard<std::mutex> lk(r.mu);
        std::copy(llr, llr + e->codeword_length, r.llr_buf.begin());
        std::fill(r.llr_buf.begin() + e->codeword_length, r.llr_buf.end(), 0.0);  // Utils.hs:55
        rc = ldpc_decode_one(r.ctx, e->max_iters, r.llr_buf.data(), r.bits_buf.data(), &iters, &conv);
        if (rc == LDPC_OK) memcpy(msg_bits, r.bits_buf.data(), (size_t)e->message_length);  // Utils.hs:72
    }
    if (rc != LDPC_OK) {  // Utils.hs:71
        for (int i = 0; i < e->message_length; i++) msg_bits[i] = llr[i] > 0.0 ? 1 : 0;
        if (ok) *ok = 0;
        return LDPC_OK;
    }
    if (ok) *ok = 1;
    return LDPC_OK;
}

}  // extern "C"
