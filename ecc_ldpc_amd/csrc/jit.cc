// jit.cc -- run-time specialisation of the fused quasi-cyclic kernels (see jit.h): the f32 split kernel and, since r03, the
// packed-fp16 and the two on-chip layered kernels (JitKind).
//   source   : generated plan + rotation table + one extern "C" kernel around the body (fused_split_body.h, fused_pk16_body.h,
//              fused_layered_body.h);
//              the device headers are embedded in the library at build time (jit_embed.inc, build.py)
//   compiler : `hipcc --genco` in a child process when the tool chain is installed, else hiprtc in-process (dlopen'ed:
//              no link-time dependency); LDPC_JIT_COMPILER=hipcc|hiprtc forces one.  See jit_compile_cached for why.
//   cache    : <cache dir>/<kernel>-<hash of source + options>[.rtc].hsaco, written atomically; LDPC_JIT_CACHE names the
//              directory (default: jit_cache/ next to libldpc_hip.so if writable, else ~/.cache/ecc_ldpc_amd, else a
//              0700 directory of this user under /tmp); builds with LDPC_JIT_EXTRA_OPTS or LDPC_JIT_NOCACHE=1 bypass it
#include "jit.h"

#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <hip/hiprtc.h>
#include <spawn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>

extern char **environ;

#include "fused_common.h"

namespace ldpc {

namespace {
struct EmbeddedHeader { const char *name; const char *text; };
#include "jit_embed.inc"   // static const EmbeddedHeader kJitHeaders[]; static const int kJitHeaderCount;

const char *const kOptions[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-w"};
constexpr int kNumOptions = sizeof(kOptions) / sizeof(kOptions[0]);

// ---------------------------------------------------------------------------------------------- hiprtc, loaded lazily
struct Rtc {
    void *h = nullptr;
    hiprtcResult (*create)(hiprtcProgram *, const char *, const char *, int, const char **, const char **) = nullptr;
    hiprtcResult (*compile)(hiprtcProgram, int, const char **) = nullptr;
    hiprtcResult (*log_size)(hiprtcProgram, size_t *) = nullptr;
    hiprtcResult (*log)(hiprtcProgram, char *) = nullptr;
    hiprtcResult (*code_size)(hiprtcProgram, size_t *) = nullptr;
    hiprtcResult (*code)(hiprtcProgram, char *) = nullptr;
    hiprtcResult (*destroy)(hiprtcProgram *) = nullptr;
    const char *(*err)(hiprtcResult) = nullptr;
    std::string why;
};

Rtc &rtc() {
    static Rtc r;
    static std::once_flag once;
    std::call_once(once, [] {
        std::vector<std::string> cand;
        if (const char *e = getenv("LDPC_HIPRTC_LIB")) cand.push_back(e);
        Dl_info di;   // the directory of the HIP runtime this process already runs on comes first
        if (dladdr((void *)&hipGetDeviceCount, &di) && di.dli_fname) {
            std::string d(di.dli_fname);
            size_t p = d.rfind('/');
            if (p != std::string::npos) { cand.push_back(d.substr(0, p) + "/libhiprtc.so.7"); cand.push_back(d.substr(0, p) + "/libhiprtc.so"); }
        }
        cand.push_back("libhiprtc.so.7");
        cand.push_back("libhiprtc.so");
        cand.push_back("/opt/rocm/lib/libhiprtc.so");
        for (auto &c : cand) {
            r.h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (r.h) break;
            const char *de = dlerror();   // (a second call would return NULL: the first clears it)
            r.why = de ? de : "dlopen failed";
        }
        if (!r.h) return;
#define SYM(field, name) r.field = (decltype(r.field))dlsym(r.h, name)
        SYM(create, "hiprtcCreateProgram"); SYM(compile, "hiprtcCompileProgram"); SYM(log_size, "hiprtcGetProgramLogSize");
        SYM(log, "hiprtcGetProgramLog"); SYM(code_size, "hiprtcGetCodeSize"); SYM(code, "hiprtcGetCode");
        SYM(destroy, "hiprtcDestroyProgram"); SYM(err, "hiprtcGetErrorString");
#undef SYM
        if (!r.create || !r.compile || !r.log_size || !r.log || !r.code_size || !r.code || !r.destroy) { r.why = "libhiprtc lacks a symbol"; dlclose(r.h); r.h = nullptr; }
    });
    return r;
}

// ---------------------------------------------------------------------------------------------- cache
uint64_t fnv1a(const std::string &s, uint64_t seed) {
    uint64_t h = 0xcbf29ce484222325ull ^ seed;
    for (unsigned char c : s) { h ^= c; h *= 0x100000001b3ull; }
    return h;
}
std::string hash_hex(const std::string &s) {
    char b[40];
    snprintf(b, sizeof(b), "%016llx%016llx", (unsigned long long)fnv1a(s, 0), (unsigned long long)fnv1a(s, 0x9e3779b97f4a7c15ull));
    return b;
}
bool dir_writable(const std::string &d) {
    if (mkdir(d.c_str(), 0755) != 0 && access(d.c_str(), F_OK) != 0) return false;
    return access(d.c_str(), W_OK | X_OK) == 0;
}
std::string g_cache_dir;
std::once_flag g_cache_once;
void pick_cache_dir() {
    std::vector<std::string> cand;
    if (const char *e = getenv("LDPC_JIT_CACHE")) cand.push_back(e);
    Dl_info di;
    if (dladdr((void *)&pick_cache_dir, &di) && di.dli_fname) {
        std::string d(di.dli_fname);
        size_t p = d.rfind('/');
        cand.push_back((p == std::string::npos ? std::string(".") : d.substr(0, p)) + "/jit_cache");
    }
    if (const char *h = getenv("HOME")) { std::string c = std::string(h) + "/.cache"; (void)mkdir(c.c_str(), 0755); cand.push_back(c + "/ecc_ldpc_amd"); }
    for (auto &c : cand)
        if (dir_writable(c)) { g_cache_dir = c; return; }
    // last resort, a world-writable parent: the directory is trusted only if it is OURS and nobody else can write to it
    // (code objects found there are loaded and run)
    const std::string t = "/tmp/ecc_ldpc_amd_jit_" + std::to_string((unsigned)getuid());
    (void)mkdir(t.c_str(), 0700);
    struct stat st;
    if (lstat(t.c_str(), &st) == 0 && S_ISDIR(st.st_mode) && st.st_uid == getuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0 &&
        access(t.c_str(), W_OK | X_OK) == 0) { g_cache_dir = t; return; }
    g_cache_dir = "";
}

bool read_all(const std::string &path, std::vector<char> &out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end);
    std::streamoff n = f.tellg();
    if (n <= 0) return false;
    out.resize((size_t)n);
    f.seekg(0);
    f.read(out.data(), n);
    return (bool)f;
}

int compile_hiprtc(const std::string &source, std::vector<char> &co) {
    Rtc &r = rtc();
    if (!r.h) return set_error(LDPC_EUNSUPPORTED, "run-time compilation unavailable: libhiprtc not loadable (%s)", r.why.c_str());
    std::vector<const char *> names, texts;
    for (int i = 0; i < kJitHeaderCount; i++) { names.push_back(kJitHeaders[i].name); texts.push_back(kJitHeaders[i].text); }
    hiprtcProgram prog = nullptr;
    hiprtcResult rc = r.create(&prog, source.c_str(), "ldpc_jit.hip", (int)names.size(), texts.data(), names.data());
    if (rc != HIPRTC_SUCCESS) return set_error(LDPC_EHIP, "hiprtcCreateProgram: %s", r.err ? r.err(rc) : "error");
    std::vector<const char *> opts(kOptions, kOptions + kNumOptions);
    std::vector<std::string> extra;   // LDPC_JIT_EXTRA_OPTS: experiments (space separated; not part of the cache key)
    if (const char *x = getenv("LDPC_JIT_EXTRA_OPTS")) { std::istringstream is(x); std::string t; while (is >> t) extra.push_back(t); }
    for (auto &t : extra) opts.push_back(t.c_str());
    rc = r.compile(prog, (int)opts.size(), opts.data());
    if (rc != HIPRTC_SUCCESS) {
        size_t n = 0;
        std::string log;
        if (r.log_size(prog, &n) == HIPRTC_SUCCESS && n > 1) { log.resize(n); r.log(prog, &log[0]); }
        if (log.size() > 380) log = log.substr(0, 380);
        r.destroy(&prog);
        return set_error(LDPC_EHIP, "hiprtcCompileProgram: %s: %s", r.err ? r.err(rc) : "error", log.c_str());
    }
    size_t n = 0;
    rc = r.code_size(prog, &n);
    if (rc == HIPRTC_SUCCESS && n > 0) { co.resize(n); rc = r.code(prog, co.data()); }
    r.destroy(&prog);
    if (rc != HIPRTC_SUCCESS || n == 0) return set_error(LDPC_EHIP, "hiprtcGetCode failed");
    return LDPC_OK;
}

// the tool-chain route: same source, same headers, same options through `hipcc --genco`
int compile_hipcc(const std::string &source, std::vector<char> &co) {
    const char *tmp = getenv("TMPDIR");
    std::string tmpl = std::string(tmp && *tmp ? tmp : "/tmp") + "/ldpc_jit_XXXXXX";
    if (!mkdtemp(&tmpl[0])) return set_error(LDPC_EHIP, "mkdtemp(%s) failed", tmpl.c_str());
    std::string d(tmpl);
    for (int i = 0; i < kJitHeaderCount; i++) { std::ofstream f(d + "/" + kJitHeaders[i].name); f << kJitHeaders[i].text; }
    { std::ofstream f(d + "/ldpc_jit.hip"); f << source; }
    const char *hipcc_env = getenv("HIPCC");
    const std::string hipcc = hipcc_env ? hipcc_env : "/opt/rocm/bin/hipcc";
    std::vector<std::string> av = {hipcc, "--genco", "-DLDPC_JIT", "-I" + d};
    for (int i = 0; i < kNumOptions; i++) av.push_back(kOptions[i]);
    if (const char *x = getenv("LDPC_JIT_EXTRA_OPTS")) { std::istringstream is(x); std::string t; while (is >> t) av.push_back(t); }   // experiments; such builds are never cached
    av.insert(av.end(), {"-x", "hip", d + "/ldpc_jit.hip", "-o", d + "/out.hsaco"});
    // The tool chain runs as a CHILD with a scrubbed environment: a profiler's preload (rocprofv3 sets LD_PRELOAD /
    // HSA_TOOLS_LIB / ROCP_*) would otherwise initialise the GPU inside hipcc, which then exec's clang -- the
    // exec-after-GPU-init chain this pool forbids.  Nothing of this process is replaced.
    std::vector<std::string> envs;
    for (char **e = environ; e && *e; e++) {
        const char *v = *e;
        static const char *const drop[] = {"LD_PRELOAD=", "HSA_TOOLS_LIB=", "HSA_TOOLS_REPORT_LOAD_FAILURE=", "ROCP_", "ROCPROF", "ROCTRACER_", "ROCTX_", "HIP_TOOLS_LIB="};
        bool skip = false;
        for (const char *p : drop) if (!strncmp(v, p, strlen(p))) skip = true;
        if (!skip) envs.push_back(v);
    }
    std::vector<char *> argv_c, env_c;
    for (auto &a : av) argv_c.push_back(&a[0]);
    argv_c.push_back(nullptr);
    for (auto &e : envs) env_c.push_back(&e[0]);
    env_c.push_back(nullptr);
    const std::string logp = d + "/log.txt";
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_addopen(&fa, 0, "/dev/null", O_RDONLY, 0);
    posix_spawn_file_actions_addopen(&fa, 1, logp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
    posix_spawn_file_actions_adddup2(&fa, 1, 2);
    pid_t pid = 0;
    int rc = posix_spawn(&pid, hipcc.c_str(), &fa, nullptr, argv_c.data(), env_c.data());
    posix_spawn_file_actions_destroy(&fa);
    if (rc == 0) {
        int status = 0;
        while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {}
        rc = WIFEXITED(status) ? WEXITSTATUS(status) : 128 + (WIFSIGNALED(status) ? WTERMSIG(status) : 0);
    } else rc = -rc;
    bool ok = rc == 0 && read_all(d + "/out.hsaco", co);
    std::string log;
    if (!ok) { std::vector<char> l; if (read_all(d + "/log.txt", l)) log.assign(l.begin(), l.begin() + std::min<size_t>(l.size(), 380)); }
    for (int i = 0; i < kJitHeaderCount; i++) (void)unlink((d + "/" + kJitHeaders[i].name).c_str());
    for (const char *f : {"/ldpc_jit.hip", "/out.hsaco", "/log.txt"}) (void)unlink((d + f).c_str());
    (void)rmdir(d.c_str());
    if (!ok) return set_error(LDPC_EHIP, "hipcc --genco failed (%d): %s", rc, log.c_str());
    return LDPC_OK;
}
// `.vgpr_spill_count` of the (single) kernel in a code object, read from its msgpack metadata note; -1 if not found
int spill_count(const std::vector<char> &co) {
    static const char key[] = ".vgpr_spill_count";
    const size_t kl = sizeof(key) - 1;
    for (size_t i = 0; i + kl + 5 <= co.size(); i++) {
        if (memcmp(&co[i], key, kl) != 0) continue;
        const unsigned char *p = (const unsigned char *)&co[i + kl];
        if (p[0] <= 0x7f) return p[0];
        if (p[0] == 0xcc) return p[1];
        if (p[0] == 0xcd) return (p[1] << 8) | p[2];
        if (p[0] == 0xce) return (int)(((unsigned)p[1] << 24) | (p[2] << 16) | (p[3] << 8) | p[4]);
        return -1;
    }
    return -1;
}
}  // namespace

const char *jit_cache_dir() {
    std::call_once(g_cache_once, pick_cache_dir);
    return g_cache_dir.c_str();
}

namespace {
std::mutex g_key_mutexes_lock;
std::map<std::string, std::shared_ptr<std::mutex>> g_key_mutexes;   // one compilation per key per process
std::atomic<unsigned> g_tmp_counter{0};
}  // namespace

int jit_compile_cached(const std::string &source, const std::string &kernel_name, std::vector<char> &co, bool *from_cache, double *seconds) {
    std::string keyed = source;
    for (int i = 0; i < kNumOptions; i++) { keyed += "\n//opt "; keyed += kOptions[i]; }
    for (int i = 0; i < kJitHeaderCount; i++) { keyed += "\n//hdr "; keyed += kJitHeaders[i].name; keyed += hash_hex(kJitHeaders[i].text); }
    const std::string key = hash_hex(keyed);
    const std::string dir = jit_cache_dir();
    // Two routes to the same code object.  The tool chain (`hipcc --genco`, a child process) comes first when it is
    // installed: it is the compiler the built-in instances were built and tuned with, whereas the compiler behind
    // hiprtc is whatever libamd_comgr the PROCESS has loaded -- inside Python that is the copy bundled with the torch
    // wheel, an older LLVM than /opt/rocm's (measured on the jpl.4096-shaped kernel: 571 spilled VGPRs against 29,
    // 4.8x slower).  Without a tool chain (run-time-only ROCm installs) hiprtc compiles in-process.
    // LDPC_JIT_COMPILER=hipcc|hiprtc forces one route.  The route is part of the cache file's name, so an object from
    // the slower compiler never stands in for the tool chain's once that is available.
    const char *which = getenv("LDPC_JIT_COMPILER");
    const bool only_hipcc = which && !strcmp(which, "hipcc"), only_hiprtc = which && !strcmp(which, "hiprtc");
    const char *hipcc_path = getenv("HIPCC") ? getenv("HIPCC") : "/opt/rocm/bin/hipcc";
    const bool have_hipcc = access(hipcc_path, X_OK) == 0;
    const bool want_hipcc = !only_hiprtc && (only_hipcc || have_hipcc);
    const std::string stem = dir.empty() ? std::string() : dir + "/" + kernel_name + "-" + key;
    const std::string path_cc = stem.empty() ? stem : stem + ".hsaco", path_rtc = stem.empty() ? stem : stem + ".rtc.hsaco";
    if (from_cache) *from_cache = false;
    if (seconds) *seconds = 0;
    const char *nc = getenv("LDPC_JIT_NOCACHE");
    const char *xo = getenv("LDPC_JIT_EXTRA_OPTS");
    // experimental builds (extra options) and LDPC_JIT_NOCACHE=1 neither read nor WRITE the cache
    const bool use_cache = !stem.empty() && !(nc && !strcmp(nc, "1")) && !(xo && *xo);
    std::shared_ptr<std::mutex> km;
    {
        std::lock_guard<std::mutex> g(g_key_mutexes_lock);
        auto &slot = g_key_mutexes[kernel_name + key];
        if (!slot) slot = std::make_shared<std::mutex>();
        km = slot;
    }
    std::lock_guard<std::mutex> one_at_a_time(*km);   // replicas created from several threads: the first compiles, the rest read
    if (use_cache) {
        // a tool-chain object is always acceptable; an in-process one only when the tool chain is not the route wanted
        if (!only_hiprtc && read_all(path_cc, co)) { if (from_cache) *from_cache = true; return LDPC_OK; }
        if (!want_hipcc && read_all(path_rtc, co)) { if (from_cache) *from_cache = true; return LDPC_OK; }
    }
    auto t0 = std::chrono::steady_clock::now();
    int rc;
    bool by_hipcc = want_hipcc;
    if (only_hiprtc) rc = compile_hiprtc(source, co);
    else if (only_hipcc) rc = compile_hipcc(source, co);
    else {
        rc = have_hipcc ? compile_hipcc(source, co) : compile_hiprtc(source, co);
        if (rc != LDPC_OK && have_hipcc) {   // tool chain present but failing: the in-process compiler
            std::string first = ldpc_last_error();
            by_hipcc = false;
            if (compile_hiprtc(source, co) != LDPC_OK) rc = set_error(LDPC_EHIP, "%s", first.c_str()); else rc = LDPC_OK;
        }
    }
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc != LDPC_OK) return rc;
    if (use_cache) {   // atomically: concurrent ranks (processes) may compile the same kernel; the temp name is unique per writer
        const std::string &path = by_hipcc ? path_cc : path_rtc;
        std::string tmp = path + ".tmp" + std::to_string((long)getpid()) + "." + std::to_string(g_tmp_counter.fetch_add(1));
        int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL, 0644);
        if (fd >= 0) {
            size_t off = 0;
            bool ok = true;
            while (off < co.size()) {
                ssize_t w = write(fd, co.data() + off, co.size() - off);
                if (w <= 0) { ok = false; break; }
                off += (size_t)w;
            }
            ok = (close(fd) == 0) && ok;
            if (!ok || rename(tmp.c_str(), path.c_str()) != 0) (void)unlink(tmp.c_str());
        }
    }
    return LDPC_OK;
}

// ---------------------------------------------------------------------------------------------- plan generation
namespace {
struct PlanChoice {
    int sz = 0, nbr = 0, nbc = 0, nedge = 0, dmax = 0, np = 1, waves_per_eu = 4, cpw = 1, v = 64;
    std::vector<int> deg, ebeg, own, rot, bc;
};

constexpr int kMsgCapMinsum = 80, kMsgCapTanh = 78;   // messages per thread that still fit the register budget (4 / 3 waves per SIMD)

// cost of dealing the block rows to `np` wave groups by `own`: (max messages per group, phase-B critical path + phase-A critical path)
void deal_cost(const PlanChoice &p, const std::vector<int> &own, int np, int *max_msgs, long *path) {
    std::vector<int> msgs(np, 0);
    for (int br = 0; br < p.nbr; br++) msgs[own[br]] += p.deg[br];
    *max_msgs = *std::max_element(msgs.begin(), msgs.end());
    // round of an edge = number of later edges in the same block column (fused_rows.h Rounds)
    std::vector<int> seen(p.nbc, 0), round_of(p.nedge, 0);
    for (int e = p.nedge - 1; e >= 0; e--) { round_of[e] = seen[p.bc[e]]++; }
    int nr = 0;
    for (int e = 0; e < p.nedge; e++) nr = std::max(nr, round_of[e] + 1);
    std::vector<int> cnt((size_t)nr * np, 0);
    for (int br = 0; br < p.nbr; br++)
        for (int e = p.ebeg[br]; e < p.ebeg[br + 1]; e++) cnt[(size_t)round_of[e] * np + own[br]]++;
    long pb = 0;
    for (int q = 0; q < nr; q++) pb += *std::max_element(cnt.begin() + (size_t)q * np, cnt.begin() + (size_t)(q + 1) * np);
    *path = pb + 4L * *max_msgs;   // phase A costs ~4x a phase-B edge
}

constexpr int kPk16MaxColumnDegree = 30;   // = pk::PK16_MAX_COLUMN_DEGREE (fused_pk16_body.h, device header)

const char *choose_plan(const ldpc_code &c, int variant, PlanChoice &p, int kind = JIT_SPLIT) {
    if (c.sz <= 0) return "code was not created from a quasi-cyclic description";
    if (c.sz < 16) return "circulant size below 16 (the generic on-chip kernel takes those)";
    if (c.sz > 1024) return "circulant size above 1024";
    p.sz = c.sz; p.nbr = c.block_rows; p.nbc = c.block_cols;
    // fused_common.h QcGeom: power-of-two sizes below 64 interleave 64/sz frames per wave; any other size runs one frame
    // on sz rounded up to whole waves (the top lanes idle, positions wrap with sub + min instead of an AND)
    const bool pow2 = (c.sz & (c.sz - 1)) == 0;
    p.cpw = (pow2 && c.sz < 64) ? 64 / c.sz : 1;
    const int vpos = c.sz * p.cpw;          // positions of a block column in LDS
    p.v = (vpos + 63) / 64 * 64;            // threads of one wave group
    p.deg.assign(p.nbr, 0); p.ebeg.assign(p.nbr + 1, 0);
    for (int br = 0; br < p.nbr; br++) {
        for (int bc = 0; bc < p.nbc; bc++) {
            int off = c.offsets[(size_t)br * p.nbc + bc];
            if (off >= 0) { p.rot.push_back(off); p.bc.push_back(bc); p.deg[br]++; }
        }
        p.ebeg[br + 1] = (int)p.rot.size();
        p.dmax = std::max(p.dmax, p.deg[br]);
        if (p.deg[br] == 0) return "an empty block row";
    }
    p.nedge = (int)p.rot.size();
    if (p.dmax > 32) return "block-row weight above 32";
    if (p.nbc > 4096 || p.nedge > 4096) return "more than 4096 block columns / circulants";
    {   // every block column must be hit (the kernel seeds lam from round 0 of each column)
        std::vector<char> hit(p.nbc, 0);
        for (int b : p.bc) hit[b] = 1;
        for (char h : hit) if (!h) return "an empty block column";
    }
    // lam + flags, and for the layered kernels the exchange area of rows split between two wave groups (fused_layered_body.h:
    // 16 + up to 3 words x 2 groups x VT lanes) and a flag word per wave
    const size_t lds = (size_t)p.nbc * vpos * 4 + 64 + ((kind == JIT_LAYERED || kind == JIT_LAYERED_PK16) ? 16 + 24 * (size_t)p.v + 4 * 16 : 0);
    if (lds > 160 * 1024) return "a frame's LLRs do not fit in 160 KB of LDS";
    const int cap = variant == LDPC_TANH ? kMsgCapTanh : kMsgCapMinsum;
    const int max_np = std::min(1024 / p.v, p.nbr);
    int best_np = 0;
    std::vector<int> best_own;
    for (int np = 1; np <= max_np && !best_np; np++) {
        // candidates: round-robin (what the shipped AR4JA instances use), longest-processing-time, then pairwise swaps
        std::vector<std::vector<int>> cands;
        std::vector<int> rr(p.nbr), lpt(p.nbr, 0);
        for (int br = 0; br < p.nbr; br++) rr[br] = br % np;
        cands.push_back(rr);
        {
            std::vector<int> order(p.nbr), load(np, 0);
            for (int i = 0; i < p.nbr; i++) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return p.deg[a] > p.deg[b]; });
            for (int br : order) { int g = (int)(std::min_element(load.begin(), load.end()) - load.begin()); lpt[br] = g; load[g] += p.deg[br]; }
            cands.push_back(lpt);
        }
        // objective, lexicographic: (messages over the register budget, critical path of a turn)
        auto score = [&](const std::vector<int> &o, int *m_out) {
            int m; long pa;
            deal_cost(p, o, np, &m, &pa);
            if (m_out) *m_out = m;
            return std::make_pair(std::max(m - cap, 0), pa);
        };
        std::vector<int> bo = cands[0];
        auto bs = score(bo, nullptr);
        for (auto &o : cands) { auto sc = score(o, nullptr); if (sc < bs) { bs = sc; bo = o; } }
        for (int pass = 0; pass < 4; pass++) {
            bool improved = false;
            for (int a = 0; a < p.nbr; a++) {
                for (int g = 0; g < np; g++) {          // move block row a to group g
                    if (g == bo[a]) continue;
                    const int old = bo[a];
                    bo[a] = g;
                    auto sc = score(bo, nullptr);
                    if (sc < bs) { bs = sc; improved = true; } else bo[a] = old;
                }
                for (int b = a + 1; b < p.nbr; b++) {   // swap the groups of a and b
                    if (bo[a] == bo[b]) continue;
                    std::swap(bo[a], bo[b]);
                    auto sc = score(bo, nullptr);
                    if (sc < bs) { bs = sc; improved = true; } else std::swap(bo[a], bo[b]);
                }
            }
            if (!improved) break;
        }
        int bm = 0;
        (void)score(bo, &bm);
        {   // every group must own at least one block row (each wave group runs the same number of barriers either way,
            // but an idle group would only burn occupancy)
            std::vector<int> used(np, 0);
            for (int g : bo) used[g] = 1;
            if (std::find(used.begin(), used.end(), 0) != used.end()) continue;
        }
        if (bm <= cap) { best_np = np; best_own = bo; }
    }
    if (!best_np) return "more messages per thread than the register budget allows even with the widest workgroup";
    p.np = best_np; p.own = best_own;
    // waves per SIMD the register allocator is asked for: messages + round-0 LLRs + row temporaries
    int msgs = 0;
    { std::vector<int> m(p.np, 0); for (int br = 0; br < p.nbr; br++) m[p.own[br]] += p.deg[br]; msgs = *std::max_element(m.begin(), m.end()); }
    const int norig = (p.nbc + p.np - 1) / p.np + 2;
    // (calibrated on the shipped AR4JA instances: 78 messages + 24 LLRs run best at 4 waves/SIMD for min-sum -- 128 VGPRs,
    //  ~30 spilled -- and at 3 for the tanh rule, which keeps ~3 transient registers per edge of a row)
    int est = msgs + norig + (variant == LDPC_TANH ? 3 * p.dmax + 10 : 24);
    // the other bodies (calibrated on the AR4JA instances, 78 messages): packed-fp16 flooding keeps magnitudes and suffix minima
    // of a row next to the LLR registers (160 VGPRs: 3 waves); the layered bodies have no LLR registers but keep a row's lam,
    // t and addresses (f32: 127 VGPRs, 4 waves; packed: 160, 3 waves)
    if (kind == JIT_PK16) est = msgs + norig + 2 * p.dmax + p.dmax / 2 + 12;
    else if (kind == JIT_LAYERED) est = msgs + 2 * p.dmax + 12;
    else if (kind == JIT_LAYERED_PK16) est = msgs + 4 * p.dmax + 8;
    // two wave groups: the layered bodies split every row between them (fused_layered_body.h, Halves) -- half the transient
    // registers; the packed kernel then runs best at 4 waves/SIMD with ~10 spilled registers (jpl.4096: 66.8 -> 73.7 Gbit/s at 3 dB)
    // (rows of weight >= 8 = LAY_SPLIT_MIN_DEG; lighter rows stay whole, and they are what sets dmax only in codes of light rows)
    if (p.np == 2 && p.dmax >= 8 && kind == JIT_LAYERED) est = msgs + p.dmax + (p.dmax + 1) / 2 + 14;
    if (p.np == 2 && p.dmax >= 8 && kind == JIT_LAYERED_PK16) est = msgs + 2 * p.dmax + 10;
    int w = est <= 64 ? 8 : est <= 80 ? 6 : est <= 96 ? 5 : est <= 128 ? 4 : est <= 168 ? 3 : 2;
    const int threads = p.np * p.v;
    const int wg_per_cu = (int)std::max<size_t>(1, (160 * 1024) / lds);
    const int w_lds = std::max(1, std::min(wg_per_cu, 2048 / threads) * threads / 64 / 4);
    p.waves_per_eu = std::max(1, std::min(w, w_lds));
    return nullptr;
}

std::string list_of(const std::vector<int> &v) {
    std::string s;
    for (size_t i = 0; i < v.size(); i++) { if (i) s += ", "; s += std::to_string(v[i]); }
    return s;
}
}  // namespace

const char *jit_split_why_not(const ldpc_code &c, int variant, int dtype, int kind) {
    const bool packed = kind == JIT_PK16 || kind == JIT_LAYERED_PK16;
    if (dtype != (packed ? LDPC_F16PK : LDPC_F32)) return packed ? "the packed-fp16 kernels serve LDPC_F16PK contexts" : "run-time specialised kernels exist for f32 only";
    if (variant != LDPC_MINSUM && variant != LDPC_TANH) return "unknown variant";
    if (kind != JIT_SPLIT && variant != LDPC_MINSUM) return "the packed-fp16 and layered on-chip kernels implement min-sum";
    if (variant == LDPC_MINSUM && c.min_row_deg < 2) return "min-sum needs check rows of weight >= 2";
    const char *e = getenv("LDPC_JIT");
    if (e && !strcmp(e, "0")) return "disabled (LDPC_JIT=0)";
    PlanChoice p;
    const char *why = choose_plan(c, variant, p, kind);
    if (why) return why;
    if (packed) {   // fused_pk16_body.h "range": the fp16 sums stay finite up to this column degree
        std::vector<int> cd(p.nbc, 0);
        for (int b : p.bc) if (++cd[b] > kPk16MaxColumnDegree) return "packed fp16: a column degree above 30 (its LLR sum could overflow fp16)";
    }
    return nullptr;
}

std::string jit_split_source(const ldpc_code &c, int variant, int dtype, JitKernel *g, int kind) {
    PlanChoice p;
    if (jit_split_why_not(c, variant, dtype, kind) || choose_plan(c, variant, p, kind)) return std::string();
    static const char *const kHeader[] = {"fused_split_body.h", "fused_pk16_body.h", "fused_layered_body.h", "fused_layered_body.h"};
    std::ostringstream s;
    s << "// generated by libldpc_hip (jit.cc) for a " << p.nbr << " x " << p.nbc << " block quasi-cyclic H, circulant size " << p.sz << "\n"
      << "#define SPLIT_RESULT_PACKED 0\n"
      << "#include \"" << kHeader[kind] << "\"\n"
      << "namespace ldpc {\n"
      << "struct JPlan {\n"
      << "    static constexpr int NBR = " << p.nbr << ", NBC = " << p.nbc << ", NEDGE = " << p.nedge << ", DMAX = " << p.dmax << ", NP = " << p.np << ";\n"
      << "    static constexpr int deg_[NBR] = {" << list_of(p.deg) << "};\n"
      << "    static constexpr int ebeg_[NBR + 1] = {" << list_of(p.ebeg) << "};\n"
      << "    static constexpr int own_[NBR] = {" << list_of(p.own) << "};\n"
      << "    static constexpr int deg(int br) { return deg_[br]; }\n"
      << "    static constexpr int ebeg(int br) { return ebeg_[br]; }\n"
      << "    static constexpr int owner_br(int br) { return own_[br]; }\n"
      << "};\n"
      << "struct JTab {\n"
      << "    static constexpr int SZ = " << p.sz << ", NBR = " << p.nbr << ", NBC = " << p.nbc << ", NEDGE = " << p.nedge << ";\n"
      << "    static constexpr uint16_t rot[NEDGE] = {" << list_of(p.rot) << "};\n"
      << "    static constexpr uint16_t bc[NEDGE] = {" << list_of(p.bc) << "};\n"
      << "};\n"
      << "}  // namespace ldpc\n";
    const std::string body = s.str();
    static const char *const kStem[] = {"split", "pk16", "layered", "layered_pk16"};
    char name[96];
    snprintf(name, sizeof(name), "ldpc_jit_%s_%s_sz%d_%s", kStem[kind], variant == LDPC_MINSUM ? "minsum" : "tanh", p.sz, hash_hex(body + kStem[kind]).substr(0, 10).c_str());
    std::ostringstream k;
    k << body << "extern \"C\" __global__ __launch_bounds__(" << p.np * p.v << ") __attribute__((amdgpu_waves_per_eu(" << p.waves_per_eu << ", " << p.waves_per_eu << ")))\n"
      << "void " << name << "(ldpc::FusedArgs A) {\n";
    if (kind == JIT_SPLIT) k << "    ldpc::split_kernel_body<float, " << (variant == LDPC_MINSUM ? "LDPC_V_MINSUM" : "LDPC_V_TANH") << ", ldpc::JPlan, " << p.sz << ", ldpc::JTab>(A);\n";
    else k << "    ldpc::" << (kind == JIT_PK16 ? "pk" : kind == JIT_LAYERED ? "lay" : "laypk") << "::kernel_body<ldpc::JPlan, " << p.sz << ", ldpc::JTab>(A);\n";
    k << "}\n";
    const bool packed = kind == JIT_PK16 || kind == JIT_LAYERED_PK16;
    if (g) { g->threads = p.np * p.v; g->frames_per_wg = (packed ? 2 : 1) * p.cpw; g->np = p.np; g->waves_per_eu = p.waves_per_eu; g->name = name; }
    return k.str();
}

void jit_destroy(JitKernel *k) {
    if (!k) return;
    if (k->mod) { (void)hipSetDevice(k->device); (void)hipModuleUnload(k->mod); }
    delete k;
}

JitKernel *jit_split_create(const ldpc_code &c, int variant, int dtype, int kind) {
    const char *why = jit_split_why_not(c, variant, dtype, kind);
    if (why) { set_error(LDPC_EUNSUPPORTED, "%s", why); return nullptr; }
    JitKernel *k = new (std::nothrow) JitKernel();
    if (!k) { set_error(LDPC_ENOMEM, "out of host memory"); return nullptr; }
    try {
        const std::string src = jit_split_source(c, variant, dtype, k, kind);
        std::vector<char> co;
        if (jit_compile_cached(src, k->name, co, &k->from_cache, &k->compile_seconds) != LDPC_OK) { delete k; return nullptr; }
        (void)hipGetDevice(&k->device);
        hipError_t e = hipModuleLoadData(&k->mod, co.data());
        if (e == hipSuccess) e = hipModuleGetFunction(&k->fn, k->mod, k->name.c_str());
        if (e != hipSuccess) {
            set_error(LDPC_EHIP, "loading the run-time compiled kernel %s: %s", k->name.c_str(), hipGetErrorString(e));
            jit_destroy(k);
            return nullptr;
        }
        if (const char *v = getenv("LDPC_JIT_VERBOSE"); v && !strcmp(v, "1"))
            fprintf(stderr, "[ldpc jit] %s: %d threads/workgroup, %d wave groups, %d waves/SIMD, %d spilled VGPRs, %s (%.1f s)\n", k->name.c_str(), k->threads,
                    k->np, k->waves_per_eu, spill_count(co), k->from_cache ? "from cache" : "compiled", k->compile_seconds);
        return k;
    } catch (...) { delete k; set_error(LDPC_ENOMEM, "out of host memory"); return nullptr; }
}

}  // namespace ldpc
