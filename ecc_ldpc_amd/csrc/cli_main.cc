// ecc-ldpc-hip -- native command-line driver over the C ABI of include/ldpc_hip.h, shaped like the reference's
// executable (main/Main.hs:38-48, NOTES.txt:2-3):
//     ecc-ldpc-hip <Eb/N0 values ...> <code names ...> [-m<frames>] [-b<batch>] [-s<seed>] [-d<dev>[,<dev>...]] [-t<rccl|host>] [-c<codes dir>] [--json]
// (--json: each row additionally as one JSON object on its own line -- value in Mbit/s, ranks, per-rank seconds, tally route:
//  the one-process twin of bench.py's N-rank line)
// e.g.  ecc-ldpc-hip 2 3 4 ldpc/hip-minsum/jpl.1024.4.5/50/4/5 -m262144 -d0,1,2,3,4,5,6,7
// Code names use the reference's grammar ldpc/<decoder>/<matrix>/<max-rounds>[/x/y] (Utils.hs:82-88,100-108),
// <decoder> in {hip-tanh, hip-minsum}[-layered][-f32|-f64|-f16].  One row per (code, Eb/N0), like eccPrinter's (NOTES.txt:3):
//     seconds  name  Eb/N0  frames  bit-errors  BER   [+ FER, mean iterations, Mbit/s, path]
// The external tester (ecc-manifold: confidence intervals, stopping rule) is not reproduced: frames from the
// library's device frame source are decoded in device batches until -m frames are done.  Everything stays on the
// GPUs: generate -> decode -> tally on one stream per device; only the four tallies come back per row.
//
// Several devices (-d0,1,...): ONE process, one host thread per GPU, one decoder replica of the record per thread
// (ldpc_ecc_create_replicas = the reference's maxThreadCount replicas, Utils.hs:53).  Frames are independent: rank r
// decodes a contiguous range of global frame ids generated on its own GPU (counter-based RNG: no scatter), and the only
// exchange is ONE all-reduce (sum) of the four uint64 tallies per row over RCCL (xGMI), -trccl (default when the
// devices are distinct).  -thost sums on the host instead (needed when a device is listed twice, which RCCL refuses).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ldpc_hip.h"

namespace {
struct Barrier {   // (std::barrier is C++20)
    std::mutex mu; std::condition_variable cv; int n, waiting = 0; long phase = 0;
    explicit Barrier(int n_) : n(n_) {}
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        const long p = phase;
        if (++waiting == n) { waiting = 0; phase++; cv.notify_all(); }
        else cv.wait(lk, [&] { return phase != p; });
    }
};

struct Shared {
    std::vector<int> devs;
    std::vector<ncclComm_t> comms;
    bool use_rccl = false;
    ldpc_ecc *ecc = nullptr;
    std::vector<double> ebn0s;
    long frames = 0;
    int batch = 0;
    uint64_t seed = 0;
    Barrier *bar = nullptr;
    std::mutex mu;
    uint64_t host_sum[4] = {0, 0, 0, 0};
    std::vector<double> rank_seconds;   // each rank's own generate+decode+tally time for the current row
    bool json = false;
    int failed = 0;
    std::string err;
};

void fail(Shared &S, const std::string &m) {
    std::lock_guard<std::mutex> lk(S.mu);
    if (!S.failed) S.err = m;
    S.failed = 1;
}

// rank r of P: frames [first, first + mine) of the global id range, block partition (first ranks take the remainder)
void shard(long total, int r, int P, long *first, long *mine) {
    const long base = total / P, extra = total % P;
    *mine = base + (r < extra ? 1 : 0);
    *first = (long)r * base + std::min<long>(r, extra);
}

void rank_main(Shared &S, int r) {
    const int P = (int)S.devs.size();
    bool ok = true;
    hipStream_t st = nullptr;
    float *d_llr = nullptr; uint8_t *d_bits = nullptr; int32_t *d_iters = nullptr; uint64_t *d_tally = nullptr;
    ldpc_ctx *ctx = ldpc_ecc_ctx_at(S.ecc, r);
    ldpc_sim *sim = ldpc_ecc_sim_at(S.ecc, r);
    const int k = ldpc_ecc_message_length(S.ecc), N = ldpc_ecc_unpunctured_length(S.ecc), iters_max = ldpc_ecc_max_iters(S.ecc);
    auto H = [&](hipError_t e, const char *what) { if (e != hipSuccess && ok) { ok = false; fail(S, std::string(what) + ": " + hipGetErrorString(e)); } };
    H(hipSetDevice(S.devs[r]), "hipSetDevice");
    H(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
    H(hipMalloc((void **)&d_llr, (size_t)S.batch * N * sizeof(float)), "hipMalloc llr");
    H(hipMalloc((void **)&d_bits, (size_t)S.batch * N), "hipMalloc bits");
    H(hipMalloc((void **)&d_iters, (size_t)S.batch * sizeof(int32_t)), "hipMalloc iters");
    H(hipMalloc((void **)&d_tally, 4 * sizeof(uint64_t)), "hipMalloc tally");
    long first = 0, mine = 0;
    shard(S.frames, r, P, &first, &mine);
    for (double db : S.ebn0s) {
        if (ok) { H(hipMemsetAsync(d_tally, 0, 4 * sizeof(uint64_t), st), "memset"); H(hipStreamSynchronize(st), "sync"); }
        if (r == 0) { std::lock_guard<std::mutex> lk(S.mu); memset(S.host_sum, 0, sizeof(S.host_sum)); }
        S.bar->wait();
        const auto t0 = std::chrono::steady_clock::now();
        for (long done = 0; ok && done < mine; done += S.batch) {
            const int b = (int)std::min<long>(S.batch, mine - done);
            int rc = ldpc_sim_generate(sim, S.seed, (uint64_t)(first + done), b, db, d_llr, nullptr, st);
            if (rc == LDPC_OK) rc = ldpc_decode_batch_dev(ctx, iters_max, b, d_llr, d_bits, d_iters, nullptr, st);
            if (rc == LDPC_OK) rc = ldpc_sim_tally(sim, b, d_bits, d_iters, d_tally, st);
            if (rc != LDPC_OK) { ok = false; fail(S, ldpc_last_error()); }
        }
        if (st) (void)hipStreamSynchronize(st);
        S.rank_seconds[r] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        S.bar->wait();                                   // every rank's frames are decoded: the row's wall time ends here
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        uint64_t t[4] = {0, 0, 0, 0};
        int anyfail;
        { std::lock_guard<std::mutex> lk(S.mu); anyfail = S.failed; }   // same answer on every rank: failures precede the barrier
        if (anyfail) break;                              // nobody enters the collective
        if (S.use_rccl) {                                // the path's only collective: 32 bytes
            if (ok) {
                ncclResult_t nr = ncclAllReduce(d_tally, d_tally, 4, ncclUint64, ncclSum, S.comms[r], st);
                if (nr != ncclSuccess) { ok = false; fail(S, std::string("ncclAllReduce: ") + ncclGetErrorString(nr)); }
                H(hipStreamSynchronize(st), "sync after all-reduce");
                H(hipMemcpy(t, d_tally, sizeof(t), hipMemcpyDeviceToHost), "copy tallies");
            }
        } else {
            if (ok) H(hipMemcpy(t, d_tally, sizeof(t), hipMemcpyDeviceToHost), "copy tallies");
            { std::lock_guard<std::mutex> lk(S.mu); for (int i = 0; i < 4; i++) S.host_sum[i] += t[i]; }
            S.bar->wait();
            { std::lock_guard<std::mutex> lk(S.mu); memcpy(t, S.host_sum, sizeof(t)); }
        }
        S.bar->wait();
        if (r == 0 && !S.failed) {
            const double f = (double)t[0];
            printf("%8.2f %s  %4.2f %8llu %8llu  %.2e   FER %.2e  iters %5.1f  %9.1f Mbit/s [%s%s]\n", dt, ldpc_ecc_name(S.ecc), db, (unsigned long long)t[0],
                   (unsigned long long)t[2], f > 0 ? (double)t[2] / (f * k) : 0.0, f > 0 ? (double)t[1] / f : 0.0, f > 0 ? (double)t[3] / f : 0.0,
                   f * k / dt / 1e6, ldpc_ctx_path(ctx) == LDPC_PATH_FUSED ? "fused" : "flood",
                   P > 1 ? (S.use_rccl ? (", " + std::to_string(P) + " GPUs, rccl").c_str() : (", " + std::to_string(P) + " ranks, host sum").c_str()) : "");
            if (S.json) {
                std::string per;
                for (int i = 0; i < P; i++) per += (i ? ", " : "") + std::to_string(S.rank_seconds[i]);
                printf("{\"metric\": \"decoded info Mbit/s\", \"value\": %.2f, \"unit\": \"Mbit/s\", \"code_name\": \"%s\", \"ebn0_db\": %g, \"ranks\": %d, "
                       "\"tally\": \"%s\", \"frames\": %llu, \"frame_errors\": %llu, \"bit_errors\": %llu, \"mean_iters\": %.3f, \"seconds\": %.6f, "
                       "\"per_rank_seconds\": [%s], \"batch\": %d, \"path\": \"%s\", \"process_model\": \"one process, one host thread per rank\"}\n",
                       f * k / dt / 1e6, ldpc_ecc_name(S.ecc), db, P, S.use_rccl ? "rccl" : "host", (unsigned long long)t[0], (unsigned long long)t[1],
                       (unsigned long long)t[2], f > 0 ? (double)t[3] / f : 0.0, dt, per.c_str(), S.batch, ldpc_ctx_path(ctx) == LDPC_PATH_FUSED ? "fused" : "flood");
            }
            fflush(stdout);
        }
        if (S.failed) break;
    }
    (void)hipFree(d_llr); (void)hipFree(d_bits); (void)hipFree(d_iters); (void)hipFree(d_tally);
    if (st) (void)hipStreamDestroy(st);
}
}  // namespace

// -H<threads>[,<coalesce>[,<wait us>]]: what a per-frame HARNESS sees.  <threads> host threads call the record's decode
// (ldpc_ecc_decode: one frame per call, host double LLRs in, message bits out -- the closure of Utils.hs:62-72) on frames
// taken from the device frame source; with <coalesce> > 0 the calls go through a batcher (ldpc_ecc_set_coalescing).
static int harness_mode(const std::string &codes_dir, const std::string &name, double db, long frames, uint64_t seed, int threads, int coalesce, int wait_us) {
    const int pool = 4096;
    ldpc_ecc *ecc = ldpc_ecc_create(codes_dir.c_str(), name.c_str(), std::max(pool, coalesce));
    if (!ecc) { fprintf(stderr, "# %s: %s\n", name.c_str(), ldpc_last_error()); return 1; }
    const int k = ldpc_ecc_message_length(ecc), n_tx = ldpc_ecc_codeword_length(ecc), N = ldpc_ecc_unpunctured_length(ecc);
    float *d_llr = nullptr;
    if (hipMalloc((void **)&d_llr, (size_t)pool * N * sizeof(float)) != hipSuccess) return 1;
    if (ldpc_sim_generate(ldpc_ecc_sim(ecc), seed, 0, pool, db, d_llr, nullptr, nullptr) != LDPC_OK) { fprintf(stderr, "%s\n", ldpc_last_error()); return 1; }
    std::vector<float> h((size_t)pool * N);
    if (hipMemcpy(h.data(), d_llr, h.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    (void)hipFree(d_llr);
    std::vector<double> llr((size_t)pool * n_tx);
    for (int f = 0; f < pool; f++) for (int i = 0; i < n_tx; i++) llr[(size_t)f * n_tx + i] = h[(size_t)f * N + i];
    if (coalesce > 0 && ldpc_ecc_set_coalescing(ecc, coalesce, wait_us) != LDPC_OK) { fprintf(stderr, "%s\n", ldpc_last_error()); return 1; }
    std::vector<std::thread> th;
    std::vector<long> ok_count(threads, 0);
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            std::vector<uint8_t> out((size_t)k);
            for (long f = t; f < frames; f += threads) {
                int ok = 0;
                (void)ldpc_ecc_decode(ecc, &llr[(size_t)(f % pool) * n_tx], out.data(), &ok);
                ok_count[t] += ok;
            }
        });
    for (auto &x : th) x.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    long calls = 0, launches = 0;
    (void)ldpc_ecc_coalescing_stats(ecc, &calls, &launches);
    printf("%8.2f %s  %4.2f %8ld frames one per call, %d threads, coalescing %d (wait %d us): %9.2f Mbit/s harness-visible, %8.0f frames/s, %.1f us per call per thread, %ld launches\n",
           dt, ldpc_ecc_name(ecc), db, frames, threads, coalesce, wait_us, frames * (double)k / dt / 1e6, frames / dt, dt / frames * threads * 1e6,
           coalesce > 0 ? launches : frames);
    ldpc_ecc_destroy(ecc);
    return 0;
}

int main(int argc, char **argv) {
    Shared S;
    std::vector<std::string> names;
    S.frames = 65536; S.batch = 16384; S.seed = 0x5EEDC0DEull;
    S.devs = {0};
    std::string tally = "auto";
    bool harness = false;
    int h_threads = 1, h_coalesce = 0, h_wait = 200;
    std::string codes_dir = getenv("LDPC_CODES_DIR") ? getenv("LDPC_CODES_DIR") : "codes";
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (!strncmp(a, "-m", 2)) S.frames = atol(a + 2);
        else if (!strncmp(a, "-b", 2)) S.batch = atoi(a + 2);
        else if (!strncmp(a, "-s", 2)) S.seed = strtoull(a + 2, nullptr, 0);
        else if (!strncmp(a, "-d", 2)) {
            S.devs.clear();
            for (const char *p = a + 2; *p;) { S.devs.push_back((int)strtol(p, (char **)&p, 10)); if (*p == ',') p++; else break; }
        }
        else if (!strncmp(a, "-t", 2)) tally = a + 2;
        else if (!strncmp(a, "-H", 2)) { harness = true; sscanf(a + 2, "%d,%d,%d", &h_threads, &h_coalesce, &h_wait); }
        else if (!strncmp(a, "-c", 2)) codes_dir = a + 2;
        else if (!strcmp(a, "--json")) S.json = true;
        else {
            char *end = nullptr;
            double v = strtod(a, &end);
            if (end != a && *end == 0) S.ebn0s.push_back(v); else names.push_back(a);
        }
    }
    if (S.ebn0s.empty() || names.empty() || S.frames <= 0 || S.batch <= 0 || S.devs.empty() || !(tally == "auto" || tally == "rccl" || tally == "host")) {
        fprintf(stderr, "usage: %s <Eb/N0 values ...> <code names ...> [-m<frames>] [-b<batch>] [-s<seed>] [-d<dev>[,<dev>...]] [-t<rccl|host>] [-c<codes dir>] [--json]\n", argv[0]);
        return 2;
    }
    const int P = (int)S.devs.size();
    S.rank_seconds.assign(P, 0.0);
    const bool distinct = std::set<int>(S.devs.begin(), S.devs.end()).size() == S.devs.size();
    S.use_rccl = tally == "rccl" || (tally == "auto" && P > 1 && distinct);
    if (S.use_rccl && !distinct) { fprintf(stderr, "-trccl needs distinct devices (RCCL has one rank per GPU); use -thost\n"); return 2; }
    if (ldpc_init(S.devs[0]) != LDPC_OK) { fprintf(stderr, "ldpc_init: %s\n", ldpc_last_error()); return 1; }
    if (harness) {
        int rc = 0;
        for (const std::string &name : names)
            for (double db : S.ebn0s) rc |= harness_mode(codes_dir, name, db, S.frames, S.seed, std::max(1, h_threads), h_coalesce, h_wait);
        ldpc_shutdown();
        return rc;
    }
    const long per_rank = (S.frames + P - 1) / P;
    if (S.batch > per_rank) S.batch = (int)per_rank;
    if (S.use_rccl) {
        S.comms.resize(P);
        ncclResult_t nr = ncclCommInitAll(S.comms.data(), P, S.devs.data());
        if (nr != ncclSuccess) { fprintf(stderr, "ncclCommInitAll: %s\n", ncclGetErrorString(nr)); return 1; }
    }
    int rc_all = 0;
    for (const std::string &name : names) {
        S.ecc = ldpc_ecc_create_replicas(codes_dir.c_str(), name.c_str(), S.batch, P, S.devs.data());
        if (!S.ecc) {
            fprintf(stderr, "# %s: %s\n", name.c_str(), ldpc_last_error());
            if (ldpc_last_error_code() != LDPC_ENOTFOUND) rc_all = 1;
            continue;
        }
        Barrier bar(P);
        S.bar = &bar; S.failed = 0;
        std::vector<std::thread> th;
        for (int r = 1; r < P; r++) th.emplace_back(rank_main, std::ref(S), r);
        rank_main(S, 0);
        for (auto &t : th) t.join();
        if (S.failed) { fprintf(stderr, "# %s: %s\n", name.c_str(), S.err.c_str()); rc_all = 1; }
        ldpc_ecc_destroy(S.ecc);
        S.ecc = nullptr;
    }
    for (auto c : S.comms) (void)ncclCommDestroy(c);
    ldpc_shutdown();
    return rc_all;
}
