// ecc-ldpc-hip -- native command-line driver over the C ABI of include/ldpc_hip.h, shaped like the reference's
// executable (main/Main.hs:38-48, NOTES.txt:2-3):
//     ecc-ldpc-hip <Eb/N0 values ...> <code names ...> [-m<frames>] [-b<batch>] [-s<seed>] [-d<device>] [-c<codes dir>]
// e.g.  ecc-ldpc-hip 2 3 4 ldpc/hip-minsum/jpl.1024.4.5/50/4/5 -m262144
// Code names use the reference's grammar ldpc/<decoder>/<matrix>/<max-rounds>[/x/y] (Utils.hs:82-88,100-108),
// <decoder> in {hip-tanh, hip-minsum}[-f32|-f64|-f16].  One row per (code, Eb/N0), like eccPrinter's (NOTES.txt:3):
//     seconds  name  Eb/N0  frames  bit-errors  BER   [+ FER, mean iterations, Mbit/s, path]
// The external tester (ecc-manifold: confidence intervals, stopping rule) is not reproduced: frames from the
// library's device frame source are decoded in device batches until -m frames are done.  Everything stays on the
// GPU: generate -> decode -> tally on one stream; only the four tallies come back per row.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <vector>

#include "../../include/ldpc_hip.h"

#define HIP_OR_DIE(x)                                                                  \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } \
    } while (0)

int main(int argc, char **argv) {
    std::vector<double> ebn0s;
    std::vector<std::string> names;
    long frames = 65536;
    int batch = 16384, device = 0;
    uint64_t seed = 0x5EEDC0DEull;
    std::string codes_dir = getenv("LDPC_CODES_DIR") ? getenv("LDPC_CODES_DIR") : "codes";
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (!strncmp(a, "-m", 2)) frames = atol(a + 2);
        else if (!strncmp(a, "-b", 2)) batch = atoi(a + 2);
        else if (!strncmp(a, "-s", 2)) seed = strtoull(a + 2, nullptr, 0);
        else if (!strncmp(a, "-d", 2)) device = atoi(a + 2);
        else if (!strncmp(a, "-c", 2)) codes_dir = a + 2;
        else {
            char *end = nullptr;
            double v = strtod(a, &end);
            if (end != a && *end == 0) ebn0s.push_back(v); else names.push_back(a);
        }
    }
    if (ebn0s.empty() || names.empty() || frames <= 0 || batch <= 0) {
        fprintf(stderr, "usage: %s <Eb/N0 values ...> <code names ...> [-m<frames>] [-b<batch>] [-s<seed>] [-d<device>] [-c<codes dir>]\n", argv[0]);
        return 2;
    }
    if (ldpc_init(device) != LDPC_OK) { fprintf(stderr, "ldpc_init: %s\n", ldpc_last_error()); return 1; }
    if (batch > frames) batch = (int)frames;
    hipStream_t st;
    HIP_OR_DIE(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int rc_all = 0;
    for (const std::string &name : names) {
        ldpc_ecc *ecc = ldpc_ecc_create(codes_dir.c_str(), name.c_str(), batch);
        if (!ecc) {
            fprintf(stderr, "# %s: %s\n", name.c_str(), ldpc_last_error());
            if (ldpc_last_error_code() != LDPC_ENOTFOUND) rc_all = 1;
            continue;
        }
        const int k = ldpc_ecc_message_length(ecc), N = ldpc_ecc_unpunctured_length(ecc), iters_max = ldpc_ecc_max_iters(ecc);
        ldpc_ctx *ctx = ldpc_ecc_ctx(ecc);
        ldpc_sim *sim = ldpc_ecc_sim(ecc);
        float *d_llr = nullptr;
        uint8_t *d_bits = nullptr;
        int32_t *d_iters = nullptr;
        uint64_t *d_tally = nullptr;
        HIP_OR_DIE(hipMalloc((void **)&d_llr, (size_t)batch * N * sizeof(float)));
        HIP_OR_DIE(hipMalloc((void **)&d_bits, (size_t)batch * N));
        HIP_OR_DIE(hipMalloc((void **)&d_iters, (size_t)batch * sizeof(int32_t)));
        HIP_OR_DIE(hipMalloc((void **)&d_tally, 4 * sizeof(uint64_t)));
        for (double db : ebn0s) {
            HIP_OR_DIE(hipMemsetAsync(d_tally, 0, 4 * sizeof(uint64_t), st));
            HIP_OR_DIE(hipStreamSynchronize(st));
            const auto t0 = std::chrono::steady_clock::now();
            int rc = LDPC_OK;
            for (long done = 0; done < frames && rc == LDPC_OK; done += batch) {
                const int b = (int)std::min<long>(batch, frames - done);
                rc = ldpc_sim_generate(sim, seed, (uint64_t)done, b, db, d_llr, nullptr, st);
                if (rc == LDPC_OK) rc = ldpc_decode_batch_dev(ctx, iters_max, b, d_llr, d_bits, d_iters, nullptr, st);
                if (rc == LDPC_OK) rc = ldpc_sim_tally(sim, b, d_bits, d_iters, d_tally, st);
            }
            HIP_OR_DIE(hipStreamSynchronize(st));
            if (rc != LDPC_OK) { fprintf(stderr, "# %s at %.2f dB: %s\n", name.c_str(), db, ldpc_last_error()); rc_all = 1; break; }
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            uint64_t t[4];
            HIP_OR_DIE(hipMemcpy(t, d_tally, sizeof(t), hipMemcpyDeviceToHost));
            const double f = (double)t[0];
            printf("%8.2f %s  %4.2f %8llu %8llu  %.2e   FER %.2e  iters %5.1f  %9.1f Mbit/s [%s]\n", dt, ldpc_ecc_name(ecc), db,
                   (unsigned long long)t[0], (unsigned long long)t[2], f > 0 ? (double)t[2] / (f * k) : 0.0, f > 0 ? (double)t[1] / f : 0.0,
                   f > 0 ? (double)t[3] / f : 0.0, f * k / dt / 1e6, ldpc_ctx_path(ctx) == LDPC_PATH_FUSED ? "fused" : "flood");
            fflush(stdout);
        }
        (void)hipFree(d_llr); (void)hipFree(d_bits); (void)hipFree(d_iters); (void)hipFree(d_tally);
        ldpc_ecc_destroy(ecc);
    }
    (void)hipStreamDestroy(st);
    ldpc_shutdown();
    return rc_all;
}
