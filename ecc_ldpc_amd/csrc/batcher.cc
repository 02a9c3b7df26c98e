// batcher.cc -- coalescing of concurrent single-frame decode calls into one launch.
//
// The reference's harness calls the decoder closure once per frame, from up to maxThreadCount Haskell threads
// (src/ECC/Code/LDPC/Utils.hs:53,63-69); its CUDA plug-ins therefore decode ONE codeword per launch sequence
// (GPU/CUDA/Arraylet2.hs:151-273).  Through ldpc_decode_one a frame costs a launch and two PCIe copies of its own
// (269 us for 50 turns of jpl.4096: 15 Mbit/s) while the device decodes 65 536 frames in 18 ms.  A batcher sits
// between the per-frame callers and one decoder replica: the first caller to arrive becomes the leader, waits until
// `max_frames` requests are queued or `max_wait_us` have passed, decodes all of them with ONE ldpc_decode_batch_f64
// call and hands every caller its own result.  Callers block exactly like ldpc_decode_one's do; results are those of
// ldpc_decode_one bit for bit (same kernels, frames are independent).
// Measured (jpl.4096, 50 turns, 16 host cores, callers = batch limit): 64 threads 338-376 Mbit/s, 128: 408, 256: 467, 512:
// 339 -- about 100 000 calls/s is what the host side sustains (wake-ups, the 40 KB copy of each call's LLRs).  Copying the
// LLRs outside the lock was tried: +14 % at 64 threads, 2.8x SLOWER at 256 (the leader then waits for descheduled copiers).
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <vector>

#include "internal.h"

using ldpc::set_error;

struct ldpc_batcher {
    ldpc_ctx *ctx = nullptr;
    int N = 0, max_frames = 0, max_wait_us = 0;
    std::mutex mu;
    std::condition_variable cv_leader, cv_done, cv_room;
    // the batch being collected
    struct Request { uint8_t *bits; int *iters, *conv; int rc; bool done; };
    std::vector<Request *> pending;
    int pending_iters = 0;
    bool decoding = false;                 // a leader is inside the decoder replica (which is not re-entrant)
    // staging in page-locked memory (ldpc_host_alloc): batches above 16 frames then run zero-copy -- the decode kernel
    // reads the LLRs and writes the bits over PCIe itself (api.cc decode_host) -- and smaller ones skip a bounce copy
    double *llr[2] = {nullptr, nullptr};   // alternating per batch: arrivals fill one while the other decodes
    int fill = 0;
    uint8_t *out_bits = nullptr, *out_conv = nullptr;
    int32_t *out_iters = nullptr;
    long calls = 0, launches = 0;
};

extern "C" {

ldpc_batcher *ldpc_batcher_create(ldpc_ctx *ctx, int max_frames, int max_wait_us) {
    if (!ctx || max_frames <= 0 || max_wait_us < 0) { set_error(LDPC_EINVAL, "ldpc_batcher_create: bad arguments"); return nullptr; }
    int M = 0, N = 0, E = 0;
    if (ldpc_code_dims(ldpc_ctx_code(ctx), &M, &N, &E) != LDPC_OK) return nullptr;
    if (max_frames > ldpc_ctx_max_batch(ctx)) { set_error(LDPC_EINVAL, "ldpc_batcher_create: %d frames exceed the context's max_batch %d", max_frames, ldpc_ctx_max_batch(ctx)); return nullptr; }
    ldpc_batcher *b = new (std::nothrow) ldpc_batcher();
    if (!b) { set_error(LDPC_ENOMEM, "out of host memory"); return nullptr; }
    try {
        b->ctx = ctx; b->N = N; b->max_frames = max_frames; b->max_wait_us = max_wait_us;
        b->pending.reserve((size_t)max_frames);
    } catch (...) { delete b; set_error(LDPC_ENOMEM, "out of host memory"); return nullptr; }
    (void)hipSetDevice(ldpc_ctx_device(ctx));
    const size_t F = (size_t)max_frames;
    b->llr[0] = (double *)ldpc_host_alloc(F * N * sizeof(double));
    b->llr[1] = (double *)ldpc_host_alloc(F * N * sizeof(double));
    b->out_bits = (uint8_t *)ldpc_host_alloc(F * N);
    b->out_conv = (uint8_t *)ldpc_host_alloc(F);
    b->out_iters = (int32_t *)ldpc_host_alloc(F * sizeof(int32_t));
    if (!b->llr[0] || !b->llr[1] || !b->out_bits || !b->out_conv || !b->out_iters) { ldpc_batcher_destroy(b); return nullptr; }
    return b;
}

void ldpc_batcher_destroy(ldpc_batcher *b) {
    if (!b) return;
    ldpc_host_free(b->llr[0]); ldpc_host_free(b->llr[1]); ldpc_host_free(b->out_bits); ldpc_host_free(b->out_conv); ldpc_host_free(b->out_iters);
    delete b;
}

int ldpc_batcher_stats(ldpc_batcher *b, long *calls, long *launches) {
    if (!b) return set_error(LDPC_EINVAL, "null batcher");
    std::lock_guard<std::mutex> lk(b->mu);
    if (calls) *calls = b->calls;
    if (launches) *launches = b->launches;
    return LDPC_OK;
}

int ldpc_batcher_decode_one(ldpc_batcher *b, int max_iters, const double *llr, uint8_t *bits, int *iters, int *converged) {
    if (!b || !llr || !bits || max_iters < 0) return set_error(LDPC_EINVAL, "ldpc_batcher_decode_one: bad arguments");
    ldpc_batcher::Request req{bits, iters, converged, LDPC_OK, false};
    std::unique_lock<std::mutex> lk(b->mu);
    // a batch has one max_iters and at most max_frames requests: wait for the next one otherwise
    b->cv_room.wait(lk, [&] { return (int)b->pending.size() < b->max_frames && (b->pending.empty() || b->pending_iters == max_iters); });
    const int slot = (int)b->pending.size();
    memcpy(&b->llr[b->fill][(size_t)slot * b->N], llr, sizeof(double) * (size_t)b->N);
    b->pending.push_back(&req);
    b->pending_iters = max_iters;
    b->calls++;
    if (slot > 0) {                         // follower: the leader of this batch will decode it
        if ((int)b->pending.size() == b->max_frames) b->cv_leader.notify_all();
        b->cv_done.wait(lk, [&] { return req.done; });
        if (req.rc != LDPC_OK) return set_error(req.rc, "coalesced decode failed");
        return LDPC_OK;
    }
    // leader: collect until full or until the wait budget is spent, and until the replica is free
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(b->max_wait_us);
    b->cv_leader.wait_until(lk, deadline, [&] { return (int)b->pending.size() == b->max_frames; });
    b->cv_leader.wait(lk, [&] { return !b->decoding; });
    std::vector<ldpc_batcher::Request *> batch;
    batch.swap(b->pending);
    const int n = (int)batch.size(), side = b->fill;
    b->fill ^= 1;
    b->decoding = true;
    b->launches++;
    b->cv_room.notify_all();                // the next batch may start collecting while this one decodes
    lk.unlock();
    int rc = ldpc_decode_batch_f64(b->ctx, max_iters, n, b->llr[side], b->out_bits, b->out_iters, b->out_conv, nullptr);
    const int code = rc == LDPC_OK ? LDPC_OK : ldpc_last_error_code();
    for (int i = 0; i < n; i++) {
        ldpc_batcher::Request *q = batch[i];
        if (rc == LDPC_OK) {
            memcpy(q->bits, &b->out_bits[(size_t)i * b->N], (size_t)b->N);
            if (q->iters) *q->iters = b->out_iters[i];
            if (q->conv) *q->conv = b->out_conv[i];
        }
        q->rc = code;
    }
    lk.lock();
    for (int i = 0; i < n; i++) batch[i]->done = true;
    b->decoding = false;
    b->cv_done.notify_all();
    b->cv_leader.notify_all();
    lk.unlock();
    return rc;
}

}  // extern "C"
