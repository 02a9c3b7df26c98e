// sim.hip -- device-side frame source and error tally for the BER/throughput harness.
//
// The reference delegates this to the external tester (ecc-manifold's eccMain, main/Main.hs:41-48:
// message generation, encode, BPSK + AWGN, BER statistics; its source is not vendored, SURVEY.md
// section 8c), so the channel model is this build's own stated one:
//   message bits uniform; systematic encode  codeword = msg ++ take (c_length - k) (msg * G)
//   (src/ECC/Code/LDPC/Utils.hs:61, Reference/Orig.hs:25-26, Fast/Encoder.hs:26-63);
//   BPSK bit b -> 2b-1 (LLR > 0 <=> bit 1, `hard x = x > 0`); noise N(0, sigma^2),
//   sigma^2 = 1/(2 R 10^(EbN0/10)), R = k/n_tx; LLR = 2y/sigma^2; punctured tail LLR = 0
//   (Utils.hs:55 `unpuncture`).
// Randomness: Philox4x32-10 keyed by the 64-bit seed, counter = (global frame id, index, stream),
// so a frame's content depends only on (seed, frame id): ranks generate disjoint frame ranges
// with no scatter.
#include "internal.h"
#include "sim.h"
#include <hip/hip_fp16.h>

namespace ldpc {

struct Philox {
    static __device__ __forceinline__ void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
        const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
        uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
        uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
        uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    static __device__ __forceinline__ void gen(uint64_t seed, uint64_t frame, uint32_t idx, uint32_t stream, uint32_t (&out)[4]) {
        uint32_t c[4] = {(uint32_t)frame, (uint32_t)(frame >> 32), idx, stream};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; r++) {
            round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
    }
};

// message words: msgw[frame][w], bit i of word w = message bit 32*w + i
__global__ void sim_msg_kernel(uint32_t *msgw, int kwords, int k, uint64_t seed, uint64_t first_frame, int batch, int zero_msg) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)batch * kwords) return;
    int f = (int)(i / kwords), w = (int)(i % kwords);
    uint32_t r[4];
    Philox::gen(seed, first_frame + f, (uint32_t)w, 0u, r);
    uint32_t v = zero_msg ? 0u : r[0];
    int rem = k - 32 * w;
    if (rem < 32) v &= (rem <= 0) ? 0u : ((1u << rem) - 1u);
    msgw[i] = v;
}

// one thread per (frame, transmitted or punctured position n)
template <typename OT>
__global__ __launch_bounds__(256) void sim_frame_kernel(SimDev s, const uint32_t *__restrict__ msgw, OT *__restrict__ llr,
                                                        uint8_t *__restrict__ msg_bytes, uint64_t seed, uint64_t first_frame,
                                                        int batch, float sigma, float llr_scale) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (n >= s.N || f >= batch) return;
    float out = 0.f;
    if (n < s.n_tx) {
        const uint32_t *mw = msgw + (size_t)f * s.kwords;
        uint32_t bit;
        if (n < s.k) {
            bit = (mw[n >> 5] >> (n & 31)) & 1u;
            if (msg_bytes) msg_bytes[(size_t)f * s.k + n] = (uint8_t)bit;
        } else if (s.gt) { // parity bit j = <msg, column j of G> over GF(2)
            const uint32_t *col = s.gt + (size_t)(n - s.k) * s.kwords;
            uint32_t acc = 0;
            for (int w = 0; w < s.kwords; w++) acc ^= mw[w] & col[w];
            bit = __popc(acc) & 1u;
        } else {
            bit = 0u; // no generator: all-zero codeword
        }
        uint32_t r[4];
        Philox::gen(seed, first_frame + f, (uint32_t)n, 1u, r);
        // Box-Muller on two 32-bit uniforms (u1 in (0,1])
        float u1 = ((float)r[0] + 1.0f) * 2.3283064365386963e-10f;
        float u2 = (float)r[1] * 2.3283064365386963e-10f;
        float z = sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
        float y = (bit ? 1.0f : -1.0f) + sigma * z;
        out = llr_scale * y;
    } else if (n < s.k && msg_bytes) {
        msg_bytes[(size_t)f * s.k + n] = (uint8_t)((msgw[(size_t)f * s.kwords + (n >> 5)] >> (n & 31)) & 1u);
    }
    if constexpr (sizeof(OT) == 2) llr[(size_t)f * s.N + n] = __float2half_rn(fminf(fmaxf(out, -65504.f), 65504.f));  // = round_f16
    else llr[(size_t)f * s.N + n] = out;
}

// tally[0..3] += {frames, frame errors, message-bit errors, sum of iterations}; one wave per frame
__global__ __launch_bounds__(256) void sim_tally_kernel(SimDev s, const uint32_t *__restrict__ msgw, const uint8_t *__restrict__ bits,
                                                        const int32_t *__restrict__ iters, unsigned long long *tally, int batch) {
    const int lane = threadIdx.x & 63;
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= batch) return;
    unsigned errs = 0;
    for (int n = lane; n < s.k; n += 64) {
        unsigned m = (msgw[(size_t)f * s.kwords + (n >> 5)] >> (n & 31)) & 1u;
        errs += (bits[(size_t)f * s.N + n] != m);
    }
    for (int o = 32; o > 0; o >>= 1) errs += __shfl_down(errs, o, 64);
    if (lane == 0) {
        atomicAdd(&tally[0], 1ull);
        if (errs) atomicAdd(&tally[1], 1ull);
        if (errs) atomicAdd(&tally[2], (unsigned long long)errs);
        if (iters) atomicAdd(&tally[3], (unsigned long long)iters[f]);
    }
}

int sim_generate(const SimDev &s, uint32_t *msgw, hipStream_t st, uint64_t seed, uint64_t first_frame, int batch,
                 double ebn0_db, void *d_llr, int llr_f16, uint8_t *d_msg) {
    const double R = (double)s.k / (double)s.n_tx;
    const double sigma2 = 1.0 / (2.0 * R * pow(10.0, ebn0_db / 10.0));
    size_t nw = (size_t)batch * s.kwords;
    hipLaunchKernelGGL(sim_msg_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, msgw, s.kwords, s.k, seed, first_frame, batch, s.gt ? 0 : 1);
    const dim3 grid((s.N + 255) / 256, batch);
    if (llr_f16)
        hipLaunchKernelGGL(sim_frame_kernel<__half>, grid, dim3(256), 0, st, s, msgw, (__half *)d_llr, d_msg, seed, first_frame, batch,
                           (float)sqrt(sigma2), (float)(2.0 / sigma2));
    else
        hipLaunchKernelGGL(sim_frame_kernel<float>, grid, dim3(256), 0, st, s, msgw, (float *)d_llr, d_msg, seed, first_frame, batch,
                           (float)sqrt(sigma2), (float)(2.0 / sigma2));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(LDPC_EHIP, "sim_generate: %s", hipGetErrorString(e));
    return LDPC_OK;
}

int sim_tally(const SimDev &s, const uint32_t *msgw, hipStream_t st, int batch, const uint8_t *d_bits, const int32_t *d_iters,
              unsigned long long *d_tally) {
    hipLaunchKernelGGL(sim_tally_kernel, dim3((batch + 3) / 4), dim3(256), 0, st, s, msgw, d_bits, d_iters, d_tally, batch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(LDPC_EHIP, "sim_tally: %s", hipGetErrorString(e));
    return LDPC_OK;
}

}  // namespace ldpc
