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
// Randomness: Philox4x32-10 keyed by the 64-bit seed, counter = (global frame id, index / 4, stream),
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

// Quasi-cyclic encoder, the reference's formulation (Fast/Encoder.hs:42-63): the message is cut into sz-bit words v'[r],
// the parity word of block column c is  XOR_r mulWord(v'[r], g[r][c]),  mulWord(w1, w2) = XOR over the set bits n of w1 of
// rotateL(w2, n).  Here: LANE = FRAME, a wave = 64 frames x one group of 16/W block columns (W = sz/32 machine words per
// circulant).  The generator is never expanded: for n = 32a + b, word w of rotateL(g, n) is word (w - a) mod W of
// rotateL(g, b), so a table of the 32 bit-rotations of every circulant (scalar loads: 16 words per (r, b), the same for
// all lanes) is all that is read -- 32 x 12 x 32 x 4 words = 196 KB for jpl.4096 instead of the 786 KB dense k x p matrix,
// and it grows with k + p, not k * p.  Per message bit and parity word: one v_bitop3 (acc ^= mask & T), the mask (0 / -1
// from bit b of the lane's message word) shared by the 16/W columns of the group.
// The four waves of a workgroup share the 64 frames and the column group and split the block rows; their partial parity
// words meet in LDS (XOR is associative: any split gives the same bits).  4x the waves of a one-wave version: the scalar
// table loads of one wave hide behind the XORs of the others (jpl.4096, 65 536 frames: 0.91 -> see profiles/r03_encoder_rate.txt).
constexpr int kQcRowSplit = 4;
template <int W>
__global__ __launch_bounds__(64 * kQcRowSplit) void sim_parity_qc_kernel(const uint32_t *__restrict__ rot, int brows, int bcols, const uint32_t *__restrict__ msgw,
                                                                         uint32_t *__restrict__ parw, int kwords, int pwords, int batch) {
    constexpr int CB = 16 / W;
    __shared__ uint32_t part[kQcRowSplit - 1][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f = blockIdx.x * 64 + lane;
    const int cg = blockIdx.y;
    const bool live = f < batch;
    const int rchunk = (brows + kQcRowSplit - 1) / kQcRowSplit;
    const int r_begin = __builtin_amdgcn_readfirstlane(wave * rchunk), r_end = min(brows, r_begin + rchunk);
    const uint32_t *mw = msgw + (size_t)(live ? f : 0) * kwords;
    const uint32_t *t = rot + (size_t)cg * brows * (32 * 16);
    uint32_t acc[CB][W];
#pragma unroll
    for (int c = 0; c < CB; c++)
#pragma unroll
        for (int w = 0; w < W; w++) acc[c][w] = 0u;
    for (int r = r_begin; r < r_end; r++) {
        uint32_t v[W];
#pragma unroll
        for (int a = 0; a < W; a++) v[a] = mw[r * W + a];
#pragma unroll
        for (int b0 = 0; b0 < 32; b0 += 4) {
            uint32_t T[4][16];       // four rotations' worth of table in flight (64 SGPRs): one wait per 4 x W masked regions
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int i = 0; i < 16; i++) T[q][i] = t[(r * 32 + b0 + q) * 16 + i];   // uniform address: scalar loads
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int a = 0; a < W; a++)
                    if ((v[a] >> (b0 + q)) & 1u) {   // the lanes (frames) whose message bit 32a + b of this block row is set: EXEC mask, one v_xor per word
                        asm volatile("" ::: "memory");   // keeps this a real EXEC-masked region (otherwise: v_cndmask + v_xor per word)
#pragma unroll
                        for (int c = 0; c < CB; c++)
#pragma unroll
                            for (int w = 0; w < W; w++) acc[c][w] ^= T[q][c * W + ((w - a) & (W - 1))];
                    }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int c = 0; c < CB; c++)
#pragma unroll
            for (int w = 0; w < W; w++) part[wave - 1][c * W + w][lane] = acc[c][w];
    }
    __syncthreads();
    if (wave > 0 || !live) return;
#pragma unroll
    for (int c = 0; c < CB; c++)
#pragma unroll
        for (int w = 0; w < W; w++)
#pragma unroll
            for (int q = 0; q < kQcRowSplit - 1; q++) acc[c][w] ^= part[q][c * W + w][lane];
#pragma unroll
    for (int c = 0; c < CB; c++) {
        const int bc = cg * CB + c;
        if (bc < bcols) {
#pragma unroll
            for (int w = 0; w < W; w++) parw[(size_t)f * pwords + bc * W + w] = acc[c][w];
        }
    }
}

// one thread per (frame, group of four consecutive positions n = 4g .. 4g+3): ONE Philox call feeds both
// Box-Muller pairs (r0,r1 -> cos and sin branch, r2,r3 likewise), i.e. four normals -- the generator is bound by
// Philox's quarter-rate 32x32 multiplies, and the first version spent a whole call per sample.
// OT = float / __half: LLRs [batch][N]; OT = uint8_t: the codeword itself, bytes [batch][n_tx], no channel (the encoder alone)
template <typename OT, bool VEC>
__global__ __launch_bounds__(256) void sim_frame_kernel(SimDev s, const uint32_t *__restrict__ msgw, const uint32_t *__restrict__ parw, OT *__restrict__ llr,
                                                        uint8_t *__restrict__ msg_bytes, uint64_t seed, uint64_t first_frame,
                                                        int batch, float sigma, float llr_scale) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    const int n0 = 4 * g;
    if (n0 >= s.N || f >= batch) return;
    const uint32_t *mw = msgw + (size_t)f * s.kwords;
    constexpr bool kBytes = sizeof(OT) == 1;
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (!kBytes && n0 < s.n_tx) {
        uint32_t r[4];
        Philox::gen(seed, first_frame + f, (uint32_t)g, 1u, r);
        // Box-Muller on two 32-bit uniforms (u1 in (0,1]); both branches of each pair are used
        const float ua = ((float)r[0] + 1.0f) * 2.3283064365386963e-10f, ub = (float)r[1] * 2.3283064365386963e-10f;
        const float uc = ((float)r[2] + 1.0f) * 2.3283064365386963e-10f, ud = (float)r[3] * 2.3283064365386963e-10f;
        const float ra = sqrtf(-2.0f * logf(ua)), rc = sqrtf(-2.0f * logf(uc));
        float sa, ca, sc, cc;
        sincospif(2.0f * ub, &sa, &ca);
        sincospif(2.0f * ud, &sc, &cc);
        z[0] = ra * ca; z[1] = ra * sa; z[2] = rc * cc; z[3] = rc * sc;
    }
    // parity bits of the group: bit j = <msg, column j of G> over GF(2); a group that lies inside the parity part
    // on a 4-aligned column reads its four columns with one 16-byte load per message word
    uint32_t pacc[4] = {0u, 0u, 0u, 0u};
    if (s.gt && !s.qc_rot && n0 + 3 >= s.k && n0 < s.n_tx) {
        const int j0 = n0 - s.k;
        if (j0 >= 0 && (j0 & 3) == 0) {
            const uint4 *col = reinterpret_cast<const uint4 *>(s.gt + j0);
            const size_t stride = (size_t)s.pp / 4;
            for (int w = 0; w < s.kwords; w++) {
                const uint32_t m = mw[w];
                const uint4 c = col[(size_t)w * stride];
                pacc[0] ^= m & c.x; pacc[1] ^= m & c.y; pacc[2] ^= m & c.z; pacc[3] ^= m & c.w;
            }
        } else {
            for (int w = 0; w < s.kwords; w++) {
                const uint32_t m = mw[w];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int j = j0 + i;
                    if (j >= 0 && j < s.pp) pacc[i] ^= m & s.gt[(size_t)w * s.pp + j];
                }
            }
        }
    }
    float out[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int n = n0 + i;
        out[i] = 0.f;
        if (n < s.n_tx) {
            uint32_t bit;
            if (n < s.k) {
                bit = (mw[n >> 5] >> (n & 31)) & 1u;
                if (msg_bytes) msg_bytes[(size_t)f * s.k + n] = (uint8_t)bit;
            } else if (s.qc_rot) {
                const int j = n - s.k;         // packed by sim_parity_qc_kernel
                bit = (parw[(size_t)f * s.pwords + (j >> 5)] >> (j & 31)) & 1u;
            } else {
                bit = __popc(pacc[i]) & 1u;   // (no generator: pacc = 0, the all-zero codeword)
            }
            out[i] = llr_scale * ((bit ? 1.0f : -1.0f) + sigma * z[i]);
        } else if (n < s.k && msg_bytes) {
            msg_bytes[(size_t)f * s.k + n] = (uint8_t)((mw[n >> 5] >> (n & 31)) & 1u);
        }
    }
    if constexpr (kBytes) {
#pragma unroll
        for (int i = 0; i < 4; i++) if (n0 + i < s.n_tx) llr[(size_t)f * s.n_tx + n0 + i] = out[i] > 0.f ? 1 : 0;   // (sigma = 0: out = +-scale)
        return;
    }
    OT *dst = llr + (size_t)f * s.N + n0;
    if constexpr (kBytes) {
    } else if constexpr (sizeof(OT) == 2) {
        __half h[4];
#pragma unroll
        for (int i = 0; i < 4; i++) h[i] = __float2half_rn(fminf(fmaxf(out[i], -65504.f), 65504.f));  // = round_f16
        if constexpr (VEC) {
            *reinterpret_cast<uint2 *>(dst) = *reinterpret_cast<const uint2 *>(h);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) if (n0 + i < s.N) dst[i] = h[i];
        }
    } else {
        if constexpr (VEC) {
            *reinterpret_cast<float4 *>(dst) = make_float4(out[0], out[1], out[2], out[3]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) if (n0 + i < s.N) dst[i] = out[i];
        }
    }
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

int sim_generate(const SimDev &s, uint32_t *msgw, uint32_t *parw, hipStream_t st, uint64_t seed, uint64_t first_frame, int batch,
                 double ebn0_db, void *d_out, int out_fmt, uint8_t *d_msg) {
    const double R = (double)s.k / (double)s.n_tx;
    const double sigma2 = 1.0 / (2.0 * R * pow(10.0, ebn0_db / 10.0));
    size_t nw = (size_t)batch * s.kwords;
    hipLaunchKernelGGL(sim_msg_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, msgw, s.kwords, s.k, seed, first_frame, batch, (s.gt || s.qc_rot) ? 0 : 1);
    if (s.qc_rot) {
        const dim3 pg((batch + 63) / 64, s.qc_ncg);
#define LDPC_QC_PARITY(W_) hipLaunchKernelGGL((sim_parity_qc_kernel<W_>), pg, dim3(64 * kQcRowSplit), 0, st, s.qc_rot, s.qc_brows, s.qc_bcols, msgw, parw, s.kwords, s.pwords, batch)
        switch (s.qc_w) {
            case 1: LDPC_QC_PARITY(1); break;
            case 2: LDPC_QC_PARITY(2); break;
            case 4: LDPC_QC_PARITY(4); break;
            case 8: LDPC_QC_PARITY(8); break;
            default: return set_error(LDPC_EUNSUPPORTED, "quasi-cyclic encoder: circulant size %d", 32 * s.qc_w);
        }
#undef LDPC_QC_PARITY
    }
    const int groups = ((out_fmt == 2 ? s.n_tx : s.N) + 3) / 4;
    const dim3 grid((groups + 255) / 256, batch);
    const float sg = (float)sqrt(sigma2), sc = (float)(2.0 / sigma2);
    // 16-byte (f32) / 8-byte (fp16) vector stores when every row starts aligned
    const bool vec = (s.N % 4 == 0) && ((uintptr_t)d_out % 16 == 0);
    if (out_fmt == 2) {
        hipLaunchKernelGGL((sim_frame_kernel<uint8_t, false>), grid, dim3(256), 0, st, s, msgw, parw, (uint8_t *)d_out, d_msg, seed, first_frame, batch, 0.f, 1.f);
    } else if (out_fmt == 1) {
        if (vec) hipLaunchKernelGGL((sim_frame_kernel<__half, true>), grid, dim3(256), 0, st, s, msgw, parw, (__half *)d_out, d_msg, seed, first_frame, batch, sg, sc);
        else hipLaunchKernelGGL((sim_frame_kernel<__half, false>), grid, dim3(256), 0, st, s, msgw, parw, (__half *)d_out, d_msg, seed, first_frame, batch, sg, sc);
    } else {
        if (vec) hipLaunchKernelGGL((sim_frame_kernel<float, true>), grid, dim3(256), 0, st, s, msgw, parw, (float *)d_out, d_msg, seed, first_frame, batch, sg, sc);
        else hipLaunchKernelGGL((sim_frame_kernel<float, false>), grid, dim3(256), 0, st, s, msgw, parw, (float *)d_out, d_msg, seed, first_frame, batch, sg, sc);
    }
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

// ---- packed result bits (ldpc_decode_batch_dev_packed): eight result bytes -> one byte, LSB first
__global__ __launch_bounds__(256) void pack_bits_kernel(const uint8_t *__restrict__ bits, uint8_t *__restrict__ packed, int batch, int N, int PB) {
    const int f = blockIdx.y;
    const uint8_t *src = bits + (size_t)f * N;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < PB; j += gridDim.x * blockDim.x) {
        uint32_t b = 0;
        if ((N & 7) == 0) {                                   // every frame starts on an 8-byte boundary of a 16-byte aligned buffer
            const uint2 v = *reinterpret_cast<const uint2 *>(src + 8 * (size_t)j);
            auto nib = [](uint32_t w) { return (w & 1u) | ((w >> 7) & 2u) | ((w >> 14) & 4u) | ((w >> 21) & 8u); };
            b = nib(v.x) | (nib(v.y) << 4);
        } else {
#pragma unroll
            for (int t = 0; t < 8; t++)
                if (8 * j + t < N) b |= (uint32_t)(src[8 * (size_t)j + t] & 1u) << t;
        }
        packed[(size_t)f * PB + j] = (uint8_t)b;
    }
}

int pack_bits(hipStream_t st, const uint8_t *d_bits, uint8_t *d_packed, int batch, int N) {
    const int PB = (N + 7) / 8;
    hipLaunchKernelGGL(pack_bits_kernel, dim3((PB + 255) / 256, batch), dim3(256), 0, st, d_bits, d_packed, batch, N, PB);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error(LDPC_EHIP, "pack_bits: %s", hipGetErrorString(e));
    return LDPC_OK;
}

}  // namespace ldpc
