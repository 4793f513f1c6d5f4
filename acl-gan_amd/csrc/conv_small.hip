// conv_small.hip -- direct (VALU) kernels for the 64 -> 4 channel 7x7 output convolution of the
// decoder (reference networks.py:260: Conv2dBlock(dim, output_dim=4, 7, 1, 3, norm none, tanh)).
//
// With Cout = 4 the implicit-GEMM MFMA kernels waste 7/8 of a 32-wide N tile (measured 14 TFLOP/s
// effective forward, 7 TFLOP/s wgrad).  On gfx950 the fp32 VALU FMA peak equals the fp32 MFMA peak
// (157 TFLOP/s), so a direct kernel with the 4 output channels as per-lane accumulators and the
// weights as wave-uniform scalars is the right tool:
//   forward : lane = output pixel, input patch (tile + 6 halo, 16 channels at a time) in LDS,
//             read back as b128; weights through the scalar cache; 4*16 FMAs per 4 LDS reads.
//   wgrad   : lane = input channel, one workgroup slab = (filter row ky, a band of padded rows);
//             every loaded activation feeds 7 taps x 4 couts = 28 FMAs with dy as scalars.
#include "common.h"
#include <cstdlib>

namespace aclgan {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int refl(int v, int n) {
    v = v < 0 ? -v : v;
    return v >= n ? 2 * (n - 1) - v : v;
}
__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == ACLGAN_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACLGAN_ACT_LRELU) return v > 0.f ? v : 0.2f * v;
    if (act == ACLGAN_ACT_TANH) return tanhf(v);
    return v;
}

// ---------------- forward: Cout == 4, stride 1, no upsample, Cin % 16 == 0 ----------------
constexpr int TH = 8, TW = 32;        // output tile (256 threads, one pixel each)
constexpr int PIXS = 20;              // LDS pixel stride in floats (16 + 4: conflict-free b128 reads)

template <int K>
__global__ void __launch_bounds__(256) conv_fwd_co4_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           int H, int W, int Ci, int act, int tiles_x, int tiles_y) {
    constexpr int P = K / 2, PH = TH + K - 1, PW = TW + K - 1;
    __shared__ __attribute__((aligned(16))) float patch[PH * PW * PIXS];
    const int tid = threadIdx.x;
    const int b = blockIdx.x / (tiles_x * tiles_y), t = blockIdx.x % (tiles_x * tiles_y);
    const int ty0 = (t / tiles_x) * TH, tx0 = (t % tiles_x) * TW;
    const int ly = tid / TW, lx = tid % TW;
    const int oy = ty0 + ly, ox = tx0 + lx;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    const int KK = K * K * Ci;   // weight row length (OHWI)
    for (int c0 = 0; c0 < Ci; c0 += 16) {
        __syncthreads();
        // stage the (reflect-padded) input patch, 16 channels: PH*PW pixels x 4 float4
        for (int i = tid; i < PH * PW * 4; i += 256) {
            const int pix = i >> 2, q = i & 3;
            const int py = pix / PW, px = pix - py * PW;
            // clamp after reflecting: halo pixels of tiles that overhang the image feed only outputs that are never stored
            const int iy = min(max(refl(ty0 + py - P, H), 0), H - 1), ix = min(max(refl(tx0 + px - P, W), 0), W - 1);
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((size_t)(b * H + iy) * W + ix) * Ci + c0 + q * 4);
            *reinterpret_cast<f32x4*>(patch + pix * PIXS + q * 4) = v;
        }
        __syncthreads();
        for (int ky = 0; ky < K; ++ky) {
#pragma unroll 1
            for (int kx = 0; kx < K; ++kx) {
                const float* pp = patch + ((ly + ky) * PW + lx + kx) * PIXS;
                const float* wp = w + (ky * K + kx) * Ci + c0;     // wave-uniform -> scalar loads
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(pp + q * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xs = xv[j];
                        acc0 = fmaf(xs, wp[q * 4 + j], acc0);
                        acc1 = fmaf(xs, wp[KK + q * 4 + j], acc1);
                        acc2 = fmaf(xs, wp[2 * KK + q * 4 + j], acc2);
                        acc3 = fmaf(xs, wp[3 * KK + q * 4 + j], acc3);
                    }
                }
            }
        }
    }
    if (oy < H && ox < W) {
        f32x4 o;
        o[0] = act_apply(acc0 + (bias ? bias[0] : 0.f), act);
        o[1] = act_apply(acc1 + (bias ? bias[1] : 0.f), act);
        o[2] = act_apply(acc2 + (bias ? bias[2] : 0.f), act);
        o[3] = act_apply(acc3 + (bias ? bias[3] : 0.f), act);
        *reinterpret_cast<f32x4*>(y + ((size_t)(b * H + oy) * W + ox) * 4) = o;
    }
}

// ---------------- wgrad: Cout == 4, Cin == 64, stride 1, no upsample ----------------
// grid: x = band of padded rows, y = ky, z = image b.  wave w of the workgroup takes padded columns
// qx = w, w+4, ...; lane = cin.  acc[kx][co] += x[refl(qy), refl(qx)][cin] * dy[qy-ky][qx-kx][co]
template <int K>
__global__ void __launch_bounds__(256) conv_wgrad_co4_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                                             int H, int W, int rows_per_band) {
    constexpr int P = K / 2;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably uniform -> scalar dy loads
    const int ky = blockIdx.y, b = blockIdx.z;
    const int Hp = H + 2 * P, Wp = W + 2 * P;
    const int q0 = blockIdx.x * rows_per_band, q1 = min(Hp, q0 + rows_per_band);
    float acc[K][4];
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][c] = 0.f;
    for (int qy = q0; qy < q1; ++qy) {
        const int oy = qy - ky;                       // output row fed by padded row qy through filter row ky
        if (oy < 0 || oy >= H) continue;              // wave-uniform
        const int iy = refl(qy - P, H);
        const float* xrow = x + (size_t)(b * H + iy) * W * 64;
        const float* drow = dy + (size_t)(b * H + oy) * W * 4;
        for (int qx = wave; qx < Wp; qx += 4) {
            const float xv = xrow[(size_t)refl(qx - P, W) * 64 + lane];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int ox = qx - kx;               // wave-uniform
                if (ox >= 0 && ox < W) {
                    const float* d = drow + ox * 4;   // scalar loads
                    acc[kx][0] = fmaf(xv, d[0], acc[kx][0]);
                    acc[kx][1] = fmaf(xv, d[1], acc[kx][1]);
                    acc[kx][2] = fmaf(xv, d[2], acc[kx][2]);
                    acc[kx][3] = fmaf(xv, d[3], acc[kx][3]);
                }
            }
        }
    }
    // combine the 4 waves through LDS, then one atomic per (co, kx, cin) per workgroup
    __shared__ float red[4][K * 4][64];
#pragma unroll
    for (int kx = 0; kx < K; ++kx)
#pragma unroll
        for (int c = 0; c < 4; ++c) red[wave][kx * 4 + c][lane] = acc[kx][c];
    __syncthreads();
    for (int i = threadIdx.x; i < K * 4 * 64; i += 256) {
        const int e = i >> 6, l = i & 63;
        const float s = (red[0][e][l] + red[1][e][l]) + (red[2][e][l] + red[3][e][l]);
        const int kx = e >> 2, c = e & 3;
        atomicAdd(dw + ((size_t)(c * K + ky) * K + kx) * 64 + l, s);
    }
}

// ---------------- wide (Cin % 8 == 0) -> thin (<= 4 channels) 7x7 / stride 1, on v_mfma_f32_4x4x1 ----------------
//   MODE 0: forward with reflect pad 3 (DO, networks.py:260): out = H x W x CN, B operand = w[co][tap][ci]
//   MODE 1: dgrad of a thin-input layer (CE0 / SE0, networks.py:216,234) onto the PADDED grid: in = dy (H x W x Cout),
//           read zero-extended by 6; out = (H+6) x (W+6) x CN; B operand = w[k][48 - tap][n] (flipped taps, transposed);
//           the reflection fold (conv_fold_kernel) follows, exactly as after the general dgrad kernel
// lane = output pixel of an 8x32 tile (A operand: its input value), lane & 3 = thin channel (B operand); the result
// register v of lane l is out[pixel of lane 4*(l/4)+v][channel l&3] (layout: scripts/microbench/mfma4x4_layout.hip).
// Input patch and weights are staged 8 channels at a time (patch stride 12 floats: conflict-free b128 reads).
template <int MODE>
__global__ void __launch_bounds__(256) conv_thin_out_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int H, int W, int Cw, int CN, int act, int tiles_x, int tiles_y) {
    constexpr int K = 7, P = MODE ? 6 : 3, PH = TH + K - 1, PW = TW + K - 1, PS = 12, CH = 8;
    __shared__ __attribute__((aligned(16))) float patch[PH * PW * PS];
    __shared__ __attribute__((aligned(16))) float wts[K * K * 4 * CH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / (tiles_x * tiles_y), t = blockIdx.x % (tiles_x * tiles_y);
    const int ty0 = (t / tiles_x) * TH, tx0 = (t % tiles_x) * TW;
    const int Ho = MODE ? H + 6 : H, Wo = MODE ? W + 6 : W;
    const int ly = 2 * wave + (lane >> 5), lx = lane & 31;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Round 6: the staging of 8-channel chunk c + 1 is FETCHED into registers before the tap loop of chunk c and written to LDS after it (until
    // then every chunk exposed one global-load latency in front of its 392 MFMAs; profiles/r06_pmc_thin_layers.txt: waves parked 36 %).  The
    // gather offsets do not depend on the chunk: computed once.
    constexpr int NPP = (PH * PW * 2 + 255) / 256, NWP = (K * K * 4 * 2 + 255) / 256;
    int poff[NPP], woff[NWP];
#pragma unroll
    for (int j = 0; j < NPP; ++j) {
        const int i = tid + 256 * j;
        poff[j] = -1;
        if (i < PH * PW * 2) {
            const int pix = i >> 1, q = i & 1;
            const int py = pix / PW, px = pix - py * PW;
            const int gy = ty0 + py - P, gx = tx0 + px - P;
            if (MODE == 0) {
                // clamp after reflecting: halo pixels of tiles that overhang the image feed only outputs that are never stored
                const int iy = min(max(refl(gy, H), 0), H - 1), ix = min(max(refl(gx, W), 0), W - 1);
                poff[j] = ((b * H + iy) * W + ix) * Cw + q * 4;
            } else if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                poff[j] = ((b * H + gy) * W + gx) * Cw + q * 4;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NWP; ++j) {
        const int i = tid + 256 * j;
        woff[j] = -1;
        if (i < K * K * 4 * 2) {
            const int q = i & 1, n = (i >> 1) & 3, tap = i >> 3;
            if (n < CN) woff[j] = MODE == 0 ? (n * K * K + tap) * Cw + q * 4 : (q * 4 * K * K + (K * K - 1 - tap)) * CN + n;
        }
    }
    f32x4 pr[NPP], wr[NWP];
    auto fetch = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NPP; ++j) {
            pr[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (poff[j] >= 0) pr[j] = *reinterpret_cast<const f32x4*>(in + (size_t)poff[j] + c0);
        }
#pragma unroll
        for (int j = 0; j < NWP; ++j) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (woff[j] >= 0) {
                if (MODE == 0) v = *reinterpret_cast<const f32x4*>(w + (size_t)woff[j] + c0);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = w[(size_t)woff[j] + (size_t)(c0 + e) * K * K * CN];
                }
            }
            wr[j] = v;
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < Cw; c0 += CH) {
        __syncthreads();                                  // the previous chunk's taps have read patch / wts
#pragma unroll
        for (int j = 0; j < NPP; ++j) {
            const int i = tid + 256 * j;
            if (i < PH * PW * 2) *reinterpret_cast<f32x4*>(patch + (i >> 1) * PS + (i & 1) * 4) = pr[j];
        }
#pragma unroll
        for (int j = 0; j < NWP; ++j) {
            const int i = tid + 256 * j;
            if (i < K * K * 4 * 2) *reinterpret_cast<f32x4*>(wts + ((i >> 3) * 4 + ((i >> 1) & 3)) * CH + (i & 1) * 4) = wr[j];
        }
        __syncthreads();
        if (c0 + CH < Cw) fetch(c0 + CH);                 // in flight under this chunk's tap loop
        const float* pa = patch + (ly * PW + lx) * PS;
        const float* pb = wts + (lane & 3) * CH;
#pragma unroll 1
        for (int ky = 0; ky < K; ++ky) {
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(pa + (ky * PW + kx) * PS);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(pa + (ky * PW + kx) * PS + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(pb + (ky * K + kx) * 4 * CH);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(pb + (ky * K + kx) * 4 * CH + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[e], b0[e], acc[e], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[e], b1[e], acc[e], 0, 0, 0);
            }
        }
    }
    const f32x4 r = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    const int co = lane & 3;
    if (co < CN) {
        const float bv = bias ? bias[co] : 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int pl = 4 * (lane >> 2) + v;                 // the lane whose pixel this register belongs to
            const int oy = ty0 + 2 * wave + (pl >> 5), ox = tx0 + (pl & 31);
            if (oy < Ho && ox < Wo) out[((size_t)(b * Ho + oy) * Wo + ox) * CN + co] = act_apply(r[v] + bv, act);
        }
    }
}

// ---------------- thin (CS = 3 or 4 channels) -> 64 channels, 7x7 / stride 1 / reflect pad 3, on v_mfma_f32_32x32x2 ----------------
// The forward of CE0 / SE0 (networks.py:216,234).  (The same scheme was tried for the dgrad of the 64 -> 4 layer, K = 196 onto
// the padded grid: 0.28 ms against 0.24 ms for the general kernel + fold -- not kept.)
// GEMM per 8x32-pixel tile: M = 256 pixels (wave = 2 image rows = 2 M-tiles), N = 64, K = 49*CS (147 / 196).  With so few
// input channels an im2col tile would be all index math; instead the (reflect / zero padded) input patch sits in LDS
// once and every A fragment is a ds_read_b32 at  pixel_base + const(k): k = (tap, c) -> patch offset k + ky*(PW-7)*CS,
// a compile-time immediate (the K loop is fully unrolled); the two lane halves (k, k+1) differ by 1 float except where
// k+1 starts a new filter row, handled by a second per-lane base.  Weights: LDS [k][64+1].
// __launch_bounds__ second argument = min waves per SIMD (what LDS allows): caps the VGPRs the fully unrolled K loop may take
template <int CS>
__global__ void __launch_bounds__(256, CS == 3 ? 3 : 2) conv_thin_in_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           int H, int W, int act, int tiles_x, int tiles_y) {
    constexpr int K = 7, P = 3, PH = TH + K - 1, PW = TW + K - 1;
    constexpr int KT = K * K * CS, KS = (KT + 1) / 2, LDW = 65, ROWJ = (PW - K) * CS;   // ROWJ: extra patch offset per filter row
    __shared__ float patch[(PH + 1) * PW * CS];          // + one zero row: the odd-K pad element reads it
    __shared__ float wl[2 * KS * LDW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int Ho = H, Wo = W;
    // weights -> LDS [k][co], zero row for the K pad
    for (int i = tid; i < 2 * KS * 64; i += 256) {
        const int co = i / (2 * KS), k = i - co * (2 * KS);          // consecutive threads walk k: coalesced along OHWI's (tap, c)
        wl[k * LDW + co] = k < KT ? w[(size_t)co * KT + k] : 0.f;
    }
    for (int tt = 0; tt < 2; ++tt) {                                 // two vertically adjacent tiles per workgroup (weights staged once)
        const int tyi = (blockIdx.x % ((tiles_y + 1) / 2)) * 2 + tt;
        const int rest = blockIdx.x / ((tiles_y + 1) / 2);
        const int txi = rest % tiles_x, b = rest / tiles_x;
        if (tyi >= tiles_y) break;                                   // block-uniform
        const int ty0 = tyi * TH, tx0 = txi * TW;
        __syncthreads();
        for (int i = tid; i < (PH + 1) * PW * CS; i += 256) {
            const int pix = i / CS, c = i - pix * CS;
            const int py = pix / PW, px = pix - py * PW;
            const int gy = ty0 + py - P, gx = tx0 + px - P;
            float v = 0.f;
            if (py < PH) {
                // clamp after reflecting: halo pixels of tiles that overhang the image feed only outputs that are never stored
                const int iy = min(max(refl(gy, H), 0), H - 1), ix = min(max(refl(gx, W), 0), W - 1);
                v = in[((size_t)(b * H + iy) * W + ix) * CS + c];
            }
            patch[i] = v;
        }
        __syncthreads();
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const float* pN = patch + ((2 * wave) * PW + l31) * CS + kh;              // lane half kh reads k+kh: +1 float ...
        const float* pX = patch + ((2 * wave) * PW + l31) * CS + kh * (1 + ROWJ);  // ... or +1 plus the row jump when k+1 opens a filter row
        const float* pb = wl + kh * LDW + l31;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k0 = 2 * s;
            const int off0 = k0 + (k0 / (K * CS)) * ROWJ;                        // compile-time after unrolling
            const bool jump = ((k0 + 1) % (K * CS)) == 0;
            const float* pa = jump ? pX : pN;
            const float a0 = pa[off0], a1 = pa[off0 + PW * CS];
            const float b0 = pb[k0 * LDW], b1 = pb[k0 * LDW + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = l31 + 32 * j;
            const float bv = bias ? bias[co] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int oy = ty0 + 2 * wave + i;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ox = tx0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (oy < Ho && ox < Wo) out[((size_t)(b * Ho + oy) * Wo + ox) * 64 + co] = act_apply(acc[i][j][r] + bv, act);
                }
            }
        }
    }
}

// ---------------- wgrad with one THIN side (<= 4 channels) and one 64-channel side, 7x7 / stride 1 / pad 3 ----------------
//   WIDE_X = true : Cin = 64, Cout = 4   (DO, networks.py:260): wide = x gathered at the tap-shifted position, thin = dy
//   WIDE_X = false: Cout = 64, Cin <= 4  (CE0 / SE0, networks.py:216,234): wide = dy, thin = x at the tap-shifted position
// v_mfma_f32_4x4x1_16B_f32 is 16 independent 4x4 outer products per wave (measured layout,
// scripts/microbench/mfma4x4_layout.hip: D[lane l][reg v] += A[lane 4*(l/4)+v] * B[lane l]).  With lane = wide
// channel as the A operand and the thin tensor's channel (lane & 3) as the B operand, one instruction accumulates
// dW[wide channel 4*(l/4)+v][thin channel l&3] for one (pixel, tap): the full MFMA rate with N = 4, where the
// 32-wide MFMA tiles waste 7/8 and the scalar-operand VALU version reached 16 TFLOP/s.
//
// Workgroup = 7 waves, wave = filter row ky (its 7 kx accumulators never leave its registers: no cross-wave
// reduction), grid = (band of output rows, column segment, image).  Every wide row segment (70 positions x 64
// channels) is staged in LDS ONCE and consumed by all 7 waves (the first version gave each ky its own workgroup
// and re-read the wide tensor 7x through a thrashing L2: 0.51 ms); the thin rows of the band are staged once per
// workgroup, zero / reflect padded so that the 7 kx taps are a sliding window (one new LDS read per position).
constexpr int THIN_SEG = 70;                 // positions per column segment (10 trips of 7: the window rotation is static)
constexpr int THIN_PX = THIN_SEG + 6;        // thin pixels a segment touches
constexpr int THIN_ROWS_X = 16;              // band height, WIDE_X  (wide rows walked: band + 6, thin rows staged: band)
constexpr int THIN_ROWS_DY = 8;              // band height, !WIDE_X (wide rows walked: band, thin rows staged: band + 6)

template <bool WIDE_X>
__global__ void __launch_bounds__(448) conv_wgrad_thin_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ dw, float* __restrict__ db,
                                                              float* __restrict__ partial, int H, int W, int CN) {
    constexpr int K = 7, P = 3, NT = 448;
    constexpr int R = WIDE_X ? THIN_ROWS_X : THIN_ROWS_DY;
    constexpr int TR = WIDE_X ? R : R + 6;                     // thin rows held
    __shared__ float wide[2][THIN_SEG * 64];
    __shared__ float thin[TR][THIN_PX * 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int ky = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave = filter row
    const int b = blockIdx.z;
    const int oy0 = blockIdx.x * R, oy1 = min(H, oy0 + R);
    const int s0 = blockIdx.y * THIN_SEG;                      // first position of this column segment
    const int AW = WIDE_X ? W + 2 * P : W;                     // positions of a wide row
    const int npos = min(THIN_SEG, AW - s0);
    const int c4 = lane & 3;

    // thin rows of the band (this segment's 76 pixels), channel-padded to 4
    for (int i = tid; i < TR * THIN_PX * 4; i += NT) {
        const int r = i / (THIN_PX * 4), rem = i - r * (THIN_PX * 4);
        const int jj = rem >> 2, c = rem & 3;
        float v = 0.f;
        if (WIDE_X) {                                           // thin = dy row oy0 + r, zero outside the image
            const int oy = oy0 + r, ox = s0 - 6 + jj;
            if (c < CN && oy < oy1 && ox >= 0 && ox < W) v = dy[((size_t)(b * H + oy) * W + ox) * CN + c];
        } else {                                                // thin = x row refl(oy0 + r - 3), reflect padded columns
            // rows past the band's last tap row are never read: do not form their (possibly twice-reflected) address
            if (c < CN && r < (oy1 - oy0) + 2 * P) {
                const int iy = refl(oy0 + r - P, H), ix = refl(min(s0 + jj, W + 2 * P - 1) - P, W);
                v = x[((size_t)(b * H + iy) * W + ix) * CN + c];
            }
        }
        thin[r][rem] = v;
    }
    // wide rows this workgroup walks: WIDE_X: padded rows qy = oy0 .. oy1+5 (x row refl(qy-3)); else output rows oy0 .. oy1-1 (dy)
    const int nwide = WIDE_X ? (oy1 - oy0) + 2 * P : (oy1 - oy0);
    // Round 6: the next wide row is FETCHED into registers before the MFMA loop of the current one and written to LDS after it.  (Until then the
    // global load and its LDS store were one statement in front of the loop: every wide row exposed one global-load latency -- MFMA pipe 55 % busy,
    // waves parked 33 %, profiles/r06_pmc_thin_layers.txt.)  THIN_SEG * 16 float4 over 448 threads: 3 per thread.
    constexpr int NWR = (THIN_SEG * 16 + NT - 1) / NT;
    f32x4 wreg[NWR];
    auto fetch_wide = [&](int wr) __attribute__((always_inline)) {
        const int row = WIDE_X ? refl(oy0 + wr - P, H) : oy0 + wr;
        const float* src = (WIDE_X ? x : dy) + (size_t)(b * H + row) * W * 64;
#pragma unroll
        for (int j = 0; j < NWR; ++j) {
            const int i = tid + j * NT;
            if (i < npos * 16) {
                const int t = i >> 4, q = i & 15;
                const int col = WIDE_X ? refl(s0 + t - P, W) : s0 + t;
                wreg[j] = *reinterpret_cast<const f32x4*>(src + (size_t)col * 64 + q * 4);
            }
        }
    };
    auto store_wide = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NWR; ++j) {
            const int i = tid + j * NT;
            if (i < npos * 16) *reinterpret_cast<f32x4*>(&wide[buf][(i >> 4) * 64 + (i & 15) * 4]) = wreg[j];
        }
    };

    f32x4 acc[K];
#pragma unroll
    for (int i = 0; i < K; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    if (nwide > 0 && npos > 0) { fetch_wide(0); store_wide(0); }
    __syncthreads();
    for (int wr = 0; wr < nwide; ++wr) {
        const int buf = wr & 1;
        if (wr + 1 < nwide) fetch_wide(wr + 1);                  // in flight under this row's MFMAs; stored below, published by the barrier
        // the thin row this wave (ky) pairs with wide row wr
        const int tr = WIDE_X ? wr - ky : wr + ky;               // WIDE_X: oy - oy0 = (qy - ky) - oy0
        const bool live = WIDE_X ? (tr >= 0 && tr < oy1 - oy0) : true;   // wave-uniform
        if (live && npos > 0) {
            const float* nb = thin[tr] + c4;
            const float* wb = wide[buf] + lane;
            float win[K];
            // slot (j mod 7) of the window holds thin pixel j of this segment: WIDE_X position t uses pixels t+6-kx, else t+kx
#pragma unroll
            for (int jj = 0; jj < K - 1; ++jj) win[jj] = nb[jj * 4];
            // NO per-position branch in here (a branch per position made hipcc split the loop into 70 basic blocks, each
            // with an exposed LDS wait and 28 accumulator copies: 0.56 ms): positions past the row get a = 0 instead.
#pragma unroll 2
            for (int t0 = 0; t0 < THIN_SEG; t0 += K) {
#pragma unroll
                for (int u = 0; u < K; ++u) {
                    const int t = t0 + u;
                    win[(u + K - 1) % K] = nb[(t + K - 1) * 4];  // newest pixel t+6 (t0 is a multiple of 7: slot = (t+6) mod 7)
                    float a = wb[t * 64];
                    a = t < npos ? a : 0.f;                      // stale LDS past the row end: select, do not multiply
                    if (!WIDE_X) bsum += a;
#pragma unroll
                    for (int kx = 0; kx < K; ++kx)
                        acc[kx] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, win[WIDE_X ? (u + K - 1 + K - kx) % K : (u + kx) % K], acc[kx], 0, 0, 0);
                }
            }
        }
        if (wr + 1 < nwide) store_wide(buf ^ 1);
        __syncthreads();
    }
    // lane l, register v of acc[kx] = dW[wide channel 4*(l/4)+v][thin channel l&3] of tap (ky, kx)
    if (partial != nullptr) {
        // two-stage reduction: thousands of workgroups adding into the same 37-50 KB of weights serialise in the
        // memory-side atomic units (measured: 0.33 of 0.43 ms); plain coalesced stores of the per-workgroup partials
        // [workgroup][ky][kx][lane] (16 B per lane) + conv_wgrad_thin_reduce_kernel instead
        const size_t wg = blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);
        f32x4* pp = reinterpret_cast<f32x4*>(partial) + (wg * K * K + (size_t)ky * K) * 64 + lane;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) pp[kx * 64] = acc[kx];
    } else if (c4 < CN) {
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int widec = 4 * (lane >> 2) + v;
                if (WIDE_X) atomicAdd(dw + ((size_t)(c4 * K + ky) * K + kx) * 64 + widec, acc[kx][v]);     // dw[co = thin][ky][kx][ci = wide]
                else atomicAdd(dw + ((size_t)(widec * K + ky) * K + kx) * CN + c4, acc[kx][v]);            // dw[co = wide][ky][kx][ci = thin]
            }
    }
    if (!WIDE_X && db != nullptr && ky == 0) atomicAdd(db + lane, bsum);   // wave 0 saw every dy value of the segment exactly once
}

// dw += sum over workgroups of partial[wg][tap][lane][v]; grid = (49 taps, groups of workgroups), 256 threads = (lane, v)
// gpart != nullptr (deterministic mode, first level): the group sums are stored as partials of the same layout, [group][tap][256],
// and a second call with ONE group adds them in order -- no two workgroups ever add into the same dw element
template <bool WIDE_X>
__global__ void __launch_bounds__(256) conv_wgrad_thin_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                                     int nwg, int per_group, int CN, float* __restrict__ gpart) {
    constexpr int K = 7;
    const int tap = blockIdx.x, l = threadIdx.x >> 2, v = threadIdx.x & 3;
    const int w0 = blockIdx.y * per_group, w1 = min(nwg, w0 + per_group);
    float s = 0.f;
    const float* pp = partial + (size_t)tap * 256 + threadIdx.x;
    int w = w0;
    for (; w + 8 <= w1; w += 8) {           // eight loads in flight, added in workgroup order (one dependent round trip per partial otherwise)
        float a[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = pp[(size_t)(w + q) * (K * K * 256)];
#pragma unroll
        for (int q = 0; q < 8; ++q) s += a[q];
    }
    for (; w < w1; ++w) s += pp[(size_t)w * (K * K * 256)];
    if (gpart) { gpart[((size_t)blockIdx.y * K * K + tap) * 256 + threadIdx.x] = s; return; }
    const int widec = 4 * (l >> 2) + v, thin = l & 3;
    if (thin < CN) {
        if (WIDE_X) atomicAdd(dw + ((size_t)thin * K * K + tap) * 64 + widec, s);
        else atomicAdd(dw + ((size_t)widec * K * K + tap) * CN + thin, s);
    }
}

// db[c] += sum of dy[pixel][c] for 4-channel maps
__global__ void __launch_bounds__(256) colsum4_kernel(const f32x4* __restrict__ dy, float* __restrict__ db, int64_t npix) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) s += dy[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s[0] += __shfl_xor(s[0], o); s[1] += __shfl_xor(s[1], o); s[2] += __shfl_xor(s[2], o); s[3] += __shfl_xor(s[3], o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(db + 0, s[0]); atomicAdd(db + 1, s[1]); atomicAdd(db + 2, s[2]); atomicAdd(db + 3, s[3]);
    }
}

bool small_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("ACLGAN_NOSMALL"); v = (e && atoi(e)) ? 0 : 1; }
    return v == 1;
}

}  // namespace

// ---------------- round 6: input gradient of the first discriminator layers (4x4 stride 2 pad 1, Cin 3 / 6 <- Cout 64; networks.py:41) ----------------
// onto the PADDED grid dxp [B][H+2][W+2][CN] (the reflection fold follows, as after the general kernel, which ran these layers on 32-column
// tiles: 32 / Cin times the necessary FLOPs, 300 us for 1.6 GFLOP).  A stride-2 transposed convolution splits by the parity (cy, cx) of the
// padded position into four 2x2-tap correlations of dy:  dxp[2m + cy][2n + cx] = sum_{a, b in {0,1}} w[.][cy + 2a][cx + 2b][.] dy[m - a][n - b]
// -- all four classes of a class-grid position (m, n) read the SAME 2x2 neighbourhood of dy, and each of the 16 filter taps is used by exactly
// one (class, a, b).  So: one thread per (m, n), the neighbourhood loaded once (float4 per 4 channels), 16 x CN accumulators, the weights as
// wave-uniform scalars (CN <= 6: 96 per output channel) -- a direct VALU kernel as conv_fwd_co4_kernel above, VALU-bound at 3072 / 6144 FMAs per
// thread (22 / 44 us at 256x256 B=16).  (An MFMA version in the scheme of conv_thin_out_kernel, 4 taps per staged 8-channel chunk, measured
// 145 / 330 us: all staging.)
template <int CN>
__global__ void __launch_bounds__(256) conv_s2k4_thin_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dxp,
                                                                   int B, int Ho, int Wo, int Co, int Hp, int Wp) {
    const int Hc = (Hp + 1) >> 1, Wc = (Wp + 1) >> 1;      // class-grid extent (the largest parity class)
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * Hc * Wc) return;
    const int n = (int)(idx % Wc), m = (int)((idx / Wc) % Hc), b = (int)(idx / ((int64_t)Wc * Hc));
    float acc[4][CN];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < CN; ++i) acc[c][i] = 0.f;
    // neighbourhood pointers (a, b) -> dy[m - a][n - b]; out-of-range taps read zero
    const float* nb[2][2]; bool ok[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            const int y = m - a, x = n - bb;
            ok[a][bb] = y >= 0 && y < Ho && x >= 0 && x < Wo;
            nb[a][bb] = dy + ((size_t)(b * Ho + (ok[a][bb] ? y : 0)) * Wo + (ok[a][bb] ? x : 0)) * Co;
        }
    for (int c0 = 0; c0 < Co; c0 += 4) {
        f32x4 d[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                d[a][bb] = *reinterpret_cast<const f32x4*>(nb[a][bb] + c0);
                if (!ok[a][bb]) d[a][bb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* wc = w + (size_t)(c0 + e) * 16 * CN;      // wave-uniform: scalar loads
#pragma unroll
            for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) {
                    const float dv = d[ky >> 1][kx >> 1][e];      // tap (ky, kx) = class (ky & 1, kx & 1), neighbour (ky >> 1, kx >> 1)
#pragma unroll
                    for (int i = 0; i < CN; ++i) acc[(ky & 1) * 2 + (kx & 1)][i] = fmaf(dv, wc[(ky * 4 + kx) * CN + i], acc[(ky & 1) * 2 + (kx & 1)][i]);
                }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int py = 2 * m + (c >> 1), px = 2 * n + (c & 1);
        if (py < Hp && px < Wp) {
            float* o = dxp + ((size_t)(b * Hp + py) * Wp + px) * CN;
#pragma unroll
            for (int i = 0; i < CN; ++i) o[i] = acc[c][i];
        }
    }
}

// ---------------- round 6: thin (CS = 3 / 4 / 6 channels) -> 64 channels, K x K / stride S / reflect pad P, pipelined ----------------
// conv_thin_in_kernel's GEMM scheme (patch in LDS, every A fragment a ds_read_b32 at pixel_base + const(k)) generalised to the first
// discriminator layers (4x4 stride 2 pad 1, Cin 3 / 6: networks.py:41 -- they ran on the general implicit-GEMM kernel with scalar gathers,
// 59 - 87 us for 67 MB of output) and restructured around what the PMC pass of the 7x7 kernel showed (profiles/r06_experiments.md section 5:
// MFMA pipe 45 % busy, 4.5 VALU instructions per MFMA of which 4 in the staging loops and the store epilogue, waves parked 42 % of the time --
// every workgroup of a CU in its load phase at once):
//   * a workgroup walks `tpw` vertically adjacent tiles with the weights staged ONCE, and fetches the patch of tile t + 1 into registers
//     before the K loop of tile t (the global latency hides behind the MFMAs; LDS holds one patch);
//   * the staging loops carry no integer division by a runtime value (rows per wave, lanes along the contiguous (x, c) run);
//   * the epilogue stores through one row pointer per output row with compile-time column offsets.
// Lane stride of the A reads is S * CS floats (stride 2, 6 channels: a 4-way bank conflict on 4 reads per 256 MFMA cycles: irrelevant).
template <int CS, int K, int S, int P>
__global__ void __launch_bounds__(256, 2) conv_thin_in2_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ out, int H, int W, int Ho, int Wo, int act, int tiles_x, int tiles_y, int tpw) {
    constexpr int PH = S * (TH - 1) + K, PW = S * (TW - 1) + K, ROWE = PW * CS;      // patch rows / columns, floats per patch row
    constexpr int KT = K * K * CS, KS = (KT + 1) / 2, LDW = 65, ROWJ = (PW - K) * CS;
    constexpr int NE = PH * ROWE, NPR = (NE + 255) / 256;                              // patch elements, per thread
    constexpr bool PF = NPR <= 16;      // prefetch the next patch under the K loop (6 channels at stride 2: 28 registers per thread -- spills; its K loop is short)
    __shared__ float patch[NE + ROWE];                   // + one zero row: the odd-K pad element reads it
    __shared__ float wl[2 * KS * LDW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    // weights -> LDS [k][co]: wave w stages couts 16 w .. 16 w + 15, its lanes walk k (coalesced along OHWI's (tap, c)); zero row for the K pad
    for (int cc = 0; cc < 16; ++cc) {
        const int co = wave * 16 + cc;
        for (int k = lane; k < 2 * KS; k += 64) wl[k * LDW + co] = k < KT ? w[(size_t)co * KT + k] : 0.f;
    }
    for (int i = tid; i < ROWE; i += 256) patch[NE + i] = 0.f;
    const int ngy = (tiles_y + tpw - 1) / tpw;
    const int gyi = blockIdx.x % ngy, rest = blockIdx.x / ngy, txi = rest % tiles_x, b = rest / tiles_x;
    const int tx0 = txi * TW;
    const float* inb = in + (size_t)b * H * W * CS;
    float pv[NPR];
    auto fetch = [&](int tyi) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NPR; ++j) {
            const int e = tid + 256 * j;
            if (NE % 256 == 0 || e < NE) {
                const int py = e / ROWE, r = e - py * ROWE, px = r / CS, c = r - px * CS;          // (constant divisors)
                // clamp after reflecting: halo pixels of tiles that overhang the image feed only outputs that are never stored
                const int iy = min(max(refl(S * tyi * TH + py - P, H), 0), H - 1), ix = min(max(refl(S * tx0 + px - P, W), 0), W - 1);
                pv[j] = inb[((size_t)iy * W + ix) * CS + c];
            }
        }
    };
    int tyi = gyi * tpw;
    if (PF && tyi < tiles_y) fetch(tyi);
    const float* pN = patch + S * ((2 * wave) * PW + l31) * CS + kh;              // lane half kh reads k + kh: + 1 float ...
    const float* pX = patch + S * ((2 * wave) * PW + l31) * CS + kh * (1 + ROWJ);  // ... or + 1 plus the row jump when k + 1 opens a filter row
    const float* pb = wl + kh * LDW + l31;
    const float bv0 = bias ? bias[l31] : 0.f, bv1 = bias ? bias[l31 + 32] : 0.f;
    for (int tt = 0; tt < tpw && tyi < tiles_y; ++tt, ++tyi) {
        if (!PF) fetch(tyi);
        __syncthreads();                                  // the previous tile's K loop has read the patch (first pass: the weights are staged)
#pragma unroll
        for (int j = 0; j < NPR; ++j) {
            const int e = tid + 256 * j;
            if (NE % 256 == 0 || e < NE) patch[e] = pv[j];
        }
        __syncthreads();
        if (PF && tt + 1 < tpw && tyi + 1 < tiles_y) fetch(tyi + 1);      // in flight under the K loop
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k0 = 2 * s;
            const int off0 = k0 + (k0 / (K * CS)) * ROWJ;                        // compile-time after unrolling
            const bool jump = ((k0 + 1) % (K * CS)) == 0;
            const float* pa = jump ? pX : pN;
            const float a0 = pa[off0], a1 = pa[off0 + S * PW * CS];
            const float b0 = pb[k0 * LDW], b1 = pb[k0 * LDW + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        const int wlim = Wo - tx0 - 4 * kh;               // columns this lane half may store
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int oy = tyi * TH + 2 * wave + i;
            if (oy >= Ho) continue;                       // (wave-uniform)
            float* orow = out + ((size_t)(b * Ho + oy) * Wo + tx0 + 4 * kh) * 64 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int oxl = (r & 3) + 8 * (r >> 2);
                if (oxl < wlim) {
                    orow[oxl * 64] = act_apply(acc[i][0][r] + bv0, act);
                    orow[oxl * 64 + 32] = act_apply(acc[i][1][r] + bv1, act);
                }
            }
        }
    }
}

static bool thin_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("ACLGAN_NOTHIN"); v = (e && atoi(e)) ? 0 : 1; }
    return v == 1;
}

// tiles per workgroup of conv_thin_in2_kernel: the most that still leaves two workgroups per CU
static int thin_in2_tpw(int B, int tx, int ty) {
    for (int t = 4; t > 1; t >>= 1)
        if ((long long)B * tx * cdiv(ty, t) >= 512) return t;
    return 1;
}
template <int CS, int K, int S, int P>
static int launch_thin_in2(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, hipStream_t st) {
    const int tx = cdiv(g.Wo, TW), ty = cdiv(g.Ho, TH), tpw = thin_in2_tpw(g.B, tx, ty);
    hipLaunchKernelGGL((conv_thin_in2_kernel<CS, K, S, P>), dim3(g.B * tx * cdiv(ty, tpw)), dim3(256), 0, st, x, w, bias, y, g.Hi, g.Wi, g.Ho, g.Wo, g.act, tx, ty, tpw);
    ACL_CHECK_LAUNCH("conv_thin_in2_kernel");
    return ACLGAN_OK;
}
// ACLGAN_THININ2=0: the round-2 kernel for the 7x7 layers, the general kernels for the first discriminator layers (A/B switch)
static bool thin_in2_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("ACLGAN_THININ2"); v = (e && !atoi(e)) ? 0 : 1; }
    return v == 1;
}
int conv_fwd_small(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, hipStream_t st) {
    // round 6: the first discriminator layers (4x4 stride 2 reflect pad 1, Cin 3 / 6 -> 64: networks.py:41)
    if (small_enabled() && thin_enabled() && thin_in2_enabled() && !g.up && g.k == 4 && g.s == 2 && g.p == 1 && g.Co == 64 && (g.Ci == 3 || g.Ci == 6) &&
        g.Hi >= 2 && g.Wi >= 2)
        return g.Ci == 3 ? launch_thin_in2<3, 4, 2, 1>(g, x, w, bias, y, st) : launch_thin_in2<6, 4, 2, 1>(g, x, w, bias, y, st);
    if (!small_enabled() || g.s != 1 || g.up || g.k != 7 || g.p != 3 || g.Hu < g.k || g.Wu < g.k) return ACLGAN_EUNSUPPORTED;
    if (thin_enabled() && g.Co <= 4 && g.Ci % 8 == 0) {
        const int tx = cdiv(g.Wi, TW), ty = cdiv(g.Hi, TH);
        hipLaunchKernelGGL(conv_thin_out_kernel<0>, dim3(g.B * tx * ty), dim3(256), 0, st, x, w, bias, y, g.Hi, g.Wi, g.Ci, g.Co, g.act, tx, ty);
        ACL_CHECK_LAUNCH("conv_thin_out_kernel<fwd>");
        return ACLGAN_OK;
    }
    if (thin_enabled() && thin_in2_enabled() && g.Co == 64 && (g.Ci == 3 || g.Ci == 4))
        return g.Ci == 3 ? launch_thin_in2<3, 7, 1, 3>(g, x, w, bias, y, st) : launch_thin_in2<4, 7, 1, 3>(g, x, w, bias, y, st);
    if (thin_enabled() && g.Co == 64 && (g.Ci == 3 || g.Ci == 4)) {
        const int tx = cdiv(g.Wi, TW), ty = cdiv(g.Hi, TH);
        const dim3 grid(g.B * tx * ((ty + 1) / 2));
        if (g.Ci == 3) hipLaunchKernelGGL((conv_thin_in_kernel<3>), grid, dim3(256), 0, st, x, w, bias, y, g.Hi, g.Wi, g.act, tx, ty);
        else hipLaunchKernelGGL((conv_thin_in_kernel<4>), grid, dim3(256), 0, st, x, w, bias, y, g.Hi, g.Wi, g.act, tx, ty);
        ACL_CHECK_LAUNCH("conv_thin_in_kernel<fwd>");
        return ACLGAN_OK;
    }
    if (g.Co != 4 || g.Ci % 16 != 0) return ACLGAN_EUNSUPPORTED;
    const int tx = cdiv(g.Wi, TW), ty = cdiv(g.Hi, TH);
    hipLaunchKernelGGL(conv_fwd_co4_kernel<7>, dim3(g.B * tx * ty), dim3(256), 0, st, x, w, bias, y, g.Hi, g.Wi, g.Ci, g.act, tx, ty);
    ACL_CHECK_LAUNCH("conv_fwd_co4_kernel");
    return ACLGAN_OK;
}

// dgrad of a thin-input 7x7 layer onto the padded grid dxp [B][H+6][W+6][Ci] (the caller folds the reflection)
int conv_dgrad_small(const ConvGeom& g, const float* dy, const float* w, float* dxp, hipStream_t st) {
    // round 6: ... and of the first discriminator layers (4x4 stride 2 pad 1) onto [B][H+2][W+2][Ci]
    if (small_enabled() && thin_enabled() && thin_in2_enabled() && !g.up && g.k == 4 && g.s == 2 && g.p == 1 && (g.Ci == 3 || g.Ci == 6) && g.Co % 4 == 0 &&
        g.Hi % 2 == 0 && g.Wi % 2 == 0) {
        const int64_t nthr = (int64_t)g.B * ((g.Hp + 1) / 2) * ((g.Wp + 1) / 2);
        const dim3 grid((unsigned)cdiv64(nthr, 256));
        if (g.Ci == 3) hipLaunchKernelGGL(conv_s2k4_thin_dgrad_kernel<3>, grid, dim3(256), 0, st, dy, w, dxp, g.B, g.Ho, g.Wo, g.Co, g.Hp, g.Wp);
        else hipLaunchKernelGGL(conv_s2k4_thin_dgrad_kernel<6>, grid, dim3(256), 0, st, dy, w, dxp, g.B, g.Ho, g.Wo, g.Co, g.Hp, g.Wp);
        ACL_CHECK_LAUNCH("conv_s2k4_thin_dgrad_kernel");
        return ACLGAN_OK;
    }
    if (!small_enabled() || !thin_enabled() || g.s != 1 || g.up || g.k != 7 || g.p != 3) return ACLGAN_EUNSUPPORTED;
    if (g.Ci > 4 || g.Co % 8 != 0) return ACLGAN_EUNSUPPORTED;
    const int tx = cdiv(g.Wi + 6, TW), ty = cdiv(g.Hi + 6, TH);
    hipLaunchKernelGGL(conv_thin_out_kernel<1>, dim3(g.B * tx * ty), dim3(256), 0, st, dy, w, (const float*)nullptr, dxp, g.Hi, g.Wi, g.Co, g.Ci,
                       (int)ACLGAN_ACT_NONE, tx, ty);
    ACL_CHECK_LAUNCH("conv_thin_out_kernel<dgrad>");
    return ACLGAN_OK;
}

static bool thin_wgrad_case(const ConvGeom& g, bool* wide_x) {
    if (!small_enabled() || !thin_enabled() || g.s != 1 || g.up || g.k != 7 || g.p != 3 || g.Hi < 7 || g.Wi < 7) return false;
    if (g.Co == 64 && g.Ci <= 4) { *wide_x = false; return true; }
    if (g.Co == 4 && g.Ci == 64) { *wide_x = true; return true; }
    return false;
}
static dim3 thin_wgrad_grid(const ConvGeom& g, bool wide_x) {
    return wide_x ? dim3(cdiv(g.Hi, THIN_ROWS_X), cdiv(g.Wi + 6, THIN_SEG), g.B) : dim3(cdiv(g.Hi, THIN_ROWS_DY), cdiv(g.Wi, THIN_SEG), g.B);
}
size_t conv_wgrad_small_scratch_bytes(const ConvGeom& g) {
    bool wx;
    if (!thin_wgrad_case(g, &wx)) return 0;
    const dim3 gr = thin_wgrad_grid(g, wx);
    const size_t nwg = (size_t)gr.x * gr.y * gr.z;
    if (deterministic())       // + second-level partials + the ordered bias column sums
        return (nwg + cdiv((int)nwg, 64)) * 49 * 256 * sizeof(float) + 256 + colsum_ordered_bytes(g.M, g.Co);
    return nwg * 49 * 256 * sizeof(float);
}

int conv_wgrad_small(const ConvGeom& g, const float* x, const float* dy, float* dw, float* db, hipStream_t st, void* scratch) {
    if (!small_enabled() || g.s != 1 || g.up || g.k != 7 || g.p != 3 || g.Hi < 7 || g.Wi < 7) return ACLGAN_EUNSUPPORTED;
    bool wx = false;
    const bool thin = thin_wgrad_case(g, &wx);
    const bool det = deterministic();
    if (det && !scratch) { set_error("conv_wgrad: deterministic mode needs the scratch buffer (aclgan_conv2d_wgrad_ws)"); return ACLGAN_EINVAL; }
    if (thin && (dw != nullptr || !wx)) {
        ACL_REQUIRE(dw != nullptr, "conv_wgrad(thin): dw must be given");
        const dim3 gr = thin_wgrad_grid(g, wx);
        float* part = (float*)scratch;       // nullptr: atomics straight into dw (correct, slower)
        const int nwg = gr.x * gr.y * gr.z, per_group = 64, ngroups = cdiv(nwg, per_group);
        float* gpart = det ? part + (size_t)nwg * 49 * 256 : nullptr;
        // deterministic mode: the fused bias sum of the kernel is a cross-workgroup atomic -> ordered column sums below instead
        if (wx) hipLaunchKernelGGL(conv_wgrad_thin_kernel<true>, gr, dim3(448), 0, st, x, dy, dw, (float*)nullptr, part, g.Hi, g.Wi, 4);
        else hipLaunchKernelGGL(conv_wgrad_thin_kernel<false>, gr, dim3(448), 0, st, x, dy, dw, det ? (float*)nullptr : db, part, g.Hi, g.Wi, g.Ci);
        ACL_CHECK_LAUNCH("conv_wgrad_thin_kernel");
        if (part) {
            if (wx) hipLaunchKernelGGL(conv_wgrad_thin_reduce_kernel<true>, dim3(49, ngroups), dim3(256), 0, st, part, dw, nwg, per_group, 4, gpart);
            else hipLaunchKernelGGL(conv_wgrad_thin_reduce_kernel<false>, dim3(49, ngroups), dim3(256), 0, st, part, dw, nwg, per_group, g.Ci, gpart);
            ACL_CHECK_LAUNCH("conv_wgrad_thin_reduce_kernel");
            if (gpart) {
                if (wx) hipLaunchKernelGGL(conv_wgrad_thin_reduce_kernel<true>, dim3(49, 1), dim3(256), 0, st, gpart, dw, ngroups, ngroups, 4, (float*)nullptr);
                else hipLaunchKernelGGL(conv_wgrad_thin_reduce_kernel<false>, dim3(49, 1), dim3(256), 0, st, gpart, dw, ngroups, ngroups, g.Ci, (float*)nullptr);
                ACL_CHECK_LAUNCH("conv_wgrad_thin_reduce_kernel(groups)");
            }
        }
        if (!wx && !det) return ACLGAN_OK;           // bias gradient fused (wide = dy)
        if (det) {
            if (!db) return ACLGAN_OK;
            char* cs = (char*)scratch + ((size_t)(nwg + ngroups) * 49 * 256 * sizeof(float) + 255) / 256 * 256;
            return colsum_ordered(dy, db, g.M, g.Co, cs, st);
        }
    } else {
        if (g.Co != 4 || g.Ci != 64 || det) return ACLGAN_EUNSUPPORTED;     // (deterministic mode: the general kernels with slice copies)
        if (dw) {
            const int Hp = g.Hi + 6, rows = 8;
            hipLaunchKernelGGL(conv_wgrad_co4_kernel<7>, dim3(cdiv(Hp, rows), 7, g.B), dim3(256), 0, st, x, dy, dw, g.Hi, g.Wi, rows);
            ACL_CHECK_LAUNCH("conv_wgrad_co4_kernel");
        }
    }
    if (db) {
        const int64_t npix = (int64_t)g.M;
        hipLaunchKernelGGL(colsum4_kernel, dim3((int)std::min<int64_t>(cdiv64(npix, 256 * 16), 1024)), dim3(256), 0, st, (const f32x4*)dy, db, npix);
        ACL_CHECK_LAUNCH("colsum4_kernel");
    }
    return ACLGAN_OK;
}

}  // namespace aclgan
