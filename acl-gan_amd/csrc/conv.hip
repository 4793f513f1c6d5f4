// conv.hip -- implicit-GEMM convolution kernels for gfx950 (MI355X), fp32 on the matrix cores.
//
// Replaces, for the ACL-GAN step, what the reference reaches through torch.nn / cuDNN:
//   forward : nn.ReflectionPad2d + nn.Conv2d (+ nn.Upsample before it) + bias + activation
//             (reference networks.py:319,366-370,256)
//   dgrad   : convolution_backward w.r.t. the input, incl. reflection_pad2d / upsample backward
//   wgrad   : convolution_backward w.r.t. weight (+ bias)          (autograd at trainer.py:169,292)
//
// Design (see DESIGN.md "conv kernels"):
//   * activations NHWC, weights OHWI -> the GEMM K axis (tap, cin) is contiguous in memory for
//     both operands of the forward GEMM; every HBM access is a 16-byte-per-lane coalesced load.
//   * reflect padding and the 2x nearest upsample are folded into the gather index of the
//     A-operand loader: the padded / 4x-upsampled tensors are never materialised.
//   * math: v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles/instr/SIMD = the 157 TF fp32 peak).
//     A wave owns a (TM*32)x(TN*32) output tile (TM*TN accumulators of 16 VGPRs); a workgroup of
//     WM*WN waves owns BM x BN.  Operands are staged through LDS in [k][m] order so that an MFMA
//     fragment read is one conflict-free ds_read_b32 per operand per k-pair; the fp32 MFMA is
//     slow enough (2048 cycles per 16-deep k-tile per wave) that LDS bandwidth is <15% utilised.
//   * global->register prefetch of k-tile t+1 is issued before the MFMAs of tile t, the
//     register->LDS write after them: one barrier per k-tile, two LDS buffers.
//   * workgroup -> tile mapping is XCD-aware: the 8 XCDs each get a contiguous range of
//     M-tiles (all N-tiles of an M-tile land on one XCD and share its L2 copy of the A rows).
#include "common.h"
#include <cstdlib>

namespace aclgan {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int reflect_idx(int v, int n) {
    v = v < 0 ? -v : v;
    return v >= n ? 2 * (n - 1) - v : v;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACLGAN_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACLGAN_ACT_LRELU) return v > 0.f ? v : 0.2f * v;
    if (act == ACLGAN_ACT_TANH) return tanhf(v);
    return v;
}

// block id -> logical tile id such that each XCD (block b runs on XCD b % 8) owns a contiguous
// range of logical ids.  Bijective for any nwg (cdna guide T1).  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

template <int V> struct Vec;
template <> struct Vec<4> { typedef float4 T; };
template <> struct Vec<1> { typedef float T; };

template <int V> __device__ __forceinline__ typename Vec<V>::T vzero();
template <> __device__ __forceinline__ float4 vzero<4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <> __device__ __forceinline__ float vzero<1>() { return 0.f; }

template <int V> __device__ __forceinline__ typename Vec<V>::T vload(const float* p) {
    return *reinterpret_cast<const typename Vec<V>::T*>(p);
}
__device__ __forceinline__ float vget(const float4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }
__device__ __forceinline__ float vget(const float& v, int) { return v; }

// Transposing store: a vector of V consecutive-k values of one m-row goes to V different k-rows.
template <int V>
__device__ __forceinline__ void lds_store_kvec(float* tile, int ld, int k, int m, const typename Vec<V>::T& v) {
#pragma unroll
    for (int j = 0; j < V; ++j) tile[(k + j) * ld + m] = vget(v, j);
}

// ------------------------------------------------------------------------------------------
// shared MFMA inner product over one 16-deep k-tile held in LDS as As[k][m], Bs[k][n]
// ------------------------------------------------------------------------------------------
template <int TM, int TN>
__device__ __forceinline__ void mma_ktile(const float* __restrict__ As, const float* __restrict__ Bs, int lda, int ldb,
                                          int am, int bn, int lane, f32x16 (&acc)[TM][TN]) {
    const int kh = lane >> 5, l31 = lane & 31;
    // all fragments of the 16-deep k-tile are fetched up front (8 k-pairs x (TM+TN) dwords) so that
    // the 8*TM*TN MFMAs issue back to back behind counted lgkmcnt waits instead of one LDS round
    // trip per k-pair.
    float a[8][TM], b[8][TN];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const int kr = 2 * ks + kh;
#pragma unroll
        for (int t = 0; t < TM; ++t) a[ks][t] = As[kr * lda + am + t * 32 + l31];
#pragma unroll
        for (int t = 0; t < TN; ++t) b[ks][t] = Bs[kr * ldb + bn + t * 32 + l31];
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the LDS reads ahead of the MFMA chain (the scheduler re-sinks them otherwise)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
struct FwdP {
    const float* x; const float* w; const float* bias; float* y;
    int Hi, Wi, Ci, Ho, Wo, Co, k, s, p, up, Hu, Wu, M, K, act, tiles_n, nwg;
};

template <int WM, int WN, int TM, int TN, int VEC>
__global__ void __launch_bounds__(WM * WN * 64) conv_fwd_kernel(FwdP p) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = 16, NT = WM * WN * 64;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int KV = BK / VEC;                 // vector columns per row
    constexpr int A_IT = (BM * KV + NT - 1) / NT;
    constexpr int B_IT = (BN * KV + NT - 1) / NT;
    typedef typename Vec<VEC>::T VT;
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LDA + LDB)];
    __shared__ int ri_y[BM], ri_x[BM], ri_b[BM];
    float* As = smem;
    float* Bs = smem + 2 * BK * LDA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile = xcd_remap(blockIdx.x, p.nwg);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;

    // per-row gather bases (output pixel -> top-left corner in padded/upsampled coordinates)
    for (int r = tid; r < BM; r += NT) {
        const int m = m0 + r;
        if (m < p.M) {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            ri_y[r] = oy * p.s - p.p; ri_x[r] = ox * p.s - p.p; ri_b[r] = b * p.Hi * p.Wi;
        } else {
            ri_y[r] = 0; ri_x[r] = 0; ri_b[r] = -1;
        }
    }
    __syncthreads();

    const int kcol = tid % KV;  // NT % KV == 0: fixed per thread
    VT ra[A_IT], rb[B_IT];

    auto load_tile = [&](int kt) {
        const int kk = kt * BK + kcol * VEC;
        const bool kok = kk < p.K;
        const int tap = kk / p.Ci, ci = kk - tap * p.Ci;
        const int ky = tap / p.k, kx = tap - ky * p.k;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int idx = tid + i * NT;
            const int row = idx / KV;
            VT v = vzero<VEC>();
            if (idx < BM * KV) {
                const int bo = ri_b[row];
                if (bo >= 0 && kok) {
                    const int iy = reflect_idx(ri_y[row] + ky, p.Hu) >> p.up;
                    const int ix = reflect_idx(ri_x[row] + kx, p.Wu) >> p.up;
                    v = vload<VEC>(p.x + (size_t)(bo + iy * p.Wi + ix) * p.Ci + ci);
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int idx = tid + i * NT;
            const int row = idx / KV;
            VT v = vzero<VEC>();
            if (idx < BN * KV) {
                const int n = n0 + row;
                if (n < p.Co && kok) v = vload<VEC>(p.w + (size_t)n * p.K + kk);
            }
            rb[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
        float* a = As + buf * BK * LDA;
        float* b = Bs + buf * BK * LDB;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int idx = tid + i * NT;
            if (idx < BM * KV) lds_store_kvec<VEC>(a, LDA, kcol * VEC, idx / KV, ra[i]);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int idx = tid + i * NT;
            if (idx < BN * KV) lds_store_kvec<VEC>(b, LDB, kcol * VEC, idx / KV, rb[i]);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        mma_ktile<TM, TN>(As + buf * BK * LDA, Bs + buf * BK * LDB, LDA, LDB, wm * TM * 32, wn * TN * 32, lane, acc);
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // epilogue: bias + activation, NHWC store (lanes 0..31 = 32 consecutive channels of one pixel)
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + l31;
        if (n >= p.Co) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < p.M) p.y[(size_t)m * p.Co + n] = apply_act(acc[i][j][r] + bv, p.act);
            }
        }
    }
}

template <int WM, int WN, int TM, int TN>
static int launch_fwd(const ConvGeom& g, FwdP p, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int tiles_m = cdiv(g.M, BM);
    p.tiles_n = cdiv(g.Co, BN);
    p.nwg = tiles_m * p.tiles_n;
    if (g.Ci % 4 == 0)
        hipLaunchKernelGGL((conv_fwd_kernel<WM, WN, TM, TN, 4>), dim3(p.nwg), dim3(WM * WN * 64), 0, st, p);
    else
        hipLaunchKernelGGL((conv_fwd_kernel<WM, WN, TM, TN, 1>), dim3(p.nwg), dim3(WM * WN * 64), 0, st, p);
    ACL_CHECK_LAUNCH("conv_fwd_kernel");
    return ACLGAN_OK;
}

size_t conv_fwd_scratch_bytes(const ConvGeom& g) { return conv_fwd_fast_scratch_bytes(g); }

int conv_fwd(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, hipStream_t st, void* scratch, float* stats, float* keepV) {
    if (keepV && (!scratch || !conv_fwd_keep_bytes(g))) { set_error("conv_fwd: this layer does not keep a Winograd input transform"); return ACLGAN_EINVAL; }
    if (stats) {      // only offered where conv_fwd_stats_chunk(g) > 0
        if (!scratch || !conv_fwd_stats_chunk(g)) { set_error("conv_fwd: this shape does not emit normalisation statistics"); return ACLGAN_EINVAL; }
        return conv_fwd_fast(g, x, w, bias, y, st, scratch, stats, keepV);
    }
    if (scratch) {
        const int rc = conv_up5_fwd(g, x, w, bias, y, scratch, st, keepV);
        if (rc != ACLGAN_EUNSUPPORTED) return rc;
    }
    FwdP p;
    p.x = x; p.w = w; p.bias = bias; p.y = y;
    p.Hi = g.Hi; p.Wi = g.Wi; p.Ci = g.Ci; p.Ho = g.Ho; p.Wo = g.Wo; p.Co = g.Co; p.k = g.k; p.s = g.s; p.p = g.p;
    p.up = g.up; p.Hu = g.Hu; p.Wu = g.Wu; p.M = g.M; p.K = g.K; p.act = g.act; p.tiles_n = 0; p.nwg = 0;
    {
        int rc = conv_fwd_small(g, x, w, bias, y, st);
        if (rc != ACLGAN_EUNSUPPORTED) return rc;
        rc = conv_fwd_fast(g, x, w, bias, y, st, scratch, nullptr, keepV);
        if (rc != ACLGAN_EUNSUPPORTED) return rc;
    }
    if (keepV) { set_error("conv_fwd: the Winograd path was not taken, nothing kept"); return ACLGAN_EINVAL; }
    if (g.Co > 64) return launch_fwd<2, 2, 2, 2>(g, p, st);   // 128 x 128
    if (g.Co > 32) return launch_fwd<4, 1, 2, 2>(g, p, st);   // 256 x 64
    return launch_fwd<4, 1, 2, 1>(g, p, st);                  // 256 x 32
}

// plain one-thread-per-output kernel: on-device cross-check of the MFMA path
__global__ void conv_fwd_naive_kernel(FwdP p) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)p.M * p.Co) return;
    const int n = (int)(idx % p.Co), m = (int)(idx / p.Co);
    const int hw = p.Ho * p.Wo, b = m / hw, rem = m - b * hw, oy = rem / p.Wo, ox = rem - oy * p.Wo;
    float acc = p.bias ? p.bias[n] : 0.f;
    for (int ky = 0; ky < p.k; ++ky)
        for (int kx = 0; kx < p.k; ++kx) {
            const int iy = reflect_idx(oy * p.s - p.p + ky, p.Hu) >> p.up;
            const int ix = reflect_idx(ox * p.s - p.p + kx, p.Wu) >> p.up;
            const float* xp = p.x + (size_t)((b * p.Hi + iy) * p.Wi + ix) * p.Ci;
            const float* wp = p.w + (size_t)n * p.K + (ky * p.k + kx) * p.Ci;
            for (int c = 0; c < p.Ci; ++c) acc = fmaf(xp[c], wp[c], acc);
        }
    p.y[idx] = apply_act(acc, p.act);
}

int conv_fwd_naive(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, hipStream_t st) {
    FwdP p;
    p.x = x; p.w = w; p.bias = bias; p.y = y;
    p.Hi = g.Hi; p.Wi = g.Wi; p.Ci = g.Ci; p.Ho = g.Ho; p.Wo = g.Wo; p.Co = g.Co; p.k = g.k; p.s = g.s; p.p = g.p;
    p.up = g.up; p.Hu = g.Hu; p.Wu = g.Wu; p.M = g.M; p.K = g.K; p.act = g.act; p.tiles_n = 0; p.nwg = 0;
    const int64_t n = (int64_t)g.M * g.Co;
    hipLaunchKernelGGL(conv_fwd_naive_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, st, p);
    ACL_CHECK_LAUNCH("conv_fwd_naive_kernel");
    return ACLGAN_OK;
}

// ------------------------------------------------------------------------------------------
// dgrad: gradient w.r.t. the padded (and upsampled) input grid, then a fold kernel that applies
// the reflection-pad backward (mirror the halo back in) and the upsample backward (2x2 sum).
// For stride s the padded grid splits into s*s parity classes, each with its own
// (k/s)x(k/s)-tap sub-filter, so no MACs are spent on structurally-zero taps.
// ------------------------------------------------------------------------------------------
struct DgP {
    const float* dy; const float* w; float* dxp;
    int Ho, Wo, Co, Ci, k, s, Hp, Wp, Hc, Wc, Mc, tiles_n, nwg;
};

template <int WM, int WN, int TM, int TN, int VA, int VB>
__global__ void __launch_bounds__(WM * WN * 64) conv_dgrad_kernel(DgP p) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = 16, NT = WM * WN * 64;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int KVA = BK / VA;
    constexpr int A_IT = (BM * KVA + NT - 1) / NT;
    constexpr int NVB = BN / VB;                      // vector columns of the B tile
    constexpr int B_IT = (BK * NVB + NT - 1) / NT;
    typedef typename Vec<VA>::T VTA;
    typedef typename Vec<VB>::T VTB;
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LDA + LDB)];
    __shared__ int ri_y[BM], ri_x[BM], ri_b[BM], ri_o[BM];
    float* As = smem;
    float* Bs = smem + 2 * BK * LDA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile = xcd_remap(blockIdx.x, p.nwg);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const int cy = blockIdx.z / p.s, cx = blockIdx.z % p.s;     // parity class
    const int Ty = (p.k - cy + p.s - 1) / p.s, Tx = (p.k - cx + p.s - 1) / p.s;
    const int Kc = Ty * Tx * p.Co;

    for (int r = tid; r < BM; r += NT) {
        const int m = m0 + r;
        int bo = -1, y2 = 0, x2 = 0, oo = -1;
        if (m < p.Mc) {
            const int hw = p.Hc * p.Wc;
            const int b = m / hw, rem = m - b * hw;
            y2 = rem / p.Wc; x2 = rem - y2 * p.Wc;
            const int py = y2 * p.s + cy, px = x2 * p.s + cx;
            if (py < p.Hp && px < p.Wp) { bo = b * p.Ho * p.Wo; oo = (b * p.Hp + py) * p.Wp + px; }
        }
        ri_y[r] = y2; ri_x[r] = x2; ri_b[r] = bo; ri_o[r] = oo;
    }
    __syncthreads();

    const int kcol = tid % KVA;
    VTA ra[A_IT];
    VTB rb[B_IT];

    auto load_tile = [&](int kt) {
        {
            const int kk = kt * BK + kcol * VA;
            const bool kok = kk < Kc;
            const int t = kk / p.Co, co = kk - t * p.Co;
            const int ty = t / Tx, tx = t - ty * Tx;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int idx = tid + i * NT;
                const int row = idx / KVA;
                VTA v = vzero<VA>();
                if (idx < BM * KVA) {
                    const int bo = ri_b[row];
                    const int oy = ri_y[row] - ty, ox = ri_x[row] - tx;
                    if (bo >= 0 && kok && (unsigned)oy < (unsigned)p.Ho && (unsigned)ox < (unsigned)p.Wo)
                        v = vload<VA>(p.dy + (size_t)(bo + oy * p.Wo + ox) * p.Co + co);
                }
                ra[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int idx = tid + i * NT;
            const int krow = idx / NVB, nc = idx - krow * NVB;
            VTB v = vzero<VB>();
            if (idx < BK * NVB) {
                const int kk = kt * BK + krow;
                const int n = n0 + nc * VB;
                if (kk < Kc && n < p.Ci) {
                    const int t = kk / p.Co, co = kk - t * p.Co;
                    const int ty = t / Tx, tx = t - ty * Tx;
                    const int ky = cy + p.s * ty, kx = cx + p.s * tx;
                    v = vload<VB>(p.w + ((size_t)(co * p.k + ky) * p.k + kx) * p.Ci + n);
                }
            }
            rb[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
        float* a = As + buf * BK * LDA;
        float* b = Bs + buf * BK * LDB;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int idx = tid + i * NT;
            if (idx < BM * KVA) lds_store_kvec<VA>(a, LDA, kcol * VA, idx / KVA, ra[i]);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int idx = tid + i * NT;
            if (idx < BK * NVB) {
                const int krow = idx / NVB, nc = idx - krow * NVB;
                *reinterpret_cast<VTB*>(b + krow * LDB + nc * VB) = rb[i];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (Kc + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        mma_ktile<TM, TN>(As + buf * BK * LDA, Bs + buf * BK * LDB, LDA, LDB, wm * TM * 32, wn * TN * 32, lane, acc);
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + l31;
        if (n >= p.Ci) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int oo = ri_o[row];
                if (oo >= 0) p.dxp[(size_t)oo * p.Ci + n] = acc[i][j][r];
            }
        }
    }
}

// reflection-pad backward + upsample backward: dx[b][i][j][c] (+)= sum of the padded-grid
// gradients of every padded/upsampled position that reads input pixel (i,j).
struct FoldP { const float* dxp; float* dx; int B, Hi, Wi, Ci, Hu, Wu, Hp, Wp, p, up, accumulate; int64_t total; };

__device__ __forceinline__ int fold_aliases(int u, int n, int p, int* q) {
    // padded positions q with reflect(q - p, n) == u
    int c = 0;
    q[c++] = u + p;
    if (u >= 1 && u <= p) q[c++] = p - u;
    if (u <= n - 2 && u >= n - 1 - p) q[c++] = p + 2 * (n - 1) - u;
    return c;
}

template <int V>
__global__ void conv_fold_kernel(FoldP f) {
    typedef typename Vec<V>::T VT;
    // grid (row blocks, Hi, B): the image row and the sample are block indices, one 32-bit division per thread (round 6; the flat 64-bit
    // index cost four 64-bit divisions per element)
    const int cv = f.Ci / V;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= f.Wi * cv) return;
    const int j = t / cv, c = (t - j * cv) * V;
    const int i = blockIdx.y, b = blockIdx.z;
    float acc[V];
#pragma unroll
    for (int t = 0; t < V; ++t) acc[t] = 0.f;
    const int nu = f.up ? 2 : 1;
    for (int du = 0; du < nu; ++du) {
        int qy[3];
        const int ny = fold_aliases((i << f.up) + du, f.Hu, f.p, qy);
        for (int dv = 0; dv < nu; ++dv) {
            int qx[3];
            const int nx = fold_aliases((j << f.up) + dv, f.Wu, f.p, qx);
            for (int a = 0; a < ny; ++a)
                for (int e = 0; e < nx; ++e) {
                    const VT v = vload<V>(f.dxp + ((size_t)(b * f.Hp + qy[a]) * f.Wp + qx[e]) * f.Ci + c);
#pragma unroll
                    for (int t = 0; t < V; ++t) acc[t] += vget(v, t);
                }
        }
    }
    float* o = f.dx + ((size_t)(b * f.Hi + i) * f.Wi + j) * f.Ci + c;
#pragma unroll
    for (int t = 0; t < V; ++t) o[t] = f.accumulate ? o[t] + acc[t] : acc[t];
}

// Round 6: the same gather restricted to a BAND of the padded grid (sub-pixel layers, conv_up5_dgrad): only the padded positions within
// `band` of the border hold values (one plain store each by the ring launch, nothing else of dxp is ever written or read), only the dx
// pixels that alias into the band are touched, and they are always accumulated (the phase launches wrote dx before).  Replaces the
// ring launch's fp32 atomics (6.2 M per launch on the 256 -> 128 layer: 445 us for 13 GFLOP; one-queue trace of round 5).
__global__ void conv_fold_band_kernel(FoldP f, int band) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= f.total) return;
    const int cv = f.Ci / 4;
    const int c = (int)(idx % cv) * 4;
    int64_t pix = idx / cv;
    const int j = (int)(pix % f.Wi); pix /= f.Wi;
    const int i = (int)(pix % f.Hi);
    const int b = (int)(pix / f.Hi);
    const int nu = f.up ? 2 : 1;
    // low-res pixels none of whose aliases lies in the band: main alias u + p >= band on both sides (the mirrored aliases belong to u <= p)
    const int bw = ((band - f.p) + nu - 1) / nu;
    if (i >= bw && i < f.Hi - bw && j >= bw && j < f.Wi - bw) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int du = 0; du < nu; ++du) {
        int qy[3];
        const int ny = fold_aliases((i << f.up) + du, f.Hu, f.p, qy);
        for (int dv = 0; dv < nu; ++dv) {
            int qx[3];
            const int nx = fold_aliases((j << f.up) + dv, f.Wu, f.p, qx);
            for (int a = 0; a < ny; ++a)
                for (int e = 0; e < nx; ++e) {
                    if (!(qy[a] < band || qy[a] >= f.Hp - band || qx[e] < band || qx[e] >= f.Wp - band)) continue;
                    const float4 v = *reinterpret_cast<const float4*>(f.dxp + ((size_t)(b * f.Hp + qy[a]) * f.Wp + qx[e]) * f.Ci + c);
                    acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
                }
        }
    }
    float4* o = reinterpret_cast<float4*>(f.dx + ((size_t)(b * f.Hi + i) * f.Wi + j) * f.Ci + c);
    float4 ov = *o;
    ov.x += acc[0]; ov.y += acc[1]; ov.z += acc[2]; ov.w += acc[3];
    *o = ov;
}
int conv_fold_band(const ConvGeom& g, const float* dxp, float* dx, int band, hipStream_t st) {
    if (g.Ci % 4 != 0) { set_error("conv_fold_band: Cin must be a multiple of 4"); return ACLGAN_EINVAL; }
    FoldP f;
    f.dxp = dxp; f.dx = dx; f.B = g.B; f.Hi = g.Hi; f.Wi = g.Wi; f.Ci = g.Ci; f.Hu = g.Hu; f.Wu = g.Wu;
    f.Hp = g.Hp; f.Wp = g.Wp; f.p = g.p; f.up = g.up; f.accumulate = 1;
    f.total = (int64_t)g.B * g.Hi * g.Wi * (g.Ci / 4);
    hipLaunchKernelGGL(conv_fold_band_kernel, dim3((unsigned)cdiv64(f.total, 256)), dim3(256), 0, st, f, band);
    ACL_CHECK_LAUNCH("conv_fold_band_kernel");
    return ACLGAN_OK;
}

size_t conv_dgrad_scratch_bytes(const ConvGeom& g) {
    // padded-grid gradient (general path) or the merged phase weights (sub-pixel path), whichever is larger
    return std::max(std::max((size_t)g.B * g.Hp * g.Wp * g.Ci * sizeof(float), conv_up5_dgrad_scratch_bytes(g)),
                    std::max(conv_wino_scratch_bytes(g), conv_s2k4_wino_scratch_bytes(g)));
}

template <int WM, int WN, int TM, int TN>
static int launch_dgrad(const ConvGeom& g, DgP p, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int tiles_m = cdiv(p.Mc, BM);
    p.tiles_n = cdiv(g.Ci, BN);
    p.nwg = tiles_m * p.tiles_n;
    dim3 grid(p.nwg, 1, g.s * g.s), block(WM * WN * 64);
    const bool va = g.Co % 4 == 0, vb = g.Ci % 4 == 0;
    if (va && vb) hipLaunchKernelGGL((conv_dgrad_kernel<WM, WN, TM, TN, 4, 4>), grid, block, 0, st, p);
    else if (va) hipLaunchKernelGGL((conv_dgrad_kernel<WM, WN, TM, TN, 4, 1>), grid, block, 0, st, p);
    else if (vb) hipLaunchKernelGGL((conv_dgrad_kernel<WM, WN, TM, TN, 1, 4>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((conv_dgrad_kernel<WM, WN, TM, TN, 1, 1>), grid, block, 0, st, p);
    ACL_CHECK_LAUNCH("conv_dgrad_kernel");
    return ACLGAN_OK;
}

// padded-grid gradient -> dx: reflection_pad2d backward (+ upsample_nearest2d backward); a gather, no atomics
int conv_fold(const ConvGeom& g, const float* dxp, float* dx, int accumulate, hipStream_t st) {
    FoldP f;
    f.dxp = dxp; f.dx = dx; f.B = g.B; f.Hi = g.Hi; f.Wi = g.Wi; f.Ci = g.Ci; f.Hu = g.Hu; f.Wu = g.Wu;
    f.Hp = g.Hp; f.Wp = g.Wp; f.p = g.p; f.up = g.up; f.accumulate = accumulate;
    if (g.Ci % 4 == 0) {
        f.total = (int64_t)g.B * g.Hi * g.Wi * (g.Ci / 4);
        hipLaunchKernelGGL(conv_fold_kernel<4>, dim3((unsigned)cdiv(f.Wi * (f.Ci / 4), 256), f.Hi, f.B), dim3(256), 0, st, f);
    } else {
        f.total = (int64_t)g.B * g.Hi * g.Wi * g.Ci;
        hipLaunchKernelGGL(conv_fold_kernel<1>, dim3((unsigned)cdiv(f.Wi * f.Ci, 256), f.Hi, f.B), dim3(256), 0, st, f);
    }
    ACL_CHECK_LAUNCH("conv_fold_kernel");
    return ACLGAN_OK;
}

int conv_dgrad(const ConvGeom& g, const float* dy, const float* w, float* dx, void* scratch, int accumulate, hipStream_t st) {
    DgP p;
    p.dy = dy; p.w = w; p.dxp = (float*)scratch;
    p.Ho = g.Ho; p.Wo = g.Wo; p.Co = g.Co; p.Ci = g.Ci; p.k = g.k; p.s = g.s; p.Hp = g.Hp; p.Wp = g.Wp;
    p.Hc = cdiv(g.Hp, g.s); p.Wc = cdiv(g.Wp, g.s); p.Mc = g.B * p.Hc * p.Wc; p.tiles_n = 0; p.nwg = 0;
    {
        const int rc0 = conv_up5_dgrad(g, dy, w, dx, accumulate, scratch, st);
        if (rc0 != ACLGAN_EUNSUPPORTED) return rc0;
    }
    bool direct = false;
    int rc = conv_dgrad_fast(g, dy, w, (float*)scratch, dx, accumulate, &direct, st);
    if (rc == ACLGAN_OK && direct) return ACLGAN_OK;   // dx complete: interior + mirrored halo written by the tuned kernel
    if (rc == ACLGAN_EUNSUPPORTED) rc = conv_dgrad_small(g, dy, w, (float*)scratch, st);   // thin input (3 channels): 4x4x1 MFMA kernel
    if (rc == ACLGAN_EUNSUPPORTED) {
        if (g.Ci > 64) rc = launch_dgrad<2, 2, 2, 2>(g, p, st);
        else if (g.Ci > 32) rc = launch_dgrad<4, 1, 2, 2>(g, p, st);
        else rc = launch_dgrad<4, 1, 2, 1>(g, p, st);
    }
    if (rc) return rc;
    return conv_fold(g, (const float*)scratch, dx, accumulate, st);
}

// ------------------------------------------------------------------------------------------
// wgrad: dW[co][tap][ci] += sum_pixels dY[pixel][co] * X[src(pixel, tap)][ci]
// GEMM M = Co, N = (tap, ci), K = B*Ho*Wo, split over K across blockIdx.z with fp32 atomics.
// ------------------------------------------------------------------------------------------
struct WgP {
    const float* x; const float* dy; float* dw;
    int Hi, Wi, Ci, Ho, Wo, Co, k, s, p, up, Hu, Wu, P, Kn, chunk, tiles_n, nwg;
    long long zs = 0;     // deterministic mode: pixel slice z accumulates into its own zeroed copy dw + z*zs (added in order afterwards)
};

template <int WM, int WN, int TM, int TN, int VA, int VB>
__global__ void __launch_bounds__(WM * WN * 64) conv_wgrad_kernel(WgP p) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BK = 16, NT = WM * WN * 64;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int MVA = BM / VA, NVB = BN / VB;
    constexpr int A_IT = (BK * MVA + NT - 1) / NT;
    constexpr int B_IT = (BK * NVB + NT - 1) / NT;
    typedef typename Vec<VA>::T VTA;
    typedef typename Vec<VB>::T VTB;
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LDA + LDB)];
    float* As = smem;
    float* Bs = smem + 2 * BK * LDA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile = xcd_remap(blockIdx.x, p.nwg);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const int pbeg = blockIdx.z * p.chunk;
    const int pend = min(p.P, pbeg + p.chunk);
    if (pbeg >= pend) return;

    VTA ra[A_IT];
    VTB rb[B_IT];
    const int hw = p.Ho * p.Wo;

    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int idx = tid + i * NT;
            const int krow = idx / MVA, mc = idx - krow * MVA;
            VTA v = vzero<VA>();
            if (idx < BK * MVA) {
                const int pix = pbeg + kt * BK + krow;
                const int m = m0 + mc * VA;
                if (pix < pend && m < p.Co) v = vload<VA>(p.dy + (size_t)pix * p.Co + m);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int idx = tid + i * NT;
            const int krow = idx / NVB, nc = idx - krow * NVB;
            VTB v = vzero<VB>();
            if (idx < BK * NVB) {
                const int pix = pbeg + kt * BK + krow;
                const int n = n0 + nc * VB;
                if (pix < pend && n < p.Kn) {
                    const int tap = n / p.Ci, ci = n - tap * p.Ci;
                    const int ky = tap / p.k, kx = tap - ky * p.k;
                    const int b = pix / hw, rem = pix - b * hw;
                    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                    const int iy = reflect_idx(oy * p.s - p.p + ky, p.Hu) >> p.up;
                    const int ix = reflect_idx(ox * p.s - p.p + kx, p.Wu) >> p.up;
                    v = vload<VB>(p.x + (size_t)((b * p.Hi + iy) * p.Wi + ix) * p.Ci + ci);
                }
            }
            rb[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
        float* a = As + buf * BK * LDA;
        float* b = Bs + buf * BK * LDB;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int idx = tid + i * NT;
            if (idx < BK * MVA) {
                const int krow = idx / MVA, mc = idx - krow * MVA;
                *reinterpret_cast<VTA*>(a + krow * LDA + mc * VA) = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int idx = tid + i * NT;
            if (idx < BK * NVB) {
                const int krow = idx / NVB, nc = idx - krow * NVB;
                *reinterpret_cast<VTB*>(b + krow * LDB + nc * VB) = rb[i];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (pend - pbeg + BK - 1) / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        mma_ktile<TM, TN>(As + buf * BK * LDA, Bs + buf * BK * LDB, LDA, LDB, wm * TM * 32, wn * TN * 32, lane, acc);
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + l31;
        if (n >= p.Kn) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < p.Co) atomicAdd(p.dw + (size_t)blockIdx.z * p.zs + (size_t)m * p.Kn + n, acc[i][j][r]);
            }
        }
    }
}

// bias gradient: db[c] += sum over pixels of dy[pixel][c]
__global__ void colsum_kernel(const float* __restrict__ dy, float* __restrict__ db, int P, int C, int rows_per_block) {
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(P, r0 + rows_per_block);
    if (c >= C) return;
    float s = 0.f;
    for (int r = r0 + threadIdx.y; r < r1; r += blockDim.y) s += dy[(size_t)r * C + c];
    __shared__ float red[16][65];
    red[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0) {
        float t = 0.f;
        for (int i = 0; i < (int)blockDim.y; ++i) t += red[i][threadIdx.x];
        atomicAdd(db + c, t);
    }
}

// same for C % 4 == 0 (C <= 1024): 16-byte loads, 256 / (C/4) rows in flight per workgroup
__global__ void __launch_bounds__(256) colsum_vec4_kernel(const float* __restrict__ dy, float* __restrict__ db, int P, int C, int rows_per_block) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int C4 = C >> 2, RG = 256 / C4;
    const int c4 = threadIdx.x % C4, rg = threadIdx.x / C4;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(P, r0 + rows_per_block);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (rg < RG)
        for (int r = r0 + rg; r < r1; r += RG) s += *reinterpret_cast<const f32x4*>(dy + (size_t)r * C + c4 * 4);
    __shared__ f32x4 red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    if (rg == 0) {
        for (int i = 1; i < RG; ++i) s += red[i * C4 + c4];
        atomicAdd(db + c4 * 4 + 0, s[0]); atomicAdd(db + c4 * 4 + 1, s[1]);
        atomicAdd(db + c4 * 4 + 2, s[2]); atomicAdd(db + c4 * 4 + 3, s[3]);
    }
}

// split K so that the grid holds ~3 workgroups per CU, each split >= 256 pixels
static void wgrad_generic_plan(int nwg, int P, int* splits, int* chunk) {
    int sp = cdiv(768, nwg);
    sp = std::max(1, std::min(sp, cdiv(P, 256)));
    *chunk = cdiv(cdiv(P, sp), 16) * 16;
    *splits = cdiv(P, *chunk);
}
template <int WM, int WN, int TM, int TN>
static int launch_wgrad(const ConvGeom& g, WgP p, hipStream_t st, void* det_part) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int tiles_m = cdiv(g.Co, BM);
    p.tiles_n = cdiv(p.Kn, BN);
    p.nwg = tiles_m * p.tiles_n;
    int splits;
    wgrad_generic_plan(p.nwg, p.P, &splits, &p.chunk);
    float* dw_out = p.dw;
    const int64_t ndw = (int64_t)g.Co * p.Kn;
    if (det_part != nullptr && splits > 1) {      // deterministic mode: one zeroed copy of dw per pixel slice, added in order below
        hipError_t e = hipMemsetAsync(det_part, 0, (size_t)splits * ndw * sizeof(float), st);
        if (e != hipSuccess) return hip_fail(e, "memset wgrad slices");
        p.dw = (float*)det_part; p.zs = ndw;
    }
    dim3 grid(p.nwg, 1, splits), block(WM * WN * 64);
    const bool va = g.Co % 4 == 0, vb = g.Ci % 4 == 0;
    if (va && vb) hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN, TM, TN, 4, 4>), grid, block, 0, st, p);
    else if (va) hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN, TM, TN, 4, 1>), grid, block, 0, st, p);
    else if (vb) hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN, TM, TN, 1, 4>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN, TM, TN, 1, 1>), grid, block, 0, st, p);
    ACL_CHECK_LAUNCH("conv_wgrad_kernel");
    if (p.zs) return reduce_slices_ordered(p.dw, ndw, splits, dw_out, st);
    return ACLGAN_OK;
}
// deterministic mode: slice copies of the general kernel (smallest tiles = most slices) + the ordered bias column sums
static size_t wgrad_generic_det_bytes(const ConvGeom& g) {
    int splits, chunk;
    wgrad_generic_plan(1, g.M, &splits, &chunk);
    return (((size_t)(splits + 1) * g.Co * g.K * sizeof(float) + 255) & ~(size_t)255) + colsum_ordered_bytes(g.M, g.Co);
}

size_t conv_wgrad_scratch_bytes(const ConvGeom& g) {
    const size_t b = std::max(conv_wgrad_fast_scratch_bytes(g), conv_wgrad_small_scratch_bytes(g));
    return (deterministic() && !conv_wgrad_fast_supported(g)) ? std::max(b, wgrad_generic_det_bytes(g)) : b;    // general kernel: slice copies
}

int conv_wgrad(const ConvGeom& g, const float* x, const float* dy, float* dw, float* db, hipStream_t st, void* scratch, const float* haveV) {
    if (scratch) {
        const int rc0 = conv_up5_wgrad(g, x, dy, dw, db, scratch, st, haveV);
        if (rc0 != ACLGAN_EUNSUPPORTED) return rc0;
    }
    WgP p;
    p.x = x; p.dy = dy; p.dw = dw;
    p.Hi = g.Hi; p.Wi = g.Wi; p.Ci = g.Ci; p.Ho = g.Ho; p.Wo = g.Wo; p.Co = g.Co; p.k = g.k; p.s = g.s; p.p = g.p;
    p.up = g.up; p.Hu = g.Hu; p.Wu = g.Wu; p.P = g.M; p.Kn = g.K; p.chunk = 0; p.tiles_n = 0; p.nwg = 0;
    int rc = conv_wgrad_small(g, x, dy, dw, db, st, scratch);
    if (rc != ACLGAN_EUNSUPPORTED) return rc;
    rc = ACLGAN_OK;
    if (dw) {
        rc = conv_wgrad_fast(g, x, dy, dw, db, st, scratch, haveV);
        if (rc == ACLGAN_OK) db = nullptr;   // bias gradient fused into the tuned kernel
        if (rc == ACLGAN_EUNSUPPORTED) {
            void* det = nullptr;
            if (deterministic()) {
                if (!scratch) { set_error("conv_wgrad: deterministic mode needs the scratch buffer (aclgan_conv2d_wgrad_ws)"); return ACLGAN_EINVAL; }
                det = scratch;
            }
            if (g.Co > 64) rc = launch_wgrad<2, 2, 2, 2>(g, p, st, det);       // 128 x 128
            else if (g.Co > 32) rc = launch_wgrad<2, 2, 1, 2>(g, p, st, det);  // 64 x 128
            else rc = launch_wgrad<1, 4, 1, 2>(g, p, st, det);                 // 32 x 256
        }
        if (rc) return rc;
    }
    if (db && deterministic()) {      // ordered column sums (the kernels below add their row blocks with fp32 atomics)
        if (!scratch) { set_error("conv_wgrad: deterministic mode needs the scratch buffer (aclgan_conv2d_wgrad_ws)"); return ACLGAN_EINVAL; }
        int splits, chunk;
        wgrad_generic_plan(1, g.M, &splits, &chunk);
        void* cs = (char*)scratch + (((size_t)(splits + 1) * g.Co * g.K * sizeof(float) + 255) & ~(size_t)255);
        return colsum_ordered(dy, db, g.M, g.Co, cs, st);
    }
    if (db && g.Co % 4 == 0 && g.Co <= 1024 && (256 % (g.Co / 4)) == 0) {
        const int rows = 512;
        hipLaunchKernelGGL(colsum_vec4_kernel, dim3(cdiv(g.M, rows)), dim3(256), 0, st, dy, db, g.M, g.Co, rows);
        ACL_CHECK_LAUNCH("colsum_vec4_kernel");
    } else if (db) {
        const int rows = 1024;
        dim3 block(64, 4), grid(cdiv(g.M, rows), cdiv(g.Co, 64));
        hipLaunchKernelGGL(colsum_kernel, grid, block, 0, st, dy, db, g.M, g.Co, rows);
        ACL_CHECK_LAUNCH("colsum_kernel");
    }
    return rc;
}

int make_geom(const aclgan_conv_desc* d, ConvGeom* g) {
    ACL_REQUIRE(d && g, "null conv desc");
    ACL_REQUIRE(d->B > 0 && d->Hi > 0 && d->Wi > 0 && d->Ci > 0 && d->Co > 0, "conv: non-positive dims");
    ACL_REQUIRE(d->k >= 1 && d->stride >= 1 && d->pad >= 0, "conv: bad k/stride/pad");
    g->B = d->B; g->Hi = d->Hi; g->Wi = d->Wi; g->Ci = d->Ci; g->Co = d->Co; g->k = d->k; g->s = d->stride;
    g->p = d->pad; g->up = d->upsample ? 1 : 0; g->act = d->act;
    g->Hu = d->Hi << g->up; g->Wu = d->Wi << g->up;
    // reflection needs pad < size (torch raises otherwise: networks.py:319 ReflectionPad2d)
    ACL_REQUIRE(g->p < g->Hu && g->p < g->Wu, "conv: reflect pad %d >= input size %dx%d", g->p, g->Hu, g->Wu);
    g->Hp = g->Hu + 2 * g->p; g->Wp = g->Wu + 2 * g->p;
    ACL_REQUIRE(g->Hp >= g->k && g->Wp >= g->k, "conv: kernel larger than padded input");
    g->Ho = (g->Hp - g->k) / g->s + 1; g->Wo = (g->Wp - g->k) / g->s + 1;
    g->M = g->B * g->Ho * g->Wo; g->K = g->k * g->k * g->Ci;
    return ACLGAN_OK;
}

}  // namespace aclgan
