// conv_wino.hip -- Winograd F(4x4, 3x3) for the 3x3 stride-1 reflect-pad-1 convolutions (the eight ResBlock convs of every
// content encoder and decoder pass, networks.py:297-310: 82 % of the step's MACs).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 4x4 output tile, per (cin, cout) pair; (.) = elementwise
//
// 36 multiplies per 16 outputs instead of 144: 4x fewer MACs on the fp32 matrix cores, at the price of two memory-bound
// transform passes.  The elementwise product summed over cin is, per frequency f = 0..35, a plain GEMM
//   M_f[T][Cout] = V_f[T][Cin] x U_f[Cout][Cin]^T          T = B * (H/4) * (W/4) tiles
// and runs on the tuned forward kernel itself (gemm_slices_f32: conv_fwd_fast_kernel as a 1x1 "conv", slice f on blockIdx.y).
//
// Precision: fp32 everywhere; the transforms (coefficients up to 8) cost ~1.5 decimal digits -- measured 1.3e-5 max-abs
// relative error per convolution against 2.7e-7 for the direct kernel (north-star tolerance of the forward: 1e-3).  The
// reference itself runs these layers through cuDNN with cudnn.benchmark = True (train.py:29), whose fp32 algorithm choice
// for 3x3 stride-1 convolutions is the same family.  ACLGAN_NOWINO=1 selects the direct kernels everywhere.
//
// The reflection padding of the forward lives in the input-transform gather; the backward w.r.t. the input uses the same
// pipeline on dy with ZERO padding and the flipped / transposed filter for the interior of dx (the halo ring of the padded grid
// keeps its small direct launch, conv_fast.hip mode 2).  The weight gradient stays on the direct kernel.
#include "common.h"
#include "st16.h"
#include <atomic>
#include <cstdlib>
#include <algorithm>

namespace aclgan {

// conv_fast.hip.  a_mod / b_mod > 0: the A (resp. B) operand of slice f is plane f % mod (the four sub-pixel phases share one input transform)
int gemm_slices_f32(const float* A, const float* Bm, float* Cm, int T, int K, int N, int nslices, int a_mod, hipStream_t st);
size_t gemm_at_b_slices_scratch(int T, int M, int N, int nslices);
int gemm_at_b_slices_f32(const float* A, const float* Bm, float* Cm, int T, int M, int N, int nslices, int b_mod, void* part, hipStream_t st);

static thread_local const WinoUCache* g_ucache = nullptr;
void set_wino_ucache(const WinoUCache* c) { g_ucache = c; }
const WinoUCache* wino_ucache() { return g_ucache; }
bool wino_u_cached(const float* wkey, int variant) {
    if (!g_ucache || !wkey) return false;
    bool fresh = true;
    float* u = g_ucache->lookup(g_ucache->user, wkey, variant, 0, &fresh);
    return u != nullptr && !fresh;
}

namespace {

// U of filter tensor `key` from the cache when the scheduler offers one (fresh: compute it now), else the scratch slice
float* cached_u(float* scratch_u, const float* key, int variant, size_t bytes, bool* fresh) {
    *fresh = true;
    if (!g_ucache || !key) return scratch_u;
    float* u = g_ucache->lookup(g_ucache->user, key, variant, bytes, fresh);
    if (!u) { *fresh = true; return scratch_u; }
    return u;
}
// Layouts a cached transform can have (bits 4.. of the cache lookup's `variant`): the engine keys an entry by (filter, variant & 15) and
// remembers the layout it was filled in; a lookup that asks for another layout (the tuning switches changed in the middle of an update)
// gets no cache entry and computes into its own scratch instead of reading a transform in the wrong order.
enum { U_PIPE = 0, U_FRAG = 1, U_X3 = 2 };
inline int uvar(int variant, int layout) { return variant | (layout << 4); }

__device__ __forceinline__ int reflw(int v, int n) {
    v = v < 0 ? -v : v;
    return v >= n ? 2 * (n - 1) - v : v;
}
__device__ __forceinline__ float actw(float v, int act) {
    if (act == ACLGAN_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACLGAN_ACT_LRELU) return v > 0.f ? v : 0.2f * v;
    if (act == ACLGAN_ACT_TANH) return tanhf(v);
    return v;
}

// the transform kernels move NV channels per thread (NHWC: NV consecutive floats = one 4 / 8 / 16-byte access)
template <int NV> struct VecOf { typedef float T __attribute__((ext_vector_type(NV))); };
template <> struct VecOf<1> { typedef float T; };
template <int NV> __device__ __forceinline__ typename VecOf<NV>::T actv(typename VecOf<NV>::T v, int act) {
    if (act == ACLGAN_ACT_NONE) return v;
#pragma unroll
    for (int e = 0; e < NV; ++e) v[e] = actw(v[e], act);
    return v;
}
template <> __device__ __forceinline__ float actv<1>(float v, int act) { return actw(v, act); }

// B^T d (input transform along one axis), G g (filter), A^T m (output)
template <typename F>
__device__ __forceinline__ void bt6(const F (&d)[6], F (&t)[6]) {
    t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    t[1] = -4.f * d[1] - 4.f * d[2] + d[3] + d[4];
    t[2] = 4.f * d[1] - 4.f * d[2] - d[3] + d[4];
    t[3] = -2.f * d[1] - d[2] + 2.f * d[3] + d[4];
    t[4] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
    t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
template <typename F>
__device__ __forceinline__ void g6(const F (&g)[3], F (&u)[6]) {
    u[0] = 0.25f * g[0];
    u[1] = -(g[0] + g[1] + g[2]) * (1.f / 6.f);
    u[2] = -(g[0] - g[1] + g[2]) * (1.f / 6.f);
    u[3] = g[0] * (1.f / 24.f) + g[1] * (1.f / 12.f) + g[2] * (1.f / 6.f);
    u[4] = g[0] * (1.f / 24.f) - g[1] * (1.f / 12.f) + g[2] * (1.f / 6.f);
    u[5] = g[2];
}
template <typename F>
__device__ __forceinline__ void at4(const F (&m)[6], F (&y)[4]) {
    y[0] = m[0] + m[1] + m[2] + m[3] + m[4];
    y[1] = m[1] - m[2] + 2.f * (m[3] - m[4]);
    y[2] = m[1] + m[2] + 4.f * (m[3] + m[4]);
    y[3] = m[1] - m[2] + 8.f * (m[3] - m[4]) + m[5];
}

// U[f][row][k]: forward: row = cout, k = cin from w[cout][ky][kx][cin];
// dgrad (flip = 1): row = cin, k = cout from the flipped filter w[cout][2-ky][2-kx][cin]
// planes > 0 (elements per plane): U is written as the three bf16 planes of the split-bf16 GEMM (gemm_bf16x3.hip) instead of fp32
// blockIdx.y = filter of a batch: filter y is read at w + y * wstr and written at U + y * ustr floats (planes: U16 + y * ustr16 halfs) -- the
// merged phase filters of a sub-pixel layer (wstr = Co * 9 * Ci, back to back) or equally shaped filters of one network (round 5: all ResBlock
// filters of an encoder / decoder in one launch, wstr = their distance in the flat parameter buffer)
__global__ void __launch_bounds__(256) wino_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int Co, int Ci, int flip, int64_t planes,
                                                          int64_t wstr, int64_t ustr, int64_t ustr16) {
    // one thread = one (row, k) pair of the OUTPUT layout, k fastest: forward (row = cout, k = cin) reads w coalesced along cin;
    // dgrad (row = cin, k = cout, flipped taps) reads w with stride 9*Cin -- 9 strided loads per thread against 36 coalesced stores
    const int R = flip ? Ci : Co, K = flip ? Co : Ci;
    const int64_t n = (int64_t)R * K;
    w += (size_t)blockIdx.y * wstr;
    unsigned short* U16 = reinterpret_cast<unsigned short*>(U) + (size_t)blockIdx.y * ustr16;
    U += (size_t)blockIdx.y * ustr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int kk = (int)(i % K), row = (int)(i / K);
        const int co = flip ? kk : row, ci = flip ? row : kk;
        float t[6][3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {            // G g: columns
            float c[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int sy = flip ? 2 - ky : ky, sx = flip ? 2 - kx : kx;
                c[ky] = w[((size_t)(co * 3 + sy) * 3 + sx) * Ci + ci];
            }
            float o[6];
            g6(c, o);
#pragma unroll
            for (int a = 0; a < 6; ++a) t[a][kx] = o[a];
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {               // (G g) G^T: rows
            float o[6];
            g6(t[a], o);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const size_t at = ((size_t)(a * 6 + j) * R + row) * K + kk;
                if (planes) {
                    unsigned short h, m, l;
                    split3(o[j], h, m, l);
                    U16[at] = h; U16[at + planes] = m; U16[at + 2 * planes] = l;
                } else {
                    U[at] = o[j];
                }
            }
        }
    }
}

// A strided window onto an NHWC tensor [B][HS][WS][C]: logical pixel (iy, ix), 0 <= iy < H, 0 <= ix < W, lives at physical
// row oy0 + sy*iy, column ox0 + sx*ix.  Identity view = the tensor itself; the sub-pixel phases of the upsample+5x5 layers are
// the views (sy = sx = 2, oy0 = 2 + py, ox0 = 2 + px) of the hi-res map (conv_fast.hip: up5_*).
struct WView { int H, W, sy, sx, oy0, ox0, HS, WS; };
struct WViews { WView v[4]; int nph; };
__host__ __device__ __forceinline__ size_t vaddr(const WView& v, int b, int iy, int ix, int C) {
    return ((size_t)(b * v.HS + v.oy0 + v.sy * iy) * v.WS + v.ox0 + v.sx * ix) * C;
}
WView ident_view(int H, int W) { WView v = {H, W, 1, 1, 0, 0, H, W}; return v; }
WView phase_view(int Hv, int Wv, int py, int px, int Hf, int Wf) { WView v = {Hv, Wv, 2, 2, 2 + py, 2 + px, Hf, Wf}; return v; }

// V[ph][f][t][c] = (B^T d B)[f] of the 6x6 patch of tile t = (b, ty, tx) of view ph: logical rows 4ty+off .. 4ty+off+5; positions
// outside the view are reflected (reflect = 1: the forward's ReflectionPad2d) or read as zero.  blockIdx.y = phase.
// planes > 0 (elements per plane = nph * 36 * T * C): V is written as three bf16 planes (h, m, l with v = h + m + l, st16.h split3) for the
// split-bf16 GEMM instead of fp32: 6 bytes per value instead of 4
template <int NV>
__global__ void __launch_bounds__(256) wino_input_kernel(const float* __restrict__ x, float* __restrict__ V, int B, WViews vs, int C, int TY, int TX, int off,
                                                         int reflect, int64_t planes) {
    typedef typename VecOf<NV>::T F;
    const WView v = vs.v[blockIdx.y];
    const int Cv = C / NV;
    const int64_t T = (int64_t)B * TY * TX, n = T * Cv;
    float* Vp = V + (size_t)blockIdx.y * 36 * T * C;
    unsigned short* Vq = reinterpret_cast<unsigned short*>(V) + (size_t)blockIdx.y * 36 * T * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cv) * NV;
        const int64_t t = i / Cv;
        const int tx = (int)(t % TX), ty = (int)((t / TX) % TY), b = (int)(t / ((int64_t)TX * TY));
        F tmp[6][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {              // columns of the patch: B^T d
            F d[6];
            const int ixr = 4 * tx + off + j;
            const bool xin = (unsigned)ixr < (unsigned)v.W;
            const int ix = reflect ? reflw(ixr, v.W) : (xin ? ixr : 0);
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const int iyr = 4 * ty + off + r;
                const bool yin = (unsigned)iyr < (unsigned)v.H;
                const int iy = reflect ? reflw(iyr, v.H) : (yin ? iyr : 0);
                const F val = *reinterpret_cast<const F*>(x + vaddr(v, b, iy, ix, C) + c);
                d[r] = (reflect || (xin && yin)) ? val : (F)(0.f);
            }
            F o[6];
            bt6(d, o);
#pragma unroll
            for (int r = 0; r < 6; ++r) tmp[r][j] = o[r];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {              // rows: (B^T d) B
            F o[6];
            bt6(tmp[r], o);
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const size_t at = ((size_t)(r * 6 + j) * T + t) * C + c;
                if (planes) {
                    if constexpr (NV == 1) {
                        unsigned short h, m, l;
                        split3(o[j], h, m, l);
                        Vq[at] = h; Vq[at + planes] = m; Vq[at + 2 * planes] = l;
                    } else {
                        typedef unsigned int U __attribute__((ext_vector_type(NV / 2)));
                        U ph, pm, pl;
#pragma unroll
                        for (int e = 0; e < NV; e += 2) {
                            unsigned short h0, m0, l0, h1, m1, l1;
                            split3(o[j][e], h0, m0, l0); split3(o[j][e + 1], h1, m1, l1);
                            const unsigned int wh = h0 | ((unsigned int)h1 << 16), wm = m0 | ((unsigned int)m1 << 16), wl = l0 | ((unsigned int)l1 << 16);
                            if constexpr (NV == 2) { ph = wh; pm = wm; pl = wl; }
                            else { ph[e / 2] = wh; pm[e / 2] = wm; pl[e / 2] = wl; }
                        }
                        *reinterpret_cast<U*>(Vq + at) = ph; *reinterpret_cast<U*>(Vq + at + planes) = pm; *reinterpret_cast<U*>(Vq + at + 2 * planes) = pl;
                    }
                } else {
                    *reinterpret_cast<F*>(Vp + at) = o[j];
                }
            }
        }
    }
}

// sum = 0: y through view ph  (+)= act((A^T M_ph A) + bias), one view per phase (forward: the phases interleave into the hi-res map);
// sum = 1: y through view 0   (+)= sum over the nph phases of A^T M_ph A  (dgrad: all phases land on the same low-res dx).
// Outputs beyond the view's logical extent (ragged last tiles) are dropped.
// (SUM as a template parameter: the forward variant then carries neither the 4 x 4 accumulator tile nor the phase loop state -- fewer
//  registers, one more wave per SIMD for a kernel that lives on memory-level parallelism)
template <int NV, int SUM>
__global__ void __launch_bounds__(256) wino_output_kernel(const float* __restrict__ M, const float* __restrict__ bias, float* __restrict__ y, int B, WViews vs,
                                                          int C, int TY, int TX, int act, int accumulate, float2* __restrict__ stats) {
    constexpr int sum = SUM;
    typedef typename VecOf<NV>::T F;
    const int Cv = C / NV;
    const int64_t T = (int64_t)B * TY * TX, n = T * Cv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cv) * NV;
        const int64_t t = i / Cv;
        const int tx = (int)(t % TX), ty = (int)((t / TX) % TY), b = (int)(t / ((int64_t)TX * TY));
        const F bv = bias ? *reinterpret_cast<const F*>(bias + c) : (F)(0.f);
        F tot[SUM ? 4 : 1][4];
#pragma unroll
        for (int r = 0; r < (SUM ? 4 : 1); ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) tot[r][j] = (F)(0.f);
        F st_sh = (F)(0.f), st_s = (F)(0.f), st_q = (F)(0.f);      // normalisation statistics of this tile (stats != nullptr), shifted sums
        const unsigned int toff = (unsigned int)t * (unsigned int)C + (unsigned int)c;      // (T * C < 2^30: checked by the launcher) -- the 36
        // plane addresses are then SGPR base + this one VGPR instead of 36 64-bit VGPR pairs (203 -> fewer registers, a third wave per SIMD)
        for (int ph = 0; ph < vs.nph; ++ph) {
            const float* Mp = M + (size_t)ph * 36 * T * C;
            F tmp[4][6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {              // columns: A^T M
                F m[6];
#pragma unroll
                for (int r = 0; r < 6; ++r) m[r] = *reinterpret_cast<const F*>(Mp + (size_t)(r * 6 + j) * T * C + toff);      // uniform plane base + one 32-bit lane offset
                F o[4];
                at4(m, o);
#pragma unroll
                for (int r = 0; r < 4; ++r) tmp[r][j] = o[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {              // rows: (A^T M) A
                F o[4];
                at4(tmp[r], o);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (sum) tot[SUM ? r : 0][j] += o[j];
                    else {
                        const WView v = vs.v[ph];
                        const int oy = 4 * ty + r, ox = 4 * tx + j;
                        if (oy < v.H && ox < v.W) {
                            F* dst = reinterpret_cast<F*>(y + vaddr(v, b, oy, ox, C) + c);
                            const F val = actv<NV>(o[j] + bv, act);
                            *dst = accumulate ? *dst + val : val;
                            if (stats) {
                                if (r == 0 && j == 0) st_sh = val;
                                const F dv = val - st_sh;
                                st_s += dv; st_q += dv * dv;
                            }
                        }
                    }
                }
            }
        }
        if (stats) {      // part[b][tile][c] = (mean, M2) of the tile's 16 outputs: the chunk partials norm_finalize_* combines (elementwise.hip)
            const F mean = st_sh + st_s * (1.f / 16.f), m2 = st_q - st_s * st_s * (1.f / 16.f);
            float2* o = stats + (size_t)t * C + c;
            if constexpr (NV == 1) o[0] = make_float2(mean, m2);
            else {
#pragma unroll
                for (int e = 0; e < NV; ++e) o[e] = make_float2(mean[e], m2[e]);
            }
        }
        if (sum) {
            const WView v = vs.v[0];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int oy = 4 * ty + r, ox = 4 * tx + j;
                    if (oy < v.H && ox < v.W) {
                        F* dst = reinterpret_cast<F*>(y + vaddr(v, b, oy, ox, C) + c);
                        const F val = actv<NV>(tot[SUM ? r : 0][j] + bv, act);
                        *dst = accumulate ? *dst + val : val;
                    }
                }
        }
    }
}

// ---- weight gradient in the Winograd domain:  dU_f[co][ci] = sum_t dM_f[t][co] V_f[t][ci],  dM = A dY A^T,  dg = G^T dU G ----
template <typename F>
__device__ __forceinline__ void a6(const F (&y)[4], F (&m)[6]) {     // A y  (A = transpose of A^T, 6 x 4)
    m[0] = y[0];
    m[1] = y[0] + y[1] + y[2] + y[3];
    m[2] = y[0] - y[1] + y[2] - y[3];
    m[3] = y[0] + 2.f * y[1] + 4.f * y[2] + 8.f * y[3];
    m[4] = y[0] - 2.f * y[1] + 4.f * y[2] - 8.f * y[3];
    m[5] = y[3];
}
__device__ __forceinline__ void gt3(const float (&u)[6], float (&g)[3]) {    // G^T u
    g[0] = 0.25f * u[0] - (u[1] + u[2]) * (1.f / 6.f) + (u[3] + u[4]) * (1.f / 24.f);
    g[1] = (u[2] - u[1]) * (1.f / 6.f) + (u[3] - u[4]) * (1.f / 12.f);
    g[2] = -(u[1] + u[2]) * (1.f / 6.f) + (u[3] + u[4]) * (1.f / 6.f) + u[5];
}

// dM[ph][f][t][c] = (A dY A^T)[f] of the 4x4 tile t of the output gradient seen through view ph (zero outside the view);
// bpart[ph][j][c] (optional) = this thread's column sum of dy (bias gradient, reduced in order by wino_bias_finish_kernel).
// Launch with gridDim.x * 256 a multiple of C / NV: a thread keeps ONE group of NV channels.  blockIdx.y = phase.
template <int NV>
__global__ void __launch_bounds__(256) wino_outgrad_kernel(const float* __restrict__ dy, float* __restrict__ dM, float* __restrict__ bpart, int B, WViews vs,
                                                           int C, int TY, int TX) {
    typedef typename VecOf<NV>::T F;
    const WView v = vs.v[blockIdx.y];
    const int Cv = C / NV;
    const int64_t T = (int64_t)B * TY * TX;
    float* dMp = dM + (size_t)blockIdx.y * 36 * T * C;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
    const int c = (int)(gid % Cv) * NV;
    F bs = (F)(0.f);
    for (int64_t t = gid / Cv; t < T; t += nth / Cv) {
        const int tx = (int)(t % TX), ty = (int)((t / TX) % TY), b = (int)(t / ((int64_t)TX * TY));
        F tmp[6][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {              // columns: A dY
            F q[4];
            const int ox = 4 * tx + j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int oy = 4 * ty + r;
                const bool in = oy < v.H && ox < v.W;
                q[r] = in ? *reinterpret_cast<const F*>(dy + vaddr(v, b, in ? oy : 0, in ? ox : 0, C) + c) : (F)(0.f);
                bs += q[r];
            }
            F o[6];
            a6(q, o);
#pragma unroll
            for (int r = 0; r < 6; ++r) tmp[r][j] = o[r];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {              // rows: (A dY) A^T
            F o[6];
            a6(tmp[r], o);
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<F*>(dMp + ((size_t)(r * 6 + j) * T + t) * C + c) = o[j];
        }
    }
    if (bpart) *reinterpret_cast<F*>(bpart + ((size_t)blockIdx.y * (nth / Cv) + gid / Cv) * C + c) = bs;     // [ph][row = gid / Cv][c]
}
// Ordered column sums of the partial rows: 16 channels x 16 row groups per workgroup, groups combined through LDS.  blockIdx.y =
// row slice: with `level2 != nullptr` the slice sums are STORED at level2[slice][c] (first level of a two-level reduction -- the
// sub-pixel layers produce 32 768 partial rows for 64 channels, far too long a serial walk for four workgroups); else db[c] += sum.
__global__ void __launch_bounds__(256) wino_bias_finish_kernel(const float* __restrict__ bpart, int rows, int C, float* __restrict__ db,
                                                               float* __restrict__ level2) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const int per_slice = (rows + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * per_slice, r1 = min(rows, r0 + per_slice);
    const int per = (max(r1 - r0, 0) + 15) >> 4;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < C) {
        const int jb = r0 + rg * per, je = min(r1, jb + per);
        int j = jb;
        for (; j + 3 < je; j += 4) {
            s0 += bpart[(size_t)j * C + c]; s1 += bpart[(size_t)(j + 1) * C + c];
            s2 += bpart[(size_t)(j + 2) * C + c]; s3 += bpart[(size_t)(j + 3) * C + c];
        }
        for (; j < je; ++j) s0 += bpart[(size_t)j * C + c];
    }
    red[rg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (threadIdx.x < 16 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][threadIdx.x];
        if (level2) level2[(size_t)blockIdx.y * C + c] = t;
        else db[c] += t;
    }
}
// dw[co][ky][kx][ci] += (G^T dU G)[ky][kx] of dU[f][co][ci]
__global__ void __launch_bounds__(256) wino_filtergrad_kernel(const float* __restrict__ dU, float* __restrict__ dw, int Co, int Ci) {
    const int64_t n = (int64_t)Co * Ci;
    dU += (size_t)blockIdx.y * 36 * n;            // blockIdx.y = phase
    dw += (size_t)blockIdx.y * 9 * n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int ci = (int)(i % Ci), co = (int)(i / Ci);
        float t[3][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {              // columns: G^T dU
            float u[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) u[r] = dU[((size_t)(r * 6 + j) * Co + co) * Ci + ci];
            float o[3];
            gt3(u, o);
#pragma unroll
            for (int r = 0; r < 3; ++r) t[r][j] = o[r];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {              // rows: (G^T dU) G
            float o[3];
            gt3(t[r], o);
#pragma unroll
            for (int j = 0; j < 3; ++j) dw[((size_t)(co * 3 + r) * 3 + j) * Ci + ci] += o[j];
        }
    }
}

bool wino_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("ACLGAN_NOWINO"); v = (e && atoi(e)) ? 0 : 1; }
    return v == 1;
}
bool wino_up5_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("ACLGAN_NOWINOUP5"); v = (e && atoi(e)) ? 0 : 1; }
    return v == 1;
}

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }
int grid_for(int64_t n, int cap) { return (int)std::min<int64_t>(cdiv64(n, 256), cap); }
const int WINO_BIAS_BLOCKS = 1024;       // x 256 threads: a multiple of every channel-group count in {16 .. 256}
const size_t WINO_BPART_BYTES = (size_t)WINO_BIAS_BLOCKS * 256 * 4 * sizeof(float);     // bias partial rows of one phase (4 channels per thread)
const size_t WINO_L2_BYTES = (size_t)64 * 1024 * sizeof(float);                          // second-level partials: 64 slices x up to 1024 channels

// channels per thread of the transform kernels (ACLGAN_WINO_VEC = 1 | 2 | 4; 2 measured best, profiles/r02_experiments.md)
int wino_vec() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("ACLGAN_WINO_VEC"); v = e ? atoi(e) : 2; if (v != 1 && v != 2 && v != 4) v = 2; }
    return v;
}
// split-bf16 GEMM slices (gemm_bf16x3.hip) for the forward-type products: ACLGAN_WINO_X3=1 / aclgan_set_tuning("wino_x3", 1).  OFF by
// default: the launch itself is 1.6x faster than the fp32 MFMA slices at fp32 accuracy (ResBlock shape 67.8 against 113 us isolated, 81
// against 102 us in the step: -8.4 ms of GEMM time per step), but the step does not get faster -- the transform writes 6 instead of 4
// bytes per value (+3.7 ms), the weight gradient loses the kept fp32 V (+1.6 ms) and, measured on the same box back to back, every
// OTHER matrix kernel of the step runs 5-8 % slower while these launches are in the mix (the chip is power-limited: the bf16 pipes at
// full rate cost the clocks of what follows): 121.3 against 119.9 ms of kernel time per step (profiles/r03_experiments.md).
std::atomic<int> g_wino_x3{-1};
bool wino_x3() {
    int v = g_wino_x3.load();
    if (v < 0) { const char* e = getenv("ACLGAN_WINO_X3"); v = e ? (atoi(e) ? 1 : 0) : 0; g_wino_x3.store(v); }
    return v == 1;
}
// channels per thread of the input transform when it writes the three bf16 planes (ACLGAN_WINO_VEC3)
int wino_vec3() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("ACLGAN_WINO_VEC3"); v = e ? atoi(e) : 2; if (v != 1 && v != 2 && v != 4) v = 2; }
    return v;
}
// planes: 0 = fp32 V; else the element count of one bf16 plane (3-plane output)
int launch_wino_input(const float* x, float* V, int B, const WViews& vs, int nph, int C, int TY, int TX, int off, int reflect, hipStream_t st, int64_t planes = 0) {
    const int nv = planes ? wino_vec3() : wino_vec();
    const dim3 grid(grid_for((int64_t)B * TY * TX * (C / nv), 16384), nph);
    if (nv == 4) hipLaunchKernelGGL(wino_input_kernel<4>, grid, dim3(256), 0, st, x, V, B, vs, C, TY, TX, off, reflect, planes);
    else if (nv == 2) hipLaunchKernelGGL(wino_input_kernel<2>, grid, dim3(256), 0, st, x, V, B, vs, C, TY, TX, off, reflect, planes);
    else hipLaunchKernelGGL(wino_input_kernel<1>, grid, dim3(256), 0, st, x, V, B, vs, C, TY, TX, off, reflect, planes);
    ACL_CHECK_LAUNCH("wino_input_kernel");
    return ACLGAN_OK;
}
int launch_wino_output(const float* M, const float* bias, float* y, int B, const WViews& vs, int C, int TY, int TX, int act, int accumulate, int sum,
                       hipStream_t st, float2* stats = nullptr) {
    const int nv = wino_vec();
    ACL_REQUIRE((int64_t)B * TY * TX * C < (1ll << 30), "wino_output: plane beyond 2^30 elements");
    const dim3 grid(grid_for((int64_t)B * TY * TX * (C / nv), 16384));
#define ACL_WO(NV_)                                                                                                                                     \
    do {                                                                                                                                                \
        if (sum) hipLaunchKernelGGL((wino_output_kernel<NV_, 1>), grid, dim3(256), 0, st, M, bias, y, B, vs, C, TY, TX, act, accumulate, stats);           \
        else hipLaunchKernelGGL((wino_output_kernel<NV_, 0>), grid, dim3(256), 0, st, M, bias, y, B, vs, C, TY, TX, act, accumulate, stats);             \
    } while (0)
    if (nv == 4) ACL_WO(4);
    else if (nv == 2) ACL_WO(2);
    else ACL_WO(1);
#undef ACL_WO
    ACL_CHECK_LAUNCH("wino_output_kernel");
    return ACLGAN_OK;
}
// returns the number of bias partial rows PER PHASE written to bpart ([ph][row][C]); C / nv must divide 256
int launch_wino_outgrad(const float* dy, float* dM, float* bpart, int B, const WViews& vs, int nph, int C, int TY, int TX, hipStream_t st, int* rows) {
    const int nv = wino_vec(), Cv = C / nv;
    const int mult = std::max(1, Cv / 256);        // gridDim.x * 256 must be a multiple of Cv
    int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(WINO_BIAS_BLOCKS, cdiv64((int64_t)B * TY * TX * Cv, 256)));
    blocks = cdiv(blocks, mult) * mult;
    *rows = blocks * 256 / Cv;
    const dim3 grid(blocks, nph);
    if (nv == 4) hipLaunchKernelGGL(wino_outgrad_kernel<4>, grid, dim3(256), 0, st, dy, dM, bpart, B, vs, C, TY, TX);
    else if (nv == 2) hipLaunchKernelGGL(wino_outgrad_kernel<2>, grid, dim3(256), 0, st, dy, dM, bpart, B, vs, C, TY, TX);
    else hipLaunchKernelGGL(wino_outgrad_kernel<1>, grid, dim3(256), 0, st, dy, dM, bpart, B, vs, C, TY, TX);
    ACL_CHECK_LAUNCH("wino_outgrad_kernel");
    return ACLGAN_OK;
}
// level2: WINO_BIAS_SLICES * C floats (carved from the tail of the partial buffer's allocation)
const int WINO_BIAS_SLICES = 64;
int launch_wino_bias_finish(const float* bpart, int rows, int C, float* db, float* level2, hipStream_t st) {
    if (rows >= 1024) {
        hipLaunchKernelGGL(wino_bias_finish_kernel, dim3(cdiv(C, 16), WINO_BIAS_SLICES), dim3(256), 0, st, bpart, rows, C, (float*)nullptr, level2);
        ACL_CHECK_LAUNCH("wino_bias_finish_kernel(level 1)");
        hipLaunchKernelGGL(wino_bias_finish_kernel, dim3(cdiv(C, 16), 1), dim3(256), 0, st, level2, WINO_BIAS_SLICES, C, db, (float*)nullptr);
    } else {
        hipLaunchKernelGGL(wino_bias_finish_kernel, dim3(cdiv(C, 16), 1), dim3(256), 0, st, bpart, rows, C, db, (float*)nullptr);
    }
    ACL_CHECK_LAUNCH("wino_bias_finish_kernel");
    return ACLGAN_OK;
}
bool wino_bias_ok(int C) { const int Cv = C / wino_vec(); return C % wino_vec() == 0 && (256 % Cv == 0 || Cv % 256 == 0); }

WViews one_view(const WView& v) { WViews w; w.v[0] = v; w.v[1] = v; w.v[2] = v; w.v[3] = v; w.nph = 1; return w; }

// carve `bytes` (256-aligned) off a scratch cursor
float* take(char*& cur, size_t bytes) { float* p = (float*)cur; cur += align256(bytes); return p; }

}  // namespace

// ------------------------------------------------------------------------------------------
// 3x3 stride-1 reflect-pad-1 layers (ResBlocks)
// ------------------------------------------------------------------------------------------
// 3x3, stride 1, reflect pad 1, no upsample, 4x4-tileable output, channel counts the GEMM kernel takes, and enough channels to pay
bool conv_wino_ok(const ConvGeom& g) {
    return wino_enabled() && g.k == 3 && g.s == 1 && g.p == 1 && g.up == 0 && g.Ho % 4 == 0 && g.Wo % 4 == 0 && g.Ci % 16 == 0 && g.Co % 16 == 0 &&
           (int64_t)g.Ci * g.Co >= 64 * 64;
}
// (U and V slots are sized for the three bf16 planes of the split-bf16 GEMM -- 6 bytes per value -- whether or not it runs; a layer's
//  cached U is fp32 or planes for the whole update: the choice depends on the layer's shape and the process-wide switch only)
// tuning / test knob behind aclgan_set_tuning("wino_x3", v); returns the previous value
int set_wino_x3(int v) { const int old = wino_x3() ? 1 : 0; g_wino_x3.store(v ? 1 : 0); return old; }
size_t conv_wino_u_bytes(const ConvGeom& g) {
    if (conv_wino_ok(g)) return align256((size_t)36 * g.Co * g.Ci * 6);
    if (conv_up5_wino_ok(g)) return align256((size_t)144 * g.Co * g.Ci * 6);
    if (conv_s2k4_wino_ok(g, 0) || conv_s2k4_wino_ok(g, 1)) return align256(wino_fused_s2k4_u_bytes(g.Ci, g.Co));
    return 0;
}
size_t conv_wino_scratch_bytes(const ConvGeom& g) {
    if (!conv_wino_ok(g)) return 0;
    const int64_t T = (int64_t)g.B * (g.Ho / 4) * (g.Wo / 4);
    const int cmax = std::max(g.Ci, g.Co);      // forward: V has Cin, M has Cout channels; dgrad the other way round
    return align256((size_t)36 * g.Co * g.Ci * 6) + align256((size_t)36 * T * cmax * 6) + align256((size_t)36 * T * cmax * sizeof(float)) + 256;
}
namespace {
// filter transform -> input transform -> 36 GEMMs -> output transform.  `in` has Cin_ channels, `out` Cout_.
int wino_run(int B, int H, int W, int Cin_, int Cout_, const float* in, const float* w, int w_co, int w_ci, int flip, const float* bias, float* out,
             int act, int accumulate, int reflect, void* scratch, hipStream_t st, float2* stats = nullptr, float* keepV = nullptr) {
    const int TY = H / 4, TX = W / 4;
    const int64_t T = (int64_t)B * TY * TX;
    char* cur = (char*)scratch;
    if (!keepV && wino_fused_ok(B, H, W, Cin_, Cout_, act)) {      // one launch (conv_wino_fused.hip): neither V nor M exists in memory
        bool fresh_f = true;
        float* Uf = cached_u((float*)scratch, w, uvar(flip ? 1 : 0, U_FRAG), (size_t)36 * Cout_ * Cin_ * 4, &fresh_f);
        if (fresh_f) { const int rcf = wino_fused_filter(w, Uf, w_co, w_ci, flip, st); if (rcf) return rcf; }
        return wino_fused_launch(B, H, W, Cin_, Cout_, in, Uf, bias, out, act, accumulate, reflect, stats, st);
    }
    const bool x3 = !keepV && wino_x3() && gemm_x3_shape_ok((int)T, Cin_, Cout_);      // (a kept V feeds the fp32 weight-gradient GEMM: fp32)
    const size_t eb = x3 ? 6 : 4;
    float* U = take(cur, (size_t)36 * Cout_ * Cin_ * 6);
    float* V = take(cur, (size_t)36 * T * Cin_ * eb);
    float* M = take(cur, (size_t)36 * T * Cout_ * 4);
    if (keepV) V = keepV;         // the caller keeps the input transform for the weight gradient of the same layer
    const WViews vw = one_view(ident_view(H, W));
    const int64_t uplanes = x3 ? (int64_t)36 * Cout_ * Cin_ : 0, vplanes = x3 ? (int64_t)36 * T * Cin_ : 0;
    bool fresh = true;
    U = cached_u(U, w, uvar(flip ? 1 : 0, x3 ? U_X3 : U_PIPE), (size_t)36 * Cout_ * Cin_ * eb, &fresh);
    if (fresh) {
        hipLaunchKernelGGL(wino_filter_kernel, dim3(grid_for((int64_t)w_co * w_ci, 4096), 1), dim3(256), 0, st, w, U, w_co, w_ci, flip, uplanes,
                           (int64_t)0, (int64_t)0, (int64_t)0);
        ACL_CHECK_LAUNCH("wino_filter_kernel");
    }
    int rc = launch_wino_input(in, V, B, vw, 1, Cin_, TY, TX, -1, reflect, st, vplanes);
    if (rc) return rc;
    rc = x3 ? gemm_slices_x3(V, (size_t)vplanes * 2, U, (size_t)uplanes * 2, M, (int)T, Cin_, Cout_, 36, 0, st) : gemm_slices_f32(V, U, M, (int)T, Cin_, Cout_, 36, 0, st);
    if (rc) return rc;
    return launch_wino_output(M, bias, out, B, vw, Cout_, TY, TX, act, accumulate, 0, st, stats);
}
}  // namespace
// Which cache entry (variant | layout << 4) the forward (dgrad = 0; keepV: the caller keeps the input transform) / the input gradient
// (dgrad = 1) of this 3x3 layer will ask for -- the same decisions as wino_run, for the scheduler's batched prefill.  -1: not a Winograd layer.
int conv_wino_u_variant(const ConvGeom& g, int dgrad, bool keepV) {
    if (!conv_wino_ok(g)) return -1;
    const int Cin_ = dgrad ? g.Co : g.Ci, Cout_ = dgrad ? g.Ci : g.Co, act = dgrad ? ACLGAN_ACT_NONE : g.act;
    const int64_t T = (int64_t)g.B * (g.Hi / 4) * (g.Wi / 4);
    if (dgrad) keepV = false;
    if (!keepV && wino_fused_ok(g.B, g.Hi, g.Wi, Cin_, Cout_, act)) return uvar(dgrad, U_FRAG);
    const bool x3 = !keepV && wino_x3() && gemm_x3_shape_ok((int)T, Cin_, Cout_);
    return uvar(dgrad, x3 ? U_X3 : U_PIPE);
}
// The transforms of `count` equally shaped 3x3 filters (w0 + i * w_stride floats, OHWI) into U0 + i * u_stride floats, in the layout of
// cache entry `uv`: ONE launch (the ~100 per-filter launches of a step were pure dispatch latency: 6 - 16 us each for 2.4 MB of output)
int conv_wino_prefill(const ConvGeom& g, int uv, const float* w0, int64_t w_stride, float* U0, int64_t u_stride, int count, hipStream_t st) {
    if (!conv_wino_ok(g) || uv < 0 || count < 1) return ACLGAN_EUNSUPPORTED;
    const int flip = uv & 1, layout = uv >> 4;
    if (layout == U_FRAG) return wino_fused_filter_batch(w0, w_stride, U0, u_stride, count, g.Co, g.Ci, flip, st);
    const int64_t planes = layout == U_X3 ? (int64_t)36 * g.Co * g.Ci : 0;
    hipLaunchKernelGGL(wino_filter_kernel, dim3(grid_for((int64_t)g.Co * g.Ci, 4096), count), dim3(256), 0, st, w0, U0, g.Co, g.Ci, flip, planes,
                       w_stride, u_stride, u_stride * 2);
    ACL_CHECK_LAUNCH("wino_filter_kernel(batch)");
    return ACLGAN_OK;
}
// stats (optional): [B][Ho/4 * Wo/4][Co] (mean, M2) pairs of the 4x4 output tiles -- the normalisation layer's chunk partials, for free
// keepV (optional, conv_wino_keep_bytes(g) bytes): receives V = B^T x B -- conv_wgrad_wino(.., haveV) of the same layer skips its
// input transform
size_t conv_wino_keep_bytes(const ConvGeom& g) {
    if (!conv_wino_ok(g) || g.Co % 64 != 0 || g.Ci % 64 != 0) return 0;
    if (wino_x3() && gemm_x3_shape_ok(1, g.Ci, g.Co)) return 0;      // the forward writes V as bf16 planes; the fp32 weight-gradient GEMM transforms x itself
    if (wino_fused_ok(g.B, g.Hi, g.Wi, g.Ci, g.Co, g.act)) return 0;         // the fused forward never materialises V
    if (wino_wgrad_fused_ok(g)) return 0;                                    // the fused weight gradient transforms x itself
    return align256((size_t)36 * g.B * (g.Ho / 4) * (g.Wo / 4) * g.Ci * sizeof(float));
}
int conv_fwd_wino(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, void* scratch, hipStream_t st, float* stats, float* keepV) {
    if (!conv_wino_ok(g) || !scratch) return ACLGAN_EUNSUPPORTED;
    return wino_run(g.B, g.Hi, g.Wi, g.Ci, g.Co, x, w, g.Co, g.Ci, 0, bias, y, g.act, 0, 1, scratch, st, (float2*)stats, keepV);
}
// the INTERIOR of the padded-grid gradient (= dx without the mirrored halo contributions): dx (+)= dy (*) flipped w^T, zero padding
int conv_dgrad_wino_interior(const ConvGeom& g, const float* dy, const float* w, float* dx, int accumulate, void* scratch, hipStream_t st) {
    if (!conv_wino_ok(g) || !scratch) return ACLGAN_EUNSUPPORTED;
    return wino_run(g.B, g.Hi, g.Wi, g.Co, g.Ci, dy, w, g.Co, g.Ci, 1, nullptr, dx, ACLGAN_ACT_NONE, accumulate, 0, scratch, st);
}

// weight (and bias) gradient.  scratch: V [36][T][Cin] | dM [36][T][Cout] | dU [36][Cout][Cin] | bias partials | GEMM partial tiles
size_t conv_wgrad_wino_scratch_bytes(const ConvGeom& g) {
    if (!conv_wino_ok(g) || g.Co % 64 != 0 || g.Ci % 64 != 0) return 0;
    const int64_t T = (int64_t)g.B * (g.Ho / 4) * (g.Wo / 4);
    const size_t pipe = align256((size_t)36 * T * g.Ci * 4) + align256((size_t)36 * T * g.Co * 4) + align256((size_t)36 * g.Co * g.Ci * 4) +
                        align256(WINO_BPART_BYTES + WINO_L2_BYTES) + gemm_at_b_slices_scratch((int)T, g.Co, g.Ci, 36) + 256;
    // (sized for either path: the fused kernel is chosen per call -- the tuning switch may change between sizing and launch)
    return std::max(pipe, wino_wgrad_fused_scratch_bytes(g));
}
int conv_wgrad_wino(const ConvGeom& g, const float* x, const float* dy, float* dw, float* db, void* scratch, hipStream_t st, const float* haveV) {
    if (!conv_wino_ok(g) || g.Co % 64 != 0 || g.Ci % 64 != 0 || !scratch || !dw) return ACLGAN_EUNSUPPORTED;
    if (wino_wgrad_fused_ok(g)) return wino_wgrad_fused(g, x, dy, dw, db, scratch, st);      // one kernel: neither V nor dM nor dU in memory
    if (!wino_bias_ok(g.Co)) return ACLGAN_EUNSUPPORTED;
    const int TY = g.Ho / 4, TX = g.Wo / 4;
    const int64_t T = (int64_t)g.B * TY * TX;
    char* cur = (char*)scratch;
    float* V = take(cur, (size_t)36 * T * g.Ci * 4);
    float* dM = take(cur, (size_t)36 * T * g.Co * 4);
    float* dU = take(cur, (size_t)36 * g.Co * g.Ci * 4);
    float* bpart = take(cur, WINO_BPART_BYTES + WINO_L2_BYTES);
    void* part = cur;
    const WViews vw = one_view(ident_view(g.Hi, g.Wi));
    int rc0 = ACLGAN_OK;
    if (haveV) V = const_cast<float*>(haveV);      // the forward pass of this layer kept its input transform (read-only here)
    else rc0 = launch_wino_input(x, V, g.B, vw, 1, g.Ci, TY, TX, -1, 1, st);
    if (rc0) return rc0;
    int brows = 0;
    rc0 = launch_wino_outgrad(dy, dM, db ? bpart : (float*)nullptr, g.B, vw, 1, g.Co, TY, TX, st, &brows);
    if (rc0) return rc0;
    if (db) {
        const int rcb = launch_wino_bias_finish(bpart, brows, g.Co, db, bpart + WINO_BPART_BYTES / sizeof(float), st);
        if (rcb) return rcb;
    }
    const int rc = gemm_at_b_slices_f32(dM, V, dU, (int)T, g.Co, g.Ci, 36, 0, part, st);
    if (rc) return rc;
    hipLaunchKernelGGL(wino_filtergrad_kernel, dim3(grid_for((int64_t)g.Co * g.Ci, 4096), 1), dim3(256), 0, st, dU, dw, g.Co, g.Ci);
    ACL_CHECK_LAUNCH("wino_filtergrad_kernel");
    return ACLGAN_OK;
}

// ------------------------------------------------------------------------------------------
// the four sub-pixel phases of "Upsample(2) + ReflectionPad2d(2) + Conv2d(5x5)" (conv_fast.hip: up5_*): each phase is a VALID 3x3
// convolution of the low-res input with a merged filter wp[phase] -- i.e. Winograd material.  All four phases share the input
// transform (forward, wgrad) and run as ONE batched GEMM launch of 4 x 36 = 144 slices.  The output ring of width 2 keeps the exact
// gather kernels (launched by the caller).  ACLGAN_NOWINOUP5=1 keeps the direct phase kernels.
// ------------------------------------------------------------------------------------------
bool conv_up5_wino_ok(const ConvGeom& g) {
    return wino_enabled() && wino_up5_enabled() && g.up == 1 && g.k == 5 && g.p == 2 && g.s == 1 && g.Hi >= 6 && g.Wi >= 6 && g.Ci % 16 == 0 &&
           g.Co % 16 == 0 && (int64_t)g.Ci * g.Co >= 64 * 64;
}
namespace {
struct Up5Geo { int Hv, Wv, TY, TX, TYd, TXd; int64_t T, Td; WViews ph; };
Up5Geo up5_geo(const ConvGeom& g) {
    Up5Geo q;
    q.Hv = g.Hi - 2; q.Wv = g.Wi - 2;                       // VALID outputs per phase
    q.TY = cdiv(q.Hv, 4); q.TX = cdiv(q.Wv, 4); q.T = (int64_t)g.B * q.TY * q.TX;
    q.TYd = cdiv(g.Hi, 4); q.TXd = cdiv(g.Wi, 4); q.Td = (int64_t)g.B * q.TYd * q.TXd;       // dgrad tiles the low-res dx
    for (int p = 0; p < 4; ++p) q.ph.v[p] = phase_view(q.Hv, q.Wv, p >> 1, p & 1, g.Ho, g.Wo);
    q.ph.nph = 4;
    return q;
}
}  // namespace
// forward: U [4][36][Co][Ci] | V [36][T][Ci] | M [4][36][T][Co]
// ... and of the merged phase filters of a sub-pixel layer (what conv_up5_wino_fwd_phases / _dgrad_phases will ask for)
int conv_up5_wino_u_variant(const ConvGeom& g, int dgrad, bool keepV) {
    if (!conv_up5_wino_ok(g)) return -1;
    const Up5Geo q = up5_geo(g);
    if (!dgrad) {
        if (!keepV && wino_fused_ok(g.B, g.Hi - 2, g.Wi - 2, g.Ci, g.Co, g.act, 4, 1)) return uvar(2, U_FRAG);
        return uvar(2, (!keepV && wino_x3() && gemm_x3_shape_ok((int)q.T, g.Ci, g.Co)) ? U_X3 : U_PIPE);
    }
    if (wino_fused_ok(g.B, g.Hi, g.Wi, g.Co, g.Ci, ACLGAN_ACT_NONE, 1, 4)) return uvar(3, U_FRAG);
    return uvar(3, (wino_x3() && gemm_x3_shape_ok((int)q.Td, g.Co, g.Ci)) ? U_X3 : U_PIPE);
}
size_t conv_up5_wino_fwd_scratch_bytes(const ConvGeom& g) {
    if (!conv_up5_wino_ok(g)) return 0;
    const Up5Geo q = up5_geo(g);
    return align256((size_t)144 * g.Co * g.Ci * 6) + align256((size_t)36 * q.T * g.Ci * 6) + align256((size_t)144 * q.T * g.Co * 4) + 256;
}
size_t conv_up5_wino_keep_bytes(const ConvGeom& g) {
    if (!conv_up5_wino_ok(g) || g.Co % 64 != 0 || g.Ci % 64 != 0) return 0;
    if (wino_x3() && gemm_x3_shape_ok(1, g.Ci, g.Co)) return 0;
    if (wino_fused_ok(g.B, g.Hi - 2, g.Wi - 2, g.Ci, g.Co, g.act, 4, 1)) return 0;      // the fused forward never materialises V
    return align256((size_t)36 * up5_geo(g).T * g.Ci * sizeof(float));
}
int conv_up5_wino_fwd_phases(const ConvGeom& g, const float* x, const float* wp, const float* bias, float* y, void* scratch, hipStream_t st, float* keepV,
                             const float* wkey) {
    if (!conv_up5_wino_ok(g) || !scratch) return ACLGAN_EUNSUPPORTED;
    const Up5Geo q = up5_geo(g);
    char* cur = (char*)scratch;
    if (!keepV && wino_fused_ok(g.B, g.Hi - 2, g.Wi - 2, g.Ci, g.Co, g.act, 4, 1)) {      // the four phases in one fused launch (conv_wino_fused.hip)
        bool fresh_f = true;
        float* Uf = cached_u((float*)scratch, wkey, uvar(2, U_FRAG), (size_t)144 * g.Co * g.Ci * 4, &fresh_f);
        if (fresh_f) { const int rcf = wino_fused_filter(wp, Uf, g.Co, g.Ci, 0, st, 4); if (rcf) return rcf; }
        return wino_fused_up5_fwd(g.B, g.Hi, g.Wi, g.Ci, g.Co, x, Uf, bias, y, g.Ho, g.Wo, g.act, st);
    }
    const bool x3 = !keepV && wino_x3() && gemm_x3_shape_ok((int)q.T, g.Ci, g.Co);
    const size_t eb = x3 ? 6 : 4;
    float* U = take(cur, (size_t)144 * g.Co * g.Ci * 6);
    float* V = take(cur, (size_t)36 * q.T * g.Ci * 6);
    float* M = take(cur, (size_t)144 * q.T * g.Co * 4);
    if (keepV) V = keepV;
    const int64_t uplanes = x3 ? (int64_t)144 * g.Co * g.Ci : 0, vplanes = x3 ? (int64_t)36 * q.T * g.Ci : 0;
    bool fresh = true;
    U = cached_u(U, wkey, uvar(2, x3 ? U_X3 : U_PIPE), (size_t)144 * g.Co * g.Ci * eb, &fresh);
    if (fresh) {
        hipLaunchKernelGGL(wino_filter_kernel, dim3(grid_for((int64_t)g.Co * g.Ci, 4096), 4), dim3(256), 0, st, wp, U, g.Co, g.Ci, 0, uplanes,
                           (int64_t)g.Co * 9 * g.Ci, (int64_t)36 * g.Co * g.Ci, (int64_t)36 * g.Co * g.Ci);
        ACL_CHECK_LAUNCH("wino_filter_kernel(up5)");
    }
    int rc = launch_wino_input(x, V, g.B, one_view(ident_view(g.Hi, g.Wi)), 1, g.Ci, q.TY, q.TX, 0, 0, st, vplanes);
    if (rc) return rc;
    // the 4 phases share V: A operand of slice f = slice f % 36
    rc = x3 ? gemm_slices_x3(V, (size_t)vplanes * 2, U, (size_t)uplanes * 2, M, (int)q.T, g.Ci, g.Co, 144, 36, st) : gemm_slices_f32(V, U, M, (int)q.T, g.Ci, g.Co, 144, 36, st);
    if (rc) return rc;
    return launch_wino_output(M, bias, y, g.B, q.ph, g.Co, q.TY, q.TX, g.act, 0, 0, st);
}
// dgrad: U' [4][36][Ci][Co] | V' [4][36][Td][Co] | M [4][36][Td][Ci];  dx (+)= sum over phases (full correlation with the flipped filter)
size_t conv_up5_wino_dgrad_scratch_bytes(const ConvGeom& g) {
    if (!conv_up5_wino_ok(g)) return 0;
    const Up5Geo q = up5_geo(g);
    return align256((size_t)144 * g.Co * g.Ci * 6) + align256((size_t)144 * q.Td * g.Co * 6) + align256((size_t)144 * q.Td * g.Ci * 4) + 256;
}
int conv_up5_wino_dgrad_phases(const ConvGeom& g, const float* dy, const float* wp, float* dx, int accumulate, void* scratch, hipStream_t st, const float* wkey) {
    if (!conv_up5_wino_ok(g) || !scratch) return ACLGAN_EUNSUPPORTED;
    const Up5Geo q = up5_geo(g);
    char* cur = (char*)scratch;
    if (wino_fused_ok(g.B, g.Hi, g.Wi, g.Co, g.Ci, ACLGAN_ACT_NONE, 1, 4)) {      // one fused launch: K loop over (phase, cout)
        bool fresh_f = true;
        float* Uf = cached_u((float*)scratch, wkey, uvar(3, U_FRAG), (size_t)144 * g.Co * g.Ci * 4, &fresh_f);
        if (fresh_f) { const int rcf = wino_fused_filter(wp, Uf, g.Co, g.Ci, 1, st, 4); if (rcf) return rcf; }
        return wino_fused_up5_dgrad(g.B, g.Hi, g.Wi, g.Ci, g.Co, dy, Uf, dx, g.Ho, g.Wo, accumulate, st);
    }
    const bool x3 = wino_x3() && gemm_x3_shape_ok((int)q.Td, g.Co, g.Ci);
    const size_t eb = x3 ? 6 : 4;
    float* U = take(cur, (size_t)144 * g.Co * g.Ci * 6);
    float* V = take(cur, (size_t)144 * q.Td * g.Co * 6);
    float* M = take(cur, (size_t)144 * q.Td * g.Ci * 4);
    const int64_t uplanes = x3 ? (int64_t)144 * g.Co * g.Ci : 0, vplanes = x3 ? (int64_t)144 * q.Td * g.Co : 0;
    bool fresh = true;
    U = cached_u(U, wkey, uvar(3, x3 ? U_X3 : U_PIPE), (size_t)144 * g.Co * g.Ci * eb, &fresh);
    if (fresh) {
        hipLaunchKernelGGL(wino_filter_kernel, dim3(grid_for((int64_t)g.Co * g.Ci, 4096), 4), dim3(256), 0, st, wp, U, g.Co, g.Ci, 1, uplanes,
                           (int64_t)g.Co * 9 * g.Ci, (int64_t)36 * g.Co * g.Ci, (int64_t)36 * g.Co * g.Ci);
        ACL_CHECK_LAUNCH("wino_filter_kernel(up5 dgrad)");
    }
    // dx[u] = sum_k wflip[k] dy_phase[u - 2 + k]: patches start 2 before the tile, zero outside the 62 x 62 phase view
    int rc = launch_wino_input(dy, V, g.B, q.ph, 4, g.Co, q.TYd, q.TXd, -2, 0, st, vplanes);
    if (rc) return rc;
    rc = x3 ? gemm_slices_x3(V, (size_t)vplanes * 2, U, (size_t)uplanes * 2, M, (int)q.Td, g.Co, g.Ci, 144, 0, st) : gemm_slices_f32(V, U, M, (int)q.Td, g.Co, g.Ci, 144, 0, st);
    if (rc) return rc;
    WViews dst = one_view(ident_view(g.Hi, g.Wi));
    dst.nph = 4;
    return launch_wino_output(M, nullptr, dx, g.B, dst, g.Ci, q.TYd, q.TXd, ACLGAN_ACT_NONE, accumulate, 1, st);
}
// wgrad: V [36][T][Ci] | dM [4][36][T][Co] | dU [4][36][Co][Ci] | bias partials [4][...] | GEMM partial tiles;  dwp[phase] += G^T dU G
size_t conv_up5_wino_wgrad_scratch_bytes(const ConvGeom& g) {
    if (!conv_up5_wino_ok(g) || g.Co % 64 != 0 || g.Ci % 64 != 0) return 0;
    const Up5Geo q = up5_geo(g);
    return align256((size_t)36 * q.T * g.Ci * 4) + align256((size_t)144 * q.T * g.Co * 4) + align256((size_t)144 * g.Co * g.Ci * 4) +
           align256(4 * WINO_BPART_BYTES + WINO_L2_BYTES) + gemm_at_b_slices_scratch((int)q.T, g.Co, g.Ci, 144) + 256;
}
int conv_up5_wino_wgrad_phases(const ConvGeom& g, const float* x, const float* dy, float* dwp, float* db, void* scratch, hipStream_t st, const float* haveV) {
    if (!conv_up5_wino_ok(g) || g.Co % 64 != 0 || g.Ci % 64 != 0 || !scratch || !wino_bias_ok(g.Co)) return ACLGAN_EUNSUPPORTED;
    const Up5Geo q = up5_geo(g);
    char* cur = (char*)scratch;
    float* V = take(cur, (size_t)36 * q.T * g.Ci * 4);
    float* dM = take(cur, (size_t)144 * q.T * g.Co * 4);
    float* dU = take(cur, (size_t)144 * g.Co * g.Ci * 4);
    float* bpart = take(cur, 4 * WINO_BPART_BYTES + WINO_L2_BYTES);
    void* part = cur;
    int rc0 = ACLGAN_OK;
    if (haveV) V = const_cast<float*>(haveV);
    else rc0 = launch_wino_input(x, V, g.B, one_view(ident_view(g.Hi, g.Wi)), 1, g.Ci, q.TY, q.TX, 0, 0, st);
    if (rc0) return rc0;
    int brows = 0;
    rc0 = launch_wino_outgrad(dy, dM, db ? bpart : (float*)nullptr, g.B, q.ph, 4, g.Co, q.TY, q.TX, st, &brows);
    if (rc0) return rc0;
    if (db) {      // interior pixels (the four phases); the ring launch of the caller adds the ring pixels
        const int rcb = launch_wino_bias_finish(bpart, 4 * brows, g.Co, db, bpart + 4 * WINO_BPART_BYTES / sizeof(float), st);
        if (rcb) return rcb;
    }
    const int rc = gemm_at_b_slices_f32(dM, V, dU, (int)q.T, g.Co, g.Ci, 144, 36, part, st);     // V shared by the phases
    if (rc) return rc;
    hipLaunchKernelGGL(wino_filtergrad_kernel, dim3(grid_for((int64_t)g.Co * g.Ci, 4096), 4), dim3(256), 0, st, dU, dwp, g.Co, g.Ci);
    ACL_CHECK_LAUNCH("wino_filtergrad_kernel(up5)");
    return ACLGAN_OK;
}

// ------------------------------------------------------------------------------------------
// Round 6: the 4x4 stride-2 reflect-pad-1 layers (ContentEncoder / StyleEncoder downsampling, MsImageDis layers 1..3: networks.py:41,
// 216-221, 236-241) as four parity phases of the fused F(4x4,3x3) kernel (conv_wino_fused.hip: wino_fused_s2k4_*).  Forward in one launch
// (K loop over phase x Cin, statistics of the following InstanceNorm from the epilogue), interior of the input gradient in one launch (four
// grid phases writing the parity views of dx); the mirrored halo ring keeps the direct kernel's small launch.  ACLGAN_NOWINOS2=1 keeps the
// direct kernels; the per-shape decision is wino_fused_s2k4_ok's cost model (aclgan_tuning("wino_fused", 2) takes every eligible shape).
// ------------------------------------------------------------------------------------------
// tuning switch "wino_s2k4" (aclgan_tuning; ACLGAN_NOWINOS2=1 sets the default to 0): 0 = the stride-2 layers stay on the direct kernels
static std::atomic<int> g_wino_s2k4{-1};
int wino_s2k4_setting() {
    int v = g_wino_s2k4.load();
    if (v < 0) { const char* e = getenv("ACLGAN_NOWINOS2"); v = (e && atoi(e)) ? 0 : 1; g_wino_s2k4.store(v); }
    return v;
}
int set_wino_s2k4(int v) { const int old = wino_s2k4_setting(); g_wino_s2k4.store(v ? 1 : 0); return old; }
bool conv_s2k4_wino_ok(const ConvGeom& g, int which) {
    if (!wino_s2k4_setting() || !wino_enabled() || g.k != 4 || g.s != 2 || g.p != 1 || g.up != 0) return false;
    if (which == 1 && deterministic()) return false;      // (the ordered ring fold of that mode is wired for the 3x3 layers only)
    return wino_fused_s2k4_ok(g.B, g.Hi, g.Wi, g.Ci, g.Co, which ? ACLGAN_ACT_NONE : g.act, which);
}
size_t conv_s2k4_wino_scratch_bytes(const ConvGeom& g) {
    return (conv_s2k4_wino_ok(g, 0) || conv_s2k4_wino_ok(g, 1)) ? align256(wino_fused_s2k4_u_bytes(g.Ci, g.Co)) + 256 : 0;
}
int conv_fwd_s2k4_wino(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, void* scratch, hipStream_t st, float* stats) {
    if (!conv_s2k4_wino_ok(g, 0) || !scratch) return ACLGAN_EUNSUPPORTED;
    bool fresh = true;
    float* Uf = cached_u((float*)scratch, w, uvar(0, U_FRAG), wino_fused_s2k4_u_bytes(g.Ci, g.Co), &fresh);
    if (fresh) { const int rcf = wino_fused_filter_s2k4(w, Uf, g.Co, g.Ci, 0, st); if (rcf) return rcf; }
    return wino_fused_s2k4_fwd(g.B, g.Hi, g.Wi, g.Ci, g.Co, x, Uf, bias, y, g.act, (float2*)stats, st);
}
int conv_dgrad_s2k4_wino_interior(const ConvGeom& g, const float* dy, const float* w, float* dx, int accumulate, void* scratch, hipStream_t st) {
    if (!conv_s2k4_wino_ok(g, 1) || !scratch) return ACLGAN_EUNSUPPORTED;
    bool fresh = true;
    float* Uf = cached_u((float*)scratch, w, uvar(1, U_FRAG), wino_fused_s2k4_u_bytes(g.Ci, g.Co), &fresh);
    if (fresh) { const int rcf = wino_fused_filter_s2k4(w, Uf, g.Co, g.Ci, 1, st); if (rcf) return rcf; }
    return wino_fused_s2k4_dgrad(g.B, g.Hi, g.Wi, g.Ci, g.Co, dy, Uf, dx, accumulate, st);
}

}  // namespace aclgan
