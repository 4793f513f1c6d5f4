// elementwise.hip -- the HBM-bound kernels of the ACL-GAN step (gfx950): normalisation layers
// (+activation, +residual) forward/backward, pooling, focus blend, losses, Adam, small dense
// layers, layout conversion.  Activations NHWC ([B][HW][C]), stored fp32 or -- wide layers of the 16-bit compute dtypes -- bf16 / fp16
// (st16.h: storage code per tensor, statistics / coefficients / arithmetic always fp32).
//
// Reference semantics restated here (file:line into the reference tree):
//   InstanceNorm2d(affine=False)          networks.py:333   biased var, 1/sqrt(var+1e-5)
//   AdaptiveInstanceNorm2d                networks.py:491-503 (F.batch_norm on a (1,B*C,H,W) view)
//   custom LayerNorm                      networks.py:520-536 unbiased std, 1/(std+1e-5), gamma/beta per channel
//   ResBlock  out += residual             networks.py:309
//   AvgPool2d(3,2,1,count_include_pad=F)  networks.py:33
//   focus_translation / focus losses      trainer.py:85-88, 146-161
//   LSGAN losses / L1                     networks.py:67,83,98 / trainer.py:61-62
//   Adam (L2 weight decay)                trainer.py:39-42
#include "common.h"
#include "st16.h"

namespace aclgan {

__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == ACLGAN_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACLGAN_ACT_LRELU) return v > 0.f ? v : 0.2f * v;
    if (act == ACLGAN_ACT_TANH) return tanhf(v);
    return v;
}
// derivative of the activation expressed through its OUTPUT y
__device__ __forceinline__ float act_grad(float y, int act) {
    if (act == ACLGAN_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == ACLGAN_ACT_LRELU) return y > 0.f ? 1.f : 0.2f;
    if (act == ACLGAN_ACT_TANH) return 1.f - y * y;
    return 1.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// block-wide sum for 256-thread blocks; result valid in every thread
__device__ __forceinline__ float block_sum256(float v, float* red /*[4]*/) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// Chan et al. pairwise combination of (count, mean, M2)
__device__ __forceinline__ void chan_combine(float& n, float& mean, float& m2, float nb, float meanb, float m2b) {
    if (nb == 0.f) return;
    if (n == 0.f) { n = nb; mean = meanb; m2 = m2b; return; }
    const float nt = n + nb, d = meanb - mean;
    mean += d * (nb / nt);
    m2 += m2b + d * d * (n * nb / nt);
    n = nt;
}

// ------------------------------------------------------------------------------------------
// normalisation forward
// ------------------------------------------------------------------------------------------
static int norm_chunk_pixels(int B, int HW) {
    // ~2048 workgroups over the batch, 32..1024 pixels per workgroup
    int c = (int)(((int64_t)B * HW + 2047) / 2048);
    c = (c + 15) / 16 * 16;
    if (c < 32) c = 32;
    if (c > 1024) c = 1024;
    if (c > HW) c = HW;
    return c;
}

// Storage template parameters of the streaming kernels below: the storage codes of the operands (ST_F32 / ST_BF16 / ST_F16) compiled in -- one
// code, or a pair (the layers at the 16-bit / fp32 boundary: x fp32 and y 16-bit, dy fp32 and x 16-bit ...) -- or ST_MIXED = per-operand
// run-time codes.  NU loads per operand are kept in flight.  With run-time codes that takes st_ld4n (the switch ONCE around an operand's NU
// loads) and still waits operand by operand (measured at fp16 B=32: 141 us against 70 for the compiled-in pair); a switch around every single
// load is waited for at its end, and unrolling buys registers and nothing else (the hand-unrolled reduce with per-load switches was 6 .. 19 %
// SLOWER than the plain loop).
constexpr int NU = 4;
constexpr int ST_MIXED = 3;
// kernels with TWO compiled-in storage codes (e.g. x | everything else): the pairs that occur -- equal codes, and fp32 on one side of the 16-bit /
// fp32 boundary (the thin image-side layers compute and store fp32, their wide neighbours store the 16-bit compute dtype).  -1: run-time codes.
static inline int st_pair(int a, int b) { return (a == b || a == ST_F32 || b == ST_F32) ? a * 3 + b : -1; }
#define ACL_ST_PAIR_DISPATCH(pair, LAUNCH)                                            \
    do {                                                                              \
        switch (pair) {                                                               \
            case 0: LAUNCH(ST_F32, ST_F32); break;                                    \
            case 1: LAUNCH(ST_F32, ST_BF16); break;                                   \
            case 2: LAUNCH(ST_F32, ST_F16); break;                                    \
            case 3: LAUNCH(ST_BF16, ST_F32); break;                                   \
            case 4: LAUNCH(ST_BF16, ST_BF16); break;                                  \
            case 6: LAUNCH(ST_F16, ST_F32); break;                                    \
            case 8: LAUNCH(ST_F16, ST_F16); break;                                    \
            default: LAUNCH(ST_MIXED, ST_MIXED); break;                               \
        }                                                                             \
    } while (0)

// partial statistics: part[b][chunk][c] = (mean, M2) over the chunk's pixels.  Threads are laid
// out C/4 float4-lanes wide (coalesced 16 B/lane along the channel axis), 256/(C/4) pixels deep.
template <int SM>
__global__ void __launch_bounds__(256) norm_stats_kernel(const void* __restrict__ x, int xst_rt, float2* __restrict__ part,
                                                         int HW, int C, int chunk, int nchunks) {
    const int xst = SM == ST_MIXED ? xst_rt : SM;
    const int C4 = C >> 2;
    const int b = blockIdx.y, ch = blockIdx.x;
    const int p0 = ch * chunk, p1 = min(HW, p0 + chunk);
    const int cg = threadIdx.x % C4, pl = threadIdx.x / C4, PL = 256 / C4;
    const int64_t xb = (int64_t)b * HW * C4;
    // shift = the chunk's first pixel (kills the cancellation in sumsq - sum^2/n)
    const st_f32x4 sh = st_ld4(x, xb + (int64_t)p0 * C4 + cg, xst);
    float4 s = make_float4(0, 0, 0, 0), q = make_float4(0, 0, 0, 0);
    auto acc = [&](st_f32x4 v) __attribute__((always_inline)) {
        const float dx = v.x - sh.x, dy = v.y - sh.y, dz = v.z - sh.z, dw = v.w - sh.w;
        s.x += dx; s.y += dy; s.z += dz; s.w += dw;
        q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
    };
    int p = p0 + pl;
    for (; p + (NU - 1) * PL < p1; p += NU * PL) {      // NU loads in flight, added in the order of the plain loop (same bits)
        st_f32x4 v[NU];
        st_ld4n<NU>(x, xb + (int64_t)p * C4 + cg, (int64_t)PL * C4, xst, v);
#pragma unroll
        for (int u = 0; u < NU; ++u) acc(v[u]);
    }
    for (; p < p1; p += PL) acc(st_ld4(x, xb + (int64_t)p * C4 + cg, xst));
    __shared__ float4 rs[256], rq[256];
    rs[threadIdx.x] = s; rq[threadIdx.x] = q;
    __syncthreads();
    if (pl == 0) {
        for (int i = 1; i < PL; ++i) {
            const float4 a = rs[i * C4 + cg], c = rq[i * C4 + cg];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            q.x += c.x; q.y += c.y; q.z += c.z; q.w += c.w;
        }
        const float n = (float)(p1 - p0), inv = 1.f / n;
        float2* o = part + ((size_t)(b * nchunks + ch) * C + cg * 4);
        o[0] = make_float2(sh.x + s.x * inv, q.x - s.x * s.x * inv);
        o[1] = make_float2(sh.y + s.y * inv, q.y - s.y * s.y * inv);
        o[2] = make_float2(sh.z + s.z * inv, q.z - s.z * s.z * inv);
        o[3] = make_float2(sh.w + s.w * inv, q.w - s.w * s.w * inv);
    }
}

// IN / AdaIN: combine chunk partials per (b,c); emit mean, rstd and the fused scale/shift.
// One workgroup = one sample x 16 channels; threads are (channel lane, chunk lane) = 16 x 16 and every
// thread keeps sixteen independent loads in flight (the kernel is pure load latency: 128 .. 256 chunks per image
// on the 64x64 maps); the 16 chunk lanes are merged through LDS.
__global__ void __launch_bounds__(256) norm_finalize_in_kernel(const float2* __restrict__ part, int C, int HW, int chunk, int nchunks,
                                                               const float* __restrict__ w, const float* __restrict__ bias, int w_stride,
                                                               float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                               float* __restrict__ scale, float* __restrict__ shift) {
    const int b = blockIdx.y, cl = threadIdx.x & 15, kl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    // eq: every chunk holds `chunk` pixels and the 16 chunk lanes see the same number of chunks (always so behind a convolution epilogue).  Then K
    // equal-count partials combine as mean = avg(mean_k), M2 = sum M2_k + count * sum (mean_k - mean)^2: no divisions and no serial dependence --
    // Chan's pairwise form costs two divisions per partial, 31 of them in a row per channel here (measured 8.4 us per launch against 4.7 us for
    // the backward's finalize, which only adds; 141 launches per step on the convolution chains).
    const bool eq = HW % chunk == 0 && nchunks % 16 == 0;
    if (c < C) {
        const float2* pb = part + (size_t)b * nchunks * C + c;
        if (eq) {
            for (int k0 = kl; k0 < nchunks; k0 += 256) {      // rounds of <= 16 partials held in registers (one round on the 64 x 64 maps)
                float2 v[16];
                int cnt = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int k = k0 + 16 * j;
                    v[j] = k < nchunks ? pb[(size_t)k * C] : make_float2(0.f, 0.f);
                    cnt += k < nchunks ? 1 : 0;
                }
                float sm_ = 0.f, sq = 0.f, sd2 = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) { sm_ += v[j].x; sq += v[j].y; }
                const float mr = sm_ / (float)cnt;
#pragma unroll
                for (int j = 0; j < 16; ++j) { const float d = v[j].x - mr; sd2 += (k0 + 16 * j < nchunks) ? d * d : 0.f; }
                chan_combine(n, mean, m2, (float)(cnt * chunk), mr, fmaf((float)chunk, sd2, sq));
            }
        } else {
            // U loads in flight per thread: 16 for >= 256 chunks, 4 otherwise (same order of combination either way)
            auto walk = [&](auto UC) __attribute__((always_inline)) {
                constexpr int U = decltype(UC)::value;
                for (int k0 = kl; k0 < nchunks; k0 += 16 * U) {
                    float2 v[U];
#pragma unroll
                    for (int j = 0; j < U; ++j) {
                        const int k = k0 + 16 * j;
                        v[j] = k < nchunks ? pb[(size_t)k * C] : make_float2(0.f, 0.f);
                    }
#pragma unroll
                    for (int j = 0; j < U; ++j) {
                        const int k = k0 + 16 * j;
                        if (k < nchunks) chan_combine(n, mean, m2, (float)(min(HW, (k + 1) * chunk) - k * chunk), v[j].x, v[j].y);
                    }
                }
            };
            if (nchunks >= 256) walk(std::integral_constant<int, 16>()); else walk(std::integral_constant<int, 4>());
        }
    }
    __shared__ float sn[16][16], sm[16][16], s2[16][16];
    sn[kl][cl] = n; sm[kl][cl] = mean; s2[kl][cl] = m2;
    __syncthreads();
    if (kl != 0 || c >= C) return;
    if (eq) {      // the 16 chunk lanes hold equal counts
        float sm_ = 0.f, sq = 0.f, sd2 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) { sm_ += sm[j][cl]; sq += s2[j][cl]; }
        mean = sm_ * (1.f / 16.f);
#pragma unroll
        for (int j = 0; j < 16; ++j) { const float d = sm[j][cl] - mean; sd2 += d * d; }
        m2 = fmaf(n, sd2, sq);
        n *= 16.f;
    } else {
        for (int j = 1; j < 16; ++j) chan_combine(n, mean, m2, sn[j][cl], sm[j][cl], s2[j][cl]);
    }
    const int i = b * C + c;
    const float rstd = rsqrtf(m2 / n + 1e-5f);
    mean_o[i] = mean; rstd_o[i] = rstd;
    const float ww = w ? w[(size_t)b * w_stride + c] : 1.f;
    const float bb = bias ? bias[(size_t)b * w_stride + c] : 0.f;
    scale[i] = rstd * ww;
    shift[i] = bb - mean * rstd * ww;
}

// LN statistics, stage 1 of 2 (round 6): workgroup (s, b) combines slice s of sample b's (chunk, channel) partials into one (count, mean, M2)
// triple, lnpart[b][s].  Stage 2 lives in norm_apply_kernel<.., true>: every workgroup combines the <= 64 triples of its sample itself.
// (Before: ONE workgroup per sample walked all partials -- 1 MB per sample behind the fused up-sampling convolutions, 64 rounds of load
// latency: 17 .. 32 us per layer on the decoder's critical path.)
constexpr int LN_SLICES = 64;
static int ln_slices(int items) { return std::max(1, std::min(LN_SLICES, items / 2048)); }
__global__ void __launch_bounds__(256) norm_finalize_ln_part_kernel(const float2* __restrict__ part, int C, int HW, int chunk, int nchunks,
                                                                    int per, float4* __restrict__ lnpart) {
    const int b = blockIdx.y, sl = blockIdx.x;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    const int items = nchunks * C;
    const int i_end = min(items, (sl + 1) * per);
    for (int i0 = sl * per + threadIdx.x; i0 < i_end; i0 += 256 * 8) {   // eight loads in flight per thread
        float2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = i0 + 256 * j;
            v[j] = i < i_end ? part[(size_t)b * items + i] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = i0 + 256 * j;
            if (i < i_end) {
                const int k = i / C;
                chan_combine(n, mean, m2, (float)(min(HW, (k + 1) * chunk) - k * chunk), v[j].x, v[j].y);
            }
        }
    }
    __shared__ float sn[256], sm[256], s2[256];
    sn[threadIdx.x] = n; sm[threadIdx.x] = mean; s2[threadIdx.x] = m2;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            float a = sn[threadIdx.x], bm = sm[threadIdx.x], c = s2[threadIdx.x];
            chan_combine(a, bm, c, sn[threadIdx.x + st], sm[threadIdx.x + st], s2[threadIdx.x + st]);
            sn[threadIdx.x] = a; sm[threadIdx.x] = bm; s2[threadIdx.x] = c;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) lnpart[b * gridDim.x + sl] = make_float4(sn[0], sm[0], s2[0], 0.f);
}
// stage 2, per wave: lane l holds triple l (or nothing), a butterfly of Chan combinations, then EVERY lane takes lane 0's result -- the
// combination is not commutative to the last bit, and all threads of all workgroups must apply the very same coefficients (norm_bwd recomputes
// the activation mask from the stored ones).  Unbiased std, eps on the std (networks.py:520-536).
__device__ __forceinline__ void ln_totals(const float4* __restrict__ lnpart, int b, int S, float& mu, float& t) {
    const int lane = threadIdx.x & 63;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    if (lane < S) { const float4 v = lnpart[b * S + lane]; n = v.x; mean = v.y; m2 = v.z; }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float nb = __shfl_xor(n, o), mb = __shfl_xor(mean, o), qb = __shfl_xor(m2, o);
        chan_combine(n, mean, m2, nb, mb, qb);
    }
    n = __shfl(n, 0); mean = __shfl(mean, 0); m2 = __shfl(m2, 0);
    const float sd = sqrtf(m2 / (n - 1.f));
    mu = mean; t = 1.f / (sd + 1e-5f);
}

// y = act(x*scale[b][c] + shift[b][c]) (+ residual).  Grid (gx, B): block row b is sample b, so no index is ever divided; 256 threads and
// the grid stride are multiples of C/4 (a power of two <= 256), so a thread's channel quad -- and its eight coefficients -- are fixed:
// loaded once.  NU float4 per operand in flight per thread.  (Round 6: the previous form -- one flat 64-bit index, i % C4 and
// i / (HW * C4) per float4, one float4 per thread -- spent more issue slots on the two divisions than on the element and ran at 3.5 TB/s.)
// SX | SY: the storage codes of x | of y and the residual, compiled in (the uniform triples and the fp32 <-> 16-bit boundary pairs: st_pair), or
// ST_MIXED = per-operand run-time codes -- there the switch is taken once per operand around its NU loads (st_ld4n).
// LN = true: scale / shift are OUTPUTS -- the workgroup derives them from the sample's LayerNorm triples (ln_totals) and gamma / beta, and
// workgroup 0 of the sample stores them with mean / rstd for the backward.
struct LnArgs { const float4* part; int S; const float* gamma; const float* beta; float* mean_o; float* rstd_o; };
template <int SX, int SY, bool LN>      // storage of x | of y and the residual (SX == ST_MIXED: run-time codes for all three)
__global__ void __launch_bounds__(256) norm_apply_kernel(const void* __restrict__ x, float* __restrict__ scale,
                                                         float* __restrict__ shift, const void* __restrict__ res,
                                                         void* __restrict__ y, NormST st, int per4, int C, int act, LnArgs ln) {
    const int C4 = C >> 2, b = blockIdx.y;
    const int cq = (threadIdx.x & (C4 - 1)) * 4, o = b * C + cq;
    float4 sc, sh;
    if (LN) {
        float mu, t;
        ln_totals(ln.part, b, ln.S, mu, t);
        const float4 g = *reinterpret_cast<const float4*>(ln.gamma + cq), be = *reinterpret_cast<const float4*>(ln.beta + cq);
        sc = make_float4(t * g.x, t * g.y, t * g.z, t * g.w);
        sh = make_float4(be.x - mu * t * g.x, be.y - mu * t * g.y, be.z - mu * t * g.z, be.w - mu * t * g.w);
        if (blockIdx.x == 0) {
            if (threadIdx.x == 0) { ln.mean_o[b] = mu; ln.rstd_o[b] = t; }
            if ((int)threadIdx.x < C4) { *reinterpret_cast<float4*>(scale + o) = sc; *reinterpret_cast<float4*>(shift + o) = sh; }
        }
    } else {
        sc = *reinterpret_cast<const float4*>(scale + o);
        sh = *reinterpret_cast<const float4*>(shift + o);
    }
    const int64_t base = (int64_t)b * per4;
    const int stride = gridDim.x * 256;
    constexpr bool RT = SX == ST_MIXED;
    const int cx = RT ? st.x : SX, cy = RT ? st.y : SY, cr = RT ? st.res : SY;
    auto one = [&](st_f32x4 v) __attribute__((always_inline)) {
        st_f32x4 r;
        r.x = act_fwd(fmaf(v.x, sc.x, sh.x), act); r.y = act_fwd(fmaf(v.y, sc.y, sh.y), act);
        r.z = act_fwd(fmaf(v.z, sc.z, sh.z), act); r.w = act_fwd(fmaf(v.w, sc.w, sh.w), act);
        return r;
    };
    auto put = [&](int64_t i, st_f32x4 v) __attribute__((always_inline)) { st_st4(y, i, v, cy); };
    int i = blockIdx.x * 256 + threadIdx.x;
    for (; i + (NU - 1) * stride < per4; i += NU * stride) {
        st_f32x4 v[NU], r[NU];
        st_ld4n<NU>(x, base + i, stride, cx, v);
        if (res) st_ld4n<NU>(res, base + i, stride, cr, r);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            st_f32x4 q = one(v[u]);
            if (res) q += r[u];
            put(base + i + u * stride, q);
        }
    }
    for (; i < per4; i += stride) {
        st_f32x4 q = one(st_ld4(x, base + i, cx));
        if (res) q += st_ld4(res, base + i, cr);
        put(base + i, q);
    }
}
// workgroups per sample of the (gx, B) elementwise grids: NU float4 per thread, at most ~8192 workgroups in all
static int rows_grid(int per4, int B) {
    const int want = cdiv(per4, 256 * NU), cap = std::max(1, 8192 / std::max(1, B));
    return std::max(1, std::min(want, cap));
}

static inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

size_t norm_scratch_bytes(int B, int HW, int C) {
    const int chunk = norm_chunk_pixels(B, HW);
    const int nchunks = cdiv(HW, chunk);
    // chunk partials | scale, shift (forward) or A, Bc, Cc (+ LN totals) (backward) | the LayerNorm slice triples (forward) | slack
    return (size_t)B * nchunks * C * sizeof(float2) + (size_t)B * C * 5 * sizeof(float) + (size_t)B * LN_SLICES * sizeof(float4) + 256;
}

// stats != nullptr: the producer of x already wrote the chunk partials [B][HW / stats_chunk][C] (mean, M2) -- conv_fwd's epilogue
// (conv_fwd_stats_chunk) -- and the statistics pass over x is skipped
int norm_fwd(int kind, int act, int B, int HW, int C, const void* x, const float* w, const float* b, int w_stride,
             const void* residual, void* y, float* mean, float* rstd, void* scratch, hipStream_t st, const float* stats, int stats_chunk,
             const NormST* sto, float* ss_out) {
    const NormST sd = sto ? *sto : NormST();
    ACL_REQUIRE(pow2(C) && C >= 4 && C <= 1024, "norm: C=%d must be a power of two in [4,1024]", C);
    ACL_REQUIRE(kind == ACLGAN_NORM_IN || kind == ACLGAN_NORM_ADAIN || kind == ACLGAN_NORM_LN, "norm: bad kind %d", kind);
    ACL_REQUIRE(!(residual && act != ACLGAN_ACT_NONE), "norm: residual requires act none");
    ACL_REQUIRE(!stats || (stats_chunk > 0 && HW % stats_chunk == 0), "norm: bad statistics chunk %d for %d pixels", stats_chunk, HW);
    const int own_chunk = norm_chunk_pixels(B, HW);
    const int chunk = stats ? stats_chunk : own_chunk, nchunks = cdiv(HW, chunk);
    float2* part = stats ? (float2*)stats : (float2*)scratch;
    // the fused coefficients y = act(x * scale + shift): in the scratch, or -- ss_out, 2 B C floats -- in a buffer the caller keeps for
    // norm_bwd, which then recomputes the activation mask from x with the very same fmaf instead of reading y back
    float* scale = ss_out ? ss_out : (float*)((float2*)scratch + (size_t)B * cdiv(HW, own_chunk) * C);
    float* shift = scale + (size_t)B * C;
    if (!stats) {
        if (sd.x == ST_F32) hipLaunchKernelGGL(norm_stats_kernel<ST_F32>, dim3(nchunks, B), dim3(256), 0, st, x, sd.x, part, HW, C, chunk, nchunks);
        else if (sd.x == ST_BF16) hipLaunchKernelGGL(norm_stats_kernel<ST_BF16>, dim3(nchunks, B), dim3(256), 0, st, x, sd.x, part, HW, C, chunk, nchunks);
        else hipLaunchKernelGGL(norm_stats_kernel<ST_F16>, dim3(nchunks, B), dim3(256), 0, st, x, sd.x, part, HW, C, chunk, nchunks);
        ACL_CHECK_LAUNCH("norm_stats_kernel");
    }
    LnArgs ln = {nullptr, 0, nullptr, nullptr, nullptr, nullptr};
    if (kind == ACLGAN_NORM_LN) {
        ACL_REQUIRE(w && b, "LN needs gamma/beta");
        const int items = nchunks * C, S = ln_slices(items), per = cdiv(items, S);
        float4* lnpart = (float4*)((char*)scratch + norm_scratch_bytes(B, HW, C) - 256 - (size_t)B * LN_SLICES * sizeof(float4));
        hipLaunchKernelGGL(norm_finalize_ln_part_kernel, dim3(S, B), dim3(256), 0, st, part, C, HW, chunk, nchunks, per, lnpart);
        ln = LnArgs{lnpart, S, w, b, mean, rstd};
    } else {
        const float* ww = kind == ACLGAN_NORM_ADAIN ? w : nullptr;
        const float* bb = kind == ACLGAN_NORM_ADAIN ? b : nullptr;
        ACL_REQUIRE(kind != ACLGAN_NORM_ADAIN || (w && b), "AdaIN needs weight/bias");
        hipLaunchKernelGGL(norm_finalize_in_kernel, dim3(cdiv(C, 16), B), dim3(256), 0, st, part, C, HW, chunk, nchunks,
                           ww, bb, w_stride, mean, rstd, scale, shift);
    }
    ACL_CHECK_LAUNCH("norm_finalize");
    ACL_REQUIRE((int64_t)HW * (C / 4) < 0x7fffff00ll, "norm: %d pixels x %d channels per sample", HW, C);
    const int per4 = HW * (C / 4);
    const dim3 grid(rows_grid(per4, B), B);
    // storage pair (x | y, residual): the uniform triples and the fp32 <-> 16-bit boundary pairs are compiled in, anything else takes run-time codes
    const int pair = (!residual || sd.res == sd.y) ? st_pair(sd.x, sd.y) : -1;
#define ACL_NA(SX, SY) do { if (kind == ACLGAN_NORM_LN) hipLaunchKernelGGL((norm_apply_kernel<SX, SY, true>), grid, dim3(256), 0, st, x, scale, shift, residual, y, sd, per4, C, act, ln); \
                            else hipLaunchKernelGGL((norm_apply_kernel<SX, SY, false>), grid, dim3(256), 0, st, x, scale, shift, residual, y, sd, per4, C, act, ln); } while (0)
    ACL_ST_PAIR_DISPATCH(pair, ACL_NA);
#undef ACL_NA
    ACL_CHECK_LAUNCH("norm_apply_kernel");
    return ACLGAN_OK;
}

// ------------------------------------------------------------------------------------------
// normalisation backward
//   g = dy * act'(y);  xhat = (x - mean) * rstd
//   reduce:   s1[b][c] = sum_hw g,  s2[b][c] = sum_hw g*xhat          (chunk partials + combine)
//   IN/AdaIN: dx = rstd*w*(g - s1/n - xhat*s2/n);  dw[b][c] += s2;  db[b][c] += s1
//   LN:       dxhat = g*gamma_c; S1_b = sum_c gamma_c s1, S2_b = sum_c gamma_c s2, n = C*HW,
//             t = 1/(std+eps), std = 1/t - eps:
//             dx = t*(dxhat - S1_b/n) - xhat*S2_b/((n-1)*std);  dgamma_c += sum_b s2;  dbeta_c += sum_b s1
//   both written as dx = A[b][c]*g + Bc[b][c]*xhat + Cc[b][c]
// ------------------------------------------------------------------------------------------
template <int SX, int SG>      // storage of x and y | of dy (SX == ST_MIXED: run-time codes)
__global__ void __launch_bounds__(256) norm_bwd_reduce_kernel(const void* __restrict__ x, const void* __restrict__ y,
                                                              const void* __restrict__ dy, NormST st, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, int per_channel_stats,
                                                              float2* __restrict__ part, int HW, int C, int chunk,
                                                              int nchunks, int act, const float* __restrict__ msc, const float* __restrict__ msh) {
    const int C4 = C >> 2;
    const int b = blockIdx.y, ch = blockIdx.x;
    const int p0 = ch * chunk, p1 = min(HW, p0 + chunk);
    const int cg = threadIdx.x % C4, pl = threadIdx.x / C4, PL = 256 / C4;
    const int64_t base = (int64_t)b * HW * C4;
    float4 mu, rs;
    if (per_channel_stats) {
        mu = *reinterpret_cast<const float4*>(mean + b * C + cg * 4);
        rs = *reinterpret_cast<const float4*>(rstd + b * C + cg * 4);
    } else {
        const float m = mean[b], r = rstd[b];
        mu = make_float4(m, m, m, m); rs = make_float4(r, r, r, r);
    }
    float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
    float4 ksc = make_float4(0, 0, 0, 0), ksh = ksc;
    if (msc) { ksc = *reinterpret_cast<const float4*>(msc + b * C + cg * 4); ksh = *reinterpret_cast<const float4*>(msh + b * C + cg * 4); }
    const bool need_y = act != ACLGAN_ACT_NONE && !msc;
    auto acc = [&](st_f32x4 xv, st_f32x4 gv, st_f32x4 yv) __attribute__((always_inline)) {
        // act-less norms (2nd norm of every ResBlock): y is not needed.  ReLU / LeakyReLU with the forward's coefficients at hand (msc): the mask
        // is the sign of the forward's own fmaf(x, scale, shift) -- recomputed, the read of y is saved
        if (msc && act != ACLGAN_ACT_NONE) { yv.x = fmaf(xv.x, ksc.x, ksh.x); yv.y = fmaf(xv.y, ksc.y, ksh.y); yv.z = fmaf(xv.z, ksc.z, ksh.z); yv.w = fmaf(xv.w, ksc.w, ksh.w); }
        const float g0 = gv.x * act_grad(yv.x, act), g1 = gv.y * act_grad(yv.y, act);
        const float g2 = gv.z * act_grad(yv.z, act), g3 = gv.w * act_grad(yv.w, act);
        s1.x += g0; s1.y += g1; s1.z += g2; s1.w += g3;
        s2.x += g0 * (xv.x - mu.x) * rs.x; s2.y += g1 * (xv.y - mu.y) * rs.y;
        s2.z += g2 * (xv.z - mu.z) * rs.z; s2.w += g3 * (xv.w - mu.w) * rs.w;
    };
    const st_f32x4 ones = {1.f, 1.f, 1.f, 1.f};
    constexpr bool RT = SX == ST_MIXED;
    const int cx = RT ? st.x : SX, cg_ = RT ? st.dy : SG, cy = RT ? st.y : SX;
    int p = p0 + pl;
    for (; p + (NU - 1) * PL < p1; p += NU * PL) {      // NU pixels' loads in flight, summed in the order of the plain loop (same bits)
        st_f32x4 xv[NU], gv[NU], yv[NU];
        const int64_t i0 = base + (int64_t)p * C4 + cg, istep = (int64_t)PL * C4;
        st_ld4n<NU>(x, i0, istep, cx, xv);
        st_ld4n<NU>(dy, i0, istep, cg_, gv);
        if (need_y) st_ld4n<NU>(y, i0, istep, cy, yv);
        else {
#pragma unroll
            for (int u = 0; u < NU; ++u) yv[u] = ones;
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) acc(xv[u], gv[u], yv[u]);
    }
    for (; p < p1; p += PL) {
        const int64_t i = base + (int64_t)p * C4 + cg;
        acc(st_ld4(x, i, cx), st_ld4(dy, i, cg_), need_y ? st_ld4(y, i, cy) : ones);
    }
    __shared__ float4 r1[256], r2[256];
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    if (pl == 0) {
        for (int i = 1; i < PL; ++i) {
            const float4 a = r1[i * C4 + cg], c = r2[i * C4 + cg];
            s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
            s2.x += c.x; s2.y += c.y; s2.z += c.z; s2.w += c.w;
        }
        float2* o = part + ((size_t)(b * nchunks + ch) * C + cg * 4);
        o[0] = make_float2(s1.x, s2.x); o[1] = make_float2(s1.y, s2.y);
        o[2] = make_float2(s1.z, s2.z); o[3] = make_float2(s1.w, s2.w);
    }
}

// (channel lane, chunk lane) = 16 x 16 threads, four loads in flight each (see norm_finalize_in_kernel)
__global__ void __launch_bounds__(256) norm_bwd_finalize_in_kernel(const float2* __restrict__ part, int C, int HW, int nchunks,
                                                                   const float* __restrict__ w, int w_stride, const float* __restrict__ rstd,
                                                                   float* __restrict__ cA, float* __restrict__ cB, float* __restrict__ cC,
                                                                   float* __restrict__ dw, float* __restrict__ db) {
    const int b = blockIdx.y, cl = threadIdx.x & 15, kl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float s1 = 0.f, s2 = 0.f;
    if (c < C) {
        const float2* pb = part + (size_t)b * nchunks * C + c;
        for (int k0 = kl; k0 < nchunks; k0 += 128) {
            float2 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + 16 * j;
                v[j] = k < nchunks ? pb[(size_t)k * C] : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { s1 += v[j].x; s2 += v[j].y; }
        }
    }
    __shared__ float r1[16][16], r2[16][16];
    r1[kl][cl] = s1; r2[kl][cl] = s2;
    __syncthreads();
    if (kl != 0 || c >= C) return;
    for (int j = 1; j < 16; ++j) { s1 += r1[j][cl]; s2 += r2[j][cl]; }
    const int i = b * C + c;
    const float ww = w ? w[(size_t)b * w_stride + c] : 1.f;
    const float a = rstd[i] * ww, inv = 1.f / (float)HW;
    cA[i] = a; cB[i] = -a * s2 * inv; cC[i] = -a * s1 * inv;
    if (dw) dw[(size_t)b * w_stride + c] += s2;
    if (db) db[(size_t)b * w_stride + c] += s1;
}

// dgamma[c] += sum_b s2[b][c], dbeta[c] += sum_b s1[b][c] from sbc [B][C][2]: fixed order, reproducible bit for bit
__global__ void __launch_bounds__(256) norm_bwd_ln_params_kernel(const float* __restrict__ sbc, int B, int C, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s1 = 0.f, s2 = 0.f;
    for (int b = 0; b < B; ++b) { s1 += sbc[2 * (b * C + c)]; s2 += sbc[2 * (b * C + c) + 1]; }
    if (dbeta) dbeta[c] += s1;
    if (dgamma) dgamma[c] += s2;
}

// LN: one workgroup per sample: per-(b,c) totals of the chunk partials, the per-sample sums S1_b, S2_b,
// the coefficients.  The parameter gradients are summed over the samples, in order, by norm_bwd_ln_params_kernel.
__global__ void __launch_bounds__(256) norm_bwd_finalize_ln_kernel(const float2* __restrict__ part, int B, int C, int HW,
                                                                   int nchunks, const float* __restrict__ gamma,
                                                                   const float* __restrict__ rstd, float* __restrict__ cA,
                                                                   float* __restrict__ cB, float* __restrict__ cC,
                                                                   float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                   float* __restrict__ sbc /* [B][C][2] */) {
    __shared__ float red[4];
    __shared__ float r1[256], r2[256];
    const int b = blockIdx.x;
    float a1 = 0.f, a2 = 0.f;
    // threads = (channel lane, chunk lane): CL = min(C, 256) channels wide, 256 / CL chunk lanes deep
    const int CL = C < 256 ? C : 256, KL = 256 / CL;
    const int cl = threadIdx.x % CL, kl = threadIdx.x / CL;
    for (int c0 = 0; c0 < C; c0 += CL) {
        const int c = c0 + cl;
        const float2* pb = part + (size_t)b * nchunks * C + c;
        float s1 = 0.f, s2 = 0.f;
        for (int k0 = kl; k0 < nchunks; k0 += 8 * KL) {   // eight loads in flight per thread
            float2 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j * KL;
                v[j] = k < nchunks ? pb[(size_t)k * C] : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { s1 += v[j].x; s2 += v[j].y; }
        }
        __syncthreads();
        r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
        __syncthreads();
        if (kl == 0) {
            for (int j = 1; j < KL; ++j) { s1 += r1[j * CL + cl]; s2 += r2[j * CL + cl]; }
            sbc[2 * (b * C + c)] = s1; sbc[2 * (b * C + c) + 1] = s2;
            const float g = gamma[c];      // (dgamma / dbeta: norm_bwd_ln_params_kernel adds the samples in order)
            a1 += g * s1; a2 += g * s2;
        }
    }
    const float S1 = block_sum256(a1, red);
    const float S2 = block_sum256(a2, red);
    const float n = (float)C * (float)HW;
    const float t = rstd[b];
    const float sd = 1.f / t - 1e-5f;
    const float kb = -S2 / ((n - 1.f) * sd), kc = -t * S1 / n;
    for (int c = threadIdx.x; c < C; c += 256) {
        cA[b * C + c] = t * gamma[c]; cB[b * C + c] = kb; cC[b * C + c] = kc;
    }
}

// dx = A g + Bc xhat + Cc, g = dy act'(y) (+ the residual branch's gradient: dres (+)= g).  Same (gx, B) grid as norm_apply_kernel: the thread's
// channel quad is fixed, its coefficients (mean, rstd, A, Bc, Cc, the forward's scale / shift) are loaded once, NU float4 per operand in flight
// (SX: storage of x, y, dx and the residual branch's gradient | SG: of dy; SX == ST_MIXED: run-time codes).
template <int SX, int SG>
__global__ void __launch_bounds__(256) norm_bwd_apply_kernel(const void* __restrict__ x, const void* __restrict__ y,
                                                             const void* __restrict__ dy, NormST st, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, int per_channel_stats,
                                                             const float* __restrict__ cA, const float* __restrict__ cB,
                                                             const float* __restrict__ cC, void* __restrict__ dx,
                                                             void* __restrict__ dres, int dres_acc, int per4, int C, int act,
                                                             const float* __restrict__ msc, const float* __restrict__ msh) {
    const int C4 = C >> 2, b = blockIdx.y;
    const int o = b * C + (threadIdx.x & (C4 - 1)) * 4;
    float4 mu, rs;
    if (per_channel_stats) {
        mu = *reinterpret_cast<const float4*>(mean + o); rs = *reinterpret_cast<const float4*>(rstd + o);
    } else {
        const float m = mean[b], r = rstd[b];
        mu = make_float4(m, m, m, m); rs = make_float4(r, r, r, r);
    }
    const float4 a = *reinterpret_cast<const float4*>(cA + o);
    const float4 kb = *reinterpret_cast<const float4*>(cB + o);
    const float4 kc = *reinterpret_cast<const float4*>(cC + o);
    float4 ksc = make_float4(0, 0, 0, 0), ksh = ksc;
    const bool need_y = act != ACLGAN_ACT_NONE, from_x = need_y && msc != nullptr;
    if (from_x) { ksc = *reinterpret_cast<const float4*>(msc + o); ksh = *reinterpret_cast<const float4*>(msh + o); }
    const int64_t base = (int64_t)b * per4;
    const int stride = gridDim.x * 256;
    constexpr bool RT = SX == ST_MIXED;
    const int cx = RT ? st.x : SX, cg_ = RT ? st.dy : SG, cy = RT ? st.y : SX, cdx = RT ? st.dx : SX, cdr = RT ? st.dres : SX;
    // g and dx of one float4 (yv: the activation's output, or -- from_x -- recomputed as the forward's own fmaf)
    auto one = [&](st_f32x4 xv, st_f32x4 gv, st_f32x4 yv, st_f32x4& g) __attribute__((always_inline)) {
        if (from_x) { yv.x = fmaf(xv.x, ksc.x, ksh.x); yv.y = fmaf(xv.y, ksc.y, ksh.y); yv.z = fmaf(xv.z, ksc.z, ksh.z); yv.w = fmaf(xv.w, ksc.w, ksh.w); }
        g.x = gv.x * act_grad(yv.x, act); g.y = gv.y * act_grad(yv.y, act);
        g.z = gv.z * act_grad(yv.z, act); g.w = gv.w * act_grad(yv.w, act);
        st_f32x4 d;
        d.x = fmaf(a.x, g.x, fmaf(kb.x, (xv.x - mu.x) * rs.x, kc.x));
        d.y = fmaf(a.y, g.y, fmaf(kb.y, (xv.y - mu.y) * rs.y, kc.y));
        d.z = fmaf(a.z, g.z, fmaf(kb.z, (xv.z - mu.z) * rs.z, kc.z));
        d.w = fmaf(a.w, g.w, fmaf(kb.w, (xv.w - mu.w) * rs.w, kc.w));
        return d;
    };
    const st_f32x4 ones = {1.f, 1.f, 1.f, 1.f};
    int i = blockIdx.x * 256 + threadIdx.x;
    for (; i + (NU - 1) * stride < per4; i += NU * stride) {
        st_f32x4 xv[NU], gv[NU], yv[NU], rv[NU];
        st_ld4n<NU>(x, base + i, stride, cx, xv);
        st_ld4n<NU>(dy, base + i, stride, cg_, gv);
        if (need_y && !from_x) st_ld4n<NU>(y, base + i, stride, cy, yv);
        else {
#pragma unroll
            for (int u = 0; u < NU; ++u) yv[u] = ones;
        }
        if (dres && dres_acc) st_ld4n<NU>(dres, base + i, stride, cdr, rv);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            st_f32x4 g;
            const st_f32x4 d = one(xv[u], gv[u], yv[u], g);
            st_st4(dx, base + i + u * stride, d, cdx);
            if (dres) {
                if (dres_acc) g += rv[u];
                st_st4(dres, base + i + u * stride, g, cdr);
            }
        }
    }
    for (; i < per4; i += stride) {
        const st_f32x4 xv = st_ld4(x, base + i, cx), gv = st_ld4(dy, base + i, cg_);
        const st_f32x4 yv = (need_y && !from_x) ? st_ld4(y, base + i, cy) : ones;
        st_f32x4 g;
        const st_f32x4 d = one(xv, gv, yv, g);
        st_st4(dx, base + i, d, cdx);
        if (dres) {
            if (dres_acc) g += st_ld4(dres, base + i, cdr);
            st_st4(dres, base + i, g, cdr);
        }
    }
}

int norm_bwd(int kind, int act, int B, int HW, int C, const void* x, const void* y, const void* dy, const float* w,
             int w_stride, const float* mean, const float* rstd, void* dx, float* dw, float* db, void* dres,
             int dres_accumulate, void* scratch, hipStream_t st, const NormST* sto, float* sbc_out, const float* ss) {
    const NormST sd = sto ? *sto : NormST();
    // ss (optional): norm_fwd's ss_out of the same layer -- only a sign-based activation can be recovered from the pre-activation
    const float* msc = (ss && (act == ACLGAN_ACT_RELU || act == ACLGAN_ACT_LRELU)) ? ss : nullptr;
    const float* msh = msc ? msc + (size_t)B * C : nullptr;
    ACL_REQUIRE(pow2(C) && C >= 4 && C <= 1024, "norm: C=%d must be a power of two in [4,1024]", C);
    const int chunk = norm_chunk_pixels(B, HW), nchunks = cdiv(HW, chunk);
    float2* part = (float2*)scratch;
    float* cA = (float*)(part + (size_t)B * nchunks * C);
    float* cB = cA + (size_t)B * C;
    float* cC = cB + (size_t)B * C;
    const int pcs = kind != ACLGAN_NORM_LN;
    {
        const bool ry = act != ACLGAN_ACT_NONE && !msc;
        const int pair = (!ry || sd.y == sd.x) ? st_pair(sd.x, sd.dy) : -1;
#define ACL_NBR(SX, SG) hipLaunchKernelGGL((norm_bwd_reduce_kernel<SX, SG>), dim3(nchunks, B), dim3(256), 0, st, x, y, dy, sd, mean, rstd, pcs, part, HW, C, chunk, \
                                           nchunks, act, msc, msh)
        ACL_ST_PAIR_DISPATCH(pair, ACL_NBR);
#undef ACL_NBR
    }
    ACL_CHECK_LAUNCH("norm_bwd_reduce_kernel");
    if (kind == ACLGAN_NORM_LN) {
        ACL_REQUIRE(w, "LN backward needs gamma");
        float* sbc = sbc_out ? sbc_out : cC + (size_t)B * C;   // [B][C][2] totals (scratch tail, see norm_scratch_bytes; or the caller's buffer)
        hipLaunchKernelGGL(norm_bwd_finalize_ln_kernel, dim3(B), dim3(256), 0, st, part, B, C, HW, nchunks, w, rstd, cA, cB, cC, dw, db, sbc);
        ACL_CHECK_LAUNCH("norm_bwd_finalize_ln_kernel");
        if ((dw || db) && !sbc_out) {
            hipLaunchKernelGGL(norm_bwd_ln_params_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, sbc, B, C, dw, db);
            ACL_CHECK_LAUNCH("norm_bwd_ln_params_kernel");
        }
    } else {
        const float* ww = kind == ACLGAN_NORM_ADAIN ? w : nullptr;
    hipLaunchKernelGGL(norm_bwd_finalize_in_kernel, dim3(cdiv(C, 16), B), dim3(256), 0, st, part, C, HW, nchunks, ww, w_stride,
                       rstd, cA, cB, cC, kind == ACLGAN_NORM_ADAIN ? dw : nullptr, kind == ACLGAN_NORM_ADAIN ? db : nullptr);
    ACL_CHECK_LAUNCH("norm_bwd_finalize_in_kernel");
    }
    ACL_REQUIRE((int64_t)HW * (C / 4) < 0x7fffff00ll, "norm: %d pixels x %d channels per sample", HW, C);
    const int per4 = HW * (C / 4);
    const dim3 grid(rows_grid(per4, B), B);
    const bool need_y = act != ACLGAN_ACT_NONE && !msc;
    const int pair = (sd.dx == sd.x && (!need_y || sd.y == sd.x) && (!dres || sd.dres == sd.x)) ? st_pair(sd.x, sd.dy) : -1;
#define ACL_NBA(SX, SG) hipLaunchKernelGGL((norm_bwd_apply_kernel<SX, SG>), grid, dim3(256), 0, st, x, y, dy, sd, mean, rstd, pcs, cA, cB, cC, dx, dres, \
                                           dres_accumulate, per4, C, act, msc, msh)
    ACL_ST_PAIR_DISPATCH(pair, ACL_NBA);
#undef ACL_NBA
    ACL_CHECK_LAUNCH("norm_bwd_apply_kernel");
    return ACLGAN_OK;
}

int norm_bwd_ln_params(const float* sbc, int B, int C, float* dgamma, float* dbeta, hipStream_t st) {
    if (!dgamma && !dbeta) return ACLGAN_OK;
    hipLaunchKernelGGL(norm_bwd_ln_params_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, sbc, B, C, dgamma, dbeta);
    ACL_CHECK_LAUNCH("norm_bwd_ln_params_kernel");
    return ACLGAN_OK;
}

}  // namespace aclgan
