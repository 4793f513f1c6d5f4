// conv_fast16.hip -- the heavy convolutions on the 16-bit matrix cores of gfx950:
// v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x16_f16 (2.5 PFLOP/s dense, 16x the fp32 MFMA rate), fp32 accumulation.
// BASELINE.json configs[2] (bf16) and configs[4] (fp16 + loss scaling).  The reference is fp32-only (no AMP anywhere,
// SURVEY.md section 0); this is the reduced-precision variant of the same three GEMMs conv_fast.hip runs in fp32:
//   forward  networks.py:366 (+256 upsample)   dgrad / wgrad  autograd of the same line (trainer.py:169,292)
//
// Precision contract ("compute dtype"): the two MFMA operands are rounded to bf16 / fp16 (round-to-nearest-even,
// v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32) on their way into LDS; products are exact in the fp32 accumulator; bias,
// activation, split-K partials, the outputs (y, dx, dw, db) and everything outside the convolutions stay fp32.
// Weights come pre-rounded from the per-step weight pack (pack_weights16: the fp32 master copy is what Adam updates).
//
// Layout / pipeline (same skeleton as conv_fast.hip, re-cut for the 16x faster MFMA):
//   * k-tile depth 32: one LDS row = 32 halfs + 8 pad = 80 bytes -- byte-identical geometry to the fp32 kernels'
//     [row][16+4] floats, so b128 row accesses stay conflict-free; a lane's MFMA fragment (8 consecutive k of its row)
//     is ONE ds_read_b128; two MFMA k-steps per tile;
//   * activations are read as fp32 (2 x global_load_dwordx4 per 8-k chunk), converted in registers (4 v_cvt_pk) and
//     stored to LDS as one b128; weights are read as packed 16-bit (1 x dwordx4 per chunk);
//   * dgrad reads the weights from the TRANSPOSED pack [tap][cin][cout]: its GEMM k axis (cout) must be contiguous
//     per B row, which a k-major LDS tile of 16-bit values cannot provide without 2-byte scattered LDS traffic;
//   * wgrad's k axis is the pixel index, which is the SLOW axis of NHWC for both operands: each staging thread loads
//     8 pixels x 4 channels (8 coalesced dwordx4 loads), transposes in registers and writes 4 b128 rows [channel][8 px];
//     rows are stored interleaved (row = (ch%4)*(rows/4) + ch/4) so that the 32 lanes of a write hit consecutive rows.
#include "conv_fast_common.h"
#include "st16.h"

namespace aclgan {
namespace {

typedef unsigned short u16;
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // 8 packed 16-bit values: the LDS / global unit
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int WG16_TARGET = 512; // wgrad workgroups in flight the split plan aims for: one resident round at 2 per CU
constexpr int BK16 = 32;        // k values per tile of the wgrad kernel (KS = 2) and the unit of the split-K plans
constexpr int LDQ = 5;          // KS = 2: LDS row stride in u32x4 units: 4 chunks of 8 halfs + 1 pad = 80 bytes
// forward / dgrad are templated on KS = k-steps (of 16) per tile: KS = 4 (64-deep tiles, row stride 9 x 16 = 144 bytes,
// b128 rows still conflict-free: 36 dwords = 4 banks further per row) when the GEMM k axis allows it.  Why: at the 16-bit
// MFMA rate a 32-deep tile is only 8 MFMAs = 256 cycles per wave, far less than one global-load latency, and the
// write-after-barrier pipeline gives a load exactly one iteration to land -- the 32-deep loop is LATENCY-bound
// (measured: 24 % MFMA utilisation, insensitive to the bytes per tile).  64-deep tiles double the bytes in flight and
// halve the barriers per MFMA.

struct PBF16 {
    typedef bf16x8 vec;
    static __device__ __forceinline__ u32x4 pack(f32x8 v) { return __builtin_bit_cast(u32x4, __builtin_convertvector(v, bf16x8)); }
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
struct PFP16 {
    typedef f16x8 vec;
    static __device__ __forceinline__ u32x4 pack(f32x8 v) { return __builtin_bit_cast(u32x4, __builtin_convertvector(v, f16x8)); }
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

template <class T>
__device__ __forceinline__ u32x4 pack8(f32x4 lo, f32x4 hi) {
    return T::pack(__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// Fragment of 32-row tile t, k-step ks, lane (l31, kh): chunk ks*2+kh of row base+t*32+l31 (row stride 2*KS+1 chunks).
template <int KS, int TN_>
__device__ __forceinline__ void read_frags16(const u32x4* __restrict__ tile, int base, int lane, u32x4 (&f)[TN_][KS]) {
    const int l31 = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int t = 0; t < TN_; ++t) {
        const u32x4* p = tile + (base + t * 32 + l31) * (2 * KS + 1) + kh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) f[t][ks] = p[2 * ks];
    }
}
template <class T, int KS, int TM, int TN>
__device__ __forceinline__ void mfma_step16(const u32x4 (&fa)[TM][KS], const u32x4 (&fb)[TN][KS], int ks, f32x16 (&acc)[TM][TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = T::mfma(fa[i][ks], fb[j][ks], acc[i][j]);
}

// Same write-after-barrier pipeline as ACL_GEMM_MAINLOOP (conv_fast.hip): `fetch(t)` issues the global loads of tile t
// into the staging registers, `stage(buf, real)` converts them and writes LDS buffer `buf`.  Both unconditional.
#define ACL_GEMM16_MAINLOOP(T_, KS_, TM_, TN_, KBEG_, NK_, AS_, BS_, ASTR_, BSTR_, AM_, BN_)             \
    do {                                                                                                 \
        const int nk__ = (NK_), kb__ = (KBEG_);                                                          \
        fetch(kb__);                                                                                     \
        stage(0, true);                                                                                  \
        fetch(kb__ + min(1, nk__ - 1));                                                                  \
        __syncthreads();                                                                                 \
        for (int kt = 0; kt < nk__; ++kt) {                                                              \
            const int cur = kt & 1;                                                                      \
            u32x4 fa[TM_][KS_], fb[TN_][KS_];                                                            \
            read_frags16<KS_, TM_>((AS_) + cur * (ASTR_), (AM_), lane, fa);                              \
            read_frags16<KS_, TN_>((BS_) + cur * (BSTR_), (BN_), lane, fb);                              \
            mfma_step16<T_, KS_, TM_, TN_>(fa, fb, 0, acc);                                              \
            stage(cur ^ 1, kt + 1 < nk__);                                                               \
            fetch(kb__ + min(kt + 2, nk__ - 1));                                                         \
            __builtin_amdgcn_sched_barrier(0x6);   /* only ALU may cross: the loads of tile kt+2 issue HERE */ \
            _Pragma("unroll") for (int ks__ = 1; ks__ < (KS_); ++ks__) mfma_step16<T_, KS_, TM_, TN_>(fa, fb, ks__, acc); \
            __syncthreads();                                                                             \
        }                                                                                                \
    } while (0)

// ------------------------------------------------------------------------------------------
// forward (Cin % 32 == 0)
// ------------------------------------------------------------------------------------------
// A16: the activations come from the producer's 16-bit copy (p.x16) -- one dwordx4 per 8-k chunk, no conversion -- instead of
// being read as fp32 and rounded here.  Same values either way (the producer rounds with the same instruction).
template <class T, int KS, int WM, int WN, int TM, int TN, bool A16>
__global__ void __launch_bounds__(WM * WN * 64) conv_fwd16_kernel(FwdFP p) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
    constexpr int NCH = 2 * KS, LDQ = NCH + 1, BKT = 16 * KS;   // chunks (of 8 k) per row, row stride, k values per tile
    constexpr int RP = NT / NCH;               // rows covered per pass
    constexpr int A_IT = BM / RP, B_IT = (BN + RP - 1) / RP;
    __shared__ u32x4 smem[2 * (BM + BN) * LDQ];
    __shared__ int ro[BM];                     // output pixel index of each tile row (-1: not stored)
    u32x4* As = smem;
    u32x4* Bs = smem + 2 * BM * LDQ;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile = xcd_map(blockIdx.x, p.nwg);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const int q = tid % NCH, r0 = tid / NCH;
    const int phase = p.phases ? (int)blockIdx.z : 0;
    const u16* wbase = p.w16 + (size_t)phase * p.Co * p.K;

    for (int r = tid; r < BM; r += NT) {
        int b, oy, ox, o = -1;
        if (fwd_row(p, m0 + r, b, oy, ox)) {
            if (p.phases) o = (b * p.Hf + 2 * (oy + 1) + (phase >> 1)) * p.Wf + 2 * (ox + 1) + (phase & 1);
            else o = (b * p.Ho + oy) * p.Wo + ox;
        }
        ro[r] = o;
    }
    int ay[A_IT], ax[A_IT], ab[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        int b = 0, oy = 0, ox = 0;
        if (!fwd_row(p, m0 + r0 + i * RP, b, oy, ox)) { b = 0; oy = 0; ox = 0; }   // past the end: any valid row (never stored)
        ay[i] = oy * p.s - p.p; ax[i] = ox * p.s - p.p; ab[i] = b * p.Hi * p.Wi;
    }
    int wo[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) wo[i] = min(n0 + r0 + i * RP, p.Co - 1) * p.K + q * 8;

    const int cpt = p.Ci / BKT;                // k-tiles per tap
    int aoff[A_IT];
    f32x4 ra[A_IT][2];
    u32x4 rah[A_IT];
    u32x4 rb[B_IT];
    int f_tap = -1;

    auto fetch = [&](int kt) __attribute__((always_inline)) {
        const int tap = kt / cpt, cc = kt - tap * cpt;
        if (tap != f_tap) {                    // wave-uniform: new filter tap -> redo the gather index math
            f_tap = tap;
            const int ky = tap / p.k, kx = tap - ky * p.k;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int iy = refl(ay[i] + ky, p.Hu) >> p.up;
                const int ix = refl(ax[i] + kx, p.Wu) >> p.up;
                aoff[i] = (ab[i] + iy * p.Wi + ix) * p.Ci + q * 8;
            }
        }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            if (A16) rah[i] = *reinterpret_cast<const u32x4*>(p.x16 + (size_t)aoff[i] + cc * BKT);
            else {
                const f32x4* s = reinterpret_cast<const f32x4*>(p.x + (size_t)aoff[i] + cc * BKT);
                ra[i][0] = s[0]; ra[i][1] = s[1];
            }
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            if (BN % RP == 0 || r0 + i * RP < BN) rb[i] = *reinterpret_cast<const u32x4*>(wbase + (size_t)wo[i] + kt * BKT);
    };
    auto stage = [&](int buf, bool real) __attribute__((always_inline)) {
        u32x4* a = As + buf * BM * LDQ;
        u32x4* b = Bs + buf * BN * LDQ;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) a[(r0 + i * RP) * LDQ + q] = A16 ? rah[i] : pack8<T>(ra[i][0], ra[i][1]);
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            if (BN % RP == 0 || r0 + i * RP < BN) b[(r0 + i * RP) * LDQ + q] = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk_all = p.K / BKT;
    const int kbeg = p.phases ? 0 : blockIdx.z * p.nkz, nk = p.phases ? nk_all : min(p.nkz, nk_all - kbeg);
    if (nk <= 0) return;
    const bool split = !p.phases && gridDim.z > 1;
    ACL_GEMM16_MAINLOOP(T, KS, TM, TN, kbeg, nk, As, Bs, BM * LDQ, BN * LDQ, wm * TM * 32, wn * TN * 32);

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
                const int rl = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int o = ro[rl];
                if (o >= 0) {
                    if (split) p.part[((size_t)blockIdx.z * p.rows + m0 + rl) * p.Co + n] = acc[i][j][r];   // ordered partials (always)
                    else st_st1(p.y, (int64_t)o * p.Co + n, act_apply(acc[i][j][r] + bv, p.act), p.yst);
                }
            }
        }
    }
}

// ordered reduction of the split-K partials into a y of any storage dtype (Co % 4 == 0)
__global__ void fwd_split_finish_st_kernel(FwdFP p, int splits) {
    const int C4 = p.Co >> 2;
    const int64_t n = (int64_t)p.rows * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / C4), c4 = (int)(i - (int64_t)m * C4);
        int b, oy, ox;
        if (!fwd_row(p, m, b, oy, ox)) continue;
        f32x4 s = *reinterpret_cast<const f32x4*>(p.part + (size_t)m * p.Co + c4 * 4);
        for (int z = 1; z < splits; ++z) s += *reinterpret_cast<const f32x4*>(p.part + ((size_t)z * p.rows + m) * p.Co + c4 * 4);
        if (p.bias) s += *reinterpret_cast<const f32x4*>(p.bias + c4 * 4);
        st_f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = act_apply(s[e], p.act);
        st_st4(p.y, ((int64_t)(b * p.Ho + oy) * p.Wo + ox) * C4 + c4, o, p.yst);
    }
}

template <class T, int KS, int WM, int WN, int TM, int TN>
int launch_fwd16(const ConvGeom& g, FwdFP p, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, BKT = 16 * KS;
    p.tiles_n = cdiv(g.Co, BN);
    const int rows = p.ring > 0 ? p.B * (p.Ho * p.Wo - std::max(0, p.Ho - 2 * p.ring) * std::max(0, p.Wo - 2 * p.ring)) : p.M;
    p.nwg = cdiv(rows, BM) * p.tiles_n;
    if (p.phases) {
        if (p.x16) hipLaunchKernelGGL((conv_fwd16_kernel<T, KS, WM, WN, TM, TN, true>), dim3(p.nwg, 1, 4), dim3(WM * WN * 64), 0, st, p);
        else hipLaunchKernelGGL((conv_fwd16_kernel<T, KS, WM, WN, TM, TN, false>), dim3(p.nwg, 1, 4), dim3(WM * WN * 64), 0, st, p);
        ACL_CHECK_LAUNCH("conv_fwd16_kernel(phases)");
        return ACLGAN_OK;
    }
    int splits = 1;
    fwd_split_plan(rows, g.Co, g.K, BKT, &splits, &p.nkz);
    p.rows = rows;
    if (splits > 1 && p.part == nullptr) { splits = 1; p.nkz = g.K / BKT; }   // no partial buffer: single pass (never atomics)
    if (p.x16) hipLaunchKernelGGL((conv_fwd16_kernel<T, KS, WM, WN, TM, TN, true>), dim3(p.nwg, 1, splits), dim3(WM * WN * 64), 0, st, p);
    else hipLaunchKernelGGL((conv_fwd16_kernel<T, KS, WM, WN, TM, TN, false>), dim3(p.nwg, 1, splits), dim3(WM * WN * 64), 0, st, p);
    ACL_CHECK_LAUNCH("conv_fwd16_kernel");
    if (splits > 1) {
        if (p.yst == ST_F32) hipLaunchKernelGGL(fwd_split_finish_kernel, dim3((int)std::min<int64_t>(cdiv64((int64_t)rows * std::max(1, g.Co / 4), 256), 4096)), dim3(256), 0, st, p, splits);
        else hipLaunchKernelGGL(fwd_split_finish_st_kernel, dim3((int)std::min<int64_t>(cdiv64((int64_t)rows * (g.Co / 4), 256), 4096)), dim3(256), 0, st, p, splits);
        ACL_CHECK_LAUNCH("fwd_split_finish_kernel");
    }
    return ACLGAN_OK;
}

// ------------------------------------------------------------------------------------------
// dgrad (Cout % 32 == 0, Cin % 32 == 0): interior rows straight into dx, halo ring mirrored in with atomics
// ------------------------------------------------------------------------------------------
template <class T, int KS, int WM, int WN, int TM, int TN>
__global__ void __launch_bounds__(WM * WN * 64) conv_dgrad16_kernel(DgFP pk) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
    constexpr int NCH = 2 * KS, LDQ = NCH + 1, BKT = 16 * KS;
    constexpr int RP = NT / NCH;
    constexpr int A_IT = BM / RP, B_IT = (BN + RP - 1) / RP;
    __shared__ u32x4 smem[2 * (BM + BN) * LDQ];
    __shared__ int ri_o[BM];
    // halo launches (mode 2): most filter taps see no valid output pixel from a ring row (top strip: only ty <= py ...), so the
    // workgroup first collects the taps that matter for ITS rows and loops over those only (3 of 9 for a 3x3 strip tile)
    __shared__ unsigned long long tap_mask;
    __shared__ int tap_list[64];
    u32x4* As = smem;
    u32x4* Bs = smem + 2 * BM * LDQ;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile = xcd_map(blockIdx.x, pk.nwg);
    int m0 = (tile / pk.tiles_n) * BM;
    const int n0 = (tile % pk.tiles_n) * BN;
    DgFP p = pk;                               // local copy: mode 3 resolves to 1 (interior tile) or 2 (halo tile) per workgroup
    if (pk.mode == 3) {
        const int tm = tile / pk.tiles_n;
        if (tm >= pk.Ti) { p.mode = 2; m0 = (tm - pk.Ti) * BM; } else p.mode = 1;
    }
    const bool merged = pk.mode == 3;
    const int cls = blockIdx.z / p.ksplit, slice = blockIdx.z - cls * p.ksplit;
    const int cy = cls / p.s, cx = cls % p.s;
    const int Tx = (p.k - cx + p.s - 1) / p.s, Ty = (p.k - cy + p.s - 1) / p.s;
    const int q = tid % NCH, r0 = tid / NCH;

    int ylo, yhi, xlo, xhi;
    dg_box(p, cy, cx, ylo, yhi, xlo, xhi);
    const bool compact = p.mode == 2 && Ty * Tx <= 64;
    if (tid == 0) tap_mask = 0ull;
    __syncthreads();
    for (int r = tid; r < BM; r += NT) {
        int oo = -1, b, y2, x2;
        if (dg_row(p, m0 + r, ylo, yhi, xlo, xhi, b, y2, x2)) {
            const int py = y2 * p.s + cy, px = x2 * p.s + cx;
            if (compact && py < p.Hp && px < p.Wp) {
                unsigned long long mk = 0ull;
                for (int t = 0; t < Ty * Tx; ++t) {
                    const int ty = t / Tx, tx = t - ty * Tx;
                    const int oy = y2 - ty, ox = x2 - tx;
                    bool ok = (unsigned)oy < (unsigned)p.Ho && (unsigned)ox < (unsigned)p.Wo;
                    if (p.band > 0) ok = ok && (oy < 2 || oy >= p.Ho - 2 || ox < 2 || ox >= p.Wo - 2);
                    if (ok) mk |= 1ull << t;
                }
                if (mk) atomicOr(&tap_mask, mk);
            }
            if (py < p.Hp && px < p.Wp) {
                if (p.mode == 0 || p.ringpad) oo = (b * p.Hp + py) * p.Wp + px;
                else oo = (b * p.Hd + (refl(py - p.pad, p.Hi) >> p.upshift)) * p.Wd + (refl(px - p.pad, p.Wi) >> p.upshift);   // mode 1: identity inside
                // merged launch: interior pixels that also receive mirrored halo rows are combined with atomics (bit 30)
                if (merged && p.mode == 1 && (dg_is_target(py - p.pad, p.Hi, p.pad) || dg_is_target(px - p.pad, p.Wi, p.pad))) oo |= 1 << 30;
            }
        }
        ri_o[r] = oo;
    }
    int ntap = Ty * Tx;
    if (compact) {
        __syncthreads();
        if (tid == 0) {
            int n = 0;
            for (int t = 0; t < Ty * Tx; ++t)
                if ((tap_mask >> t) & 1ull) tap_list[n++] = t;
            tap_list[63] = n;
        }
        __syncthreads();
        ntap = tap_list[63];
        if (ntap == 0) return;           // no row of this tile receives anything (block-uniform)
    }
    int ay[A_IT], ax[A_IT], ab[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        int b = 0, y2 = 0, x2 = 0;
        if (!dg_row(p, m0 + r0 + i * RP, ylo, yhi, xlo, xhi, b, y2, x2)) { b = 0; y2 = 0; x2 = 0; }
        ay[i] = y2; ax[i] = x2; ab[i] = b;
    }
    // B rows = input channels n, k = 32 consecutive cout of one tap: w16t[tap][n][cout]
    int bo[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) bo[i] = min(n0 + r0 + i * RP, p.Ci - 1) * p.Co + q * 8;
    const int cpt = p.Co / BKT;
    int aoff[A_IT];
    f32x4 ra[A_IT][2];
    u32x4 rb[B_IT];
    float za[A_IT];
    int f_tap = -1, tapoff = 0;

    auto fetch = [&](int kt) __attribute__((always_inline)) {
        const int tq = kt / cpt, cc = kt - tq * cpt;
        if (tq != f_tap) {
            f_tap = tq;
            const int t = compact ? tap_list[tq] : tq;
            const int ty = t / Tx, tx = t - ty * Tx;
            tapoff = ((cy + p.s * ty) * p.k + (cx + p.s * tx)) * p.Ci * p.Co;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int oy = ay[i] - ty, ox = ax[i] - tx;
                const bool ok = (unsigned)oy < (unsigned)p.Ho && (unsigned)ox < (unsigned)p.Wo;
                bool okk = ok;
                if (p.band > 0) okk = ok && (oy < 2 || oy >= p.Ho - 2 || ox < 2 || ox >= p.Wo - 2);   // ring outputs only
                const int dpix = p.dyv ? (ab[i] * p.Hf + 2 * (oy + 1) + p.py) * p.Wf + 2 * (ox + 1) + p.px
                                       : (ab[i] * p.Ho + oy) * p.Wo + ox;
                aoff[i] = okk ? dpix * p.Co + q * 8 : -1;
            }
        }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const f32x4* s = reinterpret_cast<const f32x4*>(p.dy + (size_t)(aoff[i] < 0 ? 0 : aoff[i]) + cc * BKT);
            ra[i][0] = s[0]; ra[i][1] = s[1];
            za[i] = aoff[i] < 0 ? 0.f : 1.f;       // applied in stage(): keeps the loads a full tile ahead of their first use
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            if (BN % RP == 0 || r0 + i * RP < BN) rb[i] = *reinterpret_cast<const u32x4*>(p.w16t + (size_t)tapoff + bo[i] + cc * BKT);
    };
    auto stage = [&](int buf, bool real) __attribute__((always_inline)) {
        u32x4* a = As + buf * BM * LDQ;
        u32x4* b = Bs + buf * BN * LDQ;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) a[(r0 + i * RP) * LDQ + q] = pack8<T>(ra[i][0] * za[i], ra[i][1] * za[i]);
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            if (BN % RP == 0 || r0 + i * RP < BN) b[(r0 + i * RP) * LDQ + q] = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk_all = ntap * cpt;
    const int nkz = (nk_all + p.ksplit - 1) / p.ksplit;
    const int kbeg = slice * nkz, nk = min(nkz, nk_all - kbeg);
    if (nk <= 0) return;
    ACL_GEMM16_MAINLOOP(T, KS, TM, TN, kbeg, nk, As, Bs, BM * LDQ, BN * LDQ, wm * TM * 32, wn * TN * 32);

    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + l31;
        if (n >= p.Ci) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int of = ri_o[wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
                if (of >= 0) {
                    const int oo = of & ~(1 << 30);
                    float* o = p.dxp + (size_t)oo * p.Ci + n;
                    if (p.ksplit > 1 || (p.mode == 2 && !p.ringpad) || (of >> 30)) atomicAdd(o, acc[i][j][r]);   // split-K partials / mirrored halo / its targets
                    else if (p.accumulate) *o += acc[i][j][r];
                    else *o = acc[i][j][r];
                }
            }
        }
    }
}

template <class T, int KS, int WM, int WN, int TM, int TN>
int launch_dgrad16(const ConvGeom& g, DgFP p, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    int mmax = 0;
    for (int cy = 0; cy < g.s; ++cy)
        for (int cx = 0; cx < g.s; ++cx) {
            const int ylo = g.p > cy ? (g.p - cy + g.s - 1) / g.s : 0, xlo = g.p > cx ? (g.p - cx + g.s - 1) / g.s : 0;
            const int yhi = std::min(p.Hc - 1, (g.p + g.Hi - 1 - cy) / g.s), xhi = std::min(p.Wc - 1, (g.p + g.Wi - 1 - cx) / g.s);
            int inner = (yhi - ylo + 1) * (xhi - xlo + 1);
            if (p.band > 0) inner = std::max(0, p.Hc - 2 * p.band) * std::max(0, p.Wc - 2 * p.band);
            const int rows = p.mode == 0 ? p.Hc * p.Wc : (p.mode == 1 ? inner : p.Hc * p.Wc - inner);
            mmax = std::max(mmax, g.B * rows);
        }
    if (mmax <= 0) return ACLGAN_OK;
    p.Mc = mmax;
    p.tiles_n = cdiv(g.Ci, BN);
    p.nwg = cdiv(p.Mc, BM) * p.tiles_n;
    const int nk_min = ((g.k + g.s - 1) / g.s) * ((g.k + g.s - 1) / g.s) * (g.Co / (16 * KS));   // k-tiles of the largest parity class
    const int nblk = p.nwg * g.s * g.s;
    p.ksplit = 1;
    const int nk_plan = (p.mode == 2 && p.band == 0) ? std::max(g.Co / (16 * KS), nk_min / ((g.k + g.s - 1) / g.s)) : nk_min;   // halo: useful taps only
    if (nblk < 128 && nk_plan * KS >= 32 && !deterministic()) p.ksplit = max(1, min(nk_plan * KS / 8, 512 / nblk));   // (slices combine with atomics)
    if (p.ksplit > 1 && p.mode == 1 && !p.accumulate) {
        hipError_t e = hipMemsetAsync(p.dxp, 0, (size_t)g.B * g.Hi * g.Wi * g.Ci * sizeof(float), st);
        if (e != hipSuccess) return hip_fail(e, "memset dx");
    }
    hipLaunchKernelGGL((conv_dgrad16_kernel<T, KS, WM, WN, TM, TN>), dim3(p.nwg, 1, g.s * g.s * p.ksplit), dim3(WM * WN * 64), 0, st, p);
    ACL_CHECK_LAUNCH("conv_dgrad16_kernel");
    return ACLGAN_OK;
}

// interior + halo ring in ONE launch (mode 3; see launch_dgrad_fast_merged in conv_fast.hip)
template <class T, int KS, int WM, int WN, int TM, int TN>
int launch_dgrad16_merged(const ConvGeom& g, DgFP p, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    // OPT-IN (ACLGAN_MERGEDHALO=1).  Measured (profiles/r02_experiments.md): correct, but slower -- the 34 halo tiles are a second,
    // nearly empty round after the 512 interior workgroups (one full tile duration of tail), and the divergent atomic/plain
    // epilogue of the interior tiles costs more than the 36-48 us launch it removes: fp32 step 171.7 -> 182.5 ms.
    static int off = -1;
    if (off < 0) { const char* e = getenv("ACLGAN_MERGEDHALO"); off = (e && atoi(e)) ? 0 : 1; }
    if (off || deterministic() || g.p == 0) return ACLGAN_EUNSUPPORTED;
    int mi = 0, mh = 0;
    for (int cy = 0; cy < g.s; ++cy)
        for (int cx = 0; cx < g.s; ++cx) {
            const int ylo = g.p > cy ? (g.p - cy + g.s - 1) / g.s : 0, xlo = g.p > cx ? (g.p - cx + g.s - 1) / g.s : 0;
            const int yhi = std::min(p.Hc - 1, (g.p + g.Hi - 1 - cy) / g.s), xhi = std::min(p.Wc - 1, (g.p + g.Wi - 1 - cx) / g.s);
            const int inner = std::max(0, yhi - ylo + 1) * std::max(0, xhi - xlo + 1);
            mi = std::max(mi, g.B * inner); mh = std::max(mh, g.B * (p.Hc * p.Wc - inner));
        }
    if (mi <= 0 || mh <= 0) return ACLGAN_EUNSUPPORTED;
    p.tiles_n = cdiv(g.Ci, BN);
    p.Ti = cdiv(mi, BM);
    p.nwg = (p.Ti + cdiv(mh, BM)) * p.tiles_n;
    const int nk_min = ((g.k + g.s - 1) / g.s) * ((g.k + g.s - 1) / g.s) * (g.Co / (16 * KS));
    if (p.nwg * g.s * g.s < 128 && nk_min * KS >= 32) return ACLGAN_EUNSUPPORTED;       // small grid: split-K pays more
    p.ksplit = 1; p.mode = 3; p.Mc = mi;
    if (!p.accumulate) {
        const int64_t n = (int64_t)g.B * (2 * g.p * g.Wi + 2 * g.p * g.Hi) * (g.Ci / 4);
        hipLaunchKernelGGL(dg_frame_zero_kernel, dim3((int)std::min<int64_t>(cdiv64(n, 256), 2048)), dim3(256), 0, st, p.dxp, g.B, g.Hi, g.Wi, g.Ci, g.p);
        ACL_CHECK_LAUNCH("dg_frame_zero_kernel");
    }
    hipLaunchKernelGGL((conv_dgrad16_kernel<T, KS, WM, WN, TM, TN>), dim3(p.nwg, 1, g.s * g.s), dim3(WM * WN * 64), 0, st, p);
    ACL_CHECK_LAUNCH("conv_dgrad16_kernel(merged)");
    return ACLGAN_OK;
}

// ------------------------------------------------------------------------------------------
// wgrad (Cout % 64 == 0, Cin % BN == 0 "single tap"): M = Cout, N = (tap, cin), K = pixels (split across blockIdx.z)
// A slice walks its pixel range in sub-chunks of <= 1024 pixels (the per-pixel gather table lives in LDS).  No atomics:
// with one slice the tile is added straight into dw; with several, every slice stores its tile (and its bias column
// sums) to scratch and wgrad16_finish_kernel adds the slices in ORDER -- the weight gradient is reproducible bit for bit.
// ------------------------------------------------------------------------------------------
template <class T, int WM, int WN, int TM, int TN>
__global__ void __launch_bounds__(WM * WN * 64) conv_wgrad16_kernel(WgFP p, WgPartX xp) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
    static_assert(BM % 64 == 0 && BN % 64 == 0 && BM + BN <= NT, "one staging unit (4 channels x 8 pixels) per thread, wave-uniform roles");
    constexpr int CH = 1024;                   // pixels per sub-chunk
    __shared__ u32x4 smem[2 * (BM + BN) * LDQ];
    __shared__ int2 pinfo[CH + BK16];          // per sub-chunk pixel: .x = source pixel of this workgroup's tap, .y = dy pixel
    u32x4* As = smem;
    u32x4* Bs = smem + 2 * BM * LDQ;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile = xcd_map(blockIdx.x, p.nwg);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const int pbeg = blockIdx.z * p.chunk;
    const int pend = min(p.P, pbeg + p.chunk);
    const int phase = p.phases ? (int)blockIdx.y : 0;
    float* dwbase = p.dw + (size_t)phase * p.Co * p.Kn;
    const int py = phase >> 1, px = phase & 1;
    const int st_tap = n0 / p.Ci, st_ky = st_tap / p.k, st_kx = st_tap - st_ky * p.k;   // the whole N tile lies inside ONE filter tap

    // staging roles (wave-uniform): threads [0, BM) build the A tile (dy: rows = cout), [BM, BM+BN) the B tile (x: rows =
    // cin of the tap); a unit = 4 consecutive channels x 8 consecutive chunk pixels
    const bool isA = tid < BM, isB = !isA && tid < BM + BN;
    const int u = isA ? tid : tid - BM;
    const int ncg = (isA ? BM : BN) >> 2;           // channel groups of the tile
    const int cg = u % ncg, pg = u / ncg;           // pg in 0..3: pixels 8pg .. 8pg+7 of the k-tile
    const float* src = isA ? p.dy : p.x;
    const int myst = isA ? p.dyst : p.xst;          // storage of this thread's operand (st16.h); 16-bit: src points at 16-bit data
    const int cstride = isA ? p.Co : p.Ci;
    const int chan = isA ? m0 + 4 * cg : (n0 - st_tap * p.Ci) + 4 * cg;
    u32x4* mytile = isA ? As : Bs;
    const int tstride = (isA ? BM : BN) * LDQ;
    f32x4 rr[8];
    uint2 rh[8];                               // 16-bit storage: 4 channels x 16 bit per pixel
    float zm[8];
    const bool do_bias = p.db != nullptr && (tile % p.tiles_n) == 0;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int cb = pbeg; cb < pend; cb += CH) {
        const int ce = min(pend, cb + CH);
        for (int i = tid; i < ce - cb + BK16; i += NT) {     // + BK16: the clamped tail tile reads past the end
            int b, oy, ox;
            wg_coord(p, min(cb + i, ce - 1), b, oy, ox);
            const int dyp = p.phases ? (b * p.Hf + 2 * (oy + 1) + py) * p.Wf + 2 * (ox + 1) + px : (b * p.Ho + oy) * p.Wo + ox;
            const int iy = refl(oy * p.s - p.p + st_ky, p.Hu) >> p.up, ix = refl(ox * p.s - p.p + st_kx, p.Wu) >> p.up;
            pinfo[i] = make_int2(b * p.Hi * p.Wi + iy * p.Wi + ix, dyp);
        }
        __syncthreads();

        auto fetch = [&](int kt) __attribute__((always_inline)) {
            const int pb = kt * BK16 + 8 * pg;    // sub-chunk relative
            if (isA || isB) {
                if (myst == ST_F32) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int2 pi = pinfo[pb + j];
                        rr[j] = *reinterpret_cast<const f32x4*>(src + (size_t)(isA ? pi.y : pi.x) * cstride + chan);
                        zm[j] = cb + pb + j < ce ? 1.f : 0.f;     // pixels past the sub-chunk contribute nothing (masking both operands is harmless)
                    }
                } else {      // 16-bit storage: 4 channels = 8 bytes per pixel
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int2 pi = pinfo[pb + j];
                        const unsigned int* hp = reinterpret_cast<const unsigned int*>(reinterpret_cast<const u16*>(src) + (size_t)(isA ? pi.y : pi.x) * cstride + chan);
                        rh[j] = *reinterpret_cast<const uint2*>(hp);
                        zm[j] = cb + pb + j < ce ? 1.f : 0.f;
                    }
                }
            }
        };
        auto stage = [&](int buf, bool real) __attribute__((always_inline)) {
            if (isA || isB) {
                u32x4* t = mytile + buf * tstride;
                if (myst == ST_F32) {
                    f32x4 v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = rr[j] * zm[j];
                    if (do_bias && isA && real) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) bsum += v[j];
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        f32x8 k8 = {v[0][c], v[1][c], v[2][c], v[3][c], v[4][c], v[5][c], v[6][c], v[7][c]};
                        t[(c * ncg + cg) * LDQ + pg] = T::pack(k8);     // row interleave: channel 4cg+c -> row c*ncg + cg
                    }
                } else {
                    unsigned int h[8][2];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        h[j][0] = zm[j] != 0.f ? rh[j].x : 0u;
                        h[j][1] = zm[j] != 0.f ? rh[j].y : 0u;
                    }
                    if (do_bias && isA && real) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const st_f32x2 a = st_unpack2(h[j][0], myst), b2 = st_unpack2(h[j][1], myst);
                            bsum[0] += a[0]; bsum[1] += a[1]; bsum[2] += b2[0]; bsum[3] += b2[1];
                        }
                    }
                    // 8 pixels x 4 channels of 16-bit values -> per channel 8 consecutive pixels (v_perm_b32: two halves per instruction)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const unsigned int sel = (c & 1) ? 0x07060302u : 0x05040100u;
                        u32x4 k8;
#pragma unroll
                        for (int q = 0; q < 4; ++q) k8[q] = __builtin_amdgcn_perm(h[2 * q + 1][c >> 1], h[2 * q][c >> 1], sel);
                        t[(c * ncg + cg) * LDQ + pg] = k8;
                    }
                }
            }
        };
        ACL_GEMM16_MAINLOOP(T, 2, TM, TN, 0, (ce - cb + BK16 - 1) / BK16, As, Bs, BM * LDQ, BN * LDQ, wm * TM * 32, wn * TN * 32);
        // (the main loop ends with a barrier: pinfo and both LDS buffers are free for the next sub-chunk)
    }

    const bool partial = xp.part != nullptr;
    if (do_bias) {   // block-uniform
        float* red = reinterpret_cast<float*>(smem);   // [4 pixel groups][BM]
        if (isA) *reinterpret_cast<f32x4*>(red + pg * BM + 4 * cg) = bsum;
        __syncthreads();
        if (tid < BM) {
            const float t = red[tid] + red[BM + tid] + red[2 * BM + tid] + red[3 * BM + tid];
            if (partial) xp.part_b[((size_t)blockIdx.z * xp.ny + phase) * p.Co + m0 + tid] = t;
            else p.db[m0 + tid] += t;          // single slice: this workgroup is the only writer of these channels
        }
    }
    const int l31 = lane & 31, lh = lane >> 5;
    constexpr int MQ = BM / 4, NQ = BN / 4;
    if (partial) {
        float* pt = xp.part + (((size_t)blockIdx.z * xp.ny + phase) * p.nwg + tile) * (BM * BN);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rw = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    pt[rw * BN + wn * TN * 32 + j * 32 + l31] = acc[i][j][r];      // fragment order: 128-byte coalesced rows
                }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int cr = wn * TN * 32 + j * 32 + l31;              // LDS row of the B tile -> channel 4*(row % NQ) + row / NQ
        const int n = n0 + 4 * (cr % NQ) + cr / NQ;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rw = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int m = m0 + 4 * (rw % MQ) + rw / MQ;
                dwbase[(size_t)m * p.Kn + n] += acc[i][j][r];
            }
        }
    }
}

template <class T, int WM, int WN, int TM, int TN>
int launch_wgrad16(const ConvGeom& g, WgFP p, void* part, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int ny = p.phases ? 4 : 1;
    const WgPlan q = wgrad_plan(g.Co, p.Ci, p.Kn, p.P, ny, BK16, WG16_TARGET);
    if (q.BM != BM || q.BN != BN) { set_error("wgrad16: tile plan mismatch"); return ACLGAN_EINVAL; }
    p.tiles_n = q.tiles_n; p.nwg = q.nwg; p.chunk = q.chunk;
    WgPartX xp;
    xp.ny = ny; xp.part = nullptr; xp.part_b = nullptr;
    const bool use_part = q.splits > 1 || ny > 1;
    if (use_part) {
        if (!part) { set_error("wgrad16: %d slices need the scratch buffer (aclgan_conv2d_wgrad16_scratch_bytes)", q.splits); return ACLGAN_EINVAL; }
        xp.part = (float*)part;
        xp.part_b = xp.part + (size_t)q.splits * ny * q.nwg * BM * BN;
    }
    hipLaunchKernelGGL((conv_wgrad16_kernel<T, WM, WN, TM, TN>), dim3(p.nwg, ny, q.splits), dim3(WM * WN * 64), 0, st, p, xp);
    ACL_CHECK_LAUNCH("conv_wgrad16_kernel");
    if (use_part) {
        const int64_t n = (int64_t)ny * g.Co * (p.Kn / 4);
        hipLaunchKernelGGL(wgrad_finish_kernel, dim3((int)std::min<int64_t>(cdiv64(n, 256), 4096)), dim3(256), 0, st, p, xp, BM, BN, q.splits);
        ACL_CHECK_LAUNCH("wgrad_finish_kernel");
    }
    return ACLGAN_OK;
}

// ------------------------------------------------------------------------------------------
// weight packs
// ------------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ u16 cvt1(float v);
template <> __device__ __forceinline__ u16 cvt1<PBF16>(float v) { return __builtin_bit_cast(u16, (__bf16)v); }
template <> __device__ __forceinline__ u16 cvt1<PFP16>(float v) { return __builtin_bit_cast(u16, (_Float16)v); }

// elementwise fp32 -> 16-bit over a whole flat parameter buffer (same offsets)
template <class T>
__global__ void cast16_kernel(const float* __restrict__ src, u16* __restrict__ dst, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + i * 4);
        ushort4 o = make_ushort4(cvt1<T>(v.x), cvt1<T>(v.y), cvt1<T>(v.z), cvt1<T>(v.w));
        *reinterpret_cast<ushort4*>(dst + i * 4) = o;
    }
}

// w[co][tap][ci] (fp32) -> wt[tap][ci][co] (16-bit); 32x32 (co, ci) tiles through LDS; up to PACK_MAX tensors per launch
constexpr int PACK_MAX = 96;
struct PackT { int64_t off[PACK_MAX]; int co[PACK_MAX], taps[PACK_MAX], ci[PACK_MAX], blk0[PACK_MAX + 1]; int n; };
template <class T>
__global__ void transpose16_kernel(const float* __restrict__ base, u16* __restrict__ base_t, PackT t) {
    __shared__ float tile[32][33];
    int lo = 0, hi = t.n - 1;
    const int bid = blockIdx.x;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (t.blk0[mid] <= bid) lo = mid; else hi = mid - 1; }
    const int Co = t.co[lo], Ci = t.ci[lo], taps = t.taps[lo];
    const int tc = (Co + 31) >> 5, ti = (Ci + 31) >> 5;
    int r = bid - t.blk0[lo];
    const int bi = r % ti; r /= ti;
    const int bc = r % tc; const int tap = r / tc;
    const float* w = base + t.off[lo];
    u16* wt = base_t + t.off[lo];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 8 rows per pass
    for (int rr = ty; rr < 32; rr += 8) {
        const int co = bc * 32 + rr, ci = bi * 32 + tx;
        tile[rr][tx] = (co < Co && ci < Ci) ? w[((size_t)co * taps + tap) * Ci + ci] : 0.f;
    }
    __syncthreads();
    for (int rr = ty; rr < 32; rr += 8) {
        const int ci = bi * 32 + rr, co = bc * 32 + tx;
        if (ci < Ci && co < Co) wt[((size_t)tap * Ci + ci) * Co + co] = cvt1<T>(tile[tx][rr]);
    }
}

// merged phase weights of the sub-pixel path, written in both layouts (see up5_merge_kernel):
//   wp [phase][co][a][b][ci]  (forward)      wpt[phase][a][b][ci][co]  (dgrad)
template <class T>
__global__ void up5_merge16_kernel(const float* __restrict__ w, u16* __restrict__ wp, u16* __restrict__ wpt, int Co, int Ci) {
    const int C4 = Ci >> 2;
    const int64_t n = (int64_t)4 * Co * 9 * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        int64_t t = i / C4;
        const int bb = (int)(t % 3); t /= 3;
        const int aa = (int)(t % 3); t /= 3;
        const int co = (int)(t % Co);
        const int ph = (int)(t / Co);
        int ylo, yhi, xlo, xhi;
        up5_range(ph >> 1, aa, ylo, yhi);
        up5_range(ph & 1, bb, xlo, xhi);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int ky = ylo; ky <= yhi; ++ky)
            for (int kx = xlo; kx <= xhi; ++kx)
                s += *reinterpret_cast<const f32x4*>(w + ((size_t)(co * 5 + ky) * 5 + kx) * Ci + c4 * 4);
        u16 h[4] = {cvt1<T>(s.x), cvt1<T>(s.y), cvt1<T>(s.z), cvt1<T>(s.w)};
        if (wp) *reinterpret_cast<ushort4*>(wp + i * 4) = make_ushort4(h[0], h[1], h[2], h[3]);
        if (wpt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) wpt[((((size_t)ph * 3 + aa) * 3 + bb) * Ci + c4 * 4 + e) * Co + co] = h[e];
        }
    }
}

// ACLGAN_TILE16=wide: 128 x 256 tiles (8 waves) for the 256-channel layers (A/B switch; default off, see launch_fwd16_ks)
bool wide_tiles() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("ACLGAN_TILE16"); v = (e && e[0] == 'w') ? 1 : 0; }
    return v == 1;
}

// ---- eligibility ----
bool fwd16_ok(const ConvGeom& g) { return fast_enabled() && g.Ci % 32 == 0 && g.Co % 32 == 0 && (g.up == 0 || up5_eligible(g)); }
bool dgrad16_ok(const ConvGeom& g) { return fast_enabled() && g.Ci % 32 == 0 && g.Co % 32 == 0 && (g.up == 0 || up5_eligible(g)); }
bool wgrad16_ok(const ConvGeom& g) { return fast_enabled() && g.Co % 64 == 0 && g.Ci % 64 == 0 && (g.up == 0 || up5_eligible(g)); }

size_t up5_w16_bytes(const ConvGeom& g) { return ((size_t)4 * g.Co * 9 * g.Ci * sizeof(u16) + 255) & ~(size_t)255; }

FwdFP fwd_params(const ConvGeom& g, const float* x, const u16* w16, const float* bias, float* y, const u16* x16 = nullptr, int yst = 0) {
    FwdFP p;
    p.yst = yst;
    p.part = nullptr; p.rows = 0; p.w = nullptr; p.w16 = w16; p.x16 = x16;
    p.fsl = 0; p.fsx_mod = 0; p.fs_x = p.fs_w = p.fs_y = 0;
    p.x = x; p.bias = bias; p.y = y;
    p.Hi = g.Hi; p.Wi = g.Wi; p.Ci = g.Ci; p.Ho = g.Ho; p.Wo = g.Wo; p.Co = g.Co; p.k = g.k; p.s = g.s; p.p = g.p;
    p.up = g.up; p.Hu = g.Hu; p.Wu = g.Wu; p.M = g.M; p.K = g.K; p.act = g.act; p.tiles_n = 0; p.nwg = 0; p.nkz = 0;
    p.B = g.B; p.ring = 0; p.phases = 0; p.Hf = 0; p.Wf = 0;
    return p;
}

// k-steps per tile: 64-deep tiles whenever the GEMM k axis (Cin: forward, Cout: dgrad) is a multiple of 64
// (not for the 256 x 64 tile of the 64-channel layers: 64-deep it needs 93 KB of LDS and > 256 registers -> one workgroup per CU)
inline int fwd16_bk(const ConvGeom& g) { return (g.Ci % 64 == 0 && g.Co > 64) ? 64 : 32; }
template <class T, int KS>
int launch_fwd16_ks(const ConvGeom& g, const FwdFP& p, hipStream_t st) {
    // (a 128 x 256 tile with 8 waves -- every activation row fetched once per tap -- was measured: the isolated ResBlock
    //  kernel gains 9 %, the whole step loses 2 %: ACLGAN_TILE16=wide keeps it selectable)
    if (g.Co % 256 == 0 && wide_tiles()) return launch_fwd16<T, KS, 2, 4, 2, 2>(g, p, st);
    if (g.Co > 64) return launch_fwd16<T, KS, 2, 2, 2, 2>(g, p, st);
    return launch_fwd16<T, KS, 4, 1, 2, 2>(g, p, st);      // Co 32 / 64: 256 x 64
}
template <class T>
int launch_fwd16_any(const ConvGeom& g, const FwdFP& p, hipStream_t st) {
    return fwd16_bk(g) == 64 ? launch_fwd16_ks<T, 4>(g, p, st) : launch_fwd16_ks<T, 2>(g, p, st);
}
template <class T, int KS>
int launch_dgrad16_ks(const ConvGeom& g, const DgFP& p, hipStream_t st) {
    if (g.Ci % 256 == 0 && wide_tiles()) return launch_dgrad16<T, KS, 2, 4, 2, 2>(g, p, st);
    if (g.Ci > 64) return launch_dgrad16<T, KS, 2, 2, 2, 2>(g, p, st);
    return launch_dgrad16<T, KS, 4, 1, 2, 2>(g, p, st);
}
template <class T, int KS>
int launch_dgrad16_merged_ks(const ConvGeom& g, const DgFP& p, hipStream_t st) {
    if (g.Ci % 256 == 0 && wide_tiles()) return launch_dgrad16_merged<T, KS, 2, 4, 2, 2>(g, p, st);
    if (g.Ci > 64) return launch_dgrad16_merged<T, KS, 2, 2, 2, 2>(g, p, st);
    return launch_dgrad16_merged<T, KS, 4, 1, 2, 2>(g, p, st);
}
template <class T>
int launch_dgrad16_merged_any(const ConvGeom& g, const DgFP& p, hipStream_t st) {
    return (p.Co % 64 == 0 && g.Ci > 64) ? launch_dgrad16_merged_ks<T, 4>(g, p, st) : launch_dgrad16_merged_ks<T, 2>(g, p, st);
}
template <class T>
int launch_dgrad16_any(const ConvGeom& g, const DgFP& p, hipStream_t st) {
    return (p.Co % 64 == 0 && g.Ci > 64) ? launch_dgrad16_ks<T, 4>(g, p, st) : launch_dgrad16_ks<T, 2>(g, p, st);
}
template <class T>
int launch_wgrad16_any(const ConvGeom& g, const WgFP& p, void* part, hipStream_t st) {
    if (g.Co % 128 == 0 && p.Ci % 128 == 0) return launch_wgrad16<T, 2, 2, 2, 2>(g, p, part, st);   // 128 x 128
    if (g.Co % 128 == 0) return launch_wgrad16<T, 2, 2, 2, 1>(g, p, part, st);                       // 128 x 64 (Cin = 64)
    if (p.Ci % 128 == 0) return launch_wgrad16<T, 2, 2, 1, 2>(g, p, part, st);                       // 64 x 128 (Cout = 64)
    return launch_wgrad16<T, 2, 2, 1, 1>(g, p, part, st);                                            // 64 x 64: half the threads stage
}
template <class T>
int fwd16_t(const ConvGeom& g, const float* x, const u16* x16, const float* w, const u16* w16, const float* bias, float* y, void* scratch, hipStream_t st, int yst) {
    if (up5_eligible(g)) {
        if (!scratch || !w) { set_error("conv_fwd16: the upsample+5x5 layer needs its scratch buffer and the fp32 weights"); return ACLGAN_EINVAL; }
        u16* wp = (u16*)scratch;
        const int64_t nm = (int64_t)4 * g.Co * 9 * (g.Ci / 4);
        hipLaunchKernelGGL(up5_merge16_kernel<T>, dim3((int)std::min<int64_t>(cdiv64(nm, 256), 2048)), dim3(256), 0, st, w, wp, (u16*)nullptr, g.Co, g.Ci);
        ACL_CHECK_LAUNCH("up5_merge16_kernel");
        // (1) the four phases: VALID 3x3 conv on the low-res input with the merged weights
        FwdFP p = fwd_params(g, x, wp, bias, y, x16, yst);
        p.Ho = g.Hi - 2; p.Wo = g.Wi - 2; p.k = 3; p.s = 1; p.p = 0; p.up = 0; p.Hu = g.Hi; p.Wu = g.Wi;
        p.M = g.B * p.Ho * p.Wo; p.K = 9 * g.Ci; p.phases = 1; p.Hf = g.Ho; p.Wf = g.Wo;
        ConvGeom gp = g;
        gp.M = p.M; gp.K = p.K;
        int rc = launch_fwd16_any<T>(gp, p, st);
        if (rc) return rc;
        // (2) the output ring of width 2: exact 5x5 gather (reflection at the borders of the upsampled image)
        p = fwd_params(g, x, w16, bias, y, x16, yst);
        p.ring = 2;
        p.part = (float*)((char*)scratch + up5_w16_bytes(g));
        return launch_fwd16_any<T>(g, p, st);
    }
    FwdFP p = fwd_params(g, x, w16, bias, y, x16, yst);
    p.part = (float*)scratch;
    return launch_fwd16_any<T>(g, p, st);
}

DgFP dg_params(const ConvGeom& g, const float* dy, const u16* w16t, float* dx) {
    DgFP p;
    p.dy = dy; p.w = nullptr; p.w16t = w16t; p.dxp = dx;
    p.Ho = g.Ho; p.Wo = g.Wo; p.Co = g.Co; p.Ci = g.Ci; p.k = g.k; p.s = g.s; p.Hp = g.Hp; p.Wp = g.Wp;
    p.Hc = cdiv(g.Hp, g.s); p.Wc = cdiv(g.Wp, g.s); p.Mc = 0; p.tiles_n = 0; p.nwg = 0; p.ksplit = 1;
    p.mode = 1; p.accumulate = 0; p.pad = g.p; p.B = g.B; p.Hi = g.Hu; p.Wi = g.Wu;
    p.dyv = 0; p.py = 0; p.px = 0; p.Hf = 0; p.Wf = 0; p.band = 0; p.upshift = 0; p.Hd = g.Hu; p.Wd = g.Wu;
    return p;
}

template <class T>
int dgrad16_t(const ConvGeom& g, const float* dy, const float* w, const u16* w16t, float* dx, int accumulate, void* scratch, hipStream_t st) {
    if (deterministic()) {
        // every padded-grid position has exactly one writer (no split-K, no mirrored halo); conv_fold gathers the reflection /
        // upsample backward.  The sub-pixel layers run as the plain upsample + 5x5 convolution they are: at 16-bit MFMA rates
        // one launch over the padded grid beats interior + zeroed ring + accumulate-fold (measured: bf16 step 76.8 vs 81.2 ms).
        if (!scratch) { set_error("conv_dgrad16: deterministic mode needs the scratch buffer"); return ACLGAN_EINVAL; }
        DgFP p = dg_params(g, dy, w16t, (float*)scratch);
        p.mode = 0; p.accumulate = 0;
        const int rc = launch_dgrad16_any<T>(g, p, st);
        if (rc) return rc;
        return conv_fold(g, (const float*)scratch, dx, accumulate, st);
    }
    if (up5_eligible(g)) {
        if (!scratch || !w) { set_error("conv_dgrad16: the upsample+5x5 layer needs its scratch buffer and the fp32 weights"); return ACLGAN_EINVAL; }
        u16* wpt = (u16*)scratch;
        const int64_t nm = (int64_t)4 * g.Co * 9 * (g.Ci / 4);
        hipLaunchKernelGGL(up5_merge16_kernel<T>, dim3((int)std::min<int64_t>(cdiv64(nm, 256), 2048)), dim3(256), 0, st, w, (u16*)nullptr, wpt, g.Co, g.Ci);
        ACL_CHECK_LAUNCH("up5_merge16_kernel");
        // (1) four phases: dgrad of the VALID 3x3 conv on the low-res grid, read through the phase view of dy
        ConvGeom gv = g;
        gv.k = 3; gv.s = 1; gv.p = 0; gv.up = 0; gv.Hu = g.Hi; gv.Wu = g.Wi; gv.Hp = g.Hi; gv.Wp = g.Wi;
        gv.Ho = g.Hi - 2; gv.Wo = g.Wi - 2; gv.M = g.B * gv.Ho * gv.Wo; gv.K = 9 * g.Ci;
        DgFP p = dg_params(gv, dy, wpt, dx);
        p.Hi = g.Hi; p.Wi = g.Wi; p.Hd = g.Hi; p.Wd = g.Wi;
        p.dyv = 1; p.Hf = g.Ho; p.Wf = g.Wo;
        for (int ph = 0; ph < 4; ++ph) {
            p.w16t = wpt + (size_t)ph * 9 * g.Ci * g.Co;
            p.py = ph >> 1; p.px = ph & 1;
            p.accumulate = (ph > 0 || accumulate) ? 1 : 0;
            const int rc = launch_dgrad16_any<T>(gv, p, st);
            if (rc) return rc;
        }
        // (2) contributions of the output ring (width 2) through the exact taps, folded into dx with atomics
        p = dg_params(g, dy, w16t, dx);
        p.mode = 2; p.accumulate = 1; p.pad = 2; p.band = 6; p.upshift = 1; p.Hd = g.Hi; p.Wd = g.Wi;
        return launch_dgrad16_any<T>(g, p, st);
    }
    // interior positions straight into dx, then the halo ring mirrored in: together = dgrad + reflection_pad2d backward
    DgFP p = dg_params(g, dy, w16t, dx);
    p.accumulate = accumulate;
    int rc = launch_dgrad16_merged_any<T>(g, p, st);        // one launch: interior tiles + halo tiles
    if (rc != ACLGAN_EUNSUPPORTED) return rc;
    p.mode = 1;
    rc = launch_dgrad16_any<T>(g, p, st);
    if (rc) return rc;
    if (g.p > 0) { p.mode = 2; rc = launch_dgrad16_any<T>(g, p, st); }
    return rc;
}

WgFP wg_params(const ConvGeom& g, const float* x, const float* dy, float* dw, float* db, int xst = 0, int dyst = 0) {
    WgFP p;
    p.xst = xst; p.dyst = dyst;
    p.fsl = 0; p.fsx_mod = 0; p.fs_x = p.fs_dy = 0;
    p.x = x; p.dy = dy; p.dw = dw; p.db = db;
    p.Hi = g.Hi; p.Wi = g.Wi; p.Ci = g.Ci; p.Ho = g.Ho; p.Wo = g.Wo; p.Co = g.Co; p.k = g.k; p.s = g.s; p.p = g.p;
    p.up = g.up; p.Hu = g.Hu; p.Wu = g.Wu; p.P = g.M; p.Kn = g.K; p.chunk = 0; p.tiles_n = 0; p.nwg = 0;
    p.B = g.B; p.ring = 0; p.phases = 0; p.Hf = 0; p.Wf = 0;
    return p;
}

size_t wgrad16_scratch(const ConvGeom& g) { return wgrad_part_scratch(g, BK16, WG16_TARGET); }

template <class T>
int wgrad16_t(const ConvGeom& g, const float* x, const float* dy, float* dw, float* db, void* scratch, hipStream_t st, int xst, int dyst) {
    if (wgrad16_scratch(g) && !scratch) { set_error("conv_wgrad16: this layer needs its scratch buffer"); return ACLGAN_EINVAL; }
    if (up5_eligible(g)) {
        float* dwp = (float*)scratch;
        void* part = (char*)scratch + up5_dwp_bytes(g);
        hipError_t e = hipMemsetAsync(dwp, 0, (size_t)4 * g.Co * 9 * g.Ci * sizeof(float), st);
        if (e != hipSuccess) return hip_fail(e, "memset dwp");
        WgFP p = wg_params(g, x, dy, dwp, db, xst, dyst);
        p.Ho = g.Hi - 2; p.Wo = g.Wi - 2; p.k = 3; p.s = 1; p.p = 0; p.up = 0; p.Hu = g.Hi; p.Wu = g.Wi;
        p.P = g.B * p.Ho * p.Wo; p.Kn = 9 * g.Ci; p.phases = 1; p.Hf = g.Ho; p.Wf = g.Wo;
        int rc = launch_wgrad16_any<T>(g, p, part, st);
        if (rc) return rc;
        const int64_t ns = (int64_t)g.Co * 25 * (g.Ci / 4);
        hipLaunchKernelGGL(up5_scatter_kernel, dim3((int)std::min<int64_t>(cdiv64(ns, 256), 2048)), dim3(256), 0, st, dwp, dw, g.Co, g.Ci);
        ACL_CHECK_LAUNCH("up5_scatter_kernel");
        p = wg_params(g, x, dy, dw, db, xst, dyst);
        p.ring = 2;
        p.P = up5_ring_pixels(g);
        return launch_wgrad16_any<T>(g, p, part, st);
    }
    return launch_wgrad16_any<T>(g, wg_params(g, x, dy, dw, db, xst, dyst), scratch, st);
}

}  // namespace

// ------------------------------------------------------------------------------------------
// entry points (dtype: ACLGAN_DTYPE_BF16 / ACLGAN_DTYPE_FP16)
// ------------------------------------------------------------------------------------------
bool conv16_eligible(const ConvGeom& g, int which) {
    return which == 0 ? fwd16_ok(g) : (which == 1 ? dgrad16_ok(g) : wgrad16_ok(g));
}

size_t conv_fwd16_scratch_bytes(const ConvGeom& g) {
    if (!fwd16_ok(g)) return 0;
    if (up5_eligible(g)) return up5_w16_bytes(g) + fwd_partial_bytes(g, 2, fwd16_bk(g));
    return fwd_partial_bytes(g, 0, fwd16_bk(g));
}
size_t conv_dgrad16_scratch_bytes(const ConvGeom& g) {
    if (!dgrad16_ok(g)) return 0;
    if (deterministic()) return (size_t)g.B * g.Hp * g.Wp * g.Ci * sizeof(float);     // padded-grid gradient
    return up5_eligible(g) ? up5_w16_bytes(g) : 0;
}
size_t conv_wgrad16_scratch_bytes(const ConvGeom& g) { return wgrad16_ok(g) ? std::max(wgrad16_scratch(g), conv_wgrad16s_scratch_bytes(g)) : 0; }

int conv_fwd16(const ConvGeom& g, int dtype, const float* x, const float* w, const void* w16, const float* bias, float* y, void* scratch, hipStream_t st,
               const void* x16, int y_storage) {
    if (!fwd16_ok(g)) return ACLGAN_EUNSUPPORTED;
    const int yst = y_storage ? dtype : 0;
    if (dtype == ACLGAN_DTYPE_BF16) return fwd16_t<PBF16>(g, x, (const u16*)x16, w, (const u16*)w16, bias, y, scratch, st, yst);
    if (dtype == ACLGAN_DTYPE_FP16) return fwd16_t<PFP16>(g, x, (const u16*)x16, w, (const u16*)w16, bias, y, scratch, st, yst);
    set_error("conv_fwd16: dtype %d", dtype);
    return ACLGAN_EINVAL;
}
int conv_dgrad16(const ConvGeom& g, int dtype, const float* dy, const float* w, const void* w16t, float* dx, int accumulate, void* scratch, hipStream_t st) {
    if (!dgrad16_ok(g)) return ACLGAN_EUNSUPPORTED;
    if (dtype == ACLGAN_DTYPE_BF16) return dgrad16_t<PBF16>(g, dy, w, (const u16*)w16t, dx, accumulate, scratch, st);
    if (dtype == ACLGAN_DTYPE_FP16) return dgrad16_t<PFP16>(g, dy, w, (const u16*)w16t, dx, accumulate, scratch, st);
    set_error("conv_dgrad16: dtype %d", dtype);
    return ACLGAN_EINVAL;
}
int conv_wgrad16(const ConvGeom& g, int dtype, const float* x, const float* dy, float* dw, float* db, void* scratch, hipStream_t st, int x_storage,
                 int dy_storage) {
    if (!wgrad16_ok(g) || !dw) return ACLGAN_EUNSUPPORTED;
    const int xst = x_storage ? dtype : 0, dyst = dy_storage ? dtype : 0;
    if (xst && dyst && scratch && conv_wgrad16s_ok(g)) return conv_wgrad16s(g, dtype, x, dy, dw, db, scratch, st);   // both operands 16-bit: the LDS-DMA kernel
    if (dtype == ACLGAN_DTYPE_BF16) return wgrad16_t<PBF16>(g, x, dy, dw, db, scratch, st, xst, dyst);
    if (dtype == ACLGAN_DTYPE_FP16) return wgrad16_t<PFP16>(g, x, dy, dw, db, scratch, st, xst, dyst);
    set_error("conv_wgrad16: dtype %d", dtype);
    return ACLGAN_EINVAL;
}

// fp32 flat buffer -> 16-bit copy with the same offsets (n must be a multiple of 4: flat groups are)
int cast_flat16(const float* src, void* dst, int64_t n, int dtype, hipStream_t st) {
    ACL_REQUIRE(n % 4 == 0, "cast_flat16: n %% 4 != 0");
    const int grid = (int)std::min<int64_t>(cdiv64(n / 4, 256), 8192);
    if (dtype == ACLGAN_DTYPE_BF16) hipLaunchKernelGGL(cast16_kernel<PBF16>, dim3(grid), dim3(256), 0, st, src, (u16*)dst, n / 4);
    else if (dtype == ACLGAN_DTYPE_FP16) hipLaunchKernelGGL(cast16_kernel<PFP16>, dim3(grid), dim3(256), 0, st, src, (u16*)dst, n / 4);
    else { set_error("cast_flat16: dtype %d", dtype); return ACLGAN_EINVAL; }
    ACL_CHECK_LAUNCH("cast16_kernel");
    return ACLGAN_OK;
}

// transposed 16-bit copies [tap][ci][co] of `n` conv tensors living at offs[i] inside `base` (written at the same offsets of base_t)
int transpose_flat16(const float* base, void* base_t, const int64_t* offs, const int* co, const int* taps, const int* ci, int n, int dtype, hipStream_t st) {
    for (int i0 = 0; i0 < n; i0 += PACK_MAX) {
        PackT t;
        t.n = std::min(PACK_MAX, n - i0);
        int blk = 0;
        for (int i = 0; i < t.n; ++i) {
            t.off[i] = offs[i0 + i]; t.co[i] = co[i0 + i]; t.taps[i] = taps[i0 + i]; t.ci[i] = ci[i0 + i];
            t.blk0[i] = blk;
            blk += taps[i0 + i] * cdiv(co[i0 + i], 32) * cdiv(ci[i0 + i], 32);
        }
        t.blk0[t.n] = blk;
        if (blk == 0) continue;
        if (dtype == ACLGAN_DTYPE_BF16) hipLaunchKernelGGL(transpose16_kernel<PBF16>, dim3(blk), dim3(256), 0, st, base, (u16*)base_t, t);
        else if (dtype == ACLGAN_DTYPE_FP16) hipLaunchKernelGGL(transpose16_kernel<PFP16>, dim3(blk), dim3(256), 0, st, base, (u16*)base_t, t);
        else { set_error("transpose_flat16: dtype %d", dtype); return ACLGAN_EINVAL; }
        ACL_CHECK_LAUNCH("transpose16_kernel");
    }
    return ACLGAN_OK;
}

}  // namespace aclgan
