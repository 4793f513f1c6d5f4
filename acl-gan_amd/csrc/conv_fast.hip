// conv_fast.hip -- the tuned implicit-GEMM convolution kernels (gfx950, fp32 MFMA) used whenever the
// channel counts allow 16-deep k-tiles that never straddle a filter tap (all heavy layers of the
// full-width model).  conv.hip keeps the fully general kernels (Cin = 3/6, reduced widths, tails).
//
// What changed relative to the general kernels, and why (profiles/r01_pmc_conv_fwd_v1.txt: MFMA pipe
// 59 % busy, 5.5 VALU instructions per MFMA, the two co-resident waves of a SIMD run their
// load/index phase and their MFMA phase in lockstep):
//   * per-row gather state lives in registers and the reflect/upsample index math is done once
//     per filter TAP (every Cin/16 k-tiles), not per k-tile; invalid rows are address-clamped
//     instead of branched around (garbage rows/columns are never stored);
//   * k-contiguous operands are staged in LDS as [row][16+4] and written / read with b128
//     accesses; the MFMA k index is permuted (lane half h takes k = 8h..8h+7) so one lane reads 8
//     consecutive k values of its row -- any permutation of k is legal as long as A and B agree;
//   * write-after-barrier pipeline (guide T14): global loads for tile t+2 are issued into the
//     SAME registers right after tile t+1 was written to LDS, so a load has a whole MFMA phase
//     to land and no second register set is needed;
//   * the register->LDS writes and the next loads are placed between the MFMA groups of the
//     current tile, so each wave's own instruction stream overlaps index/memory work with its
//     MFMAs instead of relying on a second wave being in the complementary phase.
#include "conv_fast_common.h"
#include <cstring>

namespace aclgan {
namespace {


constexpr int BK = 16;
constexpr int WG_MAX_CHUNK = 1024 + 16;   // wgrad: pixels per split-K chunk whose gather state is cached in LDS
constexpr int LDK = BK + 4;   // row stride (floats) of a k-contiguous LDS tile: 80 B keeps b128 rows conflict-free

// fragment fetch for one k-tile.  KC: tile stored [row][LDK]; MC: tile stored [k][ld] (row index contiguous).
template <int T, bool KC>
__device__ __forceinline__ void read_frags(const float* __restrict__ tile, int ld, int base, int lane, float (&f)[T][8]) {
    const int l31 = lane & 31, kh = lane >> 5;
    if (KC) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const f32x4* p = reinterpret_cast<const f32x4*>(tile + (base + t * 32 + l31) * LDK + kh * 8);
            const f32x4 a = p[0], b = p[1];
            f[t][0] = a.x; f[t][1] = a.y; f[t][2] = a.z; f[t][3] = a.w;
            f[t][4] = b.x; f[t][5] = b.y; f[t][6] = b.z; f[t][7] = b.w;
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int t = 0; t < T; ++t) f[t][ks] = tile[(kh * 8 + ks) * ld + base + t * 32 + l31];
    }
}

template <int TM, int TN>
__device__ __forceinline__ void mfma_step(const float (&fa)[TM][8], const float (&fb)[TN][8], int ks, f32x16 (&acc)[TM][TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][ks], fb[j][ks], acc[i][j], 0, 0, 0);
}

// The shared main loop, as a macro: `stage(buf)` writes the staged registers (tile t+1) to LDS buffer
// `buf`, `fetch(t)` issues the global loads of tile t into the staging registers; both are lambdas
// defined in the calling kernel.  (A function template taking the closures by value defeats SROA:
// hipcc then keeps the staging registers in scratch memory -- measured, 2x slower.)
// stage/fetch are deliberately UNCONDITIONAL inside the loop (tile indices clamped to nk-1, the
// redundant tail work is harmless): a conditional fetch turns the staging registers into a loop PHI
// and hipcc copies them behind an immediate s_waitcnt vmcnt, exposing the full memory latency.
#define ACL_GEMM_MAINLOOP(TM_, TN_, AKC_, BKC_, KBEG_, NK_, AS_, BS_, ASTR_, BSTR_, LDA_, LDB_, AM_, BN_) \
    do {                                                                                                 \
        const int nk__ = (NK_), kb__ = (KBEG_);                                                          \
        fetch(kb__);                                                                                     \
        stage(0, true);                                                                                  \
        fetch(kb__ + min(1, nk__ - 1));                                                                  \
        __syncthreads();                                                                                 \
        for (int kt = 0; kt < nk__; ++kt) {                                                              \
            const int cur = kt & 1;                                                                      \
            float fa[TM_][8], fb[TN_][8];                                                                \
            read_frags<TM_, AKC_>((AS_) + cur * (ASTR_), (LDA_), (AM_), lane, fa);                       \
            read_frags<TN_, BKC_>((BS_) + cur * (BSTR_), (LDB_), (BN_), lane, fb);                       \
            if (!(AKC_) && !(BKC_)) __builtin_amdgcn_sched_barrier(0);   /* wgrad: keep all b32 fragment reads ahead of the MFMA chain */ \
            mfma_step<TM_, TN_>(fa, fb, 0, acc);                                                         \
            stage(cur ^ 1, kt + 1 < nk__);                                                               \
            fetch(kb__ + min(kt + 2, nk__ - 1));                                                         \
            __builtin_amdgcn_sched_barrier(0x6);   /* only ALU may cross: the loads of tile kt+2 issue HERE, a full tile ahead of their use */ \
            mfma_step<TM_, TN_>(fa, fb, 1, acc);                                                         \
            mfma_step<TM_, TN_>(fa, fb, 2, acc);                                                         \
            mfma_step<TM_, TN_>(fa, fb, 3, acc);                                                         \
            mfma_step<TM_, TN_>(fa, fb, 4, acc);                                                         \
            mfma_step<TM_, TN_>(fa, fb, 5, acc);                                                         \
            mfma_step<TM_, TN_>(fa, fb, 6, acc);                                                         \
            mfma_step<TM_, TN_>(fa, fb, 7, acc);                                                         \
            __syncthreads();                                                                             \
        }                                                                                                \
    } while (0)


// ------------------------------------------------------------------------------------------
// forward (Cin % 16 == 0)
// ------------------------------------------------------------------------------------------

// OCC: minimum workgroups per CU the register allocator must allow (1 = no constraint: the direct convolutions; 3 for the
// batched GEMMs of the Winograd path, whose 1152-workgroup grids leave a half-empty third round at 2 workgroups per CU)
template <int WM, int WN, int TM, int TN, int OCC = 1>
__global__ void __launch_bounds__(WM * WN * 64, OCC) conv_fwd_fast_kernel(FwdFP pk) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
    constexpr int RP = NT / 4;                 // rows covered per pass (4 float4 chunks per 16-float row)
    FwdFP p = pk;
    if (pk.fsl) {   // batched GEMM slice
        const long long f = blockIdx.y;
        p.x += (pk.fsx_mod ? f % pk.fsx_mod : f) * pk.fs_x; p.w += f * pk.fs_w; p.y += f * pk.fs_y;
    }
    constexpr int A_IT = BM / RP, B_IT = (BN + RP - 1) / RP;
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDK];
    __shared__ int ro[BM];                     // output pixel index of each tile row (-1: not stored)
    float* As = smem;
    float* Bs = smem + 2 * BM * LDK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile = xcd_map(blockIdx.x, p.nwg);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const int q = tid & 3, r0 = tid >> 2;
    const int phase = p.phases ? (int)blockIdx.z : 0;
    const float* wbase = p.w + (size_t)phase * p.Co * p.K;

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
    for (int i = 0; i < B_IT; ++i) wo[i] = min(n0 + r0 + i * RP, p.Co - 1) * p.K + q * 4;

    const int cpt = p.Ci >> 4;                 // k-tiles per tap
    int aoff[A_IT];
    f32x4 ra[A_IT], rb[B_IT];
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
                aoff[i] = (ab[i] + iy * p.Wi + ix) * p.Ci + q * 4;
            }
        }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) ra[i] = *reinterpret_cast<const f32x4*>(p.x + (size_t)aoff[i] + cc * 16);
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            if (BN % RP == 0 || r0 + i * RP < BN) rb[i] = *reinterpret_cast<const f32x4*>(wbase + (size_t)wo[i] + kt * 16);
    };
    auto stage = [&](int buf, bool real) __attribute__((always_inline)) {
        float* a = As + buf * BM * LDK;
        float* b = Bs + buf * BN * LDK;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *reinterpret_cast<f32x4*>(a + (r0 + i * RP) * LDK + q * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            if (BN % RP == 0 || r0 + i * RP < BN) *reinterpret_cast<f32x4*>(b + (r0 + i * RP) * LDK + q * 4) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk_all = p.K / BK;
    const int kbeg = p.phases ? 0 : blockIdx.z * p.nkz, nk = p.phases ? nk_all : min(p.nkz, nk_all - kbeg);
    if (nk <= 0) return;
    const bool split = !p.phases && gridDim.z > 1;
    ACL_GEMM_MAINLOOP(TM, TN, true, true, kbeg, nk, As, Bs, BM * LDK, BN * LDK, LDK, LDK, wm * TM * 32, wn * TN * 32);

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
                    if (split && p.part) p.part[((size_t)blockIdx.z * p.rows + m0 + rl) * p.Co + n] = acc[i][j][r];
                    else if (split) atomicAdd(p.y + (size_t)o * p.Co + n, acc[i][j][r] + ((p.ring > 0 && blockIdx.z == 0) ? bv : 0.f));   // partial sums (ring launches: slice 0 carries the bias, act is none)
                    else p.y[(size_t)o * p.Co + n] = act_apply(acc[i][j][r] + bv, p.act);
                }
            }
        }
    }
}


template <int WM, int WN, int TM, int TN>
int launch_fwd_fast(const ConvGeom& g, FwdFP p, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    p.tiles_n = cdiv(g.Co, BN);
    int rows = p.M;
    if (p.ring > 0) rows = p.B * (p.Ho * p.Wo - std::max(0, p.Ho - 2 * p.ring) * std::max(0, p.Wo - 2 * p.ring));
    p.nwg = cdiv(rows, BM) * p.tiles_n;
    if (p.phases) {
        hipLaunchKernelGGL((conv_fwd_fast_kernel<WM, WN, TM, TN>), dim3(p.nwg, 1, 4), dim3(WM * WN * 64), 0, st, p);
        ACL_CHECK_LAUNCH("conv_fwd_fast_kernel(phases)");
        return ACLGAN_OK;
    }
    // small grids (late discriminator layers: M = B*16 .. B*256 pixels, K = 2048..4096; the ring of the sub-pixel path):
    // split K across blockIdx.z so the chip is filled.  With a partial buffer (p.part, sized by conv_fwd_scratch_bytes)
    // every slice stores its tile and fwd_split_finish_kernel adds them in order: reproducible bit for bit.  Without it
    // the slices are combined with fp32 atomics into a pre-zeroed output (+ a bias/activation pass): same value up to
    // summation order, not reproducible run to run.
    int splits = 1;
    fwd_split_plan(rows, g.Co, g.K, BK, &splits, &p.nkz);
    p.rows = rows;
    if (splits > 1 && p.part != nullptr) {
        hipLaunchKernelGGL((conv_fwd_fast_kernel<WM, WN, TM, TN>), dim3(p.nwg, 1, splits), dim3(WM * WN * 64), 0, st, p);
        ACL_CHECK_LAUNCH("conv_fwd_fast_kernel(split)");
        hipLaunchKernelGGL(fwd_split_finish_kernel, dim3((int)std::min<int64_t>(cdiv64((int64_t)rows * std::max(1, g.Co / 4), 256), 4096)), dim3(256), 0, st, p, splits);
        ACL_CHECK_LAUNCH("fwd_split_finish_kernel");
        return ACLGAN_OK;
    }
    p.part = nullptr;
    const int nk = g.K / BK;
    if (p.ring > 0 && (p.act != ACLGAN_ACT_NONE || g.Co % 4 != 0)) splits = 1, p.nkz = nk;   // ring + activation: single pass
    if (deterministic()) splits = 1, p.nkz = nk;                                             // no partial buffer: the slices would meet in atomics
    if (splits > 1 && p.ring > 0) {
        hipLaunchKernelGGL(ring_zero_kernel, dim3(cdiv(rows * (g.Co / 4), 256)), dim3(256), 0, st, p, rows);
        ACL_CHECK_LAUNCH("ring_zero_kernel");
    } else if (splits > 1) {
        hipError_t e = hipMemsetAsync(p.y, 0, (size_t)g.M * g.Co * sizeof(float), st);
        if (e != hipSuccess) return hip_fail(e, "memset y");
    }
    hipLaunchKernelGGL((conv_fwd_fast_kernel<WM, WN, TM, TN>), dim3(p.nwg, 1, splits), dim3(WM * WN * 64), 0, st, p);
    ACL_CHECK_LAUNCH("conv_fwd_fast_kernel");
    if (splits > 1 && p.ring == 0) {
        const int64_t n = (int64_t)g.M * g.Co;
        hipLaunchKernelGGL(bias_act_kernel, dim3((int)std::min<int64_t>(cdiv64(n, 256), 4096)), dim3(256), 0, st, p.y, p.bias, g.Co, p.act, n);
        ACL_CHECK_LAUNCH("bias_act_kernel");
    }
    return ACLGAN_OK;
}

// ------------------------------------------------------------------------------------------
// dgrad onto the padded grid (Cout % 16 == 0, Cin % 4 == 0); the fold kernel of conv.hip follows
// ------------------------------------------------------------------------------------------

template <int WM, int WN, int TM, int TN>
__global__ void __launch_bounds__(WM * WN * 64) conv_dgrad_fast_kernel(DgFP pk) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
    constexpr int RP = NT / 4;
    constexpr int A_IT = BM / RP;
    constexpr int LDB = BN + 4, NVB = BN / 4;
    constexpr int B_IT = (BK * NVB + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM * LDK + BK * LDB)];
    __shared__ int ri_o[BM];
    // halo launches (mode 2): most filter taps see no valid output pixel from a ring row (top strip: only ty <= py ...), so the
    // workgroup first collects the taps that matter for ITS rows and loops over those only (3 of 9 for a 3x3 strip tile)
    __shared__ unsigned long long tap_mask;
    __shared__ int tap_list[64];
    float* As = smem;
    float* Bs = smem + 2 * BM * LDK;

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
    const int q = tid & 3, r0 = tid >> 2;

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
        if (!dg_row(p, m0 + r0 + i * RP, ylo, yhi, xlo, xhi, b, y2, x2)) { b = 0; y2 = 0; x2 = 0; }   // past the end: any valid row (never stored)
        ay[i] = y2; ax[i] = x2; ab[i] = b;
    }
    // B tile rows = 16 consecutive cout of one tap, columns = cin (contiguous)
    int bo[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int idx = tid + i * NT;
        const int krow = idx / NVB, nc = idx - krow * NVB;
        const int n = n0 + nc * 4;
        bo[i] = krow * p.k * p.k * p.Ci + (n < p.Ci ? n : 0);
    }
    const int cpt = p.Co >> 4;
    int aoff[A_IT];
    f32x4 ra[A_IT], rb[B_IT];
    float za[A_IT];
    int f_tap = -1, tapoff = 0;

    auto fetch = [&](int kt) __attribute__((always_inline)) {
        const int tq = kt / cpt, cc = kt - tq * cpt;
        if (tq != f_tap) {
            f_tap = tq;
            const int t = compact ? tap_list[tq] : tq;
            const int ty = t / Tx, tx = t - ty * Tx;
            tapoff = ((cy + p.s * ty) * p.k + (cx + p.s * tx)) * p.Ci;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int oy = ay[i] - ty, ox = ax[i] - tx;
                const bool ok = (unsigned)oy < (unsigned)p.Ho && (unsigned)ox < (unsigned)p.Wo;
                bool okk = ok;
                if (p.band > 0) okk = ok && (oy < 2 || oy >= p.Ho - 2 || ox < 2 || ox >= p.Wo - 2);   // ring outputs only
                const int dpix = p.dyv ? (ab[i] * p.Hf + 2 * (oy + 1) + p.py) * p.Wf + 2 * (ox + 1) + p.px
                                       : (ab[i] * p.Ho + oy) * p.Wo + ox;
                aoff[i] = okk ? dpix * p.Co + q * 4 : -1;
            }
        }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            // the zero-fill mask is applied in stage(), NOT here: a multiply on the freshly loaded value makes
            // hipcc wait for the load inside the same iteration (s_waitcnt vmcnt right after the issue)
            ra[i] = *reinterpret_cast<const f32x4*>(p.dy + (size_t)(aoff[i] < 0 ? 0 : aoff[i]) + cc * 16);
            za[i] = aoff[i] < 0 ? 0.f : 1.f;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            if ((BK * NVB) % NT == 0 || tid + i * NT < BK * NVB)
                rb[i] = *reinterpret_cast<const f32x4*>(p.w + (size_t)bo[i] + tapoff + (size_t)cc * 16 * p.k * p.k * p.Ci);
    };
    auto stage = [&](int buf, bool real) __attribute__((always_inline)) {
        float* a = As + buf * BM * LDK;
        float* b = Bs + buf * BK * LDB;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *reinterpret_cast<f32x4*>(a + (r0 + i * RP) * LDK + q * 4) = ra[i] * za[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int idx = tid + i * NT;
            if ((BK * NVB) % NT == 0 || idx < BK * NVB) {
                const int krow = idx / NVB, nc = idx - krow * NVB;
                *reinterpret_cast<f32x4*>(b + krow * LDB + nc * 4) = rb[i];
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

    const int nk_all = ntap * cpt;
    const int nkz = (nk_all + p.ksplit - 1) / p.ksplit;
    const int kbeg = slice * nkz, nk = min(nkz, nk_all - kbeg);
    if (nk <= 0) return;
    ACL_GEMM_MAINLOOP(TM, TN, true, false, kbeg, nk, As, Bs, BM * LDK, BK * LDB, LDK, LDB, wm * TM * 32, wn * TN * 32);

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

int halo_split_override();
template <int WM, int WN, int TM, int TN>
int launch_dgrad_fast(const ConvGeom& g, DgFP p, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    // rows per class (max over the s*s parity classes) for this enumeration mode
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
    const int nk_min = ((g.k + g.s - 1) / g.s) * ((g.k + g.s - 1) / g.s) * (g.Co / 16);   // k-tiles of the largest parity class
    const int nblk = p.nwg * g.s * g.s;
    p.ksplit = 1;
    // halo launches loop over the useful taps only (about 1 / taps-per-axis of them): plan the slices for that shorter loop --
    // every slice ends in 128 x 128 fp32 atomics, which is what the old 15-slice halo launch mostly consisted of
    const int nk_plan = (p.mode == 2 && p.band == 0) ? std::max(g.Co / 16, nk_min / ((g.k + g.s - 1) / g.s)) : nk_min;
    if (nblk < 128 && nk_plan >= 32 && !deterministic() && !p.ringpad)       // (the slices combine with fp32 atomics; ringpad: one plain store per position)
        p.ksplit = max(1, min(nk_plan / 8, 512 / nblk));   // floor: stay within one round of 512 resident workgroups
    if (p.mode == 2 && p.band == 0 && !p.ringpad && !deterministic() && halo_split_override() > 0) p.ksplit = std::min(halo_split_override(), nk_plan);
    if (p.ksplit > 1 && p.mode == 0) {
        hipError_t e = hipMemsetAsync(p.dxp, 0, conv_dgrad_scratch_bytes(g), st);
        if (e != hipSuccess) return hip_fail(e, "memset dxp");
    }
    if (p.ksplit > 1 && p.mode == 1 && !p.accumulate) {
        hipError_t e = hipMemsetAsync(p.dxp, 0, (size_t)g.B * g.Hi * g.Wi * g.Ci * sizeof(float), st);
        if (e != hipSuccess) return hip_fail(e, "memset dx");
    }
    hipLaunchKernelGGL((conv_dgrad_fast_kernel<WM, WN, TM, TN>), dim3(p.nwg, 1, g.s * g.s * p.ksplit), dim3(WM * WN * 64), 0, st, p);
    ACL_CHECK_LAUNCH("conv_dgrad_fast_kernel");
    return ACLGAN_OK;
}

// interior + halo ring in ONE launch (mode 3).  Returns EUNSUPPORTED when the grid is so small that the split-K two-launch
// path is the better plan.  The separate halo launch is 34 x 15 workgroups with a full k loop: 48 us of a 354 us ResBlock dgrad
// (fp32), 36 of 116 us (bf16); as 17 extra M-tiles of the interior launch it is only +6 % work -- but see the measurement below.
template <int WM, int WN, int TM, int TN>
int launch_dgrad_fast_merged(const ConvGeom& g, DgFP p, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    // OPT-IN (ACLGAN_MERGEDHALO=1).  Measured (profiles/r02_experiments.md): correct, but slower -- the 34 halo tiles are a second,
    // nearly empty round after the 512 interior workgroups (one full tile duration of tail), and the divergent atomic/plain
    // epilogue of the interior tiles costs more than the 36-48 us launch it removes: fp32 step 171.7 -> 182.5 ms.
    static int off = -1;
    if (off < 0) { const char* e = getenv("ACLGAN_MERGEDHALO"); off = (e && atoi(e)) ? 0 : 1; }
    if (off || deterministic() || g.p == 0 || g.Ci % 4 != 0) return ACLGAN_EUNSUPPORTED;
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
    const int nk_min = ((g.k + g.s - 1) / g.s) * ((g.k + g.s - 1) / g.s) * (g.Co / 16);
    if (p.nwg * g.s * g.s < 128 && nk_min >= 32) return ACLGAN_EUNSUPPORTED;       // small grid: split-K pays more
    p.ksplit = 1; p.mode = 3; p.Mc = mi;
    if (!p.accumulate) {
        const int64_t n = (int64_t)g.B * (2 * g.p * g.Wi + 2 * g.p * g.Hi) * (g.Ci / 4);
        hipLaunchKernelGGL(dg_frame_zero_kernel, dim3((int)std::min<int64_t>(cdiv64(n, 256), 2048)), dim3(256), 0, st, p.dxp, g.B, g.Hi, g.Wi, g.Ci, g.p);
        ACL_CHECK_LAUNCH("dg_frame_zero_kernel");
    }
    hipLaunchKernelGGL((conv_dgrad_fast_kernel<WM, WN, TM, TN>), dim3(p.nwg, 1, g.s * g.s), dim3(WM * WN * 64), 0, st, p);
    ACL_CHECK_LAUNCH("conv_dgrad_fast_kernel(merged)");
    return ACLGAN_OK;
}

// the halo ring of a reflection-padded layer (mode 2): its GEMM is short and wide (3x3 ResBlock layer at 64x64 B=8: 2 016 ring rows x 256 x 768
// useful k) and latency-bound -- a single 128 x 128 slice takes 2.7 us per 16-channel k-tile, three times its MFMAs.  Measured on that layer
// (kernel trace of the operator, scripts/r06/gpu15.sh; tile x split count): 128 x 128: 131 / 74.5 / 54.2 / 45.9 / 38.5 us with 1 / 2 / 3 / 4 / 6
// slices; 64 x 64: 54.7 (1) / 39.3 (2) / 35.0 (3); 64 x 128: **34.3** (planned: 6) / 36.4 (3); 128 x 64: 35.9 (6) / 38.7 (3).  No shape gets under
// ~34 us: the ring is a fixed cost of prologue (tap lists), a short dependent loop and the mirrored atomics.  Default: 64 x 128 on the 3x3
// stride-1 layers (what was measured), the caller's tile elsewhere.  ACLGAN_HALO_TILE = 1 / 2 / 3 forces 64 x 64 / 64 x 128 / 128 x 64 on every
// layer with Cin % 64 == 0, 4 = the caller's tile everywhere; ACLGAN_HALO_SPLIT overrides the slice count.
static int halo_tile_mode() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("ACLGAN_HALO_TILE"); v = e ? atoi(e) : 0; if (v < 0 || v > 4) v = 0; }
    return v;
}
int halo_split_override() {
    static int v = -2;
    if (v == -2) { const char* e = getenv("ACLGAN_HALO_SPLIT"); v = e ? atoi(e) : 0; }
    return v;
}
template <int WM, int WN, int TM, int TN>
int launch_dgrad_halo(const ConvGeom& g, DgFP p, hipStream_t st) {
    if (p.band == 0 && !p.ringpad && g.Ci % 64 == 0) {
        int t = halo_tile_mode();
        if (t == 0 && g.k == 3 && g.s == 1 && g.Ci % 128 == 0 && WM == 2 && WN == 2 && TM == 2 && TN == 2) t = 2;
        if (t == 1) return launch_dgrad_fast<2, 2, 1, 1>(g, p, st);       // 64 x 64
        if (t == 2) return launch_dgrad_fast<2, 2, 1, 2>(g, p, st);       // 64 x 128
        if (t == 3) return launch_dgrad_fast<2, 2, 2, 1>(g, p, st);       // 128 x 64
    }
    return launch_dgrad_fast<WM, WN, TM, TN>(g, p, st);
}

template <int WM, int WN, int TM, int TN>
int dgrad_fast_all(const ConvGeom& g, DgFP p, float* dxp, float* dx, int accumulate, bool* direct, hipStream_t st) {
    // deterministic mode: the Winograd layers keep their fast interior and fold the ring in order (below); every other layer
    // takes the padded-grid plan -- one launch, one writer per element, then the fold gather
    if (g.up == 0 && dx != nullptr && (!deterministic() || g.p == 0 || (dxp && conv_wino_ok(g)))) {
        // interior positions straight into dx (balanced grid, no scratch round trip) and the halo ring mirrored in with
        // atomics: together = dgrad + reflection_pad2d backward
        *direct = true;
        p.dxp = dx; p.accumulate = accumulate;
        int rc = launch_dgrad_fast_merged<WM, WN, TM, TN>(g, p, st);
        if (rc != ACLGAN_EUNSUPPORTED) return rc;
        p.mode = 1;
        rc = (dxp && conv_wino_ok(g)) ? conv_dgrad_wino_interior(g, p.dy, p.w, dx, accumulate, dxp, st)   // interior: Winograd (zero pad, flipped w^T)
             : (dxp && conv_s2k4_wino_ok(g, 1)) ? conv_dgrad_s2k4_wino_interior(g, p.dy, p.w, dx, accumulate, dxp, st)      // 4x4 stride 2: four parity phases
                                      : launch_dgrad_fast<WM, WN, TM, TN>(g, p, st);
        if (rc) return rc;
        if (g.p > 0 && deterministic()) {
            // ordered reflection backward: the ring positions are STORED on a zeroed padded grid (one writer each; the scratch is
            // free again -- the Winograd planes of the interior are consumed) and conv_fold adds them onto their targets
            hipError_t e = hipMemsetAsync(dxp, 0, (size_t)g.B * g.Hp * g.Wp * g.Ci * sizeof(float), st);
            if (e != hipSuccess) return hip_fail(e, "memset padded grid");
            p.mode = 2; p.ringpad = 1; p.dxp = dxp; p.accumulate = 0;
            rc = launch_dgrad_fast<WM, WN, TM, TN>(g, p, st);
            if (rc) return rc;
            return conv_fold(g, dxp, dx, 1, st);
        }
        if (g.p > 0) { p.mode = 2; rc = launch_dgrad_halo<WM, WN, TM, TN>(g, p, st); }
        return rc;
    }
    *direct = false;
    p.dxp = dxp; p.mode = 0; p.accumulate = 0;
    return launch_dgrad_fast<WM, WN, TM, TN>(g, p, st);
}

// ------------------------------------------------------------------------------------------
// wgrad (Cout % 4 == 0, Cin % 4 == 0): M = Cout, N = (tap, cin), K = pixels (split across blockIdx.z)
// ------------------------------------------------------------------------------------------

// ST ("single tap"): Cin % BN == 0, so the whole N tile of a workgroup lies inside ONE filter tap: the reflected /
// upsampled source pixel of every chunk pixel is resolved once in the pixel table and the per-tile gather is one
// LDS read + one multiply-add per load instead of two reflections per load and tile.
template <int WM, int WN, int TM, int TN, bool ST>
__global__ void __launch_bounds__(WM * WN * 64) conv_wgrad_fast_kernel(WgFP p) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
    constexpr int LDA = BM + 4, LDB = BN + 4, MVA = BM / 4, NVB = BN / 4;
    constexpr int A_IT = (BK * MVA + NT - 1) / NT, B_IT = (BK * NVB + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LDA + LDB)];
    float* As = smem;
    float* Bs = smem + 2 * BK * LDA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile = xcd_map(blockIdx.x, p.nwg);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const int pbeg = blockIdx.z * p.chunk;
    const int pend = min(p.P, pbeg + p.chunk);
    if (pbeg >= pend) return;
    const int phase = p.phases ? (int)blockIdx.y : 0;
    float* dwbase = p.dw + (size_t)phase * p.Co * p.Kn + (size_t)blockIdx.z * p.dw_zs;
    const int py = phase >> 1, px = phase & 1;

    // Per-pixel gather state of this block's pixel chunk, decoded ONCE into LDS (the old per-tile
    // coordinate bookkeeping cost 6 VALU instructions per MFMA: profiles/r01_pmc notes):
    //   pinfo[i] = { oy*s - p, ox*s - p, b*Hi*Wi, dy pixel index } for chunk pixel i (clamped past the end)
    __shared__ int4 pinfo[WG_MAX_CHUNK];
    const int st_tap = n0 / p.Ci, st_ky = st_tap / p.k, st_kx = st_tap - st_ky * p.k;   // ST: this workgroup's filter tap
    for (int i = tid; i < pend - pbeg + BK; i += NT) {     // + BK: the clamped tail tile reads up to 15 entries past the end
        int b, oy, ox;
        wg_coord(p, min(pbeg + i, pend - 1), b, oy, ox);
        const int dyp = p.phases ? (b * p.Hf + 2 * (oy + 1) + py) * p.Wf + 2 * (ox + 1) + px : (b * p.Ho + oy) * p.Wo + ox;
        if (ST) {
            const int iy = refl(oy * p.s - p.p + st_ky, p.Hu) >> p.up, ix = refl(ox * p.s - p.p + st_kx, p.Wu) >> p.up;
            pinfo[i] = make_int4(b * p.Hi * p.Wi + iy * p.Wi + ix, 0, 0, dyp);   // .x = source pixel of this tap
        } else {
            pinfo[i] = make_int4(oy * p.s - p.p, ox * p.s - p.p, b * p.Hi * p.Wi, dyp);
        }
    }
    __syncthreads();

    // A: thread -> (pixel row krow, 4 consecutive cout); B: thread -> (pixel row krow, 4 consecutive cin of one tap)
    int a_m[A_IT], a_kr[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int idx = tid + i * NT;
        a_kr[i] = idx / MVA;
        const int m = m0 + (idx - a_kr[i] * MVA) * 4;
        a_m[i] = m < p.Co ? m : 0;
    }
    int b_kr[B_IT], b_ci[B_IT], b_ky[B_IT], b_kx[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int idx = tid + i * NT;
        b_kr[i] = idx / NVB;
        int n = n0 + (idx - b_kr[i] * NVB) * 4;
        if (n >= p.Kn) n = 0;
        const int tap = n / p.Ci;
        b_ci[i] = n - tap * p.Ci; b_ky[i] = tap / p.k; b_kx[i] = tap - b_ky[i] * p.k;
    }
    f32x4 ra[A_IT], rb[B_IT];
    float za[A_IT];
    // bias gradient db[m] = sum over pixels of dy[pixel][m]: the dy tile passes through this
    // thread's staging registers anyway, so the N-tile-0 workgroups keep a running column sum.
    const bool do_bias = p.db != nullptr && (tile % p.tiles_n) == 0;
    f32x4 bsum[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) bsum[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto fetch = [&](int kt) __attribute__((always_inline)) {
        const int pb = kt * BK;    // chunk-relative
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            if ((BK * MVA) % NT != 0 && tid + i * NT >= BK * MVA) continue;
            const int pi = pb + a_kr[i];
            ra[i] = *reinterpret_cast<const f32x4*>(p.dy + (size_t)pinfo[pi].w * p.Co + a_m[i]);
            za[i] = pbeg + pi < pend ? 1.f : 0.f;     // pixels past the chunk contribute nothing (zeroing A is enough); applied in stage()
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            if ((BK * NVB) % NT != 0 && tid + i * NT >= BK * NVB) continue;
            if (ST) {
                rb[i] = *reinterpret_cast<const f32x4*>(p.x + (size_t)pinfo[pb + b_kr[i]].x * p.Ci + b_ci[i]);
            } else {
                const int4 pi = pinfo[pb + b_kr[i]];
                const int iy = refl(pi.x + b_ky[i], p.Hu) >> p.up;
                const int ix = refl(pi.y + b_kx[i], p.Wu) >> p.up;
                rb[i] = *reinterpret_cast<const f32x4*>(p.x + (size_t)(pi.z + iy * p.Wi + ix) * p.Ci + b_ci[i]);
            }
        }
    };
    auto stage = [&](int buf, bool real) __attribute__((always_inline)) {
        float* a = As + buf * BK * LDA;
        float* b = Bs + buf * BK * LDB;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int idx = tid + i * NT;
            if ((BK * MVA) % NT == 0 || idx < BK * MVA) {
                const f32x4 v = ra[i] * za[i];
                *reinterpret_cast<f32x4*>(a + a_kr[i] * LDA + (idx - a_kr[i] * MVA) * 4) = v;
                bsum[i] += v * (real ? 1.f : 0.f);   // the clamped tail re-stages the last tile: count it once
            }
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int idx = tid + i * NT;
            if ((BK * NVB) % NT == 0 || idx < BK * NVB) *reinterpret_cast<f32x4*>(b + b_kr[i] * LDB + (idx - b_kr[i] * NVB) * 4) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    ACL_GEMM_MAINLOOP(TM, TN, false, false, 0, (pend - pbeg + BK - 1) / BK, As, Bs, BK * LDA, BK * LDB, LDA, LDB, wm * TM * 32, wn * TN * 32);

    if (do_bias) {   // block-uniform; the main loop ended with a barrier, LDS is free
        float* red = smem;   // [NT*A_IT/MVA row groups][BM]
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int idx = tid + i * NT;
            if ((BK * MVA) % NT == 0 || idx < BK * MVA) *reinterpret_cast<f32x4*>(red + a_kr[i] * BM + (idx - a_kr[i] * MVA) * 4) = bsum[i];
        }
        __syncthreads();
        if (tid < BM && m0 + tid < p.Co) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < BK; ++r) t += red[r * BM + tid];
            atomicAdd(p.db + (size_t)blockIdx.z * p.db_zs + m0 + tid, t);
        }
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
                if (m < p.Co) {
                    atomicAdd(dwbase + (size_t)m * p.Kn + n, acc[i][j][r]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// wgrad, k-contiguous LDS layout + ordered slices (Cout % 64 == 0, Cin % 64 == 0; the tuned path of all heavy layers).
// What changed relative to conv_wgrad_fast_kernel above (profiles/r01: MFMA pipe 68 % busy, fragments fetched with 64
// ds_read_b32 per k-tile because both operands arrive pixel-major while the MFMA wants 8 consecutive k = pixels per lane):
//   * a staging thread loads 4 pixels x 4 channels (4 coalesced dwordx4 loads), transposes them in registers and writes
//     4 b128 rows [channel][4 pixels]: the LDS tiles are [row][16+4] exactly like the forward kernel's, fragments are
//     b128 reads (16 per k-tile instead of 64 b32), the main loop is the forward's;
//   * rows are stored interleaved (LDS row (ch%4)*(rows/4) + ch/4) so that the 32 lanes of a write hit consecutive rows;
//   * no atomics: slices store their tile + bias sums, wgrad_finish_kernel adds them in order (conv_fast_common.h);
//   * a slice walks its pixel range in sub-chunks of <= 1024 pixels (gather table in LDS), so the slice count is free.
// ------------------------------------------------------------------------------------------
constexpr int WGKC_TARGET = 768;     // workgroups in flight the split plan aims for (3 per CU fit: LDS)

template <int WM, int WN, int TM, int TN>
__global__ void __launch_bounds__(WM * WN * 64) conv_wgrad_kc_kernel(WgFP pk, WgPartX xp) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
    WgFP p = pk;
    if (pk.fsl) {   // batched GEMM slice
        const long long f = blockIdx.y;
        p.x += (pk.fsx_mod ? f % pk.fsx_mod : f) * pk.fs_x; p.dy += f * pk.fs_dy;
    }
    static_assert(BM % 64 == 0 && BN % 64 == 0 && BM + BN <= NT, "one staging unit (4 channels x 4 pixels) per thread, wave-uniform roles");
    constexpr int CH = 1024;                   // pixels per sub-chunk
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDK];
    __shared__ int2 pinfo[CH + BK];            // per sub-chunk pixel: .x = source pixel of this workgroup's tap, .y = dy pixel
    float* As = smem;
    float* Bs = smem + 2 * BM * LDK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile = xcd_map(blockIdx.x, p.nwg);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const int pbeg = blockIdx.z * p.chunk;
    const int pend = min(p.P, pbeg + p.chunk);
    const int phase = (p.phases || p.fsl) ? (int)blockIdx.y : 0;
    float* dwbase = p.dw + (size_t)phase * p.Co * p.Kn;
    const int py = phase >> 1, px = phase & 1;
    const int st_tap = n0 / p.Ci, st_ky = st_tap / p.k, st_kx = st_tap - st_ky * p.k;   // the whole N tile lies inside ONE filter tap

    const bool isA = tid < BM, isB = !isA && tid < BM + BN;
    const int u = isA ? tid : tid - BM;
    const int ncg = (isA ? BM : BN) >> 2;           // channel groups of the tile
    const int cg = u % ncg, pg = u / ncg;           // pg in 0..3: pixels 4pg .. 4pg+3 of the k-tile
    const float* src = isA ? p.dy : p.x;
    const int cstride = isA ? p.Co : p.Ci;
    const int chan = isA ? m0 + 4 * cg : (n0 - st_tap * p.Ci) + 4 * cg;
    float* mytile = isA ? As : Bs;
    const int tstride = (isA ? BM : BN) * LDK;
    f32x4 rr[4];
    float zm[4];
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
        for (int i = tid; i < ce - cb + BK; i += NT) {     // + BK: the clamped tail tile reads past the end
            int b, oy, ox;
            wg_coord(p, min(cb + i, ce - 1), b, oy, ox);
            const int dyp = p.phases ? (b * p.Hf + 2 * (oy + 1) + py) * p.Wf + 2 * (ox + 1) + px : (b * p.Ho + oy) * p.Wo + ox;
            const int iy = refl(oy * p.s - p.p + st_ky, p.Hu) >> p.up, ix = refl(ox * p.s - p.p + st_kx, p.Wu) >> p.up;
            pinfo[i] = make_int2(b * p.Hi * p.Wi + iy * p.Wi + ix, dyp);
        }
        __syncthreads();

        auto fetch = [&](int kt) __attribute__((always_inline)) {
            const int pb = kt * BK + 4 * pg;    // sub-chunk relative
            if (isA || isB) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int2 pi = pinfo[pb + j];
                    rr[j] = *reinterpret_cast<const f32x4*>(src + (size_t)(isA ? pi.y : pi.x) * cstride + chan);
                    zm[j] = cb + pb + j < ce ? 1.f : 0.f;     // pixels past the sub-chunk contribute nothing; applied in stage()
                }
            }
        };
        auto stage = [&](int buf, bool real) __attribute__((always_inline)) {
            if (isA || isB) {
                float* t = mytile + buf * tstride;
                f32x4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = rr[j] * zm[j];
                if (isA) {
                    const float on = real ? 1.f : 0.f;      // the clamped tail re-stages the last tile: count it once
#pragma unroll
                    for (int j = 0; j < 4; ++j) bsum += v[j] * on;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 k4 = {v[0][c], v[1][c], v[2][c], v[3][c]};
                    *reinterpret_cast<f32x4*>(t + (c * ncg + cg) * LDK + 4 * pg) = k4;     // channel 4cg+c -> row c*ncg + cg
                }
            }
        };
        ACL_GEMM_MAINLOOP(TM, TN, true, true, 0, (ce - cb + BK - 1) / BK, As, Bs, BM * LDK, BN * LDK, LDK, LDK, wm * TM * 32, wn * TN * 32);
    }

    const bool partial = xp.part != nullptr;
    if (do_bias) {   // block-uniform; the main loop ended with a barrier, LDS is free
        float* red = smem;   // [4 pixel groups][BM]
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
        const int cr = wn * TN * 32 + j * 32 + l31;
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

// ACLGAN_BIGTILE=1: 256 x 128 workgroup tiles with 8 waves for forward / dgrad of the large layers (25 % fewer operand bytes per
// MFMA than 128 x 128; one workgroup per CU)
bool big_tiles() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("ACLGAN_BIGTILE"); v = (e && atoi(e)) ? 1 : 0; }
    return v == 1;
}

bool wgrad_kc_ok(const ConvGeom& g) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("ACLGAN_NOWGKC"); off = (e && atoi(e)) ? 1 : 0; }
    return !off && g.Co % 64 == 0 && g.Ci % 64 == 0;
}

template <int WM, int WN, int TM, int TN>
int launch_wgrad_kc(const ConvGeom& g, WgFP p, void* part, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int ny = p.fsl ? p.fsl : (p.phases ? 4 : 1);
    const WgPlan q = wgrad_plan(g.Co, p.Ci, p.Kn, p.P, ny, BK, WGKC_TARGET);
    if (q.BM != BM || q.BN != BN) { set_error("wgrad_kc: tile plan mismatch"); return ACLGAN_EINVAL; }
    p.tiles_n = q.tiles_n; p.nwg = q.nwg; p.chunk = q.chunk;
    WgPartX xp;
    xp.ny = ny; xp.part = nullptr; xp.part_b = nullptr;
    const bool use_part = q.splits > 1 || ny > 1;
    if (use_part) {
        if (!part) { set_error("wgrad_kc: %d slices need the scratch buffer", q.splits); return ACLGAN_EINVAL; }
        xp.part = (float*)part;
        xp.part_b = xp.part + (size_t)q.splits * ny * q.nwg * BM * BN;
    }
    hipLaunchKernelGGL((conv_wgrad_kc_kernel<WM, WN, TM, TN>), dim3(p.nwg, ny, q.splits), dim3(WM * WN * 64), 0, st, p, xp);
    ACL_CHECK_LAUNCH("conv_wgrad_kc_kernel");
    if (use_part) {
        const int64_t n = (int64_t)ny * g.Co * (p.Kn / 4);
        hipLaunchKernelGGL(wgrad_finish_kernel, dim3((int)std::min<int64_t>(cdiv64(n, 256), 4096)), dim3(256), 0, st, p, xp, BM, BN, q.splits);
        ACL_CHECK_LAUNCH("wgrad_finish_kernel");
    }
    return ACLGAN_OK;
}
int launch_wgrad_kc_any(const ConvGeom& g, const WgFP& p, void* part, hipStream_t st) {
    if (g.Co % 128 == 0 && p.Ci % 128 == 0) return launch_wgrad_kc<2, 2, 2, 2>(g, p, part, st);   // 128 x 128
    if (g.Co % 128 == 0) return launch_wgrad_kc<2, 2, 2, 1>(g, p, part, st);                       // 128 x 64 (Cin = 64)
    if (p.Ci % 128 == 0) return launch_wgrad_kc<2, 2, 1, 2>(g, p, part, st);                       // 64 x 128 (Cout = 64)
    return launch_wgrad_kc<2, 2, 1, 1>(g, p, part, st);                                            // 64 x 64
}

template <int WM, int WN, int TM, int TN>
int launch_wgrad_fast(const ConvGeom& g, WgFP p, hipStream_t st, void* det_part = nullptr) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    p.tiles_n = cdiv(p.Kn, BN);
    p.nwg = cdiv(g.Co, BM) * p.tiles_n;
    const int target = 1536;   // ~6 workgroups per CU in flight: measured +5..17 % over 768 on the heavy layers (latency hiding)
    const int ny = p.phases ? 4 : 1;
    // floor, not ceil: 3 workgroups fit per CU (LDS), so `target` = 2 full rounds of 768; one workgroup more
    // would add a third, nearly empty round (measured: 1548 workgroups ran 33 % longer than 1512)
    int splits = max(1, target / (p.nwg * ny));
    splits = max(1, min(splits, cdiv(p.P, 256)));
    splits = max(splits, cdiv(p.P, 1024));                // the per-chunk pixel table lives in LDS (WG_MAX_CHUNK)
    p.chunk = cdiv(cdiv(p.P, splits), 16) * 16;
    splits = cdiv(p.P, p.chunk);
    // deterministic mode: every pixel slice accumulates into its own zeroed copy of (dw, db), the copies are added in order
    float* dw_out = p.dw; float* db_out = p.db;
    const int64_t ndw = (int64_t)ny * g.Co * p.Kn;
    if (det_part != nullptr && splits > 1) {
        hipError_t e = hipMemsetAsync(det_part, 0, (size_t)splits * (ndw + g.Co) * sizeof(float), st);
        if (e != hipSuccess) return hip_fail(e, "memset wgrad slices");
        p.dw = (float*)det_part; p.dw_zs = ndw;
        if (p.db) { p.db = (float*)det_part + (size_t)splits * ndw; p.db_zs = g.Co; }
    }
    static int nost = -1;
    if (nost < 0) { const char* e = getenv("ACLGAN_NOSINGLETAP"); nost = (e && atoi(e)) ? 1 : 0; }
    if (!nost && p.Ci % BN == 0) hipLaunchKernelGGL((conv_wgrad_fast_kernel<WM, WN, TM, TN, true>), dim3(p.nwg, ny, splits), dim3(WM * WN * 64), 0, st, p);
    else hipLaunchKernelGGL((conv_wgrad_fast_kernel<WM, WN, TM, TN, false>), dim3(p.nwg, ny, splits), dim3(WM * WN * 64), 0, st, p);
    ACL_CHECK_LAUNCH("conv_wgrad_fast_kernel");
    if (p.dw_zs) {
        int rc = reduce_slices_ordered(p.dw, ndw, splits, dw_out, st);
        if (rc) return rc;
        if (db_out) { rc = reduce_slices_ordered(p.db, g.Co, splits, db_out, st); if (rc) return rc; }
    }
    return ACLGAN_OK;
}
// scratch of the deterministic mode of launch_wgrad_fast (upper bound over the tile shapes: the slice count only shrinks with them)
size_t wgrad_fast_det_bytes(const ConvGeom& g, int Kn, int P, int ny) {
    const int nwg = cdiv(g.Co, 128) * cdiv(Kn, 128);
    int splits = std::max(1, 1536 / (nwg * ny));
    splits = std::max(1, std::min(splits, cdiv(P, 256)));
    splits = std::max(splits, cdiv(P, 1024)) + 1;
    return (size_t)splits * ((size_t)ny * g.Co * Kn + g.Co) * sizeof(float) + 256;
}

// ------------------------------------------------------------------------------------------
// sub-pixel decomposition of "nn.Upsample(2) + ReflectionPad2d(2) + Conv2d(5x5)" (reference
// networks.py:256-257, the two decoder layers that hold 57 % of a decode's MACs).
// For an output pixel (2y'+py, 2x'+px) whose 5x5 window touches no reflected row/column, tap ky reads
// low-res row y' + ((py+ky-2) >> 1): the 5 taps collapse onto 3 source rows (py=0: {0,1},{2,3},{4};
// py=1: {0},{1,2},{3,4}), so each of the 4 output phases is a 3x3 VALID conv of the low-res input with
// pre-summed weights: 9 instead of 25 taps per output pixel (2.78x fewer MACs), and the 4x-upsampled
// tensor is never formed.  The output ring of width 2 (where reflection breaks the pattern) keeps the
// exact gather kernels.  Backward uses the same split (conv_up5_dgrad / conv_up5_wgrad).
// ------------------------------------------------------------------------------------------

template <int WM, int WN, int TM, int TN>
int up5_fwd_t(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, float* wp, hipStream_t st, float* keepV) {
    const int64_t nm = (int64_t)4 * g.Co * 9 * (g.Ci / 4);
    if (!(conv_up5_wino_ok(g) && wino_u_cached(w, conv_up5_wino_u_variant(g, 0, keepV != nullptr)))) {      // (the Winograd phases read only the cached transform of the merged filters)
        hipLaunchKernelGGL(up5_merge_kernel, dim3((int)std::min<int64_t>(cdiv64(nm, 256), 2048)), dim3(256), 0, st, w, wp, g.Co, g.Ci);
        ACL_CHECK_LAUNCH("up5_merge_kernel");
    }
    FwdFP p;
    p.fsl = 0; p.fsx_mod = 0; p.fs_x = p.fs_w = p.fs_y = 0; p.w16 = nullptr; p.x16 = nullptr;
    // (1) the four phases: valid 3x3 conv on the low-res input, scattered into the 2H x 2W output
    p.x = x; p.w = wp; p.bias = bias; p.y = y;
    p.Hi = g.Hi; p.Wi = g.Wi; p.Ci = g.Ci; p.Ho = g.Hi - 2; p.Wo = g.Wi - 2; p.Co = g.Co; p.k = 3; p.s = 1; p.p = 0;
    p.up = 0; p.Hu = g.Hi; p.Wu = g.Wi; p.M = g.B * p.Ho * p.Wo; p.K = 9 * g.Ci; p.act = g.act; p.tiles_n = 0; p.nwg = 0; p.nkz = 0;
    p.B = g.B; p.ring = 0; p.phases = 1; p.Hf = g.Ho; p.Wf = g.Wo; p.part = nullptr; p.rows = 0;
    ConvGeom gp = g;
    gp.M = p.M; gp.K = p.K;
    // the phases are VALID 3x3 convolutions: Winograd F(4x4,3x3), all four in one batched GEMM (conv_wino.hip); scratch follows
    // the merged filters and the ring partials
    char* wino_scr = (char*)wp + up5_merged_bytes(g) + ((fwd_partial_bytes(g, 2, BK) + 255) & ~(size_t)255);
    if (keepV && !conv_up5_wino_ok(g)) { set_error("conv_fwd: this layer does not keep a Winograd input transform"); return ACLGAN_EINVAL; }
    int rc = conv_up5_wino_ok(g) ? conv_up5_wino_fwd_phases(g, x, wp, bias, y, wino_scr, st, keepV, w) : launch_fwd_fast<WM, WN, TM, TN>(gp, p, st);
    if (rc) return rc;
    // (2) the output ring of width 2: exact gather (reflection at the borders of the upsampled image)
    p.w = w;
    p.Hi = g.Hi; p.Wi = g.Wi; p.Ho = g.Ho; p.Wo = g.Wo; p.k = 5; p.s = 1; p.p = 2; p.up = 1; p.Hu = g.Hu; p.Wu = g.Wu;
    p.M = g.M; p.K = g.K; p.ring = 2; p.phases = 0;
    p.part = (float*)((char*)wp + up5_merged_bytes(g));   // the partial buffer follows the merged phase weights
    return launch_fwd_fast<WM, WN, TM, TN>(g, p, st);
}

template <int WM, int WN, int TM, int TN>
int up5_wgrad_t(const ConvGeom& g, const float* x, const float* dy, float* dw, float* db, float* dwp, hipStream_t st, const float* haveV) {
    // (1) phase weight gradients: valid 3x3 wgrad on the low-res input against the strided views of dy
    const size_t nb = (size_t)4 * g.Co * 9 * g.Ci * sizeof(float);
    hipError_t e = hipMemsetAsync(dwp, 0, nb, st);
    if (e != hipSuccess) return hip_fail(e, "memset dwp");
    const bool kc = wgrad_kc_ok(g);
    void* part = (char*)dwp + up5_dwp_bytes(g);      // partial tiles of the ordered-slice kernels follow the phase gradients
    WgFP p;
    p.fsl = 0; p.fsx_mod = 0; p.fs_x = p.fs_dy = 0;
    p.x = x; p.dy = dy; p.dw = dwp; p.db = db;
    p.Hi = g.Hi; p.Wi = g.Wi; p.Ci = g.Ci; p.Ho = g.Hi - 2; p.Wo = g.Wi - 2; p.Co = g.Co; p.k = 3; p.s = 1; p.p = 0;
    p.up = 0; p.Hu = g.Hi; p.Wu = g.Wi; p.P = g.B * p.Ho * p.Wo; p.Kn = 9 * g.Ci; p.chunk = 0; p.tiles_n = 0; p.nwg = 0;
    p.B = g.B; p.ring = 0; p.phases = 1; p.Hf = g.Ho; p.Wf = g.Wo;
    int rc;
    if (kc && conv_up5_wino_wgrad_scratch_bytes(g)) {     // Winograd: the four phase gradients in one batched A^T B GEMM
        void* wino_scr = (char*)dwp + ((wgrad_part_scratch(g, BK, WGKC_TARGET) + 255) & ~(size_t)255);
        rc = conv_up5_wino_wgrad_phases(g, x, dy, dwp, db, wino_scr, st, haveV);
    } else rc = kc ? launch_wgrad_kc_any(g, p, part, st) : launch_wgrad_fast<WM, WN, TM, TN>(g, p, st);
    if (rc) return rc;
    // (2) fold the 4 x 3x3 phase gradients back onto the 5x5 filter
    const int64_t ns = (int64_t)g.Co * 25 * (g.Ci / 4);
    hipLaunchKernelGGL(up5_scatter_kernel, dim3((int)std::min<int64_t>(cdiv64(ns, 256), 2048)), dim3(256), 0, st, dwp, dw, g.Co, g.Ci);
    ACL_CHECK_LAUNCH("up5_scatter_kernel");
    // (3) the output ring: exact 5x5 gather wgrad restricted to the ring pixels
    p.dw = dw;
    p.Hi = g.Hi; p.Wi = g.Wi; p.Ho = g.Ho; p.Wo = g.Wo; p.k = 5; p.s = 1; p.p = 2; p.up = 1; p.Hu = g.Hu; p.Wu = g.Wu;
    p.ring = 2; p.phases = 0; p.Kn = g.K;
    p.P = g.B * (g.Ho * g.Wo - (g.Ho - 4) * (g.Wo - 4));
    return kc ? launch_wgrad_kc_any(g, p, part, st) : launch_wgrad_fast<WM, WN, TM, TN>(g, p, st);
}

// ACLGAN_UP5_BANDFOLD=1: the ring of the sub-pixel input gradient as plain stores on the padded hi-res grid + a gather over the dx pixels that
// alias into the band, instead of fp32 atomics.  MEASURED SLOWER (round 6, profiles/r06_experiments.md): the step 82.9 against 82.3 ms (3 lanes),
// 90.5 against 90.1 (one queue) -- the ring launch is bound by its k loop and its tail, not by the atomics; the extra launch and the 25 MB
// round trip cost more than the atomics did.  Off; kept as a measured option (the deterministic mode uses the full-grid fold as before).
bool up5_band_fold() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("ACLGAN_UP5_BANDFOLD"); v = (e && atoi(e)) ? 1 : 0; }
    return v == 1;
}
template <int WM, int WN, int TM, int TN>
int up5_dgrad_t(const ConvGeom& g, const float* dy, const float* w, float* dx, int accumulate, float* wp, hipStream_t st) {
    const int64_t nm = (int64_t)4 * g.Co * 9 * (g.Ci / 4);
    if (!(conv_up5_wino_ok(g) && wino_u_cached(w, conv_up5_wino_u_variant(g, 1, false)))) {
        hipLaunchKernelGGL(up5_merge_kernel, dim3((int)std::min<int64_t>(cdiv64(nm, 256), 2048)), dim3(256), 0, st, w, wp, g.Co, g.Ci);
        ACL_CHECK_LAUNCH("up5_merge_kernel");
    }
    // (1) four phases: dgrad of the VALID 3x3 conv on the low-res grid, read through the phase view of dy,
    //     accumulated straight into dx (plain read-modify-write: the launches are stream-ordered)
    ConvGeom gv = g;
    gv.k = 3; gv.s = 1; gv.p = 0; gv.up = 0; gv.Hu = g.Hi; gv.Wu = g.Wi; gv.Hp = g.Hi; gv.Wp = g.Wi;
    gv.Ho = g.Hi - 2; gv.Wo = g.Wi - 2; gv.M = g.B * gv.Ho * gv.Wo; gv.K = 9 * g.Ci;
    DgFP p;
    p.dy = dy; p.dxp = dx;
    p.Ho = gv.Ho; p.Wo = gv.Wo; p.Co = g.Co; p.Ci = g.Ci; p.k = 3; p.s = 1; p.Hp = g.Hi; p.Wp = g.Wi;
    p.Hc = g.Hi; p.Wc = g.Wi; p.Mc = 0; p.tiles_n = 0; p.nwg = 0; p.ksplit = 1;
    p.mode = 1; p.pad = 0; p.B = g.B; p.Hi = g.Hi; p.Wi = g.Wi;
    p.dyv = 1; p.Hf = g.Ho; p.Wf = g.Wo; p.band = 0; p.upshift = 0; p.Hd = g.Hi; p.Wd = g.Wi;
    if (conv_up5_wino_ok(g)) {       // Winograd: full correlation of the four phase views of dy with the flipped merged filters
        const int rc = conv_up5_wino_dgrad_phases(g, dy, wp, dx, accumulate, (char*)wp + up5_merged_bytes(g), st, w);
        if (rc) return rc;
    } else for (int ph = 0; ph < 4; ++ph) {
        p.w = wp + (size_t)ph * g.Co * 9 * g.Ci;
        p.py = ph >> 1; p.px = ph & 1;
        p.accumulate = (ph > 0 || accumulate) ? 1 : 0;
        const int rc = launch_dgrad_fast<WM, WN, TM, TN>(gv, p, st);
        if (rc) return rc;
    }
    // (2) contributions of the output ring (width 2): exact taps on the band of the padded/upsampled grid that
    //     can see ring outputs (6 rows/columns on each side), folded into dx with atomics
    p.w = w;
    p.Ho = g.Ho; p.Wo = g.Wo; p.k = 5; p.s = 1; p.Hp = g.Hp; p.Wp = g.Wp; p.Hc = g.Hp; p.Wc = g.Wp;
    p.mode = 2; p.accumulate = 1; p.pad = 2; p.Hi = g.Hu; p.Wi = g.Wu;
    p.dyv = 0; p.band = 6; p.upshift = 1; p.Hd = g.Hi; p.Wd = g.Wi;
    if (deterministic()) {     // band positions stored on a zeroed padded hi-res grid (after the merged filters), then the fold gather
        float* dxp = (float*)((char*)wp + up5_merged_bytes(g));
        hipError_t e = hipMemsetAsync(dxp, 0, (size_t)g.B * g.Hp * g.Wp * g.Ci * sizeof(float), st);
        if (e != hipSuccess) return hip_fail(e, "memset padded grid");
        p.ringpad = 1; p.dxp = dxp; p.accumulate = 0;
        const int rc = launch_dgrad_fast<WM, WN, TM, TN>(g, p, st);
        if (rc) return rc;
        return conv_fold(g, dxp, dx, 1, st);
    }
    if (up5_band_fold() && g.Ci % 4 == 0) {
        // Round 6: one plain store per band position into the (otherwise untouched) padded hi-res scratch, then a gather over the dx pixels
        // that alias into the band -- instead of 6.2 M fp32 atomics per launch (256 -> 128 layer at 256x256 B=8: 445 us for 13 GFLOP)
        float* dxp = (float*)((char*)wp + up5_merged_bytes(g));
        p.ringpad = 1; p.dxp = dxp; p.accumulate = 0;
        const int rc = launch_dgrad_fast<WM, WN, TM, TN>(g, p, st);
        if (rc) return rc;
        return conv_fold_band(g, dxp, dx, 6, st);
    }
    return launch_dgrad_fast<WM, WN, TM, TN>(g, p, st);
}


}  // namespace

size_t conv_up5_scratch_bytes(const ConvGeom& g) {
    return up5_eligible(g) ? (size_t)4 * g.Co * 9 * g.Ci * sizeof(float) : 0;
}
// dgrad of a sub-pixel layer: merged phase filters, then the Winograd planes of its four phases
size_t conv_up5_dgrad_scratch_bytes(const ConvGeom& g) {
    if (!fast_enabled() || !up5_eligible(g)) return 0;
    const size_t padded = (deterministic() || up5_band_fold()) ? (size_t)g.B * g.Hp * g.Wp * g.Ci * sizeof(float) : 0;     // ring positions on the padded hi-res grid (ordered fold / band fold)
    return up5_merged_bytes(g) + std::max(conv_up5_wino_dgrad_scratch_bytes(g), padded);
}
// weight-gradient scratch of the tuned kernels: phase gradients of the sub-pixel layers + the partial tiles of the
// ordered-slice kernel (0 when neither applies)
size_t conv_wgrad_fast_scratch_bytes(const ConvGeom& g) {
    if (!fast_enabled()) return 0;
    if (wgrad_kc_ok(g) && conv_wino_ok(g)) return conv_wgrad_wino_scratch_bytes(g);
    if (wgrad_kc_ok(g)) return ((wgrad_part_scratch(g, BK, WGKC_TARGET) + 255) & ~(size_t)255) + (up5_eligible(g) ? conv_up5_wino_wgrad_scratch_bytes(g) : 0);
    if (deterministic()) return wgrad_fast_det_bytes(g, g.K, g.M, 1);      // atomics kernel with per-slice copies (the sub-pixel split is skipped)
    return conv_up5_scratch_bytes(g);
}
// forward scratch: merged phase weights + ring split-K partials (sub-pixel layers), or the split-K partials of a small-grid layer
size_t conv_fwd_fast_scratch_bytes(const ConvGeom& g) {
    if (!fast_enabled()) return 0;
    if (conv_wino_ok(g)) return conv_wino_scratch_bytes(g);
    if (up5_eligible(g)) return up5_merged_bytes(g) + ((fwd_partial_bytes(g, 2, BK) + 255) & ~(size_t)255) + conv_up5_wino_fwd_scratch_bytes(g);
    return std::max(fwd_partial_bytes(g, 0, BK), conv_s2k4_wino_scratch_bytes(g));
}

int conv_up5_fwd(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, void* scratch, hipStream_t st, float* keepV) {
    if (!fast_enabled() || !up5_eligible(g) || !scratch) return ACLGAN_EUNSUPPORTED;
    if (g.Co > 64) return up5_fwd_t<2, 2, 2, 2>(g, x, w, bias, y, (float*)scratch, st, keepV);
    if (g.Co > 32) return up5_fwd_t<4, 1, 2, 2>(g, x, w, bias, y, (float*)scratch, st, keepV);
    return up5_fwd_t<4, 1, 2, 1>(g, x, w, bias, y, (float*)scratch, st, keepV);
}

// needs conv_up5_scratch_bytes(g) of scratch; dx complete on return (no fold kernel)
int conv_up5_dgrad(const ConvGeom& g, const float* dy, const float* w, float* dx, int accumulate, void* scratch, hipStream_t st) {
    if (!fast_enabled() || !up5_eligible(g) || !scratch) return ACLGAN_EUNSUPPORTED;
    static int off = -1;
    if (off < 0) { const char* e = getenv("ACLGAN_NOUP5DGRAD"); off = (e && atoi(e)) ? 1 : 0; }
    if (off) return ACLGAN_EUNSUPPORTED;
    if (g.Ci > 64) return up5_dgrad_t<2, 2, 2, 2>(g, dy, w, dx, accumulate, (float*)scratch, st);
    if (g.Ci > 32) return up5_dgrad_t<4, 1, 2, 2>(g, dy, w, dx, accumulate, (float*)scratch, st);
    return up5_dgrad_t<4, 1, 2, 1>(g, dy, w, dx, accumulate, (float*)scratch, st);
}

int conv_up5_wgrad(const ConvGeom& g, const float* x, const float* dy, float* dw, float* db, void* scratch, hipStream_t st, const float* haveV) {
    if (!fast_enabled() || !up5_eligible(g) || !scratch || !dw) return ACLGAN_EUNSUPPORTED;
    if (deterministic() && !wgrad_kc_ok(g)) return ACLGAN_EUNSUPPORTED;    // (the phase / ring launches of the atomics kernel share dw)
    if (g.Co > 64) return up5_wgrad_t<2, 2, 2, 2>(g, x, dy, dw, db, (float*)scratch, st, haveV);
    if (g.Co > 32) return up5_wgrad_t<2, 2, 1, 2>(g, x, dy, dw, db, (float*)scratch, st, haveV);
    return up5_wgrad_t<1, 4, 1, 2>(g, x, dy, dw, db, (float*)scratch, st, haveV);
}

// C_f[T][N] = A_f[T][K] x B_f[N][K]^T for f = 0 .. nslices-1 (fp32, exact MFMA): the tuned forward kernel run as a 1x1 "conv" over a
// T x 1 "image", slice f on blockIdx.y.  K % 16 == 0.  Used by the Winograd path (conv_wino.hip).
int gemm_slices_f32(const float* A, const float* Bm, float* Cm, int T, int K, int N, int nslices, int a_mod, hipStream_t st) {
    if (K % 16 != 0 || T <= 0 || N <= 0) { set_error("gemm_slices_f32: bad shape"); return ACLGAN_EINVAL; }
    FwdFP p;
    p.part = nullptr; p.rows = 0; p.w16 = nullptr; p.x16 = nullptr;
    p.x = A; p.w = Bm; p.bias = nullptr; p.y = Cm;
    p.Hi = T; p.Wi = 1; p.Ci = K; p.Ho = T; p.Wo = 1; p.Co = N; p.k = 1; p.s = 1; p.p = 0;
    p.up = 0; p.Hu = T; p.Wu = 1; p.M = T; p.K = K; p.act = ACLGAN_ACT_NONE; p.tiles_n = 0; p.nwg = 0; p.nkz = K / BK;
    p.B = 1; p.ring = 0; p.phases = 0; p.Hf = 0; p.Wf = 0;
    p.fsl = nslices; p.fsx_mod = a_mod; p.fs_x = (long long)T * K; p.fs_w = (long long)N * K; p.fs_y = (long long)T * N;
    if (N > 64) {
        p.tiles_n = cdiv(N, 128); p.nwg = cdiv(T, 128) * p.tiles_n;
        // K is short here (the channel count): 128 x 128 tiles leave 1.5 rounds of workgroups and exposed prologues/epilogues;
        // 64 x 128 tiles at 4 workgroups per CU measured 8 % faster on the ResBlock shape (profiles/r02_experiments.md).
        // ACLGAN_GEMM_VAR=1: 128 x 64 tiles, 3 per CU;  =2: 128 x 128 tiles, 3 per CU.
        static int var = -1;
        if (var < 0) { const char* e = getenv("ACLGAN_GEMM_VAR"); var = e ? atoi(e) : 0; }
        if (var == 1) {
            p.tiles_n = cdiv(N, 64); p.nwg = cdiv(T, 128) * p.tiles_n;
            hipLaunchKernelGGL((conv_fwd_fast_kernel<2, 2, 2, 1, 3>), dim3(p.nwg, nslices, 1), dim3(256), 0, st, p);
        } else if (var == 2) {
            hipLaunchKernelGGL((conv_fwd_fast_kernel<2, 2, 2, 2, 3>), dim3(p.nwg, nslices, 1), dim3(256), 0, st, p);
        } else if (var == 3 && N % 256 == 0) {      // 64 x 256 tiles: a workgroup covers all of N = 256, V is fetched once per row tile
            p.tiles_n = N / 256; p.nwg = cdiv(T, 64) * p.tiles_n;
            hipLaunchKernelGGL((conv_fwd_fast_kernel<1, 4, 2, 2, 3>), dim3(p.nwg, nslices, 1), dim3(256), 0, st, p);
        } else {
            p.tiles_n = cdiv(N, 128); p.nwg = cdiv(T, 64) * p.tiles_n;
            hipLaunchKernelGGL((conv_fwd_fast_kernel<2, 2, 1, 2, 4>), dim3(p.nwg, nslices, 1), dim3(256), 0, st, p);
        }
    } else if (N > 32) {
        p.tiles_n = cdiv(N, 64); p.nwg = cdiv(T, 256) * p.tiles_n;
        hipLaunchKernelGGL((conv_fwd_fast_kernel<4, 1, 2, 2>), dim3(p.nwg, nslices, 1), dim3(256), 0, st, p);
    } else {
        p.tiles_n = cdiv(N, 32); p.nwg = cdiv(T, 256) * p.tiles_n;
        hipLaunchKernelGGL((conv_fwd_fast_kernel<4, 1, 2, 1>), dim3(p.nwg, nslices, 1), dim3(256), 0, st, p);
    }
    ACL_CHECK_LAUNCH("conv_fwd_fast_kernel(gemm slices)");
    return ACLGAN_OK;
}

// returns ACLGAN_EUNSUPPORTED when the shape is not eligible (caller falls back to the general kernel)
// the Winograd output transform holds whole 4x4 output tiles per thread: it can emit the (mean, M2) partials of the following
// normalisation layer at no extra pass over y.  ACLGAN_NOSTATFUSE=1 keeps the separate statistics kernel.
int conv_fwd_stats_chunk(const ConvGeom& g) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("ACLGAN_NOSTATFUSE"); off = (e && atoi(e)) ? 1 : 0; }
    if (off || !fast_enabled() || g.Ci % 16 != 0 || g.act != ACLGAN_ACT_NONE) return 0;
    if (conv_wino_ok(g)) return 16;
    return (g.Ho % 4 == 0 && g.Wo % 4 == 0 && conv_s2k4_wino_ok(g, 0)) ? 16 : 0;      // (round 6: the 4x4 stride-2 layers through the fused kernel)
}
// bytes of the Winograd input transform conv_fwd can leave behind for conv_wgrad (same path selection as conv_fwd / conv_wgrad)
size_t conv_fwd_keep_bytes(const ConvGeom& g) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("ACLGAN_NOKEEPV"); off = (e && atoi(e)) ? 1 : 0; }
    if (off || !fast_enabled() || !wgrad_kc_ok(g)) return 0;
    if (up5_eligible(g)) return conv_up5_wino_wgrad_scratch_bytes(g) ? conv_up5_wino_keep_bytes(g) : 0;
    if (g.Ci % 16 != 0 || !conv_wino_ok(g) || !conv_wgrad_wino_scratch_bytes(g)) return 0;
    return conv_wino_keep_bytes(g);
}
int conv_fwd_fast(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, hipStream_t st, void* scratch, float* stats, float* keepV) {
    if (!fast_enabled() || g.Ci % 16 != 0) return ACLGAN_EUNSUPPORTED;
    if (scratch && conv_wino_ok(g)) return conv_fwd_wino(g, x, w, bias, y, scratch, st, stats, keepV);   // 3x3 ResBlock convs: Winograd F(4x4,3x3)
    if (scratch && !keepV && conv_s2k4_wino_ok(g, 0) && (!stats || (g.Ho % 4 == 0 && g.Wo % 4 == 0)))      // 4x4 stride-2 layers: four parity phases of the same kernel
        return conv_fwd_s2k4_wino(g, x, w, bias, y, scratch, st, stats);
    if (stats || keepV) return ACLGAN_EUNSUPPORTED;
    FwdFP p;
    p.fsl = 0; p.fsx_mod = 0; p.fs_x = p.fs_w = p.fs_y = 0; p.w16 = nullptr; p.x16 = nullptr;
    p.part = (float*)scratch; p.rows = 0;
    p.x = x; p.w = w; p.bias = bias; p.y = y;
    p.Hi = g.Hi; p.Wi = g.Wi; p.Ci = g.Ci; p.Ho = g.Ho; p.Wo = g.Wo; p.Co = g.Co; p.k = g.k; p.s = g.s; p.p = g.p;
    p.up = g.up; p.Hu = g.Hu; p.Wu = g.Wu; p.M = g.M; p.K = g.K; p.act = g.act; p.tiles_n = 0; p.nwg = 0; p.nkz = 0;
    p.B = g.B; p.ring = 0; p.phases = 0; p.Hf = 0; p.Wf = 0;
    if (g.Co % 128 == 0 && big_tiles() && g.M >= 256 * 128) return launch_fwd_fast<4, 2, 2, 2>(g, p, st);   // 256 x 128, 8 waves (experiment)
    if (g.Co > 64) return launch_fwd_fast<2, 2, 2, 2>(g, p, st);
    if (g.Co > 32) return launch_fwd_fast<4, 1, 2, 2>(g, p, st);
    return launch_fwd_fast<4, 1, 2, 1>(g, p, st);
}

// dxp: scratch for the padded-grid path; dx/accumulate: final destination.  *direct = true when dx has
// been fully produced here (no fold kernel needed).
int conv_dgrad_fast(const ConvGeom& g, const float* dy, const float* w, float* dxp, float* dx, int accumulate, bool* direct, hipStream_t st) {
    *direct = false;
    if (!fast_enabled() || g.Co % 16 != 0 || g.Ci % 4 != 0) return ACLGAN_EUNSUPPORTED;
    static int nodirect = -1;
    if (nodirect < 0) { const char* e = getenv("ACLGAN_NODIRECT"); nodirect = (e && atoi(e)) ? 1 : 0; }
    if (nodirect) dx = nullptr;
    if (deterministic() && !dxp) { set_error("conv_dgrad: deterministic mode needs the scratch buffer"); return ACLGAN_EINVAL; }
    DgFP p;
    p.dy = dy; p.w = w; p.dxp = dxp;
    p.Ho = g.Ho; p.Wo = g.Wo; p.Co = g.Co; p.Ci = g.Ci; p.k = g.k; p.s = g.s; p.Hp = g.Hp; p.Wp = g.Wp;
    p.Hc = cdiv(g.Hp, g.s); p.Wc = cdiv(g.Wp, g.s); p.Mc = 0; p.tiles_n = 0; p.nwg = 0; p.ksplit = 1;
    p.mode = 0; p.accumulate = 0; p.pad = g.p; p.B = g.B; p.Hi = g.Hu; p.Wi = g.Wu;
    p.dyv = 0; p.py = 0; p.px = 0; p.Hf = 0; p.Wf = 0; p.band = 0; p.upshift = 0; p.Hd = g.Hu; p.Wd = g.Wu;
    if (g.Ci % 128 == 0 && big_tiles() && g.M >= 256 * 128) return dgrad_fast_all<4, 2, 2, 2>(g, p, dxp, dx, accumulate, direct, st);
    if (g.Ci > 64) return dgrad_fast_all<2, 2, 2, 2>(g, p, dxp, dx, accumulate, direct, st);
    if (g.Ci > 32) return dgrad_fast_all<4, 1, 2, 2>(g, p, dxp, dx, accumulate, direct, st);
    return dgrad_fast_all<4, 1, 2, 1>(g, p, dxp, dx, accumulate, direct, st);
}

// C_f[M][N] = sum_t A_f[t][M] * B_f[t][N] for f = 0 .. nslices-1 (fp32): the ordered-slice weight-gradient kernel run as a 1x1 "conv"
// over a T x 1 "image".  M, N multiples of 64.  part: gemm_at_b_slices_scratch(...) bytes.  Used by the Winograd path.
size_t gemm_at_b_slices_scratch(int T, int M, int N, int nslices) {
    return wgrad_partial_bytes(wgrad_plan(M, N, N, T, nslices, BK, WGKC_TARGET), M, nslices) + 256;
}
int gemm_at_b_slices_f32(const float* A, const float* Bm, float* Cm, int T, int M, int N, int nslices, int b_mod, void* part, hipStream_t st) {
    if (M % 64 != 0 || N % 64 != 0 || T <= 0) { set_error("gemm_at_b_slices_f32: bad shape"); return ACLGAN_EINVAL; }
    WgFP p;
    p.x = Bm; p.dy = A; p.dw = Cm; p.db = nullptr;
    p.Hi = T; p.Wi = 1; p.Ci = N; p.Ho = T; p.Wo = 1; p.Co = M; p.k = 1; p.s = 1; p.p = 0;
    p.up = 0; p.Hu = T; p.Wu = 1; p.P = T; p.Kn = N; p.chunk = 0; p.tiles_n = 0; p.nwg = 0;
    p.B = 1; p.ring = 0; p.phases = 0; p.Hf = 0; p.Wf = 0;
    p.fsl = nslices; p.fsx_mod = b_mod; p.fs_x = (long long)T * N; p.fs_dy = (long long)T * M;
    p.dw_overwrite = 1;          // C = A^T B (the slices always go through the ordered partials + finish): no zero-fill of C needed
    ConvGeom g;
    memset(&g, 0, sizeof g);
    g.Co = M; g.Ci = N;
    return launch_wgrad_kc_any(g, p, part, st);
}

bool conv_wgrad_fast_supported(const ConvGeom& g) { return fast_enabled() && g.Co % 4 == 0 && g.Ci % 4 == 0; }
int conv_wgrad_fast(const ConvGeom& g, const float* x, const float* dy, float* dw, float* db, hipStream_t st, void* scratch, const float* haveV) {
    if (!fast_enabled() || g.Co % 4 != 0 || g.Ci % 4 != 0) return ACLGAN_EUNSUPPORTED;
    WgFP p;
    p.fsl = 0; p.fsx_mod = 0; p.fs_x = p.fs_dy = 0;
    p.x = x; p.dy = dy; p.dw = dw; p.db = db;
    p.Hi = g.Hi; p.Wi = g.Wi; p.Ci = g.Ci; p.Ho = g.Ho; p.Wo = g.Wo; p.Co = g.Co; p.k = g.k; p.s = g.s; p.p = g.p;
    p.up = g.up; p.Hu = g.Hu; p.Wu = g.Wu; p.P = g.M; p.Kn = g.K; p.chunk = 0; p.tiles_n = 0; p.nwg = 0;
    p.B = g.B; p.ring = 0; p.phases = 0; p.Hf = 0; p.Wf = 0;
    if (scratch && conv_wino_ok(g) && wgrad_kc_ok(g)) return conv_wgrad_wino(g, x, dy, dw, db, scratch, st, haveV);   // 3x3 ResBlock convs: Winograd
    // with a scratch buffer (always, inside the engine): k-contiguous tiles + ordered slices, reproducible bit for bit;
    // the scratch-less operator call keeps the atomics kernel
    if (wgrad_kc_ok(g) && (scratch || wgrad_part_scratch(g, BK, WGKC_TARGET) == 0)) return launch_wgrad_kc_any(g, p, scratch, st);
    void* det = nullptr;
    if (deterministic()) {
        if (!scratch) { set_error("conv_wgrad: deterministic mode needs the scratch buffer (aclgan_conv2d_wgrad_ws)"); return ACLGAN_EINVAL; }
        det = scratch;
    }
    if (g.Co > 64) return launch_wgrad_fast<2, 2, 2, 2>(g, p, st, det);
    if (g.Co > 32) return launch_wgrad_fast<2, 2, 1, 2>(g, p, st, det);
    return launch_wgrad_fast<1, 4, 1, 2>(g, p, st, det);
}

// ------------------------------------------------------------------------------------------
// Round 6: matrix-pipe FLOPs the chosen path of a convolution EXECUTES (which: 0 forward, 1 input gradient, 2 weight gradient; f16: the
// 16-bit kernels run it) -- the same decisions as the launchers above, counted per launch the way the hardware counts (whole tiles): the
// model behind roofline.flop_per_launch (aclgan_step_executed_flops), checked against `rocprofv3 --pmc SQ_INSTS_MFMA` (scripts/step_mfma_flops.py).
//   direct kernels            2 M Co K
//   Winograd F(4x4,3x3)       36 multiplies per 4x4 output tile and channel pair instead of 144 (fused kernel: whole blocks of 32 tiles x 64 channels)
//   sub-pixel (up5) layers    four VALID 3x3 phases (Winograd in fp32) + the exact output ring of width 2
//   4x4 stride-2 layers       four parity phases x 36 per tile when the fused kernel takes them (9 / 16 of the direct MACs)
// ------------------------------------------------------------------------------------------
namespace {
double fused_flops(int B, int OH, int OW, int Cin_, int Cout_, int gph, int kph) {      // wino_fused_go's grid x its K loop
    const int TY = cdiv(OH, 4), TX = cdiv(OW, 4);
    const double wgs = (double)B * cdiv(TY, 4) * cdiv(TX, 8) * (Cout_ / 64) * gph;
    return wgs * 2.0 * 36.0 * 32.0 * 64.0 * (double)Cin_ * kph;
}
double pipe_flops(int B, int OH, int OW, int Cin_, int Cout_, int slices) { return 2.0 * 36.0 * slices * (double)B * cdiv(OH, 4) * cdiv(OW, 4) * Cin_ * Cout_; }
}  // namespace
double conv_exec_flops(const ConvGeom& g, int which, bool f16) {
    const double direct = 2.0 * (double)g.M * g.Co * g.K;
    const double ringpix = (double)g.Ho * g.Wo - (double)std::max(0, g.Ho - 4) * std::max(0, g.Wo - 4);      // output ring of width 2 (sub-pixel layers)
    if (up5_eligible(g) && fast_enabled()) {
        const double ring_fwd = 2.0 * g.B * ringpix * g.Co * g.K;
        // input gradient: band of width 6 of the padded hi-res grid, ~9 of the 25 taps useful per position (tap lists, conv_fast_common.h::dg_row)
        const double band = (double)g.Hp * g.Wp - (double)std::max(0, g.Hp - 12) * std::max(0, g.Wp - 12);
        const double ring_dg = 2.0 * g.B * band * g.Ci * g.Co * 9.0;
        if (!f16 && conv_up5_wino_ok(g)) {
            if (which == 0) return (wino_fused_ok(g.B, g.Hi - 2, g.Wi - 2, g.Ci, g.Co, g.act, 4, 1) ? fused_flops(g.B, g.Hi - 2, g.Wi - 2, g.Ci, g.Co, 4, 1)
                                                                                                     : pipe_flops(g.B, g.Hi - 2, g.Wi - 2, g.Ci, g.Co, 4)) + ring_fwd;
            if (which == 1) return (wino_fused_ok(g.B, g.Hi, g.Wi, g.Co, g.Ci, ACLGAN_ACT_NONE, 1, 4) ? fused_flops(g.B, g.Hi, g.Wi, g.Co, g.Ci, 1, 4)
                                                                                                       : pipe_flops(g.B, g.Hi, g.Wi, g.Co, g.Ci, 4)) + ring_dg;
            return (conv_up5_wino_wgrad_scratch_bytes(g) ? pipe_flops(g.B, g.Hi - 2, g.Wi - 2, g.Ci, g.Co, 4) : 4.0 * 2.0 * g.B * (g.Hi - 2.0) * (g.Wi - 2.0) * g.Co * 9.0 * g.Ci) + ring_fwd;
        }
        const double phases = 4.0 * 2.0 * g.B * (g.Hi - 2.0) * (g.Wi - 2.0) * g.Co * 9.0 * g.Ci;
        return phases + (which == 1 ? ring_dg : ring_fwd);
    }
    if (!f16 && fast_enabled() && g.Ci % 16 == 0 && conv_wino_ok(g)) {
        const double halo = 2.0 * g.B * ((double)g.Hp * g.Wp - (double)g.Hi * g.Wi) * g.Ci * g.Co * 3.0;      // ring of the padded grid, 3 of 9 taps per position
        if (which == 0) return wino_fused_ok(g.B, g.Hi, g.Wi, g.Ci, g.Co, g.act) && !conv_fwd_keep_bytes(g) ? fused_flops(g.B, g.Hi, g.Wi, g.Ci, g.Co, 1, 1)
                                                                                                              : pipe_flops(g.B, g.Hi, g.Wi, g.Ci, g.Co, 1);
        if (which == 1) return (wino_fused_ok(g.B, g.Hi, g.Wi, g.Co, g.Ci, ACLGAN_ACT_NONE) ? fused_flops(g.B, g.Hi, g.Wi, g.Co, g.Ci, 1, 1) : pipe_flops(g.B, g.Hi, g.Wi, g.Co, g.Ci, 1)) + halo;
        if (wgrad_kc_ok(g) && conv_wgrad_wino_scratch_bytes(g)) return pipe_flops(g.B, g.Hi, g.Wi, g.Ci, g.Co, 1);
        return direct;
    }
    if (!f16 && fast_enabled() && which < 2 && conv_s2k4_wino_ok(g, which)) {
        if (which == 0) return fused_flops(g.B, g.Ho, g.Wo, g.Ci, g.Co, 1, 4);
        const double halo = 2.0 * g.B * ((double)g.Hp * g.Wp - (double)g.Hi * g.Wi) * g.Ci * g.Co * 2.0;      // 2 of the 4 taps of a parity class per ring position
        return fused_flops(g.B, g.Ho, g.Wo, g.Co, g.Ci, 4, 1) + halo;
    }
    // direct kernels: whole tiles along the narrow dimension.  The input gradient of an image-side layer (Cin 3 / 6: the first discriminator
    // layers) runs on 32-column tiles -- 32 / Cin times the necessary FLOPs; the thin-channel weight gradients (conv_small.hip, 4x4x1 MFMA
    // blocks of 4 thin channels, 7 filter rows per workgroup) measured 1.09 x (Cout 4) and 1.09 x 4 / 3 (Cin 3) of theirs (SQ_INSTS_MFMA, round 6)
    if (which == 1 && g.Ci < 32) return direct * 32.0 / g.Ci;
    if (which == 2 && g.k == 7 && (g.Ci < 16 || g.Co < 16)) return direct * 1.09 * (g.Ci == 3 ? 4.0 / 3.0 : 1.0);
    return direct;
}

}  // namespace aclgan
