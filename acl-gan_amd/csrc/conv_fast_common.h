// conv_fast_common.h -- parameter blocks, row enumerations and small helper kernels shared by the tuned fp32
// convolution kernels (conv_fast.hip) and their 16-bit MFMA counterparts (conv_fast16.hip).  Everything lives in an
// anonymous namespace: each translation unit gets its own copy (no cross-TU device symbols, no -fgpu-rdc).
#pragma once
#include "common.h"
#include <cstdlib>
#include <algorithm>

namespace aclgan {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// native vector type for the staging registers: HIP's float4 is a struct whose copies become
// llvm.memcpy between address spaces, which SROA does not promote (the registers end up in scratch)
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
__device__ __forceinline__ int xcd_map(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// position r of the ring "Hc x Wc grid minus the box [ylo,yhi] x [xlo,xhi]" (top strip, bottom strip, left, right)
__device__ __forceinline__ void ring_decode(int r, int Hc, int Wc, int ylo, int yhi, int xlo, int xhi, int& y, int& x) {
    const int ny = yhi - ylo + 1;
    const int top = ylo * Wc, bot = (Hc - 1 - yhi) * Wc, left = ny * xlo;
    if (r < top) { y = r / Wc; x = r - y * Wc; return; }
    r -= top;
    if (r < bot) { const int t = r / Wc; y = yhi + 1 + t; x = r - t * Wc; return; }
    r -= bot;
    if (r < left) { const int t = r / xlo; y = ylo + t; x = r - t * xlo; return; }
    r -= left;
    const int wr = Wc - 1 - xhi;
    const int t = r / wr; y = ylo + t; x = xhi + 1 + r - t * wr;
}
__device__ __forceinline__ int ring_count(int Hc, int Wc, int ylo, int yhi, int xlo, int xhi) {
    return Hc * Wc - (yhi - ylo + 1) * (xhi - xlo + 1);
}

struct FwdFP {
    const float* x; const float* w; const float* bias; float* y;
    int Hi, Wi, Ci, Ho, Wo, Co, k, s, p, up, Hu, Wu, M, K, act, tiles_n, nwg, nkz;   // nkz: k-tiles per blockIdx.z slice (split-K)
    // sub-pixel decomposition of the "2x nearest upsample + reflect pad 2 + 5x5" decoder convs (see conv_up5_*):
    //   ring   > 0: only the output ring of that width is produced (exact gather path); M = B * ring pixels
    //   phases = 1: blockIdx.z is the output phase (py,px); this launch is a VALID 3x3 conv on the low-res
    //               input with the phase's merged weights, scattered to y[2(oy+1)+py][2(ox+1)+px] of an Hf x Wf map
    int B, ring, phases, Hf, Wf;
    // split-K launches: part != nullptr -> slice z stores its partial tile at part[(z*rows + m)*Co + n] (plain stores) and
    // fwd_split_finish_kernel adds the slices in ORDER (+ bias, activation): bit-reproducible, unlike the atomics path
    float* part; int rows;
    const unsigned short* w16;   // 16-bit kernels: OHWI weights (or merged phase weights) as bf16 / fp16 bit patterns
    const unsigned short* x16;   // 16-bit kernels, optional: the input already rounded to the 16-bit type by its producer (same NHWC layout)
    // batched plain GEMMs (the 36 frequency planes of the Winograd path): blockIdx.y = slice f, operands / result of slice f start
    // fs_* floats after those of slice 0.  0 = off.
    int fsl; long long fs_x, fs_w, fs_y;
    int fsx_mod;                 // > 0: the x operand of slice f is plane f % fsx_mod
    int yst = 0;                 // 16-bit kernels: storage of y (st16.h: 0 fp32, else the 16-bit compute dtype); y then points at 16-bit data
};

__device__ __forceinline__ bool fwd_row(const FwdFP& p, int m, int& b, int& oy, int& ox) {
    if (p.ring > 0) {
        const int R = ring_count(p.Ho, p.Wo, p.ring, p.Ho - 1 - p.ring, p.ring, p.Wo - 1 - p.ring);
        if (m >= p.B * R) return false;
        b = m / R;
        ring_decode(m - b * R, p.Ho, p.Wo, p.ring, p.Ho - 1 - p.ring, p.ring, p.Wo - 1 - p.ring, oy, ox);
        return true;
    }
    if (m >= p.M) return false;
    const int hw = p.Ho * p.Wo;
    b = m / hw; const int rem = m - b * hw;
    oy = rem / p.Wo; ox = rem - oy * p.Wo;
    return true;
}

// ordered reduction of the split-K partials: y[o(m)][c] = act(sum_z part[z][m][c] + bias[c])
__global__ void fwd_split_finish_kernel(FwdFP p, int splits) {
    if (p.Co & 3) {                                  // narrow heads (the discriminators' 1-channel 1x1 conv): scalar
        const int64_t n = (int64_t)p.rows * p.Co;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
            const int m = (int)(i / p.Co), c = (int)(i - (int64_t)m * p.Co);
            int b, oy, ox;
            if (!fwd_row(p, m, b, oy, ox)) continue;
            float s = p.part[(size_t)m * p.Co + c];
            for (int z = 1; z < splits; ++z) s += p.part[((size_t)z * p.rows + m) * p.Co + c];
            if (p.bias) s += p.bias[c];
            p.y[((size_t)(b * p.Ho + oy) * p.Wo + ox) * p.Co + c] = act_apply(s, p.act);
        }
        return;
    }
    const int C4 = p.Co >> 2;
    const int64_t n = (int64_t)p.rows * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / C4), c4 = (int)(i - (int64_t)m * C4);
        int b, oy, ox;
        if (!fwd_row(p, m, b, oy, ox)) continue;
        f32x4 s = *reinterpret_cast<const f32x4*>(p.part + (size_t)m * p.Co + c4 * 4);
        for (int z = 1; z < splits; ++z) s += *reinterpret_cast<const f32x4*>(p.part + ((size_t)z * p.rows + m) * p.Co + c4 * 4);
        if (p.bias) s += *reinterpret_cast<const f32x4*>(p.bias + c4 * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = act_apply(s[e], p.act);
        *reinterpret_cast<f32x4*>(p.y + ((size_t)(b * p.Ho + oy) * p.Wo + ox) * p.Co + c4 * 4) = o;
    }
}

// rows of a launch and its split-K plan (shared by the launcher and the scratch-size query)
static int fwd_rows(const ConvGeom& g, int ring) {
    if (ring > 0) return g.B * (g.Ho * g.Wo - std::max(0, g.Ho - 2 * ring) * std::max(0, g.Wo - 2 * ring));
    return g.M;
}
// bk = k-tile depth of the kernel family (16: fp32, 32: 16-bit operands); a slice is never thinner than 128 k values
static void fwd_split_plan(int rows, int Co, int K, int bk, int* splits, int* nkz) {
    const int BM = Co > 64 ? 128 : 256, BN = Co > 64 ? 128 : (Co > 32 ? 64 : 32);
    const int nwg = cdiv(rows, BM) * cdiv(Co, BN), nk = K / bk;
    int sp = 1;
    static int thr = -1;      // grids below this many workgroups split K (ACLGAN_SPLIT_NWG; default 256 = one workgroup per CU; 128 through round 3: round 4 measured 98.4 -> 97.5 ms per fp32 step)
    if (thr < 0) { const char* e = getenv("ACLGAN_SPLIT_NWG"); thr = e ? atoi(e) : 256; }
    if (nwg < thr && nk * bk >= 512) sp = max(1, min(nk * bk / 128, 512 / nwg));   // floor: 512 = one full round at 2 workgroups per CU
    *nkz = cdiv(nk, sp);
    *splits = cdiv(nk, *nkz);
}
// bytes of partial storage a (ring or plain) launch wants (0: it does not split)
static size_t fwd_partial_bytes(const ConvGeom& g, int ring, int bk) {
    if (g.Ci % bk != 0) return 0;
    int sp, nkz;
    const int rows = fwd_rows(g, ring);
    fwd_split_plan(rows, g.Co, g.K, bk, &sp, &nkz);
    return sp > 1 ? (size_t)sp * rows * g.Co * sizeof(float) : 0;
}

// zero the output ring (width p.ring) ahead of a split-K ring launch
__global__ void ring_zero_kernel(FwdFP p, int rows) {
    const int C4 = p.Co >> 2;
    const int64_t n = (int64_t)rows * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / C4), c4 = (int)(i - (int64_t)m * C4);
        int b, oy, ox;
        if (fwd_row(p, m, b, oy, ox))
            *reinterpret_cast<f32x4*>(p.y + ((size_t)(b * p.Ho + oy) * p.Wo + ox) * p.Co + c4 * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

__global__ void bias_act_kernel(float* __restrict__ y, const float* __restrict__ bias, int Co, int act, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        y[i] = act_apply(y[i] + (bias ? bias[i % Co] : 0.f), act);
}

struct DgFP {
    const float* dy; const float* w; float* dxp;
    int Ho, Wo, Co, Ci, k, s, Hp, Wp, Hc, Wc, Mc, tiles_n, nwg, ksplit;   // ksplit: split-K factor (blockIdx.z = class * ksplit + slice)
    // row enumeration mode: 0 = the whole padded grid -> dxp (scratch; fold kernel follows)
    //                       1 = interior positions only (padded coords in [pad, pad+H)) -> written straight into dx
    //                       2 = the halo ring -> atomically mirrored into dx (reflection-pad backward)
    int mode, accumulate, pad, B, Hi, Wi;
    // sub-pixel path of the upsample+5x5 convs (conv_up5_dgrad):
    //   dyv = 1: dy is read at [2(oy+1)+py][2(ox+1)+px] of an Hf x Wf map (phase view of the hi-res gradient)
    //   band > 0 (mode 2): the ring is "padded grid minus the box inset by band" and only output pixels of the
    //            output ring of width 2 contribute (the interior is covered by the four phase launches);
    //            targets are folded through reflect + >>upshift into the Hd x Wd low-res dx
    int dyv, py, px, Hf, Wf, band, upshift, Hd, Wd;
    const unsigned short* w16t;  // 16-bit kernels: weights transposed to [tap][cin][cout] (cout contiguous = the GEMM k axis of dgrad)
    // mode 3 = interior AND halo in ONE launch: M-tiles [0, Ti) enumerate the interior rows (mode 1), tiles [Ti, ..) the halo
    // ring (mode 2).  Interior pixels that are mirror targets of the reflection are combined with atomics by both kinds of
    // tile (the caller zeroes that frame first unless it accumulates); all other pixels keep plain stores.
    int Ti;
    // deterministic mode: ringpad = 1 makes a mode-2 launch STORE each ring position at its padded-grid index of dxp (a zeroed
    // [B][Hp][Wp][Ci] scratch; every position has one writer) instead of adding it onto its mirror target with atomics;
    // conv_fold(accumulate) then gathers the ring into dx in a fixed order
    int ringpad = 0;
};

// image rows / columns that receive mirrored halo gradients (reflection pad p on a size-n axis): padded -j -> j, n-1+j -> n-1-j
__host__ __device__ __forceinline__ bool dg_is_target(int i, int n, int p) { return (i >= 1 && i <= p) || (i >= n - 1 - p && i <= n - 2); }

// zero the mirror-target frame of dx [B][H][W][C] (rows 1..p, H-1-p..H-2 full width; columns likewise full height)
__global__ void dg_frame_zero_kernel(float* __restrict__ dx, int B, int H, int W, int C, int p) {
    const int C4 = C >> 2, per = 2 * p * W + 2 * p * H;
    const int64_t n = (int64_t)B * per * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        int64_t t = i / C4;
        const int j = (int)(t % per), b = (int)(t / per);
        int y, x;
        if (j < 2 * p * W) { const int r = j / W; y = r < p ? 1 + r : H - 1 - p + (r - p); x = j - r * W; }
        else { const int jj = j - 2 * p * W, cc = jj / H; x = cc < p ? 1 + cc : W - 1 - p + (cc - p); y = jj - cc * H; }
        *reinterpret_cast<f32x4*>(dx + ((size_t)(b * H + y) * W + x) * C + c4 * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

// class-grid box of the interior positions for parity class (cy, cx)
__device__ __forceinline__ void dg_box(const DgFP& p, int cy, int cx, int& ylo, int& yhi, int& xlo, int& xhi) {
    if (p.band > 0) { ylo = p.band; xlo = p.band; yhi = p.Hc - 1 - p.band; xhi = p.Wc - 1 - p.band; return; }
    ylo = p.pad > cy ? (p.pad - cy + p.s - 1) / p.s : 0;
    xlo = p.pad > cx ? (p.pad - cx + p.s - 1) / p.s : 0;
    yhi = min(p.Hc - 1, (p.pad + p.Hi - 1 - cy) / p.s);
    xhi = min(p.Wc - 1, (p.pad + p.Wi - 1 - cx) / p.s);
}

// row m of this launch -> (image b, class-grid coords y2, x2); returns false past the end
__device__ __forceinline__ bool dg_row(const DgFP& p, int m, int ylo, int yhi, int xlo, int xhi, int& b, int& y2, int& x2) {
    const int ny = yhi - ylo + 1, nx = xhi - xlo + 1;
    if (p.mode == 0) {
        const int hw = p.Hc * p.Wc;
        if (m >= p.B * hw) return false;
        b = m / hw; const int rem = m - b * hw;
        y2 = rem / p.Wc; x2 = rem - y2 * p.Wc;
        return true;
    }
    if (p.mode == 1) {
        const int hw = ny * nx;
        if (m >= p.B * hw) return false;
        b = m / hw; const int rem = m - b * hw;
        const int yy = rem / nx;
        y2 = ylo + yy; x2 = xlo + rem - yy * nx;
        return true;
    }
    const int R = p.Hc * p.Wc - max(ny, 0) * max(nx, 0);
    if (R <= 0 || m >= p.B * R) return false;
    // STRIP-major over the batch: [all images' top strips][bottom strips][left strips][right strips].  A 128-row tile then lies
    // inside one strip (except at the three seams), so the per-tile tap list of a halo launch holds the k taps that strip can see
    // instead of the union over the strips of one image (3x3, pad 1: 3 taps instead of 6-8 -- the tile's k loop is that much shorter).
    // Side strips are column-major for the same reason (wide bands of the sub-pixel layers: one or two x positions per tile).
    if (p.band > 0 && ny > 0 && nx > 0) {
        // Wide bands (the ring launch of the sub-pixel layers; round 6).  A tile's k loop runs over the UNION of the filter taps its rows can
        // see (tap_list), and with the strips above every top / bottom tile contained corner columns (which see the taps of BOTH directions:
        // up to all 25) and two padded rows: 5 - 25 taps per tile, the launch as slow as its slowest tiles (445 us for 13 GFLOP on the 256 -> 128
        // layer).  Now: the four corners first (the only tiles that need many taps: scheduled first), then each strip ordered so that a tile
        // holds ONE padded row (column) across the images of the batch: ty in {r - 1, r} x 5 tx = 10 taps (5 at the outermost / innermost row).
        const int bt = p.Hc - 1 - yhi, rt = p.Wc - 1 - xhi;            // bottom rows, right columns
        const int tb = ylo + bt, lr = xlo + rt, Cn = tb * lr;
        if (m < p.B * Cn) {
            b = m / Cn; const int r = m - b * Cn; const int cy_ = r / lr, cx_ = r - cy_ * lr;
            y2 = cy_ < ylo ? cy_ : yhi + 1 + (cy_ - ylo); x2 = cx_ < xlo ? cx_ : xhi + 1 + (cx_ - xlo);
            return true;
        }
        m -= p.B * Cn;
        const int rowlen = p.B * nx, collen = p.B * ny;
        if (m < ylo * rowlen) { const int t = m / rowlen, r = m - t * rowlen; b = r / nx; y2 = t; x2 = xlo + r - b * nx; return true; }
        m -= ylo * rowlen;
        if (m < bt * rowlen) { const int t = m / rowlen, r = m - t * rowlen; b = r / nx; y2 = yhi + 1 + t; x2 = xlo + r - b * nx; return true; }
        m -= bt * rowlen;
        if (m < xlo * collen) { const int t = m / collen, r = m - t * collen; b = r / ny; x2 = t; y2 = ylo + r - b * ny; return true; }
        m -= xlo * collen;
        if (m >= rt * collen) return false;
        { const int t = m / collen, r = m - t * collen; b = r / ny; x2 = xhi + 1 + t; y2 = ylo + r - b * ny; }
        return true;
    }
    const int top = min(ylo, p.Hc) * p.Wc, bot = min(p.Hc - 1 - yhi, p.Hc - min(ylo, p.Hc)) * p.Wc, left = max(ny, 0) * xlo;
    const int right = R - top - bot - left;
    if (top > 0 && m < p.B * top) { b = m / top; const int r = m - b * top; y2 = r / p.Wc; x2 = r - y2 * p.Wc; return true; }
    m -= p.B * top;
    if (bot > 0 && m < p.B * bot) { b = m / bot; const int r = m - b * bot; const int t = r / p.Wc; y2 = yhi + 1 + t; x2 = r - t * p.Wc; return true; }
    m -= p.B * bot;
    if (left > 0 && m < p.B * left) { b = m / left; const int r = m - b * left; x2 = r / ny; y2 = ylo + r - x2 * ny; return true; }
    m -= p.B * left;
    if (right <= 0) return false;
    b = m / right;
    const int r = m - b * right, t = r / ny;
    x2 = xhi + 1 + t; y2 = ylo + r - t * ny;
    return true;
}

struct WgFP {
    const float* x; const float* dy; float* dw; float* db;
    int Hi, Wi, Ci, Ho, Wo, Co, k, s, p, up, Hu, Wu, P, Kn, chunk, tiles_n, nwg;
    // sub-pixel path of the upsample+5x5 convs (see conv_up5_*): ring > 0: only the pixels of the output ring
    // of that width are summed (exact gather); phases = 1: blockIdx.y is the output phase, dy is read at
    // [2(oy+1)+py][2(ox+1)+px] of an Hf x Wf map and the result goes to dw + phase*Co*Kn
    int B, ring, phases, Hf, Wf;
    // batched plain "A^T B" GEMMs (the 36 frequency planes of the Winograd weight gradient): blockIdx.y = slice f; x / dy of slice
    // f start fs_x / fs_dy floats after those of slice 0; the result goes to dw + f*Co*Kn (like a phase).  0 = off.
    int fsl; long long fs_x, fs_dy;
    int fsx_mod;                 // > 0: the x operand of slice f is plane f % fsx_mod
    // deterministic mode of the atomics kernel (conv_wgrad_fast_kernel): pixel slice z accumulates into its own zeroed copy
    // dw + z*dw_zs / db + z*db_zs (one writer per element), reduce_slices_ordered adds the copies in order.  0 = shared dw / db.
    long long dw_zs = 0, db_zs = 0;
    int dw_overwrite = 0;        // ordered-slice launches: wgrad_finish_kernel stores the sum (dw = ...) instead of accumulating (dw += ...)
    int xst = 0, dyst = 0;       // 16-bit kernels: storage of x / dy (st16.h); non-zero: the pointer addresses 16-bit data
};

__device__ __forceinline__ void wg_coord(const WgFP& p, int pix, int& b, int& oy, int& ox) {
    if (p.ring > 0) {
        const int R = ring_count(p.Ho, p.Wo, p.ring, p.Ho - 1 - p.ring, p.ring, p.Wo - 1 - p.ring);
        b = pix / R;
        ring_decode(pix - b * R, p.Ho, p.Wo, p.ring, p.Ho - 1 - p.ring, p.ring, p.Wo - 1 - p.ring, oy, ox);
        return;
    }
    const int hw = p.Ho * p.Wo;
    b = pix / hw; const int rem = pix - b * hw;
    oy = rem / p.Wo; ox = rem - oy * p.Wo;
}

__device__ __forceinline__ void up5_range(int ph, int a, int& lo, int& hi) {
    if (ph == 0) { lo = 2 * a; hi = min(2 * a + 1, 4); }
    else { lo = max(2 * a - 1, 0); hi = 2 * a; }
}

// wp[phase][co][a][b][ci] = sum_{ky in S(py,a)} sum_{kx in S(px,b)} w[co][ky][kx][ci]
__global__ void up5_merge_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci) {
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
        *reinterpret_cast<f32x4*>(wp + i * 4) = s;
    }
}

// dw[co][ky][kx][ci] += sum over the 4 phases of dwp[phase][co][a(py,ky)][b(px,kx)][ci]
__global__ void up5_scatter_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Co, int Ci) {
    const int C4 = Ci >> 2;
    const int64_t n = (int64_t)Co * 25 * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        int64_t t = i / C4;
        const int kx = (int)(t % 5); t /= 5;
        const int ky = (int)(t % 5);
        const int co = (int)(t / 5);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int a = (ph >> 1) ? (ky + 1) >> 1 : ky >> 1;
            const int b = (ph & 1) ? (kx + 1) >> 1 : kx >> 1;
            s += *reinterpret_cast<const f32x4*>(dwp + ((((size_t)ph * Co + co) * 3 + a) * 3 + b) * Ci + c4 * 4);
        }
        f32x4* o = reinterpret_cast<f32x4*>(dw + i * 4);
        *o += s;
    }
}

bool up5_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("ACLGAN_NOUP5"); v = (e && atoi(e)) ? 0 : 1; }
    return v == 1;
}
bool up5_eligible(const ConvGeom& g) {
    return up5_enabled() && g.up == 1 && g.k == 5 && g.p == 2 && g.s == 1 && g.Ci % 16 == 0 && g.Co % 16 == 0 && g.Hi >= 4 && g.Wi >= 4;
}

bool fast_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("ACLGAN_NOFAST"); v = (e && atoi(e)) ? 0 : 1; }
    return v == 1;
}

// ------------------------------------------------------------------------------------------
// split-K weight gradients WITHOUT atomics (conv_fast.hip: conv_wgrad_kc_kernel, conv_fast16.hip: conv_wgrad16_kernel):
// every slice stores its [BM][BN] tile in MFMA-fragment row/column order (128-byte coalesced rows) plus its bias column
// sums; wgrad_finish_kernel adds the slices in ORDER into dw / db -> the weight gradient is reproducible bit for bit.
// Tile rows are interleaved: LDS row r of a tile with R rows holds channel 4*(r % (R/4)) + r / (R/4) (the staging threads
// transpose 4 channels x n pixels in registers and consecutive lanes must hit consecutive LDS rows).
// ------------------------------------------------------------------------------------------
struct WgPartX {
    float* part;        // [slice][phase][tile][BM][BN] partial tiles (nullptr: single slice, tile added straight into dw)
    float* part_b;      // [slice][phase][Co] partial bias sums
    int ny;             // phases (4: sub-pixel launch, else 1)
};

// dw[phase][m][n..n+3] += sum over slices (in order) of the partial tiles; db[m] += sum over slices and phases
__global__ void wgrad_finish_kernel(WgFP p, WgPartX xp, int BM, int BN, int splits) {
    const int MQ = BM / 4, NQ = BN / 4, N4 = p.Kn >> 2;
    const int64_t n = (int64_t)xp.ny * p.Co * N4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int n4 = (int)(i % N4);
        int64_t t = i / N4;
        const int m = (int)(t % p.Co), phase = (int)(t / p.Co);
        const int tm = m / BM, ml = m - tm * BM, tn = (n4 * 4) / BN, nl = n4 * 4 - tn * BN;
        const int rw = (ml & 3) * MQ + (ml >> 2), q = nl >> 2;
        const int tile = tm * p.tiles_n + tn;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        const float* p0 = xp.part + ((size_t)phase * p.nwg + tile) * ((size_t)BM * BN) + (size_t)rw * BN + q;
        const size_t zs = (size_t)xp.ny * p.nwg * ((size_t)BM * BN);
        int z = 0;
        for (; z + 4 <= splits; z += 4) {       // the loads of four slices in flight, added in slice order (one dependent round trip per slice otherwise)
            float a[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* pt = p0 + (size_t)(z + u) * zs;
                a[u][0] = pt[0]; a[u][1] = pt[NQ]; a[u][2] = pt[2 * NQ]; a[u][3] = pt[3 * NQ];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { s[0] += a[u][0]; s[1] += a[u][1]; s[2] += a[u][2]; s[3] += a[u][3]; }
        }
        for (; z < splits; ++z) {
            const float* pt = p0 + (size_t)z * zs;
            s[0] += pt[0]; s[1] += pt[NQ]; s[2] += pt[2 * NQ]; s[3] += pt[3 * NQ];
        }
        f32x4* o = reinterpret_cast<f32x4*>(p.dw + ((size_t)phase * p.Co + m) * p.Kn + n4 * 4);
        if (p.dw_overwrite) *o = s; else *o += s;
    }
    if (p.db)
        for (int m = blockIdx.x * 256 + threadIdx.x; m < p.Co; m += gridDim.x * 256) {
            float s = 0.f;
            for (int z = 0; z < splits; ++z)
                for (int ph = 0; ph < xp.ny; ++ph) s += xp.part_b[((size_t)z * xp.ny + ph) * p.Co + m];
            p.db[m] += s;
        }
}

// tile shape and split plan of such a launch (shared by the launchers and the scratch-size queries).
// bk: pixels per k-tile (16 fp32 / 32 16-bit); target: workgroups in flight the launch aims for
struct WgPlan { int BM, BN, tiles_n, nwg, splits, chunk; };
static WgPlan wgrad_plan(int Co, int Ci, int Kn, int P, int ny, int bk, int target) {
    WgPlan q;
    q.BM = Co % 128 == 0 ? 128 : 64;
    q.BN = Ci % 128 == 0 ? 128 : 64;
    q.tiles_n = Kn / q.BN;
    q.nwg = (Co / q.BM) * q.tiles_n;
    int splits = std::max(1, target / (q.nwg * ny));
    splits = std::max(1, std::min(splits, cdiv(P, 512)));    // a slice is at least 512 pixels deep
    q.chunk = cdiv(cdiv(P, splits), bk) * bk;
    q.splits = cdiv(P, q.chunk);
    return q;
}
static size_t wgrad_partial_bytes(const WgPlan& q, int Co, int ny) {
    if (q.splits <= 1 && ny == 1) return 0;      // (phase launches always go through the partials: 4 phases share one db)
    return ((size_t)q.splits * ny * q.nwg * q.BM * q.BN + (size_t)q.splits * ny * Co) * sizeof(float);
}
static size_t up5_merged_bytes(const ConvGeom& g) { return ((size_t)4 * g.Co * 9 * g.Ci * sizeof(float) + 255) & ~(size_t)255; }
static size_t up5_dwp_bytes(const ConvGeom& g) { return ((size_t)4 * g.Co * 9 * g.Ci * sizeof(float) + 255) & ~(size_t)255; }
static int up5_ring_pixels(const ConvGeom& g) { return g.B * (g.Ho * g.Wo - (g.Ho - 4) * (g.Wo - 4)); }
// scratch of a partial-store weight gradient of layer g: [dwp (sub-pixel layers)][partials]
static size_t wgrad_part_scratch(const ConvGeom& g, int bk, int target) {
    if (up5_eligible(g)) {
        const WgPlan a = wgrad_plan(g.Co, g.Ci, 9 * g.Ci, g.B * (g.Hi - 2) * (g.Wi - 2), 4, bk, target);
        const WgPlan b = wgrad_plan(g.Co, g.Ci, g.K, up5_ring_pixels(g), 1, bk, target);
        return up5_dwp_bytes(g) + std::max(wgrad_partial_bytes(a, g.Co, 4), wgrad_partial_bytes(b, g.Co, 1));
    }
    return wgrad_partial_bytes(wgrad_plan(g.Co, g.Ci, g.K, g.M, 1, bk, target), g.Co, 1);
}


}  // namespace
}  // namespace aclgan
