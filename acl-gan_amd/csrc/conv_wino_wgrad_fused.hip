// conv_wino_wgrad_fused.hip -- the Winograd weight gradient of the ResBlock convolutions as ONE kernel + one finish launch (round 4):
//   dU_f[co][ci] = sum over the 4x4 tiles t of  dM_f[t][co] * V_f[t][ci],   dM = A dY A^T,   V = B^T d B,   dw = G^T dU G
// with BOTH transforms done in registers on the way into the MFMAs, so neither V (75 MB) nor dM (75 MB) nor dU exists in HBM
// (conv_wino.hip: wino_input + wino_outgrad + 36 GEMM slices + split finish + wino_filtergrad + two bias launches = 196 us per layer).
// Replaces the weight/bias gradient of ReflectionPad2d(1) + Conv2d(3x3) of the ResBlocks (networks.py:297-310, 366-370; autograd of
// F.conv2d in the reference).
//
// Work decomposition (fp32, v_mfma_f32_32x32x2_f32; same wave / frequency layout as conv_wino_fused.hip):
//   workgroup = 4 waves = 64 output channels x 32 input channels x ALL 36 frequencies, one K slice of the tiles;
//   wave (wi, wj) owns the 3 x 3 frequency block rows 3wi.., columns 3wj..: 9 frequencies x (64 x 32) = 18 accumulator tiles of 32 x 32.
//   K = tiles, walked in GROUPS of four tiles side by side (a 16 x 4 pixel strip of dy, an 18 x 6 pixel strip of x): the MFMA k index (lane
//   half h) is the tile, two k-steps are packed in one float2 (tiles h and h + 2 of the group) so the transforms run in packed form.
//   A operand = dM: every lane reads the 4 x 4 block of dy of ITS tile pair and output channel from LDS and applies A . A^T restricted to the
//               wave's 3 x 3 frequencies: 28 packed operations per 32-channel half.
//   B operand = V: the 5 x 5 sub-patch of x of the tile pair and input channel, B^T . B restricted likewise: 48 packed operations.
//   Raw dy / x strips are staged global -> registers -> LDS one group ahead (four LDS stages); all addresses are SGPR row offsets
//   (computed per group on the scalar unit: batch, tile row, reflection) + one constant lane offset.
//   epilogue = the 36 frequencies of a (co, ci) pair live in four waves: exchange through LDS (two passes of 144 KB), G^T dU G per thread,
//              partial dw of this K slice stored as [slice][co][ky][kx][ci]; the bias gradient (column sums of dy) is accumulated from the
//              staging registers by the workgroups of input-channel block 0.  wgrad_fused_finish_kernel adds the slices in order.
//
// LDS layout of a stage (bytes): x strip  [6 rows][22 pixel slots][32 ci] floats, slot(c) = c + (c >> 2)  (one empty slot after every 4 pixels);
//                                dy strip [4 rows][16 pixels x 256 B + 128 B after every 4 pixels][64 co] floats.
// The two lane halves (tiles h = 0 / 1, four pixels apart) are then 128 B (mod 256) apart: conflict-free; tiles h and h + 2 are a multiple of
// 256 B apart: one ds_read2st64_b32 fetches the packed pair.
#include "common.h"
#include <cstdlib>
#include <algorithm>

namespace aclgan {

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WCO = 64, WCI = 32;                // channels per workgroup
constexpr int XROW = 22 * 128, XB = 6 * XROW;    // x strip: 16 896 bytes
constexpr int DROW = 16 * 256 + 4 * 128, DB = 4 * DROW;      // dy strip: 18 432 bytes
constexpr int STAGE = XB + DB;
constexpr int NST = 4;
constexpr int E_BYTES = 36 * 32 * 32 * 4;        // epilogue exchange buffer (one 32-output-channel half)
constexpr unsigned int OOBV = 0x7ffffff0u;

struct WgP {
    const float* x; const float* dy; float* part; float* partdb;      // part[ks][Co][9][Ci], partdb[ks][Co]
    int B, H, W, Ci, Co, TY, TXS, G, ks, gper, nco, nci, want_db;
    long long xbytes, dybytes;
};

__device__ __forceinline__ int reflg(int v, int n) {
    v = v < 0 ? -v : v;
    return v >= n ? 2 * (n - 1) - v : v;
}
__device__ __forceinline__ f32x2 opq2(float v) {      // a constant pair the compiler cannot fold: keeps the transforms in v_pk_* form
    f32x2 r = {v, v};
    asm volatile("" : "+v"(r));
    return r;
}
__device__ __forceinline__ f32x2 fmap(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x4 fmaq(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }
struct TfK { f32x2 k4, k5n, k4n, k2, k2n; };
// three rows of B^T d (conv_wino_fused.hip bt3): W = 0: rows 0, 1, 2 from d0..d4;  W = 1: rows 3, 4, 5 from d1..d5
template <int W>
__device__ __forceinline__ void bt3g(const TfK& k, const f32x2 (&x)[5], f32x2& o0, f32x2& o1, f32x2& o2) {
    if (W == 0) {
        const f32x2 pp = fmap(k.k4n, x[2], x[4]), qq = fmap(k.k4, x[1], -x[3]);
        o0 = fmap(k.k4, x[0], fmap(k.k5n, x[2], x[4]));
        o1 = pp - qq;
        o2 = pp + qq;
    } else {
        const f32x2 pp = x[3] - x[1], sd = x[2] - x[0];
        o0 = fmap(k.k2, sd, pp);
        o1 = fmap(k.k2n, sd, pp);
        o2 = fmap(k.k4, x[0], fmap(k.k5n, x[2], x[4]));
    }
}
// three rows of A y (A = 6 x 4, the transpose of the output transform): W = 0: rows 0, 1, 2;  W = 1: rows 3, 4, 5.  4 packed operations
template <int W>
__device__ __forceinline__ void a3g(const TfK& k, const f32x2 (&y)[4], f32x2& o0, f32x2& o1, f32x2& o2) {
    if (W == 0) {
        const f32x2 s02 = y[0] + y[2], s13 = y[1] + y[3];
        o0 = y[0];
        o1 = s02 + s13;
        o2 = s02 - s13;
    } else {
        const f32x2 pp = fmap(k.k4, y[2], y[0]), qq = fmap(k.k4, y[3], y[1]);
        o0 = fmap(k.k2, qq, pp);
        o1 = fmap(k.k2n, qq, pp);
        o2 = y[3];
    }
}

// One wave's share of the K loop over the groups [g0, g1) of this workgroup's slice.
template <int WI, int WJ>
__device__ __forceinline__ void wg_wave(const WgP& p, char* smem, const int tid, const int lane, const int co0, const int ci0, const int g0, const int g1,
                                        f32x16 (&acc)[9][2], f32x4& bsum) {
    constexpr bool XW = !(WI == 1 && WJ == 1);      // waves 0..2 stage x (144 threads: 18 pixels x 8 channel quads per row), all four stage dy
    const int l31 = lane & 31, h = lane >> 5;
    const int Ci4 = p.Ci * 4;
    const int n = g1 - g0;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)p.dybytes, 0x00020000);
    const TfK tk = {opq2(4.f), opq2(-5.f), opq2(-4.f), opq2(2.f), opq2(-2.f)};

    // ---- staging constants of this thread ----
    // x: pixel column c (0..17 of the strip, column 16 sx - 1 + c of the image, reflected at the image borders), channel quad q.  The row offset
    // is scalar; the column part has four variants (strip at the left border / right border / both / neither): voffset stays non-negative.
    const bool xact = tid < 144;
    const int xc = tid >> 3, xq = tid & 7;
    const unsigned int xvo_m = xact ? (unsigned int)(xc * Ci4 + xq * 16) : OOBV;
    const unsigned int xvo_f = xact ? (unsigned int)((xc == 0 ? 1 : xc - 1) * Ci4 + xq * 16) : OOBV;
    const unsigned int xvo_l = xact ? (unsigned int)((xc == 17 ? 15 : xc) * Ci4 + xq * 16) : OOBV;
    const unsigned int xvo_fl = xact ? (unsigned int)((xc == 0 ? 1 : (xc == 17 ? 14 : xc - 1)) * Ci4 + xq * 16) : OOBV;
    const int xlw = xact ? (xc + (xc >> 2)) * 128 + xq * 16 : 4 * 128 + xq * 16;      // (inactive lanes of wave 2: the empty slot)
    const int dc = tid >> 4, dq = tid & 15;
    const unsigned int dvo = (unsigned int)(dc * p.Co * 4 + dq * 16);
    const int dlw = XB + dc * 256 + (dc >> 2) * 128 + dq * 16;
    // lane parts of the LDS read addresses
    const int xlb = h * 640 + l31 * 4;
    const int dlb = XB + h * 1152 + l31 * 4;

    // ---- the staging cursor: group cg = (image cb_, tile row cty, strip csx) ----
    int cg = g0, csx = g0 % p.TXS, cty = (g0 / p.TXS) % p.TY, cb_ = g0 / (p.TXS * p.TY);
    int xso[6], dso[4];
    unsigned int xvo = xvo_m;
    auto setgroup = [&]() __attribute__((always_inline)) {
        const bool first = csx == 0, last = csx == p.TXS - 1;
        xvo = first ? (last ? xvo_fl : xvo_f) : (last ? xvo_l : xvo_m);
        const int xcol = (first ? 0 : (16 * csx - 1) * Ci4) + ci0 * 4;
#pragma unroll
        for (int r = 0; r < 6; ++r) xso[r] = __builtin_amdgcn_readfirstlane(((cb_ * p.H + reflg(4 * cty - 1 + r, p.H)) * p.W) * Ci4 + xcol);
#pragma unroll
        for (int r = 0; r < 4; ++r) dso[r] = __builtin_amdgcn_readfirstlane((((cb_ * p.H + 4 * cty + r) * p.W + 16 * csx) * p.Co + co0) * 4);
    };
    auto advance = [&]() __attribute__((always_inline)) {      // (past the last group: stay there, the loads are redundant)
        if (cg + 1 < g1) {
            ++cg;
            if (++csx == p.TXS) { csx = 0; if (++cty == p.TY) { cty = 0; ++cb_; } }
        }
    };
    f32x4 xr[6], dr[4];
    auto issue = [&](int k) __attribute__((always_inline)) {
        if (k < 6) { if (XW) xr[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xvo, xso[k], 0)); }
        else dr[k - 6] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, dvo, dso[k - 6], 0));
    };
    auto wr = [&](int st, int k, bool count) __attribute__((always_inline)) {
        if (k < 6) { if (XW) *reinterpret_cast<f32x4*>(smem + st * STAGE + k * XROW + xlw) = xr[k]; }
        else {
            *reinterpret_cast<f32x4*>(smem + st * STAGE + (k - 6) * DROW + dlw) = dr[k - 6];
            if (count) bsum += dr[k - 6];      // bias gradient: this thread's pixel column and channel quad (count: uniform)
        }
    };

    // ---- LDS reads of one group: the 4 x 4 block of dy (per 32-channel half i) and the 5 x 5 sub-patch of x of the lane's tile pair ----
    f32x2 dyv[4][4];      // [pixel column][pixel row]
    auto rd_dy = [&](int st, int i, int cc) __attribute__((always_inline)) {
        const char* src = smem + st * STAGE + dlb + i * 128 + cc * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            dyv[cc][r] = (f32x2){*reinterpret_cast<const float*>(src + r * DROW), *reinterpret_cast<const float*>(src + r * DROW + 2304)};
    };
    f32x2 d[5][5];        // [column][row]
    auto rd_x = [&](int st, int c) __attribute__((always_inline)) {
        const int cc = WJ + c;
        const char* src = smem + st * STAGE + xlb + (cc + (cc >> 2)) * 128;
#pragma unroll
        for (int r = 0; r < 5; ++r)
            d[c][r] = (f32x2){*reinterpret_cast<const float*>(src + (WI + r) * XROW), *reinterpret_cast<const float*>(src + (WI + r) * XROW + 1280)};
    };
    auto tf_dy = [&](f32x2 (&M)[9]) __attribute__((always_inline)) {      // 28 packed operations
        f32x2 t[3][4];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) a3g<WI>(tk, dyv[cc], t[0][cc], t[1][cc], t[2][cc]);
#pragma unroll
        for (int il = 0; il < 3; ++il) a3g<WJ>(tk, t[il], M[il * 3], M[il * 3 + 1], M[il * 3 + 2]);
    };
    auto tf_x = [&](f32x2 (&V)[9]) __attribute__((always_inline)) {       // 48 packed operations
        f32x2 t[3][5];
#pragma unroll
        for (int c = 0; c < 5; ++c) bt3g<WI>(tk, d[c], t[0][c], t[1][c], t[2][c]);
#pragma unroll
        for (int il = 0; il < 3; ++il) bt3g<WJ>(tk, t[il], V[il * 3], V[il * 3 + 1], V[il * 3 + 2]);
    };
    // 18 accumulator tiles = 288 registers: 16 tiles fill the 256 AGPRs, the MFMAs of the last two are written in their VGPR form by hand
    auto mfma1 = [&](int fi, int i, float a, float b) __attribute__((always_inline)) {
        if (fi == 8) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[fi][i]) : "v"(a), "v"(b));
        else acc[fi][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[fi][i], 0, 0, 0);
    };

    // ---- prologue: groups 0 and 1 in LDS stages 0 and 1, group 2 in flight, operands of group 0 transformed ----
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
        setgroup();
#pragma unroll
        for (int k = 0; k < 10; ++k) issue(k);
#pragma unroll
        for (int k = 0; k < 10; ++k) wr(gq, k, p.want_db && gq < n);
        advance();
    }
    setgroup();
#pragma unroll
    for (int k = 0; k < 10; ++k) issue(k);
    advance();
    __syncthreads();
    f32x2 Ma[2][9], Va[9], Mb[2][9], Vb[9];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) rd_dy(0, i, cc);
        tf_dy(Ma[i]);
    }
#pragma unroll
    for (int c = 0; c < 5; ++c) rd_x(0, c);
    tf_x(Va);

    // ---- main loop: iteration j multiplies the operands of group j (registers) while the strips of group j + 1 are read from stage (j + 1) % 4 and
    // transformed in three bursts, the registers of group j + 2 are written to stage (j + 2) % 4 and the loads of group j + 3 are issued ----
    auto iter = [&](int j, int R, int Ws, f32x2 (&Mc)[2][9], f32x2 (&Vc)[9], f32x2 (&Mn)[2][9], f32x2 (&Vn)[9]) __attribute__((always_inline)) {
        const bool count = p.want_db && j + 2 < n;
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            mfma1(s, 0, Mc[0][s].x, Vc[s].x);
            mfma1(s, 1, Mc[1][s].x, Vc[s].x);
            mfma1(s, 0, Mc[0][s].y, Vc[s].y);
            mfma1(s, 1, Mc[1][s].y, Vc[s].y);
            if (s == 2) { tf_dy(Mn[0]); __builtin_amdgcn_sched_barrier(0); }
            if (s == 5) { tf_dy(Mn[1]); __builtin_amdgcn_sched_barrier(0); }
            if (s == 0) { rd_dy(R, 0, 0); rd_dy(R, 0, 1); }
            if (s == 1) { rd_dy(R, 0, 2); rd_dy(R, 0, 3); }
            if (s == 3) { rd_dy(R, 1, 0); rd_dy(R, 1, 1); }
            if (s == 4) { rd_dy(R, 1, 2); rd_dy(R, 1, 3); }
            if (s == 5) { rd_x(R, 0); rd_x(R, 1); }
            if (s == 6) { rd_x(R, 2); rd_x(R, 3); }
            if (s == 7) rd_x(R, 4);
            wr(Ws, s, count);
            if (s == 0) setgroup();
            issue(s);
            if (s == 8) { wr(Ws, 9, count); issue(9); }
            __builtin_amdgcn_sched_barrier(0);
        }
        tf_x(Vn);
        advance();
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    for (int j = 0; j < n; j += 4) {
        iter(j, 1, 2, Ma, Va, Mb, Vb);
        if (j + 1 < n) iter(j + 1, 2, 3, Mb, Vb, Ma, Va);
        if (j + 2 < n) iter(j + 2, 3, 0, Ma, Va, Mb, Vb);
        if (j + 3 < n) iter(j + 3, 0, 1, Mb, Vb, Ma, Va);
    }
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) wino_wgrad_fused_kernel(WgP p) {
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE > E_BYTES ? NST * STAGE : E_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // blockIdx.x = slice + ks * (co block + nco * ci block): with ks = 8 a K slice lives on one XCD (its dy / x strips are shared by the
    // workgroups of every channel block there, walked in lockstep)
    const int bid = blockIdx.x;
    const int slice = bid % p.ks, rest = bid / p.ks;
    const int cob = rest % p.nco, cib = rest / p.nco;
    const int co0 = cob * WCO, ci0 = cib * WCI;
    const int g0 = min(slice * p.gper, p.G), g1 = min(g0 + p.gper, p.G);

    f32x16 acc[9][2];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 bsum = (f32x4)(0.f);
    WgP q = p;
    q.want_db = (p.want_db && cib == 0) ? 1 : 0;
    if (g1 > g0) {
        if (wave == 0) wg_wave<0, 0>(q, smem, tid, lane, co0, ci0, g0, g1, acc, bsum);
        else if (wave == 1) wg_wave<0, 1>(q, smem, tid, lane, co0, ci0, g0, g1, acc, bsum);
        else if (wave == 2) wg_wave<1, 0>(q, smem, tid, lane, co0, ci0, g0, g1, acc, bsum);
        else wg_wave<1, 1>(q, smem, tid, lane, co0, ci0, g0, g1, acc, bsum);
    }

    // ---- epilogue: bias partial (16 pixel columns per channel quad, summed in order), then the frequency exchange and G^T dU G ----
    float* Es = reinterpret_cast<float*>(smem);
    if (q.want_db) {
        __syncthreads();
        *reinterpret_cast<f32x4*>(Es + tid * 4) = bsum;      // [pixel column dc][channel quad dq][4]
        __syncthreads();
        if (tid < 64) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) s += Es[c * 64 + tid];
            p.partdb[(size_t)slice * p.Co + co0 + tid] = s;
        }
    }
    const int wi = wave >> 1, wj = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int col = tid >> 3, c4 = tid & 7;
    const f32x4 q4 = (f32x4)(0.25f), s6 = (f32x4)(1.f / 6.f), s12 = (f32x4)(1.f / 12.f), s24 = (f32x4)(1.f / 24.f);
    auto gt3v = [&](const f32x4 (&u)[6], f32x4& o0, f32x4& o1, f32x4& o2) __attribute__((always_inline)) {      // G^T u
        const f32x4 a12 = u[1] + u[2], a34 = u[3] + u[4];
        o0 = fmaq(q4, u[0], fmaq(s24, a34, -(s6 * a12)));
        o1 = fmaq(s6, u[2] - u[1], s12 * (u[3] - u[4]));
        o2 = fmaq(s6, a34 - a12, u[5]);
    };
#pragma unroll
    for (int P = 0; P < 2; ++P) {
        __syncthreads();
#pragma unroll
        for (int fi = 0; fi < 9; ++fi) {
            const int f = (3 * wi + fi / 3) * 6 + 3 * wj + fi % 3;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                Es[(f * 32 + row) * 32 + l31] = acc[fi][P][r];
            }
        }
        __syncthreads();
        f32x4 t[3][6];
#pragma unroll
        for (int jf = 0; jf < 6; ++jf) {
            f32x4 u[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) u[i] = *reinterpret_cast<const f32x4*>(Es + ((i * 6 + jf) * 32 + col) * 32 + c4 * 4);
            gt3v(u, t[0][jf], t[1][jf], t[2][jf]);
        }
        float* dst = p.part + (((size_t)slice * p.Co + co0 + 32 * P + col) * 9) * p.Ci + ci0 + c4 * 4;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            f32x4 o[3];
            gt3v(t[a], o[0], o[1], o[2]);
#pragma unroll
            for (int bb = 0; bb < 3; ++bb) *reinterpret_cast<f32x4*>(dst + (size_t)(a * 3 + bb) * p.Ci) = o[bb];
        }
    }
}

// dw[i] += part[0][i] + part[1][i] + ...;  db[c] += partdb[0][c] + ...   (slices in index order: reproducible)
__global__ void __launch_bounds__(256) wgrad_fused_finish_kernel(const float* __restrict__ part, int64_t n4, int ks, float* __restrict__ dw,
                                                                 const float* __restrict__ partdb, int Co, float* __restrict__ db) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        f32x4 s = reinterpret_cast<const f32x4*>(part)[i];
        for (int z = 1; z < ks; ++z) s += reinterpret_cast<const f32x4*>(part)[(size_t)z * n4 + i];
        reinterpret_cast<f32x4*>(dw)[i] += s;
    }
    if (db && i < Co) {
        float s = partdb[i];
        for (int z = 1; z < ks; ++z) s += partdb[(size_t)z * Co + i];
        db[i] += s;
    }
}

int g_wgrad_fused = -1;

struct WgPlan { int TY, TXS, G, nblk, ks, gper; };
WgPlan wg_plan(const ConvGeom& g) {
    WgPlan q;
    q.TY = g.Ho / 4; q.TXS = g.Wo / 16; q.G = g.B * q.TY * q.TXS;
    q.nblk = (g.Co / WCO) * (g.Ci / WCI);
    q.ks = std::max(1, std::min(std::min(q.G, 64), 256 / std::max(1, q.nblk)));
    q.gper = cdiv(q.G, q.ks);
    q.ks = cdiv(q.G, q.gper);
    return q;
}

}  // namespace

// tuning / test knob behind aclgan_set_tuning("wino_wgrad_fused", v): 0 = the pipeline of conv_wino.hip, 1 = the fused kernel where it pays,
// 2 = wherever the shape is eligible; returns the previous value.  ACLGAN_WINO_WGRAD_FUSED sets the default.
int wino_wgrad_fused_mode() {
    if (g_wgrad_fused < 0) { const char* e = getenv("ACLGAN_WINO_WGRAD_FUSED"); g_wgrad_fused = e ? atoi(e) : 1; if (g_wgrad_fused < 0 || g_wgrad_fused > 2) g_wgrad_fused = 1; }
    return g_wgrad_fused;
}
int set_wino_wgrad_fused(int v) { const int old = wino_wgrad_fused_mode(); g_wgrad_fused = (v < 0 || v > 2) ? 1 : v; return old; }

// 3x3 stride-1 reflect-pad-1 layers with W a multiple of 16, H of 4, Cout of 64, Cin of 32
bool wino_wgrad_fused_ok(const ConvGeom& g) {
    const int m = wino_wgrad_fused_mode();
    if (m == 0) return false;
    const bool shape = g.k == 3 && g.s == 1 && g.p == 1 && g.up == 0 && g.Ho % 4 == 0 && g.Wo % 16 == 0 && g.Co % WCO == 0 && g.Ci % WCI == 0 && g.Hi >= 4 &&
                       (long long)g.B * g.Hi * g.Wi * std::max(g.Ci, g.Co) * 4 < 0x7fffffe0ll;
    if (!shape || m == 2) return shape;
    const WgPlan q = wg_plan(g);
    // cost model (microseconds): a workgroup walks gper groups at ~1.45 us each; the pipeline: conv_wino.hip's five launches
    const double t_fused = (25.0 + 1.45 * q.gper) * std::ceil((double)q.nblk * q.ks / 256.0) + 8.0;
    const double t_pipe = 60.0 + 0.066 * (double)g.B * q.TY * q.TXS * 4 * ((double)g.Ci * g.Co / 65536.0);
    return t_fused <= t_pipe;
}
size_t wino_wgrad_fused_scratch_bytes(const ConvGeom& g) {
    const WgPlan q = wg_plan(g);
    return (((size_t)q.ks * g.Co * 9 * g.Ci * 4 + 255) & ~(size_t)255) + (((size_t)q.ks * g.Co * 4 + 255) & ~(size_t)255) + 256;
}
int wino_wgrad_fused(const ConvGeom& g, const float* x, const float* dy, float* dw, float* db, void* scratch, hipStream_t st) {
    if (!wino_wgrad_fused_ok(g) || !scratch || !dw) return ACLGAN_EUNSUPPORTED;
    const WgPlan q = wg_plan(g);
    WgP p;
    p.x = x; p.dy = dy; p.part = (float*)scratch;
    p.partdb = (float*)((char*)scratch + (((size_t)q.ks * g.Co * 9 * g.Ci * 4 + 255) & ~(size_t)255));
    p.B = g.B; p.H = g.Hi; p.W = g.Wi; p.Ci = g.Ci; p.Co = g.Co; p.TY = q.TY; p.TXS = q.TXS; p.G = q.G; p.ks = q.ks; p.gper = q.gper;
    p.nco = g.Co / WCO; p.nci = g.Ci / WCI; p.want_db = db ? 1 : 0;
    p.xbytes = (long long)g.B * g.Hi * g.Wi * g.Ci * 4; p.dybytes = (long long)g.B * g.Ho * g.Wo * g.Co * 4;
    hipLaunchKernelGGL(wino_wgrad_fused_kernel, dim3(q.nblk * q.ks), dim3(256), 0, st, p);
    ACL_CHECK_LAUNCH("wino_wgrad_fused_kernel");
    const int64_t n4 = (int64_t)g.Co * 9 * g.Ci / 4;
    hipLaunchKernelGGL(wgrad_fused_finish_kernel, dim3((unsigned)cdiv64(std::max<int64_t>(n4, g.Co), 256)), dim3(256), 0, st, p.part, n4, q.ks, dw,
                       p.partdb, g.Co, db);
    ACL_CHECK_LAUNCH("wgrad_fused_finish_kernel");
    return ACLGAN_OK;
}

}  // namespace aclgan
