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
//   Raw dy / x strips are copied global -> LDS directly (buffer_load_dwordx4 ... lds: no staging registers, no ds_write) two groups ahead
//   through four LDS stages; all addresses are SGPR row offsets (computed per group on the scalar unit: image, tile row, reflection) + one
//   constant lane offset.
//   epilogue = the 36 frequencies of a (co, ci) pair live in four waves: exchange through LDS (two passes of 144 KB), G^T dU G per thread,
//              partial dw of this K slice stored as [slice][co][ky][kx][ci]; the bias gradient (column sums of dy) is accumulated by wave 0 of the
//              workgroups of input-channel block 0 from the blocks it reads anyway.  wgrad_fused_finish_kernel adds the slices in order.
//
// LDS layout of a stage (bytes): x strip  [6 rows][24 pixel slots, 22 used][32 ci] floats, slot(c) = c + (c >> 2)  (one empty slot after every 4 pixels);
//                                dy strip [2 co halves][4 rows][20 pixel slots][32 co] floats, the same slot rule.
// A tile is 640 bytes wide in both: the two lane halves (tiles h = 0 / 1) are 128 B (mod 256) apart: conflict-free; tiles h and h + 2 are
// 1280 B = 5 x 256 apart: one ds_read2st64_b32 fetches the packed pair; one lane base register serves both strips.
#include "common.h"
#include <cstdlib>
#include <algorithm>

namespace aclgan {

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WCO = 64, WCI = 32;                // channels per workgroup
constexpr int XROW = 24 * 128, XB = 6 * XROW;    // x strip: rows of 3 x 1 KB copy chunks (22 slots used): 18 432 bytes
constexpr int DROW = 20 * 128, DHALF = 4 * DROW, DB = 2 * DHALF;      // dy strip: two 32-channel halves of [4 rows][20 slots]: 20 480 bytes
constexpr int STAGE = XB + DB;
constexpr int NST = 4;
constexpr int E_BYTES = 36 * 32 * 32 * 4;        // epilogue exchange buffer (one 32-output-channel half)
constexpr unsigned int OOBV = 0x7ffffff0u;

struct WgP {
    const float* x; const float* dy; float* part; float* partdb;      // part[ks][Co][9][Ci], partdb[ks][Co]
    int B, H, W, Ci, Co, TY, TXS, G, ks, gper, nco, nci, want_db;
    long long xbytes, dybytes;
};

// the scalars the K loop needs, passed BY VALUE (a reference to the kernel-argument struct kept the whole struct in scratch memory here:
// every field came back as a vector register and every buffer load was wrapped in a readfirstlane waterfall loop)
struct WgS { const float* x; const float* dy; int xbytes, dybytes, H, W, Ci, Co, TY, TXS; };

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

// LDS reads of the K loop are written as ds_read2st64_b32 by hand: left to the compiler, the two halves of a packed pair were combined with
// OTHER reads (ds_read2_b32 across pairs) and reassembled with ~80 v_mov per group, and addresses beyond the 16-bit offset field cost a VALU add
// each -- all of it serial time next to the fp32 MFMAs.  base: one of four opaque lane bases (window x 128-byte parity), o0 / o1 in 256-byte units.
__device__ __forceinline__ f32x2 lds_pair(unsigned int base, int o0, int o1) {
    f32x2 v;
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(base), "i"(o0), "i"(o1));
    return v;
}
constexpr int WIN = 64000;                       // LDS window of one base register (250 x 256 bytes)
#define WG_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

constexpr int vmcnt_imm(int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); }      // s_waitcnt vmcnt(n), other counters unconstrained
#define WG_LDS_PTR(a) ((__attribute__((address_space(3))) void*)(size_t)(a))

// One wave's share of the K loop over the groups [g0, g1) of this workgroup's slice.
template <int WI, int WJ>
__device__ __forceinline__ void wg_wave(const WgS p, char* smem, const int tid, const int lane, const int co0, const int ci0, const int g0, const int g1,
                                        const bool want_db, f32x16 (&acc)[9][2], f32x2 (&bs)[2]) {
    constexpr int WV = WI * 2 + WJ;
    constexpr int NXI = WV < 3 ? 6 : 0, NDI = WV < 3 ? 4 : 8, NLD = NXI + NDI;      // global -> LDS copies per group: x rows / dy chunks of this wave
    const int l31 = lane & 31, h = lane >> 5;
    const int Ci4 = p.Ci * 4;
    const int n = g1 - g0;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.dybytes, 0x00020000);
    const TfK tk = {opq2(4.f), opq2(-5.f), opq2(-4.f), opq2(2.f), opq2(-2.f)};
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned int lds0 = (unsigned int)(size_t)(lds_char*)smem;

    // ---- staging: global -> LDS copies (buffer_load_dwordx4 ... lds: 64 lanes x 16 bytes = 1 KB of LDS in lane order, no registers) ----
    // x: wave w < 3 copies chunk w (pieces 64 w .. 64 w + 63 = (slot, channel quad) pairs) of each of the six rows; a slot is pixel column
    // c = slot - slot / 5 of the strip (image column 16 sx - 1 + c, reflected at the image borders) or empty (slot % 5 == 4, slots >= 22: the
    // lane reads out of bounds = zeros).  The row offset is scalar; at the left / right image border the column part moves by a per-lane delta.
    // (A select between precomputed variants AS VARIABLES became a select of closure field offsets -- a dynamically indexed closure that kept
    //  every captured variable of the kernel in scratch memory; deltas added under a scalar condition do not.)
    unsigned int xvo_m = OOBV;
    int xd_first = 0, xd_last = 0;
    if (NXI) {
        const int slot = WV * 8 + (lane >> 3), xq = lane & 7, xc = slot - slot / 5;
        if (slot % 5 != 4 && slot < 22) {
            xvo_m = (unsigned int)(xc * Ci4 + xq * 16);
            xd_first = xc == 0 ? Ci4 : -Ci4;          // strip at the left border: column -1 -> 1, the others relative to column 0
            xd_last = xc == 17 ? -2 * Ci4 : 0;        // strip at the right border: column W -> W - 2
        }
    }
    // dy: 20 chunks of [half][row][20 slots][8 quads]; waves 0..2 copy chunks w, w + 3, w + 6, w + 9, wave 3 chunks 12..19
    unsigned int dvo[NDI];
#pragma unroll
    for (int i = 0; i < NDI; ++i) {
        const int P = (WV < 3 ? WV + 3 * i : 12 + i) * 64 + lane;
        const int half = P / 640, rem = P % 640, row = rem / 160, rem2 = rem % 160, slot = rem2 >> 3, q = rem2 & 7;
        dvo[i] = slot % 5 == 4 ? OOBV : (unsigned int)((row * p.W + slot - slot / 5) * p.Co * 4 + half * 128 + q * 16);
    }
    // lane bases of the LDS reads, one register per 64 000-byte window and 128-byte parity (every read is base + immediate); opaque so that the
    // compiler cannot re-associate them
    unsigned int rb[6];
#pragma unroll
    for (int w = 0; w < 3; ++w) {
        rb[2 * w] = lds0 + h * 640 + l31 * 4 + w * WIN; rb[2 * w + 1] = rb[2 * w] + 128;
        asm volatile("" : "+v"(rb[2 * w]), "+v"(rb[2 * w + 1]));
    }

    // ---- the staging cursor: group cg = (image cb_, tile row cty, strip csx) ----
    int cg = g0, csx = g0 % p.TXS, cty = (g0 / p.TXS) % p.TY, cb_ = g0 / (p.TXS * p.TY);
    // copies of the cursor's group into stage st: NLD vector-memory instructions
#define WG_STAGE(st)                                                                                                               \
    do {                                                                                                                           \
        const bool first_ = csx == 0, last_ = csx == p.TXS - 1;                                                                    \
        if (NXI) {                                                                                                                 \
            const unsigned int xvo = xvo_m + (unsigned int)((first_ ? xd_first : 0) + (last_ ? xd_last : 0));                      \
            const int RS_ = p.W * Ci4;                                                                                             \
            const int xb_ = (cb_ * p.H + 4 * cty - 1) * RS_ + (first_ ? 0 : (16 * csx - 1) * Ci4) + ci0 * 4;      /* row 4 ty - 1 */ \
            _Pragma("unroll") for (int r = 0; r < NXI; ++r) {                                                                      \
                int so = xb_ + r * RS_;                                                                                            \
                if (r == 0) so = xb_ + (cty == 0 ? 2 * RS_ : 0);                          /* row -1 -> 1 */                        \
                if (r == 5) so = xb_ + (cty == p.TY - 1 ? 3 * RS_ : 5 * RS_);             /* row H -> H - 2 */                     \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, WG_LDS_PTR(lds0 + (st) * STAGE + r * XROW + WV * 1024), 16, xvo,      \
                                                         __builtin_amdgcn_readfirstlane(so), 0, 0);                                \
            }                                                                                                                      \
        }                                                                                                                          \
        const int db_ = __builtin_amdgcn_readfirstlane(((cb_ * p.H + 4 * cty) * p.W + 16 * csx) * p.Co * 4 + co0 * 4);             \
        _Pragma("unroll") for (int i = 0; i < NDI; ++i)                                                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, WG_LDS_PTR(lds0 + (st) * STAGE + XB + (WV < 3 ? WV + 3 * i : 12 + i) * 1024), 16, dvo[i], db_, 0, 0); \
    } while (0)
#define WG_ADVANCE()      /* (past the last group: stay there, the copies are redundant) */                                        \
    do {                                                                                                                           \
        if (cg + 1 < g1) {                                                                                                         \
            ++cg;                                                                                                                  \
            if (++csx == p.TXS) { csx = 0; if (++cty == p.TY) { cty = 0; ++cb_; } }                                                \
        }                                                                                                                          \
    } while (0)

    // ---- LDS reads of one group: the 4 x 4 block of dy (per 32-channel half i) and the 5 x 5 sub-patch of x of the lane's tile pair ----
    auto rdp = [&](int A) __attribute__((always_inline)) {      // A: byte offset of the h = 0 element (a multiple of 128); the pair's second half: + 1280
        const int par = (A >> 7) & 1, Ae = A - par * 128, w = Ae / WIN, o0 = (Ae - w * WIN) >> 8;
        return lds_pair(rb[2 * w + par], o0, o0 + 5);
    };
    f32x2 dyv[4][4];      // [pixel column][pixel row]
    auto rd_dy = [&](int st, int i, int cc) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dyv[cc][r] = rdp(st * STAGE + XB + i * DHALF + r * DROW + cc * 128);
    };
    f32x2 d[5][5];        // [column][row]
    auto rd_x = [&](int st, int c) __attribute__((always_inline)) {
        const int cc = WJ + c;
#pragma unroll
        for (int r = 0; r < 5; ++r) d[c][r] = rdp(st * STAGE + (WI + r) * XROW + (cc + (cc >> 2)) * 128);
    };
    // the hand-written reads are invisible to the compiler's wait-count insertion: wait here, tied to the values the next burst consumes
    auto wait_dy = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(dyv[0][0]), "+v"(dyv[0][1]), "+v"(dyv[0][2]), "+v"(dyv[0][3]), "+v"(dyv[1][0]), "+v"(dyv[1][1]), "+v"(dyv[1][2]), "+v"(dyv[1][3]),
                       "+v"(dyv[2][0]), "+v"(dyv[2][1]), "+v"(dyv[2][2]), "+v"(dyv[2][3]), "+v"(dyv[3][0]), "+v"(dyv[3][1]), "+v"(dyv[3][2]), "+v"(dyv[3][3]));
    };
    auto wait_x = [&](int c0) __attribute__((always_inline)) {      // columns 0, 1, 2 (c0 = 0) or 3, 4 (c0 = 3)
        if (c0 == 0)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[0][2]), "+v"(d[0][3]), "+v"(d[0][4]), "+v"(d[1][0]), "+v"(d[1][1]), "+v"(d[1][2]), "+v"(d[1][3]),
                           "+v"(d[1][4]), "+v"(d[2][0]), "+v"(d[2][1]), "+v"(d[2][2]), "+v"(d[2][3]), "+v"(d[2][4]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(d[3][0]), "+v"(d[3][1]), "+v"(d[3][2]), "+v"(d[3][3]), "+v"(d[3][4]), "+v"(d[4][0]), "+v"(d[4][1]), "+v"(d[4][2]), "+v"(d[4][3]),
                           "+v"(d[4][4]));
    };
    auto tf_dy = [&](f32x2 (&M)[9], int i, bool count) __attribute__((always_inline)) {      // 28 packed operations
        f32x2 t[3][4];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) a3g<WI>(tk, dyv[cc], t[0][cc], t[1][cc], t[2][cc]);
#pragma unroll
        for (int il = 0; il < 3; ++il) a3g<WJ>(tk, t[il], M[il * 3], M[il * 3 + 1], M[il * 3 + 2]);
        if (WV == 0 && count) {      // bias gradient: wave 0 of the workgroups of input-channel block 0 sums the raw block (count: uniform)
            f32x2 sm = dyv[0][0];
#pragma unroll
            for (int e = 1; e < 16; ++e) sm += dyv[e >> 2][e & 3];
            bs[i] += sm;
        }
    };
    // B^T d B in two parts (48 packed operations): the column transform of columns 0..2 runs as its own burst as soon as they are read, so that
    // only 38 registers of the sub-patch are ever live (t of three columns + two raw columns) instead of 50
    f32x2 tx[3][5];
    auto tf_x1 = [&](int c0, int c1) __attribute__((always_inline)) {
#pragma unroll
        for (int c = c0; c < c1; ++c) bt3g<WI>(tk, d[c], tx[0][c], tx[1][c], tx[2][c]);
    };
    auto tf_x2 = [&](f32x2 (&V)[9]) __attribute__((always_inline)) {
#pragma unroll
        for (int il = 0; il < 3; ++il) bt3g<WJ>(tk, tx[il], V[il * 3], V[il * 3 + 1], V[il * 3 + 2]);
    };
    // 18 accumulator tiles = 288 registers: 16 tiles fill the 256 AGPRs, the MFMAs of the last two are written in their VGPR form by hand
    auto mfma1 = [&](int fi, int i, float a, float b) __attribute__((always_inline)) {
        if (fi == 8) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[fi][i]) : "v"(a), "v"(b));
        else acc[fi][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[fi][i], 0, 0, 0);
    };

    // ---- prologue: groups 0, 1, 2 copied to stages 0, 1, 2 (group 0 landed), operands of group 0 transformed ----
#pragma unroll
    for (int gq = 0; gq < 3; ++gq) {
        WG_STAGE(gq);
        WG_ADVANCE();
    }
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(2 * NLD));
    WG_BARRIER();
    f32x2 Ma[2][9], Va[9], Mb[2][9], Vb[9];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) rd_dy(0, i, cc);
        wait_dy();
        tf_dy(Ma[i], i, want_db);
    }
#pragma unroll
    for (int c = 0; c < 5; ++c) rd_x(0, c);
    wait_x(0); wait_x(3);
    tf_x1(0, 5);
    tf_x2(Va);
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(NLD));      // group 1 has landed ...
    WG_BARRIER();                                    // ... for every wave

    // ---- main loop: iteration j multiplies the operands of group j (registers) while the strips of group j + 1 are read from stage (j + 1) % 4 and
    // transformed in bursts, and group j + 3 is copied into stage (j + 3) % 4 (read last in iteration j - 2).  At the end of the iteration the
    // wave waits for its copies of group j + 2 (all but the newest NLD), then the barrier publishes them: a copy has a whole iteration to land ----
    auto iter = [&](int j, int R, f32x2 (&Mc)[2][9], f32x2 (&Vc)[9], f32x2 (&Mn)[2][9], f32x2 (&Vn)[9]) __attribute__((always_inline)) {
        const bool count = want_db && j + 1 < n;
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            mfma1(s, 0, Mc[0][s].x, Vc[s].x);
            mfma1(s, 1, Mc[1][s].x, Vc[s].x);
            mfma1(s, 0, Mc[0][s].y, Vc[s].y);
            mfma1(s, 1, Mc[1][s].y, Vc[s].y);
            if (s == 2) { wait_dy(); tf_dy(Mn[0], 0, count); __builtin_amdgcn_sched_barrier(0); }
            if (s == 5) { wait_dy(); tf_dy(Mn[1], 1, count); __builtin_amdgcn_sched_barrier(0); }
            if (s == 7) { wait_x(0); tf_x1(0, 3); __builtin_amdgcn_sched_barrier(0); }
            if (s == 0) { rd_dy(R, 0, 0); rd_dy(R, 0, 1); }
            if (s == 1) { rd_dy(R, 0, 2); rd_dy(R, 0, 3); }
            if (s == 3) { rd_dy(R, 1, 0); rd_dy(R, 1, 1); }
            if (s == 4) { rd_dy(R, 1, 2); rd_dy(R, 1, 3); }
            if (s == 5) { rd_x(R, 0); rd_x(R, 1); }
            if (s == 6) rd_x(R, 2);
            if (s == 7) { rd_x(R, 3); rd_x(R, 4); }
            if (s == 1) WG_STAGE((R + 2) & 3);
            __builtin_amdgcn_sched_barrier(0);
        }
        wait_x(3);
        tf_x1(3, 5);
        tf_x2(Vn);
        WG_ADVANCE();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(vmcnt_imm(NLD));
        WG_BARRIER();
    };
    for (int j = 0; j < n; j += 4) {
        iter(j, 1, Ma, Va, Mb, Vb);
        if (j + 1 < n) iter(j + 1, 2, Mb, Vb, Ma, Va);
        if (j + 2 < n) iter(j + 2, 3, Ma, Va, Mb, Vb);
        if (j + 3 < n) iter(j + 3, 0, Mb, Vb, Ma, Va);
    }
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));      // the redundant copies past the last group must not land in the epilogue's buffer
#undef WG_STAGE
#undef WG_ADVANCE
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
    f32x2 bs[2] = {(f32x2)(0.f), (f32x2)(0.f)};
    const bool want_db = p.want_db && cib == 0;
    const WgS ps = {p.x, p.dy, (int)p.xbytes, (int)p.dybytes, p.H, p.W, p.Ci, p.Co, p.TY, p.TXS};
    if (g1 > g0) {
        if (wave == 0) wg_wave<0, 0>(ps, smem, tid, lane, co0, ci0, g0, g1, want_db, acc, bs);
        else if (wave == 1) wg_wave<0, 1>(ps, smem, tid, lane, co0, ci0, g0, g1, want_db, acc, bs);
        else if (wave == 2) wg_wave<1, 0>(ps, smem, tid, lane, co0, ci0, g0, g1, want_db, acc, bs);
        else wg_wave<1, 1>(ps, smem, tid, lane, co0, ci0, g0, g1, want_db, acc, bs);
    }

    // ---- epilogue: bias partial (16 pixel columns per channel quad, summed in order), then the frequency exchange and G^T dU G ----
    float* Es = reinterpret_cast<float*>(smem);
    if (want_db) {      // wave 0's lanes hold, per 32-channel half, the column sums over their two tiles of every group: add the lane halves
        __syncthreads();
        if (wave == 0) { Es[lane] = bs[0].x + bs[0].y; Es[64 + lane] = bs[1].x + bs[1].y; }
        __syncthreads();
        if (tid < 64) p.partdb[(size_t)slice * p.Co + co0 + tid] = Es[(tid >> 5) * 64 + (tid & 31)] + Es[(tid >> 5) * 64 + 32 + (tid & 31)];
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

std::atomic<int> g_wgrad_fused{-1};

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

// tuning / test knob behind aclgan_set_tuning("wino_wgrad_fused", v): 0 = the pipeline of conv_wino.hip, 1 / 2 = the fused kernel wherever the
// shape is eligible (it pays at every grid size measured); returns the previous value.  ACLGAN_WINO_WGRAD_FUSED sets the default.
int wino_wgrad_fused_mode() {
    int v = g_wgrad_fused.load();
    if (v < 0) { const char* e = getenv("ACLGAN_WINO_WGRAD_FUSED"); v = e ? atoi(e) : 1; if (v < 0 || v > 2) v = 1; g_wgrad_fused.store(v); }
    return v;
}
int set_wino_wgrad_fused(int v) { const int old = wino_wgrad_fused_mode(); g_wgrad_fused.store((v < 0 || v > 2) ? 1 : v); return old; }

// 3x3 stride-1 reflect-pad-1 layers with W a multiple of 16, H of 4, Cout of 64, Cin of 32
namespace {
bool wg_shape_ok(const ConvGeom& g) {
    return g.k == 3 && g.s == 1 && g.p == 1 && g.up == 0 && g.Ho % 4 == 0 && g.Wo % 16 == 0 && g.Wo >= 16 && g.Co % WCO == 0 && g.Ci % WCI == 0 && g.Hi >= 4 &&
           g.B >= 1 && (long long)g.B * g.Hi * g.Wi * std::max(g.Ci, g.Co) * 4 < 0x7fffffe0ll;
}
}  // namespace
bool wino_wgrad_fused_ok(const ConvGeom& g) {
    const int m = wino_wgrad_fused_mode();
    if (m == 0) return false;
    const bool shape = wg_shape_ok(g);
    // MEASURED (scripts/probe_wgrad_fused.py, 256 -> 256 channels on 64 x 64 maps, back to back, finish launch included): fused 35 / 46 / 57 / 67 / 91 /
    // 116 us at B = 1 / 2 / 3 / 4 / 6 / 8 against 57 / 80 / 99 / 117 / 127 / 176 us for the pipeline of conv_wino.hip: the K split adapts the
    // workgroup count to the tile count, so unlike the fused forward kernel there is no small-grid regime where the pipeline wins.
    return shape;
}
size_t wino_wgrad_fused_scratch_bytes(const ConvGeom& g) {
    if (!wg_shape_ok(g)) return 0;
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
