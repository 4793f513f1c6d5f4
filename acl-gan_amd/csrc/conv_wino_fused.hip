// conv_wino_fused.hip -- Winograd F(4x4, 3x3) as ONE launch (round 4): input transform, the 36 frequency GEMMs and the output
// transform in a single kernel, so neither V = B^T d B nor M = U (.) V ever exists in HBM (conv_wino.hip: three launches and
// 2 x 75.5 MB of plane traffic per ResBlock convolution -- 90 GB of the 299 GB the fp32 step moved in round 3).
// Replaces ReflectionPad2d(1) + Conv2d(3x3) of the ResBlocks (networks.py:297-310, 366-370) forward, and -- with zero padding and
// the flipped / transposed filter -- the interior of their input gradient.
//
// Work decomposition (fp32, v_mfma_f32_32x32x2_f32):
//   workgroup = 4 waves (one per SIMD, up to 512 registers each) = a block of 8 x 4 output tiles (32 x 16 pixels) x 64 output
//               channels x ALL 36 frequencies; the K loop runs over the input channels.
//   wave (wi, wj) of the 2 x 2 wave grid owns the 3 x 3 frequency block rows 3wi..3wi+2, columns 3wj..3wj+2 of the 6 x 6 Winograd
//               domain: 9 frequencies x (32 tiles x 64 channels) = 18 accumulator tiles of 32 x 32 = 288 accumulator registers.
//   A operand  = the transformed input.  The raw 18 x 34 pixel patch of the tile block is staged in LDS (KC channels at a time,
//               double-buffered); every lane reads the 5 x 5 sub-patch its wave's frequency block needs for ITS tile (MFMA row =
//               tile, lane half = channel pair) and transforms it in registers: the separable B^T d B restricted to 3 rows x 3
//               columns costs 48 packed VALU operations per 36 MFMAs.  The transform output IS the MFMA A fragment: V never
//               touches LDS.
//   B operand  = U = G g G^T, written ONCE per update by wino_filter_frag_kernel in MFMA-fragment order: a wave's 18 B
//               fragments of a 4-channel step are 18 contiguous 512-byte global loads straight into registers -- no LDS, no
//               barrier, no redundancy between the waves (each owns different frequencies).
//   epilogue   = the 36 frequency accumulators of a (tile, channel) pair live in four waves: exchanged through LDS (two passes
//               of 144 KB), A^T M A + bias + activation per thread, 128-byte rows stored to y, and the (mean, M2) of each 4 x 4
//               output tile = the chunk partials of the following normalisation layer.
//
// LDS patch layout: units of 8 bytes (one channel pair), unit(q, r, c) = (q * 18 + r) * 42 + (c & 3) * 9 + (c >> 2): the 32 lanes of
// a ds_read_b64 group (8 tile columns x 4 tile rows) read units const + 168 * ty + tx = const + 8 ty + tx (mod 32): conflict-free.
#include "common.h"
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <type_traits>

namespace aclgan {

const WinoUCache* wino_ucache();

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TBX = 8, TBY = 4;                  // output tiles per workgroup block
constexpr int PR = 4 * TBY + 2, PC = 4 * TBX + 2, NPIX = PR * PC;     // 18 x 34 input pixels
constexpr int RS = 42;                           // row stride in 8-byte units (36 used); 4 * RS = 8 (mod 32)
constexpr int QS = PR * RS;                      // units per channel-pair plane
constexpr int NBC = 64;                          // output channels per workgroup
constexpr unsigned int OOBV = 0x7ffffff0u;       // buffer offset that reads zero
constexpr int M_BYTES = 36 * 32 * 32 * 4;        // epilogue exchange buffer (one 32-channel half)

// Views: a strided window onto an NHWC tensor [B][HS][WS][C]: logical pixel (iy, ix) lives at physical row y0 + s * iy, column x0 + s * ix.
// Identity for the ResBlock layers; the sub-pixel phases of the upsample + 5x5 layers are stride-2 views of the hi-res map (conv_wino.hip).
struct WfP {
    const float* x; const float* Uf; const float* bias; float* y; float2* stats;
    int B, Cin, Cout;                                  // Cin: input channels of ONE K phase
    int IH, IW, ivs, ivy0, ivx0, IHS, IWS, off, reflect;      // input view (logical extent IH x IW), patch offset, padding mode (0 zero, 1 reflect, 2 edge)
    int nkph;                                          // K phases: the K loop runs over nkph x Cin channels (input gradient of the sub-pixel layers:
    int kph_xoff[4];                                   //   the four phase views of dy, byte offset of view ph relative to view 0) ...
    int uphase;                                        //   ... with U of K phase / grid phase ph starting uphase bytes after the previous one
    int OH, OW, ovs, OHS, OWS, ovy0[4], ovx0[4];       // output view per GRID phase (blockIdx.y; forward of the sub-pixel layers: 4), logical extent
    int TY, TX, NBY, NBX, act, accumulate, ncb, ntb, xcdmap;
    long long xbytes, ubytes;
};


__device__ __forceinline__ int reflf(int v, int n) {
    v = v < 0 ? -v : v;
    return v >= n ? 2 * (n - 1) - v : v;
}
__device__ __forceinline__ float actf(float v, int act) {
    if (act == ACLGAN_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACLGAN_ACT_LRELU) return v > 0.f ? v : 0.2f * v;
    if (act == ACLGAN_ACT_TANH) return tanhf(v);
    return v;
}

__device__ __forceinline__ void at4f(const float (&m)[6], float (&y)[4]) {
    y[0] = m[0] + m[1] + m[2] + m[3] + m[4];
    y[1] = m[1] - m[2] + 2.f * (m[3] - m[4]);
    y[2] = m[1] + m[2] + 4.f * (m[3] + m[4]);
    y[3] = m[1] - m[2] + 8.f * (m[3] - m[4]) + m[5];
}
__device__ __forceinline__ void g6f(const float (&g)[3], float (&u)[6]) {
    u[0] = 0.25f * g[0];
    u[1] = -(g[0] + g[1] + g[2]) * (1.f / 6.f);
    u[2] = -(g[0] - g[1] + g[2]) * (1.f / 6.f);
    u[3] = g[0] * (1.f / 24.f) + g[1] * (1.f / 12.f) + g[2] * (1.f / 6.f);
    u[4] = g[0] * (1.f / 24.f) - g[1] * (1.f / 12.f) + g[2] * (1.f / 6.f);
    u[5] = g[2];
}

// U in fragment order: Uf[cb][kq][wave g][fi][lane][j][e]  (floats): a lane's two 32-channel halves j of one frequency are ONE 16-byte load
//   row R (forward: cout, dgrad: cin) = cb * 64 + j * 32 + (lane & 31);  k (forward: cin, dgrad: cout) = kq * 4 + 2 * (lane >> 5) + e
//   frequency (i, jf): wave g = (i / 3) * 2 + jf / 3, fi = (i % 3) * 3 + jf % 3
// blockIdx.y = filter of a batch (w + y * wstr -> Uf + y * ustr floats): the merged phase filters of a sub-pixel layer, or -- round 5 -- all
// equally shaped ResBlock filters of an encoder / decoder (wstr = their distance in the flat parameter buffer)
// mode 0: a 3x3 filter w[Co][3][3][Ci] (flip: the flipped, transposed filter of the input gradient).
// mode 1 / 2 (round 6): the polyphase 3x3 embeddings of a 4x4 STRIDE-2 filter w[Co][4][4][Ci] (networks.py:41, 216-221, 236-241), blockIdx.y = phase
// (py, px).  A 4x4 stride-2 pad-1 convolution is the sum over the four input parity phases of a 2-tap-per-axis stride-1 convolution of the
// decimated input; each 2-tap filter sits in a 3-tap window (one tap structurally zero) so that the F(4x4,3x3) kernel runs it: 36 multiplies
// per 16 outputs and phase instead of 64.  Tap t of the window [u - 1, u, u + 1] reads filter row s2k4_tap(...) (or nothing):
//   mode 1, forward   (rows = cout, k = cin): even input rows: y[u] += w1 E[u] + w3 E[u+1];  odd rows: y[u] += w0 O[u-1] + w2 O[u]
//   mode 2, input gradient (rows = cin, k = cout; phase = parity of the dx row): dx[2m] = w3 dy[m-1] + w1 dy[m];  dx[2m+1] = w2 dy[m] + w0 dy[m+1]
__device__ __forceinline__ int s2k4_tap(int mode, int par, int t) {
    if (mode == 1) return par == 0 ? (t == 0 ? -1 : (t == 1 ? 1 : 3)) : (t == 0 ? 0 : (t == 1 ? 2 : -1));
    return par == 0 ? (t == 0 ? 3 : (t == 1 ? 1 : -1)) : (t == 0 ? -1 : (t == 1 ? 2 : 0));
}
__global__ void __launch_bounds__(256) wino_filter_frag_kernel(const float* __restrict__ w, float* __restrict__ Uf, int Co, int Ci, int flip,
                                                               int64_t wstr, int64_t ustr, int mode) {
    const int R = flip ? Ci : Co, K = flip ? Co : Ci, KQ = K >> 2;
    const int64_t n = (int64_t)R * K;
    w += (size_t)blockIdx.y * wstr;
    Uf += (size_t)blockIdx.y * ustr;
    const int py = (int)(blockIdx.y >> 1) & 1, px = (int)blockIdx.y & 1;      // (modes 1, 2)
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * 256) {
        // consecutive threads -> consecutive rows of one k (the 36 stores of a wave then fill 128-byte runs of the fragment layout)
        const int row = (int)(idx % R), kk = (int)(idx / R);
        const int co = flip ? kk : row, ci = flip ? row : kk;
        float t[6][3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            float c[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                if (mode == 0) {
                    const int sy = flip ? 2 - ky : ky, sx = flip ? 2 - kx : kx;
                    c[ky] = w[((size_t)(co * 3 + sy) * 3 + sx) * Ci + ci];
                } else {
                    const int sy = s2k4_tap(mode, py, ky), sx = s2k4_tap(mode, px, kx);
                    c[ky] = (sy >= 0 && sx >= 0) ? w[((size_t)(co * 4 + sy) * 4 + sx) * Ci + ci] : 0.f;
                }
            }
            float o[6];
            g6f(c, o);
#pragma unroll
            for (int a = 0; a < 6; ++a) t[a][kx] = o[a];
        }
        const int cb = row >> 6, j = (row >> 5) & 1, l31 = row & 31;
        const int kq = kk >> 2, h = (kk >> 1) & 1, e = kk & 1;
        float* base = Uf + ((size_t)cb * KQ + kq) * (4 * 9 * 2 * 128) + (size_t)(h * 32 + l31) * 4 + j * 2 + e;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            float o[6];
            g6f(t[a], o);
#pragma unroll
            for (int jf = 0; jf < 6; ++jf) {
                const int g = (a / 3) * 2 + jf / 3, fi = (a % 3) * 3 + jf % 3;
                base[(size_t)(g * 9 + fi) * 256] = o[jf];
            }
        }
    }
}

__device__ __forceinline__ f32x2 opaque2(float v) {      // a constant pair the compiler cannot fold: keeps the transform in v_pk_* form
    f32x2 r = {v, v};
    asm volatile("" : "+v"(r));
    return r;
}
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x4 fma4(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x4 opaque4(float v) {
    f32x4 r = {v, v, v, v};
    asm volatile("" : "+v"(r));
    return r;
}
struct BtK { f32x2 k4, k5n, k4n, k2, k2n; };
// three rows of B^T d: W = 0: rows 0, 1, 2 from d0..d4;  W = 1: rows 3, 4, 5 from d1..d5  (x = the five inputs that set needs): 6 packed operations
template <int W>
__device__ __forceinline__ void bt3(const BtK& k, const f32x2 (&x)[5], f32x2& o0, f32x2& o1, f32x2& o2) {
    if (W == 0) {
        const f32x2 pp = fma2(k.k4n, x[2], x[4]), qq = fma2(k.k4, x[1], -x[3]);
        o0 = fma2(k.k4, x[0], fma2(k.k5n, x[2], x[4]));
        o1 = pp - qq;
        o2 = pp + qq;
    } else {
        const f32x2 pp = x[3] - x[1], sd = x[2] - x[0];
        o0 = fma2(k.k2, sd, pp);
        o1 = fma2(k.k2n, sd, pp);
        o2 = fma2(k.k4, x[0], fma2(k.k5n, x[2], x[4]));
    }
}

constexpr int KC = 8;                            // input channels per staged chunk (two 4-channel MFMA sub-steps)
constexpr int NP = 5;                            // 16-byte staging pieces per thread and chunk (612 pixels x 2 pieces <= 5 x 256)
constexpr int XBUF = (KC / 2) * QS;              // 8-byte units per patch buffer
constexpr int UD = 17;                           // U fragments are loaded UD slices (almost two sub-steps) ahead of their MFMAs: two register sets

// One wave's share of the K loop.  WI, WJ: the wave's frequency block.
// MEASURED (scripts/microbench/mfma_valu_overlap.hip): v_mfma_f32_32x32x2_f32 runs on the fp32 VALU lanes -- a VALU instruction of the same
// SIMD never hides behind it (each costs its 4 cycles on top of the 64 of an MFMA, plus ~20 cycles per MFMA -> VALU -> MFMA round trip),
// while LDS and memory instructions do.  So the loop body keeps the VALU out of the MFMA stream: addresses are SGPR bases + constant lane
// offsets + immediates (buffer loads, static LDS buffer parity through a 2x unrolled chunk loop), the transform is written in packed
// form (48 v_pk_* per sub-step) and issued as ONE burst per sub-step.
template <int WI, int WJ, int ABL>
__device__ __forceinline__ void wf_wave(const WfP& p, char* smem, const int lane, const int cb, const unsigned int (&go)[NP], const int (&lub)[NP],
                                        f32x16 (&acc)[9][2]) {
    const int nchp = p.Cin / KC, nch = nchp * p.nkph, KQp = p.Cin >> 2;      // chunks per K phase, chunks in all, 4-channel steps per phase
    const int wave = WI * 2 + WJ;
    const int l31 = lane & 31, h = lane >> 5;
    const int lanebase = (h * QS + 4 * (l31 >> 3) * RS + (l31 & 7)) * 8;      // bytes
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.Uf), 0, (int)p.ubytes, 0x00020000);
    const int uvo = lane * 16;                                                 // lane part of a U fragment address (16 bytes: both 32-channel halves)
    const int ustep = 4 * 9 * 2 * 64 * 8;                                      // bytes of one 4-channel step of a channel block
    const int ubase = (cb * KQp * 4 + wave) * (9 * 2 * 64 * 8) + (int)blockIdx.y * p.uphase;      // (blockIdx.y: grid phase)
    // SCALAR byte offsets of a chunk of x (channel offset inside its K phase + the phase view's offset) and of a 4-channel step of U, computed
    // once per sub-step on the scalar unit and pinned to an SGPR.  (Round 4: written as arithmetic on comparison results, the compiler moved
    // them to the VALU and wrapped every buffer load in a readfirstlane waterfall loop -- 12 us per launch.)
    auto xoff = [&](int chunk) __attribute__((always_inline)) {
        int c = min(chunk, nch - 1), xo = 0;
        if (c >= nchp) { c -= nchp; xo = p.kph_xoff[1]; if (c >= nchp) { c -= nchp; xo = p.kph_xoff[2]; if (c >= nchp) { c -= nchp; xo = p.kph_xoff[3]; } } }
        return __builtin_amdgcn_readfirstlane(c * (KC * 4) + xo);
    };
    auto uoff = [&](int kq) __attribute__((always_inline)) {
        int q = min(kq, 2 * nch - 1), so = ubase;
        if (q >= KQp) { q -= KQp; so += p.uphase; if (q >= KQp) { q -= KQp; so += p.uphase; if (q >= KQp) { q -= KQp; so += p.uphase; } } }
        return __builtin_amdgcn_readfirstlane(so + q * ustep);
    };
    const BtK bk = {opaque2(4.f), opaque2(-5.f), opaque2(-4.f), opaque2(2.f), opaque2(-2.f)};

    // ---- staging of the raw patch: the NP pieces of a chunk are issued at window slices 0 .. NP-1 and written to LDS at slices 12 .. 12+NP-1 of
    // the 18-slice window (12 slices = ~1.5 us in flight: an L2 miss to the Infinity Cache has landed; the main loop has the registers) ----
    f32x4 xr[NP];
    auto xissue = [&](int xso, int i) __attribute__((always_inline)) {       // xso = xoff(chunk)
        xr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, go[i], xso, 0));
    };
    auto xwrite = [&](int par, int i) __attribute__((always_inline)) {       // par: parity of the chunk = its LDS buffer (compile time)
        char* dst = smem + par * XBUF * 8 + lub[i];
        *reinterpret_cast<f32x2*>(dst) = (f32x2){xr[i].x, xr[i].y};
        *reinterpret_cast<f32x2*>(dst + QS * 8) = (f32x2){xr[i].z, xr[i].w};
    };
    auto xslot = [&](int xso, int par, int wsl) __attribute__((always_inline)) {
        if (wsl >= 12 && wsl - 12 < NP) xwrite(par, wsl - 12);
        if (wsl < NP) xissue(xso, wsl);
    };

    f32x2 u[2][9][2];      // [sub-step parity][frequency][32-channel half]
    auto uload = [&](int so, int par2, int fi) __attribute__((always_inline)) {        // so = uoff(4-channel step), par2 = its parity
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ru, uvo + (fi & 3) * 1024, so + (fi >> 2) * 4096, 0));
        u[par2][fi][0] = (f32x2){v.x, v.y};
        u[par2][fi][1] = (f32x2){v.z, v.w};
    };
    // the wave's 5 x 5 sub-patch of one sub-step (read from LDS well ahead of the burst that transforms it)
    f32x2 d[5][5];      // [column][row]
    auto rdcol = [&](int par, int ss, int c) __attribute__((always_inline)) {
        const int cc = WJ + c;
        const char* src = smem + par * XBUF * 8 + lanebase;
#pragma unroll
        for (int r = 0; r < 5; ++r) d[c][r] = *reinterpret_cast<const f32x2*>(src + (2 * ss * QS + (WI + r) * RS + (cc & 3) * 9 + (cc >> 2)) * 8);
    };
    auto transform = [&](f32x2 (&V)[9]) __attribute__((always_inline)) {      // 48 packed operations
        f32x2 t[3][5];
#pragma unroll
        for (int c = 0; c < 5; ++c) bt3<WI>(bk, d[c], t[0][c], t[1][c], t[2][c]);
#pragma unroll
        for (int il = 0; il < 3; ++il) bt3<WJ>(bk, t[il], V[il * 3], V[il * 3 + 1], V[il * 3 + 2]);
    };
    // 18 accumulator tiles = 288 registers: 16 tiles fill the 256 AGPRs; the MFMAs of the last two (fi = 8) are written in their VGPR
    // form by hand (the builtin would take the AGPR form for every tile and shuttle tiles between the two files on every iteration)
    // Hazards of the hand-written form (the compiler neither inserts wait states nor s_nops inside an asm statement):
    //   * its result registers are read next by (a) another `v_mfma` on the same accumulator -- back-to-back dependent MFMAs of one size are
    //     interlocked by the hardware (SrcC forwarding), no software wait states -- and (b) VALU / LDS-store instructions of the epilogue, which
    //     come after the K loop's last barrier and >= 18 independent MFMAs (64 cycles each) later: far beyond the 18-pass XDL write -> VALU read
    //     requirement of the ISA;
    //   * its A / B operands come from VALU results (the packed transform) or loads: both are ordinary "v" inputs, so the compiler's own
    //     VALU-write -> MFMA-read handling (it sees the asm's inputs) and its s_waitcnt insertion for the loads apply;
    //   * "+v" ties the accumulator in place; `volatile` keeps the statement's position relative to the sched_barriers of the loop.
    // tests/test_gpu_ops.py::test_winograd_fused_kernels_repeat_launch_stress launches the kernel 800 times per shape next to a busy second
    // stream and compares every result bitwise with the first (advisor, round 4).
    auto mfma1 = [&](int fi, int j, float a, float b) __attribute__((always_inline)) {
        if (fi == 8) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[fi][j]) : "v"(a), "v"(b));
        else acc[fi][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[fi][j], 0, 0, 0);
    };

    // ---- prologue: chunk 0 in LDS, the first window slices of chunk 1 done, the first U fragments in flight, V of sub-step 0 computed ----
    {   // every load of the prologue is issued before the first one is waited for
        f32x4 xp[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) xp[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, go[i], 0, 0));
#pragma unroll
        for (int fi = 0; fi < 9; ++fi) uload(uoff(0), 0, fi);
#pragma unroll
        for (int fi = 0; fi < 8; ++fi) uload(uoff(1), 1, fi);      // (fragment 8 of sub-step 1 is loaded by slice 0 of sub-step 0)
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            *reinterpret_cast<f32x2*>(smem + lub[i]) = (f32x2){xp[i].x, xp[i].y};
            *reinterpret_cast<f32x2*>(smem + lub[i] + QS * 8) = (f32x2){xp[i].z, xp[i].w};
        }
        // = the state after window slices 0 .. 8 of chunk 1: every piece in flight
        const int xso1 = xoff(1);
#pragma unroll
        for (int i = 0; i < NP; ++i) xissue(xso1, i);
    }
    __syncthreads();
    f32x2 Va[9], Vb[9];
#pragma unroll
    for (int c = 0; c < 5; ++c) rdcol(0, 0, c);
    transform(Va);
    if (ABL & 1) {
#pragma unroll
        for (int i = 0; i < 9; ++i) Vb[i] = Va[i];
    }

    // ---- main loop: sub-step (ch, ss) multiplies V(ch, ss) while the patch of the next sub-step is read, U(ch, ss + 1) is loaded and
    // the chunk after next is staged; the transform of the next sub-step is ONE burst behind the last MFMA of the sub-step ----
    auto substep = [&](int ch, int par, int ss, f32x2 (&Vc)[9], f32x2 (&Vn)[9]) __attribute__((always_inline)) {
        const int kq = ch * 2 + ss;
        const int npar = ss == 0 ? par : par ^ 1, ns = ss == 0 ? 1 : 0;      // buffer / sub-step index of the NEXT sub-step's patch
        // staging window of chunk ch + 1: sub-steps (ch - 1, 1), (ch, 0); the last sub-step already stages ch + 2 (into this chunk's buffer)
        const int schunk = ss == 1 ? ch + 2 : ch + 1, spar = ss == 1 ? par : par ^ 1;
        const int wbase = ss == 1 ? 0 : 9;
        const int so_nxt = uoff(kq + 1), so_nx2 = uoff(kq + 2), xso = xoff(schunk);
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            mfma1(s, 0, Vc[s].x, u[ss][s][0].x);
            mfma1(s, 1, Vc[s].x, u[ss][s][1].x);
            mfma1(s, 0, Vc[s].y, u[ss][s][0].y);
            mfma1(s, 1, Vc[s].y, u[ss][s][1].y);
            // slice 0 fetches fragment 8 of the NEXT sub-step, slice s >= 1 fragment s - 1 of the one after (= this parity: just consumed)
            if (!(ABL & 2)) uload(s == 0 ? so_nxt : so_nx2, s == 0 ? (ss ^ 1) : ss, (s + UD) % 9);
            if (s < 5 && !(ABL & 1)) rdcol(npar, ns, s);
            if (!(ABL & 4)) xslot(xso, spar, wbase + s);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(ABL & 1)) transform(Vn);
        __builtin_amdgcn_sched_barrier(0);
        if (ss == 0 && !(ABL & 8)) __syncthreads();
    };
    for (int ch = 0; ch < ((ABL & 16) ? 0 : nch); ch += 2) {
        substep(ch, 0, 0, Va, Vb);
        substep(ch, 0, 1, Vb, Va);
        substep(ch + 1, 1, 0, Va, Vb);
        substep(ch + 1, 1, 1, Vb, Va);
    }
}

// ABL (measurement builds only): bit 0 drops the transforms, 1 the U loads, 2 the patch staging, 3 the barrier of the main loop
template <int ABL = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) wino_fused_kernel(WfP p) {
    constexpr int JN = KC / 4;
    constexpr int NSLOT = NPIX * JN;
    constexpr int MAIN_BYTES = 2 * XBUF * 8;
    __shared__ __attribute__((aligned(16))) char smem[MAIN_BYTES > M_BYTES ? MAIN_BYTES : M_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int cb, tb;
    {
        const int bid = blockIdx.x;
        if (p.xcdmap) { const int per = 8 / p.ncb, xcd = bid & 7, idx = bid >> 3; cb = xcd / per; tb = idx * per + xcd % per; }
        else { cb = bid % p.ncb; tb = bid / p.ncb; }
    }
    const int bx = tb % p.NBX, by = (tb / p.NBX) % p.NBY, b = tb / (p.NBX * p.NBY);
    const int ty0 = by * TBY, tx0 = bx * TBX;
    const int gph = blockIdx.y;                           // grid phase (forward of the sub-pixel layers): its own U and output view
    const int ovy0 = gph == 0 ? p.ovy0[0] : (gph == 1 ? p.ovy0[1] : (gph == 2 ? p.ovy0[2] : p.ovy0[3]));
    const int ovx0 = gph == 0 ? p.ovx0[0] : (gph == 1 ? p.ovx0[1] : (gph == 2 ? p.ovx0[2] : p.ovx0[3]));
    // (The workgroups of an XCD share one slice of U and walk K in lockstep.  MEASURED: starting every tile block at a different input
    //  channel -- so that a U line is a first touch for one workgroup only -- is 25 % SLOWER: 128 against 103.5 us; lockstep is what
    //  keeps the slice's lines hot in the XCD's L2.)

    // gather offsets of this thread's staging pieces (the same for every chunk: only the channel offset moves): registers
    unsigned int go[NP];
    int lub[NP];                                          // LDS byte offset of the piece's first channel pair inside a patch buffer
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int s = i * 256 + tid;
        go[i] = OOBV;
        lub[i] = 36 * 8;                                  // padding unit of row 0: never read
        if (s < NSLOT) {
            const int pix = s / JN, j = s - pix * JN;
            const int r = pix / PC, c = pix - r * PC;
            int iy = 4 * ty0 + p.off + r, ix = 4 * tx0 + p.off + c;
            if (p.reflect == 1) { iy = reflf(iy, p.IH); ix = reflf(ix, p.IW); }
            else if (p.reflect == 2) { iy = min(max(iy, 0), p.IH - 1); ix = min(max(ix, 0), p.IW - 1); }      // edge replication (parity phases of a stride-2 layer)
            if ((unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW)
                go[i] = (unsigned int)((((size_t)b * p.IHS + p.ivy0 + p.ivs * iy) * p.IWS + p.ivx0 + p.ivs * ix) * p.Cin * 4 + j * 16);
            lub[i] = (2 * j * QS + r * RS + (c & 3) * 9 + (c >> 2)) * 8;
        }
    }

    f32x16 acc[9][2];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (ABL & 512) p.y[tid] = 0.f;
    if (wave == 0) wf_wave<0, 0, ABL>(p, smem, lane, cb, go, lub, acc);
    else if (wave == 1) wf_wave<0, 1, ABL>(p, smem, lane, cb, go, lub, acc);
    else if (wave == 2) wf_wave<1, 0, ABL>(p, smem, lane, cb, go, lub, acc);
    else wf_wave<1, 1, ABL>(p, smem, lane, cb, go, lub, acc);

    // ---- epilogue: exchange the frequency accumulators through LDS (two passes of 16 tiles x 64 channels x 36 frequencies = 144 KB),
    // output transform with four channels per thread (ds_read_b128, 256-byte rows of y per 16 lanes), bias / activation / statistics ----
    float* Ms = reinterpret_cast<float*>(smem);
    const int wi = wave >> 1, wj = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int tl = tid >> 4, c4 = tid & 15;
    const int n = cb * NBC + c4 * 4;
    const f32x4 bv = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : (f32x4)(0.f);
    const float slope = p.act == ACLGAN_ACT_RELU ? 0.f : (p.act == ACLGAN_ACT_LRELU ? 0.2f : 1.f);      // act(v) = max(v, 0) + slope * min(v, 0)
    const f32x4 k2 = opaque4(2.f), k4 = opaque4(4.f), k8 = opaque4(8.f);
#pragma unroll
    for (int P = 0; P < ((ABL & 512) ? 0 : 2); ++P) {
        __syncthreads();                          // the patch buffers / the previous half are no longer read
#pragma unroll
        for (int fi = 0; fi < 9; ++fi) {
            const int f = (3 * wi + fi / 3) * 6 + 3 * wj + fi % 3;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) {
                    const int tr = (r8 & 3) + 8 * (r8 >> 2) + 4 * h;      // tile of this half
                    Ms[(f * 16 + tr) * 64 + j * 32 + l31] = acc[fi][j][8 * P + r8];
                }
        }
        __syncthreads();
        if (ABL & 256) continue;
        const int tile = 16 * P + tl;
        const int ty = ty0 + (tile >> 3), tx = tx0 + (tile & 7);
        const bool live = ty < p.TY && tx < p.TX;
        const int rows = live ? min(4, p.OH - 4 * ty) : 0, cols = live ? min(4, p.OW - 4 * tx) : 0;      // ragged last tiles of a view
        float* ybase = p.y + (((size_t)b * p.OHS + ovy0 + p.ovs * 4 * (live ? ty : 0)) * p.OWS + ovx0 + p.ovs * 4 * (live ? tx : 0)) * p.Cout + n;
        const size_t yrs = (size_t)p.ovs * p.OWS * p.Cout, ycs = (size_t)p.ovs * p.Cout;
        // A^T m along one axis, factored: 10 vector operations (y0 = m0 + s12 + s34, y1 = d12 + 2 d34, y2 = s12 + 4 s34, y3 = d12 + 8 d34 + m5)
        auto at4v = [&](const f32x4 (&m)[6], f32x4& y0, f32x4& y1, f32x4& y2, f32x4& y3) __attribute__((always_inline)) {
            const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
            y0 = m[0] + s12 + s34;
            y1 = fma4(k2, d34, d12);
            y2 = fma4(k4, s34, s12);
            y3 = fma4(k8, d34, d12) + m[5];
        };
        f32x4 tmp[4][6];
#pragma unroll
        for (int jf = 0; jf < 6; ++jf) {
            f32x4 m[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) m[i] = *reinterpret_cast<const f32x4*>(Ms + ((i * 6 + jf) * 16 + tl) * 64 + c4 * 4);
            at4v(m, tmp[0][jf], tmp[1][jf], tmp[2][jf], tmp[3][jf]);
        }
        // the store loop in the variants the step uses (every fused convolution of the step has act none: a normalisation layer or nothing follows):
        // plain store + tile statistics (forward), plain store (sub-pixel phases), accumulate (input gradients); everything else: generic
        auto finish = [&](auto ACT_, auto ACC_, auto STATS_) __attribute__((always_inline)) {
            constexpr bool ACT = decltype(ACT_)::value, ACC = decltype(ACC_)::value, STATS = decltype(STATS_)::value;
            f32x4 sh = (f32x4)(0.f), s1 = (f32x4)(0.f), s2 = (f32x4)(0.f);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                f32x4 o[4], old[4];
                if (ACC) {
#pragma unroll
                    for (int bb = 0; bb < 4; ++bb) old[bb] = (a < rows && bb < cols) ? *reinterpret_cast<const f32x4*>(ybase + a * yrs + bb * ycs) : (f32x4)(0.f);
                }
                at4v(tmp[a], o[0], o[1], o[2], o[3]);
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    f32x4 val = o[bb] + bv;
                    if (ACT) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) val[e] = fmaxf(val[e], 0.f) + slope * fminf(val[e], 0.f);
                    }
                    if (ACC) val += old[bb];
                    if (a < rows && bb < cols && !(ABL & 128)) *reinterpret_cast<f32x4*>(ybase + a * yrs + bb * ycs) = val;
                    if (STATS) {
                        if (a == 0 && bb == 0) sh = val;
                        const f32x4 dv = val - sh;
                        s1 += dv; s2 += dv * dv;
                    }
                }
            }
            if (STATS && live) {      // (mean, M2) of the tile's 16 outputs per channel: the chunk partials norm_finalize_* combines
                const f32x4 mean = sh + s1 * (1.f / 16.f), m2 = s2 - s1 * s1 * (1.f / 16.f);
                float* so = reinterpret_cast<float*>(p.stats + (((size_t)b * p.TY + ty) * p.TX + tx) * p.Cout + n);
                *reinterpret_cast<f32x4*>(so) = (f32x4){mean[0], m2[0], mean[1], m2[1]};
                *reinterpret_cast<f32x4*>(so + 4) = (f32x4){mean[2], m2[2], mean[3], m2[3]};
            }
        };
        using T_ = std::true_type; using F_ = std::false_type;
        const bool actq = p.act != ACLGAN_ACT_NONE, accq = p.accumulate != 0, stq = p.stats != nullptr;
        if (!actq && !accq && stq) finish(F_{}, F_{}, T_{});
        else if (!actq && !accq && !stq) finish(F_{}, F_{}, F_{});
        else if (!actq && accq && !stq) finish(F_{}, T_{}, F_{});
        else if (accq) { if (stq) finish(T_{}, T_{}, T_{}); else finish(T_{}, T_{}, F_{}); }
        else { if (stq) finish(T_{}, F_{}, T_{}); else finish(T_{}, F_{}, F_{}); }
    }
}

std::atomic<int> g_wino_fused{-1};
thread_local int tl_wino_force = 0;      // > 0: this thread's calls take the fused kernel wherever the shape is eligible (the operator entry point)

}  // namespace

// tuning / test knob behind aclgan_set_tuning("wino_fused", v): 0 = the three-launch pipeline of conv_wino.hip, 1 = the fused kernel where its
// cost model says it pays (wino_fused_ok), 2 = the fused kernel wherever the shape is eligible;
// returns the previous value.  ACLGAN_WINO_FUSED sets the default.  (bits 4.. select a measurement build when compiled with
// -DACLGAN_FUSED_ABLATION)
int wino_fused_mode() {
    int v = g_wino_fused.load();
    if (v < 0) { const char* e = getenv("ACLGAN_WINO_FUSED"); v = e ? atoi(e) : 1; if (v < 0 || (v & 15) > 2) v = 1; g_wino_fused.store(v); }
    return v;
}
int set_wino_fused(int v) { const int old = wino_fused_mode(); g_wino_fused.store((v < 0 || (v & 15) > 2) ? 1 : v); return old; }
// The operator entry point (aclgan_conv3x3_winograd_fused) IS the fused kernel whatever the step's switch says: a per-THREAD override
// instead of flipping the process-wide switch around the launch (round 4 did; an update running on another thread saw the mode change
// in the middle of its plan).  Returns the previous value.
int wino_fused_force(int on) { const int old = tl_wino_force; tl_wino_force = on; return old; }

// the fused kernel takes: K-side channels a multiple of 16, output channels a multiple of 64, byte offsets below 2^31; no tanh epilogue.
// Mode 1 (default) additionally asks the cost model below whether the one-launch kernel PAYS for this grid; mode 2 takes it whenever the
// shape is eligible (tests).  gph: grid phases (forward of the sub-pixel layers: 4 launch-parallel phases), kph: K phases (their input gradient).
//
// Cost model (microseconds, measured on the ResBlock channel counts, scripts/probe_fused_abl.py B = 1 .. 8, round 4): a workgroup of the fused
// kernel walks the whole K loop whatever the grid size -- 10 + 83 * K / 256 us per round of 256 workgroups -- while the three-launch pipeline
// scales with the tile count: 30 + 0.0488 us per (tile x 256 x 256 channel pair).  Crossover on the 256-channel ResBlock: ~1400 tiles (B = 5.5 at
// 64 x 64); below it (the reference's own batch_size 3, the 64 x 64 B = 1 launch-floor probe) the pipeline is faster (B = 1: 42 against 93 us).
bool wino_fused_ok(int B, int H, int W, int Cin_, int Cout_, int act, int gph, int kph) {
    const int m = tl_wino_force > 0 ? 2 : (wino_fused_mode() & 15);
    if (m == 0) return false;
    const bool shape = act != ACLGAN_ACT_TANH && Cin_ % (2 * KC) == 0 && Cout_ % NBC == 0 && (long long)4 * 36 * Cin_ * Cout_ * 4 < 0x7fffffe0ll && H >= 4 && W >= 4 &&
                       (long long)B * (2 * H + 8) * (2 * W + 8) * std::max(Cin_, Cout_) * 4 < 0x7fffffe0ll;
    if (!shape || m == 2) return shape;
    const int TY = cdiv(H, 4), TX = cdiv(W, 4);
    const double nwg = (double)B * cdiv(TY, TBY) * cdiv(TX, TBX) * (Cout_ / NBC) * gph;
    const double t_fused = (10.0 + 83.0 * (double)Cin_ * kph / 256.0) * std::ceil(nwg / 256.0);
    const double t_pipe = 30.0 + 0.0488 * (double)B * TY * TX * gph * kph * ((double)Cin_ * Cout_ / 65536.0);
    return t_fused <= t_pipe;
}
size_t wino_fused_u_bytes(int Cin_, int Cout_) { return (size_t)36 * Cin_ * Cout_ * sizeof(float); }

// Uf <- fragment-ordered G g G^T of w (flip: the flipped, transposed filter of the input gradient); nph filters back to back
int wino_fused_filter(const float* w, float* Uf, int Co, int Ci, int flip, hipStream_t st, int nph) {
    const int64_t n = (int64_t)Co * Ci;
    hipLaunchKernelGGL(wino_filter_frag_kernel, dim3((int)std::min<int64_t>(cdiv64(n, 256), 4096), nph), dim3(256), 0, st, w, Uf, Co, Ci, flip,
                       (int64_t)Co * 9 * Ci, (int64_t)36 * n, 0);
    ACL_CHECK_LAUNCH("wino_filter_frag_kernel");
    return ACLGAN_OK;
}
// `count` equally shaped filters, w_stride floats apart, into Uf0 + i * u_stride floats: one launch
int wino_fused_filter_batch(const float* w0, int64_t w_stride, float* Uf0, int64_t u_stride, int count, int Co, int Ci, int flip, hipStream_t st) {
    const int64_t n = (int64_t)Co * Ci;
    hipLaunchKernelGGL(wino_filter_frag_kernel, dim3((int)std::min<int64_t>(cdiv64(n, 256), 1024), count), dim3(256), 0, st, w0, Uf0, Co, Ci, flip,
                       w_stride, u_stride, 0);
    ACL_CHECK_LAUNCH("wino_filter_frag_kernel(batch)");
    return ACLGAN_OK;
}

namespace {
int wino_fused_go(WfP& p, int ngph, hipStream_t st) {
    p.TY = cdiv(p.OH, 4); p.TX = cdiv(p.OW, 4); p.NBY = cdiv(p.TY, TBY); p.NBX = cdiv(p.TX, TBX);
    p.ncb = p.Cout / NBC; p.ntb = p.B * p.NBY * p.NBX;
    p.xcdmap = (p.ncb <= 8 && 8 % p.ncb == 0 && p.ntb % (8 / p.ncb) == 0) ? 1 : 0;
    p.xbytes = (long long)p.B * p.IHS * p.IWS * p.Cin * 4;
    p.uphase = 36 * p.Cin * p.Cout * 4;
    p.ubytes = (long long)p.uphase * p.nkph * ngph;
    const dim3 grid(p.ncb * p.ntb, ngph);
    const int abl = wino_fused_mode() >> 4;
#ifdef ACLGAN_FUSED_ABLATION
#define ACL_ABL(A_) else if (abl == A_) hipLaunchKernelGGL((wino_fused_kernel<A_>), grid, dim3(256), 0, st, p);
    if (abl == 0) hipLaunchKernelGGL(wino_fused_kernel<0>, grid, dim3(256), 0, st, p);
    ACL_ABL(1) ACL_ABL(2) ACL_ABL(4) ACL_ABL(8) ACL_ABL(14) ACL_ABL(15) ACL_ABL(31) ACL_ABL(128)
#undef ACL_ABL
    else
#endif
    { (void)abl; hipLaunchKernelGGL(wino_fused_kernel<0>, grid, dim3(256), 0, st, p); }
    ACL_CHECK_LAUNCH("wino_fused_kernel");
    return ACLGAN_OK;
}
}  // namespace

// out[B][H][W][Cout_] (+)= act(conv3x3(in[B][H][W][Cin_]) + bias) with U already in fragment order; reflect: reflection padding 1,
// else zero padding.  stats (optional): [B][H/4 * W/4][Cout_] (mean, M2) of the 4x4 output tiles.  H, W multiples of 4.
int wino_fused_launch(int B, int H, int W, int Cin_, int Cout_, const float* in, const float* Uf, const float* bias, float* out, int act, int accumulate,
                      int reflect, float2* stats, hipStream_t st) {
    if (!wino_fused_ok(B, H, W, Cin_, Cout_, act) || H % 4 != 0 || W % 4 != 0) return ACLGAN_EUNSUPPORTED;
    WfP p;
    p.x = in; p.Uf = Uf; p.bias = bias; p.y = out; p.stats = stats;
    p.B = B; p.Cin = Cin_; p.Cout = Cout_;
    p.IH = H; p.IW = W; p.ivs = 1; p.ivy0 = 0; p.ivx0 = 0; p.IHS = H; p.IWS = W; p.off = -1; p.reflect = reflect;
    p.nkph = 1;
    for (int i = 0; i < 4; ++i) { p.kph_xoff[i] = 0; p.ovy0[i] = 0; p.ovx0[i] = 0; }
    p.OH = H; p.OW = W; p.ovs = 1; p.OHS = H; p.OWS = W;
    p.act = act; p.accumulate = accumulate;
    return wino_fused_go(p, 1, st);
}

// The four VALID 3x3 phase convolutions of an "Upsample(2) + ReflectionPad2d(2) + Conv2d(5x5)" layer (conv_wino.hip, conv_fast.hip up5_*):
// phase (py, px) of the hi-res output y[B][Hf][Wf][Cout_] at (2 (oy + 1) + py, 2 (ox + 1) + px), oy < Hi - 2, ox < Wi - 2, from the low-res input
// x[B][Hi][Wi][Cin_].  Uf: the four merged phase filters in fragment order, back to back.  One launch, blockIdx.y = phase.
int wino_fused_up5_fwd(int B, int Hi, int Wi, int Cin_, int Cout_, const float* x, const float* Uf, const float* bias, float* y, int Hf, int Wf, int act,
                       hipStream_t st) {
    if (!wino_fused_ok(B, Hi - 2, Wi - 2, Cin_, Cout_, act, 4, 1) || Hi < 6 || Wi < 6) return ACLGAN_EUNSUPPORTED;
    WfP p;
    p.x = x; p.Uf = Uf; p.bias = bias; p.y = y; p.stats = nullptr;
    p.B = B; p.Cin = Cin_; p.Cout = Cout_;
    p.IH = Hi; p.IW = Wi; p.ivs = 1; p.ivy0 = 0; p.ivx0 = 0; p.IHS = Hi; p.IWS = Wi; p.off = 0; p.reflect = 0;
    p.nkph = 1;
    for (int i = 0; i < 4; ++i) { p.kph_xoff[i] = 0; p.ovy0[i] = 2 + (i >> 1); p.ovx0[i] = 2 + (i & 1); }
    p.OH = Hi - 2; p.OW = Wi - 2; p.ovs = 2; p.OHS = Hf; p.OWS = Wf;
    p.act = act; p.accumulate = 0;
    return wino_fused_go(p, 4, st);
}
// ... and their input gradient: dx[B][Hi][Wi][Cin_] (+)= sum over the phases of the full correlation of the phase view of dy[B][Hf][Wf][Cout_]
// with the flipped merged filter: ONE K loop over (phase, cout), the sum over the phases happens in the accumulators.
int wino_fused_up5_dgrad(int B, int Hi, int Wi, int Cin_, int Cout_, const float* dy, const float* Uf, float* dx, int Hf, int Wf, int accumulate, hipStream_t st) {
    if (!wino_fused_ok(B, Hi, Wi, Cout_, Cin_, ACLGAN_ACT_NONE, 1, 4) || Hi < 6 || Wi < 6) return ACLGAN_EUNSUPPORTED;
    WfP p;
    p.x = dy; p.Uf = Uf; p.bias = nullptr; p.y = dx; p.stats = nullptr;
    p.B = B; p.Cin = Cout_; p.Cout = Cin_;
    p.IH = Hi - 2; p.IW = Wi - 2; p.ivs = 2; p.ivy0 = 2; p.ivx0 = 2; p.IHS = Hf; p.IWS = Wf; p.off = -2; p.reflect = 0;
    p.nkph = 4;
    for (int i = 0; i < 4; ++i) { p.kph_xoff[i] = ((i >> 1) * Wf + (i & 1)) * Cout_ * 4; p.ovy0[i] = 0; p.ovx0[i] = 0; }
    p.OH = Hi; p.OW = Wi; p.ovs = 1; p.OHS = Hi; p.OWS = Wi;
    p.act = ACLGAN_ACT_NONE; p.accumulate = accumulate;
    return wino_fused_go(p, 1, st);
}

// ------------------------------------------------------------------------------------------
// Round 6: the 4x4 STRIDE-2 reflect-pad-1 layers (content / style encoder downsampling, discriminator layers 1..3: networks.py:41, 216-221,
// 236-241) through the same kernel.  y[oy] = sum_k w[k] xpad[2 oy + k] splits by the parity of the input row into two 2-tap stride-1
// convolutions of the decimated input (polyphase); a 2-tap filter embedded in the 3-tap window of F(4x4,3x3) costs 36 multiplies per 16
// outputs and phase, 4 phases: 9 per output and input channel against 16 for the direct kernel (1.78x fewer MACs).
//   forward: ONE K loop over the four parity views of x (stride-2 views, K phases) x Cin.  ReflectionPad2d(1) of the full-resolution
//            map is EDGE replication in every parity view (row -1 -> row 1 = odd[0], row H -> row H - 2 = even[H/2 - 1]); the replicated
//            sample a view does not need meets its structurally zero tap (s2k4_tap).
//   input gradient (interior of the padded grid): dx of parity phase (py, px) is a 3-window correlation of dy (zero padding) with the
//            embedded taps: four grid phases writing stride-2 views of dx, K = Cout.  The mirrored halo ring keeps its small direct launch.
// Uf: the four embedded phase filters in fragment order, back to back (wino_fused_filter_s2k4).
// ------------------------------------------------------------------------------------------
int wino_fused_filter_s2k4(const float* w, float* Uf, int Co, int Ci, int dgrad, hipStream_t st) {
    const int64_t n = (int64_t)Co * Ci;
    hipLaunchKernelGGL(wino_filter_frag_kernel, dim3((int)std::min<int64_t>(cdiv64(n, 256), 4096), 4), dim3(256), 0, st, w, Uf, Co, Ci, dgrad ? 1 : 0,
                       (int64_t)0, (int64_t)36 * n, dgrad ? 2 : 1);
    ACL_CHECK_LAUNCH("wino_filter_frag_kernel(s2k4)");
    return ACLGAN_OK;
}
// Cost model (microseconds; the fused kernel as in wino_fused_ok, the direct implicit-GEMM kernels at the rates of the round-5 trace:
// 115 - 145 TFLOP/s on full grids, far less once split-K has to fill the chip) -- mode 1 asks it, mode 2 takes every eligible shape.
bool wino_fused_s2k4_ok(int B, int Hi, int Wi, int Ci, int Co, int act, int dgrad) {
    const int m = tl_wino_force > 0 ? 2 : (wino_fused_mode() & 15);
    if (m == 0) return false;
    const int Cin_ = dgrad ? Co : Ci, Cout_ = dgrad ? Ci : Co;      // the kernel's K-side / output-side channels
    const bool shape = act != ACLGAN_ACT_TANH && Hi % 2 == 0 && Wi % 2 == 0 && Hi >= 8 && Wi >= 8 && Cin_ % (2 * KC) == 0 && Cout_ % NBC == 0 &&
                       (long long)4 * 36 * Ci * Co * 4 < 0x7fffffe0ll && (long long)B * Hi * Wi * std::max(Ci, Co) * 4 < 0x7fffffe0ll;
    if (!shape || m == 2) return shape;
    const int OH = Hi / 2, OW = Wi / 2, TY = cdiv(OH, 4), TX = cdiv(OW, 4);
    const int gph = dgrad ? 4 : 1, kph = dgrad ? 1 : 4;
    const double nwg = (double)B * cdiv(TY, TBY) * cdiv(TX, TBX) * (Cout_ / NBC) * gph;
    const double t_fused = (10.0 + 83.0 * (double)Cin_ * kph / 256.0) * std::ceil(nwg / 256.0);
    const double flop = 2.0 * B * OH * OW * 16.0 * Ci * Co;
    const double nblk = (double)cdiv(B * OH * OW, 128) * cdiv(dgrad ? Ci : Co, 128) * (dgrad ? 4 : 1);
    const double t_direct = 8.0 + flop / (nblk >= 256 ? 120e6 : 60e6 + 60e6 * nblk / 256.0);
    return t_fused <= 0.95 * t_direct;
}
size_t wino_fused_s2k4_u_bytes(int Ci, int Co) { return (size_t)144 * Ci * Co * sizeof(float); }

// y[B][Hi/2][Wi/2][Co] = act(conv4x4 stride 2, reflect pad 1 (x[B][Hi][Wi][Ci]) + bias); stats (optional): (mean, M2) of the 4x4 output tiles
int wino_fused_s2k4_fwd(int B, int Hi, int Wi, int Ci, int Co, const float* x, const float* Uf, const float* bias, float* y, int act, float2* stats,
                        hipStream_t st) {
    if (!wino_fused_s2k4_ok(B, Hi, Wi, Ci, Co, act, 0)) return ACLGAN_EUNSUPPORTED;
    const int OH = Hi / 2, OW = Wi / 2;
    if (stats && (OH % 4 != 0 || OW % 4 != 0)) return ACLGAN_EUNSUPPORTED;
    WfP p;
    p.x = x; p.Uf = Uf; p.bias = bias; p.y = y; p.stats = stats;
    p.B = B; p.Cin = Ci; p.Cout = Co;
    p.IH = OH; p.IW = OW; p.ivs = 2; p.ivy0 = 0; p.ivx0 = 0; p.IHS = Hi; p.IWS = Wi; p.off = -1; p.reflect = 2;
    p.nkph = 4;
    for (int i = 0; i < 4; ++i) { p.kph_xoff[i] = ((i >> 1) * Wi + (i & 1)) * Ci * 4; p.ovy0[i] = 0; p.ovx0[i] = 0; }
    p.OH = OH; p.OW = OW; p.ovs = 1; p.OHS = OH; p.OWS = OW;
    p.act = act; p.accumulate = 0;
    return wino_fused_go(p, 1, st);
}
// dx[B][Hi][Wi][Ci] (+)= the interior of the padded-grid gradient of the same layer from dy[B][Hi/2][Wi/2][Co]
int wino_fused_s2k4_dgrad(int B, int Hi, int Wi, int Ci, int Co, const float* dy, const float* Uf, float* dx, int accumulate, hipStream_t st) {
    if (!wino_fused_s2k4_ok(B, Hi, Wi, Ci, Co, ACLGAN_ACT_NONE, 1)) return ACLGAN_EUNSUPPORTED;
    const int OH = Hi / 2, OW = Wi / 2;
    WfP p;
    p.x = dy; p.Uf = Uf; p.bias = nullptr; p.y = dx; p.stats = nullptr;
    p.B = B; p.Cin = Co; p.Cout = Ci;
    p.IH = OH; p.IW = OW; p.ivs = 1; p.ivy0 = 0; p.ivx0 = 0; p.IHS = OH; p.IWS = OW; p.off = -1; p.reflect = 0;
    p.nkph = 1;
    for (int i = 0; i < 4; ++i) { p.kph_xoff[i] = 0; p.ovy0[i] = i >> 1; p.ovx0[i] = i & 1; }
    p.OH = OH; p.OW = OW; p.ovs = 2; p.OHS = Hi; p.OWS = Wi;
    p.act = ACLGAN_ACT_NONE; p.accumulate = accumulate;
    return wino_fused_go(p, 4, st);
}

}  // namespace aclgan
