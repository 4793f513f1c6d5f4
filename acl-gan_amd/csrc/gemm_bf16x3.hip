// gemm_bf16x3.hip -- fp32-accurate batched GEMM slices on the bf16 matrix cores of gfx950 (round 3).
//
// The Winograd pipeline of the fp32 step (conv_wino.hip) is 36 (or 144) small GEMMs M_f[T][N] = V_f[T][K] U_f[N][K]^T per convolution:
// 21 % of the fp32 step on v_mfma_f32_32x32x2_f32 (157 TFLOP/s peak, measured 0.60 of it: K is only the channel count).  The bf16 matrix
// cores do 16x the multiply-adds per instruction.  An fp32 number is EXACTLY the sum of three bf16 numbers
//     x = h + m + l,   h = bf16(x),  m = bf16(x - h),  l = bf16(x - h - m)          (8 + 8 + 8 significand bits; both differences are exact in fp32)
// so a product of two fp32 numbers is the sum of nine bf16 x bf16 products, each EXACT in the fp32 accumulator of the MFMA.  The three
// smallest (m l', l m', l l': below 2^-24 of the product, i.e. below the rounding of an fp32 multiply-add) are dropped, the other six
//     h h' + h m' + m h' + h l' + m m' + l h'
// are six v_mfma_f32_32x32x16_bf16 into the same accumulator: 6/16 of the fp32 instruction count at 2x the time each = 2.67x the fp32 matrix
// peak (417 TFLOP/s equivalent), with fp32 accumulation and an operand error of 2^-26.  Measured against fp64: the same max-abs relative
// error as the fp32 MFMA kernel (tests/test_gpu_ops_misc.py::test_gemm_slices_x3).  This is fp32 arithmetic carried out on the 16-bit
// pipes, NOT a reduced-precision mode (bf16 compute -- one product, 8 bits -- is BASELINE configs[2] and stays a separate, named dtype).
//
// Operands arrive already split ("3-plane" layout, written by the producers: wino_input_kernel / wino_filter_kernel, or split3 below):
//     A3: plane p = [slices][T][K] bf16 at A3 + p * a_plane bytes;   B3: plane p = [nslices][N][K] bf16
// so the tiles go global -> LDS by LDS-DMA (no conversion in the loader), 64-byte rows (k-tile 32), XOR-swizzled 16-byte chunks.
// Kernel: 128 x 128 (or 128 x 64) tile, FOUR consumer waves (64 x 64 each: 48 MFMAs per k-tile from 24 fragment reads) + TWO producer
// waves that only copy (see conv_glds16.hip: a wave that issues LDS-DMA is held at issue while the vector-memory queue is full, so copies and
// MFMAs of one wave add up; scripts/microbench/spec_skeleton.hip: 0.87 us per k-tile specialised against 1.39 us unified), three 48 KB LDS
// stages, one bare s_barrier per k-tile.
#include "common.h"
#include "conv_fast_common.h"
#include "st16.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace aclgan {
namespace {

typedef unsigned short u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define WG_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
constexpr int vmcnt_imm(int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); }      // s_waitcnt vmcnt(n), other counters unconstrained

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* base, long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(bytes > 0x7fffffe0ll ? 0x7fffffe0ll : bytes), 0x00020000);
}

struct G3P {
    const unsigned char* a; const unsigned char* b; float* c;
    long long a_plane, b_plane;      // bytes between the planes of an operand
    int T, K, N, nslices, a_mod, tiles_m, tiles_n, nwg;
};

[[maybe_unused]] constexpr int R3 = 64;               // bytes per LDS row = 32 bf16 = one k-tile of one plane

// PERSISTENT: the grid is one workgroup per CU; workgroup w walks the output tiles xcd_map(w + i gridDim.x) (same XCD every time, a
// contiguous chunk of the slice-major tile order per XCD: the tiles of a slice share its U planes in that XCD's L2).  The k-tiles of
// all its tiles form ONE stream through the LDS stages: the producers are already copying the next tile's first k-tiles while the consumers
// store the current one (with K = 256 a tile is only 8 k-tiles: a per-tile prologue would be a third of its time).
template <int TN, int NBUF, int NP>
__global__ void __launch_bounds__((4 + NP) * 64, 2) gemm3_kernel(G3P p) {
#if defined(__HIP_DEVICE_COMPILE__)   // (device pass only: a template body with __amdgpu_buffer_rsrc_t locals leaves the host launch stub undefined)
    constexpr int TM = 2, WN = 2, NW = 4, BM = 128, BN = WN * TN * 32;
    constexpr int A_PL = BM * R3, B_PL = BN * R3;                   // bytes of one plane of a stage
    constexpr int STAGE = 3 * (A_PL + B_PL);
    constexpr int A_PC = 3 * BM / 16, B_PC = 3 * BN / 16;           // 1 KB pieces (16 rows x 64 B) per stage
    constexpr int IT = (A_PC + B_PC) / NP;                          // pieces per producer wave and k-tile
    static_assert((A_PC + B_PC) % NP == 0 && NBUF >= 2, "piece split");
    constexpr int WAIT_NEXT = vmcnt_imm((NBUF - 2) * IT);
    __shared__ __attribute__((aligned(16))) unsigned char smem[NBUF * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool prod = wave >= NW;
    const int per = p.tiles_m * p.tiles_n;
    const int nk = p.K >> 5;
    const int G = gridDim.x;
    const int ntl = (p.nwg - (int)blockIdx.x + G - 1) / G;          // tiles of this workgroup
    const int NG = ntl * nk;                                         // k-tiles of this workgroup

    if (prod) {
        const int lr = lane >> 2, lc = lane & 3;                     // row of the piece, chunk position in the LDS row
        const long long a_bytes = (long long)p.T * p.K * 2, b_bytes = (long long)p.N * p.K * 2;
        // piece q of a stage: q < A_PC: A plane q / (BM / 16), rows 16 (q % (BM / 16)) ..; else B likewise.  Producer LW owns q = LW IT + n
        // (LW as a compile-time constant: plane and operand of every piece are then known to the compiler)
        auto produce = [&](auto lwc) __attribute__((always_inline)) {
            constexpr int LW = decltype(lwc)::value;
            int vo[IT];
            __amdgpu_buffer_rsrc_t ra[3], rb3[3];
            int it_tile = -1, it_kt = nk;                            // issue cursor: (tile index of this workgroup, k-tile)
            auto issue = [&](int buf) __attribute__((always_inline)) {
                if (it_kt == nk) {                                   // next tile: new rows, new slice
                    it_kt = 0; ++it_tile;
                    const int tile = xcd_map((int)blockIdx.x + it_tile * G, p.nwg);
                    const int slice = tile / per, tin = tile - slice * per;
                    const int m0 = (tin / p.tiles_n) * BM, n0 = (tin % p.tiles_n) * BN;
                    const long long a_rows = (long long)(p.a_mod > 0 ? slice % p.a_mod : slice) * p.T, b_rows = (long long)slice * p.N;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        ra[pl] = rsrc_of(p.a + pl * p.a_plane + a_rows * p.K * 2, a_bytes);
                        rb3[pl] = rsrc_of(p.b + pl * p.b_plane + b_rows * p.K * 2, b_bytes);
                    }
#pragma unroll
                    for (int n = 0; n < IT; ++n) {
                        const int q = LW * IT + n;
                        const bool isA = q < A_PC;
                        const int qq = isA ? q : q - A_PC;
                        const int row = (isA ? qq % (BM / 16) : qq % (BN / 16)) * 16 + lr;
                        const int grow = isA ? min(m0 + row, p.T - 1) : min(n0 + row, p.N - 1);      // past the end: any valid row (never stored)
                        vo[n] = grow * p.K * 2 + ((lc ^ ((row >> 2) & 3)) << 4);
                    }
                }
                unsigned char* st = smem + buf * STAGE;
#pragma unroll
                for (int n = 0; n < IT; ++n) {
                    const int q = LW * IT + n;
                    const bool isA = q < A_PC;
                    const int qq = isA ? q : q - A_PC;
                    const int pl = isA ? qq / (BM / 16) : qq / (BN / 16);
                    // (the planes of an operand are contiguous in a stage: plane pl starts rows / 16 pieces after plane pl - 1)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(isA ? ra[pl] : rb3[pl], LDS_PTR(st + (isA ? 0 : 3 * A_PL) + qq * 1024), 16, vo[n], it_kt * R3, 0, 0);
                }
                ++it_kt;
            };
#pragma unroll
            for (int t = 0; t < NBUF - 1; ++t)
                if (t < NG) issue(t);
            if (NBUF - 1 <= NG) __builtin_amdgcn_s_waitcnt(WAIT_NEXT);
            else __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
            WG_BARRIER();                                  // k-tile 0 has landed
            int ib = NBUF - 1;
            for (int g = 0; g < NG; ++g) {
                if (g + NBUF - 1 < NG) {
                    issue(ib);                             // (into the stage the consumers read in iteration g - 1)
                    __builtin_amdgcn_s_waitcnt(WAIT_NEXT); // k-tile g + 1 has landed
                } else {
                    __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
                }
                ib = ib + 1 == NBUF ? 0 : ib + 1;
                WG_BARRIER();
            }
        };
        static_assert(NP == 2, "two producer waves");
        if (wave == NW) produce(std::integral_constant<int, 0>{});
        else produce(std::integral_constant<int, 1>{});
        return;
    }

    // ---- consumer wave: 64 x (32 TN) of the tile ----
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, kh = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // fragment addresses inside a stage: row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4), chunk = 2 ks + kh
    int ao[TM], bo[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { const int row = wm * 64 + i * 32 + l31; ao[i] = row * R3 + ((kh ^ ((row >> 2) & 3)) << 4); }
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int row = wn * TN * 32 + j * 32 + l31; bo[j] = 3 * A_PL + row * R3 + ((kh ^ ((row >> 2) & 3)) << 4); }
    // Software pipeline over the two 16-deep k-steps of a k-tile: the twelve fragment reads of the NEXT k-step are in flight under the 24
    // MFMAs of the current one (one consumer wave per SIMD: nobody else would cover the LDS latency).
    //   loop:  read(g, 1) | mma(g, 0) | lgkmcnt(0), barrier (stage g is free, k-tile g + 1 has landed) | read(g + 1, 0) | mma(g, 1)
    u32x4 fa[2][TM][3], fb[2][TN][3];
    auto rd = [&](const unsigned char* st, int ks, u32x4 (&xa)[TM][3], u32x4 (&xb)[TN][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) xa[i][pl] = *reinterpret_cast<const u32x4*>(st + pl * A_PL + (ao[i] ^ (ks << 5)));
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) xb[j][pl] = *reinterpret_cast<const u32x4*>(st + pl * B_PL + (bo[j] ^ (ks << 5)));
    };
    auto mm = [&](const u32x4 (&xa)[TM][3], const u32x4 (&xb)[TN][3]) __attribute__((always_inline)) {
        // smallest terms first (the accumulator then rounds them against the least); the six products of a tile are interleaved over the
        // TM x TN tiles so that consecutive MFMAs never wait for each other's accumulator
#define ACL_X3(pa, pb)                                                              \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                              \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = mfma_bf16(xa[i][pa], xb[j][pb], acc[i][j]);
        ACL_X3(2, 0)      // l h'
        ACL_X3(0, 2)      // h l'
        ACL_X3(1, 1)      // m m'
        ACL_X3(1, 0)      // m h'
        ACL_X3(0, 1)      // h m'
        ACL_X3(0, 0)      // h h'
#undef ACL_X3
    };
    WG_BARRIER();                                      // k-tile 0 has landed
    rd(smem, 0, fa[0], fb[0]);
    int cb = 0, kt = 0, ti = 0;
    for (int g = 0; g < NG; ++g) {
        const unsigned char* st = smem + cb * STAGE;
        rd(st, 1, fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        mm(fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        cb = cb + 1 == NBUF ? 0 : cb + 1;
        __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): every fragment read of this stage has returned
        WG_BARRIER();                                  // this stage may be refilled / the next k-tile has landed
        if (g + 1 < NG) rd(smem + cb * STAGE, 0, fa[0], fb[0]);
        __builtin_amdgcn_sched_barrier(0);
        mm(fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        if (++kt == nk) {                              // tile done: C[slice][m][n]; a lane holds column n of rows (r & 3) + 8 (r >> 2) + 4 kh of each 32 x 32 block
            const int tile = xcd_map((int)blockIdx.x + ti * G, p.nwg);
            const int slice = tile / per, tin = tile - slice * per;
            const int m0 = (tin / p.tiles_n) * BM, n0 = (tin % p.tiles_n) * BN;
            float* cs = p.c + (size_t)slice * p.T * p.N;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * TN * 32 + j * 32 + l31;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                        if (m < p.T && n < p.N) cs[(size_t)m * p.N + n] = acc[i][j][r];
                        acc[i][j][r] = 0.f;
                    }
            }
            kt = 0; ++ti;
        }
    }
#endif
}

// fp32 -> the three bf16 planes (4 values per thread)
__global__ void __launch_bounds__(256) split3_kernel(const float* __restrict__ x, u16* __restrict__ o, int64_t n4, int64_t plane_elems) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        const float f[4] = {v.x, v.y, v.z, v.w};
        u16 h[4], m[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split3(f[e], h[e], m[e], l[e]);
        reinterpret_cast<uint2*>(o)[i] = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
        reinterpret_cast<uint2*>(o + plane_elems)[i] = make_uint2(m[0] | ((unsigned)m[1] << 16), m[2] | ((unsigned)m[3] << 16));
        reinterpret_cast<uint2*>(o + 2 * plane_elems)[i] = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
    }
}

}  // namespace

// persistent grid: one workgroup per CU (three 48 KB LDS stages), a multiple of the 8 XCDs
static int x3_grid() {
    static int g = 0;
    if (!g) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount; }
        g = std::max(8, cus / 8 * 8);
    }
    return g;
}

bool gemm_x3_shape_ok(int T, int K, int N) { return T > 0 && K >= 32 && K % 32 == 0 && N >= 64 && N % 64 == 0; }

// C[s][T][N] = A_s[T][K] B_s[N][K]^T from 3-plane operands; a_mod > 0: the A operand of slice s is slice s % a_mod
int gemm_slices_x3(const void* A3, size_t a_plane, const void* B3, size_t b_plane, float* C, int T, int K, int N, int nslices, int a_mod, hipStream_t st) {
    if (!gemm_x3_shape_ok(T, K, N) || nslices <= 0) { set_error("gemm_slices_x3: bad shape (K %% 32, N %% 64)"); return ACLGAN_EINVAL; }
    ACL_REQUIRE((long long)T * K * 2 < 0x7fffffe0ll && (long long)N * K * 2 < 0x7fffffe0ll, "gemm_slices_x3: slice beyond 2 GB");
    G3P p;
    p.a = (const unsigned char*)A3; p.b = (const unsigned char*)B3; p.c = C; p.a_plane = (long long)a_plane; p.b_plane = (long long)b_plane;
    p.T = T; p.K = K; p.N = N; p.nslices = nslices; p.a_mod = a_mod;
    p.tiles_m = cdiv(T, 128);
    if (N % 128 == 0) {
        p.tiles_n = N / 128; p.nwg = p.tiles_m * p.tiles_n * nslices;
        hipLaunchKernelGGL((gemm3_kernel<2, 3, 2>), dim3(std::min(p.nwg, x3_grid())), dim3(384), 0, st, p);
    } else {
        p.tiles_n = N / 64; p.nwg = p.tiles_m * p.tiles_n * nslices;
        hipLaunchKernelGGL((gemm3_kernel<1, 3, 2>), dim3(std::min(p.nwg, x3_grid())), dim3(384), 0, st, p);
    }
    ACL_CHECK_LAUNCH("gemm3_kernel");
    return ACLGAN_OK;
}

// n fp32 values (n % 4 == 0) -> planes at out, out + plane_elems, out + 2 plane_elems (bf16 elements)
int split3_planes(const float* x, void* out, int64_t n, int64_t plane_elems, hipStream_t st) {
    ACL_REQUIRE(n % 4 == 0 && plane_elems % 4 == 0, "split3_planes: element counts must be multiples of 4");
    if (n == 0) return ACLGAN_OK;
    hipLaunchKernelGGL(split3_kernel, dim3((int)std::min<int64_t>(cdiv64(n / 4, 256), 8192)), dim3(256), 0, st, x, (u16*)out, n / 4, plane_elems);
    ACL_CHECK_LAUNCH("split3_kernel");
    return ACLGAN_OK;
}

}  // namespace aclgan
