// conv_glds16.hip -- round 3: the heavy convolutions of the 16-bit compute dtypes on 16-BIT ACTIVATIONS / GRADIENTS IN HBM, operand
// tiles loaded global -> LDS directly (buffer_load_dwordx4 ... lds), no VGPR staging, no conversion, no ds_write.
//
//   forward   networks.py:366 (ReflectionPad2d + Conv2d + bias [+ activation])           conv_fwd16s
//   dgrad     autograd of the same line w.r.t. its input (trainer.py:169,292)             conv_dgrad16s  (padded grid + ordered fold)
//
// Why a second kernel family next to conv_fast16.hip: with fp32 activations in HBM the 16-bit MFMA kernels sat at 0.21 of the
// 2.5 PFLOP/s roof (round 2, profiles/r02_pmc_conv16.txt): 6.5 VALU instructions per MFMA -- 64-bit address arithmetic of six global
// loads per k-tile, eight v_cvt_pk, the register -> LDS staging writes -- and twice the operand bytes.  Here the producers (norm_apply,
// the conv epilogues, norm_bwd_apply, the fold) store bf16 / fp16, so an operand tile row is 128 contiguous BYTES of HBM that one
// quarter-wave copies straight into LDS; the per-lane work per k-tile is eight LDS-DMA issues with an SGPR k offset.
//
// Tile: 128 (pixels) x 128 | 64 (channels) x 64 (k) per workgroup of 4 waves (2 x 2, 64 x 64 | 64 x 32 per wave, 32x32x16 MFMA),
// two LDS buffers of 32 KB | 24 KB, two workgroups per CU.  LDS image = [row][64 halfs] with no padding (LDS-DMA writes are
// lane-linear: wave-uniform base + 16 B x lane); bank conflicts of the ds_read_b128 fragment reads are removed by an XOR swizzle of
// the 16-byte chunk index with (row >> 1) & 7, applied to the per-lane SOURCE address of the DMA and to the fragment read address
// (cdna_hip_programming.md rule 21: both sides or neither).  One barrier per k-tile: the DMA of tile t+1 is issued right after the
// barrier that publishes tile t and lands while tile t's 16 MFMAs per wave run.
//
// Out-of-range operand rows (dgrad: a filter tap that reaches outside the output map) point their DMA lanes past the end of the
// tensor's buffer descriptor: a raw buffer load returns zero for such lanes.
//
// The reflection halo of dgrad: ONE launch over the padded grid (every padded position has exactly one writer), then an ordered
// gather (conv_fold_st) folds the reflection onto dx -- no atomics, reproducible bit for bit, and the only plan that can write a
// 16-bit dx.
#include "conv_fast_common.h"
#include "st16.h"

namespace aclgan {
namespace {

typedef unsigned short u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct QBF16 {
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
struct QFP16 {
    static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

constexpr int ROWB = 128;          // bytes per LDS row = 64 halfs = one k-tile
[[maybe_unused]] constexpr int OOB = 0x7ffffff0;    // voffset of a DMA lane that must read zero (beyond any num_records)

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(bytes > 0x7fffffe0ll ? 0x7fffffe0ll : bytes), 0x00020000);
}

// one k-tile of MFMAs from LDS buffers a (this wave's 64 rows) and b: TM x TN tiles, 4 k-steps of 16.  All 16 fragment reads are issued
// up front (hipcc interleaves their lgkmcnt waits with the MFMAs).  Measured and rejected (profiles/r03_experiments.md): reading the
// fragments per k-step and issuing the next tile's DMA pieces behind each k-step's MFMAs (sched_barrier-pinned) -- the exposed ds_read
// latency costs far more than the DMA issue slots it hides (forward 53 -> 59 us, weight gradient 125 -> 150 us).
template <class T, int TM, int TN>
__device__ __forceinline__ void mma_tile(const unsigned char* a, const unsigned char* b, int lane, f32x16 (&acc)[TM][TN]) {
    const int l31 = lane & 31, kh = lane >> 5;
    const int c0 = kh ^ ((l31 >> 1) & 7);            // chunk (2 ks + kh) ^ sw  =  (2 ks) ^ c0
    constexpr int KG = TN >= 4 ? 2 : 4;              // k-steps whose fragments are in flight together (2 x 4 tiles: 96 fragment VGPRs would spill)
#pragma unroll
    for (int k0 = 0; k0 < 4; k0 += KG) {
        u32x4 fa[TM][KG], fb[TN][KG];
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int ks = 0; ks < KG; ++ks) fa[t][ks] = *reinterpret_cast<const u32x4*>(a + (t * 32 + l31) * ROWB + ((c0 ^ (2 * (k0 + ks))) << 4));
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int ks = 0; ks < KG; ++ks) fb[t][ks] = *reinterpret_cast<const u32x4*>(b + (t * 32 + l31) * ROWB + ((c0 ^ (2 * (k0 + ks))) << 4));
#pragma unroll
        for (int ks = 0; ks < KG; ++ks)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = T::mfma(fa[i][ks], fb[j][ks], acc[i][j]);
    }
}

// store a wave's accumulators: element (row-index table ro[], column n) of a [rows][ld] matrix of storage `yst`.
// fp32: one dword per value.  16-bit: lanes l and l^1 hold columns n and n^1 of the same 16 rows -- they swap (DPP) and each stores
// eight packed dwords (even lane: rows with even r, odd lane: odd r) instead of sixteen 2-byte stores.
template <int TM, int TN, class F>
__device__ __forceinline__ void store_acc(const f32x16 (&acc)[TM][TN], const int* ro, int rbase, int nbase, int nmax, int ld, void* __restrict__ y,
                                          int yst, int lane, F&& fin) {
    const int l31 = lane & 31, lh = lane >> 5;
    // (compile-time column block j: a `#pragma unroll` j loop around a large `fin` is unrolled too late for the 2 x 4 tile, whose accumulator
    //  array then stays in scratch through the whole main loop)
    auto col = [&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        const int n = nbase + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (yst == ST_F32) {
                if (n < nmax) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int o = ro[rbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
                        if (o >= 0) reinterpret_cast<float*>(y)[(size_t)o * ld + n] = fin(acc[i][j][r], n);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    // values of this lane for rows r, r+1; the partner (l ^ 1) holds the neighbouring column of the same rows
                    const float v0 = n < nmax ? fin(acc[i][j][r], n) : 0.f, v1 = n < nmax ? fin(acc[i][j][r + 1], n) : 0.f;
                    const float p0 = __shfl_xor(v0, 1), p1 = __shfl_xor(v1, 1);
                    const bool odd = l31 & 1;
                    const int rr = odd ? r + 1 : r;
                    const int o = ro[rbase + i * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lh];
                    const unsigned int pk = odd ? st_pack2(p1, v1, yst) : st_pack2(v0, p0, yst);
                    if (o >= 0 && (n & ~1) < nmax) *reinterpret_cast<unsigned int*>(reinterpret_cast<u16*>(y) + (size_t)o * ld + (n & ~1)) = pk;
                }
            }
        }
    };
    col(std::integral_constant<int, 0>{});
    if constexpr (TN > 1) col(std::integral_constant<int, 1>{});
    if constexpr (TN > 2) { col(std::integral_constant<int, 2>{}); col(std::integral_constant<int, 3>{}); }
}

// dgrad epilogue with two destinations (16-bit storage st, row length ld = channels): table entry o >= 0: row o of the padded scratch;
// o <= -2: pixel -2 - o of dx, stored or (acc) accumulated: dx = round(float(dx) + float(round(v))) -- the same two roundings the ordered
// fold applies to a pixel without mirrored partners.  Lane pairs (l, l ^ 1) swap as in store_acc and write packed dwords.
template <int TM, int TN>
__device__ __forceinline__ void store_acc_dx(const f32x16 (&acc)[TM][TN], const int* ro, int rbase, int nbase, int ld, void* __restrict__ scratch,
                                             void* __restrict__ dx, int st, int accumulate, int lane) {
    const int l31 = lane & 31, lh = lane >> 5;
    auto col = [&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        const int n = nbase + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float v0 = n < ld ? acc[i][j][r] : 0.f, v1 = n < ld ? acc[i][j][r + 1] : 0.f;
                const float p0 = __shfl_xor(v0, 1), p1 = __shfl_xor(v1, 1);
                const bool odd = l31 & 1;
                const int rr = odd ? r + 1 : r;
                const int o = ro[rbase + i * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * lh];
                unsigned int pk = odd ? st_pack2(p1, v1, st) : st_pack2(v0, p0, st);
                if (o == -1 || (n & ~1) >= ld) continue;
                if (o >= 0) {
                    *reinterpret_cast<unsigned int*>(reinterpret_cast<u16*>(scratch) + (size_t)o * ld + (n & ~1)) = pk;
                } else {
                    unsigned int* d = reinterpret_cast<unsigned int*>(reinterpret_cast<u16*>(dx) + (size_t)(-2 - o) * ld + (n & ~1));
                    if (accumulate) {
                        const st_f32x2 a = st_unpack2(pk, st), b = st_unpack2(*d, st);
                        pk = st_pack2(a[0] + b[0], a[1] + b[1], st);
                    }
                    *d = pk;
                }
            }
        }
    };
    col(std::integral_constant<int, 0>{});
    if constexpr (TN > 1) col(std::integral_constant<int, 1>{});
    if constexpr (TN > 2) { col(std::integral_constant<int, 2>{}); col(std::integral_constant<int, 3>{}); }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
struct FwdSP {
    const u16* x16; const u16* w16; const float* bias; void* y;
    int B, Hi, Wi, Ci, Ho, Wo, Co, k, s, p, M, K, act, tiles_n, nwg, yst;
    float2* stats;     // optional: per (128-row tile, channel) (mean, M2) of the stored outputs -- the normalisation layer's chunk partials
};

// Tile geometry (template): WM x WN waves, each 64 rows x (TN * 32) columns -> BM = 64 WM, BN = 32 TN WN; k-tile 64.
//   <2,2,2> 128 x 128 (4 waves, 2 workgroups per CU)   <2,2,1> 128 x 64   <4,2,2> 256 x 128 (8 waves, 1 per CU)   <4,2,4> 256 x 256
// Operand bytes per MFMA: 512 at 128 x 128, 384 at 256 x 128, 256 at 256 x 256 -- and yet the 128-row tiles win in the step (see glds_tile).
// NBUF = 2: two LDS buffers, one barrier per k-tile (the DMA of tile t+1 flies under tile t's MFMAs).  NBUF = 1 (128-row tiles only, experiments):
// one buffer, two barriers per k-tile, 4 workgroups per CU.
// Wave specialisation (NP > 0; round 3, scripts/microbench/lds_dma_rate.hip): a wave that issues LDS-DMA loads is held at ISSUE while the CU's
// vector-memory queue is full -- in program order its MFMAs then wait behind its own copies, and copy time and multiply time ADD UP
// (microbenchmark, 4 waves x 12 KB + 48 MFMAs per stage: 0.69 us copies alone, 0.91 us MFMAs alone, 1.32 us together; only a second
// workgroup on the CU hides part of it).  With NP producer waves that do nothing but copy and WM x WN consumer waves that do nothing but
// multiply the same stage takes 0.98 us (= the slower of the two).  Producers run S - 1 k-tiles ahead of the consumers through S LDS
// buffers; one s_barrier per k-tile, NO vmcnt(0) in front of it (gfx950 backs the barrier off, the producers wait for exactly the tile
// the consumers need next: s_waitcnt vmcnt((S - 2) x pieces)).
constexpr int vmcnt_imm(int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); }      // s_waitcnt vmcnt(n), other counters unconstrained
#define WG_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
// Barrier that PUBLISHES global -> LDS copies: every wave waits for its own copies first.  __syncthreads() alone does not guarantee that: to
// the compiler an LDS-DMA load is a load, its release fence does not wait for loads, and the wait it inserts for the LDS reads that follow
// covers only the copies THIS wave's reads may alias -- the 256 x 256 forward tile (8 copies per wave and k-tile) came out as
// `s_waitcnt vmcnt(2); s_barrier`: another wave's reads raced the last two copies (round 4: 1 launch in 3 000 differed, found by
// tests/test_gpu_ops16s.py::test_conv_fwd16s_epilogue_statistics, reproduced by scripts/debug/stress_stats16s.py).
#define DMA_SYNCTHREADS() do { __builtin_amdgcn_s_waitcnt(vmcnt_imm(0)); __syncthreads(); } while (0)

template <class T, int WM, int WN, int TN, int NBUF, int NP>
__global__ void __launch_bounds__((WM * WN + NP) * 64, NP ? 3 : (NBUF == 1 ? 4 : 2)) conv_fwd16s_kernel(FwdSP p) {
#if defined(__HIP_DEVICE_COMPILE__)   // (device pass only: the host pass cannot instantiate a template body that declares __amdgpu_buffer_rsrc_t locals -- its launch stub would stay undefined)
    constexpr int TM = 2, NW = WM * WN, NT = (NW + NP) * 64, BM = WM * 64, BN = WN * TN * 32;
    constexpr int NL = NP ? NP : NW;                            // waves that copy
    constexpr int A_IT = BM / 8 / NL, B_IT = BN / 8 / NL;       // 1 KB DMA pieces (8 rows) per copying wave and k-tile
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int WAIT_NEXT = vmcnt_imm(NBUF >= 2 ? (NBUF - 2) * (A_IT + B_IT) : 0);      // producers: everything but the newest NBUF - 2 tiles has landed
    __shared__ __attribute__((aligned(16))) unsigned char smem[NBUF * (A_BYTES + B_BYTES) + BM * 4];
    int* ro = reinterpret_cast<int*>(smem + NBUF * (A_BYTES + B_BYTES));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool prod = NP && wave >= NW;
    const int lw = NP ? wave - NW : wave;                       // index among the copying waves
    const int wm = wave / WN, wn = wave % WN;
    const int tile = xcd_map(blockIdx.x, p.nwg);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const int hw = p.Ho * p.Wo;

    for (int r = tid; r < BM; r += NT) {
        const int m = m0 + r;
        ro[r] = m < p.M ? m : -1;                 // output pixel index == GEMM row (NHWC, no phases / rings here)
    }
    if (NP) __syncthreads();                       // (the specialised loops below use bare s_barriers: make the row table visible here)
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int cpt = p.Ci >> 6;                     // k-tiles per filter tap
    const int nk = p.K >> 6;

    if (!NP || prod) {
        // DMA rows of this lane: A rows (BM / NL) lw + 8 n + (lane >> 3); the lane copies chunk (lane & 7) ^ swz(row) of its rows
        const int lr = lane >> 3, lj = lane & 7;
        int ay[A_IT], ax[A_IT], ab[A_IT], acs[A_IT];
#pragma unroll
        for (int n = 0; n < A_IT; ++n) {
            const int row = (BM / NL) * lw + 8 * n + lr;
            const int m = min(m0 + row, p.M - 1);     // past the end: any valid row (never stored)
            const int b = m / hw, rem = m - b * hw, oy = rem / p.Wo, ox = rem - oy * p.Wo;
            ay[n] = oy * p.s - p.p; ax[n] = ox * p.s - p.p; ab[n] = b * p.Hi * p.Wi;
            acs[n] = (lj ^ ((row >> 1) & 7)) * 8;
        }
        int bvo[B_IT];
#pragma unroll
        for (int n = 0; n < B_IT; ++n) {
            const int row = (BN / NL) * lw + 8 * n + lr;
            bvo[n] = (min(n0 + row, p.Co - 1) * p.K + (lj ^ ((row >> 1) & 7)) * 8) * 2;
        }
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x16, (long long)p.B * p.Hi * p.Wi * p.Ci * 2);
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w16, (long long)p.Co * p.K * 2);
        int avo[A_IT];
        int f_tap = -1;

        auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
            const int tap = kt / cpt, cc = kt - tap * cpt;
            if (tap != f_tap) {                        // block-uniform: new filter tap -> redo the gather offsets
                f_tap = tap;
                const int ky = tap / p.k, kx = tap - ky * p.k;
#pragma unroll
                for (int n = 0; n < A_IT; ++n) {
                    const int iy = refl(ay[n] + ky, p.Hi), ix = refl(ax[n] + kx, p.Wi);
                    avo[n] = ((ab[n] + iy * p.Wi + ix) * p.Ci + acs[n]) * 2;
                }
            }
            unsigned char* da = smem + buf * A_BYTES + ((BM / NL) * lw) * ROWB;
            unsigned char* db = smem + NBUF * A_BYTES + buf * B_BYTES + ((BN / NL) * lw) * ROWB;
#pragma unroll
            for (int n = 0; n < A_IT; ++n) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(da + n * 8 * ROWB), 16, avo[n], cc * 128, 0, 0);
#pragma unroll
            for (int n = 0; n < B_IT; ++n) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(db + n * 8 * ROWB), 16, bvo[n], kt * 128, 0, 0);
        };

        if (NP) {                                      // producer wave: NBUF - 1 k-tiles ahead of the consumers
#pragma unroll
            for (int t = 0; t < NBUF - 1; ++t)
                if (t < nk) issue(t, t);
            if (NBUF - 1 <= nk) __builtin_amdgcn_s_waitcnt(WAIT_NEXT);
            else __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
            WG_BARRIER();                              // tile 0 has landed
            int ib = NBUF - 1;                         // buffer of tile kt + NBUF - 1
            for (int kt = 0; kt < nk; ++kt) {
                if (kt + NBUF - 1 < nk) {
                    issue(kt + NBUF - 1, ib);          // (the buffer the consumers read in iteration kt - 1)
                    __builtin_amdgcn_s_waitcnt(WAIT_NEXT);     // tile kt + 1 has landed
                } else {
                    __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
                }
                ib = ib + 1 == NBUF ? 0 : ib + 1;
                WG_BARRIER();
            }
        } else if (NBUF == 2) {
            issue(0, 0);
            for (int kt = 0; kt < nk; ++kt) {
                const int cur = kt & 1;
                DMA_SYNCTHREADS();                     // vmcnt(0) + barrier: tile kt has landed for every wave, buffer cur ^ 1 is free
                if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
                mma_tile<T, TM, TN>(smem + cur * A_BYTES + (wm * 64) * ROWB, smem + NBUF * A_BYTES + cur * B_BYTES + (wn * TN * 32) * ROWB, lane, acc);
            }
        } else {
            for (int kt = 0; kt < nk; ++kt) {
                issue(kt, 0);
                DMA_SYNCTHREADS();                     // tile kt has landed
                mma_tile<T, TM, TN>(smem + (wm * 64) * ROWB, smem + A_BYTES + (wn * TN * 32) * ROWB, lane, acc);
                __syncthreads();                       // every wave is done reading: the buffer may be refilled
            }
        }
    } else {                                           // consumer wave
        WG_BARRIER();
        int cb = 0;
        for (int kt = 0; kt < nk; ++kt) {
            mma_tile<T, TM, TN>(smem + cb * A_BYTES + (wm * 64) * ROWB, smem + NBUF * A_BYTES + cb * B_BYTES + (wn * TN * 32) * ROWB, lane, acc);
            cb = cb + 1 == NBUF ? 0 : cb + 1;
            WG_BARRIER();                              // this buffer may be refilled / the next tile has landed
        }
    }

    const float* bias = p.bias;
    const int act = p.act;
    if (p.stats) {
        // Normalisation statistics from the epilogue (block-uniform; the launcher offers it only when every tile is BM full rows of one
        // sample): (mean, M2) of the BM STORED outputs of each channel of this tile = one chunk partial of norm_finalize_*.  A lane holds
        // 2 x 16 rows of each of its TN columns; lanes l / l ^ 32, then the WM row blocks of the workgroup, are merged with Chan's formula.
        const int l31 = lane & 31, lh = lane >> 5;
        float2* red = reinterpret_cast<float2*>(smem);
        __syncthreads();                           // every wave is done with the operand buffers
        if (!prod) {
        // (one column block per call with a COMPILE-TIME j: as a `#pragma unroll` loop over j the 2 x 4 tile's body is unrolled too late
        //  for the accumulator array to be split into registers again -- it then lives in scratch through the whole main loop)
        auto col = [&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            const int n = n0 + wn * TN * 32 + j * 32 + l31;
            const float bv = (bias && n < p.Co) ? bias[n] : 0.f;
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = act_apply(acc[i][j][r] + bv, act);
                    if (p.yst != ST_F32) v = st_unpack2(st_pack2(v, 0.f, p.yst), p.yst)[0];     // the value as it will be stored
                    acc[i][j][r] = v;
                    sum += v;
                }
            float mean = sum * (1.f / 32.f), q = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float dv = acc[i][j][r] - mean; q += dv * dv; }
            const float om = __shfl_xor(mean, 32), oq = __shfl_xor(q, 32);
            const float dm = om - mean;
            q = q + oq + dm * dm * 16.f;           // n_a n_b / (n_a + n_b) = 32 * 32 / 64
            mean = 0.5f * (mean + om);
            if (lh == 0) red[wm * BN + wn * TN * 32 + j * 32 + l31] = make_float2(mean, q);
        };
        col(std::integral_constant<int, 0>{});
        if constexpr (TN > 1) col(std::integral_constant<int, 1>{});
        if constexpr (TN > 2) { col(std::integral_constant<int, 2>{}); col(std::integral_constant<int, 3>{}); }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.Co) {
            float cnt = 64.f, mean = red[tid].x, m2 = red[tid].y;
#pragma unroll
            for (int i = 1; i < WM; ++i) {
                const float2 o = red[i * BN + tid];
                const float dm = o.x - mean, tot = cnt + 64.f;
                mean += dm * (64.f / tot);
                m2 += o.y + dm * dm * (cnt * 64.f / tot);
                cnt = tot;
            }
            p.stats[(size_t)(m0 / BM) * p.Co + n0 + tid] = make_float2(mean, m2);
        }
        if (!prod) store_acc<TM, TN>(acc, ro, wm * 64, n0 + wn * TN * 32, p.Co, p.Co, p.y, p.yst, lane, [](float v, int) { return v; });
    } else if (!prod) {
        store_acc<TM, TN>(acc, ro, wm * 64, n0 + wn * TN * 32, p.Co, p.Co, p.y, p.yst, lane,
                          [&](float v, int n) { return act_apply(v + (bias ? bias[n] : 0.f), act); });
    }
#endif
}

// ------------------------------------------------------------------------------------------
// conv_fwd16p_kernel (round 5): the 3x3 stride-1 reflect-pad-1 convolutions (every ResBlock convolution, networks.py:297-310) with the
// INPUT PATCH resident in LDS.
//
// What bounds conv_fwd16s on these layers is the ISSUE of its LDS-DMA copies, not their bytes and not the MFMAs: a
// `buffer_load_dwordx4 ... lds` costs its wave 60 - 185 cycles at issue (MI355X_MICROARCH.md, "LDS-DMA piece issue cost"), the 128 x 128
// tile needs 32 of them per k-tile = 8 per wave against 16 MFMAs of 32 cycles (profiles/r05_experiments.md section 3).  Per filter TAP
// the A tile is the same set of input pixels shifted by one: nine taps re-copy a (rows + 2) x (W + 2) neighbourhood nine times.
// Here a workgroup owns 256 consecutive output pixels = R = 256 / W full image rows; for each block of 64 input channels it copies the
// (R + 2) x (W + 2) reflect-padded patch ONCE (W = 64: 396 pixel rows of 128 bytes instead of 9 x 256) and runs all nine taps from it --
// the tap is a constant added to the patch row a lane's fragment read starts from.  8 waves (4 x 2, 64 x 64 each, the MFMA loop of
// conv_fwd16s), 256 x 128 tile: per k-tile 16 weight pieces + ~5.5 patch pieces over 8 waves = 2.7 LDS-DMA issues per wave and 16 MFMAs
// (conv_fwd16s: 8), the patch pieces spread one per tap.  LDS: 2 patch buffers of 400 rows + 2 weight buffers of 128 rows = 132 KB,
// one workgroup per CU, two waves per SIMD.  Same swizzle (chunk ^ (row >> 1) & 7 on the DMA source side and in the fragment read), keyed
// by the PATCH row: a lane group's 16 rows are consecutive pixels of one image row (W a multiple of 32), so the reads stay conflict-free.
// Same epilogue as conv_fwd16s (bias, activation, 16-bit stores, normalisation partials per 256-row tile).
//
// PP = 1 (a measured alternative, tuning fwd16_patch = 2; PP = 0, every wave in lockstep, is the default): the two waves of every SIMD work in COUNTER-PHASE (MI355X_MICROARCH.md "Two waves per SIMD"): the interval between
// two s_barriers is a LOAD segment for one half of the workgroup (waves 0-3: 16 ds_read_b128 of its next fragments, then its share of
// the LDS-DMA issues for the tile after next) and a COMPUTE segment (16 MFMAs on fragments already in registers) for the other half
// (waves 4-7, one per SIMD), and the roles swap at every barrier.  Measured: 50 us against 47 - 48 us for the lockstep form -- the MFMAs are
// not what the kernel waits for (profiles/r05_fwd16p_ablation.txt: without its MFMAs the kernel takes the same 50 us; without fragment reads
// AND copies 41 us; with none of the three 20 us = launch + prologue + epilogue + the barriers), see profiles/r05_experiments.md section 3.  Three weight buffers (a tile is overwritten two intervals after its last reader), copies are
// waited for at the end of the issuing wave's next COMPUTE segment: a full segment of latency cover, no wait in front of a read.
constexpr int PATCH_ROWS = 400;      // >= (256 / W + 2) * (W + 2) for W = 64 (396) and W = 32 (340)
#define PP_BARRIER() do { __builtin_amdgcn_sched_barrier(0); WG_BARRIER(); __builtin_amdgcn_sched_barrier(0); } while (0)
// ABL (measurement builds only, results are wrong): 1 = no MFMAs, 2 = no fragment reads in the loop, 4 = no copies in the loop
template <class T, int WN, int TN, int PP, int ABL = 0>
__global__ void __launch_bounds__(512, 1) conv_fwd16p_kernel(FwdSP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TM = 2, WM = 4, NW = WM * WN, NT = NW * 64, BM = WM * 64, BN = WN * TN * 32;
    static_assert(NW == 8, "eight waves");
    constexpr int NB = PP ? 3 : 2;                             // weight buffers
    constexpr int P_BYTES = PATCH_ROWS * ROWB, B_BYTES = BN * ROWB;
    constexpr int P_IT = (PATCH_ROWS / 8 + NW - 1) / NW;      // patch pieces (8 rows = 1 KB) per wave: 7
    constexpr int B_IT = BN / 8 / NW;                          // weight pieces per wave and k-tile: 2
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * P_BYTES + NB * B_BYTES + BM * 4];
    unsigned char* const sB = smem + 2 * P_BYTES;
    int* ro = reinterpret_cast<int*>(smem + 2 * P_BYTES + NB * B_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int tile = xcd_map(blockIdx.x, p.nwg);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const int W = p.Wo, H = p.Ho, hw = H * W;
    const int PW = W + 2, R = BM / W, P = (R + 2) * PW, npieces = (P + 7) >> 3;
    const int bimg = m0 / hw, y0 = (m0 - bimg * hw) / W;      // the tile = rows y0 .. y0 + R - 1 of image bimg

    for (int r = tid; r < BM; r += NT) ro[r] = m0 + r;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int cpt = p.Ci >> 6;                     // 64-channel blocks
    const int nk = 9 * cpt;

    // ---- LDS-DMA sources ----
    const int lr = lane >> 3, lj = lane & 7;
    int pvo[P_IT];                                 // patch piece n of this wave = piece n * 8 + wave: byte offset of this lane's row, chunk swizzled
#pragma unroll
    for (int n = 0; n < P_IT; ++n) {
        const int piece = n * NW + wave;
        int row = piece * 8 + lr;
        if (row > P - 1) row = P - 1;             // (rows past the patch: a valid address, never read)
        const int py = row / PW, px = row - py * PW;
        const int iy = refl(y0 - 1 + py, H), ix = refl(px - 1, W);
        const int prow = piece * 8 + lr;          // the LDS row it lands in decides the swizzle
        pvo[n] = (((bimg * H + iy) * W + ix) * p.Ci + (lj ^ ((prow >> 1) & 7)) * 8) * 2;
    }
    int bvo[B_IT];
#pragma unroll
    for (int n = 0; n < B_IT; ++n) {
        const int row = (BN / NW) * wave + 8 * n + lr;
        bvo[n] = (min(n0 + row, p.Co - 1) * p.K + (lj ^ ((row >> 1) & 7)) * 8) * 2;
    }
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x16, (long long)p.B * p.Hi * p.Wi * p.Ci * 2);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w16, (long long)p.Co * p.K * 2);

    auto issue_b = [&](int tap, int cc, int buf) __attribute__((always_inline)) {
        unsigned char* db = sB + buf * B_BYTES + ((BN / NW) * wave) * ROWB;
        const int koff = (tap * cpt + cc) * 128;  // the weights' k axis is (tap, cin)
#pragma unroll
        for (int n = 0; n < B_IT; ++n) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(db + n * 8 * ROWB), 16, bvo[n], koff, 0, 0);
    };
    auto issue_p = [&](int n, int cc, int buf) __attribute__((always_inline)) {      // patch piece n of this wave, channel block cc
        const int piece = n * NW + wave;
        if (piece < npieces)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(smem + buf * P_BYTES + piece * 8 * ROWB), 16, pvo[n], cc * 128, 0, 0);
    };

    // ---- fragment read bases ----
    const int l31 = lane & 31, kh = lane >> 5;
    int pp0[TM];                                   // patch row of this lane's output pixel (tap (0, 0)), per 32-row block
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int r = wm * 64 + t * 32 + l31;
        const int ry = r / W, rx_ = r - ry * W;
        pp0[t] = ry * PW + rx_;
    }
    const int cb0 = kh ^ ((l31 >> 1) & 7);         // weight rows: swizzle by the tile row, as in mma_tile
    const unsigned char* const bbase = sB + (wn * TN * 32 + l31) * ROWB;

    if constexpr (PP == 0) {
    // ---- lockstep schedule (the default): every wave copies, reads and multiplies in the same k-tile; two weight buffers, one barrier per k-tile ----
    // (measured alternatives, profiles/r05_experiments.md section 3: three weight buffers with the copies two k-tiles ahead 53 us; the
    //  counter-phase schedule below 50 us; this one 47 - 48 us)
    // prologue: the whole first patch and the first weight tile
#pragma unroll
    for (int n = 0; n < P_IT; ++n) issue_p(n, 0, 0);
    issue_b(0, 0, 0);
    int kt = 0;
    for (int cc = 0; cc < cpt; ++cc) {
        const unsigned char* const pbuf = smem + (cc & 1) * P_BYTES;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap, ++kt) {
            const int cur = kt & 1;
            DMA_SYNCTHREADS();                     // every copy issued so far has landed, for every wave; the other buffers are free
            if (kt + 1 < nk) { if (tap < 8) issue_b(tap + 1, cc, cur ^ 1); else issue_b(0, cc + 1, cur ^ 1); }
            if (tap < P_IT && cc + 1 < cpt) issue_p(tap, cc + 1, (cc + 1) & 1);      // the next patch, one piece per tap
            const int toff = (tap / 3) * PW + (tap % 3);
            const unsigned char* const bb = bbase + cur * B_BYTES;
            u32x4 fa[TM][4], fb[TN][4];
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const int pp = pp0[t] + toff;
                const int a0 = pp * ROWB + ((kh ^ ((pp >> 1) & 7)) << 4);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fa[t][ks] = *reinterpret_cast<const u32x4*>(pbuf + (a0 ^ (ks << 5)));
            }
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fb[t][ks] = *reinterpret_cast<const u32x4*>(bb + t * 32 * ROWB + ((cb0 ^ (2 * ks)) << 4));
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = T::mfma(fa[i][ks], fb[j][ks], acc[i][j]);
        }
    }

    } else {
    // ---- counter-phase schedule ----
    // intervals I0, I1, ... between consecutive barriers:   half 0: LOAD(0) COMPUTE(0) LOAD(1) COMPUTE(1) ...
    //                                                        half 1:  idle   LOAD(0) COMPUTE(0) LOAD(1)   ...
    // tile t is read in I(2t) (half 0) and I(2t+1) (half 1); its copies are issued in LOAD(t-2) of either half (I(2t-4), I(2t-3)), waited
    // for at the end of that wave's COMPUTE(t-2) (I(2t-3), I(2t-2)) and so published by the barrier in front of I(2t-1); they overwrite
    // tile t-3, last read in I(2t-5).  The next patch (one piece per wave and tap) lands in the other patch buffer the same way.
    const int half = wave >> 2;
#pragma unroll
    for (int n = 0; n < P_IT; ++n) issue_p(n, 0, 0);
    issue_b(0, 0, 0);
    if (nk > 1) issue_b(1 % 9, 1 / 9, 1);          // (tile 1: tap 1 of channel block 0 -- nk = 9 cpt >= 9)
    __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
    PP_BARRIER();                                   // patch 0 and tiles 0, 1 are in LDS for every wave; the row table too
    if (half == 1) PP_BARRIER();                    // half 1 sits out I0
    int tap = 0, cc = 0;                            // tile kt = (cc, tap)
    int tap2 = 2, cc2 = 0;                          // tile kt + 2
    int brd = 0, bwr = 2;                           // weight buffer of tile kt / of tile kt + 2
    int ty = 0, tx = 0;                             // tap = 3 ty + tx
    u32x4 fa[TM][4], fb[TN][4];
    for (int kt = 0; kt < nk; ++kt) {
        // ---- LOAD(kt): fragments first, copies last (nothing in front of the reads that could wait for a copy issued here) ----
        const unsigned char* const pbuf = smem + (cc & 1) * P_BYTES;
        const unsigned char* const bb = bbase + brd * B_BYTES;
        const int toff = ty * PW + tx;
        if ((ABL & 2) == 0 || kt == 0) {
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int pp = pp0[t] + toff;
            const int a0 = pp * ROWB + ((kh ^ ((pp >> 1) & 7)) << 4);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fa[t][ks] = *reinterpret_cast<const u32x4*>(pbuf + (a0 ^ (ks << 5)));
        }
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fb[t][ks] = *reinterpret_cast<const u32x4*>(bb + t * 32 * ROWB + ((cb0 ^ (2 * ks)) << 4));
        }
        if ((ABL & 4) == 0 && kt + 2 < nk) issue_b(tap2, cc2, bwr);
        if ((ABL & 4) == 0 && tap < P_IT && cc + 1 < cpt) {           // the next patch, one piece per wave and tap
            const int piece = tap * NW + wave;
            if (piece < npieces) {
                int off = pvo[0];
#pragma unroll
                for (int n = 1; n < P_IT; ++n) off = tap == n ? pvo[n] : off;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(smem + ((cc + 1) & 1) * P_BYTES + piece * 8 * ROWB), 16, off, (cc + 1) * 128, 0, 0);
            }
        }
        PP_BARRIER();
        // ---- COMPUTE(kt) ----
        if constexpr ((ABL & 1) == 0) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = T::mfma(fa[i][ks], fb[j][ks], acc[i][j]);
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { acc[0][0][ks] += __builtin_bit_cast(float, fa[0][ks][0] ^ fb[0][ks][1] ^ fa[1][ks][2] ^ fb[1][ks][3]); }
        }
        __builtin_amdgcn_sched_barrier(0);          // (the wait stays BEHIND the MFMAs: hoisted to the head of the segment it would stall them)
        __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));   // the copies of LOAD(kt) have had this whole segment to land
        PP_BARRIER();
        // next tile
        if (++tx == 3) { tx = 0; ++ty; }
        if (++tap == 9) { tap = 0; ty = 0; tx = 0; ++cc; }
        if (++tap2 == 9) { tap2 = 0; ++cc2; }
        brd = brd == NB - 1 ? 0 : brd + 1;
        bwr = bwr == NB - 1 ? 0 : bwr + 1;
    }
    if (half == 0) PP_BARRIER();                    // (every wave passes the same number of barriers)
    }

    const float* bias = p.bias;
    const int act = p.act;
    if (p.stats) {
        // (mean, M2) of the BM = 256 STORED outputs of each channel of this tile = one chunk partial of norm_finalize_* (as in conv_fwd16s)
        const int lh = lane >> 5;
        float2* red = reinterpret_cast<float2*>(smem);
        __syncthreads();                           // every wave is done with the operand buffers
        auto col = [&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            const int n = n0 + wn * TN * 32 + j * 32 + l31;
            const float bv = (bias && n < p.Co) ? bias[n] : 0.f;
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = act_apply(acc[i][j][r] + bv, act);
                    if (p.yst != ST_F32) v = st_unpack2(st_pack2(v, 0.f, p.yst), p.yst)[0];     // the value as it will be stored
                    acc[i][j][r] = v;
                    sum += v;
                }
            float mean = sum * (1.f / 32.f), q = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float dv = acc[i][j][r] - mean; q += dv * dv; }
            const float om = __shfl_xor(mean, 32), oq = __shfl_xor(q, 32);
            const float dm = om - mean;
            q = q + oq + dm * dm * 16.f;
            mean = 0.5f * (mean + om);
            if (lh == 0) red[wm * BN + wn * TN * 32 + j * 32 + l31] = make_float2(mean, q);
        };
        col(std::integral_constant<int, 0>{});
        if constexpr (TN > 1) col(std::integral_constant<int, 1>{});
        if constexpr (TN > 2) { col(std::integral_constant<int, 2>{}); col(std::integral_constant<int, 3>{}); }
        __syncthreads();
        if (tid < BN && n0 + tid < p.Co) {
            float cnt = 64.f, mean = red[tid].x, m2 = red[tid].y;
#pragma unroll
            for (int i = 1; i < WM; ++i) {
                const float2 o = red[i * BN + tid];
                const float dm = o.x - mean, tot = cnt + 64.f;
                mean += dm * (64.f / tot);
                m2 += o.y + dm * dm * (cnt * 64.f / tot);
                cnt = tot;
            }
            p.stats[(size_t)(m0 / BM) * p.Co + n0 + tid] = make_float2(mean, m2);
        }
        store_acc<TM, TN>(acc, ro, wm * 64, n0 + wn * TN * 32, p.Co, p.Co, p.y, p.yst, lane, [](float v, int) { return v; });
    } else {
        __syncthreads();                           // (ro was written before the main loop's first barrier; keep the waves together for the stores)
        store_acc<TM, TN>(acc, ro, wm * 64, n0 + wn * TN * 32, p.Co, p.Co, p.y, p.yst, lane,
                          [&](float v, int n) { return act_apply(v + (bias ? bias[n] : 0.f), act); });
    }
#endif
}
// shapes the patch kernel takes, and whether it is on (aclgan_tuning "fwd16_patch" / ACLGAN_FWD16_PATCH; default: see launch_fwd16s)
std::atomic<int> g_fwd16_patch{-1};
int fwd16_patch_mode() {
    int v = g_fwd16_patch.load();
    if (v < 0) { const char* e = getenv("ACLGAN_FWD16_PATCH"); v = e ? atoi(e) : 1; if (v < 0 || (v & 15) > 2) v = 1; g_fwd16_patch.store(v); }
    return v;
}
bool fwd16p_shape_ok(const ConvGeom& g) {
    if (!(g.k == 3 && g.s == 1 && g.p == 1 && g.up == 0 && g.Hi == g.Ho && g.Wi == g.Wo)) return false;
    if (!(g.Wo == 32 || g.Wo == 64) || (g.Ho * g.Wo) % 256 != 0 || g.Ho < 2) return false;
    if (g.Ci % 64 != 0 || g.Co % 128 != 0) return false;
    if ((256 / g.Wo + 2) * (g.Wo + 2) > PATCH_ROWS) return false;
    return (g.M / 256) * (g.Co / 128) >= 64;       // (one workgroup per CU: a quarter of the chip at least; smaller grids keep the 4-wave tiles)
}

// tile choice.  Measured inside the step (profiles/r03_experiments.md, same box back to back) the 8-wave tiles LOSE although they move fewer
// operand bytes per MFMA: bf16 B=8 56.3 ms with 128-row tiles against 60.4 with "largest tile that fills the chip", fp16 B=32 182.4 against
// 186.7 -- one 8-wave workgroup per CU has nothing to run while it waits at its barrier, two 4-wave workgroups cover each other.  So:
// 128-row tiles by default; ACLGAN_GLDS_TILE = 2 / 3 forces 256 x 128 / 256 x 256 where the shape allows, 4 = the largest-tile rule
// (kept tested: tests/test_gpu_ops16s.py runs every tile).  Returns 1 / 2 / 3.
std::atomic<int> g_tile_force{-1};      // (atomics: a switch set while another thread plans an update is seen old or new, never torn)
int glds_tile(int rows, int N) {
    int force = g_tile_force.load();
    if (force < 0) { const char* e = getenv("ACLGAN_GLDS_TILE"); force = e ? atoi(e) : 0; if (force < 0) force = 0; g_tile_force.store(force); }
    const int t256 = N % 256 == 0 ? cdiv(rows, 256) * (N / 256) : 0, t128 = N % 128 == 0 ? cdiv(rows, 256) * (N / 128) : 0;
    if (force == 3 && t256) return 3;
    if (force == 2 && t128) return 2;
    if (force == 4) return t256 >= 224 ? 3 : t128 >= 224 ? 2 : 1;      // "largest tile that fills the chip"
    // default: 128-row tiles -- except on grids of 8+ rounds of them (fp16 at its per-GPU batch 32: 2048 tiles on the 256-channel maps), where the
    // largest tile that still fills the chip twice wins: fp16 B=32 step 164.3 / 165.8 / 165.9 ms against 166.1 / 166.8 / 167.3 (round 6, same box
    // back to back); at B=8 (512 tiles) it loses (bf16 46.8 against 46.6 ms)
    if (force == 0 && N % 128 == 0 && cdiv(rows, 128) * (N / 128) >= 2048) return t256 >= 448 ? 3 : t128 >= 448 ? 2 : 1;
    return 1;
}

// ACLGAN_GLDS_SPEC = 1: the wave-specialised kernels instead of the unified ones (every wave copies and multiplies).  Measured: the
// specialised 128 x 128 kernel is SLOWER (ResBlock forward 69 vs 54 us; bf16 step 65.5 vs 57.2 ms), the 256 x 128 one wins isolated
// (50.3 vs 52.7 us) and loses in the step (60.1 vs 57.2 ms) -- profiles/r03_experiments.md; kept as a tested option
int g_spec = -1;
bool glds_spec() {
    if (g_spec < 0) { const char* e = getenv("ACLGAN_GLDS_SPEC"); g_spec = e ? atoi(e) : 0; }
    return g_spec != 0;
}

template <class T>
int launch_fwd16s(const ConvGeom& g, FwdSP p, hipStream_t st) {
    if (fwd16_patch_mode() && fwd16p_shape_ok(g)) {      // 3x3 ResBlock shapes: the input patch stays in LDS for all nine taps
        p.tiles_n = g.Co / 128; p.nwg = (g.M / 256) * p.tiles_n;
#ifdef ACLGAN_FWD16P_ABLATION
        const int abl = fwd16_patch_mode() >> 4;
        if (abl == 1) { hipLaunchKernelGGL((conv_fwd16p_kernel<T, 2, 2, 1, 1>), dim3(p.nwg), dim3(512), 0, st, p); ACL_CHECK_LAUNCH("abl"); return ACLGAN_OK; }
        if (abl == 2) { hipLaunchKernelGGL((conv_fwd16p_kernel<T, 2, 2, 1, 2>), dim3(p.nwg), dim3(512), 0, st, p); ACL_CHECK_LAUNCH("abl"); return ACLGAN_OK; }
        if (abl == 4) { hipLaunchKernelGGL((conv_fwd16p_kernel<T, 2, 2, 1, 4>), dim3(p.nwg), dim3(512), 0, st, p); ACL_CHECK_LAUNCH("abl"); return ACLGAN_OK; }
        if (abl == 6) { hipLaunchKernelGGL((conv_fwd16p_kernel<T, 2, 2, 1, 6>), dim3(p.nwg), dim3(512), 0, st, p); ACL_CHECK_LAUNCH("abl"); return ACLGAN_OK; }
        if (abl == 7) { hipLaunchKernelGGL((conv_fwd16p_kernel<T, 2, 2, 1, 7>), dim3(p.nwg), dim3(512), 0, st, p); ACL_CHECK_LAUNCH("abl"); return ACLGAN_OK; }
        if (abl == 5) { hipLaunchKernelGGL((conv_fwd16p_kernel<T, 2, 2, 1, 5>), dim3(p.nwg), dim3(512), 0, st, p); ACL_CHECK_LAUNCH("abl"); return ACLGAN_OK; }
        if (abl == 3) { hipLaunchKernelGGL((conv_fwd16p_kernel<T, 2, 2, 1, 3>), dim3(p.nwg), dim3(512), 0, st, p); ACL_CHECK_LAUNCH("abl"); return ACLGAN_OK; }
#endif
        if ((fwd16_patch_mode() & 15) == 2) hipLaunchKernelGGL((conv_fwd16p_kernel<T, 2, 2, 1>), dim3(p.nwg), dim3(512), 0, st, p);      // counter-phase form (measured alternative)
        else hipLaunchKernelGGL((conv_fwd16p_kernel<T, 2, 2, 0>), dim3(p.nwg), dim3(512), 0, st, p);
        ACL_CHECK_LAUNCH("conv_fwd16p_kernel");
        return ACLGAN_OK;
    }
    const int tc = glds_tile(g.M, g.Co);
    static int force = -1;
    if (force < 0) { const char* e = getenv("ACLGAN_GLDS_NBUF"); force = e ? atoi(e) : 0; }
    const bool sp = glds_spec();
    if (tc == 3) {
        p.tiles_n = g.Co / 256; p.nwg = cdiv(g.M, 256) * p.tiles_n;
        hipLaunchKernelGGL((conv_fwd16s_kernel<T, 4, 2, 4, 2, 0>), dim3(p.nwg), dim3(512), 0, st, p);
    } else if (tc == 2) {
        p.tiles_n = g.Co / 128; p.nwg = cdiv(g.M, 256) * p.tiles_n;
        if (sp) hipLaunchKernelGGL((conv_fwd16s_kernel<T, 4, 2, 2, 3, 4>), dim3(p.nwg), dim3(768), 0, st, p);      // 8 consumers + 4 producers, 3 x 48 KB
        else hipLaunchKernelGGL((conv_fwd16s_kernel<T, 4, 2, 2, 2, 0>), dim3(p.nwg), dim3(512), 0, st, p);
    } else {
        const int BN = g.Co % 128 == 0 ? 128 : 64;
        p.tiles_n = g.Co / BN; p.nwg = cdiv(g.M, 128) * p.tiles_n;
        // (the single-buffer variant, 4 workgroups per CU, wins on ISOLATED large grids -- B=32 ResBlock shape 173 vs 200 us -- but loses
        //  badly inside the step: fp16 B=32 step 237.6 vs 185.0 ms, profiles/r03_experiments.md; ACLGAN_GLDS_NBUF=1 selects it)
        if (BN == 128 && force == 1) hipLaunchKernelGGL((conv_fwd16s_kernel<T, 2, 2, 2, 1, 0>), dim3(p.nwg), dim3(256), 0, st, p);
        else if (BN == 128 && sp) hipLaunchKernelGGL((conv_fwd16s_kernel<T, 2, 2, 2, 2, 2>), dim3(p.nwg), dim3(384), 0, st, p);      // 4 consumers + 2 producers, 2 x 32 KB, 2 workgroups per CU
        else if (BN == 128) hipLaunchKernelGGL((conv_fwd16s_kernel<T, 2, 2, 2, 2, 0>), dim3(p.nwg), dim3(256), 0, st, p);
        else if (sp) hipLaunchKernelGGL((conv_fwd16s_kernel<T, 2, 2, 1, 2, 2>), dim3(p.nwg), dim3(384), 0, st, p);
        else hipLaunchKernelGGL((conv_fwd16s_kernel<T, 2, 2, 1, 2, 0>), dim3(p.nwg), dim3(256), 0, st, p);
    }
    ACL_CHECK_LAUNCH("conv_fwd16s_kernel");
    return ACLGAN_OK;
}

// ------------------------------------------------------------------------------------------
// dgrad over the padded grid (GEMM rows = padded positions of one stride-parity class, k = (tap of the class, cout), n = cin)
// ------------------------------------------------------------------------------------------
struct DgSP {
    const u16* dy16; const u16* w16t; void* dxp;
    int B, Ho, Wo, Co, Ci, k, s, Hp, Wp, Hc, Wc, Mc, tiles_n, nwg, pst;
    // direct mode (dx stored in the same 16-bit dtype as the padded partials): a padded position whose dx pixel has NO mirrored partner is
    // written (or accumulated) straight into dx; only the ring and the border band go through the padded scratch and the ordered fold
    void* dx; int direct, accumulate, Hi, Wi, pad;
};
__device__ __forceinline__ bool fold_band(int u, int n, int p) { return (u >= 1 && u <= p) || (u <= n - 2 && u >= n - 1 - p); }      // has mirrored partners

template <class T, int WM, int WN, int TN, int NBUF, int NP>
__global__ void __launch_bounds__((WM * WN + NP) * 64, NP ? 3 : 2) conv_dgrad16s_kernel(DgSP p) {
#if defined(__HIP_DEVICE_COMPILE__)   // (device pass only, see conv_fwd16s_kernel)
    constexpr int TM = 2, NW = WM * WN, NT = (NW + NP) * 64, BM = WM * 64, BN = WN * TN * 32;
    constexpr int NL = NP ? NP : NW;
    constexpr int A_IT = BM / 8 / NL, B_IT = BN / 8 / NL;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int WAIT_NEXT = vmcnt_imm(NBUF >= 2 ? (NBUF - 2) * (A_IT + B_IT) : 0);      // producers: everything but the newest NBUF - 2 tiles has landed
    static_assert(NP || NBUF == 2, "the unified loop is double-buffered");
    __shared__ __attribute__((aligned(16))) unsigned char smem[NBUF * (A_BYTES + B_BYTES) + BM * 4];
    int* ro = reinterpret_cast<int*>(smem + NBUF * (A_BYTES + B_BYTES));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool prod = NP && wave >= NW;
    const int lw = NP ? wave - NW : wave;
    const int wm = wave / WN, wn = wave % WN;
    const int tile = xcd_map(blockIdx.x, p.nwg);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const int cls = blockIdx.z, cy = cls / p.s, cx = cls - cy * p.s;
    const int Tx = (p.k - cx + p.s - 1) / p.s, Ty = (p.k - cy + p.s - 1) / p.s;
    const int hwc = p.Hc * p.Wc;

    for (int r = tid; r < BM; r += NT) {
        const int m = m0 + r;
        int oo = -1;
        if (m < p.Mc) {
            const int b = m / hwc, rem = m - b * hwc, y2 = rem / p.Wc, x2 = rem - y2 * p.Wc;
            const int py = y2 * p.s + cy, px = x2 * p.s + cx;
            if (py < p.Hp && px < p.Wp) {
                oo = (b * p.Hp + py) * p.Wp + px;
                const int i = py - p.pad, j = px - p.pad;
                if (p.direct && (unsigned)i < (unsigned)p.Hi && (unsigned)j < (unsigned)p.Wi && !fold_band(i, p.Hi, p.pad) && !fold_band(j, p.Wi, p.pad))
                    oo = -2 - ((b * p.Hi + i) * p.Wi + j);          // (<= -2: index of the dx pixel)
            }
        }
        ro[r] = oo;
    }
    if (NP) __syncthreads();
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int cpt = p.Co >> 6;
    const int nk = Ty * Tx * cpt;

    if (!NP || prod) {
        const int lr = lane >> 3, lj = lane & 7;
        int ay[A_IT], ax[A_IT], ab[A_IT], acs[A_IT];
#pragma unroll
        for (int n = 0; n < A_IT; ++n) {
            const int row = (BM / NL) * lw + 8 * n + lr;
            const int m = m0 + row;
            if (m < p.Mc) {
                const int b = m / hwc, rem = m - b * hwc, y2 = rem / p.Wc;
                ay[n] = y2; ax[n] = rem - y2 * p.Wc; ab[n] = b;
            } else { ay[n] = -100000; ax[n] = -100000; ab[n] = 0; }      // reads zero for every tap
            acs[n] = (lj ^ ((row >> 1) & 7)) * 8;
        }
        // B rows = input channels n, k = 64 consecutive cout of one tap: w16t[tap][n][cout]
        int bvo[B_IT];
#pragma unroll
        for (int n = 0; n < B_IT; ++n) {
            const int row = (BN / NL) * lw + 8 * n + lr;
            bvo[n] = (min(n0 + row, p.Ci - 1) * p.Co + (lj ^ ((row >> 1) & 7)) * 8) * 2;
        }
        const __amdgpu_buffer_rsrc_t rdy = make_rsrc(p.dy16, (long long)p.B * p.Ho * p.Wo * p.Co * 2);
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w16t, (long long)p.k * p.k * p.Ci * p.Co * 2);
        int avo[A_IT];
        int f_tap = -1, tapoff = 0;

        auto issue = [&](int kt, int buf) __attribute__((always_inline)) {
            const int t = kt / cpt, cc = kt - t * cpt;
            if (t != f_tap) {
                f_tap = t;
                const int ty = t / Tx, tx = t - ty * Tx;
                tapoff = ((cy + p.s * ty) * p.k + (cx + p.s * tx)) * p.Ci * p.Co * 2;
#pragma unroll
                for (int n = 0; n < A_IT; ++n) {
                    const int oy = ay[n] - ty, ox = ax[n] - tx;
                    const bool ok = (unsigned)oy < (unsigned)p.Ho && (unsigned)ox < (unsigned)p.Wo;
                    avo[n] = ok ? (((ab[n] * p.Ho + oy) * p.Wo + ox) * p.Co + acs[n]) * 2 : OOB;
                }
            }
            unsigned char* da = smem + buf * A_BYTES + ((BM / NL) * lw) * ROWB;
            unsigned char* db = smem + NBUF * A_BYTES + buf * B_BYTES + ((BN / NL) * lw) * ROWB;
#pragma unroll
            for (int n = 0; n < A_IT; ++n) __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, LDS_PTR(da + n * 8 * ROWB), 16, avo[n], cc * 128, 0, 0);
#pragma unroll
            for (int n = 0; n < B_IT; ++n) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(db + n * 8 * ROWB), 16, bvo[n], tapoff + cc * 128, 0, 0);
        };

        if (NP) {                                      // producer wave (see conv_fwd16s_kernel)
#pragma unroll
            for (int t = 0; t < NBUF - 1; ++t)
                if (t < nk) issue(t, t);
            if (NBUF - 1 <= nk) __builtin_amdgcn_s_waitcnt(WAIT_NEXT);
            else __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
            WG_BARRIER();
            int ib = NBUF - 1;
            for (int kt = 0; kt < nk; ++kt) {
                if (kt + NBUF - 1 < nk) {
                    issue(kt + NBUF - 1, ib);
                    __builtin_amdgcn_s_waitcnt(WAIT_NEXT);
                } else {
                    __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
                }
                ib = ib + 1 == NBUF ? 0 : ib + 1;
                WG_BARRIER();
            }
        } else {
            issue(0, 0);
            for (int kt = 0; kt < nk; ++kt) {
                const int cur = kt & 1;
                DMA_SYNCTHREADS();
                if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
                mma_tile<T, TM, TN>(smem + cur * A_BYTES + (wm * 64) * ROWB, smem + NBUF * A_BYTES + cur * B_BYTES + (wn * TN * 32) * ROWB, lane, acc);
            }
        }
    } else {                                           // consumer wave
        WG_BARRIER();
        int cb = 0;
        for (int kt = 0; kt < nk; ++kt) {
            mma_tile<T, TM, TN>(smem + cb * A_BYTES + (wm * 64) * ROWB, smem + NBUF * A_BYTES + cb * B_BYTES + (wn * TN * 32) * ROWB, lane, acc);
            cb = cb + 1 == NBUF ? 0 : cb + 1;
            WG_BARRIER();
        }
    }
    if (!prod) {
        if (p.direct) store_acc_dx<TM, TN>(acc, ro, wm * 64, n0 + wn * TN * 32, p.Ci, p.dxp, p.dx, p.pst, p.accumulate, lane);
        else store_acc<TM, TN>(acc, ro, wm * 64, n0 + wn * TN * 32, p.Ci, p.Ci, p.dxp, p.pst, lane, [](float v, int) { return v; });
    }
#endif
}

template <class T>
int launch_dgrad16s(const ConvGeom& g, DgSP p, hipStream_t st) {
    const int tc = glds_tile(p.Mc * g.s * g.s, g.Ci);      // (rows of all stride-parity classes together fill the chip)
    const dim3 z(1, 1, g.s * g.s);
    const bool sp = glds_spec();
    if (tc == 3) {
        p.tiles_n = g.Ci / 256; p.nwg = cdiv(p.Mc, 256) * p.tiles_n;
        hipLaunchKernelGGL((conv_dgrad16s_kernel<T, 4, 2, 4, 2, 0>), dim3(p.nwg, 1, z.z), dim3(512), 0, st, p);
    } else if (tc == 2) {
        p.tiles_n = g.Ci / 128; p.nwg = cdiv(p.Mc, 256) * p.tiles_n;
        if (sp) hipLaunchKernelGGL((conv_dgrad16s_kernel<T, 4, 2, 2, 3, 4>), dim3(p.nwg, 1, z.z), dim3(768), 0, st, p);
        else hipLaunchKernelGGL((conv_dgrad16s_kernel<T, 4, 2, 2, 2, 0>), dim3(p.nwg, 1, z.z), dim3(512), 0, st, p);
    } else if (g.Ci % 128 == 0) {
        p.tiles_n = g.Ci / 128; p.nwg = cdiv(p.Mc, 128) * p.tiles_n;
        if (sp) hipLaunchKernelGGL((conv_dgrad16s_kernel<T, 2, 2, 2, 2, 2>), dim3(p.nwg, 1, z.z), dim3(384), 0, st, p);
        else hipLaunchKernelGGL((conv_dgrad16s_kernel<T, 2, 2, 2, 2, 0>), dim3(p.nwg, 1, z.z), dim3(256), 0, st, p);
    } else {
        p.tiles_n = g.Ci / 64; p.nwg = cdiv(p.Mc, 128) * p.tiles_n;
        if (sp) hipLaunchKernelGGL((conv_dgrad16s_kernel<T, 2, 2, 1, 2, 2>), dim3(p.nwg, 1, z.z), dim3(384), 0, st, p);
        else hipLaunchKernelGGL((conv_dgrad16s_kernel<T, 2, 2, 1, 2, 0>), dim3(p.nwg, 1, z.z), dim3(256), 0, st, p);
    }
    ACL_CHECK_LAUNCH("conv_dgrad16s_kernel");
    return ACLGAN_OK;
}

// padded-grid gradient (storage pst) -> dx (storage dst): reflection_pad2d backward as an ordered gather, 4 channels per thread
struct FoldSP { const void* dxp; void* dx; int B, Hi, Wi, Ci, Hp, Wp, p, accumulate, pst, dst; int64_t total; int band_only; };

__device__ __forceinline__ int fold_alias(int u, int n, int p, int* q) {     // padded positions q with reflect(q - p, n) == u
    int c = 0;
    q[c++] = u + p;
    if (u >= 1 && u <= p) q[c++] = p - u;
    if (u <= n - 2 && u >= n - 1 - p) q[c++] = p + 2 * (n - 1) - u;
    return c;
}
__global__ void __launch_bounds__(256) conv_fold_st_kernel(FoldSP f) {
    // grid (row blocks, Hi, B): the image row and the sample are block indices, one 32-bit division per thread (round 6; the flat 64-bit
    // index cost four 64-bit divisions per 8-byte element group and the kernel moved 2.6 TB/s)
    const int C4 = f.Ci >> 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= f.Wi * C4) return;
    const int j = t / C4, c4 = t - j * C4;
    const int i = blockIdx.y, b = blockIdx.z;
    int qy[3], qx[3];
    const int ny = fold_alias(i, f.Hi, f.p, qy), nx = fold_alias(j, f.Wi, f.p, qx);
    if (f.band_only && ny * nx == 1) return;          // (direct mode of conv_dgrad16s: this pixel was written by the GEMM epilogue)
    const int64_t idx = ((int64_t)(b * f.Hi + i) * f.Wi + j) * C4 + c4;
    st_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < ny; ++a)
        for (int e = 0; e < nx; ++e) acc += st_ld4(f.dxp, ((int64_t)(b * f.Hp + qy[a]) * f.Wp + qx[e]) * C4 + c4, f.pst);
    if (f.accumulate) acc += st_ld4(f.dx, idx, f.dst);
    st_st4(f.dx, idx, acc, f.dst);
}

bool shape_ok(const ConvGeom& g);
bool enabled();

// ------------------------------------------------------------------------------------------
// wgrad on 16-bit x and dy:  dW[co][tap][ci] = sum over pixels of dy[p][co] * x[src(p, tap)][ci]
// The GEMM k axis is the PIXEL index, the slow axis of both NHWC operands, while an MFMA lane wants 8 consecutive k of one channel.
// conv_fast16.hip transposes 8 pixels x 4 channels per staging thread in registers (v_perm) and writes channel-major LDS rows.  Here
// both operand tiles stay pixel-major -- [64 pixels][128 channels], a pixel row is 256 contiguous bytes of HBM, so they are plain
// LDS-DMA copies (x through a per-pixel gather table: reflection + filter tap) -- and the transposition happens in the LDS READ:
// ds_read_b64_tr_b16 hands lane l channel (l & 15) of the 4 pixel rows its 16-lane group addresses (measured layout:
// scripts/microbench/tr16_layout.hip), two of them make one 32x32x16 operand.  Bank conflicts of those reads (four pixel rows,
// 256 B apart) are removed by XOR-ing the 16-byte chunk index with (pixel & 3) << 2 on the DMA source side and in the read address.
// Pixel slices (blockIdx.z) store their 128 x 128 tiles, wgrad16s_finish_kernel adds them in order: no atomics, bit-reproducible.
// The bias gradient (column sums of dy) is a separate ordered reduction (colsum16_kernel): no operand ever passes through registers.
// ------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct WgSP {
    const u16* x16; const u16* dy16; float* part; float* part_b; float* dw; float* db;
    int B, Hi, Wi, Ci, Ho, Wo, Co, k, s, p, P, Kn, chunk, tiles_n, nwg, splits, nsub, sub;
};

template <class T>
__global__ void __launch_bounds__(256, 2) conv_wgrad16s_kernel(WgSP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BK = 64, CH = 1024, PROW = 256;       // pixels per k-tile, pixels per gather-table refill, bytes per LDS pixel row (128 channels)
    constexpr int T_BYTES = BK * PROW;                  // one operand tile
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * T_BYTES + (CH + BK) * 4];
    int* pinfo = reinterpret_cast<int*>(smem + 4 * T_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = xcd_map(blockIdx.x, p.nwg);
    const int m0 = (tile / p.tiles_n) * 128, n0 = (tile % p.tiles_n) * 128;
    const int tap = n0 / p.Ci, ci0 = n0 - tap * p.Ci, ky = tap / p.k, kx = tap - ky * p.k;     // the whole N tile lies inside one filter tap
    const int pbeg = blockIdx.z * p.chunk, pend = min(p.P, pbeg + p.chunk);
    const int hw = p.Ho * p.Wo;
    const __amdgpu_buffer_rsrc_t rdy = make_rsrc(p.dy16, (long long)p.P * p.Co * 2);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x16, (long long)p.B * p.Hi * p.Wi * p.Ci * 2);
    // DMA roles: instruction n (0..3) of wave w fills pixel rows 16 w + 4 n + (lane >> 4) of a tile, the lane its 16-byte chunk (lane & 15)
    const int lp = lane >> 4;
    const int swz = ((lane & 15) ^ (lp << 2)) * 16;     // source chunk of LDS position (lane & 15) in a row with pixel & 3 == lp
    // fragment reads: lane l addresses, as member s = l & 15 of its 16-lane group, pixel row 8 (l >> 5) + (s >> 2) [+ 4], channels
    // 16 ((l >> 4) & 1) + 4 (s & 3) .. + 3 of a 32-channel block, and receives channel (l & 31) of those 4 pixels
    const int sl = lane & 15, sq = sl >> 2;
    int fo_a[2], fo_b[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ca = wm * 64 + t * 32 + 16 * ((lane >> 4) & 1) + 4 * (sl & 3), cbn = wn * 64 + t * 32 + 16 * ((lane >> 4) & 1) + 4 * (sl & 3);
        fo_a[t] = (8 * (lane >> 5) + sq) * PROW + (((ca >> 3) ^ (sq << 2)) << 4) + (ca & 7) * 2;
        fo_b[t] = (8 * (lane >> 5) + sq) * PROW + (((cbn >> 3) ^ (sq << 2)) << 4) + (cbn & 7) * 2;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int cb = pbeg; cb < pend; cb += CH) {
        const int ce = min(pend, cb + CH);
        for (int i = tid; i < CH + BK; i += 256) {       // source pixel of this tile's tap per sub-chunk pixel; -1 past the slice: reads zero
            const int pix = cb + i;
            int v = -1;
            if (pix < ce) {
                const int b = pix / hw, rem = pix - b * hw, oy = rem / p.Wo, ox = rem - oy * p.Wo;
                v = (b * p.Hi + refl(oy * p.s - p.p + ky, p.Hi)) * p.Wi + refl(ox * p.s - p.p + kx, p.Wi);
            }
            pinfo[i] = v;
        }
        __syncthreads();
        const int nkt = (ce - cb + BK - 1) / BK;
        auto piece = [&](int kt, int n, int buf) __attribute__((always_inline)) {      // pixel rows 16 wave + 4 n .. + 3 of both operand tiles
            unsigned char* da = smem + buf * 2 * T_BYTES + (16 * wave) * PROW;
            unsigned char* db = da + T_BYTES;
            const int pl = kt * BK + 16 * wave + 4 * n + lp;
            const int sp = pinfo[pl];
            const int va = sp >= 0 ? ((cb + pl) * p.Co + m0) * 2 + swz : OOB;
            const int vb = sp >= 0 ? (sp * p.Ci + ci0) * 2 + swz : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, LDS_PTR(da + n * 4 * PROW), 16, va, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(db + n * 4 * PROW), 16, vb, 0, 0, 0);
        };
#pragma unroll
        for (int n = 0; n < 4; ++n) piece(0, n, 0);
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1;
            DMA_SYNCTHREADS();
            if (kt + 1 < nkt) {
#pragma unroll
                for (int n = 0; n < 4; ++n) piece(kt + 1, n, cur ^ 1);
            }
            const unsigned char* ta = smem + cur * 2 * T_BYTES;
            const unsigned char* tb = ta + T_BYTES;
            u32x4 fa[4][2], fb[4][2];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ta + fo_a[t] + ks * 16 * PROW));
                    const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ta + fo_a[t] + ks * 16 * PROW + 4 * PROW));
                    const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tb + fo_b[t] + ks * 16 * PROW));
                    const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tb + fo_b[t] + ks * 16 * PROW + 4 * PROW));
                    const uint2 a0u = __builtin_bit_cast(uint2, a0), a1u = __builtin_bit_cast(uint2, a1);
                    const uint2 b0u = __builtin_bit_cast(uint2, b0), b1u = __builtin_bit_cast(uint2, b1);
                    fa[ks][t] = (u32x4){a0u.x, a0u.y, a1u.x, a1u.y};
                    fb[ks][t] = (u32x4){b0u.x, b0u.y, b1u.x, b1u.y};
                }
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = T::mfma(fa[ks][i], fb[ks][j], acc[i][j]);
        }
        __syncthreads();       // the gather table and both buffers are free for the next sub-chunk
    }
    // this slice's tile, natural [128 co][128 ci] order: 128-byte coalesced rows
    float* pt = p.part + ((size_t)blockIdx.z * p.nwg + tile) * (128 * 128);
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) pt[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 128 + wn * 64 + j * 32 + l31] = acc[i][j][r];
#endif
}

// part_b[sub][c] = sum of dy16[p][c] over the sub-slice's pixels (64 channels x 16 pixel lanes per workgroup, ordered LDS combine)
__global__ void __launch_bounds__(256) colsum16_kernel(WgSP p, int st) {
    __shared__ st_f32x4 red[256];
    const int c4 = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int cblk = blockIdx.x, z = blockIdx.y;
    const int p0 = z * p.sub, p1 = min(p.P, p0 + p.sub);
    const int C4 = p.Co >> 2;
    st_f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int px = p0 + pl; px < p1; px += 16) s += st_ld4(p.dy16, (int64_t)px * C4 + cblk * 16 + c4, st);
    red[threadIdx.x] = s;
    __syncthreads();
    if (pl == 0) {
        for (int i = 1; i < 16; ++i) s += red[i * 16 + c4];
        *reinterpret_cast<st_f32x4*>(p.part_b + (size_t)z * p.Co + cblk * 64 + c4 * 4) = s;
    }
}

// dw[m][n..n+3] += sum over slices (in order) of the partial tiles; db[m] += sum over the column-sum partials (fixed order: 16 groups
// of consecutive partials per channel, then the groups).  (The first version added the up to 256 bias partials of a channel in ONE
// thread, one dependent L2 round trip each: 65 us for a launch whose GEMM takes 54 us.)
__global__ void __launch_bounds__(256) wgrad16s_finish_kernel(WgSP p) {
    const int N4 = p.Kn >> 2;
    const int64_t n = (int64_t)p.Co * N4;
    const size_t zs = (size_t)p.nwg * (128 * 128);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int n4 = (int)(i % N4), m = (int)(i / N4);
        const int tile = (m >> 7) * p.tiles_n + ((n4 * 4) >> 7);
        const float* pt = p.part + (size_t)tile * (128 * 128) + (size_t)(m & 127) * 128 + ((n4 * 4) & 127);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int z = 0;
        for (; z + 4 <= p.splits; z += 4) {          // four loads in flight, added in slice order
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(pt + (size_t)z * zs), a1 = *reinterpret_cast<const f32x4*>(pt + (size_t)(z + 1) * zs);
            const f32x4 a2 = *reinterpret_cast<const f32x4*>(pt + (size_t)(z + 2) * zs), a3 = *reinterpret_cast<const f32x4*>(pt + (size_t)(z + 3) * zs);
            s += a0; s += a1; s += a2; s += a3;
        }
        for (; z < p.splits; ++z) s += *reinterpret_cast<const f32x4*>(pt + (size_t)z * zs);
        f32x4* o = reinterpret_cast<f32x4*>(p.dw + (size_t)m * p.Kn + n4 * 4);
        *o += s;
    }
    if (p.db && (int)blockIdx.x * 16 < p.Co) {       // block b: channels 16 b .. 16 b + 15
        __shared__ float red[16][17];
        const int c = blockIdx.x * 16 + (threadIdx.x & 15), zg = threadIdx.x >> 4;
        const int per = (p.nsub + 15) / 16;
        float sg = 0.f;
        if (c < p.Co)
            for (int zz = zg * per; zz < min(p.nsub, (zg + 1) * per); ++zz) sg += p.part_b[(size_t)zz * p.Co + c];
        red[zg][threadIdx.x & 15] = sg;
        __syncthreads();
        if (threadIdx.x < 16 && c < p.Co) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += red[q][threadIdx.x];
            p.db[c] += t;
        }
    }
}

struct WgSPlan { int tiles_n, nwg, splits, chunk, nsub, sub; size_t part_bytes, partb_bytes; };
WgSPlan wgrad16s_plan(const ConvGeom& g) {
    WgSPlan q;
    q.tiles_n = g.K / 128;
    q.nwg = (g.Co / 128) * q.tiles_n;
    int splits = std::max(1, 512 / q.nwg);
    splits = std::max(1, std::min(splits, cdiv(g.M, 512)));
    q.chunk = cdiv(cdiv(g.M, splits), 64) * 64;
    q.splits = cdiv(g.M, q.chunk);
    q.nsub = std::max(1, std::min(256, g.M / 64));
    q.sub = cdiv(g.M, q.nsub);
    q.nsub = cdiv(g.M, q.sub);
    q.part_bytes = ((size_t)q.splits * q.nwg * 128 * 128 * sizeof(float) + 255) & ~(size_t)255;
    q.partb_bytes = ((size_t)q.nsub * g.Co * sizeof(float) + 255) & ~(size_t)255;
    return q;
}
bool wgrad16s_shape_ok(const ConvGeom& g) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("ACLGAN_NOWGRAD16S"); off = (e && atoi(e)) ? 1 : 0; }
    // (with the parallel bias finish it wins at every size measured against the register-transposing kernel: B=8 ResBlock 69.5 vs 117.7 us,
    //  CE2 62.6 vs 101.7, discriminator 128->256 / 256->512 51.5 / 48.3 vs 77.9 / 76.7, 16x16 maps 26.3 vs 34.9, B=32 214 vs 408 us;
    //  ACLGAN_WGRAD16S_MINPIX sets a pixel-count threshold)
    static int minpix = -1;
    if (minpix < 0) { const char* e = getenv("ACLGAN_WGRAD16S_MINPIX"); minpix = e ? atoi(e) : 64; }
    return !off && enabled() && shape_ok(g) && g.Co % 128 == 0 && g.Ci % 128 == 0 && g.M >= std::max(64, minpix);
}

bool shape_ok(const ConvGeom& g) {
    return fast_enabled() && g.up == 0 && g.Ci % 64 == 0 && g.Co % 64 == 0 && g.k >= 1 && g.p < g.Hi && g.p < g.Wi;
}
bool enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("ACLGAN_NOGLDS16"); v = (e && atoi(e)) ? 0 : 1; }
    return v == 1;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// entry points
// ------------------------------------------------------------------------------------------
// which: 0 forward, 1 dgrad.  The forward also wants a grid that fills the chip without split-K (small late-discriminator maps keep the
// split-K kernel of conv_fast16.hip, which reads the same 16-bit activations through its A16 path).
// tuning / test knob behind aclgan_set_tuning("dgrad16s_direct", v); returns the previous value
static std::atomic<int> g_dgrad_direct{-1};
int set_dgrad16s_direct(int v) {
    if (g_dgrad_direct.load() < 0) { const char* e = getenv("ACLGAN_DGRAD16S_DIRECT"); g_dgrad_direct.store(e ? (atoi(e) ? 1 : 0) : 0); }
    return g_dgrad_direct.exchange(v ? 1 : 0);
}
// tuning knob "fwd16_patch": 1 = the 3x3 stride-1 layers it fits run on conv_fwd16p_kernel (input patch resident in LDS), 0 = on conv_fwd16s
int set_fwd16_patch(int v) { const int old = fwd16_patch_mode(); g_fwd16_patch.store((v < 0 || (v & 15) > 2) ? 1 : v); return old; }
// tuning / test knob behind aclgan_set_tuning("glds_tile", v): same values as ACLGAN_GLDS_TILE; returns the previous value
int set_glds_tile(int v) {
    if (g_tile_force.load() < 0) glds_tile(1, 1);
    return g_tile_force.exchange(v < 0 ? 0 : v);
}

bool conv16s_ok(const ConvGeom& g, int which) {
    if (!enabled() || !shape_ok(g)) return false;
    if (which == 0) {
        const int BN = g.Co % 128 == 0 ? 128 : 64;
        return cdiv(g.M, 128) * (g.Co / BN) >= 96 || g.K <= 1024;      // (a grid that fills the chip without split-K)
    }
    return true;
}

// > 0: conv_fwd16s can emit the normalisation statistics of its output ((mean, M2) per 128-pixel chunk and channel) from its epilogue
int conv_fwd16s_stats_chunk(const ConvGeom& g) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("ACLGAN_NOSTATFUSE"); off = (e && atoi(e)) ? 1 : 0; }
    if (off || !conv16s_ok(g, 0)) return 0;
    const int rows = (fwd16_patch_mode() && fwd16p_shape_ok(g)) ? 256 : (glds_tile(g.M, g.Co) >= 2 ? 256 : 128);      // the statistics chunk is the launch's row tile
    return (g.Ho * g.Wo) % rows == 0 ? rows : 0;
}

int conv_fwd16s(const ConvGeom& g, int dtype, const void* x16, const void* w16, const float* bias, void* y, int yst, hipStream_t st, float* stats) {
    if (!conv16s_ok(g, 0)) return ACLGAN_EUNSUPPORTED;
    if (stats && !conv_fwd16s_stats_chunk(g)) { set_error("conv_fwd16s: statistics are not offered for this shape"); return ACLGAN_EINVAL; }
    ACL_REQUIRE(x16 && w16 && y, "conv_fwd16s: null operand");
    ACL_REQUIRE((long long)g.B * g.Hi * g.Wi * g.Ci * 2 < 0x7fffffe0ll && (long long)g.Co * g.K * 2 < 0x7fffffe0ll, "conv_fwd16s: operand beyond 2 GB");
    FwdSP p;
    p.x16 = (const u16*)x16; p.w16 = (const u16*)w16; p.bias = bias; p.y = y; p.yst = yst; p.stats = (float2*)stats;
    p.B = g.B; p.Hi = g.Hi; p.Wi = g.Wi; p.Ci = g.Ci; p.Ho = g.Ho; p.Wo = g.Wo; p.Co = g.Co; p.k = g.k; p.s = g.s; p.p = g.p;
    p.M = g.M; p.K = g.K; p.act = g.act; p.tiles_n = 0; p.nwg = 0;
    if (dtype == ACLGAN_DTYPE_BF16) return launch_fwd16s<QBF16>(g, p, st);
    if (dtype == ACLGAN_DTYPE_FP16) return launch_fwd16s<QFP16>(g, p, st);
    set_error("conv_fwd16s: dtype %d", dtype);
    return ACLGAN_EINVAL;
}

size_t conv_dgrad16s_scratch_bytes(const ConvGeom& g) {
    return conv16s_ok(g, 1) ? (((size_t)g.B * g.Hp * g.Wp * g.Ci * 2 + 255) & ~(size_t)255) : 0;
}

// dx (storage dxst) (+)= dgrad; scratch: conv_dgrad16s_scratch_bytes(g) (the padded-grid gradient, stored in the 16-bit dtype)
int conv_dgrad16s(const ConvGeom& g, int dtype, const void* dy16, const void* w16t, void* dx, int dxst, int accumulate, void* scratch, hipStream_t st) {
    if (!conv16s_ok(g, 1)) return ACLGAN_EUNSUPPORTED;
    ACL_REQUIRE(dy16 && w16t && dx && scratch, "conv_dgrad16s: null operand / scratch");
    ACL_REQUIRE((long long)g.B * g.Ho * g.Wo * g.Co * 2 < 0x7fffffe0ll && (long long)g.K * g.Co * 2 < 0x7fffffe0ll, "conv_dgrad16s: operand beyond 2 GB");
    DgSP p;
    p.dy16 = (const u16*)dy16; p.w16t = (const u16*)w16t; p.dxp = scratch; p.pst = dtype;
    p.B = g.B; p.Ho = g.Ho; p.Wo = g.Wo; p.Co = g.Co; p.Ci = g.Ci; p.k = g.k; p.s = g.s; p.Hp = g.Hp; p.Wp = g.Wp;
    p.Hc = cdiv(g.Hp, g.s); p.Wc = cdiv(g.Wp, g.s); p.Mc = g.B * p.Hc * p.Wc; p.tiles_n = 0; p.nwg = 0;
    // direct mode (ACLGAN_DGRAD16S_DIRECT=1): pixels without mirrored partners skip the scratch round trip, the fold touches the border band
    // only.  Bit-identical results (197 operator / step / determinism tests pass with it on) and no measurable gain: bf16 step 55.2 vs 55.2 ms,
    // fp16 B=32 184.1 vs 185.6 (same box, back to back) -- the fold was not on the critical path.  Off by default.
    int direct_on = g_dgrad_direct.load();
    if (direct_on < 0) { const char* e = getenv("ACLGAN_DGRAD16S_DIRECT"); direct_on = e ? (atoi(e) ? 1 : 0) : 0; g_dgrad_direct.store(direct_on); }
    p.dx = dx; p.Hi = g.Hi; p.Wi = g.Wi; p.pad = g.p; p.accumulate = accumulate;
    p.direct = (direct_on && dxst == dtype && g.Ci % 2 == 0) ? 1 : 0;
    int rc;
    if (dtype == ACLGAN_DTYPE_BF16) rc = launch_dgrad16s<QBF16>(g, p, st);
    else if (dtype == ACLGAN_DTYPE_FP16) rc = launch_dgrad16s<QFP16>(g, p, st);
    else { set_error("conv_dgrad16s: dtype %d", dtype); return ACLGAN_EINVAL; }
    if (rc) return rc;
    FoldSP f;
    f.dxp = scratch; f.dx = dx; f.B = g.B; f.Hi = g.Hi; f.Wi = g.Wi; f.Ci = g.Ci; f.Hp = g.Hp; f.Wp = g.Wp; f.p = g.p;
    f.accumulate = accumulate; f.pst = dtype; f.dst = dxst; f.band_only = p.direct;
    if (p.direct && g.p == 0) return ACLGAN_OK;            // no padding: no pixel has a mirrored partner, the epilogue wrote everything
    f.total = (int64_t)g.B * g.Hi * g.Wi * (g.Ci / 4);
    hipLaunchKernelGGL(conv_fold_st_kernel, dim3(cdiv(g.Wi * (g.Ci / 4), 256), g.Hi, g.B), dim3(256), 0, st, f);
    ACL_CHECK_LAUNCH("conv_fold_st_kernel");
    return ACLGAN_OK;
}


// weight gradient on 16-bit x AND dy (Cout % 128 == 0, Cin % 128 == 0, no upsample): dw += ..., db += ... (db may be null)
bool conv_wgrad16s_ok(const ConvGeom& g) { return wgrad16s_shape_ok(g); }
size_t conv_wgrad16s_scratch_bytes(const ConvGeom& g) {
    if (!wgrad16s_shape_ok(g)) return 0;
    const WgSPlan q = wgrad16s_plan(g);
    return q.part_bytes + q.partb_bytes;
}
int conv_wgrad16s(const ConvGeom& g, int dtype, const void* x16, const void* dy16, float* dw, float* db, void* scratch, hipStream_t st) {
    if (!wgrad16s_shape_ok(g) || !dw) return ACLGAN_EUNSUPPORTED;
    ACL_REQUIRE(x16 && dy16 && scratch, "conv_wgrad16s: null operand / scratch");
    ACL_REQUIRE((long long)g.B * g.Hi * g.Wi * g.Ci * 2 < 0x7fffffe0ll && (long long)g.M * g.Co * 2 < 0x7fffffe0ll, "conv_wgrad16s: operand beyond 2 GB");
    const WgSPlan q = wgrad16s_plan(g);
    WgSP p;
    p.x16 = (const u16*)x16; p.dy16 = (const u16*)dy16; p.part = (float*)scratch; p.part_b = (float*)((char*)scratch + q.part_bytes); p.dw = dw; p.db = db;
    p.B = g.B; p.Hi = g.Hi; p.Wi = g.Wi; p.Ci = g.Ci; p.Ho = g.Ho; p.Wo = g.Wo; p.Co = g.Co; p.k = g.k; p.s = g.s; p.p = g.p; p.P = g.M; p.Kn = g.K;
    p.chunk = q.chunk; p.tiles_n = q.tiles_n; p.nwg = q.nwg; p.splits = q.splits; p.nsub = q.nsub; p.sub = q.sub;
    if (dtype == ACLGAN_DTYPE_BF16) hipLaunchKernelGGL(conv_wgrad16s_kernel<QBF16>, dim3(q.nwg, 1, q.splits), dim3(256), 0, st, p);
    else if (dtype == ACLGAN_DTYPE_FP16) hipLaunchKernelGGL(conv_wgrad16s_kernel<QFP16>, dim3(q.nwg, 1, q.splits), dim3(256), 0, st, p);
    else { set_error("conv_wgrad16s: dtype %d", dtype); return ACLGAN_EINVAL; }
    ACL_CHECK_LAUNCH("conv_wgrad16s_kernel");
    if (db) {
        hipLaunchKernelGGL(colsum16_kernel, dim3(g.Co / 64, q.nsub), dim3(256), 0, st, p, dtype);
        ACL_CHECK_LAUNCH("colsum16_kernel");
    }
    hipLaunchKernelGGL(wgrad16s_finish_kernel, dim3((int)std::max<int64_t>(cdiv(g.Co, 16), std::min<int64_t>(cdiv64((int64_t)g.Co * (g.K / 4), 256), 4096))), dim3(256), 0, st, p);
    ACL_CHECK_LAUNCH("wgrad16s_finish_kernel");
    return ACLGAN_OK;
}

}  // namespace aclgan
