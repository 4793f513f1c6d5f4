// common.h -- shared declarations of libaclgan_hip (gfx950 only; no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <atomic>
#include "../../include/aclgan_hip.h"

namespace aclgan {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

// kernel launches issued by this library since load (measurement support: aclgan_launch_count; bench.py reports launches per step)
extern std::atomic<long long> g_launches;

#define ACL_CHECK_LAUNCH(what)                                           \
    do {                                                                 \
        ++::aclgan::g_launches;                                          \
        hipError_t e__ = hipGetLastError();                              \
        if (e__ != hipSuccess) return ::aclgan::hip_fail(e__, what);     \
    } while (0)

#define ACL_REQUIRE(cond, ...)                                           \
    do {                                                                 \
        if (!(cond)) { ::aclgan::set_error(__VA_ARGS__); return ACLGAN_EINVAL; } \
    } while (0)

// Deterministic mode (aclgan_set_deterministic / ACLGAN_DETERMINISTIC=1): every reduction that the default plan combines with fp32
// atomics (reflection halo and small-grid split-K of dgrad, the thin / odd-channel weight gradients, bias column sums, LayerNorm
// parameter gradients, the L1 loss value) takes an ordered path instead -- gradients and losses are reproducible bit for bit run to
// run, at a few % of step time.  Scratch sizes depend on the mode: set it before sizing workspaces.
bool deterministic();
void set_deterministic(int on);
// out[i] += part[0][i] + part[1][i] + ... (slices added in index order); part is [nslices][n]
int reduce_slices_ordered(const float* part, int64_t n, int nslices, float* out, hipStream_t st);
// db[c] += sum over the M rows of dy[M][C], reproducible: row chunks -> part [chunks][C] -> ordered.  part: colsum_ordered_bytes(M, C)
size_t colsum_ordered_bytes(int64_t M, int C);
int colsum_ordered(const float* dy, float* db, int64_t M, int C, void* part, hipStream_t st);

// Scheduler switches (aclgan_tuning / environment), read once per update:
//   lanes (ACLGAN_LANES, default 3): HIP streams the independent branches of an update are spread over (engine.hip); 1 = one queue
//   u_batch (ACLGAN_U_BATCH, default 1): batched Winograd filter transforms at the start of an update
int lanes_setting();
int set_lanes(int v);            // returns the previous value
int u_batch_setting();
int set_u_batch(int v);
//   norm_mask (ACLGAN_NORM_MASK, default 1): the norm backward recomputes ReLU masks from x and the forward's coefficients instead of reading y
int norm_mask_setting();
int set_norm_mask(int v);
//   mlp_fused (ACLGAN_MLP_FUSED, default 1): the generator's MLP forward as one launch (misc.hip: mlp3_fwd) instead of three linear_fwd launches
int mlp_fused_setting();
int set_mlp_fused(int v);
//   fault_at (test hook, default -1 = off): the backward replay fails with ACLGAN_EHIP after its fault_at-th closure has been enqueued --
//   the error path (lanes and side stream drained before the caller is told) is testable without breaking the GPU
int fault_at_setting();
int set_fault_at(int v);
// every aclgan_tuning call bumps this: cached results that depend on a switch (workspace checks) are keyed by it
long long tuning_epoch();

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// geometry derived from a conv descriptor
struct ConvGeom {
    int B, Hi, Wi, Ci, Co, k, s, p, up, act;
    int Hu, Wu;   // after optional upsample
    int Hp, Wp;   // after reflect pad
    int Ho, Wo;
    int M;        // B*Ho*Wo
    int K;        // k*k*Ci
};
int make_geom(const aclgan_conv_desc* d, ConvGeom* g);

// ---- kernel launchers (all async on `st`) ----
int conv_fold(const ConvGeom& g, const float* dxp, float* dx, int accumulate, hipStream_t st);
// matrix-pipe FLOPs the chosen path of this convolution executes (conv_fast.hip; which: 0 forward, 1 input gradient, 2 weight gradient)
double conv_exec_flops(const ConvGeom& g, int which, bool f16);
int conv_fold_band(const ConvGeom& g, const float* dxp, float* dx, int band, hipStream_t st);      // dx += the band (width `band`) of a padded-grid gradient
bool conv_wgrad_fast_supported(const ConvGeom& g);
// scratch (optional, conv_fwd_scratch_bytes): enables the sub-pixel path of the upsample+5x5 decoder convs
int conv_fwd(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, hipStream_t st, void* scratch = nullptr, float* stats = nullptr,
             float* keepV = nullptr);
// > 0: conv_fwd (with scratch) runs this layer through Winograd and can leave the input transform V in a caller-owned buffer of this
// many bytes (keepV); conv_wgrad(.., haveV) of the same layer and input then skips its own input transform.  0: not offered.
size_t conv_fwd_keep_bytes(const ConvGeom& g);
// > 0: conv_fwd (with scratch) can emit the normalisation statistics of its output from its epilogue: stats[B][Ho*Wo / chunk][Co] =
// (mean, M2) of groups of `chunk` output pixels (the return value), the chunk partials norm_fwd combines;  0: not for this shape
int conv_fwd_stats_chunk(const ConvGeom& g);
size_t conv_fwd_scratch_bytes(const ConvGeom& g);
size_t conv_up5_scratch_bytes(const ConvGeom& g);
int conv_up5_fwd(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, void* scratch, hipStream_t st, float* keepV = nullptr);
int conv_fwd_naive(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, hipStream_t st);
size_t conv_dgrad_scratch_bytes(const ConvGeom& g);
int conv_dgrad(const ConvGeom& g, const float* dy, const float* w, float* dx, void* scratch, int accumulate, hipStream_t st);
int conv_wgrad(const ConvGeom& g, const float* x, const float* dy, float* dw, float* db, hipStream_t st, void* scratch = nullptr, const float* haveV = nullptr);
size_t conv_wgrad_scratch_bytes(const ConvGeom& g);
int conv_up5_dgrad(const ConvGeom& g, const float* dy, const float* w, float* dx, int accumulate, void* scratch, hipStream_t st);
int conv_up5_wgrad(const ConvGeom& g, const float* x, const float* dy, float* dw, float* db, void* scratch, hipStream_t st, const float* haveV = nullptr);

// tuned kernels (conv_fast.hip); return ACLGAN_EUNSUPPORTED when the shape is not eligible
int conv_fwd_fast(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, hipStream_t st, void* scratch = nullptr, float* stats = nullptr,
                  float* keepV = nullptr);
size_t conv_fwd_fast_scratch_bytes(const ConvGeom& g);
int conv_dgrad_fast(const ConvGeom& g, const float* dy, const float* w, float* dxp, float* dx, int accumulate, bool* direct, hipStream_t st);
// also accumulates the bias gradient into db when db != nullptr
int conv_wgrad_fast(const ConvGeom& g, const float* x, const float* dy, float* dw, float* db, hipStream_t st, void* scratch = nullptr, const float* haveV = nullptr);
size_t conv_wgrad_fast_scratch_bytes(const ConvGeom& g);

// Winograd filter-transform cache (round 3): a generator's ResBlock filters are transformed 2-3 times per update (the decoder runs up
// to three times, forward and input-gradient each need U) -- the step scheduler offers a per-update cache instead.  lookup(w, variant,
// bytes, &fresh): the buffer that holds / shall hold the transform of filter tensor w (variant 0 forward, 1 flipped for the input
// gradient, 2 / 3 the same for the merged sub-pixel phase filters); fresh = it must be computed by the caller now.  nullptr: not cached.
struct WinoUCache { void* user; float* (*lookup)(void* user, const float* w, int variant, size_t bytes, bool* fresh); };
void set_wino_ucache(const WinoUCache* c);      // thread-local; nullptr = off (operator-level calls)
const WinoUCache* wino_ucache();
size_t conv_wino_u_bytes(const ConvGeom& g);    // bytes of one cached transform of this layer's filter (0: not a Winograd layer)
// Round 5.  `variant` of a lookup = (0 forward | 1 input gradient | 2 / 3 the same for merged sub-pixel phase filters) | layout << 4: an
// entry is keyed by (filter, variant & 15) and remembers the layout it was filled in (pipeline fp32 / MFMA-fragment order / bf16 planes);
// a lookup asking for another layout gets nullptr.  conv_wino_u_variant / conv_up5_wino_u_variant: the lookup a layer's forward
// (dgrad 0; keepV: the caller keeps the input transform) / input gradient will make (-1: none); conv_wino_prefill: the transforms of
// `count` equally shaped 3x3 filters w_stride floats apart into U0 + i * u_stride floats, in that layout, as ONE launch.
int conv_wino_u_variant(const ConvGeom& g, int dgrad, bool keepV);
int conv_up5_wino_u_variant(const ConvGeom& g, int dgrad, bool keepV);
int conv_wino_prefill(const ConvGeom& g, int uv, const float* w0, int64_t w_stride, float* U0, int64_t u_stride, int count, hipStream_t st);
int wino_fused_filter_batch(const float* w0, int64_t w_stride, float* Uf0, int64_t u_stride, int count, int Co, int Ci, int flip, hipStream_t st);

// Winograd F(4x4,3x3) path of the 3x3 stride-1 reflect-pad-1 layers (conv_wino.hip); EUNSUPPORTED when not eligible / no scratch
bool conv_wino_ok(const ConvGeom& g);
size_t conv_wino_scratch_bytes(const ConvGeom& g);
int conv_fwd_wino(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, void* scratch, hipStream_t st, float* stats = nullptr,
                  float* keepV = nullptr);
size_t conv_wino_keep_bytes(const ConvGeom& g);
size_t conv_up5_wino_keep_bytes(const ConvGeom& g);
int conv_dgrad_wino_interior(const ConvGeom& g, const float* dy, const float* w, float* dx, int accumulate, void* scratch, hipStream_t st);
size_t conv_wgrad_wino_scratch_bytes(const ConvGeom& g);
// the four VALID-3x3 phases of the sub-pixel upsample+5x5 path through Winograd (wp: merged fp32 phase filters [4][Co][3][3][Ci])
bool conv_up5_wino_ok(const ConvGeom& g);
size_t conv_up5_wino_fwd_scratch_bytes(const ConvGeom& g);
size_t conv_up5_wino_dgrad_scratch_bytes(const ConvGeom& g);
size_t conv_up5_wino_wgrad_scratch_bytes(const ConvGeom& g);
// wkey (optional): the ORIGINAL 5x5 filter tensor the merged phase filters wp were made from -- the key of the filter-transform cache
int conv_up5_wino_fwd_phases(const ConvGeom& g, const float* x, const float* wp, const float* bias, float* y, void* scratch, hipStream_t st, float* keepV = nullptr,
                             const float* wkey = nullptr);
int conv_up5_wino_dgrad_phases(const ConvGeom& g, const float* dy, const float* wp, float* dx, int accumulate, void* scratch, hipStream_t st,
                               const float* wkey = nullptr);
// true: the cache already holds that transform (the caller may skip preparing wp)
bool wino_u_cached(const float* wkey, int variant);
int conv_up5_wino_wgrad_phases(const ConvGeom& g, const float* x, const float* dy, float* dwp, float* db, void* scratch, hipStream_t st, const float* haveV = nullptr);
size_t conv_up5_dgrad_scratch_bytes(const ConvGeom& g);
int conv_wgrad_wino(const ConvGeom& g, const float* x, const float* dy, float* dw, float* db, void* scratch, hipStream_t st, const float* haveV = nullptr);

// 16-bit MFMA kernels (conv_fast16.hip): operands rounded to bf16 / fp16, fp32 accumulation and outputs.
// which: 0 forward, 1 dgrad, 2 wgrad.  EUNSUPPORTED when the shape is not eligible.
bool conv16_eligible(const ConvGeom& g, int which);
size_t conv_fwd16_scratch_bytes(const ConvGeom& g);
size_t conv_dgrad16_scratch_bytes(const ConvGeom& g);
size_t conv_wgrad16_scratch_bytes(const ConvGeom& g);
// w: fp32 OHWI master weights (only read by the sub-pixel path, to merge the phase filters before rounding); w16: OHWI 16-bit pack
// x16 (optional): the producer's 16-bit copy of x (same layout) -- read instead of x, same result
// y_storage != 0: y is stored in the 16-bit dtype (and points at 16-bit data)
int conv_fwd16(const ConvGeom& g, int dtype, const float* x, const float* w, const void* w16, const float* bias, float* y, void* scratch, hipStream_t st,
               const void* x16 = nullptr, int y_storage = 0);
// w16t: 16-bit pack transposed to [tap][cin][cout]; dx is complete on return (interior + mirrored halo)
int conv_dgrad16(const ConvGeom& g, int dtype, const float* dy, const float* w, const void* w16t, float* dx, int accumulate, void* scratch, hipStream_t st);
// x_storage / dy_storage != 0: that operand is stored in the 16-bit dtype (the pointer then addresses 16-bit data)
int conv_wgrad16(const ConvGeom& g, int dtype, const float* x, const float* dy, float* dw, float* db, void* scratch, hipStream_t st, int x_storage = 0,
                 int dy_storage = 0);

// 16-bit ACTIVATION STORAGE kernels (conv_glds16.hip): operands already live in HBM in the 16-bit dtype, tiles go global -> LDS directly.
// which: 0 forward, 1 dgrad.  No upsample, Cin % 64 == 0, Cout % 64 == 0.
bool conv16s_ok(const ConvGeom& g, int which);
// stats (optional, conv_fwd16s_stats_chunk(g) > 0): [B][Ho*Wo / chunk][Co] (mean, M2) pairs of the STORED outputs -- norm_fwd's chunk partials
int conv_fwd16s_stats_chunk(const ConvGeom& g);
int set_glds_tile(int v);
int set_fwd16_patch(int v);
int set_dgrad16s_direct(int v);
int set_wino_x3(int v);
// conv_wino_fused.hip (round 4): Winograd F(4x4,3x3) as ONE launch -- input transform, 36 frequency GEMMs, output transform
int wino_fused_mode();                           // 0 off, 1 fused where the cost model says it pays, 2 fused wherever eligible
int set_wino_fused(int v);                       // returns the previous mode
int wino_fused_force(int on);                    // per-thread: > 0 = eligible shapes take the fused kernel whatever the mode; returns the previous value
bool wino_fused_ok(int B, int H, int W, int Cin_, int Cout_, int act = ACLGAN_ACT_NONE, int gph = 1, int kph = 1);
int wino_fused_filter(const float* w, float* Uf, int Co, int Ci, int flip, hipStream_t st, int nph = 1);
int wino_fused_launch(int B, int H, int W, int Cin_, int Cout_, const float* in, const float* Uf, const float* bias, float* out, int act, int accumulate,
                      int reflect, float2* stats, hipStream_t st);
int wino_fused_up5_fwd(int B, int Hi, int Wi, int Cin_, int Cout_, const float* x, const float* Uf, const float* bias, float* y, int Hf, int Wf, int act,
                       hipStream_t st);
// conv_wino_wgrad_fused.hip (round 4): the Winograd weight / bias gradient of the 3x3 ResBlock layers as one kernel + one finish launch
int wino_wgrad_fused_mode();                     // 0 off, 1 where the cost model says it pays, 2 wherever eligible
int set_wino_wgrad_fused(int v);                 // returns the previous mode
bool wino_wgrad_fused_ok(const ConvGeom& g);
size_t wino_wgrad_fused_scratch_bytes(const ConvGeom& g);
int wino_wgrad_fused(const ConvGeom& g, const float* x, const float* dy, float* dw, float* db, void* scratch, hipStream_t st);
int wino_fused_up5_dgrad(int B, int Hi, int Wi, int Cin_, int Cout_, const float* dy, const float* Uf, float* dx, int Hf, int Wf, int accumulate, hipStream_t st);
// Round 6: the 4x4 stride-2 reflect-pad-1 layers as four parity phases through the same kernel (conv_wino_fused.hip, "s2k4")
bool wino_fused_s2k4_ok(int B, int Hi, int Wi, int Ci, int Co, int act, int dgrad);
size_t wino_fused_s2k4_u_bytes(int Ci, int Co);
int wino_fused_filter_s2k4(const float* w, float* Uf, int Co, int Ci, int dgrad, hipStream_t st);
int wino_fused_s2k4_fwd(int B, int Hi, int Wi, int Ci, int Co, const float* x, const float* Uf, const float* bias, float* y, int act, float2* stats, hipStream_t st);
int wino_fused_s2k4_dgrad(int B, int Hi, int Wi, int Ci, int Co, const float* dy, const float* Uf, float* dx, int accumulate, hipStream_t st);
// ... and their conv-level wrappers (conv_wino.hip): which = 0 forward, 1 input gradient (interior of the padded grid; never in deterministic mode)
bool conv_s2k4_wino_ok(const ConvGeom& g, int which);
int wino_s2k4_setting();                        // tuning switch "wino_s2k4": 1 = on (default), 0 = the stride-2 layers keep the direct kernels
int set_wino_s2k4(int v);
size_t conv_s2k4_wino_scratch_bytes(const ConvGeom& g);
int conv_fwd_s2k4_wino(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, void* scratch, hipStream_t st, float* stats = nullptr);
int conv_dgrad_s2k4_wino_interior(const ConvGeom& g, const float* dy, const float* w, float* dx, int accumulate, void* scratch, hipStream_t st);
// gemm_bf16x3.hip: fp32-accurate GEMM slices on the bf16 matrix cores from 3-plane (h, m, l) bf16 operands
bool gemm_x3_shape_ok(int T, int K, int N);
int gemm_slices_x3(const void* A3, size_t a_plane, const void* B3, size_t b_plane, float* C, int T, int K, int N, int nslices, int a_mod, hipStream_t st);
int split3_planes(const float* x, void* out, int64_t n, int64_t plane_elems, hipStream_t st);
int conv_fwd16s(const ConvGeom& g, int dtype, const void* x16, const void* w16, const float* bias, void* y, int yst, hipStream_t st, float* stats = nullptr);
size_t conv_dgrad16s_scratch_bytes(const ConvGeom& g);
// weight gradient with BOTH operands in the 16-bit dtype (Cin, Cout multiples of 128): pixel-major LDS-DMA tiles + transposing LDS reads
bool conv_wgrad16s_ok(const ConvGeom& g);
size_t conv_wgrad16s_scratch_bytes(const ConvGeom& g);
int conv_wgrad16s(const ConvGeom& g, int dtype, const void* x16, const void* dy16, float* dw, float* db, void* scratch, hipStream_t st);
int conv_dgrad16s(const ConvGeom& g, int dtype, const void* dy16, const void* w16t, void* dx, int dxst, int accumulate, void* scratch, hipStream_t st);
int cast_flat16(const float* src, void* dst, int64_t n, int dtype, hipStream_t st);
int transpose_flat16(const float* base, void* base_t, const int64_t* offs, const int* co, const int* taps, const int* ci, int n, int dtype, hipStream_t st);

// direct VALU kernels for the 64->4 channel 7x7 output conv (conv_small.hip); EUNSUPPORTED otherwise
int conv_fwd_small(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y, hipStream_t st);
int conv_wgrad_small(const ConvGeom& g, const float* x, const float* dy, float* dw, float* db, hipStream_t st, void* scratch = nullptr);
size_t conv_wgrad_small_scratch_bytes(const ConvGeom& g);
int conv_dgrad_small(const ConvGeom& g, const float* dy, const float* w, float* dxp, hipStream_t st);

size_t norm_scratch_bytes(int B, int HW, int C);
// storage codes (st16.h: 0 fp32, ACLGAN_DTYPE_BF16, ACLGAN_DTYPE_FP16) of the tensors a normalisation call touches; statistics,
// coefficients and parameter gradients are always fp32
struct NormST { int x = 0, y = 0, res = 0, dy = 0, dx = 0, dres = 0; };
int norm_fwd(int kind, int act, int B, int HW, int C, const void* x, const float* w, const float* b, int w_stride,
             const void* residual, void* y, float* mean, float* rstd, void* scratch, hipStream_t st, const float* stats = nullptr,
             int stats_chunk = 0, const NormST* sto = nullptr, float* ss_out = nullptr);
// ss_out (norm_fwd, optional, 2 B C floats) / ss (norm_bwd): the fused coefficients (scale | shift) of y = act(x * scale + shift).  Kept by
// the caller from the forward to the backward, norm_bwd recovers a ReLU / LeakyReLU mask as the sign of the same fmaf and does not read y.
// sbc_out (optional, LayerNorm only, [B][C][2] floats): receives the per-sample totals (sum g, sum g*xhat) per channel INSTEAD of the
// gamma / beta gradients being added here -- the caller adds them later with norm_bwd_ln_params (on its parameter-gradient stream)
int norm_bwd(int kind, int act, int B, int HW, int C, const void* x, const void* y, const void* dy,
             const float* w, int w_stride, const float* mean, const float* rstd, void* dx, float* dw, float* db,
             void* dres, int dres_accumulate, void* scratch, hipStream_t st, const NormST* sto = nullptr, float* sbc_out = nullptr,
             const float* ss = nullptr);
// dgamma[c] += sum_b sbc[b][c][1], dbeta[c] += sum_b sbc[b][c][0], samples added in index order
int norm_bwd_ln_params(const float* sbc, int B, int C, float* dgamma, float* dbeta, hipStream_t st);

// dy *= act'(y) (y, dy may be stored in different dtypes; n % 4 == 0 unless both are fp32)
int act_bwd_inplace(int act, const void* y, void* dy, int64_t n, hipStream_t st, int yst = 0, int gst = 0);
// dst (storage dst_st) = src (storage src_st), elementwise conversion; n % 4 == 0
int cast_storage(const void* src, int src_st, void* dst, int dst_st, int64_t n, hipStream_t st);
int positive_mask(const void* y, int yst, unsigned char* dst, int64_t n, hipStream_t st);
int positive_mask_diff(const float* a, int ca, const float* b, int cb, unsigned char* dst, int64_t npix, hipStream_t st);      // dst[i] = y[i] > 0 (diagnostics)
int avgpool3s2_fwd(int B, int H, int W, int C, const float* x, float* y, hipStream_t st);
int avgpool3s2_bwd(int B, int H, int W, int C, const float* dy, float* dx, int accumulate, hipStream_t st);
int adam_flat(float* p, const float* g, float* m, float* v, int64_t n, const aclgan_adam* o, int step, hipStream_t st);
int nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, hipStream_t st);
int nhwc_to_nchw(const float* src, float* dst, int B, int C, int H, int W, hipStream_t st);
int fill_zero(float* p, int64_t n, hipStream_t st);

// small dense layers (MLP, style head): y[b][o] = act(sum_i x[b][i] W[o][i] + bias[o])
int linear_fwd(int B, int I, int O, const float* x, const float* w, const float* bias, int act, float* y, hipStream_t st);
// the generator's three-layer MLP forward in one launch (misc.hip: bit-identical to three linear_fwd calls); EUNSUPPORTED outside S <= 64, M in {64, 128, 192, 256}
bool mlp3_fwd_ok(int S, int M);
int mlp3_fwd(int B, int S, int M, int O, const float* s, const float* W0, const float* b0, const float* W1, const float* b1, const float* W2,
             const float* b2, float* m0, float* m1, float* ap, hipStream_t st);
// dy is modified in place by the activation backward; dx overwritten (may be null); dw,db accumulate
int linear_bwd(int B, int I, int O, const float* x, const float* y, float* dy, const float* w, int act,
               float* dx, float* dw, float* db, hipStream_t st);
// the parameter half alone: dw[o][i] += sum_b dy[b][o] x[b][i], db[o] += sum_b dy[b][o]; dy = the gradient AFTER the activation backward
// (linear_bwd with dw = db = nullptr has applied it)
int linear_bwd_params(int B, int I, int O, const float* x, const float* dy, float* dw, float* db, hipStream_t st);
// global average pool NHWC [B][HW][C] (storage xst) -> [B][C] fp32; backward writes dx in storage xst
int gap_fwd(int B, int HW, int C, const void* x, float* y, hipStream_t st, int xst = 0);
int gap_bwd(int B, int HW, int C, const float* dy, void* dx, int accumulate, hipStream_t st, int xst = 0);

// trainer-level fused kernels
// focus_translation (trainer.py:85-88): dec4 [B][HW][4] (ch0-2 fg, ch3 focus), bg [B][HW][3] -> out [B][HW][3];
// pair (optional) [B][HW][6] = (pair_first, out)
int focus_blend_fwd(int B, int HW, const float* dec4, const float* bg, float* out, const float* pair_first, float* pair, hipStream_t st);
// d_out [B][HW][3] (+ optional d_pair [B][HW][6], channels 3..5 add to d_out) -> d_dec4 (+=: zero-initialised by the
// caller), d_bg (+= if accumulate else =; may be null)
int focus_blend_bwd(int B, int HW, const float* dec4, const float* bg, const float* d_out, const float* d_pair,
                    float* d_dec4, float* d_bg, int bg_accumulate, hipStream_t st);
// non-focus configuration (gen.output_dim 3, trainer.py:117-121,129-133): out = dec3, pair (optional) = (pair_first, dec3); backward adds d_out and
// d_pair[.., 3:6] onto d_dec3
int plain_pair_fwd(int B, int HW, const float* dec3, float* out, const float* pair_first, float* pair, hipStream_t st);
int plain_pair_bwd(int B, int HW, const float* d_out, const float* d_pair, float* d_dec3, hipStream_t st);
// focus_translation on the reference's own NCHW tensors (sample() / test.py: trainer.py:85-88, test.py:73-76):
// out[b][c][p] = fg[b][c][p]*m + bg[b][c][p]*(1-m), m = (focus[b][0][p]+1)/2, c < 3; *_bstride = floats between samples
int focus_translation_nchw(const float* fg, int64_t fg_bstride, const float* bg, int64_t bg_bstride, const float* focus, int64_t focus_bstride,
                           float* out, int B, int HW, hipStream_t st);
// LSGAN (networks.py:67,83,98): loss_slot += weight*mean((o-t)^2); d_o = weight*2(o-t)/n*gscale (if d_o != null)
// lscale (optional, device): fp16 dynamic loss scale; the gradient seed is multiplied by lscale[0], the reported loss is not
int lsgan_loss(const float* o, int n, float target, float weight, float* loss_slot, float* d_o, float gscale, hipStream_t st, const float* lscale = nullptr);
// the same for up to LSGAN_MAX_TERMS (map, target) terms in ONE launch: the terms are processed in index order by one workgroup, so
// every slot receives its additions in the order -- and with the values -- of the equivalent sequence of lsgan_loss calls
struct LsganTerm { const float* o; float* d_o; float* slot; int n; float target, weight, gscale; };
const int LSGAN_MAX_TERMS = 12;
int lsgan_loss_batch(const LsganTerm* terms, int nterms, hipStream_t st, const float* lscale = nullptr);
// L1 (trainer.py:61-62): loss_slot = mean|a[..,:3] - b|; a has a_stride channels (4: decoder output), b 3 channels.
const int L1_PART_FLOATS = 1024;
int l1_loss(const float* a, int a_stride, const float* b, int64_t npix, float* loss_slot, float* d_a, float gscale, int d_accumulate, hipStream_t st,
            const float* lscale = nullptr, float* part = nullptr);
// focus losses (trainer.py:146-158).  focus_sums: per-workgroup partials part[2*blk] = sum(m - upper), part[2*blk+1] =
// sum 1/(|m-.5|+eps) over the mask channel (ch3 of dec4); part holds 2*focus_sums_blocks(npix) floats
int focus_sums_blocks(int64_t npix);
int focus_sums(const float* dec4, int64_t npix, float eps, float upper, float* part, hipStream_t st);
// adds the partials in order, writes size/digit into the loss slots and adds the focus gradient into d_dec4 channel 3
// totals (optional, device float[2]): sum(m - upper) and the digit sum over a larger pixel set (the GLOBAL batch of a
// data-parallel job, npix_total pixels) to be used instead of this call's own partials
int focus_loss_finish(const float* dec4, int64_t npix, const float* part, float delta, float upper, float lower, float eps,
                      float scale, float* size_slot, float* digit_slot, float* d_dec4, hipStream_t st, const float* lscale = nullptr,
                      const float* totals = nullptr, int64_t npix_total = 0);
// totals[2*i], totals[2*i+1] = ordered sums of mask i's partials (mask i's partials start at part + i*2*focus_sums_blocks(npix))
int focus_totals(const float* part, int64_t npix, int nmask, float* totals, hipStream_t st);
// Adam under fp16 dynamic loss scaling: overflow scan, update with g/S (or skip), scale update -- all on the device
int adam_flat_scaled(float* p, const float* g, float* m, float* v, int64_t n, const aclgan_adam* o, int step, float* state, int group, hipStream_t st);

}  // namespace aclgan
