/*
 * aclgan_hip.h -- C ABI of libaclgan_hip.so, the MI355X-native (gfx950) implementation of the
 * ACL-GAN training step.
 *
 * The reference (hyperplane-lab/ACL-GAN) has NO plugin / operator / FFI interface of its own
 * (SURVEY.md section 8b): its hot path reaches ATen/cuDNN through torch.nn.  This header is the
 * boundary one level below the reference's Python surface; every entry point names the
 * reference code it replaces (file:line into the reference tree).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.  All `float*` are DEVICE pointers
 *     unless the name ends in `_host`.  Streams are passed as `void*` (a hipStream_t).
 *   - every function returns 0 on success or a negative ACLGAN_E* code; the message is available
 *     from aclgan_last_error() (thread local).  Nothing aborts or throws across the boundary.
 *   - ownership: the caller owns every device buffer (parameters, optimizer state, workspace);
 *     the library never allocates or frees device memory and keeps only the pointers registered
 *     through aclgan_bind_*.  All work is enqueued asynchronously on the given stream.
 *   - layouts: activations NHWC fp32 inside; images cross the boundary in the reference's NCHW.
 *     Convolution weights are OHWI ([Cout][kh][kw][Cin]) inside the flat parameter buffer; the
 *     Python side converts to/from the reference's OIHW at state_dict time.
 */
#ifndef ACLGAN_HIP_H
#define ACLGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACLGAN_OK 0
#define ACLGAN_EINVAL (-1)
#define ACLGAN_EUNSUPPORTED (-2)
#define ACLGAN_EHIP (-3)
#define ACLGAN_ENOMEM (-4)

/* activation enum (networks.py:343-357: relu / lrelu(0.2) / tanh / none) */
#define ACLGAN_ACT_NONE 0
#define ACLGAN_ACT_RELU 1
#define ACLGAN_ACT_LRELU 2
#define ACLGAN_ACT_TANH 3

/* normalisation enum (networks.py:327-341: none / in / adain / ln) */
#define ACLGAN_NORM_NONE 0
#define ACLGAN_NORM_IN 1
#define ACLGAN_NORM_ADAIN 2
#define ACLGAN_NORM_LN 3

/* compute dtype of the heavy convolutions (NOT in the reference, which is fp32-only: BASELINE.json configs[2] / [4]).
 * BF16 / FP16: both MFMA operands are rounded to the 16-bit type (round to nearest even), products accumulate in fp32
 * (v_mfma_f32_32x32x16_bf16 / _f16); activations, gradients, master weights, Adam state, norm statistics and losses
 * stay fp32.  FP16 training uses dynamic loss scaling (aclgan_bind_loss_scale). */
#define ACLGAN_DTYPE_FP32 0
#define ACLGAN_DTYPE_BF16 1
#define ACLGAN_DTYPE_FP16 2

/* parameter groups = the reference's two optimizers (trainer.py:37-42) */
#define ACLGAN_GROUP_GEN 0
#define ACLGAN_GROUP_DIS 1

/* networks (trainer.py:19-23) */
#define ACLGAN_NET_GEN_AB 0
#define ACLGAN_NET_GEN_BA 1
#define ACLGAN_NET_DIS_A 2
#define ACLGAN_NET_DIS_B 3
#define ACLGAN_NET_DIS_2 4

/* indices into the device loss array written by aclgan_gen_update / aclgan_dis_update: the 16
 * `loss_*` attributes the reference sets (trainer.py:136-165, 283-290; read by utils.py:174-178) */
enum {
    ACLGAN_L_GEN_ADV_A = 0, ACLGAN_L_GEN_ADV_B, ACLGAN_L_GEN_ADV_2,
    ACLGAN_L_GEN_FOCUS_B_SIZE, ACLGAN_L_GEN_FOCUS_B_DIGIT,
    ACLGAN_L_GEN_FOCUS_A_SIZE, ACLGAN_L_GEN_FOCUS_A_DIGIT,
    ACLGAN_L_GEN_FOCUS_A2_SIZE, ACLGAN_L_GEN_FOCUS_A2_DIGIT,
    ACLGAN_L_IDT_A, ACLGAN_L_IDT_B, ACLGAN_L_GEN_TOTAL,
    ACLGAN_L_DIS_A, ACLGAN_L_DIS_B, ACLGAN_L_DIS_2, ACLGAN_L_DIS_TOTAL,
    ACLGAN_L_COUNT
};

/* architecture: the `gen:` / `dis:` / `input_dim_*` keys of configs/male2female.yaml:39-59 that
 * change tensor shapes.  (activ relu / lrelu, pad reflect, norm none for D, lsgan: the only
 * branches the shipped config reaches -- SURVEY.md section 2 rows 5,6.) */
typedef struct aclgan_arch {
    int input_dim_a, input_dim_b;
    int gen_dim, gen_mlp_dim, gen_style_dim, gen_output_dim, gen_n_downsample, gen_n_res;
    int dis_dim, dis_n_layer, dis_num_scales;
} aclgan_arch;

/* per-call hyper-parameters read by gen_update / dis_update (trainer.py:93-97,142-144,164,288-290) */
typedef struct aclgan_hparams {
    float gan_w, gan_cw, recon_x_w;
    float focus_loss, focus_delta, focus_upper, focus_lower, focus_epsilon;
    float alpha;
} aclgan_hparams;

/* torch.optim.Adam as configured at trainer.py:39-42 */
typedef struct aclgan_adam {
    float lr, beta1, beta2, eps, weight_decay;
} aclgan_adam;

/* one convolution: reflect pad -> conv (networks.py:366), optionally preceded by the decoder's
 * nearest 2x upsample (networks.py:256), NHWC activations, OHWI weights */
typedef struct aclgan_conv_desc {
    int B, Hi, Wi, Ci;   /* input tensor (before the optional upsample) */
    int Co, k, stride, pad;
    int upsample;        /* 0 / 1 */
    int act;             /* ACLGAN_ACT_* fused into the forward epilogue */
} aclgan_conv_desc;

typedef struct aclgan_ctx aclgan_ctx;

/* ---- library ---- */
int aclgan_version(void);
/* kernels launched by the library since it was loaded (measurement support: launches per step on the bench line) */
long long aclgan_launch_count(void);
/* C[f][T][N] = A[f][T][K] x B[f][N][K]^T for f < nslices, fp32 MFMA: the batched-GEMM launch the Winograd F(4x4,3x3) pipeline of the
 * 3x3 convolutions (networks.py:297-310) spends its MFMA time in, exposed alone so that bench.py can time the step's dominant
 * kernel with HIP events and tests can check it against torch.bmm.  K % 16 == 0. */
int aclgan_gemm_slices_f32(const float* A, const float* B, float* C, int T, int K, int N, int nslices, void* stream);
/* Round 4 (csrc/conv_wino_fused.hip): ReflectionPad2d(1) + Conv2d(3x3) of the ResBlocks (networks.py:297-310, 366-370) as ONE launch -- Winograd
 * F(4x4,3x3) with the input transform, the 36 frequency GEMMs and the output transform (+ bias, activation, per-tile normalisation partials) fused:
 * the step's dominant kernel, exposed alone so that bench.py can time it with HIP events and tests can check it against the oracle.
 * aclgan_winograd_filter_frag: Uf (36 * Co * Ci floats) <- G g G^T of the OHWI filter w in MFMA-fragment order (flip != 0: the flipped, transposed
 * filter of the input gradient; then Uf is indexed [cin rows][cout k]).  The step computes it once per update and filter.
 * aclgan_conv3x3_winograd_fused: y[B][H][W][Cout] (+)= act(conv(x[B][H][W][Cin]) + bias); reflect != 0: reflection padding 1, else zero padding;
 * stats (optional): [B][H/4 * W/4][Cout] (mean, M2) pairs of the 4x4 output tiles.  H, W % 4 == 0, Cin % 16 == 0, Cout % 64 == 0, act != tanh. */
int aclgan_winograd_filter_frag(const float* w, float* Uf, int Co, int Ci, int flip, void* stream);
int aclgan_conv3x3_winograd_fused(const float* x, const float* Uf, const float* bias, float* y, int B, int H, int W, int Cin, int Cout, int act, int reflect,
                                  int accumulate, float* stats, void* stream);
/* The same product with fp32 accuracy on the bf16 matrix cores (round 3, csrc/gemm_bf16x3.hip): each fp32 operand is split EXACTLY into
 * three bf16 numbers (h + m + l), six of the nine cross products (everything above 2^-24 of the product) are accumulated in fp32 by
 * v_mfma_f32_32x32x16_bf16.  This entry point splits A and B into `scratch` (aclgan_gemm_slices_x3_scratch_bytes) and runs the kernel; the
 * step's Winograd transforms write the split planes directly.  A == B == NULL: reuse the planes an earlier call left in scratch (timing
 * the GEMM launch alone).  K % 32 == 0, N % 64 == 0, else ACLGAN_EUNSUPPORTED. */
size_t aclgan_gemm_slices_x3_scratch_bytes(int T, int K, int N, int nslices);
int aclgan_gemm_slices_x3(const float* A, const float* B, float* C, int T, int K, int N, int nslices, void* scratch, void* stream);
const char* aclgan_last_error(void);
/* Deterministic mode (process-wide; also ACLGAN_DETERMINISTIC=1; the counterpart of torch.use_deterministic_algorithms, which the
 * reference never turns on -- its cuDNN backward is not reproducible either, train.py:29 sets cudnn.benchmark = True).
 * Default (0): the forward, the losses and the weight gradients of the heavy layers are reproducible bit for bit; the reflection
 * halo / small-grid split-K of the input gradients and the weight gradients of the thin and odd-channel layers combine partial
 * sums with fp32 atomics (reproducible to ~1e-7).  on != 0: those take ordered paths (padded-grid gradient + fold gather, one
 * private copy of dw per pixel slice added in order, ordered column sums): a step is reproducible bit for bit run to run, at a
 * measured cost of a few %.  Scratch sizes depend on the mode: set it BEFORE any *_scratch_bytes /
 * *_workspace_bytes query, and pass the scratch buffers (the scratch-less operator calls refuse to run where they would need
 * atomics). */
int aclgan_set_deterministic(int on);
int aclgan_get_deterministic(void);

/* ---- context: replaces aclgan_Trainer.__init__ network construction (trainer.py:15-23) ---- */
int aclgan_ctx_create(const aclgan_arch* arch, aclgan_ctx** out);
void aclgan_ctx_destroy(aclgan_ctx* ctx);
/* Call once, OUTSIDE any stream capture, on a context whose updates will be captured into a HIP graph (torch.cuda.graph around
 * aclgan_gen_update / aclgan_dis_update): creates the context-private parameter-gradient stream a captured update runs its weight gradients
 * on (a captured update uses one lane; the process-wide lane streams are not captured).  Contexts that never capture should not call it:
 * every additional stream shifts HIP's stream -> hardware-queue placement (measured: 3 ms on the eager fp32 step). */
int aclgan_ctx_enable_capture(aclgan_ctx* ctx);
/* Round 6: create the lane scheduler's process-wide streams (the parameter-gradient stream and lanes 1 .. lanes - 1; lanes <= 0: the current
 * "lanes" setting) on the CURRENT device now rather than at the first update.  HIP binds streams to its hardware queues in creation order;
 * a host that creates other streams later (a process group's communicator, a prefetching loader) should call this first so that the lanes
 * keep a queue each (measured: a foreign stream created before them costs the 3-lane step 3.7 ms).  Without a device: no effect. */
int aclgan_warm_streams(int lanes);

/* Diagnostics for the parity tests (round 6; tests/test_gpu_maskfrozen.py): record the ReLU / LeakyReLU masks the following updates run with.
 * Every Conv2dBlock (networks.py:365-371) an update back-propagates through appends (output > 0) of its activated output -- exactly the
 * mask its backward applies -- as one byte per element, NHWC, to `dst` (device memory, cap_bytes; NULL turns the recording off) in the
 * order the forward builds the blocks; aclgan_debug_mask_count / _info enumerate them (dims4 = B, H, W, C; offset into dst; act = the
 * ACLGAN_ACT_* code).  The oracle's autograd (oracle/aclgan_oracle.py) replays an update with these masks in place of its own, which
 * separates the error of the backward KERNELS from the "mask lottery" (a pre-activation within rounding of zero takes a different branch
 * in two correct fp32 implementations and moves every upstream gradient).  The dense layers of the MLP are not recorded.  Each call
 * resets the list; an update whose masks do not fit returns ACLGAN_ENOMEM.  Costs one small launch per block while on. */
int aclgan_debug_capture_masks(aclgan_ctx* ctx, unsigned char* dst, size_t cap_bytes);
int aclgan_debug_mask_count(const aclgan_ctx* ctx);
int aclgan_debug_mask_info(const aclgan_ctx* ctx, int index, int* dims4, long long* offset, int* act);

/* flat parameter buffers.  Tensors are laid out back to back in the reference's
 * `parameters()` order (gen: gen_AB then gen_BA; dis: dis_A, dis_B, dis_2 -- trainer.py:37-38). */
int64_t aclgan_group_numel(const aclgan_ctx* ctx, int group);
int aclgan_tensor_count(const aclgan_ctx* ctx, int group);
/* name = "<net>/<reference state_dict key>"; shape = OIHW-order dims as the reference stores them
 * (conv: Co,Ci,k,k; linear: out,in; vectors: n); offset in floats into the group's flat buffer. */
int aclgan_tensor_info(const aclgan_ctx* ctx, int group, int index, char* name, int name_cap,
                       int64_t* offset, int* shape4, int* ndim);
int aclgan_bind_params(aclgan_ctx* ctx, int group, float* param, float* grad, float* exp_avg,
                       float* exp_avg_sq);

/* ---- reduced-precision compute (ACLGAN_DTYPE_*) ----
 * aclgan_set_compute_dtype: FP32 (default) or BF16 / FP16 for every convolution whose channel counts are multiples of
 * 32 (64 for the weight gradient); the image-side layers (Cin 3/6, Cout 1/4) and the MLP stay on the fp32 kernels.
 * aclgan_bind_params16: two caller-owned buffers of aclgan_group_numel(group) 16-bit elements per group -- w16 (same
 * offsets and OHWI layout as the fp32 parameters) and w16t (conv weights transposed to [tap][cin][cout], read by
 * dgrad).  The library refreshes them from the fp32 master copy at the start of every update / forward call. */
int aclgan_set_compute_dtype(aclgan_ctx* ctx, int dtype);
int aclgan_bind_params16(aclgan_ctx* ctx, int group, void* w16, void* w16t);
/* fp16 dynamic loss scaling.  state: device float[8], caller-owned, zero-initialised except state[0]:
 *   [0] scale S  [1] 1/S  [2] consecutive overflow-free updates  [3] overflow flag of the update in flight
 *   [4],[5] updates skipped so far (gen, dis)  [6] growth interval (0: 2000)  [7] the scale the LAST update's gradient buffers carry (written by aclgan_adam_step before it moves [0])
 * Every loss-gradient seed is multiplied by S; aclgan_adam_step first scans the group's gradients, and -- all on the
 * device, no host round trip -- either applies Adam with g/S or skips the update, halves S and counts the skip (the
 * bias-correction step excludes skipped updates); S doubles after `growth interval` clean updates.  The gradient
 * buffer read by the caller (e.g. for an all-reduce) holds S*g.  NULL unbinds. */
int aclgan_bind_loss_scale(aclgan_ctx* ctx, float* state);

/* activation workspace for one update at the given batch shape (bytes); bind before stepping */
int aclgan_workspace_bytes(aclgan_ctx* ctx, int B, int H, int W, size_t* out);
int aclgan_bind_workspace(aclgan_ctx* ctx, void* workspace, size_t bytes);
/* ACLGAN_OK iff the bound workspace holds both updates at this batch shape with the tuning switches as they are now, else ACLGAN_ENOMEM
 * with both sizes in aclgan_last_error().  Launches nothing (usable without a GPU).  aclgan_gen_update / aclgan_dis_update make the same
 * check before they enqueue anything, and the arena refuses every single request that would leave the bound range (round 5: an
 * undersized workspace fails, it never corrupts). */
int aclgan_check_workspace(aclgan_ctx* ctx, int B, int H, int W);
/* arena for ONE forward-only call below (encode / decode / discriminator forward) on (B,*,H,W) images: what
 * test.py:55-131 and trainer.sample need -- far smaller than a training step's, and without its shape constraints
 * (any H, W the networks accept, e.g. the 256x340 a Resize(256) of a non-square photo yields) */
int aclgan_forward_workspace_bytes(aclgan_ctx* ctx, int B, int H, int W, size_t* out);
/* ALGORITHMIC HBM bytes of one update at this batch shape (measurement support, SURVEY.md 8d / bench.py roofline.traffic):
 * every operator of the step (trainer.py:99-169 / 254-292: convolutions, norms, activations, blends, losses and their
 * backward) counted with its inputs read once and its outputs written once at their storage width, from a launch-free dry
 * run of the same scheduler; zero_grad (4 B / parameter) and Adam (28 B / parameter, trainer.py:170,293) included.
 * which: 0 gen_update, 1 dis_update. */
int aclgan_step_algorithmic_bytes(aclgan_ctx* ctx, int which, int B, int H, int W, double* out);
/* Round 6: the matrix-pipe FLOPs the same update EXECUTES -- every convolution at the cost of the path its launchers choose at this shape,
 * compute dtype and switch setting (direct implicit GEMM 2 M Cout K; Winograd F(4x4,3x3) 36 multiplies per 4x4 tile instead of 144, counted
 * in whole tile blocks where the fused kernel runs; the sub-pixel layers as four 3x3 phases + the exact ring; the 4x4 stride-2 layers as four
 * parity phases where the fused kernel takes them).  bench.py's roofline.flop_per_launch; checked against rocprofv3 --pmc SQ_INSTS_MFMA
 * (scripts/step_mfma_flops.py, profiles/r06_step_traffic.json).  A dry run of the scheduler: no launches. */
int aclgan_step_executed_flops(aclgan_ctx* ctx, int which, int B, int H, int W, double* out);

/* ---- the hot path ---- */
/* aclgan_Trainer.gen_update minus zero_grad/opt.step (trainer.py:92-169): forward of the whole
 * generator/discriminator graph, the 12 generator losses, backward into the GEN group's grad
 * buffer (accumulating; call aclgan_zero_grad first).  x_a, x_b: (B,3,H,W) NCHW fp32; z: device
 * (3,B,style_dim) = z_1,z_2,z_3 (trainer.py:99-101).  losses: device float[ACLGAN_L_COUNT].
 * Streams (round 5): the work is ordered after everything already enqueued on `stream` and complete, for `stream`, when the call's last
 * enqueue is (a consumer ordered after `stream` sees the finished update) -- but inside the call independent branches of the update run on
 * up to three more streams of a process-wide pool and the parameter gradients on a fourth ("lanes", aclgan_tuning "lanes", 1 .. 3; 1 = `stream`
 * only plus the parameter-gradient stream).  The calling thread's current HIP device must be the device `stream` belongs to (the pool is
 * per device); two threads may step two contexts on one device concurrently (their work interleaves on the pooled streams, ordered by
 * each update's own events), one context is used by one thread at a time.  On an error return every internal stream has been drained. */
int aclgan_gen_update(aclgan_ctx* ctx, const float* x_a, const float* x_b, const float* z,
                      int B, int H, int W, const aclgan_hparams* hp, float* losses, void* stream);
/* aclgan_Trainer.dis_update minus zero_grad/opt.step (trainer.py:249-292); grads into the DIS group */
int aclgan_dis_update(aclgan_ctx* ctx, const float* x_a, const float* x_b, const float* z,
                      int B, int H, int W, const aclgan_hparams* hp, float* losses, void* stream);
/* ---- data-parallel gradient buckets (NOT in the reference: it is single-GPU, train.py:42) ----
 * The trained group's flat gradient buffer is cut into buckets of `bucket_elems` floats.  During the backward of
 * aclgan_gen_update / aclgan_dis_update, `fn(user, group, bucket, offset, numel)` is called ON THE HOST as soon as the
 * last kernel that accumulates into that bucket has been enqueued on the step's stream -- in reverse-backward order of
 * readiness, the same order on every rank.  The caller launches that bucket's all-reduce (ordered after the work enqueued
 * so far) so that it overlaps the remaining backward, and waits for all of them before aclgan_adam_step.
 * bucket_elems = 0 or fn = NULL switches the mechanism off. */
typedef void (*aclgan_bucket_fn)(void* user, int group, int bucket, int64_t offset, int64_t numel);
int aclgan_set_grad_buckets(aclgan_ctx* ctx, int64_t bucket_elems, aclgan_bucket_fn fn, void* user);
/* the bucket completion order of one update of `group` at this batch shape, from a launch-free dry run of the same
 * scheduler (no GPU work; usable on a CPU-only host): order[0..count).  fire != 0 also invokes the callback. */
int aclgan_bucket_schedule(aclgan_ctx* ctx, int group, int B, int H, int W, int fire, int* order, int cap, int* count);

/* Optional forward sync point of aclgan_gen_update for data-parallel runs that want the reference's GLOBAL-batch focus
 * losses (trainer.py:149-161 squares a sum over the whole batch, which per-rank evaluation + gradient averaging does not
 * reproduce): after the three masks' sums are on the device, fn(user, sums, 6) is called on the host with the DEVICE
 * pointer of float[6] = {sum(m - upper), sum 1/(|m-.5|+eps)} x {B, A, A2} (inside the bound workspace); the caller
 * enqueues a SUM all-reduce over `world_size` ranks on it; the step then uses the reduced values with N = world_size *
 * local pixels.  The reported focus losses / loss_gen_total become the global-batch values.  fn = NULL switches it off. */
typedef void (*aclgan_sync_fn)(void* user, float* sums_dev, int n);
int aclgan_set_forward_sync(aclgan_ctx* ctx, aclgan_sync_fn fn, void* user, int world_size);

/* opt.zero_grad() (trainer.py:91,248) */
int aclgan_zero_grad(aclgan_ctx* ctx, int group, void* stream);
/* opt.step() (trainer.py:170,293): one fused kernel over the group's flat buffers; `step` is the
 * 1-based Adam step count used for bias correction */
int aclgan_adam_step(aclgan_ctx* ctx, int group, const aclgan_adam* opt, int step, void* stream);

/* ---- forward-only entry points (test.py:55-70 / trainer.sample: encode / decode / D forward) ---- */
/* AdaINGen.encode (networks.py:141-145): content (B,C,H/4,W/4) NCHW, style (B,style_dim) */
int aclgan_gen_encode(aclgan_ctx* ctx, int net, const float* x, int B, int H, int W,
                      float* content, float* style, void* stream);
/* AdaINGen.decode (networks.py:147-152): content (B,C,h,w) NCHW, style (B,style_dim) -> (B,out,4h,4w) */
int aclgan_gen_decode(aclgan_ctx* ctx, int net, const float* content, const float* style,
                      int B, int h, int w, float* out, void* stream);
/* MsImageDis.forward (networks.py:50-57): x (B,C,H,W) NCHW -> num_scales maps, out[s] (B,1,hs,ws) */
int aclgan_dis_forward(aclgan_ctx* ctx, int net, const float* x, int B, int H, int W,
                       float* const* outs, void* stream);

/* ---- operator level (each one is also what the step is built from) ---- */
/* reflection_pad2d + conv2d (+ upsample_nearest2d) + bias + activation (networks.py:366-370) */
int aclgan_conv2d_fwd(const aclgan_conv_desc* d, const float* x, const float* w, const float* bias,
                      float* y, void* stream);
/* same, with an optional scratch buffer (aclgan_conv2d_fwd_scratch_bytes; may be 0 / NULL): enables the
 * sub-pixel path for the decoder's "Upsample(2) + 5x5" layers (networks.py:256-257) and the ORDERED reduction
 * of split-K partials on small grids (late discriminator layers): same result as without scratch up to fp32
 * summation order, and reproducible bit for bit from call to call (the scratch-less call combines split-K
 * partials with fp32 atomics) */
int aclgan_conv2d_fwd_ws(const aclgan_conv_desc* d, const float* x, const float* w, const float* bias,
                         float* y, void* scratch, void* stream);
size_t aclgan_conv2d_fwd_scratch_bytes(const aclgan_conv_desc* d);
/* convolution_backward w.r.t. the input (autograd of networks.py:366, incl. the pad / upsample
 * backward).  scratch: aclgan_conv2d_dgrad_scratch_bytes(d) bytes.  accumulate != 0: dx += */
int aclgan_conv2d_dgrad(const aclgan_conv_desc* d, const float* dy, const float* w, float* dx,
                        void* scratch, int accumulate, void* stream);
size_t aclgan_conv2d_dgrad_scratch_bytes(const aclgan_conv_desc* d);
/* convolution_backward w.r.t. weight and bias; ALWAYS accumulates (dw +=, db +=) */
int aclgan_conv2d_wgrad(const aclgan_conv_desc* d, const float* x, const float* dy, float* dw,
                        float* db, void* stream);
/* wgrad with an optional scratch buffer of aclgan_conv2d_wgrad_scratch_bytes(d) bytes (may be 0): enables the
 * sub-pixel path of the upsample+5x5 layers and the two-stage reduction of the thin 7x7 layers (3 -> 64 and
 * 64 -> 4 channels, networks.py:216,234,260) -- same result up to fp32 summation order */
int aclgan_conv2d_wgrad_ws(const aclgan_conv_desc* d, const float* x, const float* dy, float* dw,
                           float* db, void* scratch, void* stream);
size_t aclgan_conv2d_wgrad_scratch_bytes(const aclgan_conv_desc* d);
/* ---- the same three operators on the 16-bit matrix cores (dtype = ACLGAN_DTYPE_BF16 / _FP16) ----
 * which: 0 forward, 1 dgrad, 2 wgrad -> 1 when that operator has a 16-bit kernel for this shape */
int aclgan_conv16_eligible(const aclgan_conv_desc* d, int which);
/* fp32 OHWI weights [Co][taps][Ci] -> 16-bit packs: w16 (same layout) and/or w16t ([tap][ci][co]); either may be NULL */
int aclgan_pack_weights16(const float* w, void* w16, void* w16t, int Co, int taps, int Ci, int dtype, void* stream);
/* w: the fp32 weights (needed by the upsample+5x5 layers, whose phase filters are merged before rounding; else may be
 * NULL).  scratch: aclgan_conv2d_*16_scratch_bytes (NULL allowed when that is 0) */
int aclgan_conv2d_fwd16(const aclgan_conv_desc* d, int dtype, const float* x, const float* w, const void* w16,
                        const float* bias, float* y, void* scratch, void* stream);
/* forward with the input ALREADY rounded to the 16-bit type by its producer (x16: same NHWC layout, bf16 / fp16 bit patterns;
 * aclgan_pack_weights16(x, x16, NULL, B*H*W, 1, C, ...) makes one): half the activation bytes, no conversion, same result bit for bit */
int aclgan_conv2d_fwd16_x16(const aclgan_conv_desc* d, int dtype, const void* x16, const float* w, const void* w16,
                            const float* bias, float* y, void* scratch, void* stream);
int aclgan_conv2d_dgrad16(const aclgan_conv_desc* d, int dtype, const float* dy, const float* w, const void* w16t,
                          float* dx, int accumulate, void* scratch, void* stream);
int aclgan_conv2d_wgrad16(const aclgan_conv_desc* d, int dtype, const float* x, const float* dy, float* dw,
                          float* db, void* scratch, void* stream);

/* ---- round 3: 16-bit ACTIVATION / GRADIENT STORAGE ("bf16 activations / MFMA, fp32 master") ----
 * Under a 16-bit compute dtype the step keeps the activations and activation gradients of its wide layers (C % 64 == 0) in HBM in
 * that dtype: the operand of networks.py:366's convolution is the same rounded value whether its producer or the conv loader
 * rounds it, but the bytes halve and the operand tiles can go global -> LDS directly (csrc/conv_glds16.hip).  Storage codes:
 * 0 = fp32, ACLGAN_DTYPE_BF16, ACLGAN_DTYPE_FP16.  Statistics, losses, master weights, weight gradients, Adam: fp32 as before. */
/* which: 0 forward, 1 dgrad.  1 = the LDS-DMA kernels take this shape (no upsample, Cin and Cout multiples of 64, forward: grid that
 * fills the chip without split-K) */
int aclgan_conv16s_ok(const aclgan_conv_desc* d, int which);
/* forward on 16-bit x (NHWC) and the OHWI weight pack: y (storage y_storage) = act(conv(x16) + bias) */
int aclgan_conv2d_fwd16s(const aclgan_conv_desc* d, int dtype, const void* x16, const void* w16, const float* bias, void* y, int y_storage, void* stream);
/* Process-wide tuning knobs that also exist as environment variables, settable at run time (tests use this to run every tile shape of
 * csrc/conv_glds16.hip).  key "glds_tile": 0 / 1 = 128-row tiles (default), 2 = 256 x 128, 3 = 256 x 256 where the shape allows, 4 = the
 * largest tile that still fills the chip.  key "wino_x3": 1 = the GEMM slices of the fp32 Winograd pipeline run as split-bf16
 * products on the bf16 matrix cores (fp32-accurate, see aclgan_gemm_slices_x3), 0 (default) = on the fp32 MFMA kernel.
 * key "dgrad16s_direct": 1 = aclgan_conv2d_dgrad16s stores pixels without mirrored partners straight into a 16-bit dx (bit-identical,
 * measured neutral), 0 (default) = every pixel through the padded scratch and the ordered fold.
 * key "wino_fused" (round 4): 0 = the 3x3 ResBlock / sub-pixel-phase convolutions never take the one-launch Winograd kernel
 * (csrc/conv_wino_fused.hip), 1 (default) = where its cost model says it pays, 2 = wherever the shape is eligible.
 * key "wino_wgrad_fused" (round 4): 0 = their weight gradient runs as the seven-launch pipeline of csrc/conv_wino.hip, 1 (default) / 2 = as the
 * one-kernel Winograd weight gradient (csrc/conv_wino_wgrad_fused.hip) wherever the shape is eligible (W a multiple of 16, H of 4, Cout of 64,
 * Cin of 32).
 * key "wino_s2k4" (round 6): 1 (default) = the 4x4 stride-2 reflect-pad-1 layers (networks.py:41, 216-221, 236-241) run as four parity phases
 * of the same one-launch Winograd kernel wherever "wino_fused" sends them there (1: a cost model per shape, 2: every eligible shape) --
 * forward and the interior of the input gradient; 0 = they keep the direct implicit-GEMM kernels (ACLGAN_NOWINOS2=1).
 * key "fwd16_patch" (round 5): 1 (default) = the 3x3 stride-1 layers on 32- / 64-pixel-wide maps (the ResBlock convolutions) take
 * conv_fwd16p_kernel (csrc/conv_glds16.hip: the reflect-padded input patch of a 256-pixel tile stays in LDS for all nine taps; 2 = its
 * counter-phase schedule, a measured alternative), 0 = conv_fwd16s.
 * key "lanes" (round 5): 1 .. 3 HIP streams the independent branches of an update are spread over (the two translation directions,
 * the reconstruction decodes, the discriminators and their scales: reference trainer.py:103-139, 258-286; csrc/engine.hip "Lanes");
 * 1 = one queue (the round-4 plan), default 3 (ACLGAN_LANES).  Results do not depend on it.  key "u_batch" (round 5): 1 (default) =
 * the Winograd transforms of all ResBlock filters of a network are one launch at the start of an update, 0 = one launch per filter at
 * its first use.  key "norm_mask" (round 5): 1 (default) = the backward of an activated normalisation layer recomputes the ReLU mask from x and
 * the forward's fused coefficients instead of reading y back (same mask: the same fmaf), 0 = reads y.  key "mlp_fused" (round 6): 1 (default) = the
 * generator's MLP forward (networks.py:280-292) is one launch (aclgan_mlp3_fwd), 0 = three aclgan_linear_fwd launches; the same bits either way.
 * key "fault_at" (test hook; -1 = off): the backward replay of the next updates fails with ACLGAN_EHIP after that many
 * closures have been enqueued -- exercises the error path (all internal streams drained before the call returns).
 * Returns the previous value, -1 for an unknown key.  The switches are atomics (a concurrent update sees the old or the new value, never a
 * torn one), but changing one WHILE an update is being enqueued changes that update's plan half way: do not. */
int aclgan_set_tuning(const char* key, int value);
/* The same switches with the status in the return value (round 5): ACLGAN_OK and the previous setting through *previous (may be NULL), or
 * ACLGAN_EINVAL for an unknown key (aclgan_set_tuning cannot tell -1 "unknown" from a previous value). */
int aclgan_tuning(const char* key, int value, int* previous);
/* Read a switch without touching it (round 6; no state change, no tuning-epoch bump): keys "lanes", "u_batch", "norm_mask", "mlp_fused", "wino_fused",
 * "wino_wgrad_fused", "wino_s2k4", "fault_at", and "epoch" = the number of aclgan_tuning calls so far (cached switch-dependent results -- an arena
 * size -- are valid for one epoch).  ACLGAN_EINVAL for any other key. */
int aclgan_tuning_get(const char* key, long long* value);
/* The same launch with the normalisation statistics taken from its epilogue (round 3; replaces the norm_stats pass over y that
 * follows the conv in reference networks.py:382-395 Conv2dBlock.forward -> self.norm).  aclgan_conv2d_fwd16s_stats_chunk = rows R per
 * statistics chunk (the launch's row tile: 128 or 256; 0 = not offered: Ho * Wo must be a multiple of R).  stats receives
 * [B * Ho * Wo / R][Cout] float2 (mean, M2 = sum of squared deviations) of the STORED outputs of each chunk, chunk c = rows c R .. c R + R - 1
 * of the [B * Ho * Wo][Cout] output. */
int aclgan_conv2d_fwd16s_stats_chunk(const aclgan_conv_desc* d);
int aclgan_conv2d_fwd16s_stats(const aclgan_conv_desc* d, int dtype, const void* x16, const void* w16, const float* bias, void* y, int y_storage,
                               float* stats, void* stream);
/* dgrad on 16-bit dy and the transposed weight pack: dx (storage dx_storage) (+)= ...; ONE launch over the padded grid into `scratch`
 * + an ordered fold of the reflection: no atomics, bit-reproducible */
size_t aclgan_conv2d_dgrad16s_scratch_bytes(const aclgan_conv_desc* d);
int aclgan_conv2d_dgrad16s(const aclgan_conv_desc* d, int dtype, const void* dy16, const void* w16t, void* dx, int dx_storage, int accumulate,
                           void* scratch, void* stream);
/* aclgan_conv2d_wgrad16 with either operand stored in the 16-bit dtype */
int aclgan_conv2d_wgrad16_st(const aclgan_conv_desc* d, int dtype, const void* x, int x_storage, const void* dy, int dy_storage, float* dw, float* db,
                             void* scratch, void* stream);
/* elementwise storage conversion, n % 4 == 0 */
int aclgan_cast_storage(const void* src, int src_storage, void* dst, int dst_storage, int64_t n, void* stream);
/* aclgan_norm_fwd / aclgan_norm_bwd on tensors of any storage: storage = {x, y, residual} / {x, y, dy, dx, dres} */
int aclgan_norm_fwd_st(int kind, int act, int B, int HW, int C, const void* x, const float* w, const float* b, int w_stride, const void* residual,
                       void* y, float* mean, float* rstd, void* scratch, const int* storage, void* stream);
int aclgan_norm_bwd_st(int kind, int act, int B, int HW, int C, const void* x, const void* y, const void* dy, const float* w, int w_stride,
                       const float* mean, const float* rstd, void* dx, float* dw, float* db, void* dres, int dres_accumulate, void* scratch,
                       const int* storage, void* stream);
size_t aclgan_conv2d_fwd16_scratch_bytes(const aclgan_conv_desc* d);
size_t aclgan_conv2d_dgrad16_scratch_bytes(const aclgan_conv_desc* d);
size_t aclgan_conv2d_wgrad16_scratch_bytes(const aclgan_conv_desc* d);
/* same three, but the plain one-thread-per-output kernels (no MFMA): on-device cross-check */
int aclgan_conv2d_fwd_naive(const aclgan_conv_desc* d, const float* x, const float* w,
                            const float* bias, float* y, void* stream);

/* InstanceNorm2d / AdaptiveInstanceNorm2d / custom LayerNorm + activation + residual
 * (networks.py:333,367-370,477-536,309), NHWC [B][HW][C].
 *   kind IN:    y = act((x-mu)*rstd) (+res)                      rstd = 1/sqrt(var_biased + 1e-5)
 *   kind ADAIN: y = act((x-mu)*rstd*w[b][c] + bias[b][c]) (+res)
 *   kind LN:    y = act((x-mu_b)/(std_unbiased_b + 1e-5)*gamma[c] + beta[c])
 * stats out: mean/rstd per (b,c) for IN/ADAIN ([B][C] each), per b for LN ([B] each; "rstd" holds
 * 1/(std+eps)).  scratch: aclgan_norm_scratch_bytes. */
int aclgan_norm_fwd(int kind, int act, int B, int HW, int C, const float* x, const float* w,
                    const float* b, int w_stride, const float* residual, float* y, float* mean,
                    float* rstd, void* scratch, void* stream);
/* Conv2dBlock.forward (networks.py:365-371): conv (desc.act must be none) + norm + activation (+ residual), as the step runs it.
 * y_conv receives the convolution output (kept for the backward), y the block output.  Where the forward kernel holds whole
 * output tiles (the Winograd output transform of the 3x3 ResBlock convolutions) the normalisation statistics are emitted from
 * the conv epilogue and the separate statistics pass over y_conv is skipped; *stats_fused (optional) reports which happened.
 * Same result as aclgan_conv2d_fwd_ws followed by aclgan_norm_fwd up to fp32 summation order of the statistics.
 * scratch: aclgan_conv2d_block_fwd_scratch_bytes(d) bytes. */
int aclgan_conv2d_block_fwd(const aclgan_conv_desc* d, int norm_kind, int act, const float* x, const float* w,
                            const float* bias, const float* nw, const float* nb, int n_stride, const float* residual,
                            float* y_conv, float* y, float* mean, float* rstd, void* scratch, int* stats_fused,
                            void* stream);
size_t aclgan_conv2d_block_fwd_scratch_bytes(const aclgan_conv_desc* d);
/* backward: given dy (grad of y) computes dx (grad of x), and ACCUMULATES dw/db ([B][C] for ADAIN
 * with row stride w_stride, [C] for LN; ignored for IN) and, if dres != NULL, dres (+)= the
 * activation-masked dy (the residual branch).  dres_accumulate selects += vs =. */
int aclgan_norm_bwd(int kind, int act, int B, int HW, int C, const float* x, const float* y,
                    const float* dy, const float* w, int w_stride, const float* mean,
                    const float* rstd, float* dx, float* dw, float* db, float* dres,
                    int dres_accumulate, void* scratch, void* stream);
size_t aclgan_norm_scratch_bytes(int B, int HW, int C);

/* AvgPool2d(3, stride 2, pad 1, count_include_pad=False) (networks.py:33), NHWC */
int aclgan_avgpool3s2_fwd(int B, int H, int W, int C, const float* x, float* y, void* stream);
int aclgan_avgpool3s2_bwd(int B, int H, int W, int C, const float* dy, float* dx, int accumulate,
                          void* stream);

/* LinearBlock / MLP layer (networks.py:373-418, 280-292) and the style head's 1x1 conv on a pooled vector (networks.py:223):
 * y[b][o] = act(sum_i x[b][i] W[o][i] + bias[o]).  bwd: dy is modified in place by the activation backward; dx is
 * overwritten (may be NULL); dw, db accumulate (may be NULL). */
int aclgan_linear_fwd(int B, int I, int O, const float* x, const float* w, const float* bias, int act, float* y, void* stream);
int aclgan_linear_bwd(int B, int I, int O, const float* x, const float* y, float* dy, const float* w, int act,
                      float* dx, float* dw, float* db, void* stream);
/* MLP.forward (networks.py:280-292; AdaINGen.decode's style -> AdaIN parameters, networks.py:147-151) as ONE launch:
 * m0 = relu(w0 s + b0) [B][M], m1 = relu(w1 m0 + b1) [B][M], ap = w2 m1 + b2 [B][O]; s is [B][S].  Bit-identical to three aclgan_linear_fwd
 * calls (same per-lane order, same reduction tree).  ACLGAN_EUNSUPPORTED outside S <= 64, M in {64, 128, 192, 256} (the engine then runs
 * the three launches).  The backward stays aclgan_linear_bwd per layer on m0 / m1 / ap. */
int aclgan_mlp3_fwd(int B, int S, int M, int O, const float* s, const float* w0, const float* b0, const float* w1, const float* b1,
                    const float* w2, const float* b2, float* m0, float* m1, float* ap, void* stream);
/* AdaptiveAvgPool2d(1) (networks.py:222), NHWC [B][HW][C] -> [B][C] */
int aclgan_gap_fwd(int B, int HW, int C, const float* x, float* y, void* stream);
int aclgan_gap_bwd(int B, int HW, int C, const float* dy, float* dx, int accumulate, void* stream);
/* aclgan_Trainer.focus_translation (trainer.py:85-88) fused with the 6-channel pair concat (trainer.py:132-133), NHWC:
 * dec4 [B][HW][4] (decoder output: ch0-2 image, ch3 focus), bg [B][HW][3] -> out [B][HW][3]; optional pair [B][HW][6] =
 * (pair_first, out).  bwd: d_dec4 += (zero-initialise it), d_bg (+)= (may be NULL); d_out / d_pair may each be NULL. */
int aclgan_focus_blend_fwd(int B, int HW, const float* dec4, const float* bg, float* out, const float* pair_first,
                           float* pair, void* stream);
int aclgan_focus_blend_bwd(int B, int HW, const float* dec4, const float* bg, const float* d_out, const float* d_pair,
                           float* d_dec4, float* d_bg, int bg_accumulate, void* stream);
/* the same blend on the reference's NCHW tensors, for sample() / test.py (trainer.py:179-245, test.py:73-76): fg, bg
 * (B,3,H,W), focus (B,1,H,W), out (B,3,H,W) contiguous; *_bstride = floats between consecutive samples of that input
 * (so fg / focus may be the two channel slices of one (B,4,H,W) decoder output) */
int aclgan_focus_translation_nchw(const float* fg, int64_t fg_bstride, const float* bg, int64_t bg_bstride,
                                  const float* focus, int64_t focus_bstride, float* out, int B, int HW, void* stream);
/* MsImageDis.calc_*_loss, one scale (networks.py:67,83,98): *loss_slot += weight*mean((o-target)^2);
 * d_o (may be NULL) = gscale*weight*2(o-target)/n */
int aclgan_lsgan_loss(const float* o, int n, float target, float weight, float* loss_slot, float* d_o, float gscale, void* stream);
/* ... and all scales / segments of one MsImageDis.calc_*_loss call in ONE launch (round 5; the reference loops over the scales in Python,
 * networks.py:64-67,81-83,96-98): nterms terms, arrays indexed by term; d_o (the array or single entries) may be NULL.  The terms are
 * processed in index order by one workgroup: every slot ends up with exactly the bits the equivalent sequence of aclgan_lsgan_loss
 * calls leaves there. */
int aclgan_lsgan_loss_multi(const float* const* o, const int* n, const float* target, const float* weight, float* const* loss_slot,
                            float* const* d_o, const float* gscale, int nterms, void* stream);
/* recon_criterion (trainer.py:61-62): *loss_slot += mean|a[..., :3] - b| over npix pixels; a has a_channels (3 or 4) NHWC
 * channels, b has 3; d_a (may be NULL) (+)= gscale*sign/(3*npix) on channels 0-2 */
int aclgan_l1_loss(const float* a, int a_channels, const float* b, int64_t npix, float* loss_slot, float* d_a, float gscale,
                   int d_accumulate, void* stream);
/* focus losses of one mask (trainer.py:146-158): dec4 NHWC [npix][4], m = (ch3+1)/2;
 * *size_slot = delta*(relu(sum(m-upper))^2 + relu(sum(lower-m))^2), *digit_slot = sum 1/(|m-0.5|+eps);
 * d_dec4 (may be NULL) ch3 += scale * d(size+digit)/d ch3.  scratch: aclgan_focus_loss_scratch_bytes(npix). */
size_t aclgan_focus_loss_scratch_bytes(int64_t npix);
/* the same with the two sums supplied by the caller (device float[2]: sum(m - upper), digit sum over npix_total pixels):
 * one shard of a larger batch -- see aclgan_set_forward_sync */
int aclgan_focus_loss_global(const float* dec4, int64_t npix, const float* totals, int64_t npix_total, float delta, float upper,
                             float lower, float eps, float scale, float* size_slot, float* digit_slot, float* d_dec4, void* stream);
int aclgan_focus_loss(const float* dec4, int64_t npix, float delta, float upper, float lower, float eps, float scale,
                      float* size_slot, float* digit_slot, float* d_dec4, void* scratch, void* stream);

/* torch.optim.Adam over a flat buffer (trainer.py:39-42,170,293) */
int aclgan_adam_flat(float* p, const float* g, float* m, float* v, int64_t n,
                     const aclgan_adam* opt, int step, void* stream);

/* NCHW <-> NHWC (boundary layout conversion) */
int aclgan_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, void* stream);
int aclgan_nhwc_to_nchw(const float* src, float* dst, int B, int C, int H, int W, void* stream);

/* ---- training input pipeline (reference utils.py:78-100 get_data_loader_folder / get_data_loader_list:
 * RandomHorizontalFlip -> Resize(new_size) -> RandomCrop((h, w)) -> ToTensor -> Normalize(0.5, 0.5), run by
 * torchvision 0.4.0 on PIL images, pillow==6.2.1) ----
 * One descriptor per decoded image of the batch.  The host decides the random draws (flip, crop offset) exactly
 * where torchvision does; the device does the pixel work.  Resampling is Pillow's 8-bit fixed-point two-pass
 * bilinear (src/libImaging/Resample.c): bit-identical to Image.resize(..., Image.BILINEAR). */
typedef struct aclgan_image_desc {
    int64_t src_offset;      /* byte offset of this image's HWC RGB uint8 pixels in the packed source buffer */
    int src_h, src_w;        /* decoded size */
    int res_h, res_w;        /* size after Resize(new_size) (== src size when Resize is a no-op) */
    int crop_y, crop_x;      /* RandomCrop offset (i, j) inside the resized image */
    int flip;                /* RandomHorizontalFlip drew < 0.5 */
    int tab_x, tab_y;        /* int32 index of this image's horizontal / vertical table inside `tables` */
    int ksize_x, ksize_y;    /* aclgan_image_resample_ksize(src_w, res_w) / (src_h, res_h) */
} aclgan_image_desc;

/* HOST functions (no GPU needed): Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter.
 * ksize = number of coefficient slots per output position; the table of one axis is
 *   bounds[out_size][2] = (first source index, count)  followed by  coeffs[out_size][ksize] (22-bit fixed point).
 * in_size == out_size yields the identity table (Pillow skips that pass). */
int aclgan_image_resample_ksize(int in_size, int out_size);
int aclgan_image_resample_coeffs(int in_size, int out_size, int* bounds, int* coeffs);

/* src: packed uint8 HWC RGB images (device); descs_host / descs_dev: the same n descriptors on host (validated
 * here) and on the device (read by the kernel); tables_dev: int32 tables (device); out: float32 [n][3][out_h][out_w]
 * in [-1, 1] (the x_a / x_b layout aclgan_gen_update / aclgan_dis_update consume).  Errors like torchvision's:
 * a resized image smaller than the crop is rejected. */
int aclgan_image_batch_transform(const void* src, const aclgan_image_desc* descs_host, const void* descs_dev, int n,
                                 const void* tables_dev, float* out, int out_h, int out_w, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ACLGAN_HIP_H */
