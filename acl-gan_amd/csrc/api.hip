// api.hip -- error plumbing and the operator-level C ABI of libaclgan_hip (see include/aclgan_hip.h).
#include <string.h>
#include "common.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

namespace aclgan {

static thread_local char g_err[512] = "";
std::atomic<long long> g_launches{0};

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
int hip_fail(hipError_t e, const char* what) {
    set_error("%s: HIP error %d (%s)", what, (int)e, hipGetErrorString(e));
    return ACLGAN_EHIP;
}

// scheduler switches (common.h)
static std::atomic<int> g_lanes{-1}, g_u_batch{-1}, g_norm_mask{-1}, g_mlp_fused{-1};
static std::atomic<long long> g_tuning_epoch{0};
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return (e && *e) ? atoi(e) : dflt; }
// (1 .. 3: the updates assign work to lanes 0 .. 2 only; a value of 4 used to be accepted, ran the 3-lane plan and still created a 4th pooled
//  stream, which shifts HIP's stream -> hardware-queue placement -- clamped since round 6)
int lanes_setting() { int v = g_lanes.load(); if (v < 0) { v = std::max(1, std::min(3, env_int("ACLGAN_LANES", 3))); g_lanes.store(v); } return v; }
int set_lanes(int v) { const int old = lanes_setting(); g_lanes.store(std::max(1, std::min(3, v))); return old; }
int u_batch_setting() { int v = g_u_batch.load(); if (v < 0) { v = env_int("ACLGAN_U_BATCH", 1) ? 1 : 0; g_u_batch.store(v); } return v; }
int set_u_batch(int v) { const int old = u_batch_setting(); g_u_batch.store(v ? 1 : 0); return old; }
int norm_mask_setting() { int v = g_norm_mask.load(); if (v < 0) { v = env_int("ACLGAN_NORM_MASK", 1) ? 1 : 0; g_norm_mask.store(v); } return v; }
int set_norm_mask(int v) { const int old = norm_mask_setting(); g_norm_mask.store(v ? 1 : 0); return old; }
int mlp_fused_setting() { int v = g_mlp_fused.load(); if (v < 0) { v = env_int("ACLGAN_MLP_FUSED", 1) ? 1 : 0; g_mlp_fused.store(v); } return v; }
int set_mlp_fused(int v) { const int old = mlp_fused_setting(); g_mlp_fused.store(v ? 1 : 0); return old; }
static std::atomic<int> g_fault_at{-1};
int fault_at_setting() { return g_fault_at.load(); }
int set_fault_at(int v) { return g_fault_at.exchange(v); }
long long tuning_epoch() { return g_tuning_epoch.load(); }
void bump_tuning_epoch() { ++g_tuning_epoch; }

static int g_determ = -1;
bool deterministic() {
    if (g_determ < 0) { const char* e = getenv("ACLGAN_DETERMINISTIC"); g_determ = (e && atoi(e)) ? 1 : 0; }
    return g_determ == 1;
}
void set_deterministic(int on) { g_determ = on ? 1 : 0; }

}  // namespace aclgan

using namespace aclgan;

#include <algorithm>

namespace aclgan { int gemm_slices_f32(const float* A, const float* Bm, float* Cm, int T, int K, int N, int nslices, int a_mod, hipStream_t st); }
extern "C" {

int aclgan_set_deterministic(int on) { set_deterministic(on); return ACLGAN_OK; }
int aclgan_get_deterministic(void) { return deterministic() ? 1 : 0; }
int aclgan_version(void) { return 300; }   // 0.3.0: 16-bit activation storage, measurement counters
long long aclgan_launch_count(void) { return g_launches; }
const char* aclgan_last_error(void) { return g_err; }

// the batched-GEMM launch of the Winograd pipeline alone (the step's dominant kernel): measurement / test support
size_t aclgan_gemm_slices_x3_scratch_bytes(int T, int K, int N, int nslices) {
    return (size_t)3 * 2 * ((size_t)nslices * T * K + (size_t)nslices * N * K) + 512;
}
int aclgan_gemm_slices_x3(const float* A, const float* Bm, float* Cm, int T, int K, int N, int nslices, void* scratch, void* stream) {
    ACL_REQUIRE(Cm && scratch && T > 0 && K > 0 && N > 0 && nslices > 0 && ((A && Bm) || (!A && !Bm)), "gemm_slices_x3: bad argument");
    if (!gemm_x3_shape_ok(T, K, N)) { set_error("gemm_slices_x3: K must be a multiple of 32 and N of 64"); return ACLGAN_EUNSUPPORTED; }
    const int64_t na = (int64_t)nslices * T * K, nb = (int64_t)nslices * N * K;
    char* a3 = (char*)scratch;
    char* b3 = a3 + (((size_t)6 * na + 255) & ~(size_t)255);
    int rc = ACLGAN_OK;
    if (A) {          // (A == B == NULL: the planes of an earlier call are still in scratch -- bench.py times the GEMM launch alone that way)
        rc = split3_planes(A, a3, na, na, (hipStream_t)stream);
        if (rc) return rc;
        rc = split3_planes(Bm, b3, nb, nb, (hipStream_t)stream);
        if (rc) return rc;
    }
    return gemm_slices_x3(a3, (size_t)na * 2, b3, (size_t)nb * 2, Cm, T, K, N, nslices, 0, (hipStream_t)stream);
}
int aclgan_gemm_slices_f32(const float* A, const float* Bm, float* Cm, int T, int K, int N, int nslices, void* stream) {
    ACL_REQUIRE(A && Bm && Cm && nslices >= 1, "null argument");
    return gemm_slices_f32(A, Bm, Cm, T, K, N, nslices, 0, (hipStream_t)stream);
}

int aclgan_winograd_filter_frag(const float* w, float* Uf, int Co, int Ci, int flip, void* stream) {
    ACL_REQUIRE(w && Uf && Co > 0 && Ci > 0, "winograd_filter_frag: bad argument");
    ACL_REQUIRE((flip ? Co : Ci) % 16 == 0 && (flip ? Ci : Co) % 64 == 0, "winograd_filter_frag: the K side must be a multiple of 16, the row side of 64");
    return wino_fused_filter(w, Uf, Co, Ci, flip, (hipStream_t)stream);
}
int aclgan_conv3x3_winograd_fused(const float* x, const float* Uf, const float* bias, float* y, int B, int H, int W, int Cin, int Cout, int act, int reflect,
                                  int accumulate, float* stats, void* stream) {
    ACL_REQUIRE(x && Uf && y && B > 0, "conv3x3_winograd_fused: null argument");
    const int old = wino_fused_force(1);      // (the entry point IS the fused kernel, whatever the step's switch and cost model say: per thread)
    const int rc = wino_fused_launch(B, H, W, Cin, Cout, x, Uf, bias, y, act, accumulate, reflect, (float2*)stats, (hipStream_t)stream);
    wino_fused_force(old);
    if (rc == ACLGAN_EUNSUPPORTED) set_error("conv3x3_winograd_fused: shape not eligible (H, W multiples of 4, Cin of 16, Cout of 64, no tanh)");
    return rc;
}

int aclgan_conv2d_fwd(const aclgan_conv_desc* d, const float* x, const float* w, const float* bias, float* y, void* stream) {
    ConvGeom g;
    int rc = make_geom(d, &g);
    if (rc) return rc;
    ACL_REQUIRE(x && w && y, "conv2d_fwd: null buffer");
    return conv_fwd(g, x, w, bias, y, (hipStream_t)stream);
}
int aclgan_conv2d_fwd_ws(const aclgan_conv_desc* d, const float* x, const float* w, const float* bias, float* y, void* scratch, void* stream) {
    ConvGeom g;
    int rc = make_geom(d, &g);
    if (rc) return rc;
    ACL_REQUIRE(x && w && y, "conv2d_fwd_ws: null buffer");
    return conv_fwd(g, x, w, bias, y, (hipStream_t)stream, scratch);
}
// Conv2dBlock.forward = conv + norm + activation (+ residual) as the engine runs it: where the forward kernel can, the normalisation
// statistics come out of the conv epilogue (conv_fwd_stats_chunk) and the separate statistics pass is skipped
static size_t block_stats_bytes(const ConvGeom& g) {
    const int ch = conv_fwd_stats_chunk(g);
    return ch ? (((size_t)2 * g.B * (g.Ho * g.Wo / ch) * g.Co * sizeof(float)) + 255) & ~(size_t)255 : 0;
}
size_t aclgan_conv2d_block_fwd_scratch_bytes(const aclgan_conv_desc* d) {
    ConvGeom g;
    if (make_geom(d, &g)) return 0;
    return block_stats_bytes(g) + std::max(conv_fwd_scratch_bytes(g), norm_scratch_bytes(g.B, g.Ho * g.Wo, g.Co)) + 256;
}
int aclgan_conv2d_block_fwd(const aclgan_conv_desc* d, int norm_kind, int act, const float* x, const float* w, const float* bias, const float* nw,
                            const float* nb, int n_stride, const float* residual, float* y_conv, float* y, float* mean, float* rstd, void* scratch,
                            int* stats_fused, void* stream) {
    ConvGeom g;
    int rc = make_geom(d, &g);
    if (rc) return rc;
    ACL_REQUIRE(x && w && y_conv && y && mean && rstd && scratch, "conv2d_block_fwd: null buffer");
    ACL_REQUIRE(d->act == ACLGAN_ACT_NONE, "conv2d_block_fwd: the activation follows the norm (pass it as `act`, desc.act must be none)");
    const size_t sb = block_stats_bytes(g);
    float* stats = sb ? (float*)scratch : nullptr;
    void* rest = (char*)scratch + sb;
    if (stats_fused) *stats_fused = stats ? 1 : 0;
    rc = conv_fwd(g, x, w, bias, y_conv, (hipStream_t)stream, rest, stats);
    if (rc) return rc;
    return norm_fwd(norm_kind, act, g.B, g.Ho * g.Wo, g.Co, y_conv, nw, nb, n_stride, residual, y, mean, rstd, rest, (hipStream_t)stream, stats,
                    conv_fwd_stats_chunk(g));
}
size_t aclgan_conv2d_fwd_scratch_bytes(const aclgan_conv_desc* d) {
    ConvGeom g;
    if (make_geom(d, &g)) return 0;
    return conv_fwd_scratch_bytes(g);
}
int aclgan_conv2d_fwd_naive(const aclgan_conv_desc* d, const float* x, const float* w, const float* bias, float* y, void* stream) {
    ConvGeom g;
    int rc = make_geom(d, &g);
    if (rc) return rc;
    ACL_REQUIRE(x && w && y, "conv2d_fwd_naive: null buffer");
    return conv_fwd_naive(g, x, w, bias, y, (hipStream_t)stream);
}
size_t aclgan_conv2d_dgrad_scratch_bytes(const aclgan_conv_desc* d) {
    ConvGeom g;
    if (make_geom(d, &g)) return 0;
    return conv_dgrad_scratch_bytes(g);
}
int aclgan_conv2d_dgrad(const aclgan_conv_desc* d, const float* dy, const float* w, float* dx, void* scratch, int accumulate, void* stream) {
    ConvGeom g;
    int rc = make_geom(d, &g);
    if (rc) return rc;
    ACL_REQUIRE(dy && w && dx && scratch, "conv2d_dgrad: null buffer");
    return conv_dgrad(g, dy, w, dx, scratch, accumulate, (hipStream_t)stream);
}
int aclgan_conv2d_wgrad(const aclgan_conv_desc* d, const float* x, const float* dy, float* dw, float* db, void* stream) {
    ConvGeom g;
    int rc = make_geom(d, &g);
    if (rc) return rc;
    ACL_REQUIRE(x && dy, "conv2d_wgrad: null buffer");
    return conv_wgrad(g, x, dy, dw, db, (hipStream_t)stream);
}

size_t aclgan_conv2d_wgrad_scratch_bytes(const aclgan_conv_desc* d) {
    ConvGeom g;
    if (make_geom(d, &g)) return 0;
    return conv_wgrad_scratch_bytes(g);
}
int aclgan_conv2d_wgrad_ws(const aclgan_conv_desc* d, const float* x, const float* dy, float* dw, float* db, void* scratch, void* stream) {
    ConvGeom g;
    int rc = make_geom(d, &g);
    if (rc) return rc;
    ACL_REQUIRE(x && dy, "conv2d_wgrad_ws: null buffer");
    return conv_wgrad(g, x, dy, dw, db, (hipStream_t)stream, scratch);
}

// ---- 16-bit MFMA operators ----
int aclgan_conv16_eligible(const aclgan_conv_desc* d, int which) {
    ConvGeom g;
    if (make_geom(d, &g) || which < 0 || which > 2) return 0;
    return conv16_eligible(g, which) ? 1 : 0;
}
int aclgan_pack_weights16(const float* w, void* w16, void* w16t, int Co, int taps, int Ci, int dtype, void* stream) {
    ACL_REQUIRE(w && Co > 0 && taps > 0 && Ci > 0, "pack_weights16: bad arguments");
    const int64_t n = (int64_t)Co * taps * Ci;
    ACL_REQUIRE(n % 4 == 0, "pack_weights16: Co*taps*Ci must be a multiple of 4");
    if (w16) { const int rc = cast_flat16(w, w16, n, dtype, (hipStream_t)stream); if (rc) return rc; }
    if (w16t) {
        const int64_t off = 0;
        const int rc = transpose_flat16(w, w16t, &off, &Co, &taps, &Ci, 1, dtype, (hipStream_t)stream);
        if (rc) return rc;
    }
    return ACLGAN_OK;
}
size_t aclgan_conv2d_fwd16_scratch_bytes(const aclgan_conv_desc* d) { ConvGeom g; return make_geom(d, &g) ? 0 : conv_fwd16_scratch_bytes(g); }
size_t aclgan_conv2d_dgrad16_scratch_bytes(const aclgan_conv_desc* d) { ConvGeom g; return make_geom(d, &g) ? 0 : conv_dgrad16_scratch_bytes(g); }
size_t aclgan_conv2d_wgrad16_scratch_bytes(const aclgan_conv_desc* d) { ConvGeom g; return make_geom(d, &g) ? 0 : conv_wgrad16_scratch_bytes(g); }
int aclgan_conv2d_fwd16(const aclgan_conv_desc* d, int dtype, const float* x, const float* w, const void* w16, const float* bias, float* y,
                        void* scratch, void* stream) {
    ConvGeom g;
    int rc = make_geom(d, &g);
    if (rc) return rc;
    ACL_REQUIRE(x && w16 && y, "conv2d_fwd16: null buffer");
    rc = conv_fwd16(g, dtype, x, w, w16, bias, y, scratch, (hipStream_t)stream);
    if (rc == ACLGAN_EUNSUPPORTED) set_error("conv2d_fwd16: no 16-bit kernel for this shape (Cin, Cout must be multiples of 32)");
    return rc;
}
int aclgan_conv2d_fwd16_x16(const aclgan_conv_desc* d, int dtype, const void* x16, const float* w, const void* w16, const float* bias, float* y,
                            void* scratch, void* stream) {
    ConvGeom g;
    int rc = make_geom(d, &g);
    if (rc) return rc;
    ACL_REQUIRE(x16 && w16 && y, "conv2d_fwd16_x16: null buffer");
    rc = conv_fwd16(g, dtype, nullptr, w, w16, bias, y, scratch, (hipStream_t)stream, x16);
    if (rc == ACLGAN_EUNSUPPORTED) set_error("conv2d_fwd16_x16: no 16-bit kernel for this shape (Cin, Cout must be multiples of 32)");
    return rc;
}
int aclgan_conv2d_dgrad16(const aclgan_conv_desc* d, int dtype, const float* dy, const float* w, const void* w16t, float* dx, int accumulate,
                          void* scratch, void* stream) {
    ConvGeom g;
    int rc = make_geom(d, &g);
    if (rc) return rc;
    ACL_REQUIRE(dy && w16t && dx, "conv2d_dgrad16: null buffer");
    rc = conv_dgrad16(g, dtype, dy, w, w16t, dx, accumulate, scratch, (hipStream_t)stream);
    if (rc == ACLGAN_EUNSUPPORTED) set_error("conv2d_dgrad16: no 16-bit kernel for this shape (Cin, Cout must be multiples of 32)");
    return rc;
}
int aclgan_conv2d_wgrad16(const aclgan_conv_desc* d, int dtype, const float* x, const float* dy, float* dw, float* db, void* scratch, void* stream) {
    ConvGeom g;
    int rc = make_geom(d, &g);
    if (rc) return rc;
    ACL_REQUIRE(x && dy && dw, "conv2d_wgrad16: null buffer");
    rc = conv_wgrad16(g, dtype, x, dy, dw, db, scratch, (hipStream_t)stream);
    if (rc == ACLGAN_EUNSUPPORTED) set_error("conv2d_wgrad16: no 16-bit kernel for this shape (Cin, Cout must be multiples of 64)");
    return rc;
}

// ---- small dense layers, pooling, blend and loss operators (the step is built from exactly these launchers) ----
// ---- round 3: the same operators on tensors STORED in the 16-bit compute dtype (storage codes: 0 fp32, ACLGAN_DTYPE_BF16, ACLGAN_DTYPE_FP16) ----
int aclgan_conv16s_ok(const aclgan_conv_desc* d, int which) {
    ConvGeom g;
    return (d && make_geom(d, &g) == 0 && (which == 0 || which == 1) && conv16_eligible(g, which) && conv16s_ok(g, which)) ? 1 : 0;
}
int aclgan_conv2d_fwd16s(const aclgan_conv_desc* d, int dtype, const void* x16, const void* w16, const float* bias, void* y, int y_storage, void* stream) {
    ConvGeom g;
    int rc = make_geom(d, &g);
    if (rc) return rc;
    ACL_REQUIRE(y_storage == 0 || y_storage == dtype, "conv2d_fwd16s: y storage must be fp32 or the compute dtype");
    rc = conv_fwd16s(g, dtype, x16, w16, bias, y, y_storage, (hipStream_t)stream);
    if (rc == ACLGAN_EUNSUPPORTED) set_error("conv2d_fwd16s: no 16-bit-storage kernel for this shape (no upsample, Cin and Cout multiples of 64, grid >= 96 tiles)");
    return rc;
}
// status in the return value, the previous setting through `previous` (optional): an unknown key is ACLGAN_EINVAL, never a value
int aclgan_tuning(const char* key, int value, int* previous) {
    ACL_REQUIRE(key, "aclgan_tuning: null key");
    int old = 0;
    if (!strcmp(key, "glds_tile")) old = set_glds_tile(value);
    else if (!strcmp(key, "wino_x3")) old = set_wino_x3(value);
    else if (!strcmp(key, "wino_fused")) old = set_wino_fused(value);
    else if (!strcmp(key, "wino_wgrad_fused")) old = set_wino_wgrad_fused(value);
    else if (!strcmp(key, "wino_s2k4")) old = set_wino_s2k4(value);
    else if (!strcmp(key, "dgrad16s_direct")) old = set_dgrad16s_direct(value);
    else if (!strcmp(key, "fwd16_patch")) old = set_fwd16_patch(value);
    else if (!strcmp(key, "lanes")) old = set_lanes(value);
    else if (!strcmp(key, "u_batch")) old = set_u_batch(value);
    else if (!strcmp(key, "norm_mask")) old = set_norm_mask(value);
    else if (!strcmp(key, "mlp_fused")) old = set_mlp_fused(value);
    else if (!strcmp(key, "fault_at")) old = set_fault_at(value);
    else { set_error("aclgan_tuning: unknown key '%s'", key); return ACLGAN_EINVAL; }
    bump_tuning_epoch();
    if (previous) *previous = old;
    return ACLGAN_OK;
}
// read a switch without touching it (no epoch bump, no window in which another thread sees a different value); ACLGAN_EINVAL for an unknown key.
// key "epoch": the number of aclgan_tuning calls so far (what cached, switch-dependent results are keyed by: workspace sizes)
int aclgan_tuning_get(const char* key, long long* value) {
    ACL_REQUIRE(key && value, "aclgan_tuning_get: null argument");
    if (!strcmp(key, "epoch")) { *value = tuning_epoch(); return ACLGAN_OK; }
    int old = 0;
    if (!strcmp(key, "wino_fused")) old = wino_fused_mode();
    else if (!strcmp(key, "wino_wgrad_fused")) old = wino_wgrad_fused_mode();
    else if (!strcmp(key, "wino_s2k4")) old = wino_s2k4_setting();
    else if (!strcmp(key, "lanes")) old = lanes_setting();
    else if (!strcmp(key, "u_batch")) old = u_batch_setting();
    else if (!strcmp(key, "norm_mask")) old = norm_mask_setting();
    else if (!strcmp(key, "mlp_fused")) old = mlp_fused_setting();
    else if (!strcmp(key, "fault_at")) old = fault_at_setting();
    else { set_error("aclgan_tuning_get: no getter for key '%s'", key); return ACLGAN_EINVAL; }      // (the 16-bit kernel-variant switches are write-only test knobs)
    *value = old;
    return ACLGAN_OK;
}
// (round 3 form, kept: the previous value in the return value, -1 for an unknown key)
int aclgan_set_tuning(const char* key, int value) {
    int old = 0;
    return aclgan_tuning(key, value, &old) == ACLGAN_OK ? old : -1;
}
int aclgan_conv2d_fwd16s_stats_chunk(const aclgan_conv_desc* d) { ConvGeom g; return (d && make_geom(d, &g) == 0 && conv16_eligible(g, 0)) ? conv_fwd16s_stats_chunk(g) : 0; }
int aclgan_conv2d_fwd16s_stats(const aclgan_conv_desc* d, int dtype, const void* x16, const void* w16, const float* bias, void* y, int y_storage,
                               float* stats, void* stream) {
    ConvGeom g;
    int rc = make_geom(d, &g);
    if (rc) return rc;
    ACL_REQUIRE(y_storage == 0 || y_storage == dtype, "conv2d_fwd16s_stats: y storage must be fp32 or the compute dtype");
    ACL_REQUIRE(stats, "conv2d_fwd16s_stats: null statistics buffer");
    return conv_fwd16s(g, dtype, x16, w16, bias, y, y_storage, (hipStream_t)stream, stats);
}
size_t aclgan_conv2d_dgrad16s_scratch_bytes(const aclgan_conv_desc* d) { ConvGeom g; return make_geom(d, &g) ? 0 : conv_dgrad16s_scratch_bytes(g); }
int aclgan_conv2d_dgrad16s(const aclgan_conv_desc* d, int dtype, const void* dy16, const void* w16t, void* dx, int dx_storage, int accumulate,
                           void* scratch, void* stream) {
    ConvGeom g;
    int rc = make_geom(d, &g);
    if (rc) return rc;
    ACL_REQUIRE(dx_storage == 0 || dx_storage == dtype, "conv2d_dgrad16s: dx storage must be fp32 or the compute dtype");
    rc = conv_dgrad16s(g, dtype, dy16, w16t, dx, dx_storage, accumulate, scratch, (hipStream_t)stream);
    if (rc == ACLGAN_EUNSUPPORTED) set_error("conv2d_dgrad16s: no 16-bit-storage kernel for this shape (no upsample, Cin and Cout multiples of 64)");
    return rc;
}
int aclgan_conv2d_wgrad16_st(const aclgan_conv_desc* d, int dtype, const void* x, int x_storage, const void* dy, int dy_storage, float* dw, float* db,
                             void* scratch, void* stream) {
    ConvGeom g;
    int rc = make_geom(d, &g);
    if (rc) return rc;
    ACL_REQUIRE((x_storage == 0 || x_storage == dtype) && (dy_storage == 0 || dy_storage == dtype), "conv2d_wgrad16_st: storage must be fp32 or the compute dtype");
    rc = conv_wgrad16(g, dtype, (const float*)x, (const float*)dy, dw, db, scratch, (hipStream_t)stream, x_storage, dy_storage);
    if (rc == ACLGAN_EUNSUPPORTED) set_error("conv2d_wgrad16: no 16-bit kernel for this shape (Cin, Cout must be multiples of 64)");
    return rc;
}
int aclgan_cast_storage(const void* src, int src_storage, void* dst, int dst_storage, int64_t n, void* stream) {
    ACL_REQUIRE(src && dst && src_storage >= 0 && src_storage <= 2 && dst_storage >= 0 && dst_storage <= 2, "cast_storage: bad argument");
    return cast_storage(src, src_storage, dst, dst_storage, n, (hipStream_t)stream);
}
// storage[3] = {x, y, residual}
int aclgan_norm_fwd_st(int kind, int act, int B, int HW, int C, const void* x, const float* w, const float* b, int w_stride, const void* residual,
                       void* y, float* mean, float* rstd, void* scratch, const int* storage, void* stream) {
    ACL_REQUIRE(storage, "norm_fwd_st: storage codes missing");
    NormST s;
    s.x = storage[0]; s.y = storage[1]; s.res = storage[2];
    return norm_fwd(kind, act, B, HW, C, x, w, b, w_stride, residual, y, mean, rstd, scratch, (hipStream_t)stream, nullptr, 0, &s);
}
// storage[6] = {x, y, dy, dx, dres, unused}
int aclgan_norm_bwd_st(int kind, int act, int B, int HW, int C, const void* x, const void* y, const void* dy, const float* w, int w_stride,
                       const float* mean, const float* rstd, void* dx, float* dw, float* db, void* dres, int dres_accumulate, void* scratch,
                       const int* storage, void* stream) {
    ACL_REQUIRE(storage, "norm_bwd_st: storage codes missing");
    NormST s;
    s.x = storage[0]; s.y = storage[1]; s.dy = storage[2]; s.dx = storage[3]; s.dres = storage[4];
    return norm_bwd(kind, act, B, HW, C, x, y, dy, w, w_stride, mean, rstd, dx, dw, db, dres, dres_accumulate, scratch, (hipStream_t)stream, &s);
}

int aclgan_linear_fwd(int B, int I, int O, const float* x, const float* w, const float* bias, int act, float* y, void* stream) {
    ACL_REQUIRE(B > 0 && I > 0 && O > 0 && x && w && y, "linear_fwd: bad arguments");
    return linear_fwd(B, I, O, x, w, bias, act, y, (hipStream_t)stream);
}
int aclgan_mlp3_fwd(int B, int S, int M, int O, const float* s, const float* w0, const float* b0, const float* w1, const float* b1, const float* w2,
                    const float* b2, float* m0, float* m1, float* ap, void* stream) {
    ACL_REQUIRE(B > 0 && S > 0 && M > 0 && O > 0 && s && w0 && w1 && w2 && m0 && m1 && ap, "mlp3_fwd: bad arguments");
    const int rc = mlp3_fwd(B, S, M, O, s, w0, b0, w1, b1, w2, b2, m0, m1, ap, (hipStream_t)stream);
    if (rc == ACLGAN_EUNSUPPORTED) set_error("mlp3_fwd: style width %d / hidden width %d outside the fused kernel's range (S <= 64, M in {64, 128, 192, 256})", S, M);
    return rc;
}
int aclgan_linear_bwd(int B, int I, int O, const float* x, const float* y, float* dy, const float* w, int act, float* dx, float* dw, float* db,
                      void* stream) {
    ACL_REQUIRE(B > 0 && I > 0 && O > 0 && x && y && dy && w, "linear_bwd: bad arguments");
    return linear_bwd(B, I, O, x, y, dy, w, act, dx, dw, db, (hipStream_t)stream);
}
int aclgan_gap_fwd(int B, int HW, int C, const float* x, float* y, void* stream) {
    ACL_REQUIRE(B > 0 && HW > 0 && C > 0 && x && y, "gap_fwd: bad arguments");
    return gap_fwd(B, HW, C, x, y, (hipStream_t)stream);
}
int aclgan_gap_bwd(int B, int HW, int C, const float* dy, float* dx, int accumulate, void* stream) {
    ACL_REQUIRE(B > 0 && HW > 0 && C > 0 && dy && dx, "gap_bwd: bad arguments");
    return gap_bwd(B, HW, C, dy, dx, accumulate, (hipStream_t)stream);
}
int aclgan_focus_blend_fwd(int B, int HW, const float* dec4, const float* bg, float* out, const float* pair_first, float* pair, void* stream) {
    ACL_REQUIRE(B > 0 && HW > 0 && dec4 && bg && out && (!pair == !pair_first), "focus_blend_fwd: bad arguments");
    return focus_blend_fwd(B, HW, dec4, bg, out, pair_first, pair, (hipStream_t)stream);
}
int aclgan_focus_blend_bwd(int B, int HW, const float* dec4, const float* bg, const float* d_out, const float* d_pair, float* d_dec4, float* d_bg,
                           int bg_accumulate, void* stream) {
    ACL_REQUIRE(B > 0 && HW > 0 && dec4 && bg && d_dec4 && (d_out || d_pair), "focus_blend_bwd: bad arguments");
    return focus_blend_bwd(B, HW, dec4, bg, d_out, d_pair, d_dec4, d_bg, bg_accumulate, (hipStream_t)stream);
}
int aclgan_focus_translation_nchw(const float* fg, int64_t fg_bstride, const float* bg, int64_t bg_bstride, const float* focus, int64_t focus_bstride,
                                  float* out, int B, int HW, void* stream) {
    ACL_REQUIRE(B > 0 && HW > 0 && fg && bg && focus && out, "focus_translation_nchw: bad arguments");
    return focus_translation_nchw(fg, fg_bstride, bg, bg_bstride, focus, focus_bstride, out, B, HW, (hipStream_t)stream);
}
int aclgan_lsgan_loss(const float* o, int n, float target, float weight, float* loss_slot, float* d_o, float gscale, void* stream) {
    ACL_REQUIRE(o && n > 0 && loss_slot, "lsgan_loss: bad arguments");
    return lsgan_loss(o, n, target, weight, loss_slot, d_o, gscale, (hipStream_t)stream);
}
int aclgan_lsgan_loss_multi(const float* const* o, const int* n, const float* target, const float* weight, float* const* loss_slot,
                            float* const* d_o, const float* gscale, int nterms, void* stream) {
    ACL_REQUIRE(o && n && target && weight && loss_slot && gscale && nterms >= 1 && nterms <= 1024, "lsgan_loss_multi: bad arguments");
    std::vector<LsganTerm> t((size_t)nterms);
    for (int i = 0; i < nterms; ++i) {
        ACL_REQUIRE(o[i] && n[i] > 0 && loss_slot[i], "lsgan_loss_multi: bad term %d", i);
        t[i].o = o[i]; t[i].d_o = d_o ? d_o[i] : nullptr; t[i].slot = loss_slot[i]; t[i].n = n[i];
        t[i].target = target[i]; t[i].weight = weight[i]; t[i].gscale = gscale[i];
    }
    return lsgan_loss_batch(t.data(), nterms, (hipStream_t)stream);
}
int aclgan_l1_loss(const float* a, int a_channels, const float* b, int64_t npix, float* loss_slot, float* d_a, float gscale, int d_accumulate,
                   void* stream) {
    ACL_REQUIRE(a && b && npix > 0 && loss_slot && (a_channels == 3 || a_channels == 4), "l1_loss: bad arguments (a has 3 or 4 channels)");
    return l1_loss(a, a_channels, b, npix, loss_slot, d_a, gscale, d_accumulate, (hipStream_t)stream);
}
size_t aclgan_focus_loss_scratch_bytes(int64_t npix) { return npix > 0 ? (size_t)2 * focus_sums_blocks(npix) * sizeof(float) : 0; }
int aclgan_focus_loss(const float* dec4, int64_t npix, float delta, float upper, float lower, float eps, float scale, float* size_slot,
                      float* digit_slot, float* d_dec4, void* scratch, void* stream) {
    ACL_REQUIRE(dec4 && npix > 0 && size_slot && digit_slot && scratch, "focus_loss: bad arguments");
    const int rc = focus_sums(dec4, npix, eps, upper, (float*)scratch, (hipStream_t)stream);
    if (rc) return rc;
    return focus_loss_finish(dec4, npix, (const float*)scratch, delta, upper, lower, eps, scale, size_slot, digit_slot, d_dec4, (hipStream_t)stream);
}

int aclgan_focus_loss_global(const float* dec4, int64_t npix, const float* totals, int64_t npix_total, float delta, float upper, float lower,
                             float eps, float scale, float* size_slot, float* digit_slot, float* d_dec4, void* stream) {
    ACL_REQUIRE(dec4 && npix > 0 && totals && npix_total >= npix && size_slot && digit_slot, "focus_loss_global: bad arguments");
    return focus_loss_finish(dec4, npix, nullptr, delta, upper, lower, eps, scale, size_slot, digit_slot, d_dec4, (hipStream_t)stream, nullptr, totals,
                             npix_total);
}

size_t aclgan_norm_scratch_bytes(int B, int HW, int C) { return norm_scratch_bytes(B, HW, C); }
int aclgan_norm_fwd(int kind, int act, int B, int HW, int C, const float* x, const float* w, const float* b, int w_stride,
                    const float* residual, float* y, float* mean, float* rstd, void* scratch, void* stream) {
    ACL_REQUIRE(x && y && mean && rstd && scratch, "norm_fwd: null buffer");
    return norm_fwd(kind, act, B, HW, C, x, w, b, w_stride, residual, y, mean, rstd, scratch, (hipStream_t)stream);
}
int aclgan_norm_bwd(int kind, int act, int B, int HW, int C, const float* x, const float* y, const float* dy, const float* w,
                    int w_stride, const float* mean, const float* rstd, float* dx, float* dw, float* db, float* dres,
                    int dres_accumulate, void* scratch, void* stream) {
    ACL_REQUIRE(x && y && dy && mean && rstd && dx && scratch, "norm_bwd: null buffer");
    return norm_bwd(kind, act, B, HW, C, x, y, dy, w, w_stride, mean, rstd, dx, dw, db, dres, dres_accumulate, scratch, (hipStream_t)stream);
}
int aclgan_avgpool3s2_fwd(int B, int H, int W, int C, const float* x, float* y, void* stream) {
    ACL_REQUIRE(x && y, "avgpool: null buffer");
    return avgpool3s2_fwd(B, H, W, C, x, y, (hipStream_t)stream);
}
int aclgan_avgpool3s2_bwd(int B, int H, int W, int C, const float* dy, float* dx, int accumulate, void* stream) {
    ACL_REQUIRE(dy && dx, "avgpool: null buffer");
    return avgpool3s2_bwd(B, H, W, C, dy, dx, accumulate, (hipStream_t)stream);
}
int aclgan_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, const aclgan_adam* opt, int step, void* stream) {
    ACL_REQUIRE(p && g && m && v && opt, "adam: null buffer");
    return adam_flat(p, g, m, v, n, opt, step, (hipStream_t)stream);
}
int aclgan_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W, void* stream) {
    ACL_REQUIRE(src && dst, "layout: null buffer");
    return nchw_to_nhwc(src, dst, B, C, H, W, (hipStream_t)stream);
}
int aclgan_nhwc_to_nchw(const float* src, float* dst, int B, int C, int H, int W, void* stream) {
    ACL_REQUIRE(src && dst, "layout: null buffer");
    return nhwc_to_nchw(src, dst, B, C, H, W, (hipStream_t)stream);
}

}  // extern "C"
