// engine.hip -- the native step scheduler of libaclgan_hip: context, parameter layout, activation
// arena, a static backward tape, and the two update functions.
//
// What it replaces in the reference (file:line into the reference tree):
//   network construction                      trainer.py:19-23, networks.py:21-48, 112-135, 212-292
//   gen_update forward + backward             trainer.py:99-169
//   dis_update forward + backward             trainer.py:254-292
//   torch autograd                            implicit at trainer.py:169,292 -- here an explicit tape
//
// Scheduling decisions (SURVEY.md section 8a "dead work"); none of them changes a result:
//   * dis_update: the reference back-propagates through the (non-detached) generators and then
//     throws those gradients away (trainer.py:91 zero_grad); here the generator pass of
//     dis_update records no tape at all.  encode(x_b) (trainer.py:260, unused), all style encoders
//     and the duplicate dis_A(x_a) forward (trainer.py:283-284) are not executed; the real branch
//     of loss_dis_A enters once with weight 2*0.5.
//   * gen_update: the three discriminators get dgrad only (their weight gradients would be
//     discarded by dis_opt.zero_grad, trainer.py:248); the two style-encoder passes whose output
//     is dropped (trainer.py:103,125) are skipped.
#include "common.h"
#include <dlfcn.h>

#include <functional>
#include <map>
#include <string>
#include <vector>
#include <tuple>
#include <mutex>
#include <algorithm>
#include <cstring>
#include <cstdio>
#include <cstdlib>

namespace aclgan {

struct TensorInfo {
    std::string name;   // "<net>/<reference key>"
    int64_t offset;     // floats into the group's flat buffer
    int64_t numel;
    int shape[4];       // reference (OIHW-order) dims
    int ndim;
};

struct Group {
    std::vector<TensorInfo> tensors;
    std::map<std::string, int> index;
    int64_t numel = 0;
    float *param = nullptr, *grad = nullptr, *m = nullptr, *v = nullptr;
};

// activation tensor, NHWC.  Storage (st16.h): dt = dtype of d, gdt = dtype of g -- 0: fp32; else the context's 16-bit compute dtype,
// in which case d / g address 16-bit data (the float* type is kept for the fp32 kernels' signatures).  g always OWNS 4 bytes per
// element: its dtype is settled only when the forward is complete (a consumer whose backward writes fp32 clears gdt).
struct Act {
    float* d = nullptr;
    float* g = nullptr;
    int dt = 0, gdt = 0;
    int B = 0, H = 0, W = 0, C = 0;
    bool need_grad = false;
    bool gw = false;   // gradient buffer holds valid data (first writer overwrites, later ones accumulate)
    Act* parent = nullptr;   // batch-slice view of a joint tensor: its gradient arrives through the parent
    // lane stamps (aclgan_ctx lanes): which lane last wrote d / touched g, and the index of that lane's checkpoint that covers the
    // write.  Views stamp their joint tensor (the halves of a joint discriminator batch are written on different lanes).
    struct Stamp { int lane, ck; };
    Stamp dst[4]; int ndst = 0;
    Stamp gst = {-1, 0};
    Act* root() { return parent ? parent : this; }
    int64_t numel() const { return (int64_t)B * H * W * C; }
    bool written() const { return gw || (parent && parent->gw); }
};

struct PW {   // a (weight, bias) pair inside a flat group
    const float* w = nullptr; const float* b = nullptr;
    float* dw = nullptr; float* db = nullptr;
    int64_t nw = 0, nb = 0;   // element counts (gradient-bucket bookkeeping)
    const unsigned short* w16 = nullptr;    // 16-bit packs of w (same layout) and its [tap][cin][cout] transpose
    const unsigned short* w16t = nullptr;
};

// one backward closure + the ranges of the trained group's flat gradient buffer it writes
struct TapeOp {
    std::function<int()> fn;
    int64_t goff[4] = {0, 0, 0, 0}, gnum[4] = {0, 0, 0, 0};
    int ng = 0;
    int lane = 0;      // the lane (stream) its kernels are enqueued on = the lane of the forward pass that pushed it
    int pass = -1;     // forward pass it belongs to: a lane checkpoint is taken whenever the replay moves to another pass
};

}  // namespace aclgan

using namespace aclgan;

// ---- roctx ranges per network pass (tracing): ACLGAN_ROCTX=1 wraps every forward pass of a network ("gen_AB.encode#1",
// "gen_BA.decode#2", "dis_2.forward#1", ...) and the backward closures it recorded ("bwd:gen_BA.decode#2") in roctx ranges, so that a
// rocprofv3 --marker-trace --kernel-trace run can be cut by pass.  The marker library is opened lazily (no link-time dependency).
namespace aclgan {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    bool on = false;
    Roctx() {
        const char* env = getenv("ACLGAN_ROCTX");
        if (!env || !atoi(env)) return;
        for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
            pop = (int (*)())dlsym(h, "roctxRangePop");
            if (push && pop) { on = true; return; }
        }
    }
    static Roctx& get() { static Roctx r; return r; }
};
}  // namespace aclgan

// Streams of the lanes and the parameter-gradient stream: ONE set per device for the whole process, shared by every context (created on first
// use, never destroyed).  HIP multiplexes streams onto a few hardware queues in creation order (GPU_MAX_HW_QUEUES, default 4): with a set per
// context, the second trainer of a process (bench.py's small-batch probe, a test suite) got streams that SHARE a hardware queue with the
// caller's stream, and its lanes serialised -- measured as a 5 ms spread of the same B=3 step between otherwise identical runs
// (profiles/r05_experiments.md section 2).  Contexts are used one update at a time per thread; two threads stepping two contexts at once
// would interleave their work on these streams (ordered by their own events: still correct).
namespace aclgan {
struct StreamPool {
    static const int N = 4;      // [0] = parameter-gradient (side) stream, [1..3] = lanes 1..3
    hipStream_t s[N] = {nullptr, nullptr, nullptr, nullptr};
    static StreamPool& of_device() {
        static StreamPool pools[16];
        int dev = 0;
        (void)hipGetDevice(&dev);
        return pools[(dev >= 0 && dev < 16) ? dev : 0];
    }
    int get(int i, hipStream_t* out) {
        static std::mutex mu;
        std::lock_guard<std::mutex> lock(mu);
        if (!s[i]) {
            hipError_t e = hipSuccess;
            static int prio = -1;      // ACLGAN_SIDE_PRIO=1: the parameter-gradient stream at the highest priority the device offers (measured neutral, round 4)
            if (prio < 0) { const char* pe = getenv("ACLGAN_SIDE_PRIO"); prio = pe ? atoi(pe) : 0; }
            // ACLGAN_LANE_PRIO=-1: lanes 1.. at the LOWEST priority (lane 0, the caller's stream, carries the chain everything waits for)
            static int lprio = -2;
            if (lprio == -2) { const char* pe = getenv("ACLGAN_LANE_PRIO"); lprio = pe ? atoi(pe) : 0; }
            int lo = 0, hi = 0;
            if (i == 0 && prio && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess) e = hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, prio > 0 ? hi : lo);
            else if (i > 0 && lprio && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess) e = hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, lprio > 0 ? hi : lo);
            else e = hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
            if (e != hipSuccess) { s[i] = nullptr; return hip_fail(e, "stream pool"); }
        }
        *out = s[i];
        return ACLGAN_OK;
    }
};
}  // namespace aclgan

struct aclgan_ctx {
    aclgan_arch arch;
    // roctx pass labels: index into pass_names of the pass being built (-1: none); every tape closure remembers the pass that pushed it
    std::vector<std::string> pass_names;
    std::vector<int> tape_pass;
    std::map<std::string, int> pass_count;
    int cur_pass = -1;
    int pass_begin(const char* net, const char* what) {
        if (!Roctx::get().on || dry) return -1;
        const std::string base = std::string(net) + "." + what;
        const int n = ++pass_count[base];
        pass_names.push_back(base + "#" + std::to_string(n));
        const int prev = cur_pass;
        cur_pass = (int)pass_names.size() - 1;
        // "@N": the library's launch counter when the range opens -- kernels are asynchronous, so a trace is cut by launch ORDER, not by
        // host time (scripts/rocpd_bypass.py): the range owns the library launches N .. (N of its "~end" marker) - 1
        Roctx::get().push((pass_names.back() + "@" + std::to_string((long long)g_launches)).c_str());
        return prev;
    }
    void pass_end(int prev) {
        if (!Roctx::get().on || dry) return;
        Roctx::get().pop();
        Roctx::get().push(("~end@" + std::to_string((long long)g_launches)).c_str());
        Roctx::get().pop();
        cur_pass = prev;
    }
    Group groups[2];
    char* ws = nullptr;
    size_t ws_bytes = 0;
    // per-step state
    hipStream_t st = nullptr;
    bool dry = false;
    size_t top = 0, peak = 0;
    std::vector<Act*> acts;
    std::vector<TapeOp> tape;
    // reduced-precision compute (aclgan_set_compute_dtype / aclgan_bind_params16 / aclgan_bind_loss_scale)
    int dtype = ACLGAN_DTYPE_FP32;
    unsigned short* w16[2] = {nullptr, nullptr};
    unsigned short* w16t[2] = {nullptr, nullptr};
    float* lscale = nullptr;
    std::vector<int64_t> cv_off[2];          // conv tensors of each group: offset / Cout / taps / Cin (transposed pack)
    std::vector<int> cv_co[2], cv_taps[2], cv_ci[2];
    int trained = -1;          // group whose gradients this step produces (-1: forward only)
    bool fire_dry = false;     // aclgan_bucket_schedule: invoke the bucket callback during a dry run
    // data-parallel gradient buckets (aclgan_set_grad_buckets / aclgan_set_bucket_callback)
    aclgan_sync_fn sync_fn = nullptr;   // forward sync point of gen_update (global-batch focus sums), aclgan_set_forward_sync
    void* sync_user = nullptr;
    int sync_world = 1;
    int64_t bucket_elems = 0;
    aclgan_bucket_fn bucket_fn = nullptr;
    void* bucket_user = nullptr;
    std::vector<int> bucket_order;   // buckets in the order they completed during the last update
    // workspace need of an update by (update, shape, dtype, switch setting): step_common checks the bound workspace against it
    struct NeedKey {
        int which, B, H, W, dtype; long long epoch; int determ, buckets;
        bool operator<(const NeedKey& o) const {
            return std::tie(which, B, H, W, dtype, epoch, determ, buckets) < std::tie(o.which, o.B, o.H, o.W, o.dtype, o.epoch, o.determ, o.buckets);
        }
    };
    std::map<NeedKey, size_t> need_cache;
    // ALGORITHMIC HBM bytes of the step being built (aclgan_step_algorithmic_bytes): every operator's inputs read once and
    // outputs written once at their storage width -- what a perfectly fused-per-operator implementation must move
    double alg_bytes = 0.0;
    void count(double bytes) { alg_bytes += bytes; }
    // matrix-pipe FLOPs the step being built EXECUTES (aclgan_step_executed_flops): every convolution at the cost of the path its launchers
    // choose (direct / Winograd / sub-pixel / parity phases), dense layers at 2 B in out
    double exec_flops = 0.0;
    size_t keep_total = 0;      // bytes of Winograd input transforms kept for the weight gradients of this update (conv_block)
    // diagnostics (aclgan_debug_capture_masks): the ReLU / LeakyReLU masks of every Conv2dBlock an update back-propagates through, one byte per
    // element, appended to a caller-owned device buffer in the order the forward builds them (tests/test_gpu_maskfrozen.py)
    struct MaskEnt { int B, H, W, C, act; long long off; };
    unsigned char* mask_dst = nullptr; size_t mask_cap = 0, mask_top = 0;
    std::vector<MaskEnt> mask_log;
    // Winograd filter transforms of this update, by (filter tensor, variant): computed by the first layer call that needs one, reused by
    // the later calls of the same network (common.h: WinoUCache); the buffers live in the arena until the update ends
    struct UEnt { float* u; bool filled; int layout; int lane, ck; };
    std::map<std::pair<const float*, int>, UEnt> ucache;
    WinoUCache ucache_hook;
    static bool ucache_enabled() { static int v = -1; if (v < 0) { const char* e = getenv("ACLGAN_NOUCACHE"); v = (e && atoi(e)) ? 0 : 1; } return v == 1; }
    // variant = (0 forward | 1 input gradient | 2 / 3 merged sub-pixel phase filters) | layout << 4 (conv_wino.hip).  An entry remembers the
    // layout it was filled in and the lane that filled it: another layout gets no entry (the caller computes into its scratch -- never a
    // transform read in the wrong order), another lane waits for the filling lane's checkpoint.
    static float* ucache_lookup(void* user, const float* w, int variant, size_t bytes, bool* fresh) {
        aclgan_ctx* c = (aclgan_ctx*)user;
        const int layout = variant >> 4;
        auto it = c->ucache.find(std::make_pair(w, variant & 15));
        if (it == c->ucache.end()) return nullptr;       // not reserved (operator-level call paths): the caller uses its scratch slice
        UEnt& e = it->second;
        if (!e.filled) {
            *fresh = true;
            if (bytes) { e.filled = true; e.layout = layout; e.lane = c->cur_lane; e.ck = c->nck(c->cur_lane); }      // (bytes == 0: a peek)
            return e.u;
        }
        if (e.layout != layout) return nullptr;
        *fresh = false;
        // (a failed cross-lane wait must not hand out a transform another lane may still be writing: the caller recomputes into its own scratch)
        if (bytes && c->nlanes > 1 && e.lane != c->cur_lane && c->wait_ck(c->cur_lane, e.lane, e.ck) != ACLGAN_OK) return nullptr;
        return e.u;
    }
    // make sure the arena holds a slot for the transform of (w, variant); call where an allocation may persist until the update ends
    int ucache_reserve(const float* w, int variant, size_t bytes) {
        if (!ucache_enabled() || !bytes || ucache.count(std::make_pair(w, variant))) return ACLGAN_OK;
        float* u = (float*)alloc(bytes);
        if (!u) return ACLGAN_ENOMEM;
        ucache[std::make_pair(w, variant)] = UEnt{u, false, 0, 0, 0};
        return ACLGAN_OK;
    }
    // 16-bit activation / gradient storage of the wide layers (C % 64 == 0) under a 16-bit compute dtype; co16: also the conv outputs
    // that feed a normalisation layer (ACLGAN_ACT16=0 / ACLGAN_CO16=0|1 switch them)
    bool act16() const { return dtype != ACLGAN_DTYPE_FP32 && act16_enabled(); }
    static bool act16_enabled() { static int v = -1; if (v < 0) { const char* e = getenv("ACLGAN_ACT16"); v = (e && !atoi(e)) ? 0 : 1; } return v == 1; }
    static bool co16_enabled() { static int v = -1; if (v < 0) { const char* e = getenv("ACLGAN_CO16"); v = e ? (atoi(e) ? 1 : 0) : 1; } return v == 1; }

    // Side stream of the backward (round 3): the weight gradient of a layer depends only on tensors that stay put until the update
    // ends (x, dy, the kept Winograd transform) and nothing in the backward waits for it, while the input gradient is on the critical
    // path -- so the weight-gradient pipeline of layer L (transforms: HBM-bound, GEMM: MFMA-bound, finishes: latency-bound) runs on a
    // second stream next to the input-gradient pipeline of L and the norm backward of L-1, filling each other's tails and dispatch
    // gaps.  Fork = an event on the closure's stream after dy is complete; join = before a gradient bucket is handed to the all-reduce
    // and at the end of the tape.  Its scratch is a second stack growing down from the END of the workspace (stream-ordered reuse).
    // ACLGAN_SIDE_STREAM=0 turns it off (everything on the caller's stream, as in round 2).
    // Round 5: it is THE parameter-gradient stream -- every kernel that accumulates into the trained group's flat gradient buffer
    // (convolution weight / bias gradients, LayerNorm gamma / beta, the dense layers' dw / db) runs on it, in tape order, whatever lane
    // the rest of its closure runs on: two lanes never add into the same parameter gradient concurrently, the sums keep one fixed order.
    hipStream_t st2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool side_pending = false;
    size_t top2 = 0, peak2 = 0;
    static bool side_enabled() { static int v = -1; if (v < 0) { const char* e = getenv("ACLGAN_SIDE_STREAM"); v = (e && !atoi(e)) ? 0 : 1; } return v == 1; }
    void* alloc2(size_t bytes) {
        const size_t need = top2 + ((bytes + 255) & ~(size_t)255);
        top2 = need;
        if (need > peak2) peak2 = need;
        if (dry) return (void*)(uintptr_t)4096;
        // the main stack's high-water mark of this step, not its current top: scratch the host has released may still be in use by
        // kernels in flight on a lane
        if (peak + need + 256 > ws_bytes) return nullptr;
        return ws + ((ws_bytes - need) & ~(size_t)255);
    }
    int side_fork() {       // the side stream may start once everything enqueued on the current lane so far is done
        if (dry) return ACLGAN_OK;
        if (!st2) {      // (entry points that do not go through lanes_begin)
            int rc = aclgan::StreamPool::of_device().get(0, &st2_pool);
            if (rc) return rc;
            st2 = st2_pool;
        }
        if (!ev_fork) {
            hipError_t e = hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_join, hipEventDisableTiming);
            if (e != hipSuccess) return aclgan::hip_fail(e, "side stream events");
        }
        hipError_t e = hipEventRecord(ev_fork, st);
        if (e == hipSuccess) e = hipStreamWaitEvent(st2, ev_fork, 0);
        if (e != hipSuccess) return aclgan::hip_fail(e, "side stream fork");
        side_pending = true;
        return ACLGAN_OK;
    }
    int side_join() {       // lane 0 (the caller's stream) waits for everything the side stream was given
        if (dry || !side_pending) return ACLGAN_OK;
        hipError_t e = hipEventRecord(ev_join, st2);
        if (e == hipSuccess) e = hipStreamWaitEvent(st0 ? st0 : st, ev_join, 0);
        if (e != hipSuccess) return aclgan::hip_fail(e, "side stream join");
        side_pending = false;
        return ACLGAN_OK;
    }

    // ---- Lanes (round 5): independent branches of the step on separate HIP streams.
    // The reference's step is a graph with wide independent branches -- the two translation directions (trainer.py:103-116), the
    // reconstruction decodes (trainer.py:113-114), three discriminators with three scales each (trainer.py:136-139, networks.py:50-57) --
    // which a single in-order queue serialises, so that every launch that does not fill the chip (late discriminator layers, style
    // encoder tails, split-K finishes, statistics finalizes, MLP layers) costs its full latency.  A lane is a stream; lane 0 is the
    // caller's.  The update functions assign every network pass to a lane (set_lane), every backward closure runs on the lane of the
    // pass that recorded it, and the host still builds / replays everything in ONE fixed order, so the accumulate-or-overwrite flags,
    // the order of every sum and the bucket schedule are exactly those of the single-queue plan: results do not depend on the number
    // of lanes (tests/test_gpu_determinism.py compares them bitwise).
    //   Dependencies: a lane takes a CHECKPOINT (an event) whenever the host leaves it and at every pass boundary; an activation
    //   carries a stamp (lane, checkpoint index) per writer of its data and one for the last toucher of its gradient; a consumer on
    //   another lane waits for exactly that checkpoint (need / acq).  Parameter gradients never cross lanes: they live on the side stream.
    //   Memory: the arena stays one stack.  Scratch released by the host may still be in use on its lane, so an allocation made on
    //   lane L starts above the high-water mark of every OTHER lane (hw); same-lane reuse is stream-ordered as before.  The marks
    //   reset where all lanes meet (lanes_barrier).  The dry run follows the same rule, so aclgan_workspace_bytes stays exact.
    static const int MAXL = 4;
    hipStream_t st0 = nullptr;                       // the caller's stream = lane 0 (st = the current lane's stream)
    hipStream_t lane_st[MAXL] = {nullptr, nullptr, nullptr, nullptr};
    int nlanes = 1, cur_lane = 0;
    std::vector<hipEvent_t> ev_pool; size_t ev_next = 0;
    std::vector<hipEvent_t> lane_evs[MAXL];          // lane_evs[l][k]: the k-th checkpoint of lane l in this step
    int seen[MAXL][MAXL] = {};                       // seen[d][s]: lane d has waited for the first seen[d][s] checkpoints of lane s
    size_t hw[MAXL] = {0, 0, 0, 0};
    bool lane_dirty[MAXL] = {false, false, false, false};   // something (work or a wait) was enqueued on the lane since its last checkpoint
    int pass_seq = 0, cur_pass_id = -1;              // forward pass ids (always on; the roctx labels are separate)
    hipStream_t lane_stream(int l) const { return l == 0 ? st0 : lane_st[l]; }
    int nck(int l) const { return (int)lane_evs[l].size(); }
    // lanes of this step; streams are created on first use and live as long as the context
    hipStream_t st2_pool = nullptr, st2_private = nullptr;
    // The private stream exists only for contexts that capture (aclgan_ctx_enable_capture creates it outside any capture; a capture that
    // comes without that call creates it on the spot): an extra stream, even an idle one, shifts HIP's stream -> hardware-queue placement --
    // creating it for every context cost the eager step 3 ms (89.1 against 86.2 ms, profiles/r05_experiments.md section 2).
    int make_private_side() {
        if (st2_private) return ACLGAN_OK;
        hipError_t e = hipStreamCreateWithFlags(&st2_private, hipStreamNonBlocking);
        if (e != hipSuccess) return aclgan::hip_fail(e, "private side stream");
        return ACLGAN_OK;
    }
    int lanes_begin(int want) {
        nlanes = std::max(1, std::min(want, (int)MAXL));
        if (!side_enabled()) nlanes = 1;             // parameter gradients need their own ordered stream once there is more than one lane
        // Under stream capture (aclgan_Trainer(hip_graph=True): torch.cuda.graph around the update) the update runs as in round 4: one lane
        // and a parameter-gradient stream PRIVATE to this context.  Capturing the lanes of the process-wide pool crashed hipStreamEndCapture
        // on this ROCm (round 5, tests/test_gpu_graph.py: gen_update with three pooled lanes); a replayed graph gained nothing from a second
        // queue in any regime measured, so the capture keeps the plan that is known to work.
        bool capturing = false;
        if (!dry) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) capturing = true;
            else (void)hipGetLastError();
        }
        // ACLGAN_CAPTURE_LANES=1 (investigation switch, round 6): capture the pooled lanes as they are
        static int cap_lanes = -1;
        if (cap_lanes < 0) { const char* e = getenv("ACLGAN_CAPTURE_LANES"); cap_lanes = (e && atoi(e)) ? 1 : 0; }
        const bool cap_one = capturing && !cap_lanes;
        if (cap_one) nlanes = 1;
        cur_lane = 0; st0 = st;
        for (int l = 0; l < MAXL; ++l) { lane_evs[l].clear(); hw[l] = 0; lane_dirty[l] = false; for (int m = 0; m < MAXL; ++m) seen[l][m] = 0; }
        ev_next = 0;
        if (dry) return ACLGAN_OK;
        // (parameter-gradient stream first, then the lanes: with the caller's stream that is one hardware queue each up to 3 lanes)
        if (side_enabled() && !st2_pool) { int rc = aclgan::StreamPool::of_device().get(0, &st2_pool); if (rc) return rc; }
        if (side_enabled() && cap_one) { int rc = make_private_side(); if (rc) return rc; }
        st2 = cap_one ? st2_private : st2_pool;
        for (int l = 1; l < nlanes; ++l)
            if (!lane_st[l]) { int rc = aclgan::StreamPool::of_device().get(l, &lane_st[l]); if (rc) return rc; }
        return ACLGAN_OK;
    }
    // checkpoint of lane l: everything enqueued on it so far
    int checkpoint(int l) {
        hipEvent_t ev = nullptr;
        if (!dry) {
            if (ev_next == ev_pool.size()) {
                hipEvent_t e = nullptr;
                hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
                if (rc != hipSuccess) return aclgan::hip_fail(rc, "lane event");
                ev_pool.push_back(e);
            }
            ev = ev_pool[ev_next++];
            hipError_t rc = hipEventRecord(ev, lane_stream(l));
            if (rc != hipSuccess) return aclgan::hip_fail(rc, "lane checkpoint");
        }
        lane_evs[l].push_back(ev);
        lane_dirty[l] = false;
        return ACLGAN_OK;
    }
    // pass boundary on the current lane (stamps taken before it are covered by this checkpoint)
    int mark() { return nlanes > 1 ? checkpoint(cur_lane) : ACLGAN_OK; }
    int set_lane(int l) {
        if (nlanes <= 1) return ACLGAN_OK;
        l %= nlanes;
        if (l == cur_lane) return ACLGAN_OK;
        int rc = checkpoint(cur_lane);
        if (rc) return rc;
        cur_lane = l;
        lane_dirty[l] = true;
        if (!dry) st = lane_stream(l);
        return ACLGAN_OK;
    }
    // lane d waits for checkpoint k of lane s
    int wait_ck(int d, int s, int k) {
        if (d == s || k < seen[d][s]) return ACLGAN_OK;
        if (k >= nck(s)) {                            // not checkpointed yet (s is the current lane and d is not: join paths only)
            int rc = checkpoint(s);
            if (rc) return rc;
            k = nck(s) - 1;
        }
        if (!dry) {
            hipError_t e = hipStreamWaitEvent(lane_stream(d), lane_evs[s][k], 0);
            if (e != hipSuccess) return aclgan::hip_fail(e, "lane wait");
        }
        lane_dirty[d] = true;
        seen[d][s] = k + 1;
        return ACLGAN_OK;
    }
    // the current lane is about to READ a's data
    int need(Act* a) {
        if (nlanes <= 1 || !a) return ACLGAN_OK;
        Act* r = a->root();
        for (int i = 0; i < r->ndst; ++i)
            if (r->dst[i].lane != cur_lane) { int rc = wait_ck(cur_lane, r->dst[i].lane, r->dst[i].ck); if (rc) return rc; }
        return ACLGAN_OK;
    }
    // the current lane has enqueued a writer of a's data
    void wrote(Act* a) {
        if (!a) return;
        Act* r = a->root();
        for (int i = 0; i < r->ndst; ++i)
            if (r->dst[i].lane == cur_lane) { r->dst[i].ck = nck(cur_lane); return; }
        if (r->ndst < 4) r->dst[r->ndst++] = Act::Stamp{cur_lane, nck(cur_lane)};
    }
    // the current lane is about to read or write a's GRADIENT (backward): order it after the previous toucher
    int acq(Act* a) {
        if (nlanes <= 1 || !a) return ACLGAN_OK;
        Act* r = a->root();
        if (r->gst.lane >= 0 && r->gst.lane != cur_lane) { int rc = wait_ck(cur_lane, r->gst.lane, r->gst.ck); if (rc) return rc; }
        r->gst = Act::Stamp{cur_lane, nck(cur_lane)};
        return ACLGAN_OK;
    }
    // lane 0 waits for everything enqueued on every lane and on the side stream; the current lane becomes lane 0
    int lanes_join() {
        if (nlanes > 1) {
            int rc = set_lane(0);
            if (rc) return rc;
            // (a lane that only ever received a wait -- the barrier's -- still has to come back: under stream capture every stream that
            //  joined the capture must be joined again before it ends)
            for (int l = 1; l < nlanes; ++l) {
                if (lane_dirty[l]) { rc = checkpoint(l); if (rc) return rc; }
                if (nck(l) > 0) { rc = wait_ck(0, l, nck(l) - 1); if (rc) return rc; }
            }
        }
        return side_join();
    }
    // every lane waits for everything enqueued so far anywhere: scratch of all lanes may be reused from here on
    int lanes_barrier() {
        int rc = lanes_join();
        if (rc || nlanes <= 1) return rc;
        rc = checkpoint(0);
        if (rc) return rc;
        for (int l = 1; l < nlanes; ++l) { rc = wait_ck(l, 0, nck(0) - 1); if (rc) return rc; }
        for (int l = 0; l < MAXL; ++l) hw[l] = 0;
        return ACLGAN_OK;
    }
    // error paths: nothing of this step may still be in flight on a stream the caller does not know about when control returns
    void lanes_quiesce() {
        if (dry) return;
        for (int l = 1; l < nlanes; ++l) if (lane_st[l]) (void)hipStreamSynchronize(lane_st[l]);
        if (st2) (void)hipStreamSynchronize(st2);
        side_pending = false;
        cur_lane = 0; if (st0) st = st0;
    }
    ~aclgan_ctx() {
        reset_step();
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
        if (st2_private) (void)hipStreamDestroy(st2_private);
        // (the other streams belong to the process-wide pool)
    }
    void reset_step() {
        for (Act* a : acts) delete a;
        acts.clear();
        tape.clear();
        tape_pass.clear(); pass_names.clear(); pass_count.clear(); cur_pass = -1;
        top = 0;
        top2 = 0;
        keep_total = 0;
        ucache.clear();
        for (int l = 0; l < MAXL; ++l) { lane_evs[l].clear(); hw[l] = 0; lane_dirty[l] = false; for (int m = 0; m < MAXL; ++m) seen[l][m] = 0; }
        ev_next = 0; cur_lane = 0; nlanes = 1; pass_seq = 0; cur_pass_id = -1;
        if (st0) { st = st0; st0 = nullptr; }
    }
    void* alloc(size_t bytes) {
        size_t a = (top + 255) & ~(size_t)255;
        if (nlanes > 1) for (int l = 0; l < nlanes; ++l) if (l != cur_lane && hw[l] > a) a = hw[l];
        top = a + bytes;
        if (nlanes > 1) { const size_t e = (top + 255) & ~(size_t)255; if (e > hw[cur_lane]) hw[cur_lane] = e; }
        if (top > peak) peak = top;
        if (dry) return (void*)(uintptr_t)(a + 4096);   // fake, never dereferenced
        // the side stream's stack grows down from the end of the workspace and its kernels may still be running: the main stack
        // must stay below the deepest point that stack has reached in this step (an undersized workspace fails, it does not corrupt)
        if (top + peak2 + 256 > ws_bytes) return nullptr;
        return ws + a;
    }
    float* allocf(int64_t n) { return (float*)alloc((size_t)n * sizeof(float)); }
    // st != 0: d (and, until a consumer objects, g) stored in the 16-bit compute dtype
    Act* new_act(int B, int H, int W, int C, bool need_grad, int st = 0) {
        Act* a = new Act();
        a->B = B; a->H = H; a->W = W; a->C = C; a->need_grad = need_grad; a->dt = st; a->gdt = st;
        a->d = (float*)alloc((size_t)a->numel() * (st ? 2 : 4));
        if (need_grad) a->g = allocf(a->numel());
        acts.push_back(a);
        wrote(a);
        return a;
    }
    // view of `nb` samples starting at sample b0 of a joint activation (no allocation)
    Act* new_view(Act* joint, int b0, int nb) {
        Act* a = new Act();
        a->B = nb; a->H = joint->H; a->W = joint->W; a->C = joint->C; a->need_grad = joint->need_grad;
        // (views exist only of the fp32 joint discriminator inputs)
        const int64_t off = (int64_t)b0 * joint->H * joint->W * joint->C;
        a->d = joint->d + off;
        a->g = joint->g ? joint->g + off : nullptr;
        a->parent = joint;
        acts.push_back(a);
        return a;
    }
    // append a backward closure; `grads` = pointers into the trained group's gradient buffer it accumulates into
    void push(std::function<int()> fn, std::initializer_list<std::pair<const float*, int64_t>> grads = {}) {
        TapeOp op;
        op.fn = std::move(fn);
        if (trained >= 0 && groups[trained].grad)
            for (const auto& g : grads)
                if (g.first && g.second > 0 && op.ng < 4) { op.goff[op.ng] = g.first - groups[trained].grad; op.gnum[op.ng] = g.second; ++op.ng; }
        op.lane = cur_lane; op.pass = cur_pass_id;
        tape.push_back(std::move(op));
        tape_pass.push_back(cur_pass);
    }
    PW pw(int group, int net, const std::string& key, bool with_bias = true) const;
    int64_t numel_of(int group, int net, const std::string& key) const;
    const float* param(int group, int net, const std::string& key) const;
    float* gradp(int group, int net, const std::string& key) const;
};

namespace aclgan {


#define RUN(expr)                                  \
    do {                                           \
        if (!c.dry) { int rc__ = (expr); if (rc__) return rc__; } \
    } while (0)
#define CHK(expr)                                  \
    do { int rc__ = (expr); if (rc__) return rc__; } while (0)
#define NEED(ptr)                                  \
    do { if (!(ptr)) { set_error("workspace too small (need > %zu bytes, bound %zu)", c.top, c.ws_bytes); return ACLGAN_ENOMEM; } } while (0)

// ------------------------------------------------------------------------------------------
// parameter layout (reference parameters() order; SURVEY.md 2.3 / tests/golden/param_order.json)
// ------------------------------------------------------------------------------------------
static void add_tensor(Group& g, const std::string& name, int ndim, int d0, int d1 = 1, int d2 = 1, int d3 = 1) {
    TensorInfo t;
    t.name = name; t.offset = g.numel; t.ndim = ndim;
    t.shape[0] = d0; t.shape[1] = d1; t.shape[2] = d2; t.shape[3] = d3;
    t.numel = (int64_t)d0 * d1 * d2 * d3;
    // keep every tensor 16-byte aligned inside the flat buffer (float4 loads of OHWI rows)
    g.index[name] = (int)g.tensors.size();
    g.tensors.push_back(t);
    g.numel += (t.numel + 3) / 4 * 4;
}
static void add_conv(Group& g, const std::string& prefix, int co, int ci, int k) {
    add_tensor(g, prefix + ".weight", 4, co, ci, k, k);
    add_tensor(g, prefix + ".bias", 1, co);
}

static void build_gen(Group& g, const std::string& net, const aclgan_arch& a) {
    char buf[160];
    const int nd = a.gen_n_downsample, nr = a.gen_n_res, sd = a.gen_style_dim;
    int d = a.gen_dim;
    // StyleEncoder (networks.py:212-228; 4 downsamples hard-coded at networks.py:126)
    add_conv(g, net + "/enc_style.model.0.conv", d, a.input_dim_a, 7);
    for (int i = 0; i < 2; ++i) { snprintf(buf, sizeof buf, "/enc_style.model.%d.conv", 1 + i); add_conv(g, net + buf, 2 * d, d, 4); d *= 2; }
    for (int i = 0; i < 2; ++i) { snprintf(buf, sizeof buf, "/enc_style.model.%d.conv", 3 + i); add_conv(g, net + buf, d, d, 4); }
    add_conv(g, net + "/enc_style.model.6", sd, d, 1);
    // ContentEncoder (networks.py:230-245)
    d = a.gen_dim;
    add_conv(g, net + "/enc_content.model.0.conv", d, a.input_dim_a, 7);
    for (int i = 0; i < nd; ++i) { snprintf(buf, sizeof buf, "/enc_content.model.%d.conv", 1 + i); add_conv(g, net + buf, 2 * d, d, 4); d *= 2; }
    for (int r = 0; r < nr; ++r)
        for (int j = 0; j < 2; ++j) { snprintf(buf, sizeof buf, "/enc_content.model.%d.model.%d.model.%d.conv", 1 + nd, r, j); add_conv(g, net + buf, d, d, 3); }
    // Decoder (networks.py:247-264)
    for (int r = 0; r < nr; ++r)
        for (int j = 0; j < 2; ++j) { snprintf(buf, sizeof buf, "/dec.model.0.model.%d.model.%d.conv", r, j); add_conv(g, net + buf, d, d, 3); }
    int idx = 2;
    for (int i = 0; i < nd; ++i) {
        snprintf(buf, sizeof buf, "/dec.model.%d.norm.gamma", idx); add_tensor(g, net + buf, 1, d / 2);
        snprintf(buf, sizeof buf, "/dec.model.%d.norm.beta", idx); add_tensor(g, net + buf, 1, d / 2);
        snprintf(buf, sizeof buf, "/dec.model.%d.conv", idx); add_conv(g, net + buf, d / 2, d, 5);
        d /= 2; idx += 2;
    }
    snprintf(buf, sizeof buf, "/dec.model.%d.conv", idx - 1); add_conv(g, net + buf, a.gen_output_dim, d, 7);
    // MLP (networks.py:280-292); num_adain = 2*C per AdaIN layer, 2*n_res layers (networks.py:165-171)
    const int C = a.gen_dim << nd, nad = 2 * C * 2 * nr, md = a.gen_mlp_dim;
    add_tensor(g, net + "/mlp.model.0.fc.weight", 2, md, sd); add_tensor(g, net + "/mlp.model.0.fc.bias", 1, md);
    add_tensor(g, net + "/mlp.model.1.fc.weight", 2, md, md); add_tensor(g, net + "/mlp.model.1.fc.bias", 1, md);
    add_tensor(g, net + "/mlp.model.2.fc.weight", 2, nad, md); add_tensor(g, net + "/mlp.model.2.fc.bias", 1, nad);
}

static void build_dis(Group& g, const std::string& net, int input_dim, const aclgan_arch& a) {
    char buf[160];
    for (int s = 0; s < a.dis_num_scales; ++s) {
        int d = a.dis_dim;
        snprintf(buf, sizeof buf, "/cnns.%d.0.conv", s); add_conv(g, net + buf, d, input_dim, 4);
        for (int i = 0; i < a.dis_n_layer - 1; ++i) { snprintf(buf, sizeof buf, "/cnns.%d.%d.conv", s, i + 1); add_conv(g, net + buf, 2 * d, d, 4); d *= 2; }
        snprintf(buf, sizeof buf, "/cnns.%d.%d", s, a.dis_n_layer); add_conv(g, net + buf, 1, d, 1);
    }
}

static const char* NET_NAMES[5] = {"gen_AB", "gen_BA", "dis_A", "dis_B", "dis_2"};

}  // namespace aclgan

const float* aclgan_ctx::param(int group, int net, const std::string& key) const {
    const Group& g = groups[group];
    auto it = g.index.find(std::string(NET_NAMES[net]) + "/" + key);
    if (it == g.index.end() || !g.param) return nullptr;
    return g.param + g.tensors[it->second].offset;
}
float* aclgan_ctx::gradp(int group, int net, const std::string& key) const {
    const Group& g = groups[group];
    auto it = g.index.find(std::string(NET_NAMES[net]) + "/" + key);
    if (it == g.index.end() || !g.grad) return nullptr;
    return g.grad + g.tensors[it->second].offset;
}
int64_t aclgan_ctx::numel_of(int group, int net, const std::string& key) const {
    const Group& g = groups[group];
    auto it = g.index.find(std::string(NET_NAMES[net]) + "/" + key);
    return it == g.index.end() ? 0 : g.tensors[it->second].numel;
}
PW aclgan_ctx::pw(int group, int net, const std::string& key, bool with_bias) const {
    PW p;
    p.w = param(group, net, key + ".weight"); p.dw = gradp(group, net, key + ".weight"); p.nw = numel_of(group, net, key + ".weight");
    if (with_bias) { p.b = param(group, net, key + ".bias"); p.db = gradp(group, net, key + ".bias"); p.nb = numel_of(group, net, key + ".bias"); }
    if (dtype != ACLGAN_DTYPE_FP32 && p.w && w16[group] && w16t[group]) {
        const int64_t off = p.w - groups[group].param;
        p.w16 = w16[group] + off; p.w16t = w16t[group] + off;
    }
    return p;
}

namespace aclgan {

// ------------------------------------------------------------------------------------------
// graph building blocks.  Every function runs the forward immediately and, when gradients are
// wanted, pushes one closure on the tape.
// ------------------------------------------------------------------------------------------
// arena budget for kept Winograd input transforms (ACLGAN_KEEPV_BUDGET_GB, default 64)
static size_t keepv_budget() {
    static size_t v = 0;
    if (!v) { const char* e = getenv("ACLGAN_KEEPV_BUDGET_GB"); const double gb = e ? atof(e) : 64.0; v = (size_t)(gb * 1073741824.0) + 1; }
    return v;
}

// roctx range of one forward pass of a network (ACLGAN_ROCTX=1); closes on every return path
struct PassScope {
    aclgan_ctx& c; int prev, prev_id;
    PassScope(aclgan_ctx& c_, int net, const char* what) : c(c_), prev(c_.pass_begin(NET_NAMES[net], what)), prev_id(c_.cur_pass_id) { c.cur_pass_id = c.pass_seq++; }
    ~PassScope() { c.pass_end(prev); c.cur_pass_id = prev_id; }
};

struct NormSpec {
    int kind = ACLGAN_NORM_NONE;
    const float* w = nullptr; const float* b = nullptr;   // AdaIN: rows of the MLP output; LN: gamma/beta
    float* dw = nullptr; float* db = nullptr;
    int w_stride = 0;
};

static inline void mark_written(Act* a) { a->gw = true; }

// Conv2dBlock.forward (networks.py:365-371) [+ preceding nn.Upsample, + ResBlock residual add]
//
// Storage under a 16-bit compute dtype (round 3; ctx.act16()): the block's output is stored in the 16-bit dtype when it is wide
// (Co % 64 == 0) and the caller says its consumers read 16-bit (out16: everything except the input of the 7x7 image-side output layer);
// the conv output in front of a normalisation layer likewise when the 16-bit-storage kernels run it (conv16s_ok, ctx.co16).  Gradients:
// d(out) follows out unless a consumer's input-gradient kernel can only write fp32 (the sub-pixel layers); d(conv output) is 16-bit
// exactly when this layer's backward runs on the 16-bit-storage kernels.  Kernels that take only fp32 never see a 16-bit tensor: the
// rules below keep it so, and every launch site checks.
static int conv_block(aclgan_ctx& c, const PW& W, bool train_w, Act* in, int Co, int k, int stride, int pad, int up, int act,
                      const NormSpec& ns, Act* residual, Act** out_p, int out16 = 1) {
    aclgan_conv_desc d;
    d.B = in->B; d.Hi = in->H; d.Wi = in->W; d.Ci = in->C; d.Co = Co; d.k = k; d.stride = stride; d.pad = pad; d.upsample = up;
    d.act = ns.kind == ACLGAN_NORM_NONE ? act : ACLGAN_ACT_NONE;
    ConvGeom g;
    CHK(make_geom(&d, &g));
    if (!W.w) { set_error("conv_block: parameters not bound"); return ACLGAN_EINVAL; }
    CHK(c.need(in)); CHK(c.need(residual));      // (another lane may have produced them)
    Act* const gin = in;                          // the tensor this block's input gradient is delivered to
    const bool want_grad = train_w || in->need_grad || ns.dw != nullptr;
    const bool has_norm = ns.kind != ACLGAN_NORM_NONE;
    // 16-bit MFMA path (compute dtype bf16 / fp16): per operator, whenever the shape has a 16-bit kernel
    const int dt = c.dtype;
    const bool h16 = dt != ACLGAN_DTYPE_FP32 && W.w16 != nullptr;
    const bool f16 = h16 && conv16_eligible(g, 0), d16 = h16 && conv16_eligible(g, 1), w16 = h16 && conv16_eligible(g, 2);
    const bool a16 = h16 && c.act16();
    const bool s_bwd = a16 && f16 && d16 && w16 && conv16s_ok(g, 1);                 // backward on the 16-bit-storage kernels
    const bool s_fwd = a16 && f16 && conv16s_ok(g, 0) && in->dt != 0;                 // forward on the LDS-DMA kernel
    if (in->dt != 0 && !(f16 && (w16 || !train_w))) {
        // a 16-bit activation reaches a layer whose forward or weight-gradient kernel only reads fp32 (odd widths of reduced test
        // networks: Cout or Cin a multiple of 32 but not of 64): one fp32 copy serves both; the input gradient goes to the original
        Act* in32 = c.new_act(in->B, in->H, in->W, in->C, false, 0);
        NEED(in32->d);
        RUN(cast_storage(in->d, in->dt, in32->d, 0, in->numel(), c.st));
        in = in32;                                // data source of this block; gradients still go to the original (gin)
    }
    const int out_st = (a16 && Co % 64 == 0 && out16) ? dt : 0;
    const int co_st = has_norm ? ((s_bwd && aclgan_ctx::co16_enabled()) ? dt : 0) : ((f16 || !out_st) ? out_st : 0);
    Act* co = c.new_act(g.B, g.Ho, g.Wo, Co, want_grad, co_st);
    NEED(co->d); if (want_grad) NEED(co->g);
    Act* out = co;
    float *mean = nullptr, *rstd = nullptr, *ss = nullptr;
    const int HW = g.Ho * g.Wo;
    if (has_norm) {
        out = c.new_act(g.B, g.Ho, g.Wo, Co, want_grad, out_st);
        NEED(out->d); if (want_grad) NEED(out->g);
        const int nstat = ns.kind == ACLGAN_NORM_LN ? g.B : g.B * Co;
        mean = c.allocf(nstat); rstd = c.allocf(nstat);
        NEED(mean); NEED(rstd);
        // the fused coefficients of the apply stay until the backward: its ReLU mask is then the sign of the same fmaf(x, scale, shift), and
        // neither backward pass reads y (2 of 6 / 1 of 4 tensor reads of the reduce / apply of every activated norm layer)
        if (want_grad && (act == ACLGAN_ACT_RELU || act == ACLGAN_ACT_LRELU) && norm_mask_setting()) { ss = c.allocf((int64_t)2 * g.B * Co); NEED(ss); }
        co->gdt = s_bwd ? dt : 0;             // read by this layer's dgrad / wgrad kernels
    } else if (out_st && !co_st) {            // an fp32-only kernel (image-side first layers) feeding 16-bit consumers: one conversion pass
        out = c.new_act(g.B, g.Ho, g.Wo, Co, want_grad, out_st);
        NEED(out->d); if (want_grad) NEED(out->g);
        out->gdt = 0;                         // this layer's backward kernels read fp32
    } else {
        out->gdt = s_bwd ? co_st : 0;         // (fp32 d with 16-bit-storage backward kernels: converted out of place in the backward)
    }
    if (gin->need_grad && !s_bwd) gin->gdt = 0;   // this layer's input-gradient kernels write fp32
    // Winograd layers: the forward's input transform V = B^T x B is exactly what the weight gradient needs again -- keep it (persistent
    // until the tape has run: 75 MB per ResBlock convolution at 256x256 B=8, ~6 GB per update) instead of recomputing it
    // Bounded: the kept transforms of one update may take at most keepv_budget() bytes of the arena (default 64 GB of the 288 GB; 256x256
    // B=8 needs 6.5 GB, 512x512 B=4 13 GB, B=32 26 GB); beyond it a layer's weight gradient recomputes V (same result bit for bit).
    float* keepV = nullptr;
    if (train_w && !f16 && !w16) {
        const size_t kb = conv_fwd_keep_bytes(g);
        if (kb && c.keep_total + kb <= keepv_budget()) { keepV = (float*)c.alloc(kb); NEED(keepV); c.keep_total += kb; }
    }
    const size_t ubytes = f16 ? 0 : conv_wino_u_bytes(g);          // fp32 Winograd layer: its filter transform is cached per update
    if (ubytes && c.ucache_reserve(W.w, g.up ? 2 : 0, ubytes)) { set_error("workspace too small (filter-transform cache)"); return ACLGAN_ENOMEM; }
    const double es_in = in->dt ? 2.0 : 4.0, es_co = co->dt ? 2.0 : 4.0, es_out = out->dt ? 2.0 : 4.0, es_w = f16 ? 2.0 : 4.0;
    c.count(es_in * (double)in->numel() + es_w * (double)Co * g.K + 4.0 * Co + es_co * (double)co->numel());             // conv: x, w, bias -> y
    c.exec_flops += conv_exec_flops(g, 0, f16);
    if (out != co) c.count((es_co + es_out) * (double)co->numel() + ((has_norm && residual) ? (residual->dt ? 2.0 : 4.0) * (double)co->numel() : 0.0));   // norm+act(+residual) / conversion: y -> out
    const size_t mark = c.top;
    // normalisation statistics from the conv epilogue where the forward kernel offers them (Winograd output transform)
    const int schunk = !has_norm ? 0 : (s_fwd ? conv_fwd16s_stats_chunk(g) : (!f16 ? conv_fwd_stats_chunk(g) : 0));
    float* stats = nullptr;
    if (schunk) { stats = c.allocf((size_t)2 * g.B * (HW / schunk) * Co); NEED(stats); }
    {
        const size_t fmark = c.top;
        void* fscr = nullptr;
        const size_t fb = s_fwd ? 0 : (f16 ? conv_fwd16_scratch_bytes(g) : conv_fwd_scratch_bytes(g));   // merged phase weights (upsample + 5x5 layers), split-K partials
        if (fb) { fscr = c.alloc(fb); NEED(fscr); }
        if (s_fwd) RUN(conv_fwd16s(g, dt, in->d, W.w16, W.b, co->d, co->dt, c.st, stats));
        else if (f16) RUN(conv_fwd16(g, dt, in->dt ? nullptr : in->d, W.w, W.w16, W.b, co->d, fscr, c.st, in->dt ? in->d : nullptr, co->dt));
        else RUN(conv_fwd(g, in->d, W.w, W.b, co->d, c.st, fscr, stats, keepV));
        c.top = fmark;
    }
    NormST nst;
    nst.x = co->dt; nst.y = out->dt; nst.res = residual ? residual->dt : 0;
    if (has_norm) {
        void* scr = c.alloc(norm_scratch_bytes(g.B, HW, Co));
        NEED(scr);
        RUN(norm_fwd(ns.kind, act, g.B, HW, Co, co->d, ns.w, ns.b, ns.w_stride, residual ? residual->d : nullptr, out->d, mean, rstd, scr, c.st, stats, schunk, &nst, ss));
    } else if (out != co) {
        RUN(cast_storage(co->d, co->dt, out->d, out->dt, co->numel(), c.st));
    }
    c.top = mark;
    *out_p = out;
    if (c.mask_dst && !c.dry && want_grad && (act == ACLGAN_ACT_RELU || act == ACLGAN_ACT_LRELU)) {
        // diagnostics: the mask this block's backward will apply = the sign of its activated output (out > 0 <=> pre-activation > 0)
        const size_t n = (size_t)out->numel();
        if (c.mask_top + n > c.mask_cap) { set_error("aclgan_debug_capture_masks: buffer too small (%zu bytes)", c.mask_cap); return ACLGAN_ENOMEM; }
        RUN(positive_mask(out->d, out->dt, c.mask_dst + c.mask_top, (int64_t)n, c.st));
        c.mask_log.push_back(aclgan_ctx::MaskEnt{out->B, out->H, out->W, out->C, act, (long long)c.mask_top});
        c.mask_top += n;
    }
    c.wrote(out); c.wrote(co);
    if (!want_grad) return ACLGAN_OK;
    aclgan_ctx* cp = &c;
    const bool ln_train = ns.kind == ACLGAN_NORM_LN && ns.dw != nullptr;
    c.push([=]() -> int {
        aclgan_ctx& c = *cp;
        if (!out->gw) return ACLGAN_OK;   // no gradient reached this block
        CHK(c.acq(out));
        if (residual && residual->need_grad) CHK(c.acq(residual));
        if (gin->need_grad) CHK(c.acq(gin));
        // backward of norm / activation: x (or y), dy -> dx (+ dres); wgrad: x, dy -> dw, db; dgrad: dy, w -> dx
        const double eg_out = out->gdt ? 2.0 : 4.0, eg_co = (has_norm ? co->gdt : (s_bwd ? dt : out->gdt)) ? 2.0 : 4.0, eg_in = gin->gdt ? 2.0 : 4.0;
        c.count((es_co + eg_out + eg_co) * (double)co->numel() + ((residual && residual->need_grad) ? (residual->gdt ? 2.0 : 4.0) * (double)co->numel() : 0.0));
        if (train_w) { c.count(es_in * (double)in->numel() + eg_co * (double)co->numel() + 4.0 * ((double)Co * g.K + Co)); c.exec_flops += conv_exec_flops(g, 2, w16); }
        if (gin->need_grad) { c.count(eg_co * (double)co->numel() + es_w * (double)Co * g.K + eg_in * (double)in->numel()); c.exec_flops += conv_exec_flops(g, 1, d16); }
        const bool side = train_w && aclgan_ctx::side_enabled();      // this layer's weight gradient goes to the side stream
        if (ubytes && gin->need_grad && !d16 && c.ucache_reserve(W.w, g.up ? 3 : 1, ubytes)) { set_error("workspace too small (filter-transform cache)"); return ACLGAN_ENOMEM; }
        float* g16 = nullptr;
        const size_t mark_pre = c.top;
        if (!has_norm && s_bwd && out->gdt == 0) {      // (with the side stream: kept below the closure's scratch mark -- that stream may read it later)
            g16 = (float*)c.alloc((size_t)out->numel() * 2);
            NEED(g16);
        }
        // LayerNorm gamma / beta gradients are parameter gradients: with the side stream on they are added there (the per-sample
        // totals sbc [B][C][2] stay put until then: allocated below the closure's scratch mark)
        float* sbc = nullptr;
        if (ln_train && side) { sbc = c.allocf((int64_t)2 * g.B * Co); NEED(sbc); }
        const size_t mark0 = side ? c.top : mark_pre;
        const float* dyp = nullptr;       // gradient w.r.t. the conv output, as the dgrad / wgrad kernels read it
        int dy_st = 0;
        if (has_norm) {
            const size_t mark = c.top;
            void* scr = c.alloc(norm_scratch_bytes(g.B, HW, Co));
            NEED(scr);
            float* dres = nullptr; int dacc = 0;
            if (residual && residual->need_grad) { dres = residual->g; dacc = residual->gw ? 1 : 0; mark_written(residual); }
            NormST bst;
            bst.x = co->dt; bst.y = out->dt; bst.dy = out->gdt; bst.dx = co->gdt; bst.dres = residual ? residual->gdt : 0;
            RUN(norm_bwd(ns.kind, act, g.B, HW, Co, co->d, out->d, out->g, ns.w, ns.w_stride, mean, rstd, co->g, ns.dw, ns.db, dres, dacc, scr, c.st, &bst, sbc, ss));
            c.top = mark;
            dyp = co->g; dy_st = co->gdt;
        } else if (s_bwd && out->gdt == 0) {
            // the consumers delivered an fp32 gradient, this layer's backward kernels read 16-bit: dy16 = act'(y) * dy, out of place
            RUN(cast_storage(out->g, 0, g16, dt, out->numel(), c.st));
            RUN(act_bwd_inplace(act, out->d, g16, out->numel(), c.st, out->dt, dt));
            dyp = g16; dy_st = dt;
        } else {
            RUN(act_bwd_inplace(act, co->d, out->g, out->numel(), c.st, co->dt, out->gdt));   // (co != out: the fp32 copy of the same values)
            dyp = out->g; dy_st = out->gdt;
        }
        if (train_w) {
            const size_t mark = c.top;
            void* wscr = nullptr;
            const size_t wb = w16 ? conv_wgrad16_scratch_bytes(g) : conv_wgrad_scratch_bytes(g);
            hipStream_t wst = c.st;
            if (side) {
                CHK(c.side_fork());
                if (!c.dry) wst = c.st2;
                c.top2 = 0;                   // side-stream work is stream-ordered: its scratch stack restarts with every launch group
                if (wb) { wscr = c.alloc2(wb); if (!wscr) { set_error("workspace too small for the side-stream scratch"); return ACLGAN_ENOMEM; } }
                if (sbc) RUN(norm_bwd_ln_params(sbc, g.B, Co, ns.dw, ns.db, wst));
            } else if (wb) { wscr = c.alloc(wb); NEED(wscr); }
            if (w16) RUN(conv_wgrad16(g, dt, in->d, dyp, W.dw, W.db, wscr, wst, in->dt, dy_st));
            else {
                if (in->dt || dy_st) { set_error("conv_block: fp32 weight-gradient kernel on 16-bit operands"); return ACLGAN_EINVAL; }
                RUN(conv_wgrad(g, in->d, dyp, W.dw, W.db, wst, wscr, keepV));
            }
            c.top = mark;
        }
        if (gin->need_grad) {
            const size_t mark = c.top;
            if (s_bwd) {
                if (!dy_st) { set_error("conv_block: 16-bit-storage dgrad on an fp32 gradient"); return ACLGAN_EINVAL; }
                void* scr = c.alloc(conv_dgrad16s_scratch_bytes(g));
                NEED(scr);
                RUN(conv_dgrad16s(g, dt, dyp, W.w16t, gin->g, gin->gdt, gin->gw ? 1 : 0, scr, c.st));
            } else {
                if (dy_st || gin->gdt) { set_error("conv_block: fp32-output dgrad kernel on 16-bit gradients"); return ACLGAN_EINVAL; }
                void* scr = c.alloc(d16 ? conv_dgrad16_scratch_bytes(g) + 256 : conv_dgrad_scratch_bytes(g));
                NEED(scr);
                if (d16) RUN(conv_dgrad16(g, dt, dyp, W.w, W.w16t, gin->g, gin->gw ? 1 : 0, scr, c.st));
                else RUN(conv_dgrad(g, dyp, W.w, gin->g, scr, gin->gw ? 1 : 0, c.st));
            }
            mark_written(gin);
            c.top = mark;
        }
        c.top = mark0;
        return ACLGAN_OK;
    }, {{train_w ? W.dw : nullptr, W.nw}, {train_w ? W.db : nullptr, W.nb}, {ln_train ? ns.dw : nullptr, Co}, {ln_train ? ns.db : nullptr, Co}});
    return ACLGAN_OK;
}

// Batched Winograd filter transforms (round 5).  The 2 * n_res ResBlock filters of an encoder / a decoder have one shape and sit at a
// constant distance in the flat parameter buffer (weight + bias, parameters() order): their transforms U = G g G^T for the whole update --
// forward and, when the generators are trained, input gradient -- are ONE launch per (network, encoder | decoder, direction) at the start
// of the update instead of one per filter at its first use (96 launches of 6 - 16 us per step at 256 x 256 B = 8).  They are written on lane 0
// before any other lane starts.  Layout and cache key are exactly what the layers will ask for (conv_wino_u_variant); a layer that asks for
// something else finds no entry and transforms its own filter as before.
static int prefill_wino_u(aclgan_ctx& c, int net, int B, int H, int W, bool train) {
    if (!aclgan_ctx::ucache_enabled() || !u_batch_setting()) return ACLGAN_OK;
    const aclgan_arch& a = c.arch;
    const int nd = a.gen_n_downsample, count = 2 * a.gen_n_res, C = a.gen_dim << nd;
    if (count < 2) return ACLGAN_OK;
    aclgan_conv_desc d;
    d.B = B; d.Hi = H >> nd; d.Wi = W >> nd; d.Ci = C; d.Co = C; d.k = 3; d.stride = 1; d.pad = 1; d.upsample = 0; d.act = ACLGAN_ACT_NONE;
    ConvGeom g;
    if (make_geom(&d, &g)) return ACLGAN_OK;
    const size_t ubytes = conv_wino_u_bytes(g);
    if (!ubytes) return ACLGAN_OK;
    if (c.dtype != ACLGAN_DTYPE_FP32 && c.w16[0] && conv16_eligible(g, 0)) return ACLGAN_OK;      // these layers run on the 16-bit kernels
    char buf[160];
    for (int part = 0; part < 2; ++part) {
        std::vector<const float*> ws;
        for (int r = 0; r < a.gen_n_res; ++r)
            for (int j = 0; j < 2; ++j) {
                if (part == 0) snprintf(buf, sizeof buf, "enc_content.model.%d.model.%d.model.%d.conv.weight", 1 + nd, r, j);
                else snprintf(buf, sizeof buf, "dec.model.0.model.%d.model.%d.conv.weight", r, j);
                ws.push_back(c.param(0, net, buf));
            }
        if (!ws[0]) continue;
        const int64_t wstride = ws[1] - ws[0];
        bool even = wstride > 0;
        for (int i = 1; i < count && even; ++i) even = ws[i] && ws[i] - ws[i - 1] == wstride;
        if (!even) continue;
        for (int dgrad = 0; dgrad <= (train ? 1 : 0); ++dgrad) {
            // (the forward keeps its input transform for the weight gradient only where neither runs fused: conv_block's rule)
            const bool keepV = train && !dgrad && conv_fwd_keep_bytes(g) > 0;
            const int uv = conv_wino_u_variant(g, dgrad, keepV);
            if (uv < 0 || c.ucache.count(std::make_pair(ws[0], uv & 15))) continue;
            float* u0 = nullptr;
            bool contiguous = true;
            const size_t ustride = (ubytes + 255) & ~(size_t)255;
            for (int i = 0; i < count; ++i) {
                float* u = (float*)c.alloc(ubytes);
                NEED(u);
                if (i == 0) u0 = u;
                else if ((char*)u != (char*)u0 + (size_t)i * ustride) contiguous = false;
                c.ucache[std::make_pair(ws[i], uv & 15)] = aclgan_ctx::UEnt{u, contiguous, uv >> 4, c.cur_lane, c.nck(c.cur_lane)};
            }
            if (!contiguous) {          // (cannot happen with the stack allocator; if it ever does, the layers fill their entries themselves)
                for (int i = 0; i < count; ++i) c.ucache[std::make_pair(ws[i], uv & 15)].filled = false;
                continue;
            }
            RUN(conv_wino_prefill(g, uv, ws[0], wstride, u0, (int64_t)(ustride / sizeof(float)), count, c.st));
        }
    }
    return ACLGAN_OK;
}

// Round 6: the batched transforms of an update (8 launches of ~60 us in gen_update, 4 in dis_update: 19 MB of output each) used to sit in the
// preamble on lane 0, in front of the chain everything else waits for, although the first layer that reads one is the fourth of the first
// encoder pass.  They now run on the last lane beside the first three encoder layers; the cache entries carry that lane's stamp, so a reader
// on another lane waits for exactly the launch that fills its entry (ucache_lookup).  The lane first waits for lane 0's checkpoint: the
// previous call's optimizer step wrote the parameters on the caller's stream.
static int prefill_on_side_lane(aclgan_ctx& c, int B, int H, int W, bool train) {
    const int AB = ACLGAN_NET_GEN_AB, BA = ACLGAN_NET_GEN_BA;
    static int on_lane = -1;      // ACLGAN_PREFILL_LANE=0: back in lane 0's preamble (A/B switch)
    if (on_lane < 0) { const char* e = getenv("ACLGAN_PREFILL_LANE"); on_lane = (e && !atoi(e)) ? 0 : 1; }
    if (c.nlanes <= 1 || !on_lane) { CHK(prefill_wino_u(c, AB, B, H, W, train)); return prefill_wino_u(c, BA, B, H, W, train); }
    const int LP = c.nlanes - 1;
    CHK(c.mark());
    const int k0 = c.nck(0) - 1;
    CHK(c.set_lane(LP));
    CHK(c.wait_ck(LP, 0, k0));
    CHK(prefill_wino_u(c, AB, B, H, W, train)); CHK(prefill_wino_u(c, BA, B, H, W, train));
    CHK(c.mark());
    return c.set_lane(0);
}

// ContentEncoder.forward (networks.py:230-245)
static int content_encode(aclgan_ctx& c, int net, bool train, Act* x, Act** out) {
    PassScope pass(c, net, "encode");
    const aclgan_arch& a = c.arch;
    char buf[160];
    NormSpec in_; in_.kind = ACLGAN_NORM_IN;
    Act* h = nullptr;
    int d = a.gen_dim;
    CHK(conv_block(c, c.pw(0, net, "enc_content.model.0.conv"), train, x, d, 7, 1, 3, 0, ACLGAN_ACT_RELU, in_, nullptr, &h));
    for (int i = 0; i < a.gen_n_downsample; ++i) {
        snprintf(buf, sizeof buf, "enc_content.model.%d.conv", 1 + i);
        CHK(conv_block(c, c.pw(0, net, buf), train, h, 2 * d, 4, 2, 1, 0, ACLGAN_ACT_RELU, in_, nullptr, &h));
        d *= 2;
    }
    for (int r = 0; r < a.gen_n_res; ++r) {
        Act *t = nullptr, *o = nullptr;
        snprintf(buf, sizeof buf, "enc_content.model.%d.model.%d.model.0.conv", 1 + a.gen_n_downsample, r);
        CHK(conv_block(c, c.pw(0, net, buf), train, h, d, 3, 1, 1, 0, ACLGAN_ACT_RELU, in_, nullptr, &t));
        snprintf(buf, sizeof buf, "enc_content.model.%d.model.%d.model.1.conv", 1 + a.gen_n_downsample, r);
        CHK(conv_block(c, c.pw(0, net, buf), train, t, d, 3, 1, 1, 0, ACLGAN_ACT_NONE, in_, h, &o));
        h = o;
    }
    *out = h;
    return ACLGAN_OK;
}

// dense layer with tape (launch = false: the caller runs the forward itself -- the fused MLP launch of decode() -- and this call only
// allocates the output, accounts for it and records the backward)
static int dense(aclgan_ctx& c, const PW& W, bool train_w, Act* in, int O, int act, Act** out_p, bool launch = true) {
    const int B = in->B, I = in->H * in->W * in->C;
    const bool want = train_w || in->need_grad;
    CHK(c.need(in));
    Act* out = c.new_act(B, 1, 1, O, want);
    NEED(out->d); if (want) NEED(out->g);
    if (!W.w) { set_error("dense: parameters not bound"); return ACLGAN_EINVAL; }
    if (launch) RUN(linear_fwd(B, I, O, in->d, W.w, W.b, act, out->d, c.st));
    c.count(4.0 * ((double)B * I + (double)O * I + O + (double)B * O) * (want ? 3.0 : 1.0));    // (+ backward: the same operands again, twice)
    c.wrote(out);
    *out_p = out;
    if (!want) return ACLGAN_OK;
    aclgan_ctx* cp = &c;
    c.push([=]() -> int {
        aclgan_ctx& c = *cp;
        if (!out->gw) return ACLGAN_OK;
        CHK(c.acq(out));
        if (in->need_grad) CHK(c.acq(in));
        float* dx = nullptr;
        float* tmp = nullptr;
        const size_t mark = c.top;
        if (in->need_grad) {
            if (in->gw) { tmp = c.allocf((int64_t)B * I); NEED(tmp); dx = tmp; } else dx = in->g;
        }
        // dy *= act'(y), dx on this closure's lane; dw / db (parameter gradients) on the side stream: x and dy stay put until the update ends
        const bool side = train_w && aclgan_ctx::side_enabled();
        RUN(linear_bwd(B, I, O, in->d, out->d, out->g, W.w, act, dx, (train_w && !side) ? W.dw : nullptr, (train_w && !side) ? W.db : nullptr, c.st));
        if (side) {
            CHK(c.side_fork());
            RUN(linear_bwd_params(B, I, O, in->d, out->g, W.dw, W.db, c.st2));
        }
        if (in->need_grad) {
            if (tmp) {
                // in->g += tmp  (reuse the GAP backward kernel with HW = 1: dx[i] += dy[i])
                RUN(gap_bwd(B, 1, I, tmp, in->g, 1, c.st));
            }
            mark_written(in);
        }
        c.top = mark;
        return ACLGAN_OK;
    }, {{train_w ? W.dw : nullptr, W.nw}, {train_w ? W.db : nullptr, W.nb}});
    return ACLGAN_OK;
}

// StyleEncoder.forward (networks.py:212-228)
static int style_encode(aclgan_ctx& c, int net, bool train, Act* x, Act** out) {
    PassScope pass(c, net, "style");
    const aclgan_arch& a = c.arch;
    char buf[160];
    NormSpec none;
    Act* h = nullptr;
    int d = a.gen_dim;
    CHK(conv_block(c, c.pw(0, net, "enc_style.model.0.conv"), train, x, d, 7, 1, 3, 0, ACLGAN_ACT_RELU, none, nullptr, &h));
    for (int i = 0; i < 2; ++i) {
        snprintf(buf, sizeof buf, "enc_style.model.%d.conv", 1 + i);
        CHK(conv_block(c, c.pw(0, net, buf), train, h, 2 * d, 4, 2, 1, 0, ACLGAN_ACT_RELU, none, nullptr, &h));
        d *= 2;
    }
    for (int i = 0; i < 2; ++i) {
        snprintf(buf, sizeof buf, "enc_style.model.%d.conv", 3 + i);
        CHK(conv_block(c, c.pw(0, net, buf), train, h, d, 4, 2, 1, 0, ACLGAN_ACT_RELU, none, nullptr, &h));
    }
    // AdaptiveAvgPool2d(1) (networks.py:222)
    const bool want = h->need_grad;
    Act* p = c.new_act(h->B, 1, 1, d, want);
    NEED(p->d); if (want) NEED(p->g);
    RUN(gap_fwd(h->B, h->H * h->W, d, h->d, p->d, c.st, h->dt));
    c.wrote(p);
    c.count((h->dt ? 2.0 : 4.0) * (double)h->numel() * (want ? 2.0 : 1.0));
    if (want) {
        aclgan_ctx* cp = &c;
        Act* hh = h;
        c.push([=]() -> int {
            aclgan_ctx& c = *cp;
            if (!p->gw) return ACLGAN_OK;
            CHK(c.acq(p)); CHK(c.acq(hh));
            RUN(gap_bwd(hh->B, hh->H * hh->W, hh->C, p->g, hh->g, hh->gw ? 1 : 0, c.st, hh->gdt));
            mark_written(hh);
            return ACLGAN_OK;
        });
    }
    // 1x1 conv dim -> style_dim (networks.py:223) == dense layer on [B][dim]
    CHK(dense(c, c.pw(0, net, "enc_style.model.6"), train, p, a.gen_style_dim, ACLGAN_ACT_NONE, out));
    return ACLGAN_OK;
}

// AdaINGen.decode (networks.py:147-163) + Decoder.forward (networks.py:247-264)
static int decode(aclgan_ctx& c, int net, bool train, Act* content, Act* style, Act** out) {
    PassScope pass(c, net, "decode");
    const aclgan_arch& a = c.arch;
    char buf[160];
    const int C = content->C, nr = a.gen_n_res, nad = 2 * C * 2 * nr;
    // MLP (networks.py:280-292)
    Act *m0 = nullptr, *m1 = nullptr, *ap = nullptr;
    // (round 6) one launch for the three layers where the widths allow (misc.hip: mlp3_fwd -- the same bits as three linear_fwd launches);
    // activations, accounting and the three backward closures are those of dense()
    const int sdim = style->H * style->W * style->C;
    const bool fused = mlp_fused_setting() && mlp3_fwd_ok(sdim, a.gen_mlp_dim);
    const PW W0 = c.pw(0, net, "mlp.model.0.fc"), W1 = c.pw(0, net, "mlp.model.1.fc"), W2 = c.pw(0, net, "mlp.model.2.fc");
    CHK(dense(c, W0, train, style, a.gen_mlp_dim, ACLGAN_ACT_RELU, &m0, !fused));
    CHK(dense(c, W1, train, m0, a.gen_mlp_dim, ACLGAN_ACT_RELU, &m1, !fused));
    CHK(dense(c, W2, train, m1, nad, ACLGAN_ACT_NONE, &ap, !fused));
    if (fused) RUN(mlp3_fwd(style->B, sdim, a.gen_mlp_dim, nad, style->d, W0.w, W0.b, W1.w, W1.b, W2.w, W2.b, m0->d, m1->d, ap->d, c.st));
    if (ap->need_grad) {   // AdaIN layers accumulate dw/db into disjoint column slices
        RUN(fill_zero(ap->g, ap->numel(), c.st));
        mark_written(ap);
    }
    Act* h = content;
    int j = 0;
    for (int r = 0; r < nr; ++r) {
        Act *t = nullptr, *o = nullptr;
        for (int half = 0; half < 2; ++half) {
            NormSpec ns; ns.kind = ACLGAN_NORM_ADAIN; ns.w_stride = nad;
            // assign_adain_params (networks.py:154-163): columns [2Cj, 2Cj+C) = bias, [2Cj+C, 2Cj+2C) = weight
            ns.b = ap->d + 2 * C * j; ns.w = ap->d + 2 * C * j + C;
            if (ap->need_grad) { ns.db = ap->g + 2 * C * j; ns.dw = ap->g + 2 * C * j + C; }
            ++j;
            snprintf(buf, sizeof buf, "dec.model.0.model.%d.model.%d.conv", r, half);
            if (half == 0) CHK(conv_block(c, c.pw(0, net, buf), train, h, C, 3, 1, 1, 0, ACLGAN_ACT_RELU, ns, nullptr, &t));
            else CHK(conv_block(c, c.pw(0, net, buf), train, t, C, 3, 1, 1, 0, ACLGAN_ACT_NONE, ns, h, &o));
        }
        h = o;
    }
    int d = C, idx = 2;
    for (int i = 0; i < a.gen_n_downsample; ++i) {
        NormSpec ns; ns.kind = ACLGAN_NORM_LN;
        snprintf(buf, sizeof buf, "dec.model.%d.norm.gamma", idx); ns.w = c.param(0, net, buf); ns.dw = train ? c.gradp(0, net, buf) : nullptr;
        snprintf(buf, sizeof buf, "dec.model.%d.norm.beta", idx); ns.b = c.param(0, net, buf); ns.db = train ? c.gradp(0, net, buf) : nullptr;
        snprintf(buf, sizeof buf, "dec.model.%d.conv", idx);
        // (the last of these feeds the fp32 7x7 image-side output layer: its output stays fp32)
        CHK(conv_block(c, c.pw(0, net, buf), train, h, d / 2, 5, 1, 2, 1, ACLGAN_ACT_RELU, ns, nullptr, &h, i + 1 < a.gen_n_downsample ? 1 : 0));
        d /= 2; idx += 2;
    }
    NormSpec none;
    snprintf(buf, sizeof buf, "dec.model.%d.conv", idx - 1);
    CHK(conv_block(c, c.pw(0, net, buf), train, h, a.gen_output_dim, 7, 1, 3, 0, ACLGAN_ACT_TANH, none, nullptr, out));
    return ACLGAN_OK;
}

// MsImageDis.forward (networks.py:50-57)
// lane_a / lane_b (round 5): the full-resolution scale runs on lane_a, the pooling chain and the two coarser scales on lane_b -- their
// launches are a quarter / a sixteenth of the first scale's and fill a few CUs each (9 .. 64 workgroups); next to the first scale's
// full-chip kernels they cost next to nothing (profiles/r04_microbench_coresidency.txt).  lane_a == lane_b: one queue, as before.
static int dis_forward(aclgan_ctx& c, int net, bool train, Act* x, std::vector<Act*>* outs, int lane_a = -1, int lane_b = -1) {
    PassScope pass(c, net, "forward");
    const aclgan_arch& a = c.arch;
    char buf[160];
    NormSpec none;
    Act* xin = x;
    if (lane_a < 0) lane_a = lane_b = c.cur_lane;
    for (int s = 0; s < a.dis_num_scales; ++s) {
        CHK(c.set_lane(s == 0 ? lane_a : lane_b));
        Act* h = xin;
        int d = a.dis_dim;
        for (int i = 0; i < a.dis_n_layer; ++i) {
            snprintf(buf, sizeof buf, "cnns.%d.%d.conv", s, i);
            const int co = i == 0 ? d : 2 * d;
            // (the last layer of a scale feeds the fp32 1x1 head: its output stays fp32)
            CHK(conv_block(c, c.pw(1, net, buf), train, h, co, 4, 2, 1, 0, ACLGAN_ACT_LRELU, none, nullptr, &h, i + 1 < a.dis_n_layer ? 1 : 0));
            if (i > 0) d *= 2;
        }
        snprintf(buf, sizeof buf, "cnns.%d.%d", s, a.dis_n_layer);
        Act* o = nullptr;
        CHK(conv_block(c, c.pw(1, net, buf), train, h, 1, 1, 1, 0, 0, ACLGAN_ACT_NONE, none, nullptr, &o));
        outs->push_back(o);
        if (s + 1 < a.dis_num_scales) {
            // self.downsample (networks.py:33,53)
            CHK(c.set_lane(lane_b));
            Act* src = xin;
            CHK(c.need(src));
            Act* p = c.new_act(src->B, (src->H - 1) / 2 + 1, (src->W - 1) / 2 + 1, src->C, src->need_grad);
            NEED(p->d); if (p->need_grad) NEED(p->g);
            RUN(avgpool3s2_fwd(src->B, src->H, src->W, src->C, src->d, p->d, c.st));
            c.wrote(p);
            c.count(4.0 * ((double)src->numel() + (double)p->numel()) * (src->need_grad ? 2.0 : 1.0));
            if (src->need_grad) {
                aclgan_ctx* cp = &c;
                c.push([=]() -> int {
                    aclgan_ctx& c = *cp;
                    if (!p->gw) return ACLGAN_OK;
                    CHK(c.acq(p)); CHK(c.acq(src));
                    RUN(avgpool3s2_bwd(src->B, src->H, src->W, src->C, p->g, src->g, src->gw ? 1 : 0, c.st));
                    mark_written(src);
                    return ACLGAN_OK;
                });
            }
            xin = p;
        }
    }
    return ACLGAN_OK;
}

// discriminator pass + LSGAN terms over a JOINT batch: passes that share discriminator weights (e.g.
// dis_A on x_A_fake and on x_A2_fake, trainer.py:136-137) run as one pass over the concatenated batch --
// same arithmetic per sample, half the launches and twice the rows for the small late layers.  Segment i
// covers `nb` consecutive samples with its own target / reported weight / gradient scale / loss slot.
// The loss gradient w.r.t. each scale's map is written at forward time: total = sum_i gscale_i * loss_i is
// linear in the reported losses.  Round 5: ONE launch for all scales and segments of the call (networks.py:64-67,81-83,96-98
// loop over the scales in Python), the terms added to their slots in the order of the former per-term launches.
struct LsSeg { float target, weight, gscale; float* slot; };
static int dis_lsgan(aclgan_ctx& c, int net, bool train, Act* x, int nb, const std::vector<LsSeg>& segs, int lane_a = -1, int lane_b = -1) {
    std::vector<Act*> outs;
    CHK(dis_forward(c, net, train, x, &outs, lane_a, lane_b));
    std::vector<LsganTerm> terms;
    for (Act* o : outs) {
        CHK(c.need(o));
        const int n = nb * o->H * o->W * o->C;
        for (size_t i = 0; i < segs.size(); ++i) {
            LsganTerm t;
            t.o = o->d + (size_t)i * n; t.d_o = o->need_grad ? o->g + (size_t)i * n : nullptr; t.slot = segs[i].slot;
            t.n = n; t.target = segs[i].target; t.weight = segs[i].weight; t.gscale = segs[i].gscale;
            terms.push_back(t);
        }
        if (o->need_grad) mark_written(o);
    }
    RUN(lsgan_loss_batch(terms.data(), (int)terms.size(), c.st, c.lscale));
    return ACLGAN_OK;
}

// focus_translation (trainer.py:85-88) with optional 6-channel pair (trainer.py:132-133)
// out_dst / pair_dst: optional pre-made destinations (batch-slice views of joint discriminator inputs)
static int blend(aclgan_ctx& c, Act* dec4, Act* bg, Act* pair_first, Act** out_p, Act** pair_p, Act* out_dst = nullptr, Act* pair_dst = nullptr) {
    const bool want = dec4->need_grad;
    CHK(c.need(dec4)); CHK(c.need(bg)); CHK(c.need(pair_first));
    Act* out = out_dst;
    if (!out) {
        out = c.new_act(dec4->B, dec4->H, dec4->W, 3, want);
        NEED(out->d); if (want) NEED(out->g);
    }
    Act* pair = nullptr;
    if (pair_first) {
        pair = pair_dst;
        if (!pair) {
            pair = c.new_act(dec4->B, dec4->H, dec4->W, 6, want);
            NEED(pair->d); if (want) NEED(pair->g);
        }
    }
    const bool plain = dec4->C == 3;      // non-focus configuration (trainer.py:117-121,129-130): the decoder output is the image itself
    if (plain) RUN(plain_pair_fwd(dec4->B, dec4->H * dec4->W, dec4->d, out->d, pair_first ? pair_first->d : nullptr, pair ? pair->d : nullptr, c.st));
    else RUN(focus_blend_fwd(dec4->B, dec4->H * dec4->W, dec4->d, bg->d, out->d, pair_first ? pair_first->d : nullptr, pair ? pair->d : nullptr, c.st));
    c.count(4.0 * (double)dec4->B * dec4->H * dec4->W * ((plain ? 3 : 4 + 3) + 3 + (pair_first ? 9 : 0)) * (want ? 2.0 : 1.0));
    c.wrote(out); c.wrote(pair);
    *out_p = out;
    if (pair_p) *pair_p = pair;
    if (!want) return ACLGAN_OK;
    aclgan_ctx* cp = &c;
    c.push([=]() -> int {
        aclgan_ctx& c = *cp;
        const float* dout = out->written() ? out->g : nullptr;
        const float* dpair = (pair && pair->written()) ? pair->g : nullptr;
        if (!dout && !dpair) return ACLGAN_OK;
        if (dout) CHK(c.acq(out));
        if (dpair) CHK(c.acq(pair));
        CHK(c.acq(dec4));
        if (!plain && bg->need_grad) CHK(c.acq(bg));
        if (plain) { RUN(plain_pair_bwd(dec4->B, dec4->H * dec4->W, dout, dpair, dec4->g, c.st)); return ACLGAN_OK; }
        float* dbg = bg->need_grad ? bg->g : nullptr;
        RUN(focus_blend_bwd(dec4->B, dec4->H * dec4->W, dec4->d, bg->d, dout, dpair, dec4->g, dbg, bg->gw ? 1 : 0, c.st));
        if (dbg) mark_written(bg);
        return ACLGAN_OK;
    });
    return ACLGAN_OK;
}

static int zero_grad_of(aclgan_ctx& c, Act* a) {
    if (a->need_grad) { RUN(fill_zero(a->g, a->numel(), c.st)); mark_written(a); }
    return ACLGAN_OK;
}

__global__ void scale_kernel(const float* s, float* d, float a, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = a * s[i];
}

__global__ void gen_total_kernel(float* L, aclgan_hparams hp, float focus_scale) {
    float t = hp.gan_w * L[ACLGAN_L_GEN_ADV_A] + hp.gan_w * L[ACLGAN_L_GEN_ADV_B] + hp.gan_cw * L[ACLGAN_L_GEN_ADV_2];
    t += focus_scale * (L[ACLGAN_L_GEN_FOCUS_B_SIZE] + L[ACLGAN_L_GEN_FOCUS_B_DIGIT] + L[ACLGAN_L_GEN_FOCUS_A_SIZE] +
                        L[ACLGAN_L_GEN_FOCUS_A_DIGIT] + L[ACLGAN_L_GEN_FOCUS_A2_SIZE] + L[ACLGAN_L_GEN_FOCUS_A2_DIGIT]);
    t += hp.recon_x_w * L[ACLGAN_L_IDT_A] + hp.recon_x_w * L[ACLGAN_L_IDT_B];
    L[ACLGAN_L_GEN_TOTAL] = t;
}
__global__ void dis_total_kernel(float* L, aclgan_hparams hp) {
    L[ACLGAN_L_DIS_TOTAL] = hp.gan_w * L[ACLGAN_L_DIS_A] + hp.gan_w * L[ACLGAN_L_DIS_B] + hp.gan_cw * L[ACLGAN_L_DIS_2];
}

// Refresh the 16-bit weight packs from the fp32 master parameters (both groups: each update runs the other group's
// networks forward / dgrad-only).  ~55 M elements read once, 2 x 2 bytes written: ~0.1 ms against a step of tens of ms,
// and it makes the packs immune to whatever touched the flat buffers between calls (Adam, load_state_dict, broadcast).
static int pack_params(aclgan_ctx& c) {
    if (c.dtype == ACLGAN_DTYPE_FP32 || c.dry) return ACLGAN_OK;
    for (int g = 0; g < 2; ++g) {
        if (!c.w16[g] || !c.w16t[g]) { set_error("compute dtype is 16-bit but aclgan_bind_params16 was not called for group %d", g); return ACLGAN_EINVAL; }
        CHK(cast_flat16(c.groups[g].param, c.w16[g], c.groups[g].numel, c.dtype, c.st));
        CHK(transpose_flat16(c.groups[g].param, c.w16t[g], c.cv_off[g].data(), c.cv_co[g].data(), c.cv_taps[g].data(), c.cv_ci[g].data(),
                             (int)c.cv_off[g].size(), c.dtype, c.st));
    }
    return ACLGAN_OK;
}

static int input_act(aclgan_ctx& c, const float* nchw, int B, int C, int H, int W, Act** out) {
    Act* a = c.new_act(B, H, W, C, false);
    NEED(a->d);
    RUN(nchw_to_nhwc(nchw, a->d, B, C, H, W, c.st));
    *out = a;
    return ACLGAN_OK;
}

static int wrap_vec(aclgan_ctx& c, const float* dev, int B, int n, float scale, Act** out) {
    Act* a = c.new_act(B, 1, 1, n, false);
    NEED(a->d);
    if (!c.dry) {
        hipLaunchKernelGGL(scale_kernel, dim3(cdiv(B * n, 256)), dim3(256), 0, c.st, dev, a->d, scale, B * n);
        ACL_CHECK_LAUNCH("scale_kernel");
    }
    *out = a;
    return ACLGAN_OK;
}

// Backward = the tape in reverse.  Data-parallel gradient buckets (not in the reference): the trained
// group's flat gradient buffer is cut into fixed-size buckets; a bucket is COMPLETE once the closure with the smallest
// forward index that writes into it has been enqueued (networks are used several times per step -- the decoder up to
// three times -- and every use accumulates into the same dw).  The callback fires at that point, on the host, while the
// remaining backward is still being enqueued: the caller starts that bucket's all-reduce on its communication stream
// (ordered after everything enqueued so far), so the exchange overlaps the rest of the backward.  Order = reverse-backward
// readiness, identical on every rank (the graph is static).
static void fire_bucket(aclgan_ctx& c, int b) {
    c.bucket_order.push_back(b);
    if (!c.bucket_fn || (c.dry && !c.fire_dry)) return;
    const int64_t off = (int64_t)b * c.bucket_elems;
    const int64_t n = std::min<int64_t>(c.bucket_elems, c.groups[c.trained].numel - off);
    c.bucket_fn(c.bucket_user, c.trained, b, off, n);
}
static int run_tape_impl(aclgan_ctx& c) {
    const size_t n = c.tape.size();
    const bool buckets = c.trained >= 0 && c.bucket_elems > 0;
    std::vector<std::vector<int>> done_at;
    if (buckets) {
        const int nb = (int)cdiv64(c.groups[c.trained].numel, c.bucket_elems);
        std::vector<size_t> first(nb, n);
        for (size_t i = 0; i < n; ++i)
            for (int r = 0; r < c.tape[i].ng; ++r) {
                const int64_t b0 = c.tape[i].goff[r] / c.bucket_elems, b1 = (c.tape[i].goff[r] + c.tape[i].gnum[r] - 1) / c.bucket_elems;
                for (int64_t b = b0; b <= b1 && b < nb; ++b) first[b] = std::min(first[b], i);
            }
        done_at.resize(n);
        c.bucket_order.clear();
        for (int b = 0; b < nb; ++b) {
            if (first[b] == n) fire_bucket(c, b);   // no closure writes here (alignment padding, untouched tensors): already final
            else done_at[first[b]].push_back(b);
        }
    }
    // the loss kernels ran on lane 0 after it had joined every lane: from here every lane may read everything of the forward pass
    // (activations, statistics, the gradients the loss kernels wrote) and reuse any scratch
    CHK(c.lanes_barrier());
    const bool marks = Roctx::get().on && !c.dry && c.tape_pass.size() == n;
    int open = -1, last_pass = -2;
    for (size_t i = n; i-- > 0;) {
        if (marks && c.tape_pass[i] != open) {      // consecutive closures of one forward pass = one backward range
            if (open >= 0) { Roctx::get().pop(); Roctx::get().push(("~end@" + std::to_string((long long)g_launches)).c_str()); Roctx::get().pop(); }
            open = c.tape_pass[i];
            if (open >= 0) Roctx::get().push(("bwd:" + c.pass_names[open] + "@" + std::to_string((long long)g_launches)).c_str());
        }
        // a closure runs on the lane of the forward pass that recorded it; a checkpoint at every pass boundary keeps the waits of
        // consumers on other lanes exact (they wait for the end of the producing pass, not for whatever that lane was given afterwards)
        if (c.tape[i].pass != last_pass) { if (last_pass != -2) CHK(c.mark()); last_pass = c.tape[i].pass; }
        CHK(c.set_lane(c.tape[i].lane));
        int rc_i = c.tape[i].fn();
        if (!rc_i && !c.dry && fault_at_setting() >= 0 && (int)(n - 1 - i) == fault_at_setting()) { set_error("injected fault after backward closure %d (tuning key fault_at)", fault_at_setting()); rc_i = ACLGAN_EHIP; }
        if (rc_i) { if (marks && open >= 0) Roctx::get().pop(); return rc_i; }
        if (buckets && !done_at[i].empty()) {
            // a bucket handed to the all-reduce must hold everything written into it: the side stream's parameter gradients, and the
            // caller's stream (lane 0, the one its communication stream orders after) must have seen every lane
            const int lane = c.cur_lane;
            CHK(c.lanes_join());
            for (int b : done_at[i]) fire_bucket(c, b);
            CHK(c.set_lane(lane));
        }
    }
    if (marks && open >= 0) { Roctx::get().pop(); Roctx::get().push(("~end@" + std::to_string((long long)g_launches)).c_str()); Roctx::get().pop(); }
    return c.lanes_join();
}
static int run_tape(aclgan_ctx& c) {
    const int rc = run_tape_impl(c);
    // an error leaves closures half enqueued on lanes / the side stream: nothing may still be in flight there when the caller is told
    // (it will reuse or free the workspace)
    if (rc) c.lanes_quiesce();
    return rc;
}

static int check_shape(const aclgan_ctx& c, int B, int H, int W) {
    ACL_REQUIRE(B >= 1 && H >= 64 && W >= 64, "batch shape (%d,%d,%d): need B>=1 and H,W>=64 (third discriminator scale reflect-pads a >=2x2 map)", B, H, W);
    // the reference needs exactly this: the decoder upsamples x2^n_downsample from floor(H / 2^n) and the L1 identity
    // loss / focus blend compare it with the input (trainer.py:110-116,162-163)
    const int q = 1 << c.arch.gen_n_downsample;
    ACL_REQUIRE(H % q == 0 && W % q == 0, "H, W must be multiples of %d (2^n_downsample: decoder output must match the input size)", q);
    return ACLGAN_OK;
}

// ---- gen_update (trainer.py:99-169, focus branch) ----
static int gen_update_impl(aclgan_ctx& c, const float* x_a, const float* x_b, const float* z, int B, int H, int W,
                           const aclgan_hparams& hp, float* L) {
    CHK(check_shape(c, B, H, W));
    // focus branch (trainer.py:107-116,126-128,145-161): gen.output_dim 4 = image + focus mask.  Non-focus branch (focus_loss 0,
    // trainer.py:117-121,129-130: the paper's ablation): the decoder output is the image, which only type-checks against the 3-channel
    // discriminators with gen.output_dim 3.
    const bool focus = hp.focus_loss > 0.f;
    ACL_REQUIRE(c.arch.gen_output_dim == (focus ? 4 : 3), "focus_loss %s 0 needs gen.output_dim %d (trainer.py:107-121), the context was built with %d",
                focus ? ">" : "<=", focus ? 4 : 3, c.arch.gen_output_dim);
    const int sd = c.arch.gen_style_dim;
    const int AB = ACLGAN_NET_GEN_AB, BA = ACLGAN_NET_GEN_BA;
    const int nfs = 2 * focus_sums_blocks((int64_t)B * H * W);   // per-workgroup partials of the focus sums, one set per mask
    float* sums = c.allocf((int64_t)3 * nfs);
    NEED(sums);
    if (!c.dry) {
        hipError_t e = hipMemsetAsync(L, 0, sizeof(float) * (ACLGAN_L_GEN_TOTAL + 1), c.st);
        if (e != hipSuccess) return hip_fail(e, "memset losses");
    }
    CHK(c.lanes_begin(lanes_setting()));
    CHK(pack_params(c));
    Act *xa, *xb, *z1, *z2, *z3;
    CHK(input_act(c, x_a, B, 3, H, W, &xa));
    CHK(input_act(c, x_b, B, 3, H, W, &xb));
    CHK(wrap_vec(c, z, B, sd, 1.f, &z1));
    CHK(wrap_vec(c, z + (size_t)B * sd, B, sd, hp.alpha, &z2));   // alpha multiplies only z_2 (trainer.py:109)
    CHK(wrap_vec(c, z + (size_t)2 * B * sd, B, sd, 1.f, &z3));
    // the Winograd transforms of all ResBlock filters this update will use: one batched launch per (network, encoder | decoder,
    // forward | input gradient) instead of one per filter and use
    CHK(prefill_on_side_lane(c, B, H, W, true));
    Act *c1, *c2, *s2, *c4, *s4, *c3;
    Act *dB4, *dA4, *xB, *xA, *pA1, *rA4, *rB4, *dA24, *xA2, *pA2;
    // joint discriminator inputs: (x_A_fake | x_A2_fake) for dis_A, (pair_A1 | pair_A2) for dis_2
    Act* jA = c.new_act(2 * B, H, W, 3, true);
    Act* jP = c.new_act(2 * B, H, W, 6, true);
    NEED(jA->d); NEED(jA->g); NEED(jP->d); NEED(jP->g);
    // Lanes (round 5).  Lane 0 carries the chain everything else waits for: x_a -> gen_AB -> x_B_fake -> gen_BA -> x_A2_fake
    // (trainer.py:103,108,110,125-128); lane 1 the other translation direction and the two reconstructions (trainer.py:104-105,109,111,
    // 113-114), which only meet the chain again in the discriminators.  One lane: the same host order on one queue.
    // A third lane takes the reconstruction of x_b (trainer.py:105,114: encode(x_b) -> decode, independent of everything else up to its L1
    // loss) and the coarser discriminator scales.
    const int L0 = 0, L1 = 1, L2 = c.nlanes > 2 ? 2 : 1, LS = c.nlanes > 2 ? 2 : 0;
#define PASS(expr) do { CHK(expr); CHK(c.mark()); } while (0)
    CHK(c.mark());                                                    // (the preamble: inputs, noise, filter transforms)
    CHK(c.set_lane(L0));
    PASS(content_encode(c, AB, true, xa, &c1));                      // trainer.py:103 (style dropped)
    CHK(c.set_lane(L1));
    PASS(content_encode(c, BA, true, xa, &c2));                      // trainer.py:104
    PASS(style_encode(c, BA, true, xa, &s2));
    CHK(c.set_lane(L0));
    PASS(decode(c, AB, true, c1, z1, &dB4)); CHK(zero_grad_of(c, dB4));   // trainer.py:108
    PASS(blend(c, dB4, xa, nullptr, &xB, nullptr));                  // trainer.py:110
    CHK(c.set_lane(L1));
    PASS(decode(c, BA, true, c2, z2, &dA4)); CHK(zero_grad_of(c, dA4));   // trainer.py:109
    PASS(blend(c, dA4, xa, xa, &xA, &pA1, c.new_view(jA, 0, B), c.new_view(jP, 0, B)));   // trainer.py:111,132
    CHK(c.set_lane(L0));
    PASS(content_encode(c, BA, true, xB, &c3));                      // trainer.py:125
    PASS(decode(c, BA, true, c3, z3, &dA24)); CHK(zero_grad_of(c, dA24)); // trainer.py:127
    PASS(blend(c, dA24, xB, xa, &xA2, &pA2, c.new_view(jA, B, B), c.new_view(jP, B, B)));   // trainer.py:128,133
    CHK(c.set_lane(L1));
    PASS(decode(c, BA, true, c2, s2, &rA4)); CHK(zero_grad_of(c, rA4));   // trainer.py:113
    CHK(c.set_lane(L2));
    PASS(content_encode(c, AB, true, xb, &c4));                      // trainer.py:105
    PASS(style_encode(c, AB, true, xb, &s4));
    PASS(decode(c, AB, true, c4, s4, &rB4)); CHK(zero_grad_of(c, rB4));   // trainer.py:114
    // adversarial terms (trainer.py:136-139); discriminators frozen.  dis_B only needs x_B_fake: it is enqueued first; each pass runs its
    // full-resolution scale on one lane and the coarser scales on the other
    PASS(dis_lsgan(c, ACLGAN_NET_DIS_B, false, xB, B, {{1.f, 1.f, hp.gan_w, L + ACLGAN_L_GEN_ADV_B}}, L1, LS));
    PASS(dis_lsgan(c, ACLGAN_NET_DIS_A, false, jA, B, {{1.f, 0.5f, hp.gan_w, L + ACLGAN_L_GEN_ADV_A}, {1.f, 0.5f, hp.gan_w, L + ACLGAN_L_GEN_ADV_A}}, L0, L2));
    PASS(dis_lsgan(c, ACLGAN_NET_DIS_2, false, jP, B, {{1.f, 1.f, hp.gan_cw, L + ACLGAN_L_GEN_ADV_2},     // networks.py:98: pair_A1 -> 1
                                                        {0.f, 1.f, hp.gan_cw, L + ACLGAN_L_GEN_ADV_2}}, L1, LS));  //                 pair_A2 -> 0
    CHK(c.lanes_join());                                              // the loss kernels below read all of it, on lane 0
    // focus losses (trainer.py:145-161)
    const int64_t npix = (int64_t)B * H * W;
    const float fscale = hp.focus_loss / (float)H / (float)W / (float)B / 3.f;
    struct { Act* a; int size_slot; int digit_slot; } fl[3] = {
        {dB4, ACLGAN_L_GEN_FOCUS_B_SIZE, ACLGAN_L_GEN_FOCUS_B_DIGIT},
        {dA4, ACLGAN_L_GEN_FOCUS_A_SIZE, ACLGAN_L_GEN_FOCUS_A_DIGIT},
        {dA24, ACLGAN_L_GEN_FOCUS_A2_SIZE, ACLGAN_L_GEN_FOCUS_A2_DIGIT}};
    // data parallelism with the reference's GLOBAL-batch semantics of the focus losses (trainer.py:149-161: the size loss
    // squares a sum over the whole batch): the 6 sums are all-reduced by the caller at a forward sync point, every rank
    // then differentiates relu(T_global)^2 through its own pixels; the gradient keeps the LOCAL 1/(H W b 3) scale because
    // the gradient all-reduce averages over ranks
    float* ftot = nullptr;
    if (c.sync_fn) { ftot = c.allocf(8); NEED(ftot); }
    for (int i = 0; i < 3 && focus; ++i) RUN(focus_sums(fl[i].a->d, npix, hp.focus_epsilon, hp.focus_upper, sums + (size_t)i * nfs, c.st));
    if (c.mask_dst && !c.dry) {
        // diagnostics (aclgan_debug_capture_masks): the step's two other sign decisions -- |m - 0.5| of the focus digit losses (act code 100: the
        // sign of the whole 4-channel decoder output, channel 3 = 2 m - 1) and |x_recon - x| of the identity losses (act code 101, 3 channels)
        auto cap = [&](int code, int C_, size_t n) -> int {
            if (c.mask_top + n > c.mask_cap) { set_error("aclgan_debug_capture_masks: buffer too small (%zu bytes)", c.mask_cap); return ACLGAN_ENOMEM; }
            c.mask_log.push_back(aclgan_ctx::MaskEnt{B, H, W, C_, code, (long long)c.mask_top});
            c.mask_top += n;
            return ACLGAN_OK;
        };
        for (int i = 0; i < 3 && focus; ++i) {
            const size_t n = (size_t)fl[i].a->numel();
            unsigned char* dst = c.mask_dst + c.mask_top;
            CHK(cap(100, fl[i].a->C, n));
            RUN(positive_mask(fl[i].a->d, fl[i].a->dt, dst, (int64_t)n, c.st));
        }
        Act* rr[2] = {rA4, rB4}; Act* xx[2] = {xa, xb};
        for (int i = 0; i < 2; ++i) {
            unsigned char* dst = c.mask_dst + c.mask_top;
            CHK(cap(101, 3, (size_t)npix * 3));
            RUN(positive_mask_diff(rr[i]->d, rr[i]->C, xx[i]->d, 3, dst, npix, c.st));
        }
    }
    c.count(4.0 * (double)npix * ((focus ? 3 * 2 : 0) + 2 * (4 + 3 + 4)));     // focus masks read + gradient written; L1: decoder output, image, gradient
    if (ftot && focus) {
        RUN(focus_totals(sums, npix, 3, ftot, c.st));
        if (!c.dry) c.sync_fn(c.sync_user, ftot, 6);
    }
    for (int i = 0; i < 3 && focus; ++i)
        RUN(focus_loss_finish(fl[i].a->d, npix, sums + (size_t)i * nfs, hp.focus_delta, hp.focus_upper, hp.focus_lower, hp.focus_epsilon, fscale,
                              L + fl[i].size_slot, L + fl[i].digit_slot, fl[i].a->g, c.st, c.lscale, ftot ? ftot + 2 * i : nullptr,
                              npix * (int64_t)c.sync_world));
    // identity losses (trainer.py:162-165)
    {
        const size_t mark = c.top;
        float* l1p = c.allocf(2 * L1_PART_FLOATS);     // workgroup partials of the two L1 sums: added in a fixed order
        NEED(l1p);
        RUN(l1_loss(rA4->d, rA4->C, xa->d, npix, L + ACLGAN_L_IDT_A, rA4->g, hp.recon_x_w, 1, c.st, c.lscale, l1p));
        RUN(l1_loss(rB4->d, rB4->C, xb->d, npix, L + ACLGAN_L_IDT_B, rB4->g, hp.recon_x_w, 1, c.st, c.lscale, l1p + L1_PART_FLOATS));
        c.top = mark;
    }
    if (!c.dry) {
        hipLaunchKernelGGL(gen_total_kernel, dim3(1), dim3(1), 0, c.st, L, hp, ftot ? fscale / (float)c.sync_world : fscale);
        ACL_CHECK_LAUNCH("gen_total_kernel");
    }
    return run_tape(c);   // loss_gen_total.backward() (trainer.py:169)
}

// ---- dis_update (trainer.py:254-292) ----
static int dis_update_impl(aclgan_ctx& c, const float* x_a, const float* x_b, const float* z, int B, int H, int W,
                           const aclgan_hparams& hp, float* L) {
    CHK(check_shape(c, B, H, W));
    const bool focus = hp.focus_loss > 0.f;      // trainer.py:266-276: same two branches as gen_update
    ACL_REQUIRE(c.arch.gen_output_dim == (focus ? 4 : 3), "focus_loss %s 0 needs gen.output_dim %d (trainer.py:266-276), the context was built with %d",
                focus ? ">" : "<=", focus ? 4 : 3, c.arch.gen_output_dim);
    const int sd = c.arch.gen_style_dim;
    const int AB = ACLGAN_NET_GEN_AB, BA = ACLGAN_NET_GEN_BA;
    if (!c.dry) {
        hipError_t e = hipMemsetAsync(L + ACLGAN_L_DIS_A, 0, sizeof(float) * 4, c.st);
        if (e != hipSuccess) return hip_fail(e, "memset losses");
    }
    CHK(c.lanes_begin(lanes_setting()));
    CHK(pack_params(c));
    Act *xa, *xb, *z1, *z2, *z3;
    CHK(input_act(c, x_a, B, 3, H, W, &xa));
    CHK(input_act(c, x_b, B, 3, H, W, &xb));
    CHK(wrap_vec(c, z, B, sd, 1.f, &z1));
    CHK(wrap_vec(c, z + (size_t)B * sd, B, sd, hp.alpha, &z2));
    CHK(wrap_vec(c, z + (size_t)2 * B * sd, B, sd, 1.f, &z3));
    CHK(prefill_on_side_lane(c, B, H, W, false));
    Act *c1, *c2, *c3, *dB4, *dA4, *dA24, *xB, *xA, *xA2, *pA1, *pA2;
    // joint discriminator inputs: dis_A sees (x_A_fake | x_A2_fake | x_a), dis_B (x_B_fake | x_b), dis_2 (pair_A1 | pair_A2)
    Act* jA = c.new_act(3 * B, H, W, 3, false);
    Act* jB = c.new_act(2 * B, H, W, 3, false);
    Act* jP = c.new_act(2 * B, H, W, 6, false);
    NEED(jA->d); NEED(jB->d); NEED(jP->d);
    if (!c.dry) {
        hipError_t e = hipMemcpyAsync(jA->d + (size_t)2 * xa->numel(), xa->d, sizeof(float) * xa->numel(), hipMemcpyDeviceToDevice, c.st);
        if (e == hipSuccess) e = hipMemcpyAsync(jB->d + (size_t)xb->numel(), xb->d, sizeof(float) * xb->numel(), hipMemcpyDeviceToDevice, c.st);
        if (e != hipSuccess) return hip_fail(e, "copy real images into the joint discriminator batch");
    }
    // lanes as in gen_update: lane 0 = the chain x_a -> gen_AB -> x_B_fake -> gen_BA -> x_A2_fake, lane 1 = the other direction, then
    // dis_B (which needs only x_B_fake) next to the second half of the chain; a third lane takes the coarser discriminator scales
    const int L0 = 0, L1 = 1, L2 = c.nlanes > 2 ? 2 : 1, LS = c.nlanes > 2 ? 2 : 0;
    CHK(c.mark());
    CHK(c.set_lane(L0));
    PASS(content_encode(c, AB, false, xa, &c1));
    CHK(c.set_lane(L1));
    PASS(content_encode(c, BA, false, xa, &c2));
    CHK(c.set_lane(L0));
    PASS(decode(c, AB, false, c1, z1, &dB4));
    PASS(blend(c, dB4, xa, nullptr, &xB, nullptr, c.new_view(jB, 0, B), nullptr));
    CHK(c.set_lane(L1));
    PASS(decode(c, BA, false, c2, z2, &dA4));
    PASS(blend(c, dA4, xa, xa, &xA, &pA1, c.new_view(jA, 0, B), c.new_view(jP, 0, B)));
    // calc_dis_loss(fake -> 0, real -> 1) (networks.py:60-67; trainer.py:283-286)
    PASS(dis_lsgan(c, ACLGAN_NET_DIS_B, true, jB, B, {{0.f, 1.f, hp.gan_w, L + ACLGAN_L_DIS_B}, {1.f, 1.f, hp.gan_w, L + ACLGAN_L_DIS_B}}, L1, L2));
    CHK(c.set_lane(L0));
    PASS(content_encode(c, BA, false, xB, &c3));
    PASS(decode(c, BA, false, c3, z3, &dA24));
    PASS(blend(c, dA24, xB, xa, &xA2, &pA2, c.new_view(jA, B, B), c.new_view(jP, B, B)));
    PASS(dis_lsgan(c, ACLGAN_NET_DIS_A, true, jA, B, {{0.f, 0.5f, hp.gan_w, L + ACLGAN_L_DIS_A}, {0.f, 0.5f, hp.gan_w, L + ACLGAN_L_DIS_A},
                                                       {1.f, 1.0f, hp.gan_w, L + ACLGAN_L_DIS_A}}, L0, L2));   // the real branch occurs twice x 0.5
    PASS(dis_lsgan(c, ACLGAN_NET_DIS_2, true, jP, B, {{0.f, 1.f, hp.gan_cw, L + ACLGAN_L_DIS_2}, {1.f, 1.f, hp.gan_cw, L + ACLGAN_L_DIS_2}}, L1, LS));
    CHK(c.lanes_join());
    if (!c.dry) {
        hipLaunchKernelGGL(dis_total_kernel, dim3(1), dim3(1), 0, c.st, L, hp);
        ACL_CHECK_LAUNCH("dis_total_kernel");
    }
    return run_tape(c);   // loss_dis_total.backward() (trainer.py:292)
}
#undef PASS

}  // namespace aclgan

// ------------------------------------------------------------------------------------------
// C ABI (context-level entry points)
// ------------------------------------------------------------------------------------------
extern "C" {

int aclgan_ctx_create(const aclgan_arch* arch, aclgan_ctx** out) {
    ACL_REQUIRE(arch && out, "null argument");
    ACL_REQUIRE(arch->input_dim_a == 3 && arch->input_dim_b == 6, "input_dim_a must be 3 and input_dim_b 6 (trainer.py:19-23,132-133)");
    ACL_REQUIRE(arch->gen_output_dim == 4 || arch->gen_output_dim == 3, "gen.output_dim must be 4 (image + focus mask, trainer.py:108) or 3 (non-focus configuration, trainer.py:117-121)");
    ACL_REQUIRE(arch->gen_dim >= 4 && (arch->gen_dim & (arch->gen_dim - 1)) == 0, "gen.dim must be a power of two >= 4");
    ACL_REQUIRE(arch->dis_dim >= 4 && arch->dis_dim % 4 == 0, "dis.dim must be a multiple of 4");
    ACL_REQUIRE(arch->gen_n_downsample >= 1 && arch->gen_n_res >= 1 && arch->dis_n_layer >= 1 && arch->dis_num_scales >= 1, "bad layer counts");
    ACL_REQUIRE(arch->gen_mlp_dim >= 1 && arch->gen_style_dim >= 1, "bad mlp/style dims");
    aclgan_ctx* c = new aclgan_ctx();
    c->arch = *arch;
    build_gen(c->groups[0], "gen_AB", *arch);
    build_gen(c->groups[0], "gen_BA", *arch);
    build_dis(c->groups[1], "dis_A", arch->input_dim_a, *arch);
    build_dis(c->groups[1], "dis_B", arch->input_dim_a, *arch);
    build_dis(c->groups[1], "dis_2", arch->input_dim_b, *arch);
    for (int g = 0; g < 2; ++g)
        for (const TensorInfo& t : c->groups[g].tensors)
            if (t.ndim == 4) {
                c->cv_off[g].push_back(t.offset); c->cv_co[g].push_back(t.shape[0]);
                c->cv_taps[g].push_back(t.shape[2] * t.shape[3]); c->cv_ci[g].push_back(t.shape[1]);
            }
    *out = c;
    return ACLGAN_OK;
}

void aclgan_ctx_destroy(aclgan_ctx* ctx) { delete ctx; }

int aclgan_ctx_enable_capture(aclgan_ctx* ctx) {
    ACL_REQUIRE(ctx, "null ctx");
    return ctx->make_private_side();
}

// Round 6: create the process-wide streams of the lane scheduler (parameter-gradient stream, lanes 1 .. lanes - 1) on the CURRENT device NOW instead
// of at the first update.  HIP binds streams to its 4 hardware queues in creation order: a stream another component creates first -- RCCL's, a
// prefetching loader's -- takes a queue the lanes then have to share (measured with bench.py --pre-streams: one foreign stream first costs the
// 3-lane step 3.7 ms, two cost 6.9 ms; profiles/r06_experiments.md section 8).  aclgan_Trainer calls this right after aclgan_ctx_create, before its
// rank-0 broadcast initialises the process group's communicator.  Best effort: no device, nothing happens.
int aclgan_warm_streams(int lanes) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return ACLGAN_OK; }
    if (!aclgan_ctx::side_enabled()) return ACLGAN_OK;
    lanes = std::max(1, std::min(lanes <= 0 ? lanes_setting() : lanes, 3));
    hipStream_t s = nullptr;
    int rc = aclgan::StreamPool::of_device().get(0, &s);
    for (int l = 1; l < lanes && rc == ACLGAN_OK; ++l) rc = aclgan::StreamPool::of_device().get(l, &s);
    return rc;
}

int aclgan_debug_capture_masks(aclgan_ctx* ctx, unsigned char* dst, size_t cap_bytes) {
    ACL_REQUIRE(ctx, "null ctx");
    ctx->mask_dst = dst; ctx->mask_cap = dst ? cap_bytes : 0; ctx->mask_top = 0; ctx->mask_log.clear();
    return ACLGAN_OK;
}
int aclgan_debug_mask_count(const aclgan_ctx* ctx) { return ctx ? (int)ctx->mask_log.size() : 0; }
int aclgan_debug_mask_info(const aclgan_ctx* ctx, int index, int* dims4, long long* offset, int* act) {
    ACL_REQUIRE(ctx && index >= 0 && index < (int)ctx->mask_log.size(), "mask index out of range");
    const aclgan_ctx::MaskEnt& e = ctx->mask_log[index];
    if (dims4) { dims4[0] = e.B; dims4[1] = e.H; dims4[2] = e.W; dims4[3] = e.C; }
    if (offset) *offset = e.off;
    if (act) *act = e.act;
    return ACLGAN_OK;
}

int aclgan_set_compute_dtype(aclgan_ctx* ctx, int dtype) {
    ACL_REQUIRE(ctx && dtype >= ACLGAN_DTYPE_FP32 && dtype <= ACLGAN_DTYPE_FP16, "bad ctx / dtype %d", dtype);
    ctx->dtype = dtype;
    return ACLGAN_OK;
}
int aclgan_bind_params16(aclgan_ctx* ctx, int group, void* w16, void* w16t) {
    ACL_REQUIRE(ctx && group >= 0 && group <= 1, "bad ctx/group");
    ctx->w16[group] = (unsigned short*)w16; ctx->w16t[group] = (unsigned short*)w16t;
    return ACLGAN_OK;
}
int aclgan_bind_loss_scale(aclgan_ctx* ctx, float* state) {
    ACL_REQUIRE(ctx, "null ctx");
    ctx->lscale = state;
    return ACLGAN_OK;
}

int64_t aclgan_group_numel(const aclgan_ctx* ctx, int group) {
    if (!ctx || group < 0 || group > 1) return -1;
    return ctx->groups[group].numel;
}
int aclgan_tensor_count(const aclgan_ctx* ctx, int group) {
    if (!ctx || group < 0 || group > 1) return -1;
    return (int)ctx->groups[group].tensors.size();
}
int aclgan_tensor_info(const aclgan_ctx* ctx, int group, int index, char* name, int name_cap, int64_t* offset, int* shape4, int* ndim) {
    ACL_REQUIRE(ctx && group >= 0 && group <= 1, "bad ctx/group");
    const Group& g = ctx->groups[group];
    ACL_REQUIRE(index >= 0 && index < (int)g.tensors.size(), "tensor index %d out of range", index);
    const TensorInfo& t = g.tensors[index];
    if (name && name_cap > 0) { strncpy(name, t.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (offset) *offset = t.offset;
    if (shape4) for (int i = 0; i < 4; ++i) shape4[i] = t.shape[i];
    if (ndim) *ndim = t.ndim;
    return ACLGAN_OK;
}
int aclgan_bind_params(aclgan_ctx* ctx, int group, float* param, float* grad, float* exp_avg, float* exp_avg_sq) {
    ACL_REQUIRE(ctx && group >= 0 && group <= 1, "bad ctx/group");
    ACL_REQUIRE(param, "param buffer is null");
    Group& g = ctx->groups[group];
    g.param = param; g.grad = grad; g.m = exp_avg; g.v = exp_avg_sq;
    return ACLGAN_OK;
}

// bytes one update needs (which: 0 gen_update, 1 dis_update) with the switches as they are now: a dry run of the same scheduler
static int update_need_bytes(aclgan_ctx& c, int which, int B, int H, int W, size_t* out) {
    aclgan_hparams hp;
    memset(&hp, 0, sizeof hp);
    hp.focus_loss = c.arch.gen_output_dim == 4 ? 1.f : 0.f; hp.alpha = 1.f;      // (the branch the architecture can run: gen.output_dim 4 = focus, 3 = non-focus)
    c.reset_step();
    c.dry = true; c.peak = 0; c.peak2 = 0; c.trained = -1;
    const aclgan_bucket_fn keep = c.bucket_fn;
    c.bucket_fn = nullptr;
    const int rc = which == 0 ? gen_update_impl(c, nullptr, nullptr, nullptr, B, H, W, hp, nullptr)
                              : dis_update_impl(c, nullptr, nullptr, nullptr, B, H, W, hp, nullptr);
    c.bucket_fn = keep;
    c.dry = false; c.trained = -1;
    *out = c.peak + c.peak2 + 512;
    c.reset_step();
    c.peak = 0; c.peak2 = 0;      // a dry run leaves no allocator state behind (a stale side-stack mark made later forward-only calls fail spuriously)
    return rc;
}
int aclgan_workspace_bytes(aclgan_ctx* ctx, int B, int H, int W, size_t* out) {
    ACL_REQUIRE(ctx && out, "null argument");
    aclgan_ctx& c = *ctx;
    ACL_REQUIRE(c.groups[0].param && c.groups[1].param, "bind parameters first");
    size_t best = 0;
    for (int which = 0; which < 2; ++which) {
        size_t pk = 0;
        const int rc = update_need_bytes(c, which, B, H, W, &pk);
        if (rc) return rc;
        if (pk > best) best = pk;
    }
    *out = best + 4096;
    return ACLGAN_OK;
}
// ACLGAN_OK iff the bound workspace holds both updates at this shape with the switches as they are now; ACLGAN_ENOMEM (with the two sizes
// in the message) otherwise.  No launches: callable without a GPU.  The update entry points make the same check themselves (once per
// shape and switch setting) before they enqueue anything.
int aclgan_check_workspace(aclgan_ctx* ctx, int B, int H, int W) {
    ACL_REQUIRE(ctx, "null ctx");
    ACL_REQUIRE(ctx->groups[0].param && ctx->groups[1].param, "bind parameters first");
    size_t need = 0;
    const int rc = aclgan_workspace_bytes(ctx, B, H, W, &need);
    if (rc) return rc;
    if (!ctx->ws || ctx->ws_bytes < need) {
        set_error("workspace too small: bound %zu bytes, aclgan_workspace_bytes(%d, %d, %d) = %zu", ctx->ws ? ctx->ws_bytes : (size_t)0, B, H, W, need);
        return ACLGAN_ENOMEM;
    }
    return ACLGAN_OK;
}
int aclgan_step_algorithmic_bytes(aclgan_ctx* ctx, int which, int B, int H, int W, double* out) {
    ACL_REQUIRE(ctx && out && (which == 0 || which == 1), "bad argument");
    aclgan_ctx& c = *ctx;
    ACL_REQUIRE(c.groups[0].param && c.groups[1].param, "bind parameters first");
    aclgan_hparams hp;
    memset(&hp, 0, sizeof hp);
    hp.focus_loss = c.arch.gen_output_dim == 4 ? 1.f : 0.f; hp.alpha = 1.f;      // (the branch the architecture can run: gen.output_dim 4 = focus, 3 = non-focus)
    c.reset_step();
    c.dry = true; c.peak = 0; c.trained = which; c.alg_bytes = 0.0; c.exec_flops = 0.0;
    const aclgan_bucket_fn keep = c.bucket_fn;
    c.bucket_fn = nullptr;
    const int rc = which == 0 ? gen_update_impl(c, nullptr, nullptr, nullptr, B, H, W, hp, nullptr)
                              : dis_update_impl(c, nullptr, nullptr, nullptr, B, H, W, hp, nullptr);
    c.bucket_fn = keep;
    c.dry = false; c.trained = -1;
    c.reset_step();
    c.peak = 0; c.peak2 = 0;
    if (rc) return rc;
    *out = c.alg_bytes + (4.0 + 28.0) * (double)c.groups[which].numel;    // + zero_grad + Adam (p, g, m, v read; p, m, v written)
    return ACLGAN_OK;
}
// the matrix-pipe FLOPs of the same update as the kernels EXECUTE them (conv_exec_flops: the cost of the path every layer's launchers choose
// at this shape, dtype and switch setting); a dry run, no GPU needed.  Convolutions only (the dense layers of the MLP are 1e-5 of the step).
int aclgan_step_executed_flops(aclgan_ctx* ctx, int which, int B, int H, int W, double* out) {
    double bytes = 0.0;
    const int rc = aclgan_step_algorithmic_bytes(ctx, which, B, H, W, &bytes);
    if (rc) return rc;
    *out = ctx->exec_flops;
    return ACLGAN_OK;
}
// workspace of ONE forward-only call (aclgan_gen_encode / aclgan_gen_decode / aclgan_dis_forward) at this image shape:
// inference (test.py, sample()) needs neither the training arena nor the training step's shape constraints
int aclgan_forward_workspace_bytes(aclgan_ctx* ctx, int B, int H, int W, size_t* out) {
    ACL_REQUIRE(ctx && out, "null argument");
    aclgan_ctx& c = *ctx;
    ACL_REQUIRE(c.groups[0].param && c.groups[1].param, "bind parameters first");
    ACL_REQUIRE(B >= 1 && H >= 4 && W >= 4, "bad shape (%d,%d,%d)", B, H, W);
    const int q = 1 << c.arch.gen_n_downsample, C = c.arch.gen_dim << c.arch.gen_n_downsample;
    size_t best = 0;
    for (int which = 0; which < 3; ++which) {
        c.reset_step();
        c.dry = true; c.peak = 0; c.peak2 = 0; c.trained = -1;
        Act *x = nullptr, *o = nullptr, *s = nullptr;
        std::vector<Act*> outs;
        int rc = ACLGAN_OK;
        if (which == 0) {
            rc = input_act(c, nullptr, B, 3, H, W, &x);
            if (!rc) rc = content_encode(c, ACLGAN_NET_GEN_AB, false, x, &o);
            if (!rc && o && o->dt) c.allocf(o->numel());      // fp32 copy of a 16-bit content code (aclgan_gen_encode)
            if (!rc) rc = style_encode(c, ACLGAN_NET_GEN_AB, false, x, &s);
        } else if (which == 1) {
            rc = input_act(c, nullptr, B, C, std::max(1, H / q), std::max(1, W / q), &x);
            if (!rc) rc = wrap_vec(c, nullptr, B, c.arch.gen_style_dim, 1.f, &s);
            if (!rc) rc = decode(c, ACLGAN_NET_GEN_AB, false, x, s, &o);
        } else if (H >= 64 && W >= 64) {   // smaller images cannot pass through the third discriminator scale at all
            rc = input_act(c, nullptr, B, c.arch.input_dim_b, H, W, &x);
            if (!rc) rc = dis_forward(c, ACLGAN_NET_DIS_2, false, x, &outs);
        }
        c.dry = false;
        const size_t pk = c.peak;
        c.reset_step();
        if (rc) return rc;
        if (pk > best) best = pk;
    }
    *out = best + 4096;
    return ACLGAN_OK;
}
int aclgan_bind_workspace(aclgan_ctx* ctx, void* workspace, size_t bytes) {
    ACL_REQUIRE(ctx, "null ctx");
    ctx->ws = (char*)workspace; ctx->ws_bytes = bytes;
    return ACLGAN_OK;
}

static int step_common(aclgan_ctx* ctx, const float* x_a, const float* x_b, const float* z, const aclgan_hparams* hp, float* losses, void* stream, int group_trained,
                       int B, int H, int W) {
    ACL_REQUIRE(ctx && x_a && x_b && z && hp && losses, "null argument");
    ACL_REQUIRE(ctx->ws, "bind a workspace first (aclgan_workspace_bytes / aclgan_bind_workspace)");
    ACL_REQUIRE(ctx->groups[0].param && ctx->groups[1].param, "bind parameters first");
    ACL_REQUIRE(ctx->groups[group_trained].grad, "gradient buffer of the trained group is not bound");
    // the bound workspace against this update's need (a dry run, cached per shape / dtype / switch setting): an undersized workspace is
    // refused here, before anything is enqueued (the allocator checks every request as well)
    if (check_shape(*ctx, B, H, W) == ACLGAN_OK) {
        const aclgan_ctx::NeedKey key{group_trained, B, H, W, ctx->dtype, tuning_epoch(), deterministic() ? 1 : 0, ctx->bucket_elems > 0 ? 1 : 0};
        auto it = ctx->need_cache.find(key);
        if (it == ctx->need_cache.end()) {
            size_t need = 0;
            const int rc = update_need_bytes(*ctx, group_trained, B, H, W, &need);
            if (rc) return rc;
            it = ctx->need_cache.emplace(key, need).first;
        }
        if (it->second > ctx->ws_bytes) {
            set_error("workspace too small: bound %zu bytes, this update needs %zu (aclgan_workspace_bytes)", ctx->ws_bytes, it->second);
            return ACLGAN_ENOMEM;
        }
    }
    ctx->reset_step();
    ctx->st = (hipStream_t)stream; ctx->dry = false; ctx->peak = 0; ctx->peak2 = 0; ctx->trained = group_trained;
    return ACLGAN_OK;
}

int aclgan_gen_update(aclgan_ctx* ctx, const float* x_a, const float* x_b, const float* z, int B, int H, int W,
                      const aclgan_hparams* hp, float* losses, void* stream) {
    int rc = step_common(ctx, x_a, x_b, z, hp, losses, stream, 0, B, H, W);
    if (rc) return rc;
    ctx->ucache_hook = WinoUCache{ctx, &aclgan_ctx::ucache_lookup};
    set_wino_ucache(&ctx->ucache_hook);
    rc = gen_update_impl(*ctx, x_a, x_b, z, B, H, W, *hp, losses);
    set_wino_ucache(nullptr);
    if (rc) ctx->lanes_quiesce();      // (an error in the middle of the forward: lanes may hold work)
    ctx->reset_step();
    return rc;
}
int aclgan_dis_update(aclgan_ctx* ctx, const float* x_a, const float* x_b, const float* z, int B, int H, int W,
                      const aclgan_hparams* hp, float* losses, void* stream) {
    int rc = step_common(ctx, x_a, x_b, z, hp, losses, stream, 1, B, H, W);
    if (rc) return rc;
    ctx->ucache_hook = WinoUCache{ctx, &aclgan_ctx::ucache_lookup};
    set_wino_ucache(&ctx->ucache_hook);
    rc = dis_update_impl(*ctx, x_a, x_b, z, B, H, W, *hp, losses);
    set_wino_ucache(nullptr);
    if (rc) ctx->lanes_quiesce();
    ctx->reset_step();
    return rc;
}

int aclgan_set_forward_sync(aclgan_ctx* ctx, aclgan_sync_fn fn, void* user, int world_size) {
    ACL_REQUIRE(ctx && world_size >= 1, "bad ctx / world size");
    ctx->sync_fn = fn; ctx->sync_user = user; ctx->sync_world = fn ? world_size : 1;
    return ACLGAN_OK;
}

int aclgan_set_grad_buckets(aclgan_ctx* ctx, int64_t bucket_elems, aclgan_bucket_fn fn, void* user) {
    ACL_REQUIRE(ctx && bucket_elems >= 0, "bad ctx/bucket size");
    ctx->bucket_elems = bucket_elems; ctx->bucket_fn = fn; ctx->bucket_user = user;
    return ACLGAN_OK;
}

int aclgan_bucket_schedule(aclgan_ctx* ctx, int group, int B, int H, int W, int fire, int* order, int cap, int* count) {
    ACL_REQUIRE(ctx && count && group >= 0 && group <= 1, "bad argument");
    aclgan_ctx& c = *ctx;
    ACL_REQUIRE(c.groups[0].param && c.groups[1].param && c.groups[group].grad, "bind parameters (and the group's gradient buffer) first");
    ACL_REQUIRE(c.bucket_elems > 0, "aclgan_set_grad_buckets first");
    aclgan_hparams hp;
    memset(&hp, 0, sizeof hp);
    hp.focus_loss = c.arch.gen_output_dim == 4 ? 1.f : 0.f; hp.alpha = 1.f;      // (the branch the architecture can run: gen.output_dim 4 = focus, 3 = non-focus)
    c.reset_step();
    c.dry = true; c.peak = 0; c.trained = group; c.fire_dry = fire != 0;
    const int rc = group == 0 ? gen_update_impl(c, nullptr, nullptr, nullptr, B, H, W, hp, nullptr)
                              : dis_update_impl(c, nullptr, nullptr, nullptr, B, H, W, hp, nullptr);
    c.dry = false; c.fire_dry = false; c.trained = -1;
    c.reset_step();
    c.peak = 0; c.peak2 = 0;
    if (rc) return rc;
    *count = (int)c.bucket_order.size();
    for (int i = 0; i < *count && i < cap && order; ++i) order[i] = c.bucket_order[i];
    return ACLGAN_OK;
}

int aclgan_zero_grad(aclgan_ctx* ctx, int group, void* stream) {
    ACL_REQUIRE(ctx && group >= 0 && group <= 1, "bad ctx/group");
    Group& g = ctx->groups[group];
    ACL_REQUIRE(g.grad, "gradient buffer not bound");
    return fill_zero(g.grad, g.numel, (hipStream_t)stream);
}
int aclgan_adam_step(aclgan_ctx* ctx, int group, const aclgan_adam* opt, int step, void* stream) {
    ACL_REQUIRE(ctx && opt && group >= 0 && group <= 1, "bad ctx/group/opt");
    Group& g = ctx->groups[group];
    ACL_REQUIRE(g.param && g.grad && g.m && g.v, "param/grad/exp_avg/exp_avg_sq must all be bound");
    if (ctx->lscale) return adam_flat_scaled(g.param, g.grad, g.m, g.v, g.numel, opt, step, ctx->lscale, group, (hipStream_t)stream);
    return adam_flat(g.param, g.grad, g.m, g.v, g.numel, opt, step, (hipStream_t)stream);
}

// ---- forward-only entry points ----
static int fwd_begin(aclgan_ctx* ctx, void* stream) {
    ACL_REQUIRE(ctx && ctx->ws, "bind a workspace first");
    ACL_REQUIRE(ctx->groups[0].param && ctx->groups[1].param, "bind parameters first");
    ctx->reset_step();
    // (peak2: the side stack's high-water mark of whatever ran before -- an update, a dry run -- is not this call's: a forward-only arena has no side stack)
    ctx->st = (hipStream_t)stream; ctx->dry = false; ctx->peak = 0; ctx->peak2 = 0; ctx->trained = -1;
    return ACLGAN_OK;
}

int aclgan_gen_encode(aclgan_ctx* ctx, int net, const float* x, int B, int H, int W, float* content, float* style, void* stream) {
    int rc = fwd_begin(ctx, stream);
    if (rc) return rc;
    ACL_REQUIRE(net == ACLGAN_NET_GEN_AB || net == ACLGAN_NET_GEN_BA, "encode: net must be a generator");
    aclgan_ctx& c = *ctx;
    rc = pack_params(c);
    if (rc) return rc;
    Act *xa = nullptr, *cc = nullptr, *ss = nullptr;
    rc = input_act(c, x, B, 3, H, W, &xa);
    if (!rc && content) {
        rc = content_encode(c, net, false, xa, &cc);
        const float* src = cc ? cc->d : nullptr;
        if (!rc && cc->dt) {      // the content code lives in HBM in the 16-bit dtype: the public surface hands out fp32 (test.py:58-70)
            float* tmp = c.allocf(cc->numel());
            if (!tmp) { set_error("workspace too small"); rc = ACLGAN_ENOMEM; }
            else { rc = cast_storage(cc->d, cc->dt, tmp, 0, cc->numel(), c.st); src = tmp; }
        }
        if (!rc) rc = nhwc_to_nchw(src, content, cc->B, cc->C, cc->H, cc->W, c.st);
    }
    if (!rc && style) {
        rc = style_encode(c, net, false, xa, &ss);
        if (!rc) { hipError_t e = hipMemcpyAsync(style, ss->d, sizeof(float) * ss->numel(), hipMemcpyDeviceToDevice, c.st); if (e != hipSuccess) rc = hip_fail(e, "copy style"); }
    }
    c.reset_step();
    return rc;
}

int aclgan_gen_decode(aclgan_ctx* ctx, int net, const float* content, const float* style, int B, int h, int w, float* out, void* stream) {
    int rc = fwd_begin(ctx, stream);
    if (rc) return rc;
    ACL_REQUIRE(net == ACLGAN_NET_GEN_AB || net == ACLGAN_NET_GEN_BA, "decode: net must be a generator");
    aclgan_ctx& c = *ctx;
    rc = pack_params(c);
    if (rc) return rc;
    const int C = c.arch.gen_dim << c.arch.gen_n_downsample;
    Act *cc = nullptr, *ss = nullptr, *o = nullptr;
    rc = input_act(c, content, B, C, h, w, &cc);
    if (!rc) rc = wrap_vec(c, style, B, c.arch.gen_style_dim, 1.f, &ss);
    if (!rc) rc = decode(c, net, false, cc, ss, &o);
    if (!rc) rc = nhwc_to_nchw(o->d, out, o->B, o->C, o->H, o->W, c.st);
    c.reset_step();
    return rc;
}

int aclgan_dis_forward(aclgan_ctx* ctx, int net, const float* x, int B, int H, int W, float* const* outs, void* stream) {
    int rc = fwd_begin(ctx, stream);
    if (rc) return rc;
    ACL_REQUIRE(net >= ACLGAN_NET_DIS_A && net <= ACLGAN_NET_DIS_2, "dis_forward: net must be a discriminator");
    aclgan_ctx& c = *ctx;
    rc = pack_params(c);
    if (rc) return rc;
    const int Cin = net == ACLGAN_NET_DIS_2 ? c.arch.input_dim_b : c.arch.input_dim_a;
    Act* xa = nullptr;
    std::vector<Act*> o;
    rc = input_act(c, x, B, Cin, H, W, &xa);
    if (!rc) rc = dis_forward(c, net, false, xa, &o);
    for (size_t s = 0; !rc && s < o.size(); ++s) {
        hipError_t e = hipMemcpyAsync(outs[s], o[s]->d, sizeof(float) * o[s]->numel(), hipMemcpyDeviceToDevice, c.st);   // C == 1: NHWC == NCHW
        if (e != hipSuccess) rc = hip_fail(e, "copy dis out");
    }
    c.reset_step();
    return rc;
}

}  // extern "C"
