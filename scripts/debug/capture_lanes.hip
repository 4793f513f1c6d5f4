// Minimal reproductions for the round-5 crash "hipStreamEndCapture segfaults when an update with pooled lanes is captured"
// (acl-gan_amd/csrc/engine.hip lanes_begin: a captured update falls back to one lane).  Every variant builds the stream / event pattern the
// lane scheduler produces, under stream capture of an origin stream, in a CHILD process (a segfault is reported, not fatal):
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/capture_lanes scripts/debug/capture_lanes.hip && /tmp/capture_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <unistd.h>
#include <sys/wait.h>

__global__ void k(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("    %s -> %s\n", #x, hipGetErrorString(e_)); fflush(stdout); return 1; } } while (0)

static int run(int v, hipStreamCaptureMode mode) {
    const int N = 1 << 16;
    float* buf = nullptr;
    CK(hipMalloc(&buf, 8 * N * sizeof(float)));
    CK(hipMemset(buf, 0, 8 * N * sizeof(float)));
    hipStream_t origin, s[4];
    CK(hipStreamCreateWithFlags(&origin, hipStreamNonBlocking));
    for (int i = 0; i < 4; ++i) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(4096);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    int ne = 0;
    auto K = [&](hipStream_t st, int slot) { hipLaunchKernelGGL(k, dim3(N / 256), dim3(256), 0, st, buf + slot * N, N); return hipGetLastError(); };
    auto edge = [&](hipStream_t from, hipStream_t to) -> hipError_t { hipEvent_t e = ev[ne++]; hipError_t r = hipEventRecord(e, from); if (r != hipSuccess) return r; return hipStreamWaitEvent(to, e, 0); };
    if (v == 7) { CK(K(s[1], 1)); CK(hipEventRecord(ev[4000], s[1])); }      // an event recorded BEFORE the capture begins (eager work of the previous step)
    CK(hipStreamBeginCapture(origin, mode));
    if (v == 4) CK(hipMemsetAsync(buf + 7 * N, 0, N * sizeof(float), origin));
    CK(K(origin, 0));
    switch (v) {
    case 1: case 4:      // plain fork / join over three lanes
        for (int l = 0; l < 3; ++l) { CK(edge(origin, s[l])); CK(K(s[l], l + 1)); }
        for (int l = 0; l < 3; ++l) CK(edge(s[l], origin));
        break;
    case 2:      // a lane that only ever receives a wait, joined again
        CK(edge(origin, s[0])); CK(K(s[0], 1)); CK(edge(origin, s[1]));
        CK(edge(s[0], origin)); CK(edge(s[1], origin));
        break;
    case 3:      // ... and NOT joined again (must be an error code, not a crash)
        CK(edge(origin, s[0])); CK(K(s[0], 1)); CK(edge(origin, s[1]));
        CK(edge(s[0], origin));
        break;
    case 5: {    // ONE event pair re-recorded for every fork / join of a side stream (ev_fork / ev_join of the engine), 200 times
        hipEvent_t f = ev[ne++], j = ev[ne++];
        for (int i = 0; i < 200; ++i) { CK(K(origin, 0)); CK(hipEventRecord(f, origin)); CK(hipStreamWaitEvent(s[3], f, 0)); CK(K(s[3], 4)); }
        CK(hipEventRecord(j, s[3])); CK(hipStreamWaitEvent(origin, j, 0));
        break; }
    case 6:      // cross-lane waits, several waiters on one event, a lane waiting for an OLD checkpoint of another lane
        CK(edge(origin, s[0])); CK(edge(origin, s[1])); CK(K(s[0], 1)); CK(K(s[1], 2));
        { hipEvent_t e = ev[ne++]; CK(hipEventRecord(e, s[0])); CK(K(s[0], 1)); CK(hipStreamWaitEvent(s[1], e, 0)); CK(hipStreamWaitEvent(s[2], e, 0)); CK(hipStreamWaitEvent(origin, e, 0)); }
        CK(K(s[2], 3)); CK(K(s[1], 2));
        for (int l = 0; l < 3; ++l) CK(edge(s[l], origin));
        break;
    case 7:      // a captured stream waits for an event recorded before the capture
        CK(hipStreamWaitEvent(origin, ev[4000], 0)); CK(edge(origin, s[0])); CK(K(s[0], 2)); CK(edge(s[0], origin));
        break;
    case 8: {    // the size of a real update: 2400 kernels over the origin, three lanes and a side stream, a cross edge every 8 launches
        //              (CAP_N / CAP_E: number of kernels / launches per cross edge, for the sweep at the end of main)
        const int nk = getenv("CAP_N") ? atoi(getenv("CAP_N")) : 2400, ep = getenv("CAP_E") ? atoi(getenv("CAP_E")) : 8;
        const int ns = getenv("CAP_S") ? atoi(getenv("CAP_S")) : 4, nt = ns + 1;      // side streams (the origin is stream index ns)
        for (int l = 0; l < ns; ++l) CK(edge(origin, s[l]));
        for (int i = 0; i < nk; ++i) {
            const int l = i % nt; hipStream_t st = l == ns ? origin : s[l];
            CK(K(st, l));
            if (i % ep == ep - 1) { const int t = (i / ep) % nt; hipStream_t to = t == ns ? origin : s[t]; if (to != st) CK(edge(st, to)); }
        }
        for (int l = 0; l < ns; ++l) CK(edge(s[l], origin));
        break; }
    case 9: {    // the join pattern of lanes_join + side_join, then MORE work and a second join (bucket hand-outs in the middle of the backward)
        for (int r = 0; r < 6; ++r) {
            for (int l = 0; l < 4; ++l) { CK(edge(origin, s[l])); CK(K(s[l], l + 1)); }
            for (int l = 0; l < 4; ++l) CK(edge(s[l], origin));
            CK(K(origin, 0));
        }
        break; }
    case 10: {   // CAP_S side streams exchanging events in a RING (s0 -> s1 -> ... -> s0), CAP_N rounds: the smallest pattern of the sweep's failures
        const int ns = getenv("CAP_S") ? atoi(getenv("CAP_S")) : 3, rounds = getenv("CAP_N") ? atoi(getenv("CAP_N")) : 1;
        for (int l = 0; l < ns; ++l) { CK(edge(origin, s[l])); CK(K(s[l], l + 1)); }
        for (int r = 0; r < rounds; ++r)
            for (int l = 0; l < ns; ++l) { CK(edge(s[l], s[(l + 1) % ns])); CK(K(s[(l + 1) % ns], (l + 1) % ns + 1)); }
        for (int l = 0; l < ns; ++l) CK(edge(s[l], origin));
        break; }
    case 11: {   // random cross edges among the origin and CAP_S side streams (seed CAP_E), CAP_N kernels, one edge per 6 launches
        const int ns = getenv("CAP_S") ? atoi(getenv("CAP_S")) : 3, nk = getenv("CAP_N") ? atoi(getenv("CAP_N")) : 600;
        unsigned int rng = 12345u + (getenv("CAP_E") ? atoi(getenv("CAP_E")) : 0) * 7919u;
        auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return (rng >> 10); };
        for (int l = 0; l < ns; ++l) CK(edge(origin, s[l]));
        for (int i = 0; i < nk; ++i) {
            const int l = rnd() % (ns + 1); hipStream_t st = l == ns ? origin : s[l];
            CK(K(st, l));
            if (i % 6 == 5) { const int a = rnd() % (ns + 1), b = rnd() % (ns + 1); if (a != b) CK(edge(a == ns ? origin : s[a], b == ns ? origin : s[b])); }
        }
        for (int l = 0; l < ns; ++l) CK(edge(s[l], origin));
        break; }
    default: break;
    }
    CK(K(origin, 0));
    hipGraph_t g = nullptr;
    printf("    hipStreamEndCapture ..."); fflush(stdout);
    CK(hipStreamEndCapture(origin, &g));
    hipGraphExec_t ge = nullptr;
    printf(" hipGraphInstantiate ..."); fflush(stdout);
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    printf(" hipGraphLaunch ...\n"); fflush(stdout);
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, origin));
    CK(hipStreamSynchronize(origin));
    size_t nodes = 0; CK(hipGraphGetNodes(g, nullptr, &nodes));
    printf("    ok: %zu nodes, 3 replays\n", nodes);
    return 0;
}

static void child(int v, hipStreamCaptureMode mode) {
    pid_t pid = fork();
    if (pid == 0) { const int rc = run(v, mode); fflush(stdout); _exit(rc); }
    int status = 0; waitpid(pid, &status, 0);
    if (WIFSIGNALED(status)) printf("    CRASHED: signal %d\n", WTERMSIG(status));
    else if (WEXITSTATUS(status)) printf("    failed (rc %d)\n", WEXITSTATUS(status));
    fflush(stdout);
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "rings")) {      // which ring sizes / round counts / random patterns survive hipStreamEndCapture?
        for (int st = 2; st <= 4; ++st)
            for (int rounds : {1, 2, 8, 64}) {
                char a[32], c[32]; snprintf(a, sizeof a, "%d", rounds); snprintf(c, sizeof c, "%d", st);
                setenv("CAP_N", a, 1); setenv("CAP_S", c, 1);
                printf("[ring] %d side streams, %d rounds\n", st, rounds); fflush(stdout);
                child(10, hipStreamCaptureModeThreadLocal);
            }
        for (int st = 2; st <= 4; ++st)
            for (int seed = 0; seed < 4; ++seed) {
                char b[32], c[32]; snprintf(b, sizeof b, "%d", seed); snprintf(c, sizeof c, "%d", st);
                setenv("CAP_N", "600", 1); setenv("CAP_E", b, 1); setenv("CAP_S", c, 1);
                printf("[random] origin + %d side streams, 600 kernels, 100 random cross edges, seed %d\n", st, seed); fflush(stdout);
                child(11, hipStreamCaptureModeThreadLocal);
            }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "sweep")) {      // how large / how connected may a multi-stream capture be?
        const int ns[] = {100, 400, 2400}, es[] = {4, 8, 32, 128, 100000};
        for (int st = 1; st <= 4; ++st)
            for (int e : es)
                for (int n : ns) {
                    char a[32], b[32], c[32]; snprintf(a, sizeof a, "%d", n); snprintf(b, sizeof b, "%d", e); snprintf(c, sizeof c, "%d", st);
                    setenv("CAP_N", a, 1); setenv("CAP_E", b, 1); setenv("CAP_S", c, 1);
                    printf("[sweep] %d kernels over the origin + %d streams, one cross edge per %d launches\n", n, st, e); fflush(stdout);
                    child(8, hipStreamCaptureModeThreadLocal);
                }
        return 0;
    }
    const char* names[] = {"", "fork/join 3 lanes", "wait-only lane, rejoined", "wait-only lane, NOT rejoined (expect an error code)", "memset on the origin before the fork",
                           "one event pair re-recorded 200 times", "cross-lane waits, shared event", "wait on an event recorded before the capture",
                           "2400 kernels, 5 streams, 300 cross edges", "six fork/join rounds with work in between"};
    for (int mode = 0; mode < 2; ++mode)
        for (int v = 1; v <= 9; ++v) {
            printf("[%s] variant %d: %s\n", mode == 0 ? "global" : "thread-local", v, names[v]); fflush(stdout);
            pid_t pid = fork();
            if (pid == 0) { const int rc = run(v, mode == 0 ? hipStreamCaptureModeGlobal : hipStreamCaptureModeThreadLocal); fflush(stdout); _exit(rc); }
            int status = 0; waitpid(pid, &status, 0);
            if (WIFSIGNALED(status)) printf("    CRASHED: signal %d\n", WTERMSIG(status));
            else if (WEXITSTATUS(status)) printf("    failed (rc %d)\n", WEXITSTATUS(status));
            fflush(stdout);
        }
    return 0;
}
