// coresidency.hip -- what a second HIP stream can and cannot overlap on gfx950 (round 4).
// A memory-bound kernel (streaming copy: every wave slot of the chip, or half of them) on stream 1, an fp32-MFMA-bound kernel on stream 2, timed
// alone and together:
//   small  : full-chip MFMA kernel with few registers (its workgroups could share CUs with the copy's waves)
//   big    : the same MFMA work in a kernel that OWNS the register file (256 VGPR + 256 AGPR per lane, one wave per SIMD), like the fused
//            Winograd kernels
//   narrow : an under-filled launch (32 workgroups)
// Hypothesis when written: co-residency of memory-bound and matrix-bound kernels is what a second stream buys (copy || small ~ max, copy || big
// ~ sum).  MEASURED, the opposite: copy || small hides 0 - 3 % (two full-chip kernels time-share the chip, and the copy's own VALU instructions
// queue behind fp32 MFMAs), copy || big 20 - 35 % (it takes over CUs as they drain), copy || narrow 84 - 89 %: a second stream hides
// under-filled launches (profiles/r04_microbench_coresidency.txt).
//   hipcc --offload-arch=gfx950 -O3 -o coresidency coresidency.hip && ./coresidency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) copy_kernel(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 v = a[i];
        v.x *= 1.0001f;
        b[i] = v;
    }
}

template <int BIG>
__global__ void __launch_bounds__(256) mfma_kernel(float* __restrict__ out, int iters) {
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
    }
    if (BIG) asm volatile("" ::: "v250", "a250");      // reserves the whole register file: one wave per SIMD
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[3];
}

int main() {
    const size_t bytes = (size_t)1 << 30;             // 1 GiB read + 1 GiB written per copy launch
    float4 *a, *b; float* out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&out, 1024 * 256 * 4));
    CK(hipMemset(a, 0, bytes));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const int reps = 20, iters_big = 4000;            // big: 256 workgroups x 4 waves x 2 x 4000 MFMAs; small: 512 workgroups, half the iterations
    int copy_grid = 4096;                             // 4096: the copy's waves take every wave slot of the chip; 1024: half of them (4 workgroups per CU)
    auto copy = [&]() { hipLaunchKernelGGL(copy_kernel, dim3(copy_grid), dim3(256), 0, s1, a, b, bytes / 16); };
    auto narrow = [&]() { hipLaunchKernelGGL(mfma_kernel<0>, dim3(32), dim3(256), 0, s2, out, iters_big / 2); };      // an under-filled launch: 32 of 256 CUs
    auto small = [&]() { hipLaunchKernelGGL(mfma_kernel<0>, dim3(512), dim3(256), 0, s2, out, iters_big / 2); };
    auto big = [&]() { hipLaunchKernelGGL(mfma_kernel<1>, dim3(256), dim3(256), 0, s2, out, iters_big); };
    auto timeit = [&](bool c, int which) {
        double best = 1e30;
        for (int t = 0; t < 3; ++t) {
            (void)hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; ++r) {
                if (c) copy();
                if (which == 1) small();
                if (which == 2) big();
                if (which == 3) narrow();
            }
            (void)hipDeviceSynchronize();
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
            best = us < best ? us : best;
        }
        return best;
    };
    for (int w = 0; w < 3; ++w) { copy(); small(); big(); narrow(); }
    int occ_s = 0, occ_b = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_s, mfma_kernel<0>, 256, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_b, mfma_kernel<1>, 256, 0);
    printf("per launch, microseconds (best of 3 x %d launches per stream); MFMA small: up to %d workgroups / CU, big: %d (512 registers per lane)\n", reps, occ_s, occ_b);
    auto row = [&](const char* name, double tc, double tk, double both) {
        printf("%-34s copy %7.1f  kernel %7.1f  together %7.1f  (sum %7.1f, max %7.1f)  hidden %4.0f %% of the shorter\n", name, tc, tk, both, tc + tk,
               tc > tk ? tc : tk, 100.0 * (tc + tk - both) / (tc < tk ? tc : tk));
    };
    for (int cg : {4096, 1024}) {
        copy_grid = cg;
        const double tc = timeit(true, 0);
        printf("copy grid %d (%.2f TB/s alone)\n", cg, 2.0 * bytes / tc / 1e6);
        row("  || MFMA small (512 workgroups)", tc, timeit(false, 1), timeit(true, 1));
        row("  || MFMA big (256 workgroups)", tc, timeit(false, 2), timeit(true, 2));
        row("  || MFMA narrow (32 workgroups)", tc, timeit(false, 3), timeit(true, 3));
    }
    return 0;
}
