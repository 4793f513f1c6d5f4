// Round 6, review item 4 ("fp32 products on the bf16 pipes INSIDE the fused Winograd kernel"): is the inner loop worth building?
//
// Today's loop of wino_fused_kernel (conv_wino_fused.hip), per wave and per 16 input channels: 18 accumulator tiles x 8 v_mfma_f32_32x32x2_f32
// = 144 MFMAs (64 cycles each on the SIMD, and the fp32 MFMA shares the pipe with the VALU: measured in round 4) + the packed input transform
// (48 v_pk_* per 4 channels = 192).  The proposal: split every fp32 operand EXACTLY into three bf16 pieces (h + m + l, 8 + 8 + 8 mantissa
// bits by truncation) and issue the six products that matter (hh, hm, mh, hl, lh, mm: error below the fp32 MFMA's, gemm_bf16x3.hip) as
// v_mfma_f32_32x32x16_bf16: 18 x 6 = 108 MFMAs of 32 cycles per 16 channels.  The filter side is split once per update; the V side is produced
// in registers by the transform, so its split is VALU work in the loop: per channel pair  h = perm(x, y); (hx, hy) = (x, y) & 0xffff0000;
// r = (x, y) - (hx, hy); m = perm(r); (mx, my) = r & mask; r2 = r - m; l = perm(r2)  = 9 instructions per pair, 36 pairs (9 frequencies x 8
// channels / 2) = 324 per 16 channels, on top of the 192 of the transform.
//
// This benchmark runs both loop bodies register-resident (no memory traffic at all: the best case for either), one wave per SIMD as the real
// kernel (16 accumulator tiles instead of 18 so that the builtin keeps them in the AGPRs; the VALU counts are scaled by 16 / 18):
//   fp32   : 128 MFMA 32x32x2 f32, the transform as four bursts of 43 v_pk_fma_f32 (the real kernel's schedule)
//   bf16x6 : 96 MFMA 32x32x16 bf16, 171 v_pk_fma_f32 + 288 split instructions spread evenly (4 - 5 per MFMA gap, pinned by sched_barrier; two split
//            chains interleaved so that dependent instructions are not adjacent; the fragments the MFMAs read are those of the PREVIOUS step)
//   and the two streams alone.  Prints cycles per 16-channel step and the ratio.
//   hipcc --offload-arch=gfx950 -O3 -o split_bf16_loop split_bf16_loop.hip && ./split_bf16_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pk[(i) % 24]) : "v"(pk[((i) + 11) % 24]))
// the bf16x6 step as straight-line code (generated: gen_split_bf16_body.py): DO_MFMA / DO_VALU select its two instruction streams
template <int WHAT>
__device__ __forceinline__ void bf16x6_step(f32x16 (&acc)[16], f32x2 (&pk)[24], u32x4 (&A)[3], u32x4 (&A2)[3], const u32x4 (&Bf)[3]);
#define DO_MFMA 1
#define DO_VALU 1
template <> __device__ __forceinline__ void bf16x6_step<1>(f32x16 (&acc)[16], f32x2 (&pk)[24], u32x4 (&A)[3], u32x4 (&A2)[3], const u32x4 (&Bf)[3])
#include "split_bf16_loop_body.inc"
#undef DO_VALU
#define DO_VALU 0
template <> __device__ __forceinline__ void bf16x6_step<2>(f32x16 (&acc)[16], f32x2 (&pk)[24], u32x4 (&A)[3], u32x4 (&A2)[3], const u32x4 (&Bf)[3])
#include "split_bf16_loop_body.inc"
#undef DO_VALU
#undef DO_MFMA
#define DO_VALU 1
#define DO_MFMA 0
template <> __device__ __forceinline__ void bf16x6_step<3>(f32x16 (&acc)[16], f32x2 (&pk)[24], u32x4 (&A)[3], u32x4 (&A2)[3], const u32x4 (&Bf)[3])
#include "split_bf16_loop_body.inc"

// MODE 0: fp32 loop; 1: bf16x6 loop with all the VALU; 2: bf16 MFMAs alone; 3: the VALU of mode 1 alone; 4: fp32 MFMAs alone
template <int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) loop(float* out, long long* cyc, int iters, float a, float b) {
    f32x16 acc[16];
    for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    f32x2 pk[24];
    for (int i = 0; i < 24; ++i) pk[i] = (f32x2){a + i + threadIdx.x * 1e-6f, b - i};
    u32x4 A[3], A2[3], Bf[3];      // A fragments being built (h, m, l planes of one frequency), B fragments (the filter, split once per update)
    for (int s = 0; s < 3; ++s) { A[s] = (u32x4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; A2[s] = A[s]; Bf[s] = (u32x4){0x3c003c00u + threadIdx.x, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}; }
    const float x = a + threadIdx.x * 1e-6f, y = b;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 4) {
#pragma unroll
            for (int ss = 0; ss < 4; ++ss) {      // four 4-channel sub-steps: 32 MFMAs, then the transform of the next sub-step as ONE burst
#pragma unroll
                for (int m = 0; m < 32; ++m) { acc[m & 15] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[m & 15], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
                if (MODE == 0) {
#pragma unroll
                    for (int k = 0; k < 43; ++k) PKFMA(k);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            bf16x6_step<MODE>(acc, pk, A, A2, Bf);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int i = 0; i < 24; ++i) s += pk[i].x + pk[i].y;
    for (int q = 0; q < 3; ++q) s += (float)(A[q][0] ^ A[q][1] ^ A2[q][2] ^ A2[q][3] ^ A2[q][0] ^ A2[q][1]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
static double run(const char* what) {
    const int nblk = 256, iters = 400;
    float* out; long long* cyc; hipMalloc(&out, (size_t)nblk * 256 * sizeof(float)); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((loop<MODE>), dim3(nblk), dim3(256), 0, 0, out, cyc, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((loop<MODE>), dim3(nblk), dim3(256), 0, 0, out, cyc, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-58s %8.0f cycles per 16-channel step (s_memtime), %7.2f us per step per wave, %.3f ms\n", what, (double)c / iters, ms * 1e3 / iters, ms);
    hipFree(out); hipFree(cyc);
    return ms;
}

int main() {
    const double f4 = run<4>("fp32: 128 x mfma_f32_32x32x2_f32 alone");
    const double f0 = run<0>("fp32: + transform (4 bursts of 43 v_pk_fma_f32) = today");
    const double b2 = run<2>("bf16x6: 96 x mfma_f32_32x32x16_bf16 alone");
    const double b3 = run<3>("bf16x6: transform + split VALU alone (459 instructions)");
    const double b1 = run<1>("bf16x6: MFMAs + transform + split, interleaved");
    printf("ratio fp32 loop / bf16x6 loop = %.2f (MFMAs alone: %.2f; the review's bar for building the kernel: >= 1.5)\n", f0 / b1, f4 / b2);
    printf("bf16x6 loop vs max(MFMA, VALU) alone: %.2f (1.0 = the VALU hides completely behind the bf16 MFMAs)\n", b1 / (b2 > b3 ? b2 : b3));
    return 0;
}
