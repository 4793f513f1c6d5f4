// Does a wave's own VALU / LDS work hide behind its v_mfma_f32_32x32x2_f32 stream?  One wave per SIMD (256-thread workgroups, one per CU,
// 512 registers), loop body = 4 MFMAs on two accumulators, each followed by K filler instructions (pinned by sched_barrier):
//   K x v_fma_f32 | K x v_pk_fma_f32 | K x ds_read_b64 | nothing.   Prints shader cycles per MFMA (s_memtime) and wall time.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KIND: 0 v_fma_f32, 1 v_pk_fma_f32, 2 ds_read_b64, 3 v_fma_f32 with the MFMAs in their VGPR form
template <int K, int KIND, int WPS>
__global__ void __launch_bounds__(256 * WPS) __attribute__((amdgpu_waves_per_eu(WPS, WPS))) loop(float* out, long long* cyc, int iters, float a, float b) {
    __shared__ f32x2 lds[1024];
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    lds[threadIdx.x & 1023] = (f32x2){a, b};
    __syncthreads();
    float v[16]; f32x2 pk[16];
    for (int i = 0; i < 16; ++i) { v[i] = a + i + threadIdx.x * 1e-6f; pk[i] = (f32x2){a + i, b + i}; }
    const float x = a + threadIdx.x * 1e-6f, y = b;
    const f32x2* lp = lds + (threadIdx.x & 63);
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (KIND == 3) {
                if (m & 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc1) : "v"(x), "v"(y));
                else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc0) : "v"(x), "v"(y));
            } else {
                if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc0, 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0 || KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k % 16]) : "v"(x), "v"(y));
                else if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pk[k % 16]) : "v"(pk[(k + 8) % 16]));
                else { f32x2 t = lp[(k % 8) * 64]; asm volatile("" :: "v"(t)); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    for (int i = 0; i < 16; ++i) s += v[i] + pk[i].x + pk[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int K, int KIND, int WPS>
static void run(const char* what) {
    const int nblk = 256, iters = 4000;
    float* out; long long* cyc; hipMalloc(&out, (size_t)nblk * 256 * WPS * sizeof(float)); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((loop<K, KIND, WPS>), dim3(nblk), dim3(256 * WPS), 0, 0, out, cyc, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((loop<K, KIND, WPS>), dim3(nblk), dim3(256 * WPS), 0, 0, out, cyc, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-14s K=%2d waves/SIMD=%d: %7.1f s_memtime ticks per MFMA, %6.1f ns per MFMA per wave, %.3f ms\n", what, K, WPS, (double)c / (iters * 4.0), ms * 1e6 / (iters * 4.0), ms);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, 0, 1>("none");
    run<2, 0, 1>("v_fma_f32"); run<4, 0, 1>("v_fma_f32"); run<8, 0, 1>("v_fma_f32"); run<12, 0, 1>("v_fma_f32"); run<16, 0, 1>("v_fma_f32");
    run<2, 1, 1>("v_pk_fma_f32"); run<4, 1, 1>("v_pk_fma_f32"); run<8, 1, 1>("v_pk_fma_f32");
    run<1, 2, 1>("ds_read_b64"); run<2, 2, 1>("ds_read_b64"); run<4, 2, 1>("ds_read_b64");
    run<0, 3, 1>("vgpr-form"); run<4, 3, 1>("vgpr-form+fma"); run<8, 3, 1>("vgpr-form+fma");
    run<0, 0, 2>("none"); run<4, 0, 2>("v_fma_f32"); run<8, 0, 2>("v_fma_f32"); run<16, 0, 2>("v_fma_f32");
    return 0;
}
