// ds_read_b64_tr_b16 on gfx950: which LDS element does (lane l, result element j) receive?  Every lane supplies its own 8-byte
// address; LDS holds lds[i] = i.  Prints, for the address pattern used by conv_wgrad16s_kernel (a 16-lane group reads a 4-pixel x
// 16-channel block of a [pixel][channel] image, lane s at pixel s >> 2, channels 4 (s & 3) ..), the element each lane ends up with.
//   hipcc --offload-arch=gfx950 -O2 scripts/microbench/tr16_layout.hip -o scripts/microbench/tr16_layout && scripts/microbench/tr16_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int ROW = 128;   // channels per pixel row
__global__ void k(s16x4* out) {
    __shared__ __attribute__((aligned(16))) short smem[64 * ROW];
    for (int i = threadIdx.x; i < 64 * ROW; i += 64) smem[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, s = l & 15;
    const int pixel = 8 * (l >> 5) + (s >> 2), chan = 16 * ((l >> 4) & 1) + 4 * (s & 3);
    out[l] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + pixel * ROW + chan));
}
int main() {
    s16x4* d; hipMalloc(&d, 64 * sizeof(s16x4));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[64][4]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            const int e = h[l][j], p = e / ROW, c = e % ROW;
            printf("  (px %d, ch %2d)", p, c);
            // expected by the kernel: lane l gets channel (l & 31) of pixels 8 (l >> 5) + j
            if (p != 8 * (l >> 5) + j || c != (l & 31)) ok = 0;
        }
        printf("\n");
    }
    printf("%s\n", ok ? "TR16 LAYOUT AS EXPECTED" : "TR16 LAYOUT DIFFERS");
    return 0;
}
