// st16.h -- storage-dtype helpers of the HBM-bound kernels (gfx950 only).
//
// Round 3: with a 16-bit compute dtype (BASELINE configs[2] bf16, configs[4] fp16) the activations and activation gradients of the wide
// layers (C % 64 == 0) live in HBM in that dtype ("bf16 activations / MFMA, fp32 master").  A tensor's storage is a
// run-time code -- the same kernel serves fp32 tensors (everything image-side, every fp32-compute run) and 16-bit ones -- because these
// kernels are HBM-bound: a wave-uniform branch per 16-byte access costs nothing next to the access itself.
//
//   ST_F32 = 0 (ACLGAN_DTYPE_FP32)    ST_BF16 = 1 (ACLGAN_DTYPE_BF16)    ST_F16 = 2 (ACLGAN_DTYPE_FP16)
//
// Conversions round to nearest even (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32: the same instruction the conv loaders used when they rounded
// fp32 activations on their way into LDS, so a conv operand has the same value whether its producer or its consumer rounded it).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace aclgan {

enum { ST_F32 = 0, ST_BF16 = 1, ST_F16 = 2 };

typedef float st_f32x4 __attribute__((ext_vector_type(4)));
typedef float st_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int st_u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 st_bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 st_f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 st_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 st_f16x2 __attribute__((ext_vector_type(2)));

__host__ __device__ __forceinline__ int st_bytes(int st) { return st == ST_F32 ? 4 : 2; }

// 4 consecutive elements starting at element index 4*i4 of a tensor stored as `st`
__device__ __forceinline__ st_f32x4 st_ld4(const void* __restrict__ p, int64_t i4, int st) {
    if (st == ST_F32) return reinterpret_cast<const st_f32x4*>(p)[i4];
    const st_u32x2 r = reinterpret_cast<const st_u32x2*>(p)[i4];
    if (st == ST_BF16) {      // bf16 -> fp32 is a 16-bit shift
        st_f32x4 o;
        o[0] = __builtin_bit_cast(float, r[0] << 16); o[1] = __builtin_bit_cast(float, r[0] & 0xffff0000u);
        o[2] = __builtin_bit_cast(float, r[1] << 16); o[3] = __builtin_bit_cast(float, r[1] & 0xffff0000u);
        return o;
    }
    return __builtin_convertvector(__builtin_bit_cast(st_f16x4, r), st_f32x4);
}
// N groups of 4 elements at group indices i0, i0 + step, ...: the storage branch is taken ONCE around the N loads.  (Inside an unrolled loop the
// branch sits around every load and the compiler waits for a load at the end of its branch: one load in flight, whatever the unroll -- measured
// in round 6 on the normalisation kernels; with a compile-time code the branch folds away and both forms are the same.)
template <int N>
__device__ __forceinline__ void st_ld4n(const void* __restrict__ p, int64_t i0, int64_t step, int st, st_f32x4 (&o)[N]) {
    if (st == ST_F32) {
#pragma unroll
        for (int u = 0; u < N; ++u) o[u] = reinterpret_cast<const st_f32x4*>(p)[i0 + u * step];
        return;
    }
    st_u32x2 r[N];
#pragma unroll
    for (int u = 0; u < N; ++u) r[u] = reinterpret_cast<const st_u32x2*>(p)[i0 + u * step];
    if (st == ST_BF16) {
#pragma unroll
        for (int u = 0; u < N; ++u) {
            o[u][0] = __builtin_bit_cast(float, r[u][0] << 16); o[u][1] = __builtin_bit_cast(float, r[u][0] & 0xffff0000u);
            o[u][2] = __builtin_bit_cast(float, r[u][1] << 16); o[u][3] = __builtin_bit_cast(float, r[u][1] & 0xffff0000u);
        }
    } else {
#pragma unroll
        for (int u = 0; u < N; ++u) o[u] = __builtin_convertvector(__builtin_bit_cast(st_f16x4, r[u]), st_f32x4);
    }
}
__device__ __forceinline__ void st_st4(void* __restrict__ p, int64_t i4, st_f32x4 v, int st) {
    if (st == ST_F32) { reinterpret_cast<st_f32x4*>(p)[i4] = v; return; }
    if (st == ST_BF16) reinterpret_cast<st_u32x2*>(p)[i4] = __builtin_bit_cast(st_u32x2, __builtin_convertvector(v, st_bf16x4));
    else reinterpret_cast<st_u32x2*>(p)[i4] = __builtin_bit_cast(st_u32x2, __builtin_convertvector(v, st_f16x4));
}
// single element i
__device__ __forceinline__ float st_ld1(const void* __restrict__ p, int64_t i, int st) {
    if (st == ST_F32) return reinterpret_cast<const float*>(p)[i];
    const unsigned short r = reinterpret_cast<const unsigned short*>(p)[i];
    if (st == ST_BF16) return __builtin_bit_cast(float, (unsigned int)r << 16);
    return (float)__builtin_bit_cast(_Float16, r);
}
__device__ __forceinline__ void st_st1(void* __restrict__ p, int64_t i, float v, int st) {
    if (st == ST_F32) { reinterpret_cast<float*>(p)[i] = v; return; }
    if (st == ST_BF16) reinterpret_cast<unsigned short*>(p)[i] = __builtin_bit_cast(unsigned short, (__bf16)v);
    else reinterpret_cast<unsigned short*>(p)[i] = __builtin_bit_cast(unsigned short, (_Float16)v);
}
// two packed 16-bit values <-> two floats
__device__ __forceinline__ st_f32x2 st_unpack2(unsigned int r, int st) {
    st_f32x2 o;
    if (st == ST_BF16) { o[0] = __builtin_bit_cast(float, r << 16); o[1] = __builtin_bit_cast(float, r & 0xffff0000u); return o; }
    return __builtin_convertvector(__builtin_bit_cast(st_f16x2, r), st_f32x2);
}
__device__ __forceinline__ unsigned int st_pack2(float a, float b, int st) {
    st_f32x2 v = {a, b};
    if (st == ST_BF16) return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, st_bf16x2));
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, st_f16x2));
}
// fp32 -> three bf16 numbers with x = h + m + l to 2^-26 |x| (csrc/gemm_bf16x3.hip): both differences are exact in fp32
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    const __bf16 bh = (__bf16)x;
    const float r1 = x - (float)bh;
    const __bf16 bm = (__bf16)r1;
    const float r2 = r1 - (float)bm;
    const __bf16 bl = (__bf16)r2;
    h = __builtin_bit_cast(unsigned short, bh); m = __builtin_bit_cast(unsigned short, bm); l = __builtin_bit_cast(unsigned short, bl);
}
// byte offset of element i
__host__ __device__ __forceinline__ const void* st_at(const void* p, int64_t i, int st) { return (const char*)p + i * st_bytes(st); }
__host__ __device__ __forceinline__ void* st_at(void* p, int64_t i, int st) { return (char*)p + i * st_bytes(st); }

}  // namespace aclgan
