// pk_math.h - complex arithmetic on the packed-f32 VALU instructions of gfx950
#pragma once
#include "common.h"

namespace lrhip {

// Complex values live in one 64-bit VGPR pair (re, im) so that the butterflies run on the packed-f32 VALU ops
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two flops per lane-instruction - the plain f32 ops are half rate on CDNA).
// Swizzles (re <-> im, broadcast) are expressed as shuffles, which hipcc folds into op_sel; the two products that need a
// one-sided negation (neg_lo / neg_hi) are written as inline VOP3P instructions because the compiler materialises such a
// negation as a separate v_xor.  LRHIP_FFT_PACKED=0 keeps the scalar formulation for A/B measurements.
#ifndef LRHIP_FFT_PACKED
#define LRHIP_FFT_PACKED 1
#endif
typedef float cf __attribute__((ext_vector_type(2)));

__device__ __forceinline__ cf cf_from(float2 a) { return cf{a.x, a.y}; }
__device__ __forceinline__ float2 cf_to(cf a) { return make_float2(a.x, a.y); }

// Every function is __host__ __device__: the device pass compiles the packed body (LRHIP_FFT_PACKED), the host pass the scalar one, so that
// tools/host_fft_check.hip can run the butterflies of the overlap-save kernels on the CPU (index algebra of a kernel checked without a GPU).
#if LRHIP_FFT_PACKED && defined(__HIP_DEVICE_COMPILE__)
#define LRHIP_PK_DEVICE 1
#else
#define LRHIP_PK_DEVICE 0
#endif
__host__ __device__ __forceinline__ cf cadd(cf a, cf b) { return a + b; }
__host__ __device__ __forceinline__ cf csub(cf a, cf b) { return a - b; }
// a + j*b, a - j*b: one packed fma with a (-1, 1) / (1, -1) constant pair; exact (the product is +-b)
__host__ __device__ __forceinline__ cf add_j(cf a, cf b)
{
#if LRHIP_PK_DEVICE
    return __builtin_elementwise_fma(__builtin_shufflevector(b, b, 1, 0), cf{-1.f, 1.f}, a);
#else
    return cf{a.x - b.y, a.y + b.x};
#endif
}
__host__ __device__ __forceinline__ cf sub_j(cf a, cf b)
{
#if LRHIP_PK_DEVICE
    return __builtin_elementwise_fma(__builtin_shufflevector(b, b, 1, 0), cf{1.f, -1.f}, a);
#else
    return cf{a.x + b.y, a.y - b.x};
#endif
}
__host__ __device__ __forceinline__ cf cmul(cf a, cf w)
{
#if LRHIP_PK_DEVICE
    cf t = __builtin_shufflevector(a, a, 0, 0) * w, r;          // (a.x w.x, a.x w.y)
    // (-a.y w.y + t.x, a.y w.x + t.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
#else
    return cf{fmaf(a.x, w.x, -a.y * w.y), fmaf(a.x, w.y, a.y * w.x)};
#endif
}
// a * conj(w)
__host__ __device__ __forceinline__ cf cmulc(cf a, cf w)
{
#if LRHIP_PK_DEVICE
    cf t;                                                       // (a.x w.x, -a.x w.y)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    return __builtin_elementwise_fma(__builtin_shufflevector(a, a, 1, 1), __builtin_shufflevector(w, w, 1, 0), t);
#else
    return cf{fmaf(a.x, w.x, a.y * w.y), fmaf(a.y, w.x, -a.x * w.y)};
#endif
}
// a * (WX + j WY) for a compile-time constant: both operand pairs are constants, no negation needed
__host__ __device__ __forceinline__ cf cmul_const(cf a, float wx, float wy)
{
#if LRHIP_PK_DEVICE
    cf t = __builtin_shufflevector(a, a, 0, 0) * cf{wx, wy};
    return __builtin_elementwise_fma(__builtin_shufflevector(a, a, 1, 1), cf{-wy, wx}, t);
#else
    return cmul(a, cf{wx, wy});
#endif
}

}  // namespace lrhip
