// kernels_elem.h - the HBM-bound element-wise blocks: frequency translator, FM discriminator, downsampler.
// All are streaming kernels: coalesced 8/16-B-per-lane loads, grid-stride, no LDS.
#pragma once
#include "common.h"

namespace lrhip {

// ------------------------------------------------------------------------------------------------
// FrequencyTranslatorBlock  (reference: radio/blocks/signal/frequencytranslator.lua:93-110)
//   y[n] = x[n] * exp(j*omega*n)
// The reference carries a running phase (double accumulator in Lua, f32 phasor in VOLK); here the phase of
// sample n is closed-form from a 64-bit sample counter, so any chunking / any launch split gives the same
// values and there is no drift:   turns(n) = frac(n * omega/2pi)  computed EXACTLY in 0.64 fixed point
// (unsigned 64-bit wrap-around multiply == mod 1 turn).
// Algorithmic traffic: 16 B/sample (8 in + 8 out).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void phasor_from_turns(uint64_t turns_fx, float &c, float &s)
{
    // signed fraction of a turn in [-0.5, 0.5) -> angle in [-pi, pi)
    const double k = 6.283185307179586476925286766559 / 18446744073709551616.0;   // 2*pi / 2^64
    float a = (float)((double)(int64_t)turns_fx * k);
    sincosf(a, &s, &c);
}

__global__ __launch_bounds__(256) void rotator_kernel(const float2 *__restrict__ x, float2 *__restrict__ y,
                                                      unsigned long n, uint64_t step_fx, uint64_t count0)
{
    unsigned long stride = (unsigned long)gridDim.x * blockDim.x;
    for (unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float2 v = x[i];
        float c, s;
        phasor_from_turns(step_fx * (count0 + i), c, s);
        // complex multiply, each component rounded once (complexfloat32.lua:79-81 computes in double)
        double xr = v.x, xi = v.y;
        float2 o;
        o.x = (float)(xr * (double)c - xi * (double)s);
        o.y = (float)(xr * (double)s + xi * (double)c);
        y[i] = o;
    }
}

// ------------------------------------------------------------------------------------------------
// FrequencyDiscriminatorBlock  (reference: radio/blocks/signal/frequencydiscriminator.lua:68-88)
//   tmp[n] = x[n] * conj(x[n-1]);  y[n] = atan2(Im tmp, Re tmp) * (1/gain)
// x[-1] is the carried previous sample (prev_in), zero initially (:34).  The kernel also publishes the last
// sample of the chunk to prev_out (ping-pong, so no block reads what another writes).
// Algorithmic traffic: 12 B/sample (8 in + 4 out); the x[n-1] re-read hits L1/L2.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float discriminate(float2 a, float2 b, double inv_gain)
{
    double ar = a.x, ai = a.y, br = b.x, bi = -(double)b.y;
    float tr = (float)(ar * br - ai * bi), ti = (float)(ar * bi + ai * br);
    return (float)((double)atan2f(ti, tr) * inv_gain);
}

__global__ __launch_bounds__(256) void fmdiscrim_kernel(const float2 *__restrict__ x, float *__restrict__ y,
                                                        unsigned long n, double inv_gain,
                                                        const float2 *__restrict__ prev_in, float2 *__restrict__ prev_out)
{
    unsigned long stride = (unsigned long)gridDim.x * blockDim.x;
    for (unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float2 a = x[i];
        float2 b = i ? x[i - 1] : *prev_in;
        y[i] = discriminate(a, b, inv_gain);
        if (i == n - 1) *prev_out = a;
    }
}

// ------------------------------------------------------------------------------------------------
// DownsamplerBlock  (reference: radio/blocks/signal/downsampler.lua:45-56):  y[m] = x[index + m*factor].
// Bit-exact copies.  Traffic: E/factor out + between E/factor and E in (sector granularity).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void downsample_kernel(const T *__restrict__ x, T *__restrict__ y,
                                                         unsigned long n_out, unsigned long index, unsigned long factor)
{
    unsigned long stride = (unsigned long)gridDim.x * blockDim.x;
    for (unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += stride)
        y[i] = x[index + i * factor];
}

}  // namespace lrhip
