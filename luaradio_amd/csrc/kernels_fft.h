// kernels_fft.h - spectrum_utils.DFT / IDFT / PSD (radio/utilities/spectrum_utils.lua:25-113, :259-349,
// :522-640) and fftshift (:654-667) as one fused LDS-resident kernel per frame batch:
//     load (+ window multiply)  ->  radix-2 Stockham FFT in LDS  ->  |X|^2/scale (-> 10 log10)  ->  store
// so a PSD frame costs one 8-B read and one 4-B write per sample (12 B/sample algorithmic) instead of the
// reference's five passes (window multiply, FFT, magnitude, normalise, shift).
// Twiddles come from a host-built table W[m] = exp(-2*pi*i*m/N), m < N/2, computed in double.
#pragma once
#include "common.h"

namespace lrhip {

enum { FFT_OUT_COMPLEX = 0, FFT_OUT_REAL = 1, FFT_OUT_PSD = 2, FFT_OUT_PSD_LOG = 3 };

// 10 log10(p) on the hardware log2 (v_log_f32, 1 ulp): libm's log10f is a dozen instructions per bin.  v_log_f32 flushes denormal inputs, so
// powers below 1e-30 are scaled by 2^64 first; p = 0 gives -inf as in the reference (spectrum_utils.lua:636).
__device__ __forceinline__ float psd_db(float p)
{
    const bool tiny = p < 1e-30f;
    const float l2 = __builtin_amdgcn_logf(tiny ? p * 0x1p+64f : p) - (tiny ? 64.0f : 0.0f);
    return 3.0102999566398120f * l2;
}

// in-LDS autosort (Stockham) FFT over `frames` independent frames laid out back to back (N complex each) in buffer a;
// b is scratch of the same size; tw = W_N^m, m < N/2, in LDS.  Radix-4 passes (half the barriers and LDS round trips of
// radix-2), one radix-2 pass when log2 N is odd.  Returns the buffer holding the result.  All 256 threads call it.
__device__ __forceinline__ float2 tw_at(const float2 *tw, int m, int half, bool inverse)
{
    float2 w = tw[m & (half - 1)];                     // W^(m + N/2) = -W^m
    if (m & half) w = make_float2(-w.x, -w.y);
    if (inverse) w.y = -w.y;
    return w;
}

__device__ __forceinline__ float2 cmulf(float2 a, float2 w) { return make_float2(fmaf(a.x, w.x, -a.y * w.y), fmaf(a.x, w.y, a.y * w.x)); }

__device__ __forceinline__ float2 *fft_lds(float2 *a, float2 *b, int N, int log2n, int frames, const float2 *tw, bool inverse)
{
    const int half = N >> 1, quarter = N >> 2;
    int p = 1, lp = 0;
    for (; lp + 2 <= log2n; lp += 2, p <<= 2) {
        const int twstride = quarter >> lp;            // W_{4p}^k = W_N^{k N/(4p)}
        const int total = frames * quarter;
        for (int w = threadIdx.x; w < total; w += blockDim.x) {
            const int f = w >> (log2n - 2), i = w & (quarter - 1);
            const int k = i & (p - 1);
            const int j = ((i - k) << 2) + k;
            const float2 *src = a + (f << log2n);
            float2 *dst = b + (f << log2n);
            float2 x0 = src[i], x1 = src[i + quarter], x2 = src[i + 2 * quarter], x3 = src[i + 3 * quarter];
            if (p > 1) {
                x1 = cmulf(x1, tw_at(tw, k * twstride, half, inverse));
                x2 = cmulf(x2, tw_at(tw, 2 * k * twstride, half, inverse));
                x3 = cmulf(x3, tw_at(tw, 3 * k * twstride, half, inverse));
            }
            const float2 s0 = make_float2(x0.x + x2.x, x0.y + x2.y), d0 = make_float2(x0.x - x2.x, x0.y - x2.y);
            const float2 s1 = make_float2(x1.x + x3.x, x1.y + x3.y), d1 = make_float2(x1.x - x3.x, x1.y - x3.y);
            // forward: -j * d1 = (d1.y, -d1.x); inverse: +j * d1
            const float2 jd = inverse ? make_float2(-d1.y, d1.x) : make_float2(d1.y, -d1.x);
            dst[j] = make_float2(s0.x + s1.x, s0.y + s1.y);
            dst[j + p] = make_float2(d0.x + jd.x, d0.y + jd.y);
            dst[j + 2 * p] = make_float2(s0.x - s1.x, s0.y - s1.y);
            dst[j + 3 * p] = make_float2(d0.x - jd.x, d0.y - jd.y);
        }
        __syncthreads();
        float2 *t = a; a = b; b = t;
    }
    if (lp < log2n) {                                   // last pass radix-2, p = N/2
        const int total = frames * half;
        for (int w = threadIdx.x; w < total; w += blockDim.x) {
            const int f = w >> (log2n - 1), i = w & (half - 1);
            const float2 *src = a + (f << log2n);
            float2 *dst = b + (f << log2n);
            float2 u0 = src[i], u1 = cmulf(src[i + half], tw_at(tw, i, half, inverse));      // k = i, twstride = 1
            dst[i] = make_float2(u0.x + u1.x, u0.y + u1.y);
            dst[i + half] = make_float2(u0.x - u1.x, u0.y - u1.y);
        }
        __syncthreads();
        float2 *t = a; a = b; b = t;
    }
    return a;
}

// One workgroup transforms FPW frames of N samples.  IN_REAL: Float32 input (imag = 0).
template <bool IN_REAL>
__global__ __launch_bounds__(256) void fft_frames_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                         long nframes, int N, int fpw, const float2 *__restrict__ tw,
                                                         const float *__restrict__ window, int inverse, int out_kind,
                                                         float out_scale, int shift)
{
    extern __shared__ __attribute__((aligned(16))) float2 fbuf[];
    float2 *a = fbuf, *b = fbuf + (size_t)fpw * N, *twl = fbuf + (size_t)2 * fpw * N;      // [a | b | N/2 twiddles]
    const int log2n = 31 - __builtin_clz(N);
    for (int i = threadIdx.x; i < N / 2; i += blockDim.x) twl[i] = tw[i];
    const long f0 = (long)blockIdx.x * fpw;
    int frames = (nframes - f0) < fpw ? (int)(nframes - f0) : fpw;
    const int total = frames * N;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        int s = i & (N - 1);
        long g = f0 * N + i;
        float2 v = IN_REAL ? make_float2(x[g], 0.f) : reinterpret_cast<const float2 *>(x)[g];
        if (window) { float w = window[s]; v.x *= w; v.y *= w; }
        a[i] = v;
    }
    __syncthreads();
    float2 *r = fft_lds(a, b, N, log2n, frames, twl, inverse != 0);
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        int f = i >> log2n, s = i & (N - 1);
        int src = shift ? ((s + N / 2) & (N - 1)) : s;       // fftshift: out[s] = X[(s + N/2) mod N]
        float2 v = r[f * N + src];
        long g = f0 * N + i;
        if (out_kind == FFT_OUT_COMPLEX) {
            reinterpret_cast<float2 *>(y)[g] = make_float2(v.x * out_scale, v.y * out_scale);
        } else if (out_kind == FFT_OUT_REAL) {
            y[g] = v.x * out_scale;
        } else {
            // spectrum_utils.lua:631-638: abs_squared()/scale, optionally 10*log10
            float p = fmaf(v.x, v.x, v.y * v.y) * out_scale;
            y[g] = out_kind == FFT_OUT_PSD_LOG ? psd_db(p) : p;
        }
    }
}

}  // namespace lrhip
