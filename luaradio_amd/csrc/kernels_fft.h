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

// in-LDS autosort radix-2 FFT over `frames` independent frames laid out back to back (N complex each) in
// buffer a; b is scratch of the same size.  Returns the buffer holding the result.  All 256 threads call it.
__device__ __forceinline__ float2 *fft_lds(float2 *a, float2 *b, int N, int frames, const float2 *__restrict__ tw,
                                           bool inverse)
{
    const int half = N >> 1;
    const int total = frames * half;
    for (int p = 1; p < N; p <<= 1) {
        const int twstride = half / p;          // W_{2p}^k = W_N^{k*N/(2p)}
        for (int w = threadIdx.x; w < total; w += blockDim.x) {
            int f = w / half, i = w - f * half;
            int k = i & (p - 1);
            int j = ((i - k) << 1) + k;
            float2 wv = tw[k * twstride];
            if (inverse) wv.y = -wv.y;
            const float2 *src = a + f * N;
            float2 *dst = b + f * N;
            float2 u0 = src[i], x1 = src[i + half];
            float2 u1 = make_float2(fmaf(x1.x, wv.x, -x1.y * wv.y), fmaf(x1.x, wv.y, x1.y * wv.x));
            dst[j] = make_float2(u0.x + u1.x, u0.y + u1.y);
            dst[j + p] = make_float2(u0.x - u1.x, u0.y - u1.y);
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
    float2 *a = fbuf, *b = fbuf + (size_t)fpw * N;
    const long f0 = (long)blockIdx.x * fpw;
    int frames = (nframes - f0) < fpw ? (int)(nframes - f0) : fpw;
    const int total = frames * N;
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        int s = i % N;
        long g = f0 * N + i;
        float2 v = IN_REAL ? make_float2(x[g], 0.f) : reinterpret_cast<const float2 *>(x)[g];
        if (window) { float w = window[s]; v.x *= w; v.y *= w; }
        a[i] = v;
    }
    __syncthreads();
    float2 *r = fft_lds(a, b, N, frames, tw, inverse != 0);
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        int f = i / N, s = i - f * N;
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
            y[g] = out_kind == FFT_OUT_PSD_LOG ? 10.0f * log10f(p) : p;
        }
    }
}

}  // namespace lrhip
