// Micro-benchmark: a 1:1 read/write stream (2^26 ComplexFloat32 samples = 512 MiB in, 512 MiB out) with the two thread -> address mappings of the library's
// streaming kernels:  (a) coalesced - lane l of a workgroup's pass j touches float4 (256 j + l): consecutive lanes, consecutive 16 bytes;
//                     (b) chunked   - thread t owns CH consecutive float4 (what a recurrence's zero-state run wants: iir_stream_kernel, 16 samples per thread),
//                                     i.e. a load instruction touches 64 lanes x 16 B at a stride of CH x 16 B.
// build: hipcc --offload-arch=gfx950 -O3 -o mb_chunk tools/mb_chunk.hip ; run: ./mb_chunk
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
template <int CH, bool CHUNKED>
__global__ __launch_bounds__(256) void k(const float4 *__restrict__ x, float4 *__restrict__ y, long n4)
{
    const long base = (long)blockIdx.x * 256 * CH;
    float4 v[CH];
#pragma unroll
    for (int j = 0; j < CH; j++) {
        const long i = CHUNKED ? base + (long)threadIdx.x * CH + j : base + 256L * j + threadIdx.x;
        v[j] = i < n4 ? x[i] : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < CH; j++) {
        const long i = CHUNKED ? base + (long)threadIdx.x * CH + j : base + 256L * j + threadIdx.x;
        if (i < n4) y[i] = make_float4(v[j].x * 1.5f, v[j].y, v[j].z, v[j].w);
    }
}
// (d) read : write ratios other than 1 : 1, ideal mapping (one 16-byte access per lane and instruction, one-shot grid): RD float4 loads and WR float4 stores per thread
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int RD, int WR, bool NT = false, bool NTL = false>
__global__ __launch_bounds__(256) void k_rw(const float4 *__restrict__ x, float4 *__restrict__ y, long nthreads)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= nthreads) return;
    float4 a = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < RD; j++) {
        float4 v;
        if (NTL) { const f32x4 w = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(x + ((long)blockIdx.x * RD + j) * 256 + threadIdx.x)); v = make_float4(w[0], w[1], w[2], w[3]); }
        else v = x[((long)blockIdx.x * RD + j) * 256 + threadIdx.x];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
#pragma unroll
    for (int j = 0; j < WR; j++) {
        float4 *d = y + ((long)blockIdx.x * WR + j) * 256 + threadIdx.x;
        if (NT) { const f32x4 v = {a.x + j, a.y, a.z, a.w}; __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(d)); }
        else *d = make_float4(a.x + j, a.y, a.z, a.w);
    }
    if (WR == 0 && a.x == 12345.678f) y[t] = a;
}
template <int RD, int WR, bool NT = false, bool NTL = false>
static void run_rw(const char *name, const float4 *x, float4 *y, long n4)
{
    const long nthreads = n4 / (RD > WR ? RD : WR);
    const unsigned grid = (unsigned)(nthreads / 256);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k_rw<RD, WR, NT, NTL>), dim3(grid), dim3(256), 0, 0, x, y, nthreads);
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k_rw<RD, WR, NT, NTL>), dim3(grid), dim3(256), 0, 0, x, y, nthreads);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
    printf("%-44s %8.4f ms  %7.1f GB/s (%d read + %d written float4 per thread)\n", name, ms, 16.0 * (RD + WR) * nthreads / ms / 1e6, RD, WR);
}
// (c) the overlap-save FIR's traffic with no arithmetic: a wave per 1024-sample window at a hop of 896 (16 loads of 64 lanes x 8 B, 14 stores), windows dealt to
//     waves one-shot (a wave per window, address order) or to a persistent grid (PERSIST workgroups per CU, stride = the grid)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NTF>      // bit 0: non-temporal loads, bit 1: non-temporal stores
__global__ __launch_bounds__(256) void k_os_nt(const float2 *__restrict__ x, float2 *__restrict__ y, long n, long nblocks, int persist)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long first = (long)blockIdx.x * 4 + wave, step = persist ? (long)gridDim.x * 4 : nblocks;
    for (long fb = first; fb < nblocks; fb += step) {
        const long p0 = fb * 896;
        f32x2 v[16];
        if (p0 + 1024 <= n) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const f32x2 *s = reinterpret_cast<const f32x2 *>(x + p0 + 64 * i) + lane;
                v[i] = (NTF & 1) ? __builtin_nontemporal_load(s) : *s;
            }
#pragma unroll
            for (int i = 2; i < 16; i++) {
                f32x2 *d = reinterpret_cast<f32x2 *>(y + p0 + 64 * i) + lane;
                const f32x2 o = {v[i][0] * 1.5f + v[i & 1][1], v[i][1]};
                if (NTF & 2) __builtin_nontemporal_store(o, d); else *d = o;
            }
        }
    }
}
template <int NTF>
static void run_os_nt(const char *name, const float2 *x, float2 *y, long n, int wgs_per_cu)
{
    const long nblocks = n / 896;
    const unsigned grid = wgs_per_cu ? 256u * wgs_per_cu : (unsigned)((nblocks + 3) / 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k_os_nt<NTF>, dim3(grid), dim3(256), 0, 0, x, y, n, nblocks, wgs_per_cu);
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_os_nt<NTF>, dim3(grid), dim3(256), 0, 0, x, y, n, nblocks, wgs_per_cu);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
    printf("%-44s %8.4f ms  %7.1f GB/s (16 B per sample)\n", name, ms, 16.0 * n / ms / 1e6);
}
__global__ __launch_bounds__(256) void k_os(const float2 *__restrict__ x, float2 *__restrict__ y, long n, long nblocks, int persist)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long first = (long)blockIdx.x * 4 + wave, step = persist ? (long)gridDim.x * 4 : nblocks;
    for (long fb = first; fb < nblocks; fb += step) {
        const long p0 = fb * 896;
        float2 v[16];
        if (p0 + 1024 <= n) {
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = x[p0 + 64 * i + lane];
#pragma unroll
            for (int i = 2; i < 16; i++) y[p0 + 64 * i + lane] = make_float2(v[i].x * 1.5f + v[i & 1].y, v[i].y);
        }
    }
}
// the same with 16-byte accesses (8 loads of 64 lanes x 16 B) and / or without the overlap (hop 1024): what each ingredient of the pattern costs
template <int HOP>
__global__ __launch_bounds__(256) void k_os4(const float4 *__restrict__ x, float4 *__restrict__ y, long n, long nblocks)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long fb = (long)blockIdx.x * 4 + wave;
    if (fb >= nblocks) return;
    const long p0 = fb * (HOP / 2);      // in float4 = two samples
    float4 v[8];
    if (2 * p0 + 1024 <= n) {
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = x[p0 + 64 * i + lane];
#pragma unroll
        for (int i = (1024 - HOP) / 128; i < 8; i++) y[p0 + 64 * i + lane] = make_float4(v[i].x * 1.5f + v[i & 1].y, v[i].y, v[i].z, v[i].w);
    }
}
template <int HOP>
static void run_os4(const char *name, const float4 *x, float4 *y, long n)
{
    const long nblocks = n / HOP;
    const unsigned grid = (unsigned)((nblocks + 3) / 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k_os4<HOP>, dim3(grid), dim3(256), 0, 0, x, y, n, nblocks);
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_os4<HOP>, dim3(grid), dim3(256), 0, 0, x, y, n, nblocks);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
    printf("%-44s %8.4f ms  %7.1f GB/s (16 B per sample)\n", name, ms, 16.0 * n / ms / 1e6);
}
static void run_os(const char *name, const float2 *x, float2 *y, long n, int wgs_per_cu)
{
    const long nblocks = n / 896;
    const unsigned grid = wgs_per_cu ? 256u * wgs_per_cu : (unsigned)((nblocks + 3) / 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k_os, dim3(grid), dim3(256), 0, 0, x, y, n, nblocks, wgs_per_cu);
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_os, dim3(grid), dim3(256), 0, 0, x, y, n, nblocks, wgs_per_cu);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
    printf("%-44s %8.4f ms  %7.1f GB/s (16 B per sample)\n", name, ms, 16.0 * n / ms / 1e6);
}
template <int CH, bool CHUNKED>
static void run(const char *name, const float4 *x, float4 *y, long n4)
{
    const unsigned grid = (unsigned)((n4 + 256L * CH - 1) / (256L * CH));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k<CH, CHUNKED>), dim3(grid), dim3(256), 0, 0, x, y, n4);
    hipEventRecord(a);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k<CH, CHUNKED>), dim3(grid), dim3(256), 0, 0, x, y, n4);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
    printf("%-34s %8.4f ms  %7.1f GB/s\n", name, ms, 32.0 * n4 / ms / 1e6);
}
int main()
{
    const long n4 = 1L << 25;      // float4s: 512 MiB
    float4 *x, *y;
    hipMalloc(&x, n4 * 16); hipMalloc(&y, n4 * 16);
    hipMemset(x, 0, n4 * 16);
    for (int r = 0; r < 2; r++) {
        run<1, false>("coalesced, 1 float4 per thread", x, y, n4);
        run<4, false>("coalesced, 4 float4 per thread", x, y, n4);
        run<8, false>("coalesced, 8 float4 per thread", x, y, n4);
        run<4, true>("chunked, 4 float4 (64 B) per thread", x, y, n4);
        run<8, true>("chunked, 8 float4 (128 B) per thread", x, y, n4);
    }
    const long n = n4 * 2;      // ComplexFloat32 samples in the same buffers
    for (int r = 0; r < 2; r++) {
        run_os("overlap-save traffic, wave per window", (const float2 *)x, (float2 *)y, n, 0);
        run_os("overlap-save traffic, persistent 2 WG / CU", (const float2 *)x, (float2 *)y, n, 2);
        run_os("overlap-save traffic, persistent 4 WG / CU", (const float2 *)x, (float2 *)y, n, 4);
        run_os("overlap-save traffic, persistent 8 WG / CU", (const float2 *)x, (float2 *)y, n, 8);
        run_rw<1, 0>("read only", x, y, n4);
        run_rw<0, 1>("write only", x, y, n4);
        run_rw<1, 1>("1 : 1", x, y, n4);
        run_rw<2, 1>("2 : 1 (discriminator, PSD)", x, y, n4);
        run_rw<5, 1>("5 : 1 (decimator)", x, y, n4);
        run_rw<1, 2>("1 : 2 (Hilbert)", x, y, n4);
        run_rw<1, 5>("1 : 5 (interpolator)", x, y, n4);
        run_rw<0, 1, true>("write only, non-temporal stores", x, y, n4);
        run_rw<1, 1, true>("1 : 1, non-temporal stores", x, y, n4);
        run_rw<2, 1, true>("2 : 1, non-temporal stores", x, y, n4);
        run_rw<1, 2, true>("1 : 2, non-temporal stores", x, y, n4);
        run_rw<1, 5, true>("1 : 5, non-temporal stores", x, y, n4);
        run_rw<1, 0, false, true>("read only, non-temporal loads", x, y, n4);
        run_rw<1, 1, true, true>("1 : 1, non-temporal loads + stores", x, y, n4);
        run_rw<1, 1, false, true>("1 : 1, non-temporal loads only", x, y, n4);
        run_rw<2, 1, true, true>("2 : 1, non-temporal loads + stores", x, y, n4);
        run_rw<5, 1, true, true>("5 : 1, non-temporal loads + stores", x, y, n4);
        run_rw<1, 2, true, true>("1 : 2, non-temporal loads + stores", x, y, n4);
        run_os_nt<0>("overlap-save, persistent 4 WG / CU, plain", (const float2 *)x, (float2 *)y, n, 4);
        run_os_nt<1>("overlap-save, persistent 4, NT loads", (const float2 *)x, (float2 *)y, n, 4);
        run_os_nt<2>("overlap-save, persistent 4, NT stores", (const float2 *)x, (float2 *)y, n, 4);
        run_os_nt<3>("overlap-save, persistent 4, NT both", (const float2 *)x, (float2 *)y, n, 4);
        run_os4<896>("overlap-save traffic, 16-B accesses", x, y, n);
        run_os4<1024>("same windows without overlap, 16-B accesses", x, y, n);
    }
    return 0;
}
