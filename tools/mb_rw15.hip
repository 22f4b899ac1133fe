// Micro-benchmark: the Interpolator(5)'s HBM traffic at its real size with no arithmetic - 2^26 ComplexFloat32 samples read (512 MiB), five times as many written
// (2.5 GiB; far beyond the 256 MB Infinity Cache, unlike the 1 : 5 case of tools/mb_chunk.hip whose 512 MiB output mostly stays on the die).
//   (a) one-shot grid, a thread reads one float4 and writes five (consecutive lanes, consecutive 16 bytes in every instruction), plain / non-temporal stores;
//   (b) the interpolator's tile: a workgroup reads 1280 samples (10 KB) and writes 6400 (50 KB) as 12.5 float4 per thread, one tile per workgroup;
//   (c) write only (the fill ceiling), same store shape.
// build: hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o mb_rw15 tools/mb_rw15.hip ; run: ./mb_rw15
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ void st(float4 *d, float4 v)
{
    if (NT) { const f32x4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<f32x4 *>(d)); }
    else *d = v;
}
template <bool NT, bool RD>
__global__ __launch_bounds__(256) void k15(const float4 *__restrict__ x, float4 *__restrict__ y, long nthreads)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= nthreads) return;
    float4 a = RD ? x[t] : make_float4(1.f, 2.f, 3.f, 4.f);
#pragma unroll
    for (int j = 0; j < 5; j++) st<NT>(y + ((long)blockIdx.x * 5 + j) * 256 + threadIdx.x, make_float4(a.x + j, a.y, a.z, a.w));
}
// tile form: 640 float4 in (2.5 per thread), 3200 float4 out (12.5 per thread)
template <bool NT>
__global__ __launch_bounds__(256) void ktile(const float4 *__restrict__ x, float4 *__restrict__ y, long ntiles)
{
    const long t = blockIdx.x;
    if (t >= ntiles) return;
    const float4 *src = x + t * 640;
    float4 a = src[threadIdx.x], b = src[256 + threadIdx.x], c = threadIdx.x < 128 ? src[512 + threadIdx.x] : a;
    a.x += b.x + c.x;
    float4 *dst = y + t * 3200;
#pragma unroll
    for (int j = 0; j < 12; j++) st<NT>(dst + 256 * j + threadIdx.x, make_float4(a.x + j, a.y, a.z, a.w));
    if (threadIdx.x < 128) st<NT>(dst + 3072 + threadIdx.x, a);
}
template <typename F>
static void timeit(const char *name, double bytes, F launch)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) launch();
    hipEventRecord(a);
    for (int i = 0; i < 10; i++) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("%-72s %8.4f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6);
}
int main()
{
    const long n4 = (1L << 26) / 2;                  // float4 = two ComplexFloat32 samples
    float4 *x, *y;
    hipMalloc(&x, n4 * 16); hipMalloc(&y, n4 * 16 * 5);
    hipMemset(x, 0, n4 * 16); hipMemset(y, 0, n4 * 16 * 5);
    const double rw = 16.0 * n4 * 6, w = 16.0 * n4 * 5;
    const unsigned grid = (unsigned)(n4 / 256);
    for (int rep = 0; rep < 2; rep++) {
        timeit("(a) 1 read : 5 written float4 per thread, plain stores", rw, [&] { hipLaunchKernelGGL((k15<false, true>), dim3(grid), dim3(256), 0, 0, x, y, n4); });
        timeit("(a) 1 read : 5 written float4 per thread, non-temporal stores", rw, [&] { hipLaunchKernelGGL((k15<true, true>), dim3(grid), dim3(256), 0, 0, x, y, n4); });
        timeit("(b) tile: 10 KB read, 50 KB written per workgroup, plain stores", rw, [&] { hipLaunchKernelGGL((ktile<false>), dim3((unsigned)(n4 / 640)), dim3(256), 0, 0, x, y, n4 / 640); });
        timeit("(b) tile: 10 KB read, 50 KB written per workgroup, non-temporal stores", rw, [&] { hipLaunchKernelGGL((ktile<true>), dim3((unsigned)(n4 / 640)), dim3(256), 0, 0, x, y, n4 / 640); });
        timeit("(c) write only, 5 float4 per thread, plain stores", w, [&] { hipLaunchKernelGGL((k15<false, false>), dim3(grid), dim3(256), 0, 0, x, y, n4); });
        timeit("(c) write only, 5 float4 per thread, non-temporal stores", w, [&] { hipLaunchKernelGGL((k15<true, false>), dim3(grid), dim3(256), 0, 0, x, y, n4); });
    }
    return 0;
}
