// Micro-benchmark: a 1:1 read/write stream (2^26 ComplexFloat32 samples = 512 MiB in, 512 MiB out) with the two thread -> address mappings of the library's
// streaming kernels:  (a) coalesced - lane l of a workgroup's pass j touches float4 (256 j + l): consecutive lanes, consecutive 16 bytes;
//                     (b) chunked   - thread t owns CH consecutive float4 (what a recurrence's zero-state run wants: iir_stream_kernel, 16 samples per thread),
//                                     i.e. a load instruction touches 64 lanes x 16 B at a stride of CH x 16 B.
// build: hipcc --offload-arch=gfx950 -O3 -o mb_chunk tools/mb_chunk.hip ; run: ./mb_chunk
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-result"
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
    return 0;
}
