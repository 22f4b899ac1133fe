// mb_stream.hip - streaming-floor micro-benchmark for MI355X (gfx950): what does a 1:1 read/write stream reach, and
// with which access shape?  Build: hipcc --offload-arch=gfx950 -O3 -o mb_stream mb_stream.hip ; run: ./mb_stream [log2_bytes]
// Prints one line per variant: GB/s counting bytes read + bytes written.  Used to pick the shape of the library's
// element-wise kernels (kernels_elem.h) and the bench's streaming yardstick.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// 0: plain, 1: nontemporal store, 2: nontemporal load + store
template <typename T, int NT>
__device__ __forceinline__ T ld(const T *p) { return NT >= 2 ? __builtin_nontemporal_load(p) : *p; }
template <typename T, int NT>
__device__ __forceinline__ void st(T *p, T v) { if (NT >= 1) __builtin_nontemporal_store(v, p); else *p = v; }

// interleaved grid-stride, U accesses in flight per thread per iteration
template <typename T, int U, int NT>
__global__ __launch_bounds__(256) void copy_gs(const T *__restrict__ x, T *__restrict__ y, size_t n, float s)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = ld<T, NT>(x + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; u++) st<T, NT>(y + i + u * stride, v[u] * s);
    }
    for (; i < n; i += stride) st<T, NT>(y + i, ld<T, NT>(x + i) * s);
}

// contiguous chunk per workgroup: block b owns [b*chunk, (b+1)*chunk), walks it 256*U elements at a time
template <typename T, int U, int NT>
__global__ __launch_bounds__(256) void copy_chunk(const T *__restrict__ x, T *__restrict__ y, size_t n, float s)
{
    const size_t chunk = (n + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256 * U) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; u++) if (i + u * 256 < hi) v[u] = ld<T, NT>(x + i + u * 256);
#pragma unroll
        for (int u = 0; u < U; u++) if (i + u * 256 < hi) st<T, NT>(y + i + u * 256, v[u] * s);
    }
}

// one shot: every thread does U elements, grid covers n exactly (no loop)
template <typename T, int U, int NT>
__global__ __launch_bounds__(256) void copy_once(const T *__restrict__ x, T *__restrict__ y, size_t n, float s)
{
    const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    T v[U];
#pragma unroll
    for (int u = 0; u < U; u++) if (base + u * 256 < n) v[u] = ld<T, NT>(x + base + u * 256);
#pragma unroll
    for (int u = 0; u < U; u++) if (base + u * 256 < n) st<T, NT>(y + base + u * 256, v[u] * s);
}

template <typename T, int U>
__global__ __launch_bounds__(256) void read_only(const T *__restrict__ x, float *__restrict__ sink, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    T acc = {};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; u++) if (i + u * stride < n) acc += x[i + u * stride];
    }
    if (acc[0] == 12345.678f) sink[0] = acc[0];
}

template <typename T, int NT>
__global__ __launch_bounds__(256) void write_only(T *__restrict__ y, size_t n, float s)
{
    const size_t stride = (size_t)gridDim.x * 256;
    T v = {};
    v[0] = s;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) st<T, NT>(y + i, v);
}

static hipEvent_t e0, e1;
template <typename F>
static double timeit(F f, int iters = 10)
{
    f(); f();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) f();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main(int argc, char **argv)
{
    int lg = argc > 1 ? atoi(argv[1]) : 31;
    size_t bytes = (size_t)1 << lg;
    float *x, *y;
    CK(hipMalloc(&x, bytes));
    CK(hipMalloc(&y, bytes));
    CK(hipMemset(x, 0, bytes));
    CK(hipMemset(y, 0, bytes));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("buffers: %zu MiB in + %zu MiB out\n", bytes >> 20, bytes >> 20);
    auto report = [&](const char *name, double ms, double nbytes) { printf("%-44s %8.4f ms  %8.1f GB/s\n", name, ms, nbytes / ms / 1e6); fflush(stdout); };
    const double rw = 2.0 * bytes;
    const size_t n2 = bytes / 8, n4 = bytes / 16;
    char nm[128];
    report("hipMemcpyDtoD", timeit([&] { CK(hipMemcpyAsync(y, x, bytes, hipMemcpyDeviceToDevice, 0)); }), rw);
    for (int g : {2048, 4096, 8192, 16384}) {
        snprintf(nm, sizeof nm, "f2 gs U1 grid %d", g);
        report(nm, timeit([&] { copy_gs<f2, 1, 0><<<g, 256>>>((const f2 *)x, (f2 *)y, n2, 1.f); }), rw);
    }
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
        snprintf(nm, sizeof nm, "f4 gs U1 grid %d", g);
        report(nm, timeit([&] { copy_gs<f4, 1, 0><<<g, 256>>>((const f4 *)x, (f4 *)y, n4, 1.f); }), rw);
        snprintf(nm, sizeof nm, "f4 gs U1 nt-store grid %d", g);
        report(nm, timeit([&] { copy_gs<f4, 1, 1><<<g, 256>>>((const f4 *)x, (f4 *)y, n4, 1.f); }), rw);
        snprintf(nm, sizeof nm, "f4 gs U2 nt-store grid %d", g);
        report(nm, timeit([&] { copy_gs<f4, 2, 1><<<g, 256>>>((const f4 *)x, (f4 *)y, n4, 1.f); }), rw);
        snprintf(nm, sizeof nm, "f4 gs U4 grid %d", g);
        report(nm, timeit([&] { copy_gs<f4, 4, 0><<<g, 256>>>((const f4 *)x, (f4 *)y, n4, 1.f); }), rw);
        snprintf(nm, sizeof nm, "f4 gs U4 nt-store grid %d", g);
        report(nm, timeit([&] { copy_gs<f4, 4, 1><<<g, 256>>>((const f4 *)x, (f4 *)y, n4, 1.f); }), rw);
        snprintf(nm, sizeof nm, "f4 gs U4 nt-load+store grid %d", g);
        report(nm, timeit([&] { copy_gs<f4, 4, 2><<<g, 256>>>((const f4 *)x, (f4 *)y, n4, 1.f); }), rw);
        snprintf(nm, sizeof nm, "f4 chunk U4 nt-store grid %d", g);
        report(nm, timeit([&] { copy_chunk<f4, 4, 1><<<g, 256>>>((const f4 *)x, (f4 *)y, n4, 1.f); }), rw);
    }
    {
        unsigned g = (unsigned)((n4 + 256 * 4 - 1) / (256 * 4));
        report("f4 once U4", timeit([&] { copy_once<f4, 4, 0><<<g, 256>>>((const f4 *)x, (f4 *)y, n4, 1.f); }), rw);
        report("f4 once U4 nt-store", timeit([&] { copy_once<f4, 4, 1><<<g, 256>>>((const f4 *)x, (f4 *)y, n4, 1.f); }), rw);
        g = (unsigned)((n4 + 256 * 8 - 1) / (256 * 8));
        report("f4 once U8 nt-store", timeit([&] { copy_once<f4, 8, 1><<<g, 256>>>((const f4 *)x, (f4 *)y, n4, 1.f); }), rw);
        g = (unsigned)((n4 + 255) / 256);
        report("f4 once U1", timeit([&] { copy_once<f4, 1, 0><<<g, 256>>>((const f4 *)x, (f4 *)y, n4, 1.f); }), rw);
        report("f4 once U1 nt-store", timeit([&] { copy_once<f4, 1, 1><<<g, 256>>>((const f4 *)x, (f4 *)y, n4, 1.f); }), rw);
        g = (unsigned)((n2 + 255) / 256);
        report("f2 once U1", timeit([&] { copy_once<f2, 1, 0><<<g, 256>>>((const f2 *)x, (f2 *)y, n2, 1.f); }), rw);
        g = (unsigned)((n2 + 256 * 4 - 1) / (256 * 4));
        report("f2 once U4 nt-store", timeit([&] { copy_once<f2, 4, 1><<<g, 256>>>((const f2 *)x, (f2 *)y, n2, 1.f); }), rw);
    }
    for (int g : {2048, 4096, 8192}) {
        snprintf(nm, sizeof nm, "read-only f4 U4 grid %d", g);
        report(nm, timeit([&] { read_only<f4, 4><<<g, 256>>>((const f4 *)x, y, n4); }), (double)bytes);
        snprintf(nm, sizeof nm, "write-only f4 grid %d", g);
        report(nm, timeit([&] { write_only<f4, 0><<<g, 256>>>((f4 *)y, n4, 1.f); }), (double)bytes);
        snprintf(nm, sizeof nm, "write-only f4 nt grid %d", g);
        report(nm, timeit([&] { write_only<f4, 1><<<g, 256>>>((f4 *)y, n4, 1.f); }), (double)bytes);
    }
    return 0;
}
