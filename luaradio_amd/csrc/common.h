// common.h - runtime plumbing for liblrhip.so: error channel, context/stream, device + pinned buffers.
#pragma once
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

namespace lrhip {

// ---- error channel: nothing in the library throws, aborts or prints (include/lrhip.h) -----------------
inline char *err_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}
inline int set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return -1;
}
#define LR_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return lrhip::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)
#define LR_HIP_NULL(call)                                                                              \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess) {                                                                       \
            lrhip::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return nullptr;                                                                            \
        }                                                                                              \
    } while (0)

// ---- context: lazily created after fork (SURVEY.md section 7 hard part 2) -----------------------------------
struct Context {
    bool ready = false;
    long pid = 0;                   // process that created the stream: a forked child must not reuse the parent's handles
    int device = 0;
    int num_cus = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;   // own_stream, or an adopted external stream
};
inline Context &ctx()
{
    static Context c;
    return c;
}
inline int ensure_init(int device = -1)
{
    Context &c = ctx();
    if (c.ready && c.pid == (long)getpid()) return 0;
    if (c.ready) {
        // forked since the stream was created (CompositeBlock forks one process per block after initialize(),
        // radio/core/composite.lua:443 vs :569): the parent's stream handle means nothing here - forget it (do not destroy it:
        // it belongs to the parent's context) and build this process's own
        c.ready = false;
        c.own_stream = nullptr;
        c.stream = nullptr;
    }
    int count = 0;
    LR_HIP(hipGetDeviceCount(&count));
    if (count < 1) return set_error("no HIP device visible");
    if (device >= 0) {
        if (device >= count) return set_error("device %d out of range (%d visible)", device, count);
        LR_HIP(hipSetDevice(device));
    }
    LR_HIP(hipGetDevice(&c.device));
    hipDeviceProp_t prop;
    LR_HIP(hipGetDeviceProperties(&prop, c.device));
    c.num_cus = prop.multiProcessorCount;
    LR_HIP(hipStreamCreateWithFlags(&c.own_stream, hipStreamNonBlocking));
    c.stream = c.own_stream;
    c.pid = (long)getpid();
    c.ready = true;
    return 0;
}

// ---- growable device / pinned-host buffers --------------------------------------------------------------
struct DeviceBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return 0;
        size_t want = cap ? cap : 4096;
        while (want < bytes) want *= 2;
        if (p) {
            // the old buffer may still be in use by queued kernels
            LR_HIP(hipStreamSynchronize(ctx().stream));
            LR_HIP(hipFree(p));
            p = nullptr;
            cap = 0;
        }
        LR_HIP(hipMalloc(&p, want));
        cap = want;
        return 0;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    ~DeviceBuf() { release(); }
};

struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return 0;
        size_t want = cap ? cap : 4096;
        while (want < bytes) want *= 2;
        if (p) {
            LR_HIP(hipStreamSynchronize(ctx().stream));
            LR_HIP(hipHostFree(p));
            p = nullptr;
            cap = 0;
        }
        LR_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
        return 0;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
    ~PinnedBuf() { release(); }
};

// Streaming kernels are launched ONE-SHOT: a workgroup per 256 work items, every thread one item (the grid-stride loops in
// the kernels then run once and only guard grids clipped at 2^31-1).  Measured on MI355X (tools/mb_stream.hip, 2 GiB in +
// 2 GiB out): one item per thread 6.2-6.4 TB/s; a persistent grid of 2048-16384 workgroups striding through the same
// buffers 4.4-5.4 TB/s - the dispatcher hands out workgroups in address order, so the set of DRAM pages in flight stays
// compact, while persistent waves drift apart.
inline unsigned grid_for(unsigned long work_items, unsigned per_block, unsigned max_blocks = 0)
{
    unsigned long g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (max_blocks && g > max_blocks) g = max_blocks;
    if (g > 0x7fffffffUL) g = 0x7fffffffUL;
    return (unsigned)g;
}

}  // namespace lrhip
