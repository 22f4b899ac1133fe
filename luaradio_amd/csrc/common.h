// common.h - runtime plumbing for liblrhip.so: error channel, context/stream, device + pinned buffers.
#pragma once
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <sys/stat.h>
#include <cerrno>
#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace lrhip {

// Set by host_execute (chain.h) around a launch whose input / output pointers are the caller's registered HOST memory: kernels that know of it launch a short
// persistent grid (0 = device memory, the normal grids).  LRHIP_HOST_IO_GRID overrides the workgroup count (A/B).
inline int &host_io_grid_ref() { static thread_local int v = 0; return v; }
inline int host_io_grid()
{
    static const int env = getenv("LRHIP_HOST_IO_GRID") ? atoi(getenv("LRHIP_HOST_IO_GRID")) : 0;
    return env < 0 ? host_io_grid_ref() > 0 ? -env : 0 : env > 0 ? env : host_io_grid_ref();
}


// A 16-byte store that tells the caches the line will not be read again by this launch.  tools/mb_chunk.hip, 2^26 ComplexFloat32 samples: a 1 : 1 stream moves
// 6.27 -> 6.44 TB/s with it, one read per two writes 5.95 -> 8.07, one per five 5.48 -> 6.88: the more of a kernel's traffic is output, the more it matters.
#ifdef __HIPCC__
typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store(float4 *dst, float4 v)
{
    const nt_f32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<nt_f32x4 *>(dst));
}
// wave-uniform read-only tables (filter taps) through the scalar cache: a load from the constant address space at a uniform address is an
// s_load, its result lives in SGPRs and feeds VALU instructions as a scalar operand - no LDS read, no vector register
__device__ __forceinline__ float4 uniform_load4(const float *p)
{
    const nt_f32x4 v = *(const __attribute__((address_space(4))) nt_f32x4 *)p;
    return make_float4(v[0], v[1], v[2], v[3]);
}
typedef float nt_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ nt_f32x2 uniform_load2(const float *p) { return *(const __attribute__((address_space(4))) nt_f32x2 *)p; }
__device__ __forceinline__ void nt_store(float2 *dst, float2 v)
{
    const nt_f32x2 w = {v.x, v.y};
    __builtin_nontemporal_store(w, reinterpret_cast<nt_f32x2 *>(dst));
}
#endif


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
        if (e__ != hipSuccess) {                                                                       \
            (void)hipGetLastError();      /* the error is reported HERE: do not leave it for the next launch check to trip over */ \
            return lrhip::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
        }                                                                                              \
    } while (0)
#define LR_HIP_NULL(call)                                                                              \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess) {                                                                       \
            (void)hipGetLastError();                                                                   \
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
    if (c.ready && c.pid == (long)getpid()) {
        if (device >= 0 && device != c.device)
            return set_error("this process is bound to device %d; lrhip_init(%d) refused (one device per process)", c.device, device);
        return 0;
    }
    if (c.ready) {
        // forked since the device context was created.  CompositeBlock forks one process per block after initialize()
        // (radio/core/composite.lua:443 vs :569), which is why device blocks create their objects on the first process(); a parent that DID
        // touch the device before fork() has left this child a copy of a HIP runtime whose threads, queues and doorbells exist only in the
        // parent - any HIP call from here may hang.  Refuse with a message instead (the block process then exits 1, composite.lua:625-629).
        return set_error("the device was initialised in process %ld before fork(); a forked child (process %ld) cannot use it - "
                         "create device objects after fork(): LuaRadio device blocks do so on their first process()", c.pid, (long)getpid());
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

// ---- host-side copies into / out of the pinned slots --------------------------------------------------------------------------
// A LuaRadio process() hands the library a pageable vector (radio/core/vector.lua:19-37); the copy into the pinned ring slot ran on ONE
// host thread in round 2 and was the whole host-path bound (~10 GB/s: 1.3 GS/s of ComplexFloat32 at the pipe's 131 072-sample chunks,
// DESIGN.md section 7).  Copies of half a megabyte and more are split over a small pool of helper threads (LRHIP_COPY_THREADS, default 8
// including the caller, at most half the hardware threads; 1 = off).  The pool is created lazily in the process that uses it (a forked block process builds its own: the
// parent's threads do not exist in the child) and never touches HIP.
struct CopyPool {
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_work;
    long pid = 0;
    int nthreads = 1;
    // one job at a time (the caller blocks until it is done)
    char *dst = nullptr;
    const char *src = nullptr;
    size_t bytes = 0, part = 0;
    std::atomic<size_t> next{0};
    std::atomic<int> generation{0}, running{0};
    bool stop = false;
    std::mutex job;                              // one job at a time: a second host thread calling into the library meanwhile (ctypes releases the GIL;
                                                 // two chains on two Python threads) copies on its own instead of overwriting dst / src / next

    static inline void cpu_relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#elif defined(__aarch64__)
        __asm__ __volatile__("yield");
#else
        std::this_thread::yield();
#endif
    }

    // a job is a memcpy, or (fd >= 0) a positional read of a regular file into dst: the page cache hands a single read(2) stream ~10 GB/s, several
    // streams on several cores several times that
    int fd = -1;
    long long file_off = 0;
    std::atomic<int> io_errno{0};
    void work()
    {
        for (;;) {
            const size_t off = next.fetch_add(part);
            if (off >= bytes) break;
            const size_t len = off + part <= bytes ? part : bytes - off;
            if (fd < 0) { memcpy(dst + off, src + off, len); continue; }
            size_t got = 0;
            while (got < len) {
                const ssize_t r = pread(fd, dst + off + got, len - got, (off_t)(file_off + (long long)(off + got)));
                if (r < 0) { if (errno == EINTR) continue; io_errno.store(errno); break; }
                if (r == 0) { io_errno.store(EIO); break; }      // the caller sized the job from fstat(): a short file now is an error
                got += (size_t)r;
            }
        }
    }
    void loop()
    {
        int seen = 0;
        for (;;) {
            // a megabyte copies in ~25 us on four threads - less than a condition-variable wake-up: spin for a while first (chunks arrive back to
            // back while a stream runs), sleep when the stream pauses
            // (LRHIP_COPY_SPIN iterations, default 20000 = a few hundred microseconds; a flow graph with many device blocks per host can lower it - round 4 measured
            // 4000 on the pipelined host path: the pieces of one call arrive further apart than that, the workers slept in between, and the staged
            // stand-alone block fell from 2.4 to 1.0-1.7 GS/s)
            int spins = 0;
            while (generation.load(std::memory_order_acquire) == seen && spins < spin_limit) {
                cpu_relax();
                spins++;
            }
            if (generation.load(std::memory_order_acquire) == seen) {
                std::unique_lock<std::mutex> lk(m);
                cv_work.wait(lk, [&] { return stop || generation.load(std::memory_order_acquire) != seen; });
                if (stop) return;
            }
            seen = generation.load(std::memory_order_acquire);
            work();
            running.fetch_sub(1, std::memory_order_release);
        }
    }
    int spin_limit = 20000;
    void start()
    {
        if (const char *sp = getenv("LRHIP_COPY_SPIN")) { spin_limit = atoi(sp); if (spin_limit < 0) spin_limit = 0; }
        const char *e = getenv("LRHIP_COPY_THREADS");
        int n = e ? atoi(e) : 8;
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw && (unsigned)n > (hw + 1) / 2) n = (int)((hw + 1) / 2);
        nthreads = n < 1 ? 1 : n > 16 ? 16 : n;
        pid = (long)getpid();
        for (int i = 1; i < nthreads; i++) workers.emplace_back([this] { loop(); });
        for (auto &t : workers) t.detach();        // they live as long as the process (no join at exit: the library never unloads cleanly under LuaJIT)
    }
    // fd_ >= 0: read n bytes of the file from byte offset off_ into d (returns 0, or an errno)
    int copy(void *d, const void *s_, size_t n, int fd_ = -1, long long off_ = 0)
    {
        std::unique_lock<std::mutex> one(job, std::try_to_lock);
        if (!one.owns_lock()) {
            if (fd_ < 0) { memcpy(d, s_, n); return 0; }
            one.lock();                                   // a file job has no lock-free fallback worth having: wait for the pool
        }
        dst = (char *)d; src = (const char *)s_; bytes = n;
        fd = fd_; file_off = off_;
        io_errno.store(0);
        part = ((n / (size_t)(4 * nthreads)) + 4095) & ~(size_t)4095;      // a few parts per thread, page multiples
        if (part < 65536) part = 65536;
        next.store(0);
        running.store(nthreads - 1, std::memory_order_release);
        {
            std::lock_guard<std::mutex> lk(m);
            generation.fetch_add(1, std::memory_order_release);
        }
        cv_work.notify_all();
        work();
        while (running.load(std::memory_order_acquire) != 0) cpu_relax();
        fd = -1;
        return io_errno.load();
    }
};
inline CopyPool *copy_pool()
{
    static CopyPool *pool = nullptr;
    if (!pool || pool->pid != (long)getpid()) {
        pool = new (std::nothrow) CopyPool();      // after a fork the parent's pool object is abandoned (its threads do not exist here)
        if (pool) pool->start();
    }
    return pool;
}
inline void host_copy(void *dst, const void *src, size_t bytes)
{
    if (bytes < (512u << 10)) { memcpy(dst, src, bytes); return; }
    CopyPool *pool = copy_pool();
    if (!pool || pool->nthreads <= 1) { memcpy(dst, src, bytes); return; }
    (void)pool->copy(dst, src, bytes);
}
// bytes of a regular file, from byte offset `offset`, into dst - on the copy threads when the job is large.  0, or an errno.
inline int host_pread(void *dst, int fd, long long offset, size_t bytes)
{
    CopyPool *pool = bytes >= (512u << 10) ? copy_pool() : nullptr;
    if (pool && pool->nthreads > 1) return pool->copy(dst, nullptr, bytes, fd, offset);
    size_t got = 0;
    while (got < bytes) {
        const ssize_t r = pread(fd, (char *)dst + got, bytes - got, (off_t)(offset + (long long)got));
        if (r < 0) { if (errno == EINTR) continue; return errno; }
        if (r == 0) return EIO;
        got += (size_t)r;
    }
    return 0;
}

// Ablation switches (LRHIP_RX_DBG, LRHIP_DECFFT_DBG, LRHIP_INTERP_DBG) remove parts of a kernel to time the rest: the results are WRONG by
// design.  A release build ignores them (a stray environment variable must not silently corrupt audio); `make ABLATION=1` builds the
// library that honours them, and says so on stderr the first time one is used.
inline int ablation_bits(const char *name)
{
#ifdef LRHIP_ABLATION
    const char *e = getenv(name);
    const int v = e ? atoi(e) : 0;
    if (v) {
        static bool told = false;
        if (!told) { fprintf(stderr, "liblrhip: ablation build, %s=%d - results are wrong on purpose\n", name, v); told = true; }
    }
    return v;
#else
    (void)name;
    return 0;
#endif
}

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
