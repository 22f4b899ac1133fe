// stage.h - stage base: lrhip_stage (what every block object derives from) and the launch bookkeeping
// (part of liblrhip.so; included by lrhip.hip in this order, one translation unit)
#pragma once

// =====================================================================================================
// stage base
// =====================================================================================================
struct lrhip_stage {
    int in_size = 8, out_size = 8;     // bytes per sample
    PinnedBuf h_in, h_out;             // pinned staging for the host-pointer execute
    DeviceBuf d_in, d_out;
    virtual ~lrhip_stage() {}
    virtual unsigned long max_output(unsigned long n_in) const { return n_in; }
    virtual long run(const void *in_dev, unsigned long n_in, void *out_dev, unsigned long cap) = 0;
    virtual long run2(const void *, const void *, unsigned long, void *, unsigned long) { return set_error("%s is not a two-input stage", kind()); }
    virtual int reset() = 0;
    virtual const char *kind() const = 0;
    // ---- time-axis sharding of one stream: a partition that starts at absolute input sample n0
    // seek(): forget every carried sample (as reset()) and set the absolute counters - rotator phase, decimation phase - as if n0 input
    // samples had been consumed; *n0_out = output samples this stage has emitted by then (the next stage's n0).
    virtual int seek(unsigned long long n0, unsigned long long *n0_out)
    {
        if (reset()) return -1;
        *n0_out = n0;
        return 0;
    }
    // memory(): input samples after which a stage started from zero state carries exactly what the uninterrupted stream would
    // (filter history, previous sample; a recurrence whose zero start has decayed out of Float32); -1 = unbounded (no time sharding)
    virtual long memory() const { return 0; }
    // align(): partitions that start on a multiple of this many input samples of the stage see the same tile grid in its scan kernels as the
    // uninterrupted stream, which makes a recurrence's output bit-identical as well (1: the stage has no such grid)
    virtual unsigned long align() const { return 1; }
    // round 5: the stage reads its input ONCE and writes its output once, so the host-pointer entry points may hand it the caller's registered host memory
    // itself (kernels load and store across the link; host_execute's direct mode in chain.h) instead of staging through device buffers
    virtual bool direct_io_ok() const { return false; }
    // input samples per output sample as a ratio (decimation num/den), for mapping memories back to the chain input
    virtual void rate(unsigned long *num, unsigned long *den) const { *num = 1; *den = 1; }
};

static int upload(DeviceBuf &b, const void *src, size_t bytes)
{
    if (b.reserve(bytes ? bytes : 4)) return -1;
    if (bytes) LR_HIP(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    return 0;
}
static int zero_fill(DeviceBuf &b, size_t bytes)
{
    if (b.reserve(bytes ? bytes : 4)) return -1;
    LR_HIP(hipMemsetAsync(b.p, 0, bytes ? bytes : 4, ctx().stream));
    return 0;
}
