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
