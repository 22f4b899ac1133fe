// stage_elem3.h - one-input element-wise blocks, Delay, HilbertTransform
// (part of liblrhip.so; included by lrhip.hip in this order, one translation unit)
#pragma once

// =====================================================================================================
// one-input element-wise blocks, DelayBlock, HilbertTransformBlock
// =====================================================================================================
struct UnaryStage : lrhip_stage {
    int op = 0;
    float cr = 0.f, ci = 0.f;
    const char *kind() const override { return "unary"; }
    // round 6, host_execute's direct mode (registered vectors read and written across the link by the kernel itself): measured against the staged piece
    // pipeline on one box (tools/ab_direct_elem.py, profiles/r06_ab_direct_elem.txt) it pays for the translator (+14 % at 2^20-sample vectors) and the
    // stages that write less than they read (Downsampler +11 %, complex -> real +3-6 %); 1 : 1 kernels without arithmetic (MultiplyConstant, conjugate)
    // are equal at 2^20 and 7 % SLOWER at 2^24 - they stay staged
    bool direct_io_ok() const override { return out_size < in_size; }
    int reset() override { return 0; }
    long run(const void *in_dev, unsigned long n, void *out_dev, unsigned long cap) override
    {
        if (n > cap) return set_error("unary: output capacity %lu < %lu", cap, n);
        if (!n) return 0;
        const float *x = (const float *)in_dev;
        float *y = (float *)out_dev;
        // 16-byte accesses (the one-sample kernel when a pointer is not 16-byte aligned)
        static const bool no_vec = getenv("LRHIP_ELEM_SCALAR") != nullptr;      // A/B knob: one sample per thread (round 2)
        if (!no_vec && ((uintptr_t)in_dev % 16) == 0 && ((uintptr_t)out_dev % 16) == 0) {
            const unsigned long per = (unsigned long)unary_vec_samples(op), items = n / per;
            const unsigned vg = grid_for(items + 1, 256);
#define LR_UNV(OP) case OP: hipLaunchKernelGGL(unary_vec_kernel<OP>, dim3(vg), dim3(256), 0, ctx().stream, x, y, items, n, cr, ci); break
            switch (op) {
                LR_UNV(UN_CMAG); LR_UNV(UN_CPHASE); LR_UNV(UN_CREAL); LR_UNV(UN_CIMAG); LR_UNV(UN_CCONJ); LR_UNV(UN_R2C); LR_UNV(UN_ABS);
                LR_UNV(UN_ADDC_REAL); LR_UNV(UN_ADDC_CPLX_BY_REAL); LR_UNV(UN_ADDC_CPLX);
                default: return set_error("unary: bad op");
            }
#undef LR_UNV
            LR_LAUNCH_CHECK();
            return (long)n;
        }
        return run_scalar(x, y, n);
    }
    long run_scalar(const float *x, float *y, unsigned long n)
    {
        unsigned grid = grid_for(n, 256);
#define LR_UN(OP) case OP: hipLaunchKernelGGL(unary_kernel<OP>, dim3(grid), dim3(256), 0, ctx().stream, x, y, n, cr, ci); break
        switch (op) {
            LR_UN(UN_CMAG); LR_UN(UN_CPHASE); LR_UN(UN_CREAL); LR_UN(UN_CIMAG); LR_UN(UN_CCONJ); LR_UN(UN_R2C); LR_UN(UN_ABS);
            LR_UN(UN_ADDC_REAL); LR_UN(UN_ADDC_CPLX_BY_REAL); LR_UN(UN_ADDC_CPLX);
            default: return set_error("unary: bad op");
        }
#undef LR_UN
        LR_LAUNCH_CHECK();
        return (long)n;
    }
};

struct DelayStage : lrhip_stage {
    unsigned long D = 1;
    DeviceBuf state[2];
    int cur = 0;
    const char *kind() const override { return "delay"; }
    long memory() const override { return (long)D; }
    int reset() override
    {
        cur = 0;
        return (zero_fill(state[0], D * in_size) || zero_fill(state[1], D * in_size)) ? -1 : 0;
    }
    long run(const void *in_dev, unsigned long n, void *out_dev, unsigned long cap) override
    {
        if (n > cap) return set_error("delay: output capacity %lu < %lu", cap, n);
        if (!n) return 0;
        static const bool no_vec = getenv("LRHIP_ELEM_SCALAR") != nullptr;
        if (!no_vec && (D * in_size) % 16 == 0 && (n * in_size) % 16 == 0 && ((uintptr_t)in_dev % 16) == 0 && ((uintptr_t)out_dev % 16) == 0) {
            const unsigned long n4 = n * in_size / 16, D4 = D * in_size / 16;
            hipLaunchKernelGGL(delay_vec_kernel, dim3(grid_for(n4 + D4, 256)), dim3(256), 0, ctx().stream, (const float4 *)state[cur].p, (const float4 *)in_dev,
                               (float4 *)out_dev, (float4 *)state[cur ^ 1].p, n4, D4);
            LR_LAUNCH_CHECK();
            cur ^= 1;
            return (long)n;
        }
        unsigned grid = grid_for(n + D, 256);
        if (in_size == 8)
            hipLaunchKernelGGL(delay_kernel<float2>, dim3(grid), dim3(256), 0, ctx().stream, (const float2 *)state[cur].p, (const float2 *)in_dev,
                               (float2 *)out_dev, (float2 *)state[cur ^ 1].p, n, D);
        else
            hipLaunchKernelGGL(delay_kernel<float>, dim3(grid), dim3(256), 0, ctx().stream, (const float *)state[cur].p, (const float *)in_dev,
                               (float *)out_dev, (float *)state[cur ^ 1].p, n, D);
        LR_LAUNCH_CHECK();
        cur ^= 1;
        return (long)n;
    }
};

struct HilbertStage : lrhip_stage {
    std::unique_ptr<FirStage> fir;     // real taps, Float32 stream: the imaginary part
    DeviceBuf tmp;
    const char *kind() const override { return "hilbert"; }
    long memory() const override { return fir->M - 1; }
    int reset() override { return fir->reset(); }
    long run(const void *in_dev, unsigned long n, void *out_dev, unsigned long cap) override
    {
        if (n > cap) return set_error("hilbert: output capacity %lu < %lu", cap, n);
        if (!n) return 0;
        static const bool two_pass = getenv("LRHIP_HILBERT_TWO_PASS") != nullptr;      // A/B knob: filter, then combine (round 2)
        if (!two_pass && fir->hilbert_ok() && ((uintptr_t)in_dev % 4) == 0) {
            // one launch: the filter's epilogue writes (delayed input, filtered input) pairs (hilberttransform.lua:107-124 is one loop as well)
            if (fir->launch_hilbert((const float *)in_dev, (long)n, (float *)out_dev)) return -1;
            return (long)n;
        }
        if (tmp.reserve(n * sizeof(float))) return -1;
        long got = fir->core((const float *)in_dev, (long)n, (float *)tmp.p, n);
        if (got < 0) return got;
        // core() has swapped the ping-pong history: the history that was current for this chunk is the other one
        const float *old_hist = (const float *)fir->hist[fir->cur ^ 1].p;
        unsigned grid = grid_for(n, 256);
        hipLaunchKernelGGL(hilbert_combine_kernel, dim3(grid), dim3(256), 0, ctx().stream, old_hist, (const float *)in_dev, (const float *)tmp.p,
                           (float2 *)out_dev, n, fir->M);
        LR_LAUNCH_CHECK();
        return (long)n;
    }
};
