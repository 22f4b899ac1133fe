// stage_elem2.h - file sample formats, two-input blocks, MultiplyConstant, Upsampler
// (part of liblrhip.so; included by lrhip.hip in this order, one translation unit)
#pragma once

// =====================================================================================================
// IQFileSource / RealFileSource format conversion
// =====================================================================================================
struct FormatStage : lrhip_stage {
    int fmt = 0;          // index into kFormats
    int scalars = 1;      // raw scalars per sample (2 for I/Q)
    bool pack = false;    // sink direction: Float32 / ComplexFloat32 -> raw records
    const char *kind() const override { return "format"; }
    int reset() override { return 0; }
    long run(const void *in_dev, unsigned long n, void *out_dev, unsigned long cap) override;
};

struct FormatDesc {
    const char *name;
    int bytes;        // per scalar
    int cls;          // 0 u8, 1 s8, 2 u16, 3 s16, 4 u32, 5 s32, 6 f32, 7 f64
    bool swap;        // file byte order differs from the (little-endian) device
    double offset, scale;
};
// radio/utilities/format_utils.lua:82-97
static const FormatDesc kFormats[] = {
    {"u8", 1, 0, false, 127.5, 127.5},           {"s8", 1, 1, false, 0.0, 127.5},
    {"u16le", 2, 2, false, 32767.5, 32767.5},    {"u16be", 2, 2, true, 32767.5, 32767.5},
    {"s16le", 2, 3, false, 0.0, 32767.5},        {"s16be", 2, 3, true, 0.0, 32767.5},
    {"u32le", 4, 4, false, 2147483647.5, 2147483647.5}, {"u32be", 4, 4, true, 2147483647.5, 2147483647.5},
    {"s32le", 4, 5, false, 0.0, 2147483647.5},   {"s32be", 4, 5, true, 0.0, 2147483647.5},
    {"f32le", 4, 6, false, 0.0, 1.0},            {"f32be", 4, 6, true, 0.0, 1.0},
    {"f64le", 8, 7, false, 0.0, 1.0},            {"f64be", 8, 7, true, 0.0, 1.0},
};

long FormatStage::run(const void *in_dev, unsigned long n, void *out_dev, unsigned long cap)
{
    if (n > cap) return set_error("format: output capacity %lu < %lu", cap, n);
    if (!n) return 0;
    const FormatDesc &f = kFormats[fmt];
    unsigned long ns = n * scalars;
    unsigned grid = grid_for(ns, 256);
    float *out = (float *)out_dev;
    if (pack) {
#define LR_PACK(RAW, VAL)                                                                                                  \
    do {                                                                                                                   \
        if (f.swap) hipLaunchKernelGGL((format_pack_kernel<RAW, VAL, true>), dim3(grid), dim3(256), 0, ctx().stream, (const float *)in_dev, (RAW *)out_dev, ns, f.offset, f.scale); \
        else hipLaunchKernelGGL((format_pack_kernel<RAW, VAL, false>), dim3(grid), dim3(256), 0, ctx().stream, (const float *)in_dev, (RAW *)out_dev, ns, f.offset, f.scale);      \
    } while (0)
        switch (f.cls) {
            case 0: LR_PACK(uint8_t, uint8_t); break;
            case 1: LR_PACK(uint8_t, int8_t); break;
            case 2: LR_PACK(uint16_t, uint16_t); break;
            case 3: LR_PACK(uint16_t, int16_t); break;
            case 4: LR_PACK(uint32_t, uint32_t); break;
            case 5: LR_PACK(uint32_t, int32_t); break;
            case 6: LR_PACK(uint32_t, float); break;
            default: LR_PACK(uint64_t, double); break;
        }
#undef LR_PACK
        LR_LAUNCH_CHECK();
        return (long)n;
    }
    static const bool no_vec = getenv("LRHIP_ELEM_SCALAR") != nullptr;      // A/B knob: one scalar per thread (round 2)
    const size_t in_align = (size_t)(f.bytes >= 8 ? 16 : 4 * f.bytes);
    if (!no_vec && ((uintptr_t)in_dev % in_align) == 0 && ((uintptr_t)out_dev % 16) == 0) {
        const unsigned long items = ns / 4;
        const unsigned vg = grid_for(items + 1, 256);
#define LR_FMTV(RAW, VAL)                                                                                                  \
    do {                                                                                                                   \
        if (f.swap) hipLaunchKernelGGL((format_convert_vec_kernel<RAW, VAL, true>), dim3(vg), dim3(256), 0, ctx().stream, (const RAW *)in_dev, out, items, ns, f.offset, f.scale); \
        else hipLaunchKernelGGL((format_convert_vec_kernel<RAW, VAL, false>), dim3(vg), dim3(256), 0, ctx().stream, (const RAW *)in_dev, out, items, ns, f.offset, f.scale);      \
    } while (0)
        switch (f.cls) {
            case 0: LR_FMTV(uint8_t, uint8_t); break;
            case 1: LR_FMTV(uint8_t, int8_t); break;
            case 2: LR_FMTV(uint16_t, uint16_t); break;
            case 3: LR_FMTV(uint16_t, int16_t); break;
            case 4: LR_FMTV(uint32_t, uint32_t); break;
            case 5: LR_FMTV(uint32_t, int32_t); break;
            case 6: LR_FMTV(uint32_t, float); break;
            default: LR_FMTV(uint64_t, double); break;
        }
#undef LR_FMTV
        LR_LAUNCH_CHECK();
        return (long)n;
    }
#define LR_FMT(RAW, VAL)                                                                                                   \
    do {                                                                                                                   \
        if (f.swap) hipLaunchKernelGGL((format_convert_kernel<RAW, VAL, true>), dim3(grid), dim3(256), 0, ctx().stream, (const RAW *)in_dev, out, ns, f.offset, f.scale); \
        else hipLaunchKernelGGL((format_convert_kernel<RAW, VAL, false>), dim3(grid), dim3(256), 0, ctx().stream, (const RAW *)in_dev, out, ns, f.offset, f.scale);      \
    } while (0)
    switch (f.cls) {
        case 0: LR_FMT(uint8_t, uint8_t); break;
        case 1: LR_FMT(uint8_t, int8_t); break;
        case 2: LR_FMT(uint16_t, uint16_t); break;
        case 3: LR_FMT(uint16_t, int16_t); break;
        case 4: LR_FMT(uint32_t, uint32_t); break;
        case 5: LR_FMT(uint32_t, int32_t); break;
        case 6: LR_FMT(uint32_t, float); break;
        default: LR_FMT(uint64_t, double); break;
    }
#undef LR_FMT
    LR_LAUNCH_CHECK();
    return (long)n;
}

// =====================================================================================================
// MultiplyBlock / MultiplyConjugateBlock / AddBlock / SubtractBlock
// =====================================================================================================
struct BinaryStage : lrhip_stage {
    int op = BIN_MULTIPLY;
    PinnedBuf h_in2;
    DeviceBuf d_in2;
    const char *kind() const override { return "binary"; }
    int reset() override { return 0; }
    long run(const void *, unsigned long, void *, unsigned long) override { return set_error("binary stage needs two inputs: use lrhip_stage_execute2"); }
    long run2(const void *a, const void *b, unsigned long n, void *y, unsigned long cap) override
    {
        if (n > cap) return set_error("binary: output capacity %lu < %lu", cap, n);
        if (!n) return 0;
        if (op == BIN_F2C) {
            if (((uintptr_t)a % 8) == 0 && ((uintptr_t)b % 8) == 0 && ((uintptr_t)y % 16) == 0)
                hipLaunchKernelGGL(float_to_complex_vec_kernel, dim3(grid_for(n / 2 + 1, 256)), dim3(256), 0, ctx().stream, (const float2 *)a, (const float2 *)b, (float4 *)y, n / 2, n);
            else
                hipLaunchKernelGGL(float_to_complex_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx().stream, (const float *)a, (const float *)b, (float2 *)y, n);
            LR_LAUNCH_CHECK();
            return (long)n;
        }
        // 16 B per lane when the three vectors allow it (the odd tail, at most 3 floats, goes through the scalar kernels below)
        const unsigned long nf = n * (in_size / 4), nf4 = nf & ~3ul;
        if (nf4 && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)y) & 15) == 0) {
            unsigned g4 = grid_for(nf4 / 4, 256);
#define LR_BIN4(OP, CM) hipLaunchKernelGGL((binary_vec4_kernel<OP, CM>), dim3(g4), dim3(256), 0, ctx().stream, (const float *)a, (const float *)b, (float *)y, nf4)
            if (op == BIN_ADD) LR_BIN4(BIN_ADD, 0);
            else if (op == BIN_SUBTRACT) LR_BIN4(BIN_SUBTRACT, 0);
            else if (in_size == 4) LR_BIN4(BIN_MULTIPLY, 0);
            else if (op == BIN_MULTIPLY) LR_BIN4(BIN_MULTIPLY, 1);
            else LR_BIN4(BIN_MULTIPLY_CONJ, 2);
#undef LR_BIN4
            LR_LAUNCH_CHECK();
            const unsigned long done = nf4 / (in_size / 4);        // samples covered
            if (done == n) return (long)n;
            a = (const char *)a + done * in_size; b = (const char *)b + done * in_size; y = (char *)y + done * in_size;
            const unsigned long rest = n - done;
            long r = run2_scalar(a, b, rest, y);
            return r < 0 ? r : (long)n;
        }
        long r = run2_scalar(a, b, n, y);
        return r < 0 ? r : (long)n;
    }
    long run2_scalar(const void *a, const void *b, unsigned long n, void *y)
    {
        unsigned grid = grid_for(n, 256);
#define LR_BIN(K, OP, T) hipLaunchKernelGGL((K<OP>), dim3(grid), dim3(256), 0, ctx().stream, (const T *)a, (const T *)b, (T *)y, n)
        if (in_size == 8) {
            switch (op) {
                case BIN_MULTIPLY: LR_BIN(binary_complex_kernel, BIN_MULTIPLY, float2); break;
                case BIN_MULTIPLY_CONJ: LR_BIN(binary_complex_kernel, BIN_MULTIPLY_CONJ, float2); break;
                case BIN_ADD: LR_BIN(binary_complex_kernel, BIN_ADD, float2); break;
                default: LR_BIN(binary_complex_kernel, BIN_SUBTRACT, float2); break;
            }
        } else {
            switch (op) {
                case BIN_MULTIPLY: LR_BIN(binary_real_kernel, BIN_MULTIPLY, float); break;
                case BIN_ADD: LR_BIN(binary_real_kernel, BIN_ADD, float); break;
                default: LR_BIN(binary_real_kernel, BIN_SUBTRACT, float); break;
            }
        }
#undef LR_BIN
        LR_LAUNCH_CHECK();
        return (long)n;
    }
};

// =====================================================================================================
// MultiplyConstantBlock, UpsamplerBlock
// =====================================================================================================
struct MulConstStage : lrhip_stage {
    float cr = 1.f, ci = 0.f;
    int mode = 0;
    const char *kind() const override { return "multiplyconstant"; }
    int reset() override { return 0; }
    long run(const void *in_dev, unsigned long n, void *out_dev, unsigned long cap) override
    {
        if (n > cap) return set_error("multiplyconstant: output capacity %lu < %lu", cap, n);
        if (!n) return 0;
        const float *x = (const float *)in_dev;
        float *y = (float *)out_dev;
        const unsigned long nf = n * (in_size / 4), nf4 = nf & ~3ul;
        if (nf4 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
            unsigned g4 = grid_for(nf4 / 4, 256);
            if (mode != 2) hipLaunchKernelGGL(multiply_constant_vec4_kernel<0>, dim3(g4), dim3(256), 0, ctx().stream, x, y, nf4, cr, ci);
            else hipLaunchKernelGGL(multiply_constant_vec4_kernel<2>, dim3(g4), dim3(256), 0, ctx().stream, x, y, nf4, cr, ci);
            LR_LAUNCH_CHECK();
            const unsigned long done = nf4 / (in_size / 4);
            if (done == n) return (long)n;
            long r = run_scalar(x + nf4, y + nf4, n - done);      // the odd tail (< 4 floats)
            return r < 0 ? r : (long)n;
        }
        return run_scalar(x, y, n);
    }
    long run_scalar(const float *x, float *y, unsigned long n)
    {
        unsigned grid = grid_for(n, 256);
        if (mode == 0) hipLaunchKernelGGL(multiply_constant_kernel<0>, dim3(grid), dim3(256), 0, ctx().stream, x, y, n, cr, ci);
        else if (mode == 1) hipLaunchKernelGGL(multiply_constant_kernel<1>, dim3(grid), dim3(256), 0, ctx().stream, x, y, n, cr, ci);
        else hipLaunchKernelGGL(multiply_constant_kernel<2>, dim3(grid), dim3(256), 0, ctx().stream, x, y, n, cr, ci);
        LR_LAUNCH_CHECK();
        return (long)n;
    }
};

struct UpsamplerStage : lrhip_stage {
    unsigned long factor = 1;
    const char *kind() const override { return "upsampler"; }
    int reset() override { return 0; }
    int seek(unsigned long long n0, unsigned long long *n0_out) override { *n0_out = n0 * factor; return 0; }
    void rate(unsigned long *num, unsigned long *den) const override { *num = 1; *den = factor; }
    unsigned long max_output(unsigned long n) const override { return n * factor; }
    long run(const void *in_dev, unsigned long n, void *out_dev, unsigned long cap) override
    {
        unsigned long n_out = n * factor;                 // upsampler.lua:46
        if (n_out > cap) return set_error("upsampler: output capacity %lu < %lu", cap, n_out);
        if (!n_out) return 0;
        static const bool no_vec = getenv("LRHIP_ELEM_SCALAR") != nullptr;      // A/B knob: one output sample per thread (round 2)
        const unsigned long per = 16 / in_size;
        if (!no_vec && ((uintptr_t)out_dev % 16) == 0) {
            const unsigned long items = n_out / per;
            if (in_size == 8)
                hipLaunchKernelGGL((upsample_vec_kernel<float2, 2>), dim3(grid_for(items + 1, 256 * UPS_U)), dim3(256), 0, ctx().stream, (const float2 *)in_dev, (float4 *)out_dev, items, factor, n_out);
            else
                hipLaunchKernelGGL((upsample_vec_kernel<float, 4>), dim3(grid_for(items + 1, 256 * UPS_U)), dim3(256), 0, ctx().stream, (const float *)in_dev, (float4 *)out_dev, items, factor, n_out);
            LR_LAUNCH_CHECK();
            return (long)n_out;
        }
        unsigned grid = grid_for(n_out, 256);
        if (in_size == 8)
            hipLaunchKernelGGL(upsample_kernel<float2>, dim3(grid), dim3(256), 0, ctx().stream, (const float2 *)in_dev, (float2 *)out_dev, n_out, factor);
        else
            hipLaunchKernelGGL(upsample_kernel<float>, dim3(grid), dim3(256), 0, ctx().stream, (const float *)in_dev, (float *)out_dev, n_out, factor);
        LR_LAUNCH_CHECK();
        return (long)n_out;
    }
};
