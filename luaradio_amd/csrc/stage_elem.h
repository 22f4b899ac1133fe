// stage_elem.h - FrequencyTranslator, Downsampler, FrequencyDiscriminator, FrequencyModulator stages
// (part of liblrhip.so; included by lrhip.hip in this order, one translation unit)
#pragma once

// =====================================================================================================
// FrequencyTranslatorBlock
// =====================================================================================================
static uint64_t turns_fixed(double omega)
{
    long double turns = (long double)omega / (2.0L * 3.14159265358979323846264338327950288L);
    turns -= floorl(turns);
    return (uint64_t)(turns * 18446744073709551616.0L);
}

struct RotatorStage : lrhip_stage {
    double omega = 0;
    uint64_t step = 0, count = 0;
    const char *kind() const override { return "rotator"; }
    bool direct_io_ok() const override { return true; }      // round 6: one read, one write per sample (host_execute's direct mode; measured: stage_elem3.h)
    int reset() override { count = 0; return 0; }
    int seek(unsigned long long n0, unsigned long long *n0_out) override { count = n0; *n0_out = n0; return 0; }      // phase = step * absolute index
    long run(const void *in_dev, unsigned long n, void *out_dev, unsigned long cap) override
    {
        if (n > cap) return set_error("rotator: output capacity %lu < %lu", cap, n);
        if (!n) return 0;
        unsigned grid = grid_for(n, 256);
        if ((((uintptr_t)in_dev | (uintptr_t)out_dev) & 15) == 0)
            hipLaunchKernelGGL(rotator_kernel<2>, dim3(grid_for((n / 2 + 1) / 2 + 1, 256)), dim3(256), 0, ctx().stream, (const float2 *)in_dev, (float2 *)out_dev, n, step, count);
        else
            hipLaunchKernelGGL(rotator_kernel<1>, dim3(grid), dim3(256), 0, ctx().stream, (const float2 *)in_dev, (float2 *)out_dev, n, step, count);
        LR_LAUNCH_CHECK();
        count += n;
        return (long)n;
    }
};

// =====================================================================================================
// DownsamplerBlock
// =====================================================================================================
struct DownsamplerStage : lrhip_stage {
    unsigned long factor = 1, index = 0;
    const char *kind() const override { return "downsampler"; }
    bool direct_io_ok() const override { return true; }
    int reset() override { index = 0; return 0; }
    int seek(unsigned long long n0, unsigned long long *n0_out) override
    {
        index = (unsigned long)((factor - n0 % factor) % factor);       // the kept samples are the absolute indices 0 mod factor
        *n0_out = (n0 + factor - 1) / factor;
        return 0;
    }
    void rate(unsigned long *num, unsigned long *den) const override { *num = factor; *den = 1; }
    unsigned long max_output(unsigned long n) const override { return n / factor + 1; }
    long run(const void *in_dev, unsigned long n, void *out_dev, unsigned long cap) override
    {
        unsigned long n_out = n > index ? (n - index + factor - 1) / factor : 0;   // downsampler.lua:46
        if (n_out > cap) return set_error("downsampler: output capacity %lu < %lu", cap, n_out);
        if (n_out) {
            unsigned grid = grid_for(n_out, 256);
            if (in_size == 8)
                hipLaunchKernelGGL(downsample_kernel<float2>, dim3(grid), dim3(256), 0, ctx().stream, (const float2 *)in_dev, (float2 *)out_dev, n_out, index, factor);
            else
                hipLaunchKernelGGL(downsample_kernel<float>, dim3(grid), dim3(256), 0, ctx().stream, (const float *)in_dev, (float *)out_dev, n_out, index, factor);
            LR_LAUNCH_CHECK();
        }
        index = index + n_out * factor - n;                                          // downsampler.lua:53
        return (long)n_out;
    }
};

// =====================================================================================================
// FrequencyDiscriminatorBlock
// =====================================================================================================
struct FmDiscrimStage : lrhip_stage {
    double gain = 1;
    DeviceBuf prev;     // two float2 slots, ping-pong
    int cur = 0;
    const char *kind() const override { return "fmdiscrim"; }
    int reset() override { cur = 0; return zero_fill(prev, 4 * sizeof(float)); }
    long memory() const override { return 1; }
    long run(const void *in_dev, unsigned long n, void *out_dev, unsigned long cap) override
    {
        if (n > cap) return set_error("fmdiscrim: output capacity %lu < %lu", cap, n);
        if (!n) return 0;
        float2 *p = (float2 *)prev.p;
        static const bool disc_vec2 = getenv("LRHIP_DISC_VEC2") != nullptr;      // A/B knob: the 8-byte-store form
        if (!disc_vec2 && (((uintptr_t)in_dev | (uintptr_t)out_dev) & 15) == 0)
            hipLaunchKernelGGL(fmdiscrim_vec4_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, ctx().stream, (const float2 *)in_dev, (float *)out_dev, n, 1.0 / gain,
                               (const float2 *)(p + cur), p + (cur ^ 1));
        else if ((((uintptr_t)in_dev & 15) | ((uintptr_t)out_dev & 7)) == 0)
            hipLaunchKernelGGL(fmdiscrim_vec2_kernel, dim3(grid_for(n / 2 + 1, 256)), dim3(256), 0, ctx().stream, (const float2 *)in_dev, (float *)out_dev, n, 1.0 / gain,
                               (const float2 *)(p + cur), p + (cur ^ 1));
        else
            hipLaunchKernelGGL(fmdiscrim_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx().stream, (const float2 *)in_dev, (float *)out_dev, n, 1.0 / gain,
                               (const float2 *)(p + cur), p + (cur ^ 1));
        LR_LAUNCH_CHECK();
        cur ^= 1;
        return (long)n;
    }
};

struct FmModStage : lrhip_stage {
    double k = 0;
    DeviceBuf phase, tile_sum;     // phase: two uint64 slots, ping-pong
    int cur = 0;
    const char *kind() const override { return "fmmod"; }
    int reset() override { cur = 0; return zero_fill(phase, 2 * sizeof(uint64_t)); }
    long memory() const override { return -1; }      // the phase is the integral of the whole input
    long run(const void *in_dev, unsigned long n, void *out_dev, unsigned long cap) override
    {
        if (n > cap) return set_error("fmmod: output capacity %lu < %lu", cap, n);
        if (!n) return 0;
        unsigned long ntiles = (n + FMOD_TILE - 1) / FMOD_TILE;
        if (tile_sum.reserve(ntiles * sizeof(uint64_t))) return -1;
        uint64_t *ph = (uint64_t *)phase.p, *ts = (uint64_t *)tile_sum.p;
        hipLaunchKernelGGL(fmod_tile_sum_kernel, dim3((unsigned)ntiles), dim3(256), 0, ctx().stream, (const float *)in_dev, n, k, ts);
        hipLaunchKernelGGL(fmod_tile_scan_kernel, dim3(1), dim3(256), 0, ctx().stream, ts, ntiles, (const uint64_t *)(ph + cur), ph + (cur ^ 1));
        hipLaunchKernelGGL(fmod_emit_kernel, dim3((unsigned)ntiles), dim3(256), 0, ctx().stream, (const float *)in_dev, (float2 *)out_dev, n, k,
                           (const uint64_t *)ts);
        LR_LAUNCH_CHECK();
        cur ^= 1;
        return (long)n;
    }
};

// P x P matrix helpers (double, host) for the transition powers
static void matmul(const std::vector<double> &A, const std::vector<double> &B, std::vector<double> &C, int P)
{
    std::vector<double> T((size_t)P * P, 0.0);
    for (int r = 0; r < P; r++)
        for (int c = 0; c < P; c++) {
            double acc = 0;
            for (int k = 0; k < P; k++) acc += A[r * P + k] * B[k * P + c];
            T[r * P + c] = acc;
        }
    C = T;
}
