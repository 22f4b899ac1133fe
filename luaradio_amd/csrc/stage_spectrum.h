// stage_spectrum.h - DFT / IDFT / PSD stage and the Welch-averaged spectrum
// (part of liblrhip.so; included by lrhip.hip in this order, one translation unit)
#pragma once

// =====================================================================================================
// DFT / IDFT / PSD
// =====================================================================================================
struct FftStage : lrhip_stage {
    int seek(unsigned long long, unsigned long long *) override { return set_error("seek: not supported by framed (spectrum) stages"); }
    long memory() const override { return -1; }
    int N = 0, inverse = 0, out_kind = FFT_OUT_COMPLEX, shift = 0, in_real = 0, fpw = 1;
    float out_scale = 1.f;
    bool has_window = false;
    DeviceBuf tw, window, spec_tables;    // spec_tables: tw1 | tw2 of the one-wave-per-frame N = 1024 engine
    int spec_blocks_per_cu = 0;
    const char *kind() const override { return "fft"; }
    int reset() override { return 0; }
    unsigned long max_output(unsigned long n) const override { return n - n % N; }
    long run(const void *in_dev, unsigned long n, void *out_dev, unsigned long cap) override
    {
        if (n % N) return set_error("fft: input length %lu is not a multiple of the frame length %d", n, N);
        if (n > cap) return set_error("fft: output capacity %lu < %lu", cap, n);
        if (!n) return 0;
        long nframes = (long)(n / N);
        if (N == FFTN) {
            size_t lds_bytes = (size_t)SPEC_LDS_ELEMS * sizeof(float2);
            int mode = inverse ? (out_kind == FFT_OUT_REAL ? SPEC_INV_REAL : SPEC_INV_COMPLEX)
                               : (out_kind == FFT_OUT_PSD ? SPEC_FWD_PSD : out_kind == FFT_OUT_PSD_LOG ? SPEC_FWD_PSD_LOG : SPEC_FWD_COMPLEX);
            const float *wp = has_window ? (const float *)window.p : nullptr;
            auto go = [&](auto kern) -> int {
                if (!spec_blocks_per_cu) {
                    if (lds_bytes > 48 * 1024) LR_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                    int nb = 0;
                    LR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds_bytes));
                    spec_blocks_per_cu = nb < 1 ? 1 : nb;
                }
                long slots = (long)ctx().num_cus * spec_blocks_per_cu, want = (nframes + 3) / 4;
                static const int rounds_env = getenv("LRHIP_SPEC_ROUNDS") ? atoi(getenv("LRHIP_SPEC_ROUNDS")) : -1;      // A/B knob
                int rounds = rounds_env >= 0 ? rounds_env : 0;      // measured, 2^26 samples, same box: persistent 0.1572 ms, one-shot with 4 / 8 / 16 batches per workgroup 0.1628 / 0.159 / 0.1694
                if (want <= slots) rounds = 0;
                unsigned g = rounds > 0 ? (unsigned)((want + rounds - 1) / rounds) : (unsigned)(want < slots ? want : slots);
                hipLaunchKernelGGL(kern, dim3(g), dim3(256), lds_bytes, ctx().stream, (const float *)in_dev, (float *)out_dev, nframes,
                                   (const float2 *)spec_tables.p, wp, mode, out_scale, shift, rounds);
                return 0;
            };
            int rc = in_real ? go(spectrum1024_kernel<true>) : go(spectrum1024_kernel<false>);
            if (rc) return rc;
            LR_LAUNCH_CHECK();
            return (long)n;
        }
        unsigned grid = (unsigned)((nframes + fpw - 1) / fpw);
        size_t lds = ((size_t)2 * fpw * N + N / 2) * sizeof(float2);
        const float *w = has_window ? (const float *)window.p : nullptr;
        if (in_real) {
            auto kern = fft_frames_kernel<true>;
            if (lds > 48 * 1024) LR_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, ctx().stream, (const float *)in_dev, (float *)out_dev, nframes, N, fpw,
                               (const float2 *)tw.p, w, inverse, out_kind, out_scale, shift);
        } else {
            auto kern = fft_frames_kernel<false>;
            if (lds > 48 * 1024) LR_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, ctx().stream, (const float *)in_dev, (float *)out_dev, nframes, N, fpw,
                               (const float2 *)tw.p, w, inverse, out_kind, out_scale, shift);
        }
        LR_LAUNCH_CHECK();
        return (long)n;
    }
};

static FftStage *fft_build(unsigned n)
{
    if (n < 8 || n > 4096 || (n & (n - 1))) { set_error("fft: frame length must be a power of two in [8, 4096] (got %u)", n); return nullptr; }
    if (ensure_init()) return nullptr;
    std::unique_ptr<FftStage> q(new (std::nothrow) FftStage());
    if (!q) { set_error("out of memory"); return nullptr; }
    q->N = (int)n;
    q->fpw = n >= 512 ? 1 : (int)(512 / n);
    std::vector<float> tw(n);   // n/2 complex
    for (unsigned m = 0; m < n / 2; m++) {
        double ang = -2.0 * 3.14159265358979323846 * m / n;
        tw[2 * m] = (float)std::cos(ang);
        tw[2 * m + 1] = (float)std::sin(ang);
    }
    if (upload(q->tw, tw.data(), tw.size() * sizeof(float))) return nullptr;
    if (n == FFTN) {
        const double PI2 = 6.283185307179586476925286766559;
        std::vector<float> tab((size_t)SPEC_TABLE_ELEMS * 2);
        for (int k1 = 0; k1 < 16; k1++)
            for (int t = 0; t < 64; t++) {
                double a = -PI2 * (double)((k1 * t) % FFTN) / FFTN;
                tab[2 * (k1 * 64 + t)] = (float)std::cos(a);
                tab[2 * (k1 * 64 + t) + 1] = (float)std::sin(a);
            }
        for (int k2 = 0; k2 < 16; k2++)
            for (int t2 = 0; t2 < 4; t2++) {
                double a = -PI2 * (double)((k2 * t2) % 64) / 64.0;
                size_t o = (size_t)16 * 64 + k2 * 4 + t2;
                tab[2 * o] = (float)std::cos(a);
                tab[2 * o + 1] = (float)std::sin(a);
            }
        if (upload(q->spec_tables, tab.data(), tab.size() * sizeof(float))) return nullptr;
    }
    return q.release();
}

// =====================================================================================================
// Welch-averaged spectrum (the arithmetic of GnuplotSpectrumSink)
// =====================================================================================================
struct WelchStage : lrhip_stage {
    int seek(unsigned long long, unsigned long long *) override { return set_error("seek: not supported by framed (spectrum) stages"); }
    long memory() const override { return -1; }
    std::unique_ptr<FftStage> psd;
    int N = 0, hop = 0;
    unsigned long P = 0;            // pending samples (< N once a frame could be cut)
    long count = 0;                 // frames accumulated
    DeviceBuf pending[2], frames, spectra, partial, sum;
    PinnedBuf h_avg;
    int cur = 0;
    const char *kind() const override { return "welch"; }
    unsigned long max_output(unsigned long) const override { return 0; }
    int clear()
    {
        count = 0;
        return zero_fill(sum, (size_t)N * sizeof(float));
    }
    int reset() override
    {
        P = 0; cur = 0;
        if (pending[0].reserve((size_t)N * in_size) || pending[1].reserve((size_t)N * in_size)) return -1;
        return clear();
    }
    template <typename T>
    long run_t(const T *x, unsigned long n)
    {
        unsigned long total = P + n;
        unsigned long nf = total >= (unsigned long)N ? (total - N) / hop + 1 : 0;
        const T *pend = (const T *)pending[cur].p;
        if (nf) {
            const void *frames_in;
            if (P == 0 && hop == N) frames_in = x;             // contiguous frames: no gather
            else {
                if (frames.reserve(nf * N * sizeof(T))) return -1;
                hipLaunchKernelGGL(welch_gather_kernel<T>, dim3(grid_for(nf * N, 256)), dim3(256), 0, ctx().stream, pend, P, x,
                                   (T *)frames.p, nf, N, hop);
                LR_LAUNCH_CHECK();
                frames_in = frames.p;
            }
            if (spectra.reserve(nf * N * sizeof(float))) return -1;
            long got = psd->run(frames_in, nf * N, spectra.p, nf * N);
            if (got < 0) return got;
            unsigned long nchunks = (nf + WELCH_CHUNK - 1) / WELCH_CHUNK;
            if (partial.reserve(nchunks * N * sizeof(float))) return -1;
            if (nchunks > 65535) return set_error("welch: more than %d frames in one call", 65535 * WELCH_CHUNK);
            dim3 g((N + 255) / 256, (unsigned)nchunks);
            hipLaunchKernelGGL(welch_partial_kernel, g, dim3(256), 0, ctx().stream, (const float *)spectra.p, (float *)partial.p, nf, N);
            hipLaunchKernelGGL(welch_final_kernel, dim3((N + 255) / 256), dim3(256), 0, ctx().stream, (const float *)partial.p, nchunks, (float *)sum.p, N);
            LR_LAUNCH_CHECK();
            count += (long)nf;
        }
        // what is left after the last frame start + hop: the overlap of the last frame plus the unconsumed tail
        unsigned long start = nf * hop, left = total - start;
        if (left) {
            hipLaunchKernelGGL(welch_pending_kernel<T>, dim3((unsigned)((left + 255) / 256)), dim3(256), 0, ctx().stream, pend, P, x, start,
                               (T *)pending[cur ^ 1].p, left);
            LR_LAUNCH_CHECK();
        }
        cur ^= 1;
        P = left;
        return 0;
    }
    long run(const void *in_dev, unsigned long n, void *, unsigned long) override
    {
        if (!n) return 0;
        return in_size == 8 ? run_t<float2>((const float2 *)in_dev, n) : run_t<float>((const float *)in_dev, n);
    }
};
