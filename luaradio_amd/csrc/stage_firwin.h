// stage_firwin.h - stages on the register-window FIR kernels (kernels_firwin.h)
// (part of liblrhip.so; included by lrhip.hip in this order, one translation unit)
#pragma once

// =====================================================================================================
// FIRFilterBlock (Float32 stream, real taps) -> first-order IIRFilterBlock -> [DownsamplerBlock], one launch:
// the audio tail of the WBFM receiver (LowpassFilter(128) -> FMDeemphasisFilter -> Downsampler(5),
// examples/rtlsdr_wbfm_mono.lua:14-16).  IIR = false: the plain Float32 filter.
// =====================================================================================================
struct FirWinRealStage : lrhip_stage {
    int M = 0;
    bool iir = false;
    std::vector<float> taps_rev;
    DeviceBuf d_taps, d_ptab, hist[2], st[2];      // st: {y[-1], v[-1]} ping-pong
    int cur = 0;
    float b0 = 0.f, b1 = 0.f, na1 = 0.f;
    int nb = 1, warm_waves = 4;
    unsigned long D = 1, index = 0;
    int blocks_per_cu = 0;
    const char *kind() const override { return iir ? "fir+iir" : "firwin"; }
    unsigned long max_output(unsigned long n) const override { return D == 1 ? n : n / D + 1; }
    int seek(unsigned long long n0, unsigned long long *n0_out) override
    {
        if (reset()) return -1;
        index = (unsigned long)((D - n0 % D) % D);
        *n0_out = (n0 + D - 1) / D;
        return 0;
    }
    long memory() const override { return M - 1 + (iir ? 1024L * warm_waves + 1 : 0); }
    void rate(unsigned long *num, unsigned long *den) const override { *num = D; *den = 1; }
    unsigned long align() const override { return iir ? (unsigned long)FWR_TILE : 1UL; }
    int reset() override
    {
        cur = 0; index = 0;
        for (int i = 0; i < 2; i++)
            if (zero_fill(hist[i], sizeof(float) * (size_t)(M > 1 ? M - 1 : 1)) || zero_fill(st[i], 4 * sizeof(float))) return -1;
        return 0;
    }
    static bool supported_taps(int m) { return m == 32 || m == 64 || m == 128; }
    // is the recurrence's memory short enough for the in-launch warm-up?  |a1|^(1024 w) < 1e-12 for w in {1, 2, 4}
    static int warm_waves_for(double a1)
    {
        const double p = std::fabs(a1);
        if (!(p < 1.0)) return 0;
        for (int w = 1; w <= 4; w *= 2)
            if (std::pow(p, 1024.0 * w) < 1e-12) return w;
        return 0;
    }
    template <int MM, bool II>
    int launch(const float *x, long n, float *y)
    {
        using G = FwrGeom<MM>;
        const size_t lds_bytes = (size_t)G::LDS_FLOATS * sizeof(float);
        auto kern = fir_win_real_kernel<MM, II>;
        if (!blocks_per_cu) {
            if (lds_bytes > 48 * 1024) LR_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            int nb_ = 0;
            LR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_, kern, 256, lds_bytes));
            blocks_per_cu = nb_ < 1 ? 1 : nb_;
        }
        const long ntiles = (n + FWR_TILE - 1) / FWR_TILE;
        const long slots = (long)ctx().num_cus * blocks_per_cu;
        long run = (ntiles + slots - 1) / slots;
        if (II && run < 2 && ntiles > slots / 2) run = 2;          // bound the warm-up share
        FwrParams pr;
        memset(&pr, 0, sizeof(pr));
        pr.hist = (const float *)hist[cur].p; pr.x = x; pr.n = n; pr.taps_rev = (const float *)d_taps.p; pr.y = y;
        pr.hist_out = M > 1 ? (float *)hist[cur ^ 1].p : nullptr;
        pr.run = run;
        pr.b0 = b0; pr.b1 = b1; pr.na1 = na1; pr.nb = nb; pr.ptab = (const float *)d_ptab.p;
        const float *si = (const float *)st[cur].p;
        float *so = (float *)st[cur ^ 1].p;
        pr.state_in = si; pr.vhist = si + 1; pr.state_out = so; pr.vhist_out = so + 1;
        pr.dec = (long)D; pr.dfirst = (long)index; pr.warm_waves = warm_waves;
        const unsigned grid = (unsigned)((ntiles + run - 1) / run);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, ctx().stream, pr);
        LR_LAUNCH_CHECK();
        cur ^= 1;
        return 0;
    }
    template <bool II>
    int dispatch(const float *x, long n, float *y)
    {
        switch (M) {
            case 32: return launch<32, II>(x, n, y);
            case 64: return launch<64, II>(x, n, y);
            case 128: return launch<128, II>(x, n, y);
            default: return set_error("internal: no window-kernel instantiation for %d taps", M);
        }
    }
    long run(const void *in_dev, unsigned long n, void *out_dev, unsigned long cap) override
    {
        if (!n) return 0;
        unsigned long n_out = D == 1 ? n : (n > index ? (n - index + D - 1) / D : 0);
        if (n_out > cap) return set_error("%s: output capacity %lu < %lu", kind(), cap, n_out);
        int rc = iir ? dispatch<true>((const float *)in_dev, (long)n, (float *)out_dev) : dispatch<false>((const float *)in_dev, (long)n, (float *)out_dev);
        if (rc) return rc;
        if (D > 1) index = index + n_out * D - n;       // downsampler.lua:53
        return (long)n_out;
    }
};

// taps in natural order (as FIRFilterBlock takes them); b / a as IIRFilterBlock takes them (nb <= 2, na == 2), or null for the plain filter
static FirWinRealStage *firwin_real_build(const float *taps, int M, const float *b, int nb, const float *a, int na, unsigned long D)
{
    if (!FirWinRealStage::supported_taps(M)) { set_error("firwin: no instantiation for %d taps", M); return nullptr; }
    if (ensure_init()) return nullptr;
    std::unique_ptr<FirWinRealStage> q(new (std::nothrow) FirWinRealStage());
    if (!q) { set_error("out of memory"); return nullptr; }
    q->M = M; q->D = D;
    q->in_size = q->out_size = 4;
    q->taps_rev.resize((size_t)M);
    for (int i = 0; i < M; i++) q->taps_rev[(size_t)i] = taps[M - 1 - i];
    if (upload(q->d_taps, q->taps_rev.data(), (size_t)M * sizeof(float))) return nullptr;
    if (b) {
        if (nb < 1 || nb > 2 || na != 2 || a[0] == 0.f) { set_error("fir+iir: first-order recurrence with at most two feed-forward taps only"); return nullptr; }
        const double a1 = (double)a[1] / (double)a[0];
        q->warm_waves = FirWinRealStage::warm_waves_for(a1);
        if (!q->warm_waves) { set_error("fir+iir: the recurrence's memory is too long for the in-launch warm-up"); return nullptr; }
        q->iir = true;
        q->nb = nb;
        // the coefficients exactly as lrhip_iir_create rounds them (IirCoeffs: b[j]/a0, a[i]/a0 in Float32)
        q->b0 = (float)((double)b[0] / (double)a[0]);
        q->b1 = nb > 1 ? (float)((double)b[1] / (double)a[0]) : 0.f;
        const float a1f = (float)a1;
        q->na1 = -a1f;
        std::vector<float> ptab(64);
        double p16 = 1.0;
        for (int k = 0; k < 16; k++) p16 *= -(double)a1f;
        double acc = 1.0;
        for (int l = 0; l < 64; l++) { acc *= p16; ptab[(size_t)l] = (float)acc; }
        if (upload(q->d_ptab, ptab.data(), ptab.size() * sizeof(float))) return nullptr;
    } else {
        if (D != 1) { set_error("firwin: decimation only behind the fused recurrence"); return nullptr; }
        if (q->d_ptab.reserve(256)) return nullptr;
    }
    if (q->reset()) return nullptr;
    return q.release();
}
