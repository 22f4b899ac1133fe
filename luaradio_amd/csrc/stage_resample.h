// stage_resample.h - polyphase rational resampler and the MFMA channelizer
// (part of liblrhip.so; included by lrhip.hip in this order, one translation unit)
#pragma once

// =====================================================================================================
// polyphase rational resampler (chains: [MultiplyConstant] -> Upsampler -> FIR -> [Downsampler])
// =====================================================================================================
struct ResampleStage : lrhip_stage {
    int S = 2, M = 0, L = 1, HQ = 0;
    unsigned long D = 1;
    float c = 1.f;
    DeviceBuf d_taps, d_ttab, hist[2];
    int cur = 0;
    uint64_t Q0 = 0, m0 = 0;          // absolute input samples consumed / outputs emitted so far
    int interp_J = 0;                 // > 0: the register-window interpolator kernel applies (kernels_interp.h), taps per phase
    int rat_blocks_per_cu = 0;
    // (L, D) pairs with a kept-phases instantiation (fir_rational_kernel): 128 taps, ComplexFloat32, gcd(L, D) = 1
    static bool rational_supported(int L, unsigned long D) { return (L == 3 && D == 2) || (L == 2 && D == 3) || (L == 4 && D == 3) || (L == 3 && D == 4) || (L == 5 && D == 4) || (L == 4 && D == 5); }
    int interp_blocks_per_cu = 0;     // its resident workgroups per CU (occupancy query, cached)
    // Tiles per workgroup of the register-window kernels (kernels_interp.h `rounds`): runs of consecutive tiles, workgroups handed out in address order.
    // 0 = persistent workgroups with a grid stride and a register prefetch of the next tile (rounds 2-3; launches that do not fill the slots keep it).
    // Measured on 2^26 input samples, same box, two alternations (profiles/r04_resampler_tile_order.txt): ONE tile per workgroup is the fastest order for
    // every RationalResampler shape - (3,2) 0.256 against 0.281 ms persistent, (4,3) 0.224 / 0.254, (5,4) 0.207 / 0.234, (3,4) 0.156 / 0.170, (4,5) 0.169 / 0.184,
    // (2,3) 0.166 / 0.182; runs of 2 / 3 / 4 / 16 tiles give back 2 / 4 / 6 / 12 % of it; Interpolator(5) 0.745 against 0.758.
    // LRHIP_RESAMPLE_ROUNDS in the environment forces a run length (A/B; 0 = persistent)
    static int resample_rounds(long ntiles, long slots)
    {
        static const char *e = getenv("LRHIP_RESAMPLE_ROUNDS");
        if (e) return atoi(e) < 0 ? 0 : atoi(e);
        return ntiles <= slots ? 0 : 1;
    }
    static constexpr int SPAN_MAX = 6144;
    const char *kind() const override { return "resample"; }
    unsigned long max_output(unsigned long n) const override { return (n * (unsigned long)L) / D + 2; }
    int seek(unsigned long long n0, unsigned long long *n0_out) override
    {
        if (reset()) return -1;
        Q0 = n0;
        m0 = (n0 * (uint64_t)L + D - 1) / D;
        *n0_out = m0;
        return 0;
    }
    long memory() const override { return HQ; }
    void rate(unsigned long *num, unsigned long *den) const override { *num = D; *den = (unsigned long)L; }
    static bool fits(int M, int L, unsigned long D) { return L >= 1 && 256 * D / (unsigned long)L + (unsigned long)((M - 1) / L) + 4 <= (unsigned long)SPAN_MAX; }
    int reset() override
    {
        cur = 0; Q0 = 0; m0 = 0;
        size_t hb = (size_t)(HQ > 0 ? HQ : 1) * S * sizeof(float);
        return (zero_fill(hist[0], hb) || zero_fill(hist[1], hb)) ? -1 : 0;
    }
    long run(const void *in_dev, unsigned long n, void *out_dev, unsigned long cap) override
    {
        if (!n) return 0;
        // outputs m with m*D inside the upsampled positions [Q0*L, (Q0+n)*L)
        uint64_t hi = (Q0 + n) * (uint64_t)L;
        uint64_t m_end = (hi + D - 1) / D;                     // first m with m*D >= hi
        long n_out = (long)(m_end - m0);
        if ((unsigned long)n_out > cap) return set_error("resample: output capacity %lu < %ld", cap, n_out);
        const float *h = (const float *)hist[cur].p;
        float *ho = (float *)hist[cur ^ 1].p;
        if (interp_J > 0 && D == 1 && ((uintptr_t)in_dev & 7) == 0 && ((uintptr_t)out_dev & 7) == 0) {
            // ComplexFloat32 Interpolator(L), 128 taps: a lane owns 5 input positions and all L phases (fir_interp_kernel)
            auto gi = [&](auto kern, size_t lds_bytes, int tq) -> int {
                if (lds_bytes > 48 * 1024) LR_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                if (!interp_blocks_per_cu) {
                    int nb = 0;
                    LR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, FIP_NT, lds_bytes));
                    interp_blocks_per_cu = nb < 1 ? 1 : nb;
                }
                const long ntiles = ((long)n + tq - 1) / tq, slots = (long)ctx().num_cus * interp_blocks_per_cu;
                const int rounds = resample_rounds(ntiles, slots);
                const long wgs = rounds > 0 ? (ntiles + rounds - 1) / rounds : (ntiles < slots ? ntiles : slots);
                hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(FIP_NT), lds_bytes, ctx().stream, h, (const float *)in_dev, (const float *)d_ttab.p,
                                   (float *)out_dev, (long)n, HQ, c, ho, ablation_bits("LRHIP_INTERP_DBG"), rounds);
                return 0;
            };
#define LR_INTERP(LL, JJ) gi(fir_interp_kernel<LL, JJ>, (size_t)FipGeom<LL, JJ>::LDS_FLOATS * sizeof(float), FipGeom<LL, JJ>::TQ)
            int rc = L == 2 ? LR_INTERP(2, 64) : L == 3 ? LR_INTERP(3, 43) : L == 4 ? LR_INTERP(4, 32) : LR_INTERP(5, 26);
#undef LR_INTERP
            if (rc) return rc;
            LR_LAUNCH_CHECK();
            cur ^= 1;
            Q0 += n;
            m0 = m_end;
            return n_out;
        }
        static const bool no_rational = getenv("LRHIP_NO_RATIONAL_WIN") != nullptr;      // A/B knob: one output per thread (fir_resample_kernel)
        if (!no_rational && interp_J > 0 && D > 1 && ((uintptr_t)in_dev & 7) == 0 && ((uintptr_t)out_dev & 7) == 0) {
            // ComplexFloat32 RationalResampler(L, D), 128 taps: only the kept (position, phase) pairs, register window (fir_rational_kernel)
            auto gr = [&](auto kern, size_t lds_bytes, int tq) -> int {
                if (lds_bytes > 48 * 1024) LR_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                if (!rat_blocks_per_cu) {
                    int nb = 0;
                    LR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds_bytes));
                    rat_blocks_per_cu = nb < 1 ? 1 : nb;
                }
                const long ntiles = ((long)n + (long)(Q0 % D) + tq - 1) / tq, slots = (long)ctx().num_cus * rat_blocks_per_cu;
                const int rounds = resample_rounds(ntiles, slots);
                const long wgs = rounds > 0 ? (ntiles + rounds - 1) / rounds : (ntiles < slots ? ntiles : slots);
                hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(256), lds_bytes, ctx().stream, h, (const float *)in_dev, (const float *)d_ttab.p,
                                   (float *)out_dev, (long)n, n_out, m0, Q0, HQ, c, ho, rounds);
                return 0;
            };
#define LR_RAT(LL, DD, JJ, RR) gr(fir_rational_kernel<LL, DD, JJ, RR>, (size_t)FrrGeom<LL, DD, JJ, RR>::LDS_FLOATS * sizeof(float), FrrGeom<LL, DD, JJ, RR>::TQ)
            // positions per lane (R) by measurement on MI355X, 2^26 input samples.  With the taps read from the LDS the shapes were sharply sensitive to it
            // ((3,4) R = 4 / 8 / 12 -> 0.33 / 0.70 / 0.49 ms, (5,4) R = 4 / 8 -> 0.86 / 0.24): the LDS pipe was the bound.  With the taps in SGPRs
            // (FW_TAPS_SGPR, kernels_firwin.h) a sweep of R = D .. 4 D moves no shape by more than 5 % ((2,3) R = 3 / 6 / 9 / 12 -> 0.242 / 0.190 / 0.180 /
            // 0.187, (3,4) R = 4 / 8 / 12 / 16 -> 0.180 / 0.174 / 0.174 / 0.172, (3,2) 0.289-0.293 for R = 4 .. 10)
            int rc = (L == 3 && D == 2) ? LR_RAT(3, 2, 43, 6) : (L == 2 && D == 3) ? LR_RAT(2, 3, 64, 9) : (L == 4 && D == 3) ? LR_RAT(4, 3, 32, 6)
                   : (L == 3 && D == 4) ? LR_RAT(3, 4, 43, 8) : (L == 5 && D == 4) ? LR_RAT(5, 4, 26, 8) : LR_RAT(4, 5, 32, 10);
#undef LR_RAT
            if (rc) return rc;
            LR_LAUNCH_CHECK();
            cur ^= 1;
            Q0 += n;
            m0 = m_end;
            return n_out;
        }
        // per-workgroup input span: 256 outputs advance 256*D/L input samples, plus the (M-1)/L samples of filter memory
        int span_cap = (int)(256 * D / (unsigned long)L) + (M - 1) / L + 4;
        size_t lds_bytes = ((size_t)((((M - 1) / L + 1) * L + 3) & ~3) + (size_t)span_cap * S) * sizeof(float);
        auto go = [&](auto kern) -> int {
            if (lds_bytes > 48 * 1024) LR_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            unsigned grid = n_out > 0 ? (unsigned)((n_out + 255) / 256) : 1;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, ctx().stream, h, (const float *)in_dev, (const float *)d_taps.p, (float *)out_dev, M, L,
                               (long)D, (long)n, n_out, m0, Q0, HQ, c, span_cap, ho);
            return 0;
        };
        int rc = S == 2 ? go(fir_resample_kernel<2>) : go(fir_resample_kernel<1>);
        if (rc) return rc;
        LR_LAUNCH_CHECK();
        cur ^= 1;
        Q0 += n;
        m0 = m_end;
        return n_out;
    }
};

// =====================================================================================================
// polyphase channelizer as a dense MFMA GEMM
// =====================================================================================================
struct ChannelizerStage : lrhip_stage {
    int seek(unsigned long long, unsigned long long *) override { return set_error("seek: not supported by the channelizer stage"); }
    long memory() const override { return -1; }
    int M = 0, K = 0;
    DeviceBuf W, hist[2];
    int cur = 0;
    unsigned long index = 0;
    const char *kind() const override { return "channelizer"; }
    unsigned long max_output(unsigned long n) const override { return (n / K + 1) * K; }
    int reset() override
    {
        cur = 0; index = 0;
        size_t hb = (size_t)(M - 1) * 2 * sizeof(float);
        return (zero_fill(hist[0], hb) || zero_fill(hist[1], hb)) ? -1 : 0;
    }
    template <int NCT>
    int launch(const float *x, long n, float *y, long nframes)
    {
        constexpr int K2 = 16 * NCT;
        int nflt = 2 * ((CHAN_MT - 1) * K + M);
        size_t dsize = (size_t)((nflt + 2 * (nflt / K2) + 2 + 3) / 4) * 4;
        size_t lds_bytes = (dsize + (size_t)2 * CHAN_KSLAB * (K2 + 16)) * sizeof(float);
        auto kern = channelizer_kernel<NCT>;
        if (lds_bytes > 48 * 1024) LR_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        unsigned grid = (unsigned)((nframes + CHAN_MT - 1) / CHAN_MT);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, ctx().stream, (const float *)hist[cur].p, x, (const float *)W.p, y, M, n,
                           nframes, (long)index);
        LR_LAUNCH_CHECK();
        return 0;
    }
    long run(const void *in_dev, unsigned long n_in, void *out_dev, unsigned long cap) override
    {
        long n = (long)n_in;
        if (n <= 0) return 0;
        long nframes = n_in > index ? (long)((n_in - index + K - 1) / K) : 0;
        if ((unsigned long)(nframes * K) > cap) return set_error("channelizer: output capacity %lu < %ld", cap, nframes * K);
        const float *x = (const float *)in_dev;
        if (nframes > 0) {
            int rc = K == 32 ? launch<4>(x, n, (float *)out_dev, nframes) : launch<8>(x, n, (float *)out_dev, nframes);
            if (rc) return rc;
        }
        unsigned grid = grid_for((unsigned long)(M - 1) * 2, 256);
        hipLaunchKernelGGL(fir_history_kernel<2>, dim3(grid), dim3(256), 0, ctx().stream, (const float *)hist[cur].p, x, (float *)hist[cur ^ 1].p, M, n);
        LR_LAUNCH_CHECK();
        cur ^= 1;
        index = index + (unsigned long)nframes * K - n_in;
        return nframes * K;
    }
};
