// stage_fir.h - FIRFilterBlock stage: direct Toeplitz-MFMA / overlap-save FFT / decimating kernels, fused rotator, downsampler, discriminator; fir_build()
// (part of liblrhip.so; included by lrhip.hip in this order, one translation unit)
#pragma once

// =====================================================================================================
// FIRFilterBlock (+ fused FrequencyTranslatorBlock in front, + fused DownsamplerBlock behind)
// =====================================================================================================
struct FirStage : lrhip_stage {
    int M = 0, S = 2, taps_complex = 0;
    unsigned D = 1;
    bool use_fft = false;
    std::vector<float> taps_rev;          // host copy, reversed (firfilter.lua:234-238)
    DeviceBuf d_taps, d_atab, d_ctaps4;
    int ksteps = 0;                       // 0 => MFMA path unavailable for this (M, D)
    int mfma_blocks_per_cu = 0;           // resident workgroups of the persistent kernel (occupancy query, cached)
    int hist_pad = 0;                     // leading pad floats in the history buffers (1 for complex taps, see launch_mfma_cc)
    DeviceBuf hist[2];
    int cur = 0;
    unsigned long index = 0;              // carried downsampler index (downsampler.lua:53)
    bool rot = false;                     // fused rotator in front
    // rotator + discriminator epilogue on the persistent Toeplitz kernel: window-relative phasors (kernels_fir.h, REL) unless the environment asks
    // for the stand-alone rotator's phasors bit for bit
    bool rel_rot = !LRHIP_DISC_EPI_LDS && getenv("LRHIP_TUNER_EXACT") == nullptr;
    bool rel_nw1 = getenv("LRHIP_TUNER_NW1") != nullptr;      // A/B knob: one-wave workgroups for that kernel (every wave stages its own window, no barriers: measured equal)
    uint64_t rot_step = 0, count = 0;     // absolute index of the next input sample
    // overlap-save emission framing (firfilter.lua:451-485)
    long L = 0, fill = 0;
    DeviceBuf pending, work;
    // overlap-save ARITHMETIC (fused 1024-point FFT kernel); independent of the emission framing
    static constexpr int FFT_PART = 512;   // taps per overlap-save partition (V = 512, L = 512 of the 1024-point block)
    bool fft_arith = false;
    // round 3: IQFileSource's format stage (u8 / s8 / s16le records) in front of a fused Tuner is folded into it: the persistent kernel converts the records
    // on the way into LDS (kernels_fir.h FMT); the stage itself (not owned) converts for every other launch form
    int in_fmt = 0;                       // RX_FMT_*
    lrhip_stage *fmt_stage = nullptr;
    DeviceBuf converted;
    bool raw_now = false;                 // set around core() while x holds raw records
    DeviceBuf d_fft_tables;
    DeviceBuf d_fft4k_tables;             // 513 .. 1281 taps on a ComplexFloat32 stream: the 4096-point kernel (kernels_firfft4k.h)
    DeviceBuf d_fft64_tables;             // ... and its one-wave-per-block form (kernels_firfft64.h)
    int fft64_np = 0;                     // round 5: 1 282 .. 2 049 taps (1) / 2 050 .. 4 097 taps (2 partitions) on that form at an overlap of 2 048
    int fft4k_V = 0, fft4k_blocks = 0;    // its overlap (768 / 1024 / 1280; 0 = not built)
    int fft_blocks_per_cu = 0;
    // decimating polyphase-FFT form (kernels_firdecfft.h): ComplexFloat32 stream, D >= 2, ceil(M / D) <= 32
    bool decfft = false;
    int mode_req = 0;                     // use_fft as the caller passed it (0..3), before 3 = automatic was resolved
    DeviceBuf d_dec_tables;
    int dec_blocks_per_cu = 0;
    double rot_omega = 0.0;
    static bool decfft_supported(unsigned d, int m, int s) { return s == 2 && (d == 2 || d == 4 || d == 5 || d == 8) && (m + (int)d - 1) / (int)d <= DF_V && m >= 8; }
    // fused FrequencyDiscriminatorBlock in front (chains): input is ComplexFloat32, the filter runs on arg(c[i] conj c[i-1])/gain
    bool hist_in_kernel = false;          // set by a launch that also wrote the next history buffer
    bool pre_disc = false;
    // fused FrequencyDiscriminatorBlock behind the filter (chains): ComplexFloat32 in, Float32 out (persistent MFMA kernel epilogue)
    bool post_disc = false;
    int post_unary = 0;      // 1 + UN_CMAG / UN_CPHASE / UN_CREAL / UN_CIMAG folded into the LDS-staged decimator's store (chains: tuner -> ComplexMagnitude ...), Float32 out
    bool can_post_unary() const { return S == 2 && D > 1 && !ksteps && decim_lds_ok() && !decfft && !pre_disc && !post_disc; }
    DeviceBuf edge;
    // fix-up of the wave-first discriminator outputs (disc_epilogue): done by fir_disc_fixup_kernel, or - defer_fixup - left to the next
    // stage of the chain, a pair-mode window filter that patches the samples as it stages them (FwcParams::fix_edge): one launch less
    bool defer_fixup = false, fix_ready = false;
    long fix_nunits = 0; int fix_unit = 0;  // the deferred fix-up as fir_disc_fixup_kernel would have been launched (consumer chunks that emit nothing)
    int fix_shift = 8;                     // log2 of the outputs per edge record pair
    const float2 *fix_prev_ptr = nullptr;
    FirStage *fix_src = nullptr;           // consumer side: the stage whose edge records are to be applied
    double disc_gain = 1.0;
    DeviceBuf disc_prev;
    int disc_cur = 0;

    const char *kind() const override { return "fir"; }
    unsigned long max_output(unsigned long n) const override
    {
        if (use_fft) return (unsigned long)(((fill + (long)n) / L) * L);
        return D == 1 ? n : n / D + 1;
    }
    int reset() override
    {
        cur = 0; index = 0; count = 0; fill = 0; disc_cur = 0; iir_cur = 0;
        if (iir_fused && (zero_fill(iir_state[0], 4 * sizeof(float)) || zero_fill(iir_state[1], 4 * sizeof(float)))) return -1;
        if ((pre_disc || post_disc) && zero_fill(disc_prev, 4 * sizeof(float))) return -1;
        size_t hb = ((size_t)(M > 1 ? M - 1 : 1) * S + hist_pad) * sizeof(float);
        if (zero_fill(hist[0], hb) || zero_fill(hist[1], hb)) return -1;
        return 0;
    }

    int seek(unsigned long long n0, unsigned long long *n0_out) override
    {
        if (use_fft && n0 % (unsigned long long)L) return set_error("fir(fft framing): a partition must start on a block boundary (multiple of %ld samples)", L);
        if (reset()) return -1;
        count = n0;                                                     // fused rotator: phase = step * absolute index
        index = (unsigned long)((D - n0 % D) % D);                      // fused downsampler: kept outputs are the absolute indices 0 mod D
        *n0_out = (n0 + D - 1) / D;
        return 0;
    }
    long memory() const override
    {
        long m = M - 1 + (pre_disc ? 1 : 0);
        if (post_disc) m += D;                                          // one filter output earlier
        if (iir_fused) m += (long)D * 320 * iir_warm;                   // the low-rate recurrence's in-launch warm-up length, in outputs
        return m;
    }
    void rate(unsigned long *num, unsigned long *den) const override { *num = D; *den = 1; }
    unsigned long align() const override
    {
        if (iir_fused) return 2UL * 256 * 5 * D;                        // pair-mode tile: 2 x 256 lanes x 5 outputs
        if (fft_arith) {
            // overlap-save arithmetic: the 1024-point blocks advance by Lf samples from the start of a chunk (two blocks ride together on
            // a Float32 stream); the same grid gives the same rounding
            unsigned long l = 1;
            const int nparts = (M + FFT_PART - 1) / FFT_PART;
            for (int part = 0; part < nparts; part++) {
                const int Mp = part + 1 < nparts ? FFT_PART : M - part * FFT_PART;
                unsigned long a = (unsigned long)(FFTN - ((Mp - 1 + 63) / 64) * 64) * (S == 1 ? 2UL : 1UL), x = l, y = a;
                while (y) { unsigned long t = x % y; x = y; y = t; }
                l = l / x * a;
            }
            return l;
        }
        if (rot && post_disc && !decfft && !win_cplx_ok() && rel_rot) {
            // tuner + discriminator on the persistent Toeplitz kernel: a tile's window is rotated relative to its first sample
            // (kernels_fir.h, REL), so the rounding follows the tile grid, which starts with the chunk
            const int nacc5 = getenv("LRHIP_FIR_D5_NACC") ? atoi(getenv("LRHIP_FIR_D5_NACC")) : 2;
            if (D == 1) return (unsigned long)FirMfmaGeom<2, 1>::tile_out(LRHIP_FIR_D1_NACC);
            if (D == 5) return 5UL * FirMfmaGeom<2, 5>::tile_out(nacc5 == 1 ? 1 : 2, ksteps == 51 && rel_nw1 ? 1 : 4);
        }
        return 1UL;
    }

    // MFMA steps of the 128-tap ComplexFloat32 Toeplitz filter at decimation d (fir_mfma_ksteps(128, d, 2)): the shapes with a discriminator epilogue
    static constexpr int disc_ksteps(int d) { return (1 + 15 * d + 128 + 3) / 4; }
    template <int SS, int DD, int NACC>
    int launch_mfma(const float *x, long n, float *y, long n_out)
    {
        // the shapes that matter most get the persistent, fully unrolled instantiation:
        // M = 128 at D = 1 (36 MFMA steps, the headline) and M = 128 at D = 5 (60 steps, the WBFM tuner)
        if constexpr (DD == 1) {
            if (ksteps == 36) return launch_mfma_ks<SS, DD, NACC, 36>(x, n, y, n_out);     // M = 128, cf32
            if (ksteps == 37) return launch_mfma_ks<SS, DD, NACC, 37>(x, n, y, n_out);     // M = 128, f32 (slack up to 3 samples)
        }
        if constexpr (DD == 5) {
            if (ksteps == 51) return launch_mfma_ks<SS, DD, NACC, 51>(x, n, y, n_out);     // M = 128 at D = 5 (Tuner / Decimator(5))
        }
        // Round 5: the Tuner of an FM receiver at OTHER input rates - decimation 4, 8, 10 at 128 taps - with the discriminator epilogue (only that
        // combination: the plain Tuner / Decimator at these decimations keep the generic kernel)
        if constexpr (SS == 2 && (DD == 4 || DD == 8 || DD == 10)) {
            constexpr int KSD = disc_ksteps(DD);
            if (post_disc && rot && ksteps == KSD) return launch_mfma_ks<SS, DD, NACC, KSD>(x, n, y, n_out);
        }
        return launch_mfma_ks<SS, DD, NACC, 0>(x, n, y, n_out);
    }

    template <typename K>
    int prepare_kernel(K kern, size_t lds_bytes, int *blocks_per_cu, int threads = 256)
    {
        if (lds_bytes > 48 * 1024) LR_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        if (blocks_per_cu) {
            int nb = 0;
            LR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, lds_bytes));
            *blocks_per_cu = nb < 1 ? 1 : nb;
        }
        return 0;
    }

    // a stage launches several instantiations over its life (raw records / converted samples on edge chunks, the bit-exact and the window-relative
    // rotator): each gets its own attribute call and its own occupancy figure, queried once.  Returns workgroups per CU, or -1.
    std::vector<std::pair<const void *, int>> prepared;
    template <typename K>
    int prepared_blocks(K kern, size_t lds_bytes, int threads = 256)
    {
        const void *kp = (const void *)kern;
        for (const auto &e : prepared)
            if (e.first == kp) return e.second;
        int nb = 0;
        if (prepare_kernel(kern, lds_bytes, &nb, threads)) return -1;
        prepared.emplace_back(kp, nb);
        return nb;
    }

    template <int SS, int DD, int NACC, int KS, int NW = 4>
    int launch_mfma_ks(const float *x, long n, float *y, long n_out)
    {
        using G = FirMfmaGeom<SS, DD>;
        constexpr int TILE_OUT = G::tile_out(NACC, NW);
        if constexpr (NW == 4 && SS == 2 && DD == 5 && KS == 51) {
            // tuner + discriminator with window-relative phasors: one-wave workgroups (every wave stages its own window, no barriers)
            if (rot && post_disc && rel_rot && rel_nw1) return launch_mfma_ks<SS, DD, NACC, KS, 1>(x, n, y, n_out);
        }
        // alignment slack so that the tile's first staged sample is 16-B aligned in global memory (raw records: the 4- / 8-byte word of two samples)
        const unsigned esz = raw_now ? (in_fmt == RX_FMT_S16LE ? 4u : 2u) : (unsigned)(4 * SS);
        if (((uintptr_t)x % esz) != 0) {
            if (rot || post_disc) return set_error("fir: fused rotator / discriminator needs a sample-aligned input pointer");
            return launch_direct(x, n, y, n_out);
        }
        long sample_addr = (long)((uintptr_t)x / esz);
        int q = 4 / SS;
        long v = sample_addr + (long)index - (M - 1);
        int e = (int)(((v % q) + q) % q);
        int span = G::span(NACC, ksteps, NW);
        size_t lds_floats = (size_t)fir_taps_len(DD, ksteps) + (size_t)G::phys(SS * span) + G::PAD + 8;
        size_t lds_bytes = lds_floats * sizeof(float);
        long ntiles = (n_out + TILE_OUT - 1) / TILE_OUT;
        const float *atab = (const float *)d_atab.p;          // zero-padded reversed taps
        const float *h = (const float *)hist[cur].p + hist_pad;
        int out_aligned = ((uintptr_t)y % 16) == 0;
        uint64_t rs = rot ? rot_step : 0, rc = rot ? count : 0;
        if constexpr (KS > 0) {
            auto launch = [&](auto kern) -> int {
                if ((mfma_blocks_per_cu = prepared_blocks(kern, lds_bytes, 64 * NW)) < 0) return -1;     // queried once per instantiation
                long slots = (long)ctx().num_cus * mfma_blocks_per_cu;
                // tile order: persistent grid stride (0), or runs of `rounds` consecutive tiles per workgroup in address order (LRHIP_FIR_ROUNDS, A/B)
                static const int rounds_env = getenv("LRHIP_FIR_ROUNDS") ? atoi(getenv("LRHIP_FIR_ROUNDS")) : 0;
                const int rounds = ntiles > slots && rounds_env > 0 ? rounds_env : 0;
                unsigned grid = rounds > 0 ? (unsigned)((ntiles + rounds - 1) / rounds) : (unsigned)(ntiles < slots ? ntiles : slots);
                if (post_disc && edge.reserve((size_t)ntiles * 2 * NW * sizeof(float2))) return -1;
                float *ho = M > 1 ? (float *)hist[cur ^ 1].p + hist_pad : nullptr;
                hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds_bytes, ctx().stream, h, x, atab, y, M, n, n_out, (long)index, e,
                                   ntiles, out_aligned, rs, rc, (float2 *)edge.p, post_disc ? (float2 *)disc_prev.p + (disc_cur ^ 1) : nullptr, 1.0 / disc_gain, ho, rounds);
                hist_in_kernel = ho != nullptr;
                return 0;
            };
            int rc2;
            if constexpr (SS == 2 && (DD == 4 || DD == 8 || DD == 10)) {
                // only the Tuner + discriminator form is instantiated at these decimations (launch_mfma)
                if (!(post_disc && rot)) return set_error("internal: decimation %d has a persistent kernel for tuner + discriminator only", DD);
            }
            if constexpr (SS == 2 && (DD == 1 || DD == 5 || DD == 4 || DD == 8 || DD == 10)) {
                if (post_disc) {
                    if constexpr (DD == 4 || DD == 8 || DD == 10)
                        rc2 = rel_rot ? launch(fir_mfma_persistent_kernel<2, DD, NACC, true, KS, 1, true>) : launch(fir_mfma_persistent_kernel<2, DD, NACC, true, KS, 1, false>);
                    else if constexpr (NW == 1) rc2 = launch(fir_mfma_persistent_kernel<2, DD, NACC, true, KS, 1, true, 1>);
                    else if constexpr (LRHIP_DISC_EPI_LDS) rc2 = rot ? launch(fir_mfma_persistent_kernel<2, DD, NACC, true, KS, 1>) : launch(fir_mfma_persistent_kernel<2, DD, NACC, false, KS, 1>);
                    else rc2 = !rot ? launch(fir_mfma_persistent_kernel<2, DD, NACC, false, KS, 1>)
                             : rel_rot ? launch(fir_mfma_persistent_kernel<2, DD, NACC, true, KS, 1, true>) : launch(fir_mfma_persistent_kernel<2, DD, NACC, true, KS, 1, false>);
                    if (rc2) return rc2;
                    LR_LAUNCH_CHECK();
                    float2 *dp = (float2 *)disc_prev.p;
                    // records per unit of FIX_UNIT outputs: a wave's range in the in-register epilogue, a whole tile in the LDS one
                    constexpr int FIX_UNIT = LRHIP_DISC_EPI_LDS ? TILE_OUT : TILE_OUT / NW;
                    const long nunits = LRHIP_DISC_EPI_LDS ? ntiles : NW * ntiles;
                    // (the consumer patches at most 32 samples per half window: a record pair per >= 256 outputs)
                    if (defer_fixup && (FIX_UNIT & (FIX_UNIT - 1)) == 0 && FIX_UNIT >= 256) {
                        fix_ready = true;
                        fix_nunits = nunits; fix_unit = FIX_UNIT;
                        fix_prev_ptr = (const float2 *)(dp + disc_cur);
                        fix_shift = 0;
                        while ((1 << fix_shift) < FIX_UNIT) fix_shift++;
                    } else {
                        hipLaunchKernelGGL(fir_disc_fixup_kernel, dim3((unsigned)((nunits + 255) / 256)), dim3(256), 0, ctx().stream, (const float2 *)edge.p,
                                           nunits, FIX_UNIT, y, n_out, (const float2 *)(dp + disc_cur), 1.0 / disc_gain);
                        LR_LAUNCH_CHECK();
                    }
                    disc_cur ^= 1;
                    return 0;
                }
            }
            if (post_disc) return set_error("internal: discriminator epilogue without a persistent kernel variant");
            if constexpr (SS == 2 && DD == 5 && KS == 51 && NW == 4) {
                if (raw_now) {
                    if (rot)
                        rc2 = in_fmt == RX_FMT_U8 ? launch(fir_mfma_persistent_kernel<2, DD, NACC, true, KS, 0, false, 4, RX_FMT_U8>)
                            : in_fmt == RX_FMT_S8 ? launch(fir_mfma_persistent_kernel<2, DD, NACC, true, KS, 0, false, 4, RX_FMT_S8>)
                                                  : launch(fir_mfma_persistent_kernel<2, DD, NACC, true, KS, 0, false, 4, RX_FMT_S16LE>);
                    else
                        rc2 = in_fmt == RX_FMT_U8 ? launch(fir_mfma_persistent_kernel<2, DD, NACC, false, KS, 0, false, 4, RX_FMT_U8>)
                            : in_fmt == RX_FMT_S8 ? launch(fir_mfma_persistent_kernel<2, DD, NACC, false, KS, 0, false, 4, RX_FMT_S8>)
                                                  : launch(fir_mfma_persistent_kernel<2, DD, NACC, false, KS, 0, false, 4, RX_FMT_S16LE>);
                    if (rc2) return rc2;
                    LR_LAUNCH_CHECK();
                    return 0;
                }
            }
            if (raw_now) return set_error("internal: raw records reached a kernel without a record instantiation");
            if constexpr (DD == 4 || DD == 8 || DD == 10) return set_error("internal: no plain persistent kernel at decimation %d", DD);
            else {
                if constexpr (SS == 2) rc2 = rot ? launch(fir_mfma_persistent_kernel<2, DD, NACC, true, KS>) : launch(fir_mfma_persistent_kernel<2, DD, NACC, false, KS>);
                else rc2 = rot ? set_error("rotator fusion needs complex input") : launch(fir_mfma_persistent_kernel<1, DD, NACC, false, KS>);
                if (rc2) return rc2;
            }
        } else {
            if (post_disc) return set_error("internal: discriminator epilogue without a persistent kernel variant");
            auto launch = [&](auto kern) -> int {
                if (prepare_kernel(kern, lds_bytes, nullptr)) return -1;
                hipLaunchKernelGGL(kern, dim3((unsigned)ntiles), dim3(256), lds_bytes, ctx().stream, h, x, atab, y, M, n, n_out, (long)index, e,
                                   ksteps, out_aligned, rs, rc);
                return 0;
            };
            int rc2;
            if constexpr (SS == 2) rc2 = rot ? launch(fir_mfma_kernel<2, DD, NACC, true, 1>) : launch(fir_mfma_kernel<2, DD, NACC, false, 1>);
            else rc2 = rot ? set_error("rotator fusion needs complex input") : launch(fir_mfma_kernel<1, DD, NACC, false, 1>);
            if (rc2) return rc2;
        }
        LR_LAUNCH_CHECK();
        return 0;
    }

    // HilbertTransformBlock in one launch: the generic Float32 Toeplitz kernel with the pair epilogue (kernels_fir.h, HILB).  y2 receives n
    // ComplexFloat32 samples (delayed input, filtered input); history / index bookkeeping is core()'s (D = 1: index stays 0)
    bool hilbert_ok() const { return S == 1 && !taps_complex && D == 1 && ksteps > 0 && !rot && !fft_arith && !use_fft && !pre_disc && !post_disc; }
    // the window form (kernels_firwin.h hilbert_win_kernel): the reference's tap counts, taps at even distance from the centre exactly zero
    int hilb_sparse = -1, hilb_blocks = 0;
    template <int MM>
    int launch_hilbert_win(const float *x, long n, float *y2)
    {
        using G = FwhGeom<MM>;
        const size_t lds_bytes = (size_t)G::LDS_FLOATS * sizeof(float);
        auto kern = hilbert_win_kernel<MM>;
        if (!hilb_blocks) {
            if (lds_bytes > 48 * 1024) LR_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            int nb_ = 0;
            LR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_, kern, 256, lds_bytes));
            hilb_blocks = nb_ < 1 ? 1 : nb_;
        }
        static const int run_knob = getenv("LRHIP_HILBERT_RUN") ? atoi(getenv("LRHIP_HILBERT_RUN")) : 0;      // A/B knob: tiles per workgroup
        const long ntiles = (n + FWR_TILE - 1) / FWR_TILE, slots = (long)ctx().num_cus * hilb_blocks;
        long run = (ntiles + 4 * slots - 1) / (4 * slots);
        if (run_knob > 0) run = run_knob;
        const unsigned grid = (unsigned)((ntiles + run - 1) / run);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, ctx().stream, (const float *)hist[cur].p + hist_pad, x, (const float *)d_taps.p, y2, n, run,
                           (float *)hist[cur ^ 1].p + hist_pad);
        LR_LAUNCH_CHECK();
        cur ^= 1;
        count += (uint64_t)n;
        return 0;
    }
    int launch_hilbert(const float *x, long n, float *y2)
    {
        if (hilb_sparse < 0) {
            hilb_sparse = (M == 65 || M == 129) ? 1 : 0;
            for (int j = 0; j < M && hilb_sparse; j += 2)
                if (taps_rev[(size_t)j] != 0.0f) hilb_sparse = 0;
        }
        static const bool no_win = getenv("LRHIP_HILBERT_MFMA") != nullptr;      // A/B knob: the matrix-core pair epilogue of round 3
        if (hilb_sparse && !no_win && index == 0) return M == 65 ? launch_hilbert_win<65>(x, n, y2) : launch_hilbert_win<129>(x, n, y2);
        constexpr int NACC = 4;
        using G = FirMfmaGeom<1, 1>;
        constexpr int TILE_OUT = G::tile_out(NACC);
        if (((uintptr_t)x % 4) != 0) return set_error("hilbert: unaligned input pointer");
        const long v = (long)((uintptr_t)x / 4) + (long)index - (M - 1);
        const int e = (int)(((v % 4) + 4) % 4);
        const int span = G::span(NACC, ksteps);
        const size_t lds_bytes = ((size_t)fir_taps_len(1, ksteps) + (size_t)G::phys(span) + G::PAD + 8) * sizeof(float);
        const long ntiles = (n + TILE_OUT - 1) / TILE_OUT;
        auto kern = fir_mfma_kernel<1, 1, NACC, false, 1, true>;
        if (prepare_kernel(kern, lds_bytes, nullptr)) return -1;
        hipLaunchKernelGGL(kern, dim3((unsigned)ntiles), dim3(256), lds_bytes, ctx().stream, (const float *)hist[cur].p + hist_pad, x, (const float *)d_atab.p, y2, M, n, n,
                           (long)index, e, ksteps, (int)(((uintptr_t)y2 % 16) == 0), (uint64_t)0, (uint64_t)0);
        LR_LAUNCH_CHECK();
        // history carry, as core() does for kernels that do not write it themselves
        if (M > 1) {
            unsigned grid = grid_for((unsigned long)(M - 1), 256);
            hipLaunchKernelGGL(fir_history_kernel<1>, dim3(grid), dim3(256), 0, ctx().stream, (const float *)hist[cur].p + hist_pad, x, (float *)hist[cur ^ 1].p + hist_pad, M, n);
            LR_LAUNCH_CHECK();
            cur ^= 1;
        }
        count += (uint64_t)n;
        return 0;
    }

    // complex taps: two real Toeplitz filters (re / im) of 2M taps over the interleaved float stream, decimation 2D,
    // sharing every B fragment.  Stream position of output k in float units is 2*q_k + 1 once the float stream is
    // given one leading pad float (so the history is the 2M-1 floats the S = 1 kernel expects).
    template <int DD2, int NACC>
    int launch_mfma_cc(const float *x, long n, float *y, long n_out)
    {
        using G = FirMfmaGeom<1, DD2>;
        constexpr int TILE_OUT = G::tile_out(NACC);
        if (((uintptr_t)x % 8) != 0) return launch_direct(x, n, y, n_out);
        const int M2 = 2 * M;
        const long first2 = 2 * (long)index + 1, n2 = 2 * n;
        long v = (long)((uintptr_t)x / 4) + first2 - (M2 - 1);
        int e = (int)(((v % 4) + 4) % 4);
        int span = G::span(NACC, ksteps);
        size_t lds_bytes = ((size_t)2 * fir_taps_len(DD2, ksteps) + (size_t)G::phys(span) + G::PAD + 8) * sizeof(float);
        long ntiles = (n_out + TILE_OUT - 1) / TILE_OUT;
        const float *atab = (const float *)d_atab.p;          // [re taps | im taps], each zero-padded
        const float *h = (const float *)hist[cur].p;          // includes the pad float
        int out_aligned = ((uintptr_t)y % 16) == 0;
        auto kern = fir_mfma_kernel<1, DD2, NACC, false, 2>;
        if (prepare_kernel(kern, lds_bytes, nullptr)) return -1;
        hipLaunchKernelGGL(kern, dim3((unsigned)ntiles), dim3(256), lds_bytes, ctx().stream, h, x, atab, y, M2, n2, n_out, first2, e,
                           ksteps, out_aligned, (uint64_t)0, (uint64_t)0);
        LR_LAUNCH_CHECK();
        return 0;
    }

    int dispatch_mfma_cc(const float *x, long n, float *y, long n_out)
    {
        switch (D) {
            case 1: return launch_mfma_cc<2, 4>(x, n, y, n_out);
            case 2: return launch_mfma_cc<4, 2>(x, n, y, n_out);
            case 3: return launch_mfma_cc<6, 1>(x, n, y, n_out);
            case 4: return launch_mfma_cc<8, 1>(x, n, y, n_out);
            case 5: return launch_mfma_cc<10, 1>(x, n, y, n_out);
            default: return decim_lds_ok() ? launch_decim_lds(x, n, y, n_out) : launch_direct(x, n, y, n_out);
        }
    }

    template <int VV, int NG>
    int launch_fft4k_ng(const float *x, long n, float *y, long n_out, int *blocks_per_cu)
    {
        constexpr long Lf = F4K_N - VV;
        const size_t lds_bytes = (size_t)f4k_lds_elems(NG) * sizeof(float2);
        auto kern = fir_fft4k_kernel<VV, NG>;
        if (!*blocks_per_cu && prepare_kernel(kern, lds_bytes, blocks_per_cu, 256 * NG)) return -1;
        // XCD-major block order (kernels_firfft4k.h), measured on 2^26 samples, same box: 1 276 taps 0.479 -> 0.448 ms (the 31 % overlap becomes L2 hits),
        // 768 taps equal; LRHIP_F4K_XCD_MAP=0 is the plain order
        static const int xcd_map = getenv("LRHIP_F4K_XCD_MAP") ? atoi(getenv("LRHIP_F4K_XCD_MAP")) : 1;
        const long nblocks = (n_out + Lf - 1) / Lf, nslots = (nblocks + NG - 1) / NG, slots = (long)ctx().num_cus * *blocks_per_cu;
        const unsigned grid = (unsigned)(nslots < slots ? nslots : slots);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * NG), lds_bytes, ctx().stream, (const float *)hist[cur].p + hist_pad, x, (const float2 *)d_fft4k_tables.p, y, M, n,
                           n_out, nblocks, M > 1 ? (float *)hist[cur ^ 1].p + hist_pad : (float *)nullptr, xcd_map);
        LR_LAUNCH_CHECK();
        hist_in_kernel = true;
        return 0;
    }
    int fft4k_blocks2 = 0;
    // one WAVE per 4096-point block as 64 x 64 (kernels_firfft64.h): eight waves per CU with the conjugate-symmetric H of real taps, four with complex taps
    template <int VV, int WAVES, int SS = 2, bool HG = false>
    int launch_fft64(const float *x, long n, float *y, long n_out)
    {
        constexpr long Lf = F4K_N - VV;
        const size_t lds_bytes = (size_t)f64_lds_elems(WAVES) * sizeof(float2);
        auto kern = fir_fft64_kernel<VV, WAVES, 1, SS, HG>;
        if (prepared_blocks(kern, lds_bytes, 64 * WAVES) < 0) return -1;
        static const int xcd_map = getenv("LRHIP_F4K_XCD_MAP") ? atoi(getenv("LRHIP_F4K_XCD_MAP")) : 1;
        // (Float32 stream: two stream blocks per transform - the kernel's block count is the number of transforms)
        const long nblocks = ((n_out + Lf - 1) / Lf + (2 - SS)) / (3 - SS), nslots = (nblocks + WAVES - 1) / WAVES;
        const unsigned grid = (unsigned)(nslots < ctx().num_cus ? nslots : ctx().num_cus);      // 108 / 158 KB of LDS: one workgroup per CU
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), lds_bytes, ctx().stream, (const float *)hist[cur].p + hist_pad, x, (const float2 *)d_fft64_tables.p, y, M, n,
                           n_out, nblocks, M > 1 ? (float *)hist[cur ^ 1].p + hist_pad : (float *)nullptr, xcd_map, 0L, 0);
        LR_LAUNCH_CHECK();
        hist_in_kernel = true;
        return 0;
    }
    // round 5: V = 2 048, one partition (eight waves on real taps, four on complex ones) or two (four waves, a run of consecutive blocks per wave)
    template <int NP, int SS = 2>
    int launch_fft64_long(const float *x, long n, float *y, long n_out)
    {
        constexpr int VV = 2048, WAVES = 4;
        constexpr long Lf = F4K_N - VV;
        const size_t lds_bytes = (size_t)f64_lds_elems(WAVES, NP) * sizeof(float2);
        auto kern = fir_fft64_kernel<VV, WAVES, NP, SS>;
        if (prepared_blocks(kern, lds_bytes, 64 * WAVES) < 0) return -1;
        const long nblocks = (n_out + Lf - 1) / Lf, nslots = (nblocks + WAVES - 1) / WAVES;
        const unsigned grid = (unsigned)(nslots < ctx().num_cus ? nslots : ctx().num_cus);
        // round 6: 4 098 .. 8 193 taps (fft64_np = 3, 4) = a second launch with partitions 2 and 3 on the stream delayed by 4 096 samples, adding to y
        const int launches = NP == 2 && fft64_np > 2 ? 2 : 1;
        for (int l = 0; l < launches; l++) {
            hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), lds_bytes, ctx().stream, (const float *)hist[cur].p + hist_pad, x,
                               (const float2 *)d_fft64_tables.p + (size_t)l * F64_TABLE_ELEMS2, y, M, n, n_out, nblocks,
                               (M > 1 && l == 0) ? (float *)hist[cur ^ 1].p + hist_pad : (float *)nullptr, 0, (long)l * F4K_N, l);
            LR_LAUNCH_CHECK();
        }
        hist_in_kernel = true;
        return 0;
    }
    template <int VV>
    int launch_fft64_v(const float *x, long n, float *y, long n_out)
    {
        if (S == 1) return launch_fft64<VV, 8, 1>(x, n, y, n_out);
        // round 6, measured and left OFF: complex taps at eight waves per CU with H read from the global table (LRHIP_F64_HG=1) are 8-10 % SLOWER than four waves
        // with H in the LDS (1 276 taps, 2^26 samples, three alternations on one box: 0.447 / 0.444 / 0.437 against 0.404 / 0.406 / 0.405 ms, profiles/r06_ab_hg.txt) -
        // 64 more global loads per block in a kernel whose block is already a third memory-instruction issue
        static const int hg_knob = getenv("LRHIP_F64_HG") ? atoi(getenv("LRHIP_F64_HG")) : 0;
        if (taps_complex) return hg_knob ? launch_fft64<VV, 8, 2, true>(x, n, y, n_out) : launch_fft64<VV, 4>(x, n, y, n_out);
        return launch_fft64<VV, 8>(x, n, y, n_out);
    }
    template <int VV>
    int launch_fft4k(const float *x, long n, float *y, long n_out)
    {
        // A/B knob: blocks per workgroup.  2 (512 threads, shared tables, 16 waves per CU instead of 12) measured SLOWER: 0.505 against 0.479 ms - the
        // barriers then couple eight waves; the kernel is bound by its five workgroup barriers per block, not by occupancy (counters: VALU 27 %, LDS 42 % busy)
        static const int ng = getenv("LRHIP_F4K_NG") ? atoi(getenv("LRHIP_F4K_NG")) : 1;
        return ng == 2 ? launch_fft4k_ng<VV, 2>(x, n, y, n_out, &fft4k_blocks2) : launch_fft4k_ng<VV, 1>(x, n, y, n_out, &fft4k_blocks);
    }
    // ---- partitioned overlap-save (kernels_firpols.h, round 4): 513 taps and more in one launch per 1 536 taps, ComplexFloat32 or Float32 stream
    int pols_blocks = 0;
    template <int SS, int PP>
    int launch_pols_p(const float *x, long n, float *y, long n_out, int part0)
    {
        const size_t lds_bytes = (size_t)pols_lds_elems(SS, PP) * sizeof(float2);
        constexpr int POLS_WPB = pols_wpb(SS, PP);
        auto kern = fir_pols_kernel<SS, PP>;
        if (prepared_blocks(kern, lds_bytes, 64 * POLS_WPB) < 0) return -1;
        // a wave owns a run of consecutive blocks (its spectra delay line); runs of ~40 blocks keep the P - 1 warm-up blocks of a run under 5 %,
        // and the number of runs is a whole number of rounds of (CUs x waves) where the launch is long enough
        constexpr long RPW = SS == 2 ? 1 : 2;
        const long nblocks = (n_out + POLS_HOP - 1) / POLS_HOP, per_round = (long)ctx().num_cus * POLS_WPB * RPW;
        static const long run_env = getenv("LRHIP_POLS_RUN") ? atol(getenv("LRHIP_POLS_RUN")) : 0;      // A/B knob
        long rounds = (nblocks + per_round * 20) / (per_round * 40);
        if (rounds < 1) rounds = 1;
        long run = run_env > 0 ? run_env : (nblocks + per_round * rounds - 1) / (per_round * rounds);
        if (run < 4 * (PP - 1) + 4) run = 4 * (PP - 1) + 4;
        const long nruns = (nblocks + run - 1) / run, nslots = (nruns + POLS_WPB * RPW - 1) / (POLS_WPB * RPW);
        const unsigned grid = (unsigned)(nslots < ctx().num_cus ? nslots : ctx().num_cus);
        float *ho = (M > 1 && part0 == 0) ? (float *)hist[cur ^ 1].p + hist_pad : nullptr;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * POLS_WPB), lds_bytes, ctx().stream, (const float *)hist[cur].p + hist_pad, x, (const float2 *)d_fft_tables.p, y, M, n,
                           n_out, nblocks, run, part0, part0 > 0 ? 1 : 0, ho);
        if (ho) hist_in_kernel = true;
        LR_LAUNCH_CHECK();
        return 0;
    }
    template <int SS>
    int launch_pols(const float *x, long n, float *y, long n_out)
    {
        const int nparts = (M + FFT_PART - 1) / FFT_PART;
        hist_in_kernel = false;
        // partitions per launch: up to 3 at twelve waves per CU (a delay line of two spectra in registers, 168 registers per lane).  Round 5: a ComplexFloat32 filter of
        // four partitions and more (1 537 taps up) takes FOUR per launch at eight waves per CU (three spectra, 256 registers): 4 096 taps in two launches instead
        // of three - same box, three alternations (profiles/r05_ab_pols_p4.txt): 1.4846 / 1.4822 / 1.4842 -> 1.1281 / 1.1296 / 1.1158 ms on 2^26 samples.
        // LRHIP_POLS_P=3 is the round-4 split (A/B knob).
        static const int pmax_env = getenv("LRHIP_POLS_P") ? atoi(getenv("LRHIP_POLS_P")) : 4;
        const int PMAX = (pmax_env == 4 && SS == 2 && nparts >= 4) ? 4 : 3;
        for (int p0 = 0; p0 < nparts; p0 += PMAX) {
            const int P = nparts - p0 < PMAX ? nparts - p0 : PMAX;
            int rc;
            if constexpr (SS == 2) rc = P == 4 ? launch_pols_p<SS, 4>(x, n, y, n_out, p0) : P == 3 ? launch_pols_p<SS, 3>(x, n, y, n_out, p0) : P == 2 ? launch_pols_p<SS, 2>(x, n, y, n_out, p0) : launch_pols_p<SS, 1>(x, n, y, n_out, p0);
            else rc = P == 3 ? launch_pols_p<SS, 3>(x, n, y, n_out, p0) : P == 2 ? launch_pols_p<SS, 2>(x, n, y, n_out, p0) : launch_pols_p<SS, 1>(x, n, y, n_out, p0);
            if (rc) return rc;
        }
        return 0;
    }

    int launch_fft(const float *x, long n, float *y, long n_out)
    {
        static const bool no_4k = getenv("LRHIP_FFT_NO_4K") != nullptr;      // A/B knob: partitions of the 1024-point kernel (round 2)
        // more than 512 taps: the partitioned form (one launch per 1 536 taps) wherever the 4096-point kernels do not apply - Float32 streams, more than
        // 1 281 taps - instead of one accumulating pass of the 1024-point kernel per 512 taps.  LRHIP_FFT_POLS=1 / 0 forces it on (also for 513 .. 1 281
        // taps on a ComplexFloat32 stream) / off (A/B)
        static const int pols_knob = getenv("LRHIP_FFT_POLS") ? atoi(getenv("LRHIP_FFT_POLS")) : -1;
        // round 5: 1 282 .. 4 097 taps on a ComplexFloat32 stream as ONE launch of the 64 x 64 kernel at an overlap of 2 048 (two partitions above 2 049 taps) once
        // a wave's run is long enough to pay for its warm-up block; LRHIP_F64_LONG=0 keeps the partitioned 1024-point kernel (A/B)
        static const int long_knob = getenv("LRHIP_F64_LONG") ? atoi(getenv("LRHIP_F64_LONG")) : 1;
        // round 6: Float32 streams (real taps) ride the same kernels, two stream blocks per transform: LRHIP_F64_F32=0 keeps the partitioned kernel for them (A/B)
        static const int f32_knob = getenv("LRHIP_F64_F32") ? atoi(getenv("LRHIP_F64_F32")) : 1;
        if (fft64_np && long_knob && pols_knob != 1 && !pre_disc && !post_disc && (S == 2 || f32_knob)) {
            const long nb = (n_out + 2047) / 2048;
            // (size sweep 2^20 .. 2^26 samples, same box: faster than the partitioned kernel at every size - 4 096 taps 0.072 / 0.106 / 0.196 / 0.575 ms against
            // 0.188 / 0.208 / 0.243 / 1.104 at 2^20 / 2^22 / 2^24 / 2^26, 2 048 taps 0.048 against 0.093 at 2^22 - so there is no lower bound; LRHIP_F64_LONG_MIN = blocks per CU)
            static const long long_min = getenv("LRHIP_F64_LONG_MIN") ? atol(getenv("LRHIP_F64_LONG_MIN")) : 0;
            if (nb >= long_min * ctx().num_cus) {
                if (S == 1) return fft64_np == 1 ? launch_fft64<2048, 8, 1>(x, n, y, n_out) : launch_fft64_long<2, 1>(x, n, y, n_out);      // (np = 2, 3, 4: two partitions per launch)
                if (fft64_np == 1) return launch_fft64_v<2048>(x, n, y, n_out);
                return launch_fft64_long<2>(x, n, y, n_out);
            }
        }
        // one wave per 4096-point block (fir_fft64_kernel, one 512- / 256-thread workgroup per CU) once the launch has enough blocks per CU (below); smaller
        // launches keep the workgroup-per-block form, which spreads over more CUs.  LRHIP_F4K_WAVE=1 / 0 forces one or the other (A/B)
        static const int wave_knob = getenv("LRHIP_F4K_WAVE") ? atoi(getenv("LRHIP_F4K_WAVE")) : -1;
        const long nblocks4k = fft4k_V ? (n_out + (F4K_N - fft4k_V) - 1) / (F4K_N - fft4k_V) : 0;
        // Float32 streams (round 6, size sweep 2^18 .. 2^26 on one box, profiles/r06_f32_long_filter_sizes.txt): the wave-per-block kernel beats the partitioned
        // one at EVERY size (1 276 taps: 0.034-0.051 against 0.064-0.071 ms up to 2^23 samples - the partitioned kernel has a 40-65 us floor) except where the
        // launch is a little more than one round of the chip's 8 x CUs waves and the filter short (768 taps at 2^24: 2 521 transforms = 1.23 rounds, 0.070 against
        // 0.052 ms): only that window keeps the partitioned kernel
        const long transforms = (nblocks4k + 1) / 2, one_round = 8L * ctx().num_cus;
        const bool f32_window = fft4k_V == 768 && transforms > one_round && 20 * transforms <= 27 * one_round;
        // ComplexFloat32 streams: the workgroup-per-block kernel up to 20 blocks per CU (real taps; 32 with complex taps, whose wave-per-block form runs four waves per
        // CU) - re-measured in round 6 on the same sweep: at 2^23 samples (2 521-2 979 blocks, the old bound of 8 per CU already on the wave kernel) it is 15-40 %
        // faster (1 276 taps 0.057 against 0.067 ms, 768 taps 0.048 / 0.068, complex taps 0.057 / 0.081), at 2^24 the two cross (0.108 / 0.097, 0.091 / 0.095, 0.108 / 0.119)
        const bool wave4k = wave_knob >= 0 ? wave_knob != 0 : S == 1 ? !f32_window : nblocks4k >= (taps_complex ? 32L : 20L) * ctx().num_cus;
        // (a Float32 stream has no workgroup-per-block kernel: where the wave-per-block kernel is not taken it stays partitioned)
        const bool f32_part = S == 1 && (!f32_knob || !wave4k);
        if (M > FFT_PART && !pre_disc && !post_disc && pols_knob != 0 && (pols_knob == 1 || !fft4k_V || no_4k || f32_part))
            return S == 2 ? launch_pols<2>(x, n, y, n_out) : launch_pols<1>(x, n, y, n_out);
        if (fft4k_V && !no_4k && !pre_disc && !post_disc && wave4k) {
            switch (fft4k_V) {
                case 768: return launch_fft64_v<768>(x, n, y, n_out);
                case 1024: return launch_fft64_v<1024>(x, n, y, n_out);
                default: return launch_fft64_v<1280>(x, n, y, n_out);
            }
        }
        if (fft4k_V && !no_4k && !pre_disc && !post_disc && S == 2) {      // (ComplexFloat32 only; a Float32 stream that gets here - LRHIP_FFT_POLS=0 - takes the per-partition passes below)
            switch (fft4k_V) {
                case 768: return launch_fft4k<768>(x, n, y, n_out);
                case 1024: return launch_fft4k<1024>(x, n, y, n_out);
                default: return launch_fft4k<1280>(x, n, y, n_out);
            }
        }
        size_t lds_bytes = (size_t)FFT_LDS_ELEMS * sizeof(float2);
        static const long lds_pad = getenv("LRHIP_FFT_LDS_PAD") ? atol(getenv("LRHIP_FFT_LDS_PAD")) : 0;      // A/B knob: unused LDS per workgroup -> fewer resident workgroups per CU
        lds_bytes += (size_t)lds_pad;
        const float *h = (const float *)hist[cur].p + hist_pad;
        hist_in_kernel = false;
        // one launch per partition of at most FFT_PART taps (a plain filter has one); partitions after the first accumulate
        const int nparts = (M + FFT_PART - 1) / FFT_PART;
        for (int part = 0; part < nparts; part++) {
            const int Mp = part + 1 < nparts ? FFT_PART : M - part * FFT_PART;
            const long Lf = FFTN - ((Mp - 1 + 63) / 64) * 64;      // block advance of the fused kernel (overlap rounded to 64)
            long nblocks = (n_out + Lf - 1) / Lf;
            long nffts = S == 2 ? nblocks : (nblocks + 1) / 2;
            const float2 *tables = (const float2 *)d_fft_tables.p + (size_t)part * FFT_TABLE_ELEMS;
            auto go = [&](auto kern) -> int {
                if (!fft_blocks_per_cu && prepare_kernel(kern, lds_bytes, &fft_blocks_per_cu, 64 * FFT_WPB)) return -1;
                long slots = (long)ctx().num_cus * fft_blocks_per_cu;
                long want = (nffts + FFT_WPB - 1) / FFT_WPB;
                static const int rounds_env = getenv("LRHIP_FFT_ROUNDS") ? atoi(getenv("LRHIP_FFT_ROUNDS")) : -1;      // A/B knob, read once
                // one-shot order with 8 batches per workgroup once the launch exceeds the resident slots: 316 GS/s against 263-314
                // (run-to-run spread) for the persistent stride on 2^28 samples, same box, alternating
                // (only when the launch is many times the resident slots: a 2^26-sample chain's 1/5-rate audio filter, 1 873 workgroups
                // on 768 slots, keeps the persistent walk - 8 batches per workgroup would leave two thirds of the CUs idle)
                // Round 3, re-measured after the early first-block loads and the wave-uniform addressing (two boxes, alternating, HIP events): the persistent walk is
                // now the faster order at every size - 2^26 samples 0.2155 against 0.2368 ms, 2^27 0.4199 / 0.4357, 2^28 0.8359 / 0.8412 and 0.8591 / 0.8646; the
                // Float32 and complex-taps filters at 2^26 gain 9 % - a launch of 8-batch workgroups ends with a ragged last round that the one-block stride
                // does not have.  The one-shot order stays as LRHIP_FFT_ROUNDS=8.
                int rounds = rounds_env >= 0 ? rounds_env : 0;
                if (want <= slots) rounds = 0;
                unsigned grid = rounds > 0 ? (unsigned)((want + rounds - 1) / rounds) : (unsigned)(want < slots ? want : slots);
                // input / output in HOST memory (host_execute's direct mode, chain.h): a short persistent grid, so that the reads of one block and the writes of
                // another share the link instead of every wave loading, then every wave storing
                if (host_io_grid() > 0 && rounds == 0 && grid > (unsigned)host_io_grid()) grid = (unsigned)host_io_grid();
                // tapered tail of the one-shot order (kernels_firfft.h): the last three "waves" of workgroups own rounds/2, rounds/4, rounds/8 batches
                // (measured equal on 2^28 samples, same box: 0.865-0.878 ms with, 0.860-0.867 without - the ~45 us fixed cost the size sweep shows is not
                // the tail of long workgroups; opt-in, LRHIP_FFT_TAPER=1)
                static const bool use_taper = getenv("LRHIP_FFT_TAPER") != nullptr && atoi(getenv("LRHIP_FFT_TAPER")) > 0;      // A/B knob
                int n_full = 0, taper = 0;
                if (rounds >= 2 && use_taper) {
                    const long r1 = rounds / 2, r2 = rounds / 4 > 0 ? rounds / 4 : 1, r3 = rounds / 8 > 0 ? rounds / 8 : 1;
                    const long tail_b = slots * (r1 + r2 + r3);
                    if (want > 2 * tail_b) {
                        taper = (int)slots;
                        n_full = (int)((want - tail_b) / rounds);
                        long rem = want - (long)n_full * rounds;                         // >= tail_b
                        long k1 = slots, k2 = slots;
                        rem -= k1 * r1 + k2 * r2;
                        long k3 = rem > 0 ? (rem + r3 - 1) / r3 : 0;
                        grid = (unsigned)(n_full + k1 + k2 + k3);
                    }
                }
                const float2 *dp = pre_disc ? (const float2 *)disc_prev.p + disc_cur : nullptr;
                float *ho = (!pre_disc && M > 1 && part == 0) ? (float *)hist[cur ^ 1].p + hist_pad : nullptr;
#ifdef LRHIP_FFT_TRACE
                static unsigned long long *trace = nullptr;
                static long trace_launches = 0;
                const size_t trace_n = (size_t)8 * FFT_WPB * 32 * 12;
                if (!trace) {
                    LR_HIP(hipMalloc(&trace, trace_n * 8));
                    LR_HIP(hipMemcpyToSymbol(HIP_SYMBOL(lrhip_fft_trace), &trace, sizeof(trace)));
                }
                if (++trace_launches == 12) LR_HIP(hipMemsetAsync(trace, 0, trace_n * 8, ctx().stream));
#endif
                hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * FFT_WPB), lds_bytes, ctx().stream, h, x, tables, y, Mp, n, n_out, nblocks,
                                   1.0 / disc_gain, dp, ho, M, (long)part * FFT_PART, part > 0 ? 1 : 0, rounds, n_full, taper);
#ifdef LRHIP_FFT_TRACE
                if (trace_launches == 12) {
                    // stamps: 0 loop top, 1 loads issued (the first dft16 waits for them), 2 stage 1 done, 3 E1 exchanged, 4 stage 2 + inner transpose done,
                    // 5 stage 3 / H / inverse stage 3 + inner transpose done, 6 inverse stage 2 done, 7 E1 back, 8 inverse stage 1 done, 9 stores issued
                    LR_HIP(hipStreamSynchronize(ctx().stream));
                    std::vector<unsigned long long> tr(trace_n);
                    LR_HIP(hipMemcpy(tr.data(), trace, trace_n * 8, hipMemcpyDeviceToHost));
                    static const char *names[9] = {"issue", "load+st1", "E1", "st2+T", "st3 H st3 T", "ist2", "E1back", "ist1", "store"};
                    double sum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, blk = 0;
                    int cnt = 0;
                    for (int w = 0; w < 8 * FFT_WPB; w++)
                        for (int t = 1; t < 30; t++) {
                            const unsigned long long *s = &tr[((size_t)w * 32 + t) * 12], *nx = s + 12;
                            if (!s[0] || !s[9] || !nx[0]) continue;
                            for (int i = 0; i < 9; i++) sum[i] += (double)(s[i + 1] - s[i]);
                            blk += (double)(nx[0] - s[0]);
                            cnt++;
                        }
                    if (cnt) {
                        fprintf(stderr, "fft trace (%d blocks):", cnt);
                        for (int i = 0; i < 9; i++) fprintf(stderr, "  %s %.0f", names[i], sum[i] / cnt);
                        fprintf(stderr, "  | block %.0f clocks\n", blk / cnt);
                    }
                }
#endif
                if (ho) hist_in_kernel = true;
                return 0;
            };
            int rc = S == 2 ? go(fir_fft_kernel<2, 0>) : pre_disc ? go(fir_fft_kernel<1, 1>) : go(fir_fft_kernel<1, 0>);
            if (rc) return rc;
            LR_LAUNCH_CHECK();
        }
        return 0;
    }

    // decimations without a Toeplitz instantiation (and taps too long for its LDS table): LDS-staged one-output-per-thread kernel
    bool decim_lds_ok() const { return !fft_arith && !use_fft && M + 255 <= DECIM_SPAN_MAX && !(taps_complex && rot); }
    int decim_blocks_per_cu = 0;
    // round 5: the second form (kernels_firdecim.h) for a ComplexFloat32 stream and real taps; LRHIP_DECIM_V1=1 keeps the first
    bool decim_lds2_ok() const
    {
        static const bool v1_env = getenv("LRHIP_DECIM_V1") != nullptr;
        return !v1_env && S == 2 && !taps_complex && D >= 2 && M + 255 <= DECIM2_SPAN_MAX;
    }
    int launch_decim_lds(const float *x, long n, float *y, long n_out)
    {
        const bool v2 = decim_lds2_ok();
        // staged samples per tile: the kernel's registers allow DECIM_SPAN_MAX; LRHIP_DECIM_SPAN (A/B) asks for less = smaller tiles, more workgroups per CU
        static const long span_env = getenv("LRHIP_DECIM_SPAN") ? atol(getenv("LRHIP_DECIM_SPAN")) : 0;
        const long span_cap = v2 ? DECIM2_SPAN_MAX : DECIM_SPAN_MAX;
        const long span_max = span_env >= 512 && span_env < span_cap && span_env >= M + 64 ? span_env : span_cap;
        const int dsc = post_disc ? 1 : 0;       // discriminator epilogue (second form only): OW counts the stored outputs, every tile computes one more in front
        if (dsc && !v2) return set_error("internal: discriminator epilogue on the first LDS-staged decimator form");
        long ow = (span_max - M) / (long)D + 1 - dsc;
        int OW = (int)(ow > 256 - 4 * dsc ? 256 - 4 * dsc : ow < 1 ? 1 : ow);
        long ntiles = (n_out + OW - 1) / OW;
        long span = (long)(OW - 1 + dsc) * D + M;
        size_t lds_bytes = v2 ? ((size_t)((M + 3) & ~3) + (size_t)2 * decim2_slots((int)span, (long)D)) * sizeof(float)
                              : ((size_t)(((taps_complex ? 2 : 1) * M + 3) & ~3) + (size_t)S * (span + (span >> 5) + 2)) * sizeof(float);
        const float *h = (const float *)hist[cur].p + hist_pad;
        float *ho = M > 1 ? (float *)hist[cur ^ 1].p + hist_pad : nullptr;
        auto go = [&](auto kern) -> int {
            if ((decim_blocks_per_cu = prepared_blocks(kern, lds_bytes)) < 0) return -1;
            long slots = (long)ctx().num_cus * decim_blocks_per_cu;
            static const int rounds_env = getenv("LRHIP_DECIM_ROUNDS") ? atoi(getenv("LRHIP_DECIM_ROUNDS")) : 0;     // A/B: runs of consecutive tiles, address order
            const int rounds = ntiles > slots && rounds_env > 0 ? rounds_env : 0;
            unsigned grid = rounds > 0 ? (unsigned)((ntiles + rounds - 1) / rounds) : (unsigned)(ntiles < slots ? ntiles : slots);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, ctx().stream, h, x, (const float *)d_taps.p, y, M, n, n_out, (long)index, (long)D, OW,
                               ntiles, rot ? rot_step : (uint64_t)0, rot ? count : (uint64_t)0, ho, post_unary, rounds);
            hist_in_kernel = ho != nullptr;
            return 0;
        };
        auto go2 = [&](auto kern) -> int {
            if ((decim_blocks_per_cu = prepared_blocks(kern, lds_bytes)) < 0) return -1;
            long slots = (long)ctx().num_cus * decim_blocks_per_cu;
            static const int rounds_env = getenv("LRHIP_DECIM_ROUNDS") ? atoi(getenv("LRHIP_DECIM_ROUNDS")) : 0;
            const int rounds = ntiles > slots && rounds_env > 0 ? rounds_env : 0;
            unsigned grid = rounds > 0 ? (unsigned)((ntiles + rounds - 1) / rounds) : (unsigned)(ntiles < slots ? ntiles : slots);
            float2 *dp = (float2 *)disc_prev.p;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, ctx().stream, h, x, (const float *)d_taps.p, y, M, n, n_out, (long)index, (long)D, OW,
                               ntiles, rot ? rot_step : (uint64_t)0, rot ? count : (uint64_t)0, ho, post_unary, rounds, 1.0 / disc_gain,
                               dsc ? (const float2 *)(dp + disc_cur) : (const float2 *)nullptr, dsc ? dp + (disc_cur ^ 1) : (float2 *)nullptr,
                               rel_rot ? 0 : 1);      // an exact chain keeps one fmaf chain per output (no tap split over the half-waves)
            hist_in_kernel = ho != nullptr;
            if (dsc) disc_cur ^= 1;
            return 0;
        };
        int rc;
        if (v2) {
            auto pick = [&](auto ph) -> int {
                constexpr bool PH = decltype(ph)::value;
                if (raw_now)
                    return rot ? (in_fmt == RX_FMT_U8 ? go2(fir_decim_lds2_kernel<true, RX_FMT_U8, PH>) : in_fmt == RX_FMT_S8 ? go2(fir_decim_lds2_kernel<true, RX_FMT_S8, PH>)
                                                                                                        : go2(fir_decim_lds2_kernel<true, RX_FMT_S16LE, PH>))
                               : (in_fmt == RX_FMT_U8 ? go2(fir_decim_lds2_kernel<false, RX_FMT_U8, PH>) : in_fmt == RX_FMT_S8 ? go2(fir_decim_lds2_kernel<false, RX_FMT_S8, PH>)
                                                                                                         : go2(fir_decim_lds2_kernel<false, RX_FMT_S16LE, PH>));
                return rot ? go2(fir_decim_lds2_kernel<true, RX_FMT_CF32, PH>) : go2(fir_decim_lds2_kernel<false, RX_FMT_CF32, PH>);
            };
            rc = decim2_esh((long)D) ? pick(std::true_type{}) : pick(std::false_type{});
        } else if (raw_now) {
            if (taps_complex || S != 2) return set_error("internal: raw records reached a kernel without a record instantiation");
            rc = rot ? (in_fmt == RX_FMT_U8 ? go(fir_decim_lds_kernel<2, true, false, RX_FMT_U8>) : in_fmt == RX_FMT_S8 ? go(fir_decim_lds_kernel<2, true, false, RX_FMT_S8>)
                                                                                               : go(fir_decim_lds_kernel<2, true, false, RX_FMT_S16LE>))
                     : (in_fmt == RX_FMT_U8 ? go(fir_decim_lds_kernel<2, false, false, RX_FMT_U8>) : in_fmt == RX_FMT_S8 ? go(fir_decim_lds_kernel<2, false, false, RX_FMT_S8>)
                                                                                               : go(fir_decim_lds_kernel<2, false, false, RX_FMT_S16LE>));
        } else
        rc = taps_complex ? go(fir_decim_lds_kernel<2, false, true>)
                 : S == 2 ? (rot ? go(fir_decim_lds_kernel<2, true>) : go(fir_decim_lds_kernel<2, false>))
                        : (rot ? set_error("rotator fusion needs complex input") : go(fir_decim_lds_kernel<1, false>));
        if (rc) return rc;
        LR_LAUNCH_CHECK();
        return 0;
    }

    template <int DD>
    int launch_decfft_d(const float *x, long n, float *y, long n_out)
    {
        const size_t lds_bytes = (size_t)df_lds_elems(DD) * sizeof(float2);
        DfParams pr;
        pr.M = M; pr.n = n; pr.n_out = n_out; pr.first = (long)index;
        pr.nblocks = (n_out + DF_LO - 1) / DF_LO;
        pr.rot_step_fx = rot ? rot_step : 0; pr.rot_count0 = rot ? count : 0;
        const double wD = rot ? rot_omega * (double)DD : 0.0;
        pr.cD = make_float2((float)std::cos(wD), (float)std::sin(wD));
        pr.inv_gain = 1.0 / disc_gain;
        pr.taps_rev = (const float *)d_taps.p; pr.taps_complex = taps_complex;
        pr.dbg = ablation_bits("LRHIP_DECFFT_DBG");      // ablation knob (tools/ab_decfft.py): 0 unless the library was built with -DLRHIP_ABLATION
        const float *h = (const float *)hist[cur].p + hist_pad;
        float *ho = M > 1 ? (float *)hist[cur ^ 1].p + hist_pad : nullptr;
        auto go = [&](auto kern) -> int {
            if (!dec_blocks_per_cu && prepare_kernel(kern, lds_bytes, &dec_blocks_per_cu)) return -1;
            const long nquads = (pr.nblocks + 3) / 4, slots = (long)ctx().num_cus * dec_blocks_per_cu;
            static const int rounds_env = getenv("LRHIP_DECFFT_ROUNDS") ? atoi(getenv("LRHIP_DECFFT_ROUNDS")) : 0;      // A/B knob, read once
            // one-shot order (common.h grid_for): workgroups of 4 waves x `rounds` consecutive quads, handed out in address order.  More
            // quads per wave amortise the table load and the first (unhidden) window request; same-box A/B at 2^26 samples:
            // rounds 1 / 2 / 4 = 0.171 / 0.167 / 0.164 ms
            int rounds = rounds_env > 0 ? rounds_env : (nquads >= 16 * slots ? 4 : nquads >= 8 * slots ? 2 : 1);
            pr.rounds = rounds;
            const long wgs = (nquads + 4L * rounds - 1) / (4L * rounds);
            float2 *dp = (float2 *)disc_prev.p;
            hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(256), lds_bytes, ctx().stream, h, x, (const float2 *)d_dec_tables.p, y, pr,
                               post_disc ? (const float2 *)(dp + disc_cur) : nullptr, post_disc ? dp + (disc_cur ^ 1) : nullptr, ho);
            hist_in_kernel = ho != nullptr;
            return 0;
        };
        int rc = post_disc ? go(fir_decfft_kernel<DD, 1>) : go(fir_decfft_kernel<DD, 0>);
        if (rc) return rc;
        LR_LAUNCH_CHECK();
        if (post_disc) disc_cur ^= 1;
        return 0;
    }
    int launch_decfft(const float *x, long n, float *y, long n_out)
    {
        switch (D) {
            case 2: return launch_decfft_d<2>(x, n, y, n_out);
            case 4: return launch_decfft_d<4>(x, n, y, n_out);
            case 5: return launch_decfft_d<5>(x, n, y, n_out);
            default: return launch_decfft_d<8>(x, n, y, n_out);
        }
    }

    int launch_direct(const float *x, long n, float *y, long n_out)
    {
        if (rot) return set_error("internal: direct FIR kernel has no fused rotator");
        unsigned grid = grid_for((unsigned long)n_out, 256);
        const float *h = (const float *)hist[cur].p + hist_pad, *t = (const float *)d_taps.p;
        if (S == 1)
            hipLaunchKernelGGL(fir_direct_kernel<0>, dim3(grid), dim3(256), 0, ctx().stream, h, x, t, y, M, n, n_out, (long)index, (long)D);
        else if (!taps_complex)
            hipLaunchKernelGGL(fir_direct_kernel<1>, dim3(grid), dim3(256), 0, ctx().stream, h, x, t, y, M, n, n_out, (long)index, (long)D);
        else
            hipLaunchKernelGGL(fir_direct_kernel<2>, dim3(grid), dim3(256), 0, ctx().stream, h, x, t, y, M, n, n_out, (long)index, (long)D);
        LR_LAUNCH_CHECK();
        return 0;
    }

    // Float32 stream at D = 1 on the register-window kernel (kernels_firwin.h): every issued packed FMA is useful work,
    // against 89 % for the Toeplitz product
    int win_blocks_per_cu = 0;
    bool win_real_ok() const
    {
        // opt-in (LRHIP_FIR_WIN_REAL=1): 128 taps on 2^26 Float32 samples run at 0.25 ms here against 0.19 ms on the Toeplitz-MFMA kernel
        // (same box) - at D = 1 the Toeplitz product wastes only 11 % of its MACs and keeps more waves resident
        static const bool on = getenv("LRHIP_FIR_WIN_REAL") != nullptr && getenv("LRHIP_NO_FIR_WIN") == nullptr;
        return on && S == 1 && !taps_complex && D == 1 && !rot && !pre_disc && !post_disc && !fft_arith && (M == 32 || M == 64 || M == 128);
    }
    template <int MM, bool ONESHOT = false>
    int launch_win_real_m(const float *x, long n, float *y)
    {
        using G = FwrGeom<MM>;
        const size_t lds_bytes = (size_t)G::LDS_FLOATS * sizeof(float);
        auto kern = fir_win_real_kernel<MM, false>;
        if (!win_blocks_per_cu && prepare_kernel(kern, lds_bytes, &win_blocks_per_cu)) return -1;
        const long ntiles = (n + FWR_TILE - 1) / FWR_TILE;
        const long slots = (long)ctx().num_cus * win_blocks_per_cu;
        FwrParams pr;
        memset(&pr, 0, sizeof(pr));
        pr.hist = (const float *)hist[cur].p + hist_pad; pr.x = x; pr.n = n; pr.taps_rev = (const float *)d_taps.p; pr.y = y;
        pr.hist_out = M > 1 ? (float *)hist[cur ^ 1].p + hist_pad : nullptr;
        pr.run = ONESHOT ? 1 : (ntiles + slots - 1) / slots;
        pr.dec = 1;
        hipLaunchKernelGGL(kern, dim3((unsigned)((ntiles + pr.run - 1) / pr.run)), dim3(256), lds_bytes, ctx().stream, pr);
        LR_LAUNCH_CHECK();
        hist_in_kernel = pr.hist_out != nullptr;
        return 0;
    }
    int launch_win_real(const float *x, long n, float *y)
    {
        return M == 32 ? launch_win_real_m<32>(x, n, y) : M == 64 ? launch_win_real_m<64>(x, n, y) : launch_win_real_m<128>(x, n, y);
    }

    // ComplexFloat32 stream with decimation (Decimator / Tuner [+ discriminator]) and the decimating Float32 filter with a fused
    // first-order recurrence, on the register-window kernel (kernels_firwin2.h)
    int winc_blocks_per_cu = 0;
    bool iir_fused = false;               // pair mode: y[k] = iir_b0 v[k] + iir_na1 y[k-1] behind the filter
    float iir_b0 = 1.f, iir_na1 = 0.f, iir_na1_lo = 0.f;
    int iir_warm = 1;
    DeviceBuf d_iir_ptab, iir_state[2];
    int iir_cur = 0;
    static bool win_off()
    {
        static const bool off = getenv("LRHIP_NO_FIR_WIN") != nullptr;      // A/B knob
        return off;
    }
    // the ComplexFloat32 form is opt-in (LRHIP_FIR_WIN_CPLX=1): same-box A/B on the WBFM tuner + discriminator, 2^26 samples: 0.204 ms against
    // 0.150 ms + 0.005 ms (fix-up) for the Toeplitz-MFMA kernel - both are bound by the shared MFMA / VALU datapath (rocprofv3: 1 250 VALU
    // instructions per wave and 6 360-sample tile, 640 of them the filter), and the Toeplitz kernel keeps 3 workgroups per CU resident against 2
    static bool win_cplx_on()
    {
        static const bool on = getenv("LRHIP_FIR_WIN_CPLX") != nullptr;
        return on;
    }
    bool win_cplx_ok() const { return win_cplx_on() && !win_off() && S == 2 && !taps_complex && D == 5 && M == 128 && !fft_arith && !use_fft && !decfft && !pre_disc; }
    bool win_pair_ok() const { return !win_off() && S == 1 && !taps_complex && D == 5 && M == 136 && !fft_arith && !use_fft && !rot && !pre_disc && !post_disc; }
    // first-order recurrence behind the pair-mode filter: needs |a1|^(320 w) < 1e-12 for the in-launch warm-up (w waves of 64 lanes x 5 outputs)
    // The pole q (a double: p^D of the polyphase identity) is carried as a Float32 pair hi + lo: rounded to one Float32 its relative error of
    // 2^-24 moves the DC gain by 2^-24 q / (1 - q), which for a slow filter is far above the 1e-6 parity bar of the recurrence it replaces.
    int fuse_iir1(double b0, double q)
    {
        if (!win_pair_ok()) return -1;
        const double a1 = -q, p = std::fabs(q);
        int w = 0;
        for (int c = 1; c <= 4 && !w; c *= 2)
            if (p < 1.0 && std::pow(p, 320.0 * c) < 1e-12) w = c;
        if (!w) return -1;
        std::vector<float> ptab(64);
        double pR = 1.0, acc = 1.0;
        for (int k = 0; k < 5; k++) pR *= -(double)a1;
        for (int l = 0; l < 64; l++) { acc *= pR; ptab[(size_t)l] = (float)acc; }
        if (upload(d_iir_ptab, ptab.data(), ptab.size() * sizeof(float))) return -1;
        iir_fused = true; iir_b0 = (float)b0; iir_na1 = (float)q; iir_na1_lo = (float)(q - (double)iir_na1); iir_warm = w;
        return reset();
    }
    // short ComplexFloat32-taps filters at D = 1 (the reference suite's 16-complex-taps entry): 2 M packed FMAs per output on the window kernel,
    // a streaming problem like the real-taps case (the two-Toeplitz-filter form pays 2 x 2 M taps in fixed 16-output blocks)
    bool win_short_c_ok() const
    {
        static const bool off = getenv("LRHIP_NO_FIR_WIN_SHORT") != nullptr;      // A/B knob
        return !off && !win_off() && S == 2 && taps_complex && D == 1 && (M == 16 || M == 32) && d_ctaps4.p && !rot && !fft_arith && !use_fft && !pre_disc && !post_disc;
    }
    int launch_win_short_c(const float *x, long n, float *y, long n_out)
    {
        return M == 16 ? launch_win_cplx_m<16, FWC_CTAPS, 1, true>(x, n, y, n_out) : launch_win_cplx_m<32, FWC_CTAPS, 1, true>(x, n, y, n_out);
    }
    template <int MM, int MODE, int DD = 5, bool ONESHOT = false>
    int launch_win_cplx_m(const float *x, long n, float *y, long n_out)
    {
        using G = FwcGeom<DD, 5, MM, MODE>;
        const size_t lds_bytes = (size_t)G::LDS_FLOATS * sizeof(float);
        auto kern = fir_win_cplx_kernel<DD, 5, MM, MODE>;
        if (!winc_blocks_per_cu && prepare_kernel(kern, lds_bytes, &winc_blocks_per_cu)) return -1;
        FwcParams pr;
        memset(&pr, 0, sizeof(pr));
        pr.hist = (const float *)hist[cur].p + hist_pad; pr.x = x; pr.n = n; pr.taps_rev = (const float *)(G::CTAPS ? d_ctaps4.p : d_taps.p); pr.y = y;
        pr.n_out = n_out; pr.first = (long)index;
        pr.hist_out = M > 1 ? (float *)hist[cur ^ 1].p + hist_pad : nullptr;
        pr.ntiles = G::PAIR ? (n_out + 2L * G::TO - 1) / (2L * G::TO) : (n_out + G::TA - 1) / G::TA;
        pr.rot_step_fx = rot ? rot_step : 0; pr.rot_count0 = rot ? count : 0;
        if (G::DISC) {
            float2 *dp = (float2 *)disc_prev.p;
            pr.prev_in = dp + disc_cur; pr.prev_out = dp + (disc_cur ^ 1);
            pr.inv_gain = 1.0 / disc_gain;
        }
        const long slots = (long)ctx().num_cus * winc_blocks_per_cu;
        unsigned grid;
        if (G::IIR) {
            pr.b0 = iir_b0; pr.na1 = iir_na1; pr.na1_lo = iir_na1_lo; pr.ptab = (const float *)d_iir_ptab.p; pr.warm_waves = iir_warm;
            pr.state_in = (const float *)iir_state[iir_cur].p; pr.state_out = (float *)iir_state[iir_cur ^ 1].p;
            pr.fix_edge = pr.fix_prev = (const float2 *)d_iir_ptab.p;          // readable dummies (64 floats): the kernel loads unconditionally
            pr.fix_inv_gain = 1.0;
            if (fix_src && fix_src->fix_ready) {
                pr.fix_edge = (const float2 *)fix_src->edge.p; pr.fix_prev = fix_src->fix_prev_ptr; pr.fix_inv_gain = 1.0 / fix_src->disc_gain;
                pr.fix_on = 1;
                pr.fix_shift = fix_src->fix_shift;
                fix_src->fix_ready = false;
            }
            pr.run = (pr.ntiles + slots - 1) / slots;
            static const long tail_run_env = getenv("LRHIP_TAIL_RUN") ? atol(getenv("LRHIP_TAIL_RUN")) : 0;      // A/B knob, read once
            if (tail_run_env > 0) pr.run = tail_run_env;
            grid = (unsigned)((pr.ntiles + pr.run - 1) / pr.run);
        } else {
            pr.warm_waves = 4; pr.run = 1;
            // ONESHOT: a workgroup per tile, handed out in address order (short filters are a streaming problem: common.h grid_for)
            static const bool oneshot_env = getenv("LRHIP_FIR_WIN_ONESHOT") != nullptr;      // A/B knob
            grid = (unsigned)((ONESHOT || oneshot_env || pr.ntiles < slots) ? pr.ntiles : slots);
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, ctx().stream, pr);
        LR_LAUNCH_CHECK();
        hist_in_kernel = pr.hist_out != nullptr;
        if (G::DISC) disc_cur ^= 1;
        if (G::IIR) iir_cur ^= 1;
        return 0;
    }
    int launch_win_cplx(const float *x, long n, float *y, long n_out)
    {
        if (post_disc) return rot ? launch_win_cplx_m<128, FWC_ROT | FWC_DISC>(x, n, y, n_out) : launch_win_cplx_m<128, FWC_DISC>(x, n, y, n_out);
        return rot ? launch_win_cplx_m<128, FWC_ROT>(x, n, y, n_out) : launch_win_cplx_m<128, 0>(x, n, y, n_out);
    }
    // short filters on the ComplexFloat32 stream at D = 1 (the reference suite's 16-tap entries): at 16 taps the filter is 16 packed FMAs per
    // output - nothing against its 16 B of traffic - and the Toeplitz product pays its fixed 16-output blocks (K = 15 + 16, half of it zeros).
    // One-shot window kernel, same box, 2^26 samples: 16 taps 0.175 against 0.222 ms (6.1 TB/s = the copy yardstick), 32 taps 0.233 / 0.260,
    // 64 taps 0.301 / 0.326 (there the overlap-save kernel, 0.221, is what `automatic` picks)
    bool win_short_ok() const
    {
        static const bool off = getenv("LRHIP_NO_FIR_WIN_SHORT") != nullptr;      // A/B knob
        return !off && !win_off() && S == 2 && !taps_complex && D == 1 && (M == 16 || M == 32 || M == 64) && !rot && !fft_arith && !use_fft && !pre_disc && !post_disc;
    }
    int launch_win_short(const float *x, long n, float *y, long n_out)
    {
        return M == 16 ? launch_win_cplx_m<16, 0, 1, true>(x, n, y, n_out) : M == 32 ? launch_win_cplx_m<32, 0, 1, true>(x, n, y, n_out)
                                                                            : launch_win_cplx_m<64, 0, 1, true>(x, n, y, n_out);
    }
    // (the Float32-stream window kernel was measured the same way and lost: 0.153 / 0.150 ms against 0.136 / 0.136 for the Toeplitz kernel at 16 / 32 taps;
    // what wins there is the plain streaming form, fir_short_real_kernel: four outputs per thread, loads shared through L1)
    bool short_real_ok() const
    {
        static const bool off = getenv("LRHIP_NO_FIR_WIN_SHORT") != nullptr;
        return !off && S == 1 && !taps_complex && D == 1 && (M == 16 || M == 32) && !rot && !fft_arith && !use_fft && !pre_disc && !post_disc;
    }
    int launch_short_real(const float *x, long n, float *y)
    {
        const float *h = (const float *)hist[cur].p + hist_pad, *t = (const float *)d_taps.p;
        float *ho = M > 1 ? (float *)hist[cur ^ 1].p + hist_pad : nullptr;
        const unsigned grid = grid_for((unsigned long)((n + 3) / 4), 256);
        if (M == 16) hipLaunchKernelGGL(fir_short_real_kernel<16>, dim3(grid), dim3(256), 0, ctx().stream, h, x, t, y, n, ho);
        else hipLaunchKernelGGL(fir_short_real_kernel<32>, dim3(grid), dim3(256), 0, ctx().stream, h, x, t, y, n, ho);
        LR_LAUNCH_CHECK();
        hist_in_kernel = ho != nullptr;
        return 0;
    }
    int launch_win_pair(const float *x, long n, float *y, long n_out)
    {
        return iir_fused ? launch_win_cplx_m<136, FWC_PAIR | FWC_IIR>(x, n, y, n_out) : launch_win_cplx_m<136, FWC_PAIR>(x, n, y, n_out);
    }

    template <int SS>
    int dispatch_mfma(const float *x, long n, float *y, long n_out)
    {
        switch (D) {
            case 1: return launch_mfma<SS, 1, LRHIP_FIR_D1_NACC>(x, n, y, n_out);
            case 2: return launch_mfma<SS, 2, 4>(x, n, y, n_out);
            case 3: return launch_mfma<SS, 3, 2>(x, n, y, n_out);
            case 4: return launch_mfma<SS, 4, 2>(x, n, y, n_out);
            case 5: {
                static const int nacc5 = getenv("LRHIP_FIR_D5_NACC") ? atoi(getenv("LRHIP_FIR_D5_NACC")) : 2;      // A/B knob: accumulators per wave at D = 5
                return nacc5 == 1 ? launch_mfma<SS, 5, 1>(x, n, y, n_out) : launch_mfma<SS, 5, 2>(x, n, y, n_out);
            }
            case 6: return launch_mfma<SS, 6, 1>(x, n, y, n_out);
            case 7: return launch_mfma<SS, 7, 1>(x, n, y, n_out);
            case 8: return launch_mfma<SS, 8, 1>(x, n, y, n_out);
            case 10: return launch_mfma<SS, 10, 1>(x, n, y, n_out);
            default: return decim_lds_ok() ? launch_decim_lds(x, n, y, n_out) : launch_direct(x, n, y, n_out);
        }
    }

    static bool mfma_supported_decim(unsigned d) { return (d >= 1 && d <= 8) || d == 10; }
    // the discriminator epilogue exists for the persistent instantiations of the complex-stream, real-taps kernel
    // (round 5: and for the Tuner - rotator fused - at decimation 4, 8, 10 with 128 taps: FM receivers at other input rates)
    bool can_post_disc() const
    {
        if (decfft || win_cplx_ok()) return true;
        if (!(S == 2 && !taps_complex && !fft_arith && !use_fft)) return false;
        if ((D == 1 && ksteps == 36) || (D == 5 && ksteps == 51)) return true;
        // round 5: the second LDS-staged decimator form (kernels_firdecim.h) has the epilogue at every decimation it takes - Tuner(.., 50) / (.., 80) +
        // FrequencyDiscriminator of rtlsdr_nbfm.lua, rtlsdr_pocsag.lua, rtlsdr_ax25.lua: one launch less, the ComplexFloat32 tuner output never reaches HBM
        static const bool no_lds_disc = getenv("LRHIP_NO_DISC_EPI_LDS") != nullptr;      // A/B knob
        if (!no_lds_disc && ksteps == 0 && D > 1 && decim_lds_ok() && decim_lds2_ok()) return true;
        static const bool off = getenv("LRHIP_NO_DISC_EPI_OTHER_D") != nullptr;      // A/B knob: the round-4 behaviour
        return !off && rot && (D == 4 || D == 8 || D == 10) && ksteps == disc_ksteps(D);
    }

    // round 5 (host_execute's direct mode, the ring's in-place input): the forms that stage their input through LDS ONCE in one launch and never read
    // their output back - overlap-save in one launch, the Toeplitz kernels, the LDS-staged decimators, the polyphase-FFT decimator.  Not the last-resort
    // direct kernel (M global reads per output), the multi-launch partitioned filters (they accumulate into y), the opt-in window kernels
    bool direct_io_ok() const override
    {
        if (pre_disc || fix_src) return false;
        if (use_fft) return false;            // the reference's block-emission framing: run() copies x into `pending` / `work` first (a second pass, device-to-device)
        if (decfft) return true;
        if (fft_arith) return M <= FFT_PART || (S == 2 && (fft4k_V || (fft64_np && fft64_np <= 2)));      // (4 098 taps and more: the second launch re-reads y)
        if (win_real_ok() || win_cplx_ok() || short_real_ok()) return false;
        return ksteps != 0 || (D > 1 && decim_lds_ok());
    }

    // filter n inputs (device), emit the retained outputs; advances history / index / count
    long core(const float *x, long n, float *y, unsigned long cap)
    {
        if (n <= 0) return 0;
        hist_in_kernel = false;
        fix_ready = false;
        long n_out = (unsigned long)n > index ? (long)((n - index + D - 1) / D) : 0;
        if ((unsigned long)n_out > cap) return set_error("fir: output capacity %lu < %ld", cap, n_out);
        if (fix_src && fix_src->fix_ready && n_out == 0) {
            // the producer left its wave-first discriminator outputs to this stage's staging, but this chunk launches no window kernel
            // (fewer inputs than the decimation index): patch them in place before they enter the history
            hipLaunchKernelGGL(fir_disc_fixup_kernel, dim3((unsigned)((fix_src->fix_nunits + 255) / 256)), dim3(256), 0, ctx().stream,
                               (const float2 *)fix_src->edge.p, fix_src->fix_nunits, fix_src->fix_unit, const_cast<float *>(x), n, fix_src->fix_prev_ptr,
                               1.0 / fix_src->disc_gain);
            LR_LAUNCH_CHECK();
            fix_src->fix_ready = false;
        }
        if (n_out > 0) {
            int rc = (decfft && ((uintptr_t)x & 7) == 0) ? launch_decfft(x, n, y, n_out)
                     : fft_arith ? launch_fft(x, n, y, n_out)
                     : win_real_ok() ? launch_win_real(x, n, y)
                     : win_cplx_ok() ? launch_win_cplx(x, n, y, n_out)
                     : win_short_ok() ? launch_win_short(x, n, y, n_out)
                     : win_short_c_ok() ? launch_win_short_c(x, n, y, n_out)
                     : short_real_ok() ? launch_short_real(x, n, y)
                     : win_pair_ok() ? launch_win_pair(x, n, y, n_out)
                     : !ksteps ? (decim_lds_ok() ? launch_decim_lds(x, n, y, n_out) : launch_direct(x, n, y, n_out))
                     : taps_complex ? dispatch_mfma_cc(x, n, y, n_out)
                     : S == 1 ? dispatch_mfma<1>(x, n, y, n_out) : dispatch_mfma<2>(x, n, y, n_out);
            if (rc) return rc;
        }
        if (pre_disc) {
            unsigned grid = grid_for((unsigned long)(M > 1 ? M - 1 : 1), 256);
            float2 *dp = (float2 *)disc_prev.p;
            hipLaunchKernelGGL(fir_fft_pre_history_kernel, dim3(grid), dim3(256), 0, ctx().stream, (const float *)hist[cur].p, x, (float *)hist[cur ^ 1].p, M, n,
                               1.0 / disc_gain, (const float2 *)(dp + disc_cur), dp + (disc_cur ^ 1));
            LR_LAUNCH_CHECK();
            cur ^= 1;
            disc_cur ^= 1;
        } else if (hist_in_kernel) {
            cur ^= 1;
        } else if (M > 1) {
            unsigned grid = grid_for((unsigned long)(M - 1) * S, 256);
            const float *hi = (const float *)hist[cur].p + hist_pad;
            float *ho = (float *)hist[cur ^ 1].p + hist_pad;
            if (S == 1)
                hipLaunchKernelGGL(fir_history_kernel<1>, dim3(grid), dim3(256), 0, ctx().stream, hi, x, ho, M, n);
            else
                hipLaunchKernelGGL(fir_history_kernel<2>, dim3(grid), dim3(256), 0, ctx().stream, hi, x, ho, M, n);
            LR_LAUNCH_CHECK();
            cur ^= 1;
        }
        index = index + (unsigned long)n_out * D - (unsigned long)n;
        count += (uint64_t)n;
        return n_out;
    }

    // does this chunk reach the persistent Tuner kernel, the one with a record instantiation?  (core()'s dispatch, the D = 5 / 128-tap shape)
    bool raw_path_ok(const void *in_dev, unsigned long n_in) const
    {
        if (!in_fmt || post_disc || pre_disc || use_fft || decfft || fft_arith || taps_complex || S != 2) return false;
        // the two kernels with record instantiations: the persistent Toeplitz kernel (128 taps, decimation 5) and the LDS-staged decimator (no Toeplitz shape)
        if (!((D == 5 && ksteps == 51) || (ksteps == 0 && decim_lds_ok()))) return false;
        if (win_cplx_ok() || win_pair_ok() || win_short_ok() || win_short_c_ok() || win_real_ok() || short_real_ok()) return false;
        if (n_in <= index) return false;                                  // no output: nothing launches, the history kernel would read x
        return ((uintptr_t)in_dev % (in_fmt == RX_FMT_S16LE ? 4u : 2u)) == 0;
    }
    long run(const void *in_dev, unsigned long n_in, void *out_dev, unsigned long cap) override
    {
        const float *x = (const float *)in_dev;
        float *y = (float *)out_dev;
        if (in_fmt) {
            static const bool no_raw = getenv("LRHIP_TUNER_NO_RAW") != nullptr;      // A/B knob: conversion launch first
            if (!n_in) return 0;
            if (!no_raw && raw_path_ok(in_dev, n_in)) {
                raw_now = true;
                const long rc = core(x, (long)n_in, y, cap);
                raw_now = false;
                return rc;
            }
            if (converted.reserve((size_t)n_in * 8 + 16)) return -1;
            const long m = fmt_stage->run(in_dev, n_in, converted.p, n_in);
            if (m < 0) return m;
            x = (const float *)converted.p;
        }
        if (!use_fft) return core(x, (long)n_in, y, cap);
        // overlap-save framing: emit only whole L-blocks, keep the tail pending (firfilter.lua:451-485)
        long total = fill + (long)n_in, emit = (total / L) * L;
        size_t ss = (size_t)S * sizeof(float);
        if (emit == 0) {
            if (n_in) LR_HIP(hipMemcpyAsync((char *)pending.p + fill * ss, x, n_in * ss, hipMemcpyDeviceToDevice, ctx().stream));
            fill = total;
            return 0;
        }
        if ((unsigned long)emit > cap) return set_error("fir(fft framing): output capacity %lu < %ld", cap, emit);
        if (work.reserve((size_t)((long)n_in + L) * ss)) return -1;      // the largest total this chunk size can see: no regrowth as `fill` moves
        if (fill) LR_HIP(hipMemcpyAsync(work.p, pending.p, fill * ss, hipMemcpyDeviceToDevice, ctx().stream));
        LR_HIP(hipMemcpyAsync((char *)work.p + fill * ss, x, n_in * ss, hipMemcpyDeviceToDevice, ctx().stream));
        long rc = core((const float *)work.p, emit, y, cap);
        if (rc < 0) return rc;
        fill = total - emit;
        if (fill) LR_HIP(hipMemcpyAsync(pending.p, (char *)work.p + emit * ss, fill * ss, hipMemcpyDeviceToDevice, ctx().stream));
        return emit;
    }
};

// Tables of fir_decfft_kernel (layout: df_table_elems): 256-point twiddles, the D polyphase branch responses G_rho (1/256 of the
// inverse transform folded in) in the lane order of the forward batches, and the output phasors e^{j w D k}.  All in double.
// Window position j = D mm + rho of a block is input x[first + D m - r] with r = D - 1 - rho, so branch rho uses g_r[q] = g[D q + r],
// g[i] = h[i] e^{-j w i}.
static void decfft_build_tables(const float *taps, int M, int taps_complex, int D, double omega, std::vector<float> &out)
{
    const double PI2 = 6.283185307179586476925286766559;
    const int A = D / 4, C = D % 4;
    out.assign((size_t)df_table_elems(D) * 2, 0.f);
    float *tw = out.data(), *Gf = tw + 2 * 256, *Gl = Gf + 2 * (size_t)A * 1024, *rt = Gl + 2 * (size_t)C * 256;
    for (int k1 = 0; k1 < 16; k1++)
        for (int u = 0; u < 16; u++) {
            double a = -PI2 * (double)(k1 * u) / DF_N;
            tw[2 * (k1 * 16 + u)] = (float)std::cos(a);
            tw[2 * (k1 * 16 + u) + 1] = (float)std::sin(a);
        }
    long double turns = (long double)omega / (2.0L * 3.14159265358979323846264338327950288L);
    turns -= floorl(turns);
    std::vector<double> gr((size_t)M), gi((size_t)M);
    for (int i = 0; i < M; i++) {
        long double f = turns * (long double)i;
        f -= floorl(f);
        double a = -PI2 * (double)f, c = std::cos(a), sn = std::sin(a);
        double hr = taps_complex ? taps[2 * i] : taps[i], hi = taps_complex ? taps[2 * i + 1] : 0.0;
        gr[i] = hr * c - hi * sn;
        gi[i] = hr * sn + hi * c;
    }
    std::vector<double> Gr(DF_N), Gi(DF_N);
    for (int rho = 0; rho < D; rho++) {
        const int r = D - 1 - rho;
        for (int k = 0; k < DF_N; k++) {
            double sr = 0, si = 0;
            for (int qq = 0; D * qq + r < M; qq++) {
                double a = -PI2 * (double)((qq * k) % DF_N) / DF_N, c = std::cos(a), sn = std::sin(a);
                sr += gr[D * qq + r] * c - gi[D * qq + r] * sn;
                si += gr[D * qq + r] * sn + gi[D * qq + r] * c;
            }
            Gr[k] = sr / DF_N;
            Gi[k] = si / DF_N;
        }
        for (int k2 = 0; k2 < 16; k2++)
            for (int k1 = 0; k1 < 16; k1++) {
                const int k = k1 + 16 * k2;
                float *dst = rho < 4 * A ? Gf + 2 * ((size_t)(rho / 4) * 1024 + k2 * 64 + 16 * (rho % 4) + k1)
                                         : Gl + 2 * ((size_t)(rho - 4 * A) * 256 + k2 * 16 + k1);
                dst[0] = (float)Gr[k];
                dst[1] = (float)Gi[k];
            }
    }
    for (int w = 0; w < DF_N; w++) {
        long double f = turns * (long double)D * (long double)w;
        f -= floorl(f);
        double a = PI2 * (double)f;
        rt[2 * w] = (float)std::cos(a);
        rt[2 * w + 1] = (float)std::sin(a);
    }
}

static FirStage *fir_build(const float *taps, unsigned ntaps, int taps_complex, int input_complex, unsigned decim,
                           int use_fft, bool rot, double omega)
{
    if (!taps || ntaps < 1) { set_error("fir: need at least one tap"); return nullptr; }
    if (taps_complex && !input_complex) { set_error("fir: complex taps require ComplexFloat32 input (firfilter.lua:69-74)"); return nullptr; }
    if (decim < 1) { set_error("fir: decimation must be >= 1"); return nullptr; }
    const int mode_req = use_fft;
    // decimating filters: overlap-save arithmetic exists in the polyphase form (kernels_firdecfft.h) for the complex stream
    bool want_decfft = false;
    if (decim > 1 && (use_fft == 2 || use_fft == 3)) {
        // automatic (3) keeps the direct form for decimating filters: on MI355X the Toeplitz MFMA kernel is the faster one there
        // (Tuner + discriminator, 2^26 samples: 0.155 ms against 0.165 ms, same box) and it is bit-exact
        want_decfft = use_fft == 2 && FirStage::decfft_supported(decim, (int)ntaps, input_complex ? 2 : 1);
        // "fast" is an arithmetic preference, not a framing: where no polyphase-FFT kernel exists for (taps, decimation) the direct form runs -
        // stand-alone exactly as lrhip_chain_create does when it fuses filter and downsampler (one behaviour for both front doors)
        use_fft = 0;
    }
    if (use_fft == 3) use_fft = (decim == 1 && !rot && ntaps >= 48 && ntaps <= 16 * FirStage::FFT_PART && (input_complex || !taps_complex)) ? 2 : 0;
    if (use_fft && decim != 1) { set_error("fir: the reference's block-emission framing (use_fft = 1) cannot be combined with decimation"); return nullptr; }
    if (use_fft < 0 || use_fft > 2) { set_error("fir: use_fft must be 0 (direct form), 1 (overlap-save as the reference: block emission), 2 (overlap-save arithmetic, sample-exact emission) or 3 (automatic)"); return nullptr; }
    if (ntaps > (1u << 20)) { set_error("fir: too many taps"); return nullptr; }
    if (ensure_init()) return nullptr;
    std::unique_ptr<FirStage> q(new (std::nothrow) FirStage());
    if (!q) { set_error("out of memory"); return nullptr; }
    q->M = (int)ntaps; q->S = input_complex ? 2 : 1; q->taps_complex = taps_complex; q->D = decim; q->mode_req = mode_req;
    q->use_fft = use_fft == 1; q->rot = rot;     // 1: reference emission framing; 2: FFT arithmetic, sample-exact emission
    q->in_size = q->out_size = 4 * q->S;
    int ts = taps_complex ? 2 : 1;
    q->taps_rev.resize((size_t)ntaps * ts);
    for (unsigned i = 0; i < ntaps; i++)
        for (int c = 0; c < ts; c++) q->taps_rev[(size_t)i * ts + c] = taps[(size_t)(ntaps - 1 - i) * ts + c];
    if (upload(q->d_taps, q->taps_rev.data(), q->taps_rev.size() * sizeof(float))) return nullptr;
    if (!taps_complex && FirStage::mfma_supported_decim(decim)) {
        int ks = fir_mfma_ksteps(q->M, (int)decim, q->S);
        if ((size_t)fir_taps_len((int)decim, ks) * sizeof(float) <= 24 * 1024) {   // tap array must leave LDS room for the tile
            std::vector<float> tab;
            fir_mfma_build_taps(q->taps_rev.data(), q->M, (int)decim, ks, tab);
            if (upload(q->d_atab, tab.data(), tab.size() * sizeof(float))) return nullptr;
            q->ksteps = ks;
        }
    }
    if (taps_complex && decim == 1 && input_complex && (ntaps == 16 || ntaps == 32)) {
        // short complex filters as a streaming kernel (launch_win_short_c): (re, im, -im, re) per reversed tap
        std::vector<float> t4((size_t)4 * ntaps);
        for (unsigned j = 0; j < ntaps; j++) {
            const float hr = q->taps_rev[2 * j], hi = q->taps_rev[2 * j + 1];
            t4[4 * j] = hr; t4[4 * j + 1] = hi; t4[4 * j + 2] = -hi; t4[4 * j + 3] = hr;
        }
        if (upload(q->d_ctaps4, t4.data(), t4.size() * sizeof(float))) return nullptr;
    }
    if (taps_complex && decim <= 5) {
        // taps'_re = interleave(hr_rev, -hi_rev), taps'_im = interleave(hi_rev, hr_rev) over the float stream
        int M2 = 2 * q->M, D2 = 2 * (int)decim;
        int ks = fir_mfma_ksteps(M2, D2, 1, 2);          // 8-B aligned complex input => float slack e in {0, 2}
        if ((size_t)2 * fir_taps_len(D2, ks) * sizeof(float) <= 32 * 1024) {
            std::vector<float> tre((size_t)M2), tim((size_t)M2), tab;
            for (int j = 0; j < q->M; j++) {
                float hr = q->taps_rev[2 * j], hi = q->taps_rev[2 * j + 1];
                tre[2 * j] = hr; tre[2 * j + 1] = -hi;
                tim[2 * j] = hi; tim[2 * j + 1] = hr;
            }
            std::vector<float> are, aim;
            fir_mfma_build_taps(tre.data(), M2, D2, ks, are);
            fir_mfma_build_taps(tim.data(), M2, D2, ks, aim);
            tab = are;
            tab.insert(tab.end(), aim.begin(), aim.end());
            if (upload(q->d_atab, tab.data(), tab.size() * sizeof(float))) return nullptr;
            q->ksteps = ks;
            q->hist_pad = 1;
        }
    }
    if (want_decfft) {
        std::vector<float> tab;
        decfft_build_tables(taps, (int)ntaps, taps_complex, (int)decim, rot ? omega : 0.0, tab);
        if (upload(q->d_dec_tables, tab.data(), tab.size() * sizeof(float))) return nullptr;
        q->decfft = true;
        q->rot_omega = rot ? omega : 0.0;
    }
    if (rot) {
        if (!q->decfft && !q->ksteps && !(input_complex && !taps_complex && (int)ntaps + 255 <= DECIM_SPAN_MAX)) {
            set_error("fir: rotator fusion unavailable for this tap count / decimation");
            return nullptr;
        }
        long double turns = (long double)omega / (2.0L * 3.14159265358979323846264338327950288L);
        turns -= floorl(turns);
        q->rot_step = (uint64_t)(turns * 18446744073709551616.0L);
    }
    if (use_fft && decim == 1 && !rot && ntaps >= 32 && ntaps <= 16 * FirStage::FFT_PART && (input_complex || !taps_complex)) {
        // fused overlap-save kernel tables, one set per partition of <= FFT_PART taps: tw1[k1][t] | Hperm[4j+k3][lane] | tw2[k2][t2]
        const double PI2 = 6.283185307179586476925286766559;
        const int nparts = ((int)ntaps + FirStage::FFT_PART - 1) / FirStage::FFT_PART;
        std::vector<float> tab((size_t)nparts * FFT_TABLE_ELEMS * 2);
        for (int part = 0; part < nparts; part++) {
            float *tp = tab.data() + (size_t)part * FFT_TABLE_ELEMS * 2;
            const unsigned m0 = (unsigned)part * FirStage::FFT_PART, m1 = std::min<unsigned>(ntaps, m0 + FirStage::FFT_PART);
            for (int k1 = 0; k1 < 16; k1++)
                for (int t = 0; t < 64; t++) {
                    double a = -PI2 * (double)((k1 * t) % FFTN) / FFTN;
                    tp[2 * (k1 * 64 + t)] = (float)std::cos(a);
                    tp[2 * (k1 * 64 + t) + 1] = (float)std::sin(a);
                }
            std::vector<double> Hr(FFTN, 0.0), Hi(FFTN, 0.0);
            for (int k = 0; k < FFTN; k++) {
                double sr = 0, si = 0;
                for (unsigned m = m0; m < m1; m++) {
                    double a = -PI2 * (double)((k * (long)(m - m0)) % FFTN) / FFTN, c = std::cos(a), sn = std::sin(a);
                    double hr = taps_complex ? taps[2 * m] : taps[m], hi = taps_complex ? taps[2 * m + 1] : 0.0;
                    sr += hr * c - hi * sn;
                    si += hr * sn + hi * c;
                }
                Hr[k] = sr / FFTN;      // the 1/N of the inverse transform (spectrum_utils.lua:335-338) folded in
                Hi[k] = si / FFTN;
            }
            for (int j = 0; j < 4; j++)
                for (int k3 = 0; k3 < 4; k3++)
                    for (int lane = 0; lane < 64; lane++) {
                        int qq = lane & 3, k1 = lane >> 2;
                        int k = k1 + 16 * (4 * j + qq) + 256 * k3;
                        size_t o = (size_t)16 * 64 + (size_t)(4 * j + k3) * 64 + lane;
                        tp[2 * o] = (float)Hr[k];
                        tp[2 * o + 1] = (float)Hi[k];
#if LRHIP_FFT_E2_SWAP
                        // fir_fft_kernel's numbering with the register <-> row transposes: q = the lane's row of 16, k1 = its position in the row
                        const int qs = lane >> 4, k1w = lane & 15, ks = k1w + 16 * (4 * j + qs) + 256 * k3;
                        const size_t os = (size_t)FFT_TABLE_HSW + (size_t)(4 * j + k3) * 64 + lane;
                        tp[2 * os] = (float)Hr[ks];
                        tp[2 * os + 1] = (float)Hi[ks];
#endif
                    }
            for (int k2 = 0; k2 < 16; k2++)
                for (int t2 = 0; t2 < 4; t2++) {
                    double a = -PI2 * (double)((k2 * t2) % 64) / 64.0;
                    size_t o = (size_t)2 * 16 * 64 + k2 * 4 + t2;
                    tp[2 * o] = (float)std::cos(a);
                    tp[2 * o + 1] = (float)std::sin(a);
                }
        }
        if (upload(q->d_fft_tables, tab.data(), tab.size() * sizeof(float))) return nullptr;
        q->fft_arith = true;
        // (round 6: Float32 streams with real taps take the 64 x 64 kernel too - two stream blocks per transform; the workgroup-per-block kernel of small
        // launches stays ComplexFloat32-only, small Float32 launches keep the partitioned kernel)
        if ((input_complex || !taps_complex) && (int)ntaps > FirStage::FFT_PART && ntaps <= 1281) {
            // tables of the 4096-point kernel (kernels_firfft4k.h): tw1 | tw2 | c[w][i] = W_64^(i w) | b[w][t] = W_4096^(t w) | H[w][16 x 64] of the bins w + 4 k'
            std::vector<float> t4((size_t)F4K_TABLE_ELEMS * 2);
            auto put = [&](size_t o, double a) { t4[2 * o] = (float)std::cos(a); t4[2 * o + 1] = (float)std::sin(a); };
            for (int k1 = 0; k1 < 16; k1++)
                for (int t = 0; t < 64; t++) put((size_t)k1 * 64 + t, -PI2 * (double)((k1 * t) % FFTN) / FFTN);
            for (int k2 = 0; k2 < 16; k2++)
                for (int t2 = 0; t2 < 4; t2++) put((size_t)16 * 64 + k2 * 4 + t2, -PI2 * (double)((k2 * t2) % 64) / 64.0);
            for (int w = 0; w < 4; w++) {
                for (int i = 0; i < 16; i++) put((size_t)16 * 64 + 64 + w * 16 + i, -PI2 * (double)((i * w) % 64) / 64.0);
                for (int t = 0; t < 64; t++) put((size_t)F4K_TAB_B + w * 64 + t, -PI2 * (double)(t * w) / F4K_N);
            }
            std::vector<double> cs(F4K_N), sn(F4K_N), Hr(F4K_N), Hi(F4K_N);
            for (int k = 0; k < F4K_N; k++) { cs[k] = std::cos(-PI2 * k / F4K_N); sn[k] = std::sin(-PI2 * k / F4K_N); }
            for (int k = 0; k < F4K_N; k++) {
                double sr = 0, si = 0;
                for (unsigned m = 0; m < ntaps; m++) {
                    const int a = (int)(((long)k * m) % F4K_N);
                    const double hr = taps_complex ? taps[2 * m] : taps[m], hi = taps_complex ? taps[2 * m + 1] : 0.0;
                    sr += hr * cs[a] - hi * sn[a];
                    si += hr * sn[a] + hi * cs[a];
                }
                Hr[k] = sr / F4K_N;
                Hi[k] = si / F4K_N;
            }
            for (int w = 0; w < 4; w++)
                for (int j = 0; j < 4; j++)
                    for (int k3 = 0; k3 < 4; k3++)
                        for (int lane = 0; lane < 64; lane++) {
                            const int qq = lane & 3, k1 = lane >> 2, kp = k1 + 16 * (4 * j + qq) + 256 * k3, k = w + 4 * kp;
                            const size_t o = (size_t)F4K_TAB_H + (size_t)w * 1024 + (size_t)(4 * j + k3) * 64 + lane;
                            t4[2 * o] = (float)Hr[k];
                            t4[2 * o + 1] = (float)Hi[k];
                        }
            if (input_complex && upload(q->d_fft4k_tables, t4.data(), t4.size() * sizeof(float))) return nullptr;
            // tables of the 64 x 64 form: C[c][t] = W_1024^(t c) | D[d][t] = W_4096^(t d) | H[r][l] = H(64 k1(r) + l) | Hsym[k1][l <= 32] = H(64 k1 + l)
            std::vector<float> t6((size_t)F64_TABLE_ELEMS * 2, 0.f);
            auto put6 = [&](size_t o, double a) { t6[2 * o] = (float)std::cos(a); t6[2 * o + 1] = (float)std::sin(a); };
            for (int c = 0; c < 16; c++)
                for (int t = 0; t < 64; t++) put6((size_t)c * 64 + t, -PI2 * (double)((c * t) % FFTN) / FFTN);
            for (int d = 0; d < 4; d++)
                for (int t = 0; t < 64; t++) put6((size_t)F64_TAB_D + d * 64 + t, -PI2 * (double)(t * d) / F4K_N);
            for (int r = 0; r < 64; r++)
                for (int l = 0; l < 64; l++) {
                    const int k = 64 * f64_index(r) + l;
                    const size_t o = (size_t)F64_TAB_H + (size_t)r * 64 + l;
                    t6[2 * o] = (float)Hr[k];
                    t6[2 * o + 1] = (float)Hi[k];
                    if (l <= 32) {
                        const size_t os = (size_t)F64_TAB_HSYM + (size_t)f64_index(r) * F64_HSYM_ROW + l;
                        t6[2 * os] = (float)Hr[k];
                        t6[2 * os + 1] = (float)Hi[k];
                    }
                }
            if (upload(q->d_fft64_tables, t6.data(), t6.size() * sizeof(float))) return nullptr;
            q->fft4k_V = (int)((ntaps - 1 + 255) / 256) * 256;
            if (q->fft4k_V < 768) q->fft4k_V = 768;
        }
        if ((input_complex || !taps_complex) && ntaps > 1281 && ntaps <= 8193) {
            // round 5: the 64 x 64 form at an overlap of 2 048 - one partition to 2 049 taps, two (taps [0, 2 048) and [2 048, ntaps)) above;
            // round 6: three / four partitions of 2 048 taps (the last up to 2 049) to 8 193 taps - a second table set for the second launch
            const int np = (int)((ntaps + 2046) / 2048);                      // ceil((ntaps - 1) / 2 048): 1 to 2 049 taps, 2 to 4 097, 3 to 6 145, 4 to 8 193
            const int nsets = np > 2 ? 2 : 1;
            std::vector<float> t6((size_t)F64_TABLE_ELEMS2 * 2 * nsets, 0.f);
            auto put6 = [&](size_t o, double a) { t6[2 * o] = (float)std::cos(a); t6[2 * o + 1] = (float)std::sin(a); };
            for (int set = 0; set < nsets; set++) {
                const size_t so = (size_t)set * F64_TABLE_ELEMS2;
                for (int c = 0; c < 16; c++)
                    for (int t = 0; t < 64; t++) put6(so + (size_t)c * 64 + t, -PI2 * (double)((c * t) % FFTN) / FFTN);
                for (int d = 0; d < 4; d++)
                    for (int t = 0; t < 64; t++) put6(so + (size_t)F64_TAB_D + d * 64 + t, -PI2 * (double)(t * d) / F4K_N);
            }
            std::vector<double> cs(F4K_N), sn(F4K_N), Hr(F4K_N), Hi(F4K_N);
            for (int k = 0; k < F4K_N; k++) { cs[k] = std::cos(-PI2 * k / F4K_N); sn[k] = std::sin(-PI2 * k / F4K_N); }
            for (int part = 0; part < np; part++) {
                const unsigned m0 = np == 1 ? 0 : part * 2048u, m1 = np == 1 ? ntaps : (part + 1 < np ? (part + 1) * 2048u : ntaps);
                const size_t so = (size_t)(part / 2) * F64_TABLE_ELEMS2;
                const bool second = (part & 1) != 0;
                for (int k = 0; k < F4K_N; k++) {
                    double sr = 0, si = 0;
                    for (unsigned m = m0; m < m1; m++) {
                        const int a = (int)(((long)k * (m - m0)) % F4K_N);
                        const double hr = taps_complex ? taps[2 * m] : taps[m], hi = taps_complex ? taps[2 * m + 1] : 0.0;
                        sr += hr * cs[a] - hi * sn[a];
                        si += hr * sn[a] + hi * cs[a];
                    }
                    Hr[k] = sr / F4K_N;
                    Hi[k] = si / F4K_N;
                }
                for (int r = 0; r < 64; r++)
                    for (int l = 0; l < 64; l++) {
                        const int k = 64 * f64_index(r) + l;
                        const size_t o = so + (size_t)(second ? F64_TAB_H1 : F64_TAB_H) + (size_t)r * 64 + l;
                        t6[2 * o] = (float)Hr[k];
                        t6[2 * o + 1] = (float)Hi[k];
                        if (part == 0 && l <= 32) {
                            const size_t os = (size_t)F64_TAB_HSYM + (size_t)f64_index(r) * F64_HSYM_ROW + l;
                            t6[2 * os] = (float)Hr[k];
                            t6[2 * os + 1] = (float)Hi[k];
                        }
                    }
            }
            if (upload(q->d_fft64_tables, t6.data(), t6.size() * sizeof(float))) return nullptr;
            q->fft64_np = np;
        }
    }
    if (q->use_fft) {
        long N = 1L << (long)std::floor(std::log(8.0 * ntaps) / std::log(2.0));   // firfilter.lua:329
        q->L = N - (long)ntaps + 1;
        if (q->pending.reserve((size_t)q->L * q->S * sizeof(float))) return nullptr;
    }
    if (q->reset()) return nullptr;
    return q.release();
}
