// stage_rx.h - the FM receiver's device path as one stage: tuner + discriminator (FirStage A) and the polyphase audio tail (FirStage B)
// run by ONE launch of rx_fused_kernel (kernels_rx.h), with the two-launch form of round 2 behind it for the chunks and settings the
// single launch does not take (LRHIP_CHAIN_NO_SINGLE_LAUNCH, an unaligned source, chunks that emit no audio sample).
// (part of liblrhip.so; included by lrhip.hip in this order, one translation unit)
#pragma once

struct RxStage : lrhip_stage {
    std::unique_ptr<FirStage> A, B;       // state lives in the two stages, in their own formats: both forms can alternate chunk by chunk
    DeviceBuf mid;                        // two-launch form: the discriminator stream between them
    DeviceBuf d_ptab4;
    int blocks_per_cu = 0, blocks_per_cu8 = 0;
    int in_fmt = RX_FMT_CF32;             // RX_FMT_* of the folded format stage
    // round 3: an IQFileSource format stage for unsigned 8-bit records directly in front of the receiver is folded into it (lrhip_chain_create): the single
    // launch reads the 2-byte records and converts them on the way into LDS (kernels_rx.h, U8); the stage itself (not owned) serves the fall-back form
    bool in_u8 = false;
    lrhip_stage *fmt = nullptr;
    DeviceBuf converted;
    bool single_launch = true;
    int last_form = 0;                    // diagnostics: 1 = single launch, 2 = two launches, 3 = single launch on u8 records (last run)

    const char *kind() const override { return "fm-receiver"; }
    bool direct_io_ok() const override { return true; }      // both forms read the RF input once (round 5: host_execute's direct mode, the ring's in-place input)
    unsigned long max_output(unsigned long n) const override { return B->max_output(A->max_output(n)); }
    int reset() override { return (A->reset() || B->reset()) ? -1 : 0; }
    int seek(unsigned long long n0, unsigned long long *n0_out) override
    {
        unsigned long long m = 0;
        if (A->seek(n0, &m) || B->seek(m, n0_out)) return -1;
        return 0;
    }
    long memory() const override
    {
        const long ma = A->memory(), mb = B->memory();
        return (ma < 0 || mb < 0) ? -1 : ma + (long)A->D * mb;
    }
    void rate(unsigned long *num, unsigned long *den) const override { *num = (unsigned long)A->D * B->D; *den = 1; }
    unsigned long align() const override
    {
        // tile grids of the two-launch form (which a partition may fall back to) and the batch grid of the single launch
        auto gcd = [](unsigned long x, unsigned long y) { while (y) { unsigned long t = x % y; x = y; y = t; } return x; };
        auto lcm = [&](unsigned long x, unsigned long y) { return x / gcd(x, y) * y; };
        return lcm(lcm(A->align(), (unsigned long)A->D * B->align()), (unsigned long)RX_BATCH * RX_D);
    }

    // The single launch restarts the low-rate recurrence of every workgroup run from a ZERO state RX_WARM_OUTPUTS audio samples early
    // (kernels_rx.h: the tile in front of the run).  What is left of the true state there is q^75 of it: 1.4e-10 for the stock de-emphasis
    // (75 us at 220.5 kHz: q = 0.739), but fuse_iir1() admits poles up to q^1280 < 1e-12 (the two-launch form warms up over up to four waves).
    // A slower pole - the same receiver at 2.048 / 2.4 MS/s, a longer time constant - would leave a seam at every run boundary, so such chains
    // keep the two-launch form.  Bound: half an ulp of unit-scale audio.
    static constexpr int RX_WARM_OUTPUTS = 75;
    static bool pole_ok(const FirStage *b)
    {
        const double q = std::fabs((double)b->iir_na1 + (double)b->iir_na1_lo);
        return q < 1.0 && std::pow(q, (double)RX_WARM_OUTPUTS) <= 0x1p-25;
    }

    static bool shapes_ok(const FirStage *a, const FirStage *b)
    {
        return a && b && b->iir_fused && pole_ok(b) && a->S == 2 && !a->taps_complex && a->D == 5 && a->M == RX_M && a->ksteps == RX_KS && a->rot && a->rel_rot && a->post_disc &&
               !a->decfft && !a->fft_arith && !a->use_fft && !a->win_cplx_ok() && b->S == 1 && b->M == RX_MT && b->D == 5 && b->ksteps == RX_KST && b->d_atab.p && b->iir_fused && b->win_pair_ok();
    }

    int prepare()
    {
        in_size = in_u8 ? (in_fmt == RX_FMT_S16LE ? 4 : 2) : A->in_size;
        out_size = B->out_size;
        const double q = (double)B->iir_na1 + (double)B->iir_na1_lo;
        std::vector<float> pt(64);
        const double q4 = q * q * q * q;
        double acc = 1.0;
        for (int l = 0; l < 64; l++) { acc *= q4; pt[(size_t)l] = (float)acc; }      // q^(4 (l+1)): a lane owns four audio outputs
        return upload(d_ptab4, pt.data(), pt.size() * sizeof(float));
    }

    long run_two(const void *in_dev, unsigned long n_in, void *out_dev, unsigned long cap)
    {
        const unsigned long need = A->max_output(n_in);
        if (mid.reserve((size_t)need * sizeof(float) + 16)) return -1;
        long m = A->run(in_dev, n_in, mid.p, need);
        if (m < 0) return m;
        last_form = 2;
        return B->run(mid.p, (unsigned long)m, out_dev, cap);
    }

    long run(const void *in_dev, unsigned long n_in, void *out_dev, unsigned long cap) override
    {
        static const bool env_off = getenv("LRHIP_NO_SINGLE_LAUNCH") != nullptr;      // A/B knob
        static const bool no_u8 = getenv("LRHIP_RX_NO_U8") != nullptr;                // A/B knob: convert first (the file-format kernel), then the receiver
        const long n = (long)n_in;
        const long n_out_a = (unsigned long)n > A->index ? (long)((n - (long)A->index + RX_D - 1) / RX_D) : 0;
        const long n_out_b = (unsigned long)n_out_a > B->index ? (long)((n_out_a - (long)B->index + RX_D - 1) / RX_D) : 0;
        const bool one = single_launch && !env_off && n_out_b >= 1;
        if (in_u8 && (!one || no_u8 || ((uintptr_t)in_dev % (in_fmt == RX_FMT_S16LE ? 4 : 2)) != 0)) {
            // the forms that take ComplexFloat32: convert the records with the format stage's own kernel first
            if (!n_in) return 0;
            if (converted.reserve((size_t)n_in * 8 + 16)) return -1;
            const long m = fmt->run(in_dev, n_in, converted.p, n_in);
            if (m < 0) return m;
            return run_cf32(converted.p, n_in, out_dev, cap, one, n_out_a, n_out_b, false);
        }
        return run_cf32(in_dev, n_in, out_dev, cap, one, n_out_a, n_out_b, in_u8);
    }

    long run_cf32(const void *in_dev, unsigned long n_in, void *out_dev, unsigned long cap, bool one, long n_out_a, long n_out_b, bool u8)
    {
        const long n = (long)n_in;
        if (!one || (!u8 && ((uintptr_t)in_dev % 8) != 0)) return run_two(in_dev, n_in, out_dev, cap);
        if ((unsigned long)n_out_b > cap) return set_error("fm-receiver: output capacity %lu < %ld", cap, n_out_b);
        const float *x = (const float *)in_dev;
        const size_t lds_bytes = (size_t)RX_LDS_FLOATS * sizeof(float);
        int &bpc = u8 ? blocks_per_cu8 : blocks_per_cu;
        const int fmt_k = u8 ? in_fmt : RX_FMT_CF32;
        auto with_kernel = [&](auto fn) -> int {
            switch (fmt_k) {
                case RX_FMT_U8: return fn(rx_fused_kernel<RX_FMT_U8>);
                case RX_FMT_S8: return fn(rx_fused_kernel<RX_FMT_S8>);
                case RX_FMT_S16LE: return fn(rx_fused_kernel<RX_FMT_S16LE>);
                default: return fn(rx_fused_kernel<RX_FMT_CF32>);
            }
        };
        if (!bpc) {
            const int rc0 = with_kernel([&](auto kern) -> int {
                LR_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                int nb = 0;
                LR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, lds_bytes));
                bpc = nb < 1 ? 1 : nb;
                return 0;
            });
            if (rc0) return rc0;
        }
        RxParams pr;
        memset(&pr, 0, sizeof(pr));
        pr.hist = (const float *)A->hist[A->cur].p; pr.x = x; pr.n = n; pr.taps_pad = (const float *)A->d_atab.p;
        pr.n_out_a = n_out_a; pr.first_a = (long)A->index;
        // alignment of the staged window (16-byte loads of ComplexFloat32 pairs, 4-byte loads of record pairs): slack of 0 or 1 sample
        const long v = (long)((uintptr_t)x / (u8 ? (in_fmt == RX_FMT_S16LE ? 4 : 2) : 8)) + (long)A->index - (RX_M - 1);
        pr.e = (int)(((v % 2) + 2) % 2);
        pr.ntiles = (n_out_a + RX_TILE - 1) / RX_TILE;
        pr.rot_step_fx = A->rot_step; pr.rot_count0 = A->count;
        float2 *dp = (float2 *)A->disc_prev.p;
        pr.prev_in = dp + A->disc_cur; pr.prev_out = dp + (A->disc_cur ^ 1);
        pr.inv_gain = 1.0 / A->disc_gain;
        pr.hist_out = (float *)A->hist[A->cur ^ 1].p;
        pr.g_pad = (const float *)B->d_atab.p;
        pr.thist_in = (const float *)B->hist[B->cur].p; pr.thist_out = (float *)B->hist[B->cur ^ 1].p;
        pr.first_b = (long)B->index; pr.n_out_b = n_out_b; pr.y = (float *)out_dev;
        pr.b0 = B->iir_b0; pr.na1 = B->iir_na1; pr.na1_lo = B->iir_na1_lo; pr.ptab4 = (const float *)d_ptab4.p;
        pr.state_in = (const float *)B->iir_state[B->iir_cur].p; pr.state_out = (float *)B->iir_state[B->iir_cur ^ 1].p;
        // one round of workgroups: as many as fit the chip at once; a run costs one extra tile, so short chunks take fewer, longer runs
        long wgs = (long)ctx().num_cus * bpc;
        static const long env_wgs = getenv("LRHIP_RX_WGS_PER_CU") ? atol(getenv("LRHIP_RX_WGS_PER_CU")) : 0;       // A/B knob, read once
        if (env_wgs > 0) wgs = (long)ctx().num_cus * env_wgs;
        const long most = (pr.ntiles + 7) / 8;
        if (wgs < 1 || wgs > most) wgs = most;
        pr.dbg = ablation_bits("LRHIP_RX_DBG");      // ablation bits (WRONG results): 0 unless the library was built with -DLRHIP_ABLATION
        const unsigned grid = (unsigned)wgs;
#ifdef LRHIP_RX_TRACE
        static unsigned long long *trace = nullptr;
        static long trace_launches = 0;
        const size_t trace_n = (size_t)8 * 4 * 64 * 8;
        if (!trace) {
            LR_HIP(hipMalloc(&trace, trace_n * 8));
            LR_HIP(hipMemcpyToSymbol(HIP_SYMBOL(lrhip_rx_trace), &trace, sizeof(trace)));
        }
        if (++trace_launches == 12) LR_HIP(hipMemsetAsync(trace, 0, trace_n * 8, ctx().stream));
#endif
        (void)with_kernel([&](auto kern) -> int { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, ctx().stream, pr); return 0; });
        LR_LAUNCH_CHECK();
#ifdef LRHIP_RX_TRACE
        if (trace_launches == 12) {
            // stamps: 0 loop top, 1 staged, 2 behind barrier A, 3 prefetch issued, 4 matrix product done, 5 exchange done, 6 behind barrier B, 7 angles in P; the
            // time from 7 to the next tile's 0 is the audio tail (every RX_TPB-th tile)
            LR_HIP(hipStreamSynchronize(ctx().stream));
            std::vector<unsigned long long> tr(trace_n);
            LR_HIP(hipMemcpy(tr.data(), trace, trace_n * 8, hipMemcpyDeviceToHost));
            static const char *names[8] = {"stage", "barrierA", "prefetch", "mfma", "exchange", "barrierB", "disc", "tail"};
            for (int w = 0; w < 4; w++) {
                double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tile = 0;
                int cnt = 0;
                for (int b = 0; b < 8; b++)
                    for (int t = 2; t < 62; t++) {
                        const unsigned long long *s = &tr[(((size_t)b * 4 + w) * 64 + t) * 8], *nx = s + 8;
                        if (!s[0] || !s[7] || !nx[0]) continue;
                        for (int i = 0; i < 7; i++) sum[i] += (double)(s[i + 1] - s[i]);
                        sum[7] += (double)(nx[0] - s[7]);
                        tile += (double)(nx[0] - s[0]);
                        cnt++;
                    }
                if (!cnt) continue;
                fprintf(stderr, "rx trace wave %d (%d tiles):", w, cnt);
                for (int i = 0; i < 8; i++) fprintf(stderr, "  %s %.0f", names[i], sum[i] / cnt);
                fprintf(stderr, "  | tile %.0f clocks\n", tile / cnt);
            }
        }
#endif
        // what FirStage::core() does for each of the two stages
        A->hist_in_kernel = true; A->fix_ready = false;
        A->cur ^= 1; A->disc_cur ^= 1;
        A->index = A->index + (unsigned long)n_out_a * RX_D - (unsigned long)n;
        A->count += (uint64_t)n;
        B->cur ^= 1; B->iir_cur ^= 1;
        B->index = B->index + (unsigned long)n_out_b * RX_D - (unsigned long)n_out_a;
        B->count += (uint64_t)n_out_a;
        last_form = u8 ? 3 : 1;
        return n_out_b;
    }
};
