// stage_iir.h - IIRFilterBlock and AGC / PowerSquelch stages (scan kernels)
// (part of liblrhip.so; included by lrhip.hip in this order, one translation unit)
#pragma once

// =====================================================================================================
// IIRFilterBlock
// =====================================================================================================
struct IirStage : lrhip_stage {
    int S = 1, nb = 0, na = 0, P = 0;
    bool scan = false;
    IirCoeffs co;
    DeviceBuf d_tpow, d_ttile, d_tseg;    // A^(LC*2^k), k = 0..8 (9 PxP matrices, float up to order 4, double above); A^TILE in double; per-launch carry powers
    IirSeqCoeffs seq;
    std::vector<double> Ttile;            // A^TILE in double (row-major PxP) for the per-launch carry powers
    int warm_tiles = 0;                   // > 0: A^(warm_tiles*TILE) underflows Float32 -> single-launch iir_stream_kernel
    int warm_chunks = 0;                  // > 0 (first order): |p|^(16 warm_chunks) < 1e-12 with at most half a tile -> one-shot launch, a workgroup per tile
    DeviceBuf xhist[2], state[2], tile_end, tile_start, seq_xs, seq_ys;
    int cur = 0;
    unsigned long D = 1, index = 0;       // fused DownsamplerBlock behind the filter (chains)
    const char *kind() const override { return "iir"; }
    unsigned long max_output(unsigned long n) const override { return D == 1 ? n : n / D + 1; }
    int seek(unsigned long long n0, unsigned long long *n0_out) override
    {
        if (reset()) return -1;
        index = (unsigned long)((D - n0 % D) % D);
        *n0_out = (n0 + D - 1) / D;
        return 0;
    }
    // a recurrence forgets its start only if it decays: the single-launch form already requires A^(warm_tiles * TILE) to underflow Float32
    long memory() const override { return warm_tiles > 0 ? (long)warm_tiles * IIR_TILE + nb : -1; }
    void rate(unsigned long *num, unsigned long *den) const override { *num = D; *den = 1; }
    unsigned long align() const override { return scan ? (unsigned long)IIR_TILE : 1UL; }
    int reset() override
    {
        cur = 0; index = 0;
        for (int i = 0; i < 2; i++) {
            if (zero_fill(xhist[i], sizeof(float) * S * IIR_MAX_NB)) return -1;
            if (zero_fill(state[i], sizeof(float) * S * (IIR_MAX_P + 1))) return -1;
        }
        if (zero_fill(seq_xs, sizeof(float) * S * IIR_SEQ_MAX) || zero_fill(seq_ys, sizeof(float) * S * IIR_SEQ_MAX)) return -1;
        return 0;
    }
    template <int SS, int PP, int NBT>
    int run_scan_nb(const float *x, float *y, long n)
    {
        using ST = typename IirScanT<PP>::T;
        long ntiles = (n + IIR_TILE - 1) / IIR_TILE;
        const float *xh = (const float *)xhist[cur].p, *st = (const float *)state[cur].p;
        float *st_out = (float *)state[cur ^ 1].p;
        const ST *tp = (const ST *)d_tpow.p;
        if (warm_tiles > 0) {
            // tiles per workgroup: enough workgroups to fill the chip a few times over, at most 8 tiles each
            long slots = (long)ctx().num_cus * 8;
            int run = (int)(ntiles / slots);
            run = run < 1 ? 1 : run > 8 ? 8 : run;
            if (run < 2 * warm_tiles && ntiles > 4 * warm_tiles) run = 2 * warm_tiles;      // bound the re-read overhead
            int wc = 0;
            static const bool no_oneshot = getenv("LRHIP_IIR_NO_ONESHOT") != nullptr;      // A/B knob: whole warm-up tiles (round 2)
            static const bool no_coal = getenv("LRHIP_IIR_NO_COAL") != nullptr;            // A/B knob: per-thread chunk loads / stores (round 2)
            // A/B knob: ordinary stores instead of non-temporal ones (measured, 2^26 samples: Float32 first order 0.100 -> 0.082 ms and two poles 0.117 -> 0.091 with
            // them; the ComplexFloat32 kernels do not move - 155 registers = 3 workgroups per CU hold them at 4.8 TB/s; forcing 4 spills 80 bytes and loses 40 %)
            static const bool plain_st = getenv("LRHIP_IIR_PLAIN_STORES") != nullptr;
            static const int run_knob = getenv("LRHIP_IIR_RUN") ? atoi(getenv("LRHIP_IIR_RUN")) : 0;       // A/B knob: tiles per workgroup
            if (warm_chunks > 0 && !no_oneshot) {
                // partial warm-up tile: only its last warm_chunks chunks are read.  Tiles per workgroup by measurement (2^26 samples, one box, ms):
                //   first order Float32        1: 0.104   2: 0.098   4: 0.109   8: 0.112
                //   2 poles, 5 taps, Complex   1: 0.267   2: 0.226   4: 0.219   8: 0.213   16: 0.247      (whole warm-up tile, 8: 0.232)
                wc = warm_chunks;
                if (PP == 1) run = ntiles >= 8L * ctx().num_cus ? 2 : 1;
                if (run_knob > 0) run = run_knob;
            }
            unsigned grid = (unsigned)((ntiles + run - 1) / run);
            hipLaunchKernelGGL((iir_stream_kernel<SS, PP, NBT>), dim3(grid), dim3(256), 0, ctx().stream, x, y, n, xh, st, st_out, (long)D, (long)index, run,
                               wc ? 1 : warm_tiles, wc, co, (float *)xhist[cur ^ 1].p, tp, (no_coal ? 1 : 0) | (plain_st ? 2 : 0));
            LR_LAUNCH_CHECK();
            cur ^= 1;
            return 0;
        }
        if (tile_end.reserve(sizeof(ST) * ntiles * SS * PP) || tile_start.reserve(sizeof(ST) * ntiles * SS * PP) ||
            d_tseg.reserve(sizeof(ST) * 8 * PP * PP)) return -1;
        if (ntiles > 1) {
            hipLaunchKernelGGL((iir_scan_kernel<SS, PP, false, NBT>), dim3((unsigned)ntiles), dim3(256), 0, ctx().stream, x, (float *)nullptr, n, xh,
                               (const ST *)nullptr, (ST *)tile_end.p, st, st_out, 1L, 0L, co, tp);
            LR_LAUNCH_CHECK();
        }
        // carry scan: 256 segments of `seg` tiles; the powers A^(TILE*seg*2^k) are computed on the device in double
        long nt = ntiles > 1 ? ntiles : 1, seg = (nt + 255) / 256;
        hipLaunchKernelGGL((iir_tseg_kernel<PP, ST>), dim3(1), dim3(1), 0, ctx().stream, (const double *)d_ttile.p, seg, (ST *)d_tseg.p);
        hipLaunchKernelGGL((iir_carry_kernel<SS, PP>), dim3(1), dim3(256), 0, ctx().stream, (const ST *)tile_end.p, (ST *)tile_start.p,
                           nt, seg, st, (const ST *)d_tseg.p, tp);
        LR_LAUNCH_CHECK();
        hipLaunchKernelGGL((iir_scan_kernel<SS, PP, true, NBT>), dim3((unsigned)ntiles), dim3(256), 0, ctx().stream, x, y, n, xh,
                           (const ST *)tile_start.p, (ST *)nullptr, st, st_out, (long)D, (long)index, co, tp);
        LR_LAUNCH_CHECK();
        if (nb > 1) {
            hipLaunchKernelGGL(iir_state_kernel<SS>, dim3(1), dim3(64), 0, ctx().stream, x, n, nb, xh, (float *)xhist[cur ^ 1].p);
            LR_LAUNCH_CHECK();
        }
        cur ^= 1;
        return 0;
    }
    template <int SS, int PP>
    int run_scan(const float *x, float *y, long n)
    {
        // the feed-forward loop is unrolled to NBT taps (terms beyond nb are predicated off, not free): 2 = single-pole filters, 4 = biquads and the
        // reference suite's 4-ff-tap entry, 16 = the rest
        // (8 = the reference suite's "5 ff 3 fb" entry on orders up to 4: 0.373 -> measured below with half the predicated terms and 8 fewer staged samples)
        if (nb <= 2) return run_scan_nb<SS, PP, 2>(x, y, n);
        if (nb <= 4) return run_scan_nb<SS, PP, 4>(x, y, n);
        if constexpr (PP <= 4) {
            if (nb <= 8) return run_scan_nb<SS, PP, 8>(x, y, n);
        }
        return run_scan_nb<SS, PP, 16>(x, y, n);
    }
    long run(const void *in_dev, unsigned long n, void *out_dev, unsigned long cap) override
    {
        if (!n) return 0;
        unsigned long n_out = D == 1 ? n : (n > index ? (n - index + D - 1) / D : 0);
        if (n_out > cap) return set_error("iir: output capacity %lu < %lu", cap, n_out);
        const float *x = (const float *)in_dev;
        float *y = (float *)out_dev;
        int rc = 0;
        if (scan) {
            // with a single tile the carry kernel just seeds tile_start[0] from the carried state
#define LR_IIR_P(SS, PP) case PP: rc = run_scan<SS, PP>(x, y, (long)n); break
            if (S == 1) switch (P) { LR_IIR_P(1, 1); LR_IIR_P(1, 2); LR_IIR_P(1, 3); LR_IIR_P(1, 4); LR_IIR_P(1, 5); LR_IIR_P(1, 6); LR_IIR_P(1, 7); default: rc = run_scan<1, 8>(x, y, (long)n); }
            else switch (P) { LR_IIR_P(2, 1); LR_IIR_P(2, 2); LR_IIR_P(2, 3); LR_IIR_P(2, 4); LR_IIR_P(2, 5); LR_IIR_P(2, 6); LR_IIR_P(2, 7); default: rc = run_scan<2, 8>(x, y, (long)n); }
#undef LR_IIR_P
        } else {
            if (S == 1) hipLaunchKernelGGL(iir_seq_kernel<1>, dim3(1), dim3(64), 0, ctx().stream, x, y, (long)n, seq, (float *)seq_xs.p, (float *)seq_ys.p);
            else hipLaunchKernelGGL(iir_seq_kernel<2>, dim3(1), dim3(64), 0, ctx().stream, x, y, (long)n, seq, (float *)seq_xs.p, (float *)seq_ys.p);
            LR_LAUNCH_CHECK();
        }
        if (rc) return rc;
        if (D > 1) index = index + n_out * D - n;       // downsampler.lua:53
        return (long)n_out;
    }
};

// =====================================================================================================
// AGCBlock
// =====================================================================================================
struct AgcStage : lrhip_stage {
    AgcParams p;
    int S = 1;
    bool squelch = false;                              // PowerSquelchBlock: power scan + gate only
    DeviceBuf state, mapsP, mapsG, startP, startG;     // state: two (P, G) double pairs, ping-pong
    int cur = 0;
    const char *kind() const override { return "agc"; }
    long memory() const override { return -1; }
    int reset() override { cur = 0; return zero_fill(state, 4 * sizeof(double)); }
    template <int SS>
    int go(const float *x, float *y, unsigned long n)
    {
        unsigned long nt = (n + AGC_TILE - 1) / AGC_TILE;
        if (mapsP.reserve(nt * 2 * sizeof(double)) || mapsG.reserve(nt * 2 * sizeof(double)) || startP.reserve(nt * sizeof(double)) ||
            startG.reserve(nt * sizeof(double))) return -1;
        double *st = (double *)state.p + 2 * cur, *st_out = (double *)state.p + 2 * (cur ^ 1);
        double *mp = (double *)mapsP.p, *mg = (double *)mapsG.p, *sp = (double *)startP.p, *sg = (double *)startG.p;
        dim3 g((unsigned)nt), b(256);
        hipLaunchKernelGGL((agc_pass_kernel<SS, 0>), g, b, 0, ctx().stream, x, y, n, p, mp, mg, (const double *)sp, (const double *)sg, st_out);
        hipLaunchKernelGGL(agc_carry_kernel, dim3(1), b, 0, ctx().stream, (const double *)mp, nt, (const double *)st, sp);
        if (squelch) {
            hipLaunchKernelGGL((agc_pass_kernel<SS, 3>), g, b, 0, ctx().stream, x, y, n, p, mp, mg, (const double *)sp, (const double *)sg, st_out);
            LR_LAUNCH_CHECK();
            cur ^= 1;
            return 0;
        }
        hipLaunchKernelGGL((agc_pass_kernel<SS, 1>), g, b, 0, ctx().stream, x, y, n, p, mp, mg, (const double *)sp, (const double *)sg, st_out);
        hipLaunchKernelGGL(agc_carry_kernel, dim3(1), b, 0, ctx().stream, (const double *)mg, nt, (const double *)(st + 1), sg);
        hipLaunchKernelGGL((agc_pass_kernel<SS, 2>), g, b, 0, ctx().stream, x, y, n, p, mp, mg, (const double *)sp, (const double *)sg, st_out);
        LR_LAUNCH_CHECK();
        cur ^= 1;
        return 0;
    }
    long run(const void *in_dev, unsigned long n, void *out_dev, unsigned long cap) override
    {
        if (n > cap) return set_error("agc: output capacity %lu < %lu", cap, n);
        if (!n) return 0;
        int rc = S == 2 ? go<2>((const float *)in_dev, (float *)out_dev, n) : go<1>((const float *)in_dev, (float *)out_dev, n);
        return rc ? rc : (long)n;
    }
};
