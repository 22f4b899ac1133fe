// lrhip.hip - liblrhip.so: the C ABI of include/lrhip.h over the CDNA4 kernels in kernels_*.h.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o luaradio_amd/liblrhip.so lrhip.hip
#include "../../include/lrhip.h"

#include <cmath>
#include <memory>

#include "common.h"
#ifndef LRHIP_FIR_D1_NACC
#define LRHIP_FIR_D1_NACC 8      /* accumulators per wave of the D = 1 Toeplitz kernel (A/B: 4 with more waves per SIMD) */
#endif
#include "kernels_elem.h"
#include "kernels_fft.h"
#include "kernels_fir.h"
#include "kernels_firfft.h"
#include "kernels_channelizer.h"
#include "kernels_iir.h"
#include "kernels_agc.h"

using namespace lrhip;

static int g_launches = 0;   // kernels enqueued since the counter was last cleared (chain diagnostics)
#define LR_LAUNCH_CHECK()                                                                              \
    do {                                                                                               \
        g_launches++;                                                                                  \
        hipError_t e__ = hipGetLastError();                                                            \
        if (e__ != hipSuccess) return set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

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

// =====================================================================================================
// FIRFilterBlock (+ fused FrequencyTranslatorBlock in front, + fused DownsamplerBlock behind)
// =====================================================================================================
struct FirStage : lrhip_stage {
    int M = 0, S = 2, taps_complex = 0;
    unsigned D = 1;
    bool use_fft = false;
    std::vector<float> taps_rev;          // host copy, reversed (firfilter.lua:234-238)
    DeviceBuf d_taps, d_atab;
    int ksteps = 0;                       // 0 => MFMA path unavailable for this (M, D)
    int mfma_blocks_per_cu = 0;           // resident workgroups of the persistent kernel (occupancy query, cached)
    int hist_pad = 0;                     // leading pad floats in the history buffers (1 for complex taps, see launch_mfma_cc)
    DeviceBuf hist[2];
    int cur = 0;
    unsigned long index = 0;              // carried downsampler index (downsampler.lua:53)
    bool rot = false;                     // fused rotator in front
    uint64_t rot_step = 0, count = 0;     // absolute index of the next input sample
    // overlap-save emission framing (firfilter.lua:451-485)
    long L = 0, fill = 0;
    DeviceBuf pending, work;
    // overlap-save ARITHMETIC (fused 1024-point FFT kernel); independent of the emission framing
    static constexpr int FFT_PART = 512;   // taps per overlap-save partition (V = 512, L = 512 of the 1024-point block)
    bool fft_arith = false;
    DeviceBuf d_fft_tables;
    int fft_blocks_per_cu = 0;
    // fused FrequencyDiscriminatorBlock in front (chains): input is ComplexFloat32, the filter runs on arg(c[i] conj c[i-1])/gain
    bool hist_in_kernel = false;          // set by a launch that also wrote the next history buffer
    bool pre_disc = false;
    // fused FrequencyDiscriminatorBlock behind the filter (chains): ComplexFloat32 in, Float32 out (persistent MFMA kernel epilogue)
    bool post_disc = false;
    DeviceBuf edge;
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
        cur = 0; index = 0; count = 0; fill = 0; disc_cur = 0;
        if ((pre_disc || post_disc) && zero_fill(disc_prev, 4 * sizeof(float))) return -1;
        size_t hb = ((size_t)(M > 1 ? M - 1 : 1) * S + hist_pad) * sizeof(float);
        if (zero_fill(hist[0], hb) || zero_fill(hist[1], hb)) return -1;
        return 0;
    }

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

    template <int SS, int DD, int NACC, int KS>
    int launch_mfma_ks(const float *x, long n, float *y, long n_out)
    {
        using G = FirMfmaGeom<SS, DD>;
        constexpr int TILE_OUT = G::tile_out(NACC);
        // alignment slack so that the tile's first staged sample is 16-B aligned in global memory
        if (((uintptr_t)x % (4 * SS)) != 0) {
            if (rot || post_disc) return set_error("fir: fused rotator / discriminator needs a sample-aligned input pointer");
            return launch_direct(x, n, y, n_out);
        }
        long sample_addr = (long)((uintptr_t)x / (4 * SS));
        int q = 4 / SS;
        long v = sample_addr + (long)index - (M - 1);
        int e = (int)(((v % q) + q) % q);
        int span = G::span(NACC, ksteps);
        size_t lds_floats = (size_t)fir_taps_len(DD, ksteps) + (size_t)G::phys(SS * span) + G::PAD + 8;
        size_t lds_bytes = lds_floats * sizeof(float);
        long ntiles = (n_out + TILE_OUT - 1) / TILE_OUT;
        const float *atab = (const float *)d_atab.p;          // zero-padded reversed taps
        const float *h = (const float *)hist[cur].p + hist_pad;
        int out_aligned = ((uintptr_t)y % 16) == 0;
        uint64_t rs = rot ? rot_step : 0, rc = rot ? count : 0;
        if constexpr (KS > 0) {
            auto launch = [&](auto kern) -> int {
                if (!mfma_blocks_per_cu && prepare_kernel(kern, lds_bytes, &mfma_blocks_per_cu)) return -1;     // queried once per stage
                long slots = (long)ctx().num_cus * mfma_blocks_per_cu;
                unsigned grid = (unsigned)(ntiles < slots ? ntiles : slots);
                if (post_disc && edge.reserve((size_t)ntiles * 8 * sizeof(float2))) return -1;
                float *ho = M > 1 ? (float *)hist[cur ^ 1].p + hist_pad : nullptr;
                hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, ctx().stream, h, x, atab, y, M, n, n_out, (long)index, e,
                                   ntiles, out_aligned, rs, rc, (float2 *)edge.p, post_disc ? (float2 *)disc_prev.p + (disc_cur ^ 1) : nullptr, 1.0 / disc_gain, ho);
                hist_in_kernel = ho != nullptr;
                return 0;
            };
            int rc2;
            if constexpr (SS == 2 && (DD == 1 || DD == 5)) {
                if (post_disc) {
                    rc2 = rot ? launch(fir_mfma_persistent_kernel<2, DD, NACC, true, KS, 1>) : launch(fir_mfma_persistent_kernel<2, DD, NACC, false, KS, 1>);
                    if (rc2) return rc2;
                    LR_LAUNCH_CHECK();
                    float2 *dp = (float2 *)disc_prev.p;
                    hipLaunchKernelGGL(fir_disc_fixup_kernel, dim3((unsigned)((4 * ntiles + 255) / 256)), dim3(256), 0, ctx().stream, (const float2 *)edge.p,
                                       4 * ntiles, TILE_OUT / 4, y, n_out, (const float2 *)(dp + disc_cur), 1.0 / disc_gain);
                    LR_LAUNCH_CHECK();
                    disc_cur ^= 1;
                    return 0;
                }
            }
            if (post_disc) return set_error("internal: discriminator epilogue without a persistent kernel variant");
            if constexpr (SS == 2) rc2 = rot ? launch(fir_mfma_persistent_kernel<2, DD, NACC, true, KS>) : launch(fir_mfma_persistent_kernel<2, DD, NACC, false, KS>);
            else rc2 = rot ? set_error("rotator fusion needs complex input") : launch(fir_mfma_persistent_kernel<1, DD, NACC, false, KS>);
            if (rc2) return rc2;
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

    int launch_fft(const float *x, long n, float *y, long n_out)
    {
        size_t lds_bytes = (size_t)FFT_LDS_ELEMS * sizeof(float2);
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
                unsigned grid = (unsigned)(want < slots ? want : slots);      // persistent; a dynamic one-batch-per-workgroup grid measured 3-7 % slower even for 2.4 blocks per wave
                const float2 *dp = pre_disc ? (const float2 *)disc_prev.p + disc_cur : nullptr;
                float *ho = (!pre_disc && M > 1 && part == 0) ? (float *)hist[cur ^ 1].p + hist_pad : nullptr;
                hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * FFT_WPB), lds_bytes, ctx().stream, h, x, tables, y, Mp, n, n_out, nblocks,
                                   1.0 / disc_gain, dp, ho, M, (long)part * FFT_PART, part > 0 ? 1 : 0);
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
    int launch_decim_lds(const float *x, long n, float *y, long n_out)
    {
        long ow = (DECIM_SPAN_MAX - M) / (long)D + 1;
        int OW = (int)(ow > 256 ? 256 : ow < 1 ? 1 : ow);
        long ntiles = (n_out + OW - 1) / OW;
        long span = (long)(OW - 1) * D + M;
        size_t lds_bytes = ((size_t)(((taps_complex ? 2 : 1) * M + 3) & ~3) + (size_t)S * (span + (span >> 5) + 2)) * sizeof(float);
        const float *h = (const float *)hist[cur].p + hist_pad;
        float *ho = M > 1 ? (float *)hist[cur ^ 1].p + hist_pad : nullptr;
        auto go = [&](auto kern) -> int {
            if (!decim_blocks_per_cu && prepare_kernel(kern, lds_bytes, &decim_blocks_per_cu)) return -1;
            long slots = (long)ctx().num_cus * decim_blocks_per_cu;
            unsigned grid = (unsigned)(ntiles < slots ? ntiles : slots);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, ctx().stream, h, x, (const float *)d_taps.p, y, M, n, n_out, (long)index, (long)D, OW,
                               ntiles, rot ? rot_step : (uint64_t)0, rot ? count : (uint64_t)0, ho);
            hist_in_kernel = ho != nullptr;
            return 0;
        };
        int rc = taps_complex ? go(fir_decim_lds_kernel<2, false, true>)
                 : S == 2 ? (rot ? go(fir_decim_lds_kernel<2, true>) : go(fir_decim_lds_kernel<2, false>))
                        : (rot ? set_error("rotator fusion needs complex input") : go(fir_decim_lds_kernel<1, false>));
        if (rc) return rc;
        LR_LAUNCH_CHECK();
        return 0;
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

    template <int SS>
    int dispatch_mfma(const float *x, long n, float *y, long n_out)
    {
        switch (D) {
            case 1: return launch_mfma<SS, 1, LRHIP_FIR_D1_NACC>(x, n, y, n_out);
            case 2: return launch_mfma<SS, 2, 4>(x, n, y, n_out);
            case 3: return launch_mfma<SS, 3, 2>(x, n, y, n_out);
            case 4: return launch_mfma<SS, 4, 2>(x, n, y, n_out);
            case 5: return launch_mfma<SS, 5, 2>(x, n, y, n_out);
            case 6: return launch_mfma<SS, 6, 1>(x, n, y, n_out);
            case 7: return launch_mfma<SS, 7, 1>(x, n, y, n_out);
            case 8: return launch_mfma<SS, 8, 1>(x, n, y, n_out);
            case 10: return launch_mfma<SS, 10, 1>(x, n, y, n_out);
            default: return decim_lds_ok() ? launch_decim_lds(x, n, y, n_out) : launch_direct(x, n, y, n_out);
        }
    }

    static bool mfma_supported_decim(unsigned d) { return (d >= 1 && d <= 8) || d == 10; }
    // the discriminator epilogue exists for the persistent instantiations of the complex-stream, real-taps kernel
    bool can_post_disc() const { return S == 2 && !taps_complex && !fft_arith && !use_fft && ((D == 1 && ksteps == 36) || (D == 5 && ksteps == 51)); }

    // filter n inputs (device), emit the retained outputs; advances history / index / count
    long core(const float *x, long n, float *y, unsigned long cap)
    {
        if (n <= 0) return 0;
        hist_in_kernel = false;
        long n_out = (unsigned long)n > index ? (long)((n - index + D - 1) / D) : 0;
        if ((unsigned long)n_out > cap) return set_error("fir: output capacity %lu < %ld", cap, n_out);
        if (n_out > 0) {
            int rc = fft_arith ? launch_fft(x, n, y, n_out)
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

    long run(const void *in_dev, unsigned long n_in, void *out_dev, unsigned long cap) override
    {
        const float *x = (const float *)in_dev;
        float *y = (float *)out_dev;
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

static FirStage *fir_build(const float *taps, unsigned ntaps, int taps_complex, int input_complex, unsigned decim,
                           int use_fft, bool rot, double omega)
{
    if (!taps || ntaps < 1) { set_error("fir: need at least one tap"); return nullptr; }
    if (taps_complex && !input_complex) { set_error("fir: complex taps require ComplexFloat32 input (firfilter.lua:69-74)"); return nullptr; }
    if (decim < 1) { set_error("fir: decimation must be >= 1"); return nullptr; }
    if (use_fft == 3) use_fft = (decim == 1 && !rot && ntaps >= 48 && ntaps <= 16 * FirStage::FFT_PART && (input_complex || !taps_complex)) ? 2 : 0;
    if (use_fft && decim != 1) { set_error("fir: overlap-save cannot be combined with decimation"); return nullptr; }
    if (use_fft < 0 || use_fft > 2) { set_error("fir: use_fft must be 0 (direct form), 1 (overlap-save as the reference: block emission), 2 (overlap-save arithmetic, sample-exact emission) or 3 (automatic)"); return nullptr; }
    if (ntaps > (1u << 20)) { set_error("fir: too many taps"); return nullptr; }
    if (ensure_init()) return nullptr;
    std::unique_ptr<FirStage> q(new (std::nothrow) FirStage());
    if (!q) { set_error("out of memory"); return nullptr; }
    q->M = (int)ntaps; q->S = input_complex ? 2 : 1; q->taps_complex = taps_complex; q->D = decim;
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
    if (rot) {
        if (!q->ksteps && !(input_complex && !taps_complex && (int)ntaps + 255 <= DECIM_SPAN_MAX)) {
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
    }
    if (q->use_fft) {
        long N = 1L << (long)std::floor(std::log(8.0 * ntaps) / std::log(2.0));   // firfilter.lua:329
        q->L = N - (long)ntaps + 1;
        if (q->pending.reserve((size_t)q->L * q->S * sizeof(float))) return nullptr;
    }
    if (q->reset()) return nullptr;
    return q.release();
}

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
    int reset() override { count = 0; return 0; }
    long run(const void *in_dev, unsigned long n, void *out_dev, unsigned long cap) override
    {
        if (n > cap) return set_error("rotator: output capacity %lu < %lu", cap, n);
        if (!n) return 0;
        unsigned grid = grid_for(n, 256, ctx().num_cus * 16);
        if ((((uintptr_t)in_dev | (uintptr_t)out_dev) & 15) == 0)
            hipLaunchKernelGGL(rotator_kernel<2>, dim3(grid), dim3(256), 0, ctx().stream, (const float2 *)in_dev, (float2 *)out_dev, n, step, count);
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
    int reset() override { index = 0; return 0; }
    unsigned long max_output(unsigned long n) const override { return n / factor + 1; }
    long run(const void *in_dev, unsigned long n, void *out_dev, unsigned long cap) override
    {
        unsigned long n_out = n > index ? (n - index + factor - 1) / factor : 0;   // downsampler.lua:46
        if (n_out > cap) return set_error("downsampler: output capacity %lu < %lu", cap, n_out);
        if (n_out) {
            unsigned grid = grid_for(n_out, 256, ctx().num_cus * 16);
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
    long run(const void *in_dev, unsigned long n, void *out_dev, unsigned long cap) override
    {
        if (n > cap) return set_error("fmdiscrim: output capacity %lu < %lu", cap, n);
        if (!n) return 0;
        unsigned grid = grid_for(n, 256, ctx().num_cus * 16);
        float2 *p = (float2 *)prev.p;
        hipLaunchKernelGGL(fmdiscrim_kernel, dim3(grid), dim3(256), 0, ctx().stream, (const float2 *)in_dev, (float *)out_dev, n, 1.0 / gain,
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
    DeviceBuf xhist[2], state[2], tile_end, tile_start, seq_xs, seq_ys;
    int cur = 0;
    unsigned long D = 1, index = 0;       // fused DownsamplerBlock behind the filter (chains)
    const char *kind() const override { return "iir"; }
    unsigned long max_output(unsigned long n) const override { return D == 1 ? n : n / D + 1; }
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
            unsigned grid = (unsigned)((ntiles + run - 1) / run);
            hipLaunchKernelGGL((iir_stream_kernel<SS, PP, NBT>), dim3(grid), dim3(256), 0, ctx().stream, x, y, n, xh, st, st_out, (long)D, (long)index, run,
                               warm_tiles, co, (float *)xhist[cur ^ 1].p, tp);
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
        return nb <= 2 ? run_scan_nb<SS, PP, 2>(x, y, n) : run_scan_nb<SS, PP, 16>(x, y, n);
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

// =====================================================================================================
// DFT / IDFT / PSD
// =====================================================================================================
struct FftStage : lrhip_stage {
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
                unsigned g = (unsigned)(want < slots ? want : slots);
                hipLaunchKernelGGL(kern, dim3(g), dim3(256), lds_bytes, ctx().stream, (const float *)in_dev, (float *)out_dev, nframes,
                                   (const float2 *)spec_tables.p, wp, mode, out_scale, shift);
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
                hipLaunchKernelGGL(welch_gather_kernel<T>, dim3(grid_for(nf * N, 256, ctx().num_cus * 16)), dim3(256), 0, ctx().stream, pend, P, x,
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

// =====================================================================================================
// IQFileSource / RealFileSource format conversion
// =====================================================================================================
struct FormatStage : lrhip_stage {
    int fmt = 0;          // index into kFormats
    int scalars = 1;      // raw scalars per sample (2 for I/Q)
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
    unsigned grid = grid_for(ns, 256, ctx().num_cus * 16);
    float *out = (float *)out_dev;
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
        unsigned grid = grid_for(n, 256, ctx().num_cus * 16);
        if (op == BIN_F2C) {
            hipLaunchKernelGGL(float_to_complex_kernel, dim3(grid), dim3(256), 0, ctx().stream, (const float *)a, (const float *)b, (float2 *)y, n);
            LR_LAUNCH_CHECK();
            return (long)n;
        }
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
        unsigned grid = grid_for(n, 256, ctx().num_cus * 16);
        const float *x = (const float *)in_dev;
        float *y = (float *)out_dev;
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
    unsigned long max_output(unsigned long n) const override { return n * factor; }
    long run(const void *in_dev, unsigned long n, void *out_dev, unsigned long cap) override
    {
        unsigned long n_out = n * factor;                 // upsampler.lua:46
        if (n_out > cap) return set_error("upsampler: output capacity %lu < %lu", cap, n_out);
        if (!n_out) return 0;
        unsigned grid = grid_for(n_out, 256, ctx().num_cus * 16);
        if (in_size == 8)
            hipLaunchKernelGGL(upsample_kernel<float2>, dim3(grid), dim3(256), 0, ctx().stream, (const float2 *)in_dev, (float2 *)out_dev, n_out, factor);
        else
            hipLaunchKernelGGL(upsample_kernel<float>, dim3(grid), dim3(256), 0, ctx().stream, (const float *)in_dev, (float *)out_dev, n_out, factor);
        LR_LAUNCH_CHECK();
        return (long)n_out;
    }
};

// =====================================================================================================
// polyphase rational resampler (chains: [MultiplyConstant] -> Upsampler -> FIR -> [Downsampler])
// =====================================================================================================
struct ResampleStage : lrhip_stage {
    int S = 2, M = 0, L = 1, HQ = 0;
    unsigned long D = 1;
    float c = 1.f;
    DeviceBuf d_taps, hist[2];
    int cur = 0;
    uint64_t Q0 = 0, m0 = 0;          // absolute input samples consumed / outputs emitted so far
    static constexpr int SPAN_MAX = 6144;
    const char *kind() const override { return "resample"; }
    unsigned long max_output(unsigned long n) const override { return (n * (unsigned long)L) / D + 2; }
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

// =====================================================================================================
// one-input element-wise blocks, DelayBlock, HilbertTransformBlock
// =====================================================================================================
struct UnaryStage : lrhip_stage {
    int op = 0;
    float cr = 0.f, ci = 0.f;
    const char *kind() const override { return "unary"; }
    int reset() override { return 0; }
    long run(const void *in_dev, unsigned long n, void *out_dev, unsigned long cap) override
    {
        if (n > cap) return set_error("unary: output capacity %lu < %lu", cap, n);
        if (!n) return 0;
        unsigned grid = grid_for(n, 256, ctx().num_cus * 16);
        const float *x = (const float *)in_dev;
        float *y = (float *)out_dev;
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
    int reset() override
    {
        cur = 0;
        return (zero_fill(state[0], D * in_size) || zero_fill(state[1], D * in_size)) ? -1 : 0;
    }
    long run(const void *in_dev, unsigned long n, void *out_dev, unsigned long cap) override
    {
        if (n > cap) return set_error("delay: output capacity %lu < %lu", cap, n);
        if (!n) return 0;
        unsigned grid = grid_for(n + D, 256, ctx().num_cus * 16);
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
    int reset() override { return fir->reset(); }
    long run(const void *in_dev, unsigned long n, void *out_dev, unsigned long cap) override
    {
        if (n > cap) return set_error("hilbert: output capacity %lu < %lu", cap, n);
        if (!n) return 0;
        if (tmp.reserve(n * sizeof(float))) return -1;
        long got = fir->core((const float *)in_dev, (long)n, (float *)tmp.p, n);
        if (got < 0) return got;
        // core() has swapped the ping-pong history: the history that was current for this chunk is the other one
        const float *old_hist = (const float *)fir->hist[fir->cur ^ 1].p;
        unsigned grid = grid_for(n, 256, ctx().num_cus * 16);
        hipLaunchKernelGGL(hilbert_combine_kernel, dim3(grid), dim3(256), 0, ctx().stream, old_hist, (const float *)in_dev, (const float *)tmp.p,
                           (float2 *)out_dev, n, fir->M);
        LR_LAUNCH_CHECK();
        return (long)n;
    }
};

// =====================================================================================================
// chain
// =====================================================================================================
struct lrhip_chain {
    struct Op {
        lrhip_stage *stage;
        bool owned;
    };
    std::vector<Op> ops;
    std::vector<std::unique_ptr<DeviceBuf>> edges;   // edges[i] = output of op i (all but the last)
    PinnedBuf h_in, h_out;
    DeviceBuf d_in, d_out;
    int last_launches = 0;
    // ---- pipelined ring (lrhip_chain_set_ring)
    struct Slot {
        PinnedBuf h_in, h_out;
        DeviceBuf d_in, d_out;
        hipEvent_t ev_in = nullptr, ev_done = nullptr, ev_out = nullptr;   // H2D done, kernels done, D2H done
        bool used = false;          // events have been recorded at least once
        long n_out = 0;
    };
    std::vector<std::unique_ptr<Slot>> ring;
    unsigned long ring_chunk = 0;
    unsigned head = 0, inflight = 0;       // next slot to submit into; chunks submitted and not collected
    hipStream_t s_in = nullptr, s_out = nullptr;
    ~lrhip_chain()
    {
        for (auto &sl : ring) {
            if (sl->ev_in) (void)hipEventDestroy(sl->ev_in);
            if (sl->ev_done) (void)hipEventDestroy(sl->ev_done);
            if (sl->ev_out) (void)hipEventDestroy(sl->ev_out);
        }
        if (s_in) (void)hipStreamDestroy(s_in);
        if (s_out) (void)hipStreamDestroy(s_out);
        for (auto &o : ops)
            if (o.owned) delete o.stage;
    }
};

// host-pointer path shared by stages and chains: pinned staging in, run, pinned staging out
template <typename Runner>
static long host_execute(PinnedBuf &h_in, PinnedBuf &h_out, DeviceBuf &d_in, DeviceBuf &d_out, int in_size, int out_size,
                         unsigned long max_out, const void *in_host, unsigned long n_in, void *out_host,
                         unsigned long out_capacity, Runner run)
{
    if (n_in && !in_host) return set_error("null input buffer");
    size_t in_bytes = (size_t)n_in * in_size;
    unsigned long cap = max_out < out_capacity ? max_out : out_capacity;
    if (max_out > out_capacity) return set_error("output capacity %lu < required %lu", out_capacity, max_out);
    if (max_out && !out_host) return set_error("null output buffer");
    if (h_in.reserve(in_bytes ? in_bytes : 16) || d_in.reserve(in_bytes ? in_bytes : 16)) return -1;
    if (h_out.reserve((size_t)cap * out_size + 16) || d_out.reserve((size_t)cap * out_size + 16)) return -1;
    if (in_bytes) {
        memcpy(h_in.p, in_host, in_bytes);
        LR_HIP(hipMemcpyAsync(d_in.p, h_in.p, in_bytes, hipMemcpyHostToDevice, ctx().stream));
    }
    long n_out = run(d_in.p, n_in, d_out.p, cap);
    if (n_out < 0) return n_out;
    if (n_out) LR_HIP(hipMemcpyAsync(h_out.p, d_out.p, (size_t)n_out * out_size, hipMemcpyDeviceToHost, ctx().stream));
    LR_HIP(hipStreamSynchronize(ctx().stream));
    if (n_out) memcpy(out_host, h_out.p, (size_t)n_out * out_size);
    return n_out;
}

// =====================================================================================================
// C ABI
// =====================================================================================================
extern "C" {

const char *lrhip_version(void) { return "lrhip 0.1 (gfx950)"; }
const char *lrhip_strerror(void) { return err_buf(); }

int lrhip_init(int device)
{
    err_buf()[0] = 0;
    return ensure_init(device);
}

int lrhip_device_count(void)
{
    int count = 0;
    LR_HIP(hipGetDeviceCount(&count));
    return count;
}

int lrhip_set_stream(void *hip_stream)
{
    if (ensure_init()) return -1;
    ctx().stream = hip_stream ? (hipStream_t)hip_stream : ctx().own_stream;
    return 0;
}

int lrhip_synchronize(void)
{
    if (ensure_init()) return -1;
    LR_HIP(hipStreamSynchronize(ctx().stream));
    return 0;
}

lrhip_stage_t *lrhip_fir_create(const float *taps, unsigned ntaps, int taps_complex, int input_complex, unsigned decim, int use_fft)
{
    return fir_build(taps, ntaps, taps_complex, input_complex, decim, use_fft, false, 0.0);
}

lrhip_stage_t *lrhip_rotator_create(double omega)
{
    if (!std::isfinite(omega)) { set_error("rotator: omega must be finite"); return nullptr; }
    if (ensure_init()) return nullptr;
    RotatorStage *q = new (std::nothrow) RotatorStage();
    if (!q) { set_error("out of memory"); return nullptr; }
    q->omega = omega;
    q->step = turns_fixed(omega);
    q->in_size = q->out_size = 8;
    return q;
}

lrhip_stage_t *lrhip_downsampler_create(unsigned factor, int elem_size)
{
    if (factor < 1) { set_error("downsampler: factor must be >= 1"); return nullptr; }
    if (elem_size != 4 && elem_size != 8) { set_error("downsampler: element size must be 4 or 8"); return nullptr; }
    if (ensure_init()) return nullptr;
    DownsamplerStage *q = new (std::nothrow) DownsamplerStage();
    if (!q) { set_error("out of memory"); return nullptr; }
    q->factor = factor;
    q->in_size = q->out_size = elem_size;
    return q;
}

lrhip_stage_t *lrhip_fmdiscrim_create(double gain)
{
    if (!(gain != 0.0) || !std::isfinite(gain)) { set_error("fmdiscrim: gain must be finite and non-zero"); return nullptr; }
    if (ensure_init()) return nullptr;
    std::unique_ptr<FmDiscrimStage> q(new (std::nothrow) FmDiscrimStage());
    if (!q) { set_error("out of memory"); return nullptr; }
    q->gain = gain;
    q->in_size = 8; q->out_size = 4;
    if (q->reset()) return nullptr;
    return q.release();
}

lrhip_stage_t *lrhip_iir_create(const float *b, unsigned nb, const float *a, unsigned na, int input_complex)
{
    if (!b || !a || nb < 1 || na < 1) { set_error("iir: need b_taps and at least one a_tap (iirfilter.lua:56)"); return nullptr; }
    if (nb > IIR_SEQ_MAX || na > IIR_SEQ_MAX) { set_error("iir: at most %d taps supported", IIR_SEQ_MAX); return nullptr; }
    if (a[0] == 0.0f) { set_error("iir: a[0] must be non-zero"); return nullptr; }
    if (ensure_init()) return nullptr;
    std::unique_ptr<IirStage> q(new (std::nothrow) IirStage());
    if (!q) { set_error("out of memory"); return nullptr; }
    q->S = input_complex ? 2 : 1;
    q->in_size = q->out_size = 4 * q->S;
    q->nb = (int)nb; q->na = (int)na; q->P = (int)na - 1;
    q->seq.nb = (int)nb; q->seq.na = (int)na;
    for (unsigned i = 0; i < nb; i++) q->seq.b[i] = b[i];
    for (unsigned i = 0; i < na; i++) q->seq.a[i] = a[i];
    q->scan = q->P >= 1 && q->P <= IIR_MAX_P && nb <= IIR_MAX_NB;
    if (q->scan) {
        int P = q->P;
        IirCoeffs &co = q->co;
        memset(&co, 0, sizeof(co));
        co.nb = (int)nb; co.P = P;
        for (unsigned i = 0; i < nb; i++) co.b[i] = (float)((double)b[i] / (double)a[0]);
        for (int i = 0; i < P; i++) co.a[i] = (float)((double)a[i + 1] / (double)a[0]);
        // companion matrix of the homogeneous recurrence for the state (y[n-1], ..., y[n-P])
        std::vector<double> A((size_t)P * P, 0.0), T;
        for (int k = 0; k < P; k++) A[k] = -(double)co.a[k];
        for (int r = 1; r < P; r++) A[r * P + r - 1] = 1.0;
        T = A;
        for (int s = 1; s < IIR_LC; s <<= 1) matmul(T, T, T, P);      // A^LC (LC is a power of two)
        std::vector<float> tpow((size_t)9 * P * P);
        std::vector<double> tpow64((size_t)9 * P * P);
        for (int k = 0; k <= 8; k++) {
            for (int i = 0; i < P * P; i++) { tpow[(size_t)k * P * P + i] = (float)T[i]; tpow64[(size_t)k * P * P + i] = T[i]; }
            if (k == 8) q->Ttile = T;
            matmul(T, T, T, P);
        }
        if (P > 4 ? upload(q->d_tpow, tpow64.data(), tpow64.size() * sizeof(double)) : upload(q->d_tpow, tpow.data(), tpow.size() * sizeof(float))) return nullptr;
        if (upload(q->d_ttile, q->Ttile.data(), q->Ttile.size() * sizeof(double))) return nullptr;
        // memory shorter than `w` tiles in Float32 terms?  (every entry of A^(w*TILE) underflows)  -> single-launch kernel
        static const bool force_3pass = getenv("LRHIP_IIR_3PASS") != nullptr;       // A/B knob
        std::vector<double> W = q->Ttile;
        for (int w = 1; w <= 4 && !force_3pass && !q->warm_tiles; w *= 2) {
            double mx = 0.0;
            bool finite = true;
            for (double v : W) { finite = finite && std::isfinite(v); mx = std::fabs(v) > mx ? std::fabs(v) : mx; }
            if (finite && mx < 1e-46) q->warm_tiles = w;
            matmul(W, W, W, P);
        }
    }
    if (q->reset()) return nullptr;
    return q.release();
}

lrhip_stage_t *lrhip_psd_create(unsigned n, const float *window, double scale, int logarithmic, int input_complex, int fftshift)
{
    if (!(scale > 0.0)) { set_error("psd: scale must be positive"); return nullptr; }
    FftStage *q = fft_build(n);
    if (!q) return nullptr;
    q->in_real = !input_complex;
    q->in_size = input_complex ? 8 : 4;
    q->out_size = 4;
    q->out_kind = logarithmic ? FFT_OUT_PSD_LOG : FFT_OUT_PSD;
    q->out_scale = (float)(1.0 / scale);
    q->shift = fftshift != 0;
    if (window) {
        if (upload(q->window, window, n * sizeof(float))) { delete q; return nullptr; }
        q->has_window = true;
    }
    return q;
}

lrhip_stage_t *lrhip_welch_create(unsigned n, const float *window, double scale, int logarithmic, int input_complex, unsigned overlap)
{
    if (overlap >= n) { set_error("welch: overlap %u must be smaller than the frame length %u", overlap, n); return nullptr; }
    std::unique_ptr<FftStage> psd((FftStage *)lrhip_psd_create(n, window, scale, logarithmic, input_complex, 1));
    if (!psd) return nullptr;
    std::unique_ptr<WelchStage> q(new (std::nothrow) WelchStage());
    if (!q) { set_error("out of memory"); return nullptr; }
    q->N = (int)n; q->hop = (int)(n - overlap);
    q->in_size = input_complex ? 8 : 4; q->out_size = 4;
    q->psd = std::move(psd);
    if (q->reset()) return nullptr;
    return q.release();
}

long lrhip_welch_read(lrhip_stage_t *q, float *avg_host, int reset)
{
    WelchStage *w = q ? dynamic_cast<WelchStage *>(q) : nullptr;
    if (!w) return set_error("welch_read: not a welch stage");
    long frames = w->count;
    if (frames > 0) {
        if (!avg_host) return set_error("null output buffer");
        size_t bytes = (size_t)w->N * sizeof(float);
        if (w->h_avg.reserve(bytes)) return -1;
        LR_HIP(hipMemcpyAsync(w->h_avg.p, w->sum.p, bytes, hipMemcpyDeviceToHost, ctx().stream));
        LR_HIP(hipStreamSynchronize(ctx().stream));
        const float *src = (const float *)w->h_avg.p;
        for (int i = 0; i < w->N; i++) avg_host[i] = src[i] / (float)frames;      // gnuplotspectrum.lua:179-181
    }
    if (reset && w->clear()) return -1;
    return frames;
}

lrhip_stage_t *lrhip_dft_create(unsigned n, int inverse, int real_side)
{
    FftStage *q = fft_build(n);
    if (!q) return nullptr;
    q->inverse = inverse != 0;
    if (!inverse) {
        q->in_real = real_side != 0;
        q->in_size = real_side ? 4 : 8;
        q->out_size = 8;
        q->out_kind = FFT_OUT_COMPLEX;
        q->out_scale = 1.0f;
    } else {
        q->in_real = 0;
        q->in_size = 8;
        q->out_size = real_side ? 4 : 8;
        q->out_kind = real_side ? FFT_OUT_REAL : FFT_OUT_COMPLEX;
        q->out_scale = (float)(1.0 / (double)n);      // spectrum_utils.lua:335-338
    }
    return q;
}

lrhip_stage_t *lrhip_format_convert_create(const char *format, int complex_out)
{
    if (!format) { set_error("format: missing format name"); return nullptr; }
    int idx = -1;
    for (size_t i = 0; i < sizeof(kFormats) / sizeof(kFormats[0]); i++)
        if (!strcmp(kFormats[i].name, format)) idx = (int)i;
    if (idx < 0) { set_error("Unsupported format (\"%s\")", format); return nullptr; }     // iqfile.lua:48
    if (ensure_init()) return nullptr;
    FormatStage *q = new (std::nothrow) FormatStage();
    if (!q) { set_error("out of memory"); return nullptr; }
    q->fmt = idx;
    q->scalars = complex_out ? 2 : 1;
    q->in_size = kFormats[idx].bytes * q->scalars;
    q->out_size = 4 * q->scalars;
    return q;
}

lrhip_stage_t *lrhip_multiply_constant_create(float re, float im, int constant_complex, int input_complex)
{
    if (constant_complex && !input_complex) { set_error("multiplyconstant: a complex constant takes ComplexFloat32 input only (multiplyconstant.lua:42-44)"); return nullptr; }
    if (ensure_init()) return nullptr;
    MulConstStage *q = new (std::nothrow) MulConstStage();
    if (!q) { set_error("out of memory"); return nullptr; }
    q->cr = re; q->ci = constant_complex ? im : 0.f;
    q->mode = !input_complex ? 0 : (constant_complex ? 2 : 1);
    q->in_size = q->out_size = input_complex ? 8 : 4;
    return q;
}

lrhip_stage_t *lrhip_upsampler_create(unsigned factor, int elem_size)
{
    if (factor < 1) { set_error("upsampler: factor must be >= 1"); return nullptr; }
    if (elem_size != 4 && elem_size != 8) { set_error("upsampler: element size must be 4 or 8"); return nullptr; }
    if (ensure_init()) return nullptr;
    UpsamplerStage *q = new (std::nothrow) UpsamplerStage();
    if (!q) { set_error("out of memory"); return nullptr; }
    q->factor = factor;
    q->in_size = q->out_size = elem_size;
    return q;
}

lrhip_stage_t *lrhip_agc_create(double power_alpha, double gain_alpha, double target, double threshold, int input_complex)
{
    if (!(power_alpha > 0.0 && power_alpha <= 1.0) || !(gain_alpha > 0.0 && gain_alpha <= 1.0)) { set_error("agc: alphas must be in (0, 1]"); return nullptr; }
    if (!(target > 0.0) || !(threshold >= 0.0)) { set_error("agc: target and threshold are linear powers (> 0)"); return nullptr; }
    if (ensure_init()) return nullptr;
    std::unique_ptr<AgcStage> q(new (std::nothrow) AgcStage());
    if (!q) { set_error("out of memory"); return nullptr; }
    q->p = AgcParams{power_alpha, gain_alpha, target, threshold};
    q->S = input_complex ? 2 : 1;
    q->in_size = q->out_size = 4 * q->S;
    if (q->reset()) return nullptr;
    return q.release();
}

lrhip_stage_t *lrhip_powersquelch_create(double alpha, double threshold, int input_complex)
{
    if (!(alpha > 0.0 && alpha <= 1.0)) { set_error("powersquelch: alpha must be in (0, 1]"); return nullptr; }
    AgcStage *q = (AgcStage *)lrhip_agc_create(alpha, 1.0, 1.0, threshold, input_complex);
    if (q) q->squelch = true;
    return q;
}

lrhip_stage_t *lrhip_fmmod_create(double modulation_index)
{
    if (!std::isfinite(modulation_index)) { set_error("fmmod: modulation index must be finite"); return nullptr; }
    if (ensure_init()) return nullptr;
    std::unique_ptr<FmModStage> q(new (std::nothrow) FmModStage());
    if (!q) { set_error("out of memory"); return nullptr; }
    q->k = modulation_index;
    q->in_size = 4; q->out_size = 8;
    if (q->reset()) return nullptr;
    return q.release();
}

lrhip_stage_t *lrhip_unary_create(const char *op, float re, float im, int constant_complex, int input_complex)
{
    if (!op) { set_error("unary: missing operation name"); return nullptr; }
    struct { const char *name; int code, in, out; } T[] = {
        {"complexmagnitude", UN_CMAG, 8, 4}, {"complexphase", UN_CPHASE, 8, 4}, {"complextoreal", UN_CREAL, 8, 4},
        {"complextoimag", UN_CIMAG, 8, 4},   {"complexconjugate", UN_CCONJ, 8, 8}, {"realtocomplex", UN_R2C, 4, 8},
        {"absolutevalue", UN_ABS, 4, 4}};
    int code = -1, in = 0, out = 0;
    for (auto &t : T)
        if (!strcmp(t.name, op)) { code = t.code; in = t.in; out = t.out; }
    if (!strcmp(op, "addconstant")) {
        if (constant_complex && !input_complex) { set_error("addconstant: a complex constant takes ComplexFloat32 input only (addconstant.lua:42-44)"); return nullptr; }
        code = !input_complex ? UN_ADDC_REAL : (constant_complex ? UN_ADDC_CPLX : UN_ADDC_CPLX_BY_REAL);
        in = out = input_complex ? 8 : 4;
    }
    if (code < 0) { set_error("unary: unknown operation \"%s\"", op); return nullptr; }
    if (ensure_init()) return nullptr;
    UnaryStage *q = new (std::nothrow) UnaryStage();
    if (!q) { set_error("out of memory"); return nullptr; }
    q->op = code; q->cr = re; q->ci = im;
    q->in_size = in; q->out_size = out;
    return q;
}

lrhip_stage_t *lrhip_delay_create(unsigned num_samples, int elem_size)
{
    if (num_samples < 1) { set_error("Number of samples must be greater than 0"); return nullptr; }      // delay.lua:28
    if (elem_size != 4 && elem_size != 8) { set_error("delay: element size must be 4 or 8"); return nullptr; }
    if (ensure_init()) return nullptr;
    std::unique_ptr<DelayStage> q(new (std::nothrow) DelayStage());
    if (!q) { set_error("out of memory"); return nullptr; }
    q->D = num_samples;
    q->in_size = q->out_size = elem_size;
    if (q->reset()) return nullptr;
    return q.release();
}

lrhip_stage_t *lrhip_hilbert_create(const float *taps, unsigned ntaps)
{
    if (!taps || (ntaps % 2) != 1) { set_error("Number of taps must be odd"); return nullptr; }          // hilberttransform.lua:29
    std::unique_ptr<HilbertStage> q(new (std::nothrow) HilbertStage());
    if (!q) { set_error("out of memory"); return nullptr; }
    q->fir.reset(fir_build(taps, ntaps, 0, 0, 1, 0, false, 0.0));
    if (!q->fir) return nullptr;
    q->in_size = 4; q->out_size = 8;
    return q.release();
}

lrhip_stage_t *lrhip_channelizer_create(const float *taps, unsigned ntaps, unsigned nchannels)
{
    if (!taps || ntaps < 32 || (ntaps % 32) != 0 || ntaps > 8192) { set_error("channelizer: ntaps must be a multiple of 32 in [32, 8192]"); return nullptr; }
    if (nchannels != 32 && nchannels != 64) { set_error("channelizer: nchannels must be 32 or 64"); return nullptr; }
    if (ensure_init()) return nullptr;
    std::unique_ptr<ChannelizerStage> q(new (std::nothrow) ChannelizerStage());
    if (!q) { set_error("out of memory"); return nullptr; }
    int M = (int)ntaps, K = (int)nchannels, K2 = 2 * K;
    q->M = M; q->K = K;
    q->in_size = q->out_size = 8;
    // W[2i][2c] = Re g, W[2i+1][2c] = -Im g, W[2i][2c+1] = Im g, W[2i+1][2c+1] = Re g,
    // g_c[i] = h[M-1-i] * exp(+j*2*pi*c*(M-1-i)/K)   (kernels_channelizer.h)
    std::vector<float> W((size_t)2 * M * K2);
    const double PI2 = 6.283185307179586476925286766559;
    for (int i = 0; i < M; i++)
        for (int c = 0; c < K; c++) {
            int j = M - 1 - i;
            double a = PI2 * (double)(((long)c * j) % K) / K, h = taps[j];
            float gr = (float)(h * std::cos(a)), gi = (float)(h * std::sin(a));
            W[(size_t)(2 * i) * K2 + 2 * c] = gr;
            W[(size_t)(2 * i + 1) * K2 + 2 * c] = -gi;
            W[(size_t)(2 * i) * K2 + 2 * c + 1] = gi;
            W[(size_t)(2 * i + 1) * K2 + 2 * c + 1] = gr;
        }
    if (upload(q->W, W.data(), W.size() * sizeof(float))) return nullptr;
    if (q->reset()) return nullptr;
    return q.release();
}

lrhip_stage_t *lrhip_binary_create(const char *op, int input_complex)
{
    if (!op) { set_error("binary: missing operation name"); return nullptr; }
    int code = !strcmp(op, "multiply") ? BIN_MULTIPLY : !strcmp(op, "multiplyconjugate") ? BIN_MULTIPLY_CONJ
             : !strcmp(op, "add") ? BIN_ADD : !strcmp(op, "subtract") ? BIN_SUBTRACT : !strcmp(op, "floattocomplex") ? BIN_F2C : -1;
    if (code < 0) { set_error("binary: unknown operation \"%s\"", op); return nullptr; }
    if (code == BIN_MULTIPLY_CONJ && !input_complex) { set_error("binary: multiplyconjugate takes ComplexFloat32 inputs (multiplyconjugate.lua:26)"); return nullptr; }
    if (ensure_init()) return nullptr;
    BinaryStage *q = new (std::nothrow) BinaryStage();
    if (!q) { set_error("out of memory"); return nullptr; }
    q->op = code;
    q->in_size = q->out_size = input_complex ? 8 : 4;
    if (code == BIN_F2C) { q->in_size = 4; q->out_size = 8; }
    return q;
}

void lrhip_stage_destroy(lrhip_stage_t *q)
{
    if (!q) return;
    if (ctx().ready) (void)hipStreamSynchronize(ctx().stream);
    delete q;
}

int lrhip_stage_reset(lrhip_stage_t *q) { return q ? q->reset() : set_error("null stage"); }
int lrhip_stage_input_size(const lrhip_stage_t *q) { return q ? q->in_size : set_error("null stage"); }
int lrhip_stage_output_size(const lrhip_stage_t *q) { return q ? q->out_size : set_error("null stage"); }
unsigned long lrhip_stage_max_output(const lrhip_stage_t *q, unsigned long n_in) { return q ? q->max_output(n_in) : 0; }

long lrhip_stage_execute_device(lrhip_stage_t *q, const void *in_dev, unsigned long n_in, void *out_dev, unsigned long out_capacity)
{
    if (!q) return set_error("null stage");
    if (n_in && (!in_dev || (!out_dev && q->max_output(n_in)))) return set_error("null buffer");
    return q->run(in_dev, n_in, out_dev, out_capacity);
}

long lrhip_stage_execute(lrhip_stage_t *q, const void *in_host, unsigned long n_in, void *out_host, unsigned long out_capacity)
{
    if (!q) return set_error("null stage");
    return host_execute(q->h_in, q->h_out, q->d_in, q->d_out, q->in_size, q->out_size, q->max_output(n_in), in_host, n_in,
                        out_host, out_capacity,
                        [&](const void *di, unsigned long n, void *dout, unsigned long cap) { return q->run(di, n, dout, cap); });
}

long lrhip_stage_execute2_device(lrhip_stage_t *q, const void *in1_dev, const void *in2_dev, unsigned long n_in, void *out_dev,
                                 unsigned long out_capacity)
{
    if (!q) return set_error("null stage");
    if (n_in && (!in1_dev || !in2_dev || !out_dev)) return set_error("null buffer");
    return q->run2(in1_dev, in2_dev, n_in, out_dev, out_capacity);
}

long lrhip_stage_execute2(lrhip_stage_t *q, const void *in1_host, const void *in2_host, unsigned long n_in, void *out_host,
                          unsigned long out_capacity)
{
    if (!q) return set_error("null stage");
    BinaryStage *b = dynamic_cast<BinaryStage *>(q);
    if (!b) return set_error("%s is not a two-input stage", q->kind());
    if (n_in && !in2_host) return set_error("null input buffer");
    size_t bytes = (size_t)n_in * q->in_size;
    if (b->h_in2.reserve(bytes ? bytes : 16) || b->d_in2.reserve(bytes ? bytes : 16)) return -1;
    if (bytes) {
        memcpy(b->h_in2.p, in2_host, bytes);
        LR_HIP(hipMemcpyAsync(b->d_in2.p, b->h_in2.p, bytes, hipMemcpyHostToDevice, ctx().stream));
    }
    return host_execute(q->h_in, q->h_out, q->d_in, q->d_out, q->in_size, q->out_size, n_in, in1_host, n_in, out_host, out_capacity,
                        [&](const void *di, unsigned long n, void *dout, unsigned long cap) { return q->run2(di, b->d_in2.p, n, dout, cap); });
}

// ---- chains ---------------------------------------------------------------------------------------------
lrhip_chain_t *lrhip_chain_create(lrhip_stage_t **stages, unsigned nstages)
{
    if (!stages || nstages < 1) { set_error("chain: need at least one stage"); return nullptr; }
    for (unsigned i = 0; i < nstages; i++)
        if (!stages[i]) { set_error("chain: stage %u is null", i); return nullptr; }
    for (unsigned i = 0; i + 1 < nstages; i++)
        if (stages[i]->out_size != stages[i + 1]->in_size) {
            set_error("chain: stage %u (%s) emits %d-byte samples but stage %u (%s) takes %d-byte samples", i, stages[i]->kind(),
                      stages[i]->out_size, i + 1, stages[i + 1]->kind(), stages[i + 1]->in_size);
            return nullptr;
        }
    std::unique_ptr<lrhip_chain> c(new (std::nothrow) lrhip_chain());
    if (!c) { set_error("out of memory"); return nullptr; }
    unsigned i = 0;
    while (i < nstages) {
        // fusion: [multiplyconstant(real)] upsampler fir(real taps, plain) [downsampler]  ->  one polyphase resampling launch
        {
            static const bool no_resample_fusion = getenv("LRHIP_NO_RESAMPLE_FUSION") != nullptr;      // A/B knob
            unsigned k = i;
            MulConstStage *mc = dynamic_cast<MulConstStage *>(stages[k]);
            if (mc && mc->mode <= 1) k++; else mc = nullptr;
            UpsamplerStage *up = k < nstages ? dynamic_cast<UpsamplerStage *>(stages[k]) : nullptr;
            FirStage *rf = (up && k + 1 < nstages) ? dynamic_cast<FirStage *>(stages[k + 1]) : nullptr;
            if (!no_resample_fusion && up && rf && !rf->taps_complex && !rf->use_fft && !rf->fft_arith && rf->D == 1 && !rf->rot && !rf->pre_disc) {
                DownsamplerStage *rd = k + 2 < nstages ? dynamic_cast<DownsamplerStage *>(stages[k + 2]) : nullptr;
                unsigned long D = rd ? rd->factor : 1;
                int L = (int)up->factor;
                if (ResampleStage::fits(rf->M, L, D)) {
                    std::unique_ptr<ResampleStage> q(new (std::nothrow) ResampleStage());
                    if (!q) { set_error("out of memory"); return nullptr; }
                    q->S = rf->S; q->M = rf->M; q->L = L; q->D = D;
                    q->HQ = (rf->M - 1) / L + 1;
                    q->c = mc ? mc->cr : 1.f;
                    q->in_size = q->out_size = rf->S * 4;
                    std::vector<float> taps((size_t)rf->M);
                    for (int t = 0; t < rf->M; t++) taps[t] = rf->taps_rev[rf->M - 1 - t];
                    if (upload(q->d_taps, taps.data(), taps.size() * sizeof(float)) || q->reset()) return nullptr;
                    c->ops.push_back({q.release(), true});
                    i = k + 2 + (rd ? 1 : 0);
                    continue;
                }
            }
        }
        // fusion: [rotator] fir(real taps, plain) [downsampler]  ->  one decimating Toeplitz-MFMA launch
        RotatorStage *rot = dynamic_cast<RotatorStage *>(stages[i]);
        unsigned j = rot ? i + 1 : i;
        FirStage *fir = j < nstages ? dynamic_cast<FirStage *>(stages[j]) : nullptr;
        bool fusable_fir = fir && !fir->use_fft && fir->D == 1 && !fir->rot;
        if (rot && fusable_fir && fir->taps_complex) {      // rotator folding is implemented for real taps only
            c->ops.push_back({stages[i], false});
            i++;
            continue;
        }
        DownsamplerStage *ds = (fusable_fir && j + 1 < nstages) ? dynamic_cast<DownsamplerStage *>(stages[j + 1]) : nullptr;
        // ... [discriminator]: runs as the epilogue of the persistent kernel (ComplexFloat32 outputs never reach HBM)
        static const bool no_disc_fusion = getenv("LRHIP_NO_DISC_FUSION") != nullptr;      // A/B knob
        unsigned after = j + 1 + (ds ? 1 : 0);
        FmDiscrimStage *dsc_after = (!no_disc_fusion && fusable_fir && fir->S == 2 && !fir->taps_complex && !fir->fft_arith && after < nstages)
                                        ? dynamic_cast<FmDiscrimStage *>(stages[after]) : nullptr;
        if (fusable_fir && (rot || ds || dsc_after)) {
            unsigned D = ds ? (unsigned)ds->factor : 1;
            int ts = fir->taps_complex ? 2 : 1;
            std::vector<float> taps((size_t)fir->M * ts);
            for (int t = 0; t < fir->M; t++)
                for (int cc = 0; cc < ts; cc++) taps[(size_t)t * ts + cc] = fir->taps_rev[(size_t)(fir->M - 1 - t) * ts + cc];
            bool want_rot = rot && fir->S == 2;
            FirStage *fused = nullptr;
            if (FirStage::mfma_supported_decim(D) || !rot || (fir->S == 2 && !fir->taps_complex && fir->M + 255 <= DECIM_SPAN_MAX))
                fused = fir_build(taps.data(), (unsigned)fir->M, fir->taps_complex, fir->S == 2, D, 0, want_rot, want_rot ? rot->omega : 0.0);
            if (fused && rot && !want_rot) { delete fused; fused = nullptr; }
            bool with_disc = fused && dsc_after && fused->can_post_disc();
            if (fused && !rot && !ds && !with_disc) { delete fused; fused = nullptr; }      // nothing was fused
            if (fused) {
                if (with_disc) {
                    fused->post_disc = true;
                    fused->disc_gain = dsc_after->gain;
                    fused->out_size = 4;                // ComplexFloat32 in, Float32 out
                    if (fused->reset()) { delete fused; return nullptr; }
                }
                c->ops.push_back({fused, true});
                i = j + (ds ? 2 : 1) + (with_disc ? 1 : 0);
                continue;
            }
        }
        // fusion: discriminator -> overlap-save FIR on the real stream: the discriminator runs in the FFT kernel's load stage
        {
            FmDiscrimStage *dsc = dynamic_cast<FmDiscrimStage *>(stages[i]);
            FirStage *f1 = (dsc && i + 1 < nstages) ? dynamic_cast<FirStage *>(stages[i + 1]) : nullptr;
            if (!no_disc_fusion && f1 && f1->fft_arith && f1->M <= FirStage::FFT_PART && !f1->use_fft && f1->S == 1 && f1->D == 1 && !f1->rot && !f1->pre_disc) {
                std::vector<float> taps((size_t)f1->M);
                for (int t = 0; t < f1->M; t++) taps[t] = f1->taps_rev[f1->M - 1 - t];
                FirStage *fused = fir_build(taps.data(), (unsigned)f1->M, 0, 0, 1, 2, false, 0.0);
                if (fused) {
                    fused->pre_disc = true;
                    fused->disc_gain = dsc->gain;
                    fused->in_size = 8;                 // ComplexFloat32 in, Float32 out
                    if (fused->reset()) { delete fused; return nullptr; }
                    c->ops.push_back({fused, true});
                    i += 2;
                    continue;
                }
            }
        }
        // fusion: iir (scan path) -> downsampler: the final scan pass stores only the kept samples
        {
            IirStage *iir = dynamic_cast<IirStage *>(stages[i]);
            DownsamplerStage *ds2 = (iir && iir->scan && iir->D == 1 && i + 1 < nstages) ? dynamic_cast<DownsamplerStage *>(stages[i + 1]) : nullptr;
            if (ds2 && ds2->factor > 1) {
                IirStage *f = (IirStage *)lrhip_iir_create(iir->seq.b, (unsigned)iir->nb, iir->seq.a, (unsigned)iir->na, iir->S == 2);
                if (f) {
                    f->D = ds2->factor;
                    c->ops.push_back({f, true});
                    i += 2;
                    continue;
                }
            }
        }
        c->ops.push_back({stages[i], false});
        i++;
    }
    for (size_t k = 0; k + 1 < c->ops.size(); k++) c->edges.emplace_back(new DeviceBuf());
    return c.release();
}

void lrhip_chain_destroy(lrhip_chain_t *c)
{
    if (!c) return;
    if (ctx().ready) (void)hipStreamSynchronize(ctx().stream);
    delete c;
}

unsigned long lrhip_chain_max_output(const lrhip_chain_t *c, unsigned long n_in)
{
    if (!c) return 0;
    unsigned long n = n_in;
    for (auto &o : c->ops) n = o.stage->max_output(n);
    return n;
}

long lrhip_chain_execute_device(lrhip_chain_t *c, const void *in_dev, unsigned long n_in, void *out_dev, unsigned long out_capacity)
{
    if (!c) return set_error("null chain");
    g_launches = 0;
    const void *cur = in_dev;
    unsigned long n = n_in;
    for (size_t k = 0; k < c->ops.size(); k++) {
        lrhip_stage *s = c->ops[k].stage;
        bool last = k + 1 == c->ops.size();
        unsigned long need = s->max_output(n);
        void *dst;
        unsigned long cap;
        if (last) {
            dst = out_dev;
            cap = out_capacity;
        } else {
            if (c->edges[k]->reserve((size_t)need * s->out_size + 16)) return -1;
            dst = c->edges[k]->p;
            cap = need;
        }
        long got = s->run(cur, n, dst, cap);
        if (got < 0) return got;
        cur = dst;
        n = (unsigned long)got;
    }
    c->last_launches = g_launches;
    return (long)n;
}

long lrhip_chain_execute(lrhip_chain_t *c, const void *in_host, unsigned long n_in, void *out_host, unsigned long out_capacity)
{
    if (!c) return set_error("null chain");
    return host_execute(c->h_in, c->h_out, c->d_in, c->d_out, c->ops.front().stage->in_size, c->ops.back().stage->out_size,
                        lrhip_chain_max_output(c, n_in), in_host, n_in, out_host, out_capacity,
                        [&](const void *di, unsigned long n, void *dout, unsigned long cap) { return lrhip_chain_execute_device(c, di, n, dout, cap); });
}

int lrhip_chain_last_launches(const lrhip_chain_t *c) { return c ? c->last_launches : set_error("null chain"); }

int lrhip_chain_in_flight(const lrhip_chain_t *c) { return c ? (int)c->inflight : set_error("null chain"); }

int lrhip_chain_set_ring(lrhip_chain_t *c, unsigned depth, unsigned long max_chunk)
{
    if (!c) return set_error("null chain");
    if (depth < 1 || depth > 16) return set_error("ring depth must be 1..16");
    if (max_chunk < 1) return set_error("ring: max_chunk must be >= 1");
    if (c->inflight) return set_error("ring: %u chunks still in flight", c->inflight);
    if (ensure_init()) return -1;
    if (!c->s_in) LR_HIP(hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking));
    if (!c->s_out) LR_HIP(hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking));
    LR_HIP(hipStreamSynchronize(ctx().stream));
    c->ring.clear();
    int in_size = c->ops.front().stage->in_size, out_size = c->ops.back().stage->out_size;
    // bound on the output of one chunk whatever the carried state (rate-changing stages may emit one extra sample)
    unsigned long max_out = lrhip_chain_max_output(c, max_chunk) + 64;
    if (max_out < max_chunk + 64) max_out = max_chunk + 64;
    for (unsigned i = 0; i < depth; i++) {
        std::unique_ptr<lrhip_chain::Slot> sl(new (std::nothrow) lrhip_chain::Slot());
        if (!sl) return set_error("out of memory");
        if (sl->h_in.reserve((size_t)max_chunk * in_size) || sl->d_in.reserve((size_t)max_chunk * in_size)) return -1;
        if (sl->h_out.reserve((size_t)max_out * out_size) || sl->d_out.reserve((size_t)max_out * out_size)) return -1;
        LR_HIP(hipEventCreateWithFlags(&sl->ev_in, hipEventDisableTiming));
        LR_HIP(hipEventCreateWithFlags(&sl->ev_done, hipEventDisableTiming));
        LR_HIP(hipEventCreateWithFlags(&sl->ev_out, hipEventDisableTiming));
        c->ring.push_back(std::move(sl));
    }
    // size the device-resident edges once, so no reallocation happens while chunks are in flight
    unsigned long nmax = max_chunk;
    for (size_t k = 0; k + 1 < c->ops.size(); k++) {
        unsigned long grow = c->ops[k].stage->max_output(nmax);
        nmax = (grow > nmax ? grow : nmax) + 64;
        if (c->edges[k]->reserve((size_t)nmax * c->ops[k].stage->out_size + 16)) return -1;
    }
    c->ring_chunk = max_chunk;
    c->head = 0;
    c->inflight = 0;
    return 0;
}

long lrhip_chain_submit(lrhip_chain_t *c, const void *in_host, unsigned long n_in)
{
    if (!c) return set_error("null chain");
    if (c->ring.empty()) return set_error("chain has no ring: call lrhip_chain_set_ring first");
    if (n_in > c->ring_chunk) return set_error("chunk of %lu samples exceeds the ring's max_chunk %lu", n_in, c->ring_chunk);
    if (n_in && !in_host) return set_error("null input buffer");
    if (c->inflight == c->ring.size()) return set_error("ring full: collect a chunk first (%u in flight)", c->inflight);
    lrhip_chain::Slot &sl = *c->ring[c->head];
    int in_size = c->ops.front().stage->in_size, out_size = c->ops.back().stage->out_size;
    size_t bytes = (size_t)n_in * in_size;
    // the slot was collected (ev_out waited) before it can be reused, so its buffers are free on host and device
    if (bytes) {
        memcpy(sl.h_in.p, in_host, bytes);
        LR_HIP(hipMemcpyAsync(sl.d_in.p, sl.h_in.p, bytes, hipMemcpyHostToDevice, c->s_in));
    }
    LR_HIP(hipEventRecord(sl.ev_in, c->s_in));
    LR_HIP(hipStreamWaitEvent(ctx().stream, sl.ev_in, 0));
    unsigned long cap = (unsigned long)(sl.d_out.cap / out_size);
    long n_out = lrhip_chain_execute_device(c, sl.d_in.p, n_in, sl.d_out.p, cap);
    if (n_out < 0) return n_out;
    LR_HIP(hipEventRecord(sl.ev_done, ctx().stream));
    LR_HIP(hipStreamWaitEvent(c->s_out, sl.ev_done, 0));
    if (n_out) LR_HIP(hipMemcpyAsync(sl.h_out.p, sl.d_out.p, (size_t)n_out * out_size, hipMemcpyDeviceToHost, c->s_out));
    LR_HIP(hipEventRecord(sl.ev_out, c->s_out));
    sl.n_out = n_out;
    sl.used = true;
    c->head = (c->head + 1) % c->ring.size();
    c->inflight++;
    return n_out;
}

long lrhip_chain_collect(lrhip_chain_t *c, void *out_host, unsigned long out_capacity)
{
    if (!c) return set_error("null chain");
    if (!c->inflight) { set_error("nothing in flight"); return -2; }
    unsigned tail = (c->head + (unsigned)c->ring.size() - c->inflight) % c->ring.size();
    lrhip_chain::Slot &sl = *c->ring[tail];
    if ((unsigned long)sl.n_out > out_capacity) return set_error("output capacity %lu < %ld", out_capacity, sl.n_out);
    if (sl.n_out && !out_host) return set_error("null output buffer");
    LR_HIP(hipEventSynchronize(sl.ev_out));
    if (sl.n_out) memcpy(out_host, sl.h_out.p, (size_t)sl.n_out * c->ops.back().stage->out_size);
    c->inflight--;
    return sl.n_out;
}

// ---- memory helpers ----------------------------------------------------------------------------------------
void *lrhip_malloc(unsigned long bytes)
{
    if (ensure_init()) return nullptr;
    void *p = nullptr;
    LR_HIP_NULL(hipMalloc(&p, bytes ? bytes : 4));
    return p;
}
void lrhip_free(void *p)
{
    if (p) (void)hipFree(p);
}
int lrhip_memcpy_h2d(void *dst, const void *src, unsigned long bytes)
{
    if (ensure_init()) return -1;
    LR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx().stream));
    LR_HIP(hipStreamSynchronize(ctx().stream));
    return 0;
}
int lrhip_memcpy_d2h(void *dst, const void *src, unsigned long bytes)
{
    if (ensure_init()) return -1;
    LR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx().stream));
    LR_HIP(hipStreamSynchronize(ctx().stream));
    return 0;
}
int lrhip_memcpy_d2d(void *dst, const void *src, unsigned long bytes)
{
    if (ensure_init()) return -1;
    if (bytes) LR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx().stream));
    return 0;
}
void *lrhip_host_alloc(unsigned long bytes)
{
    if (ensure_init()) return nullptr;
    void *p = nullptr;
    LR_HIP_NULL(hipHostMalloc(&p, bytes ? bytes : 4, hipHostMallocDefault));
    return p;
}
void lrhip_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

// ---- timers ----------------------------------------------------------------------------------------------------
struct lrhip_timer {
    hipEvent_t a, b;
};
lrhip_timer_t *lrhip_timer_create(void)
{
    if (ensure_init()) return nullptr;
    lrhip_timer *t = new (std::nothrow) lrhip_timer();
    if (!t) { set_error("out of memory"); return nullptr; }
    if (hipEventCreate(&t->a) != hipSuccess || hipEventCreate(&t->b) != hipSuccess) {
        set_error("hipEventCreate failed");
        delete t;
        return nullptr;
    }
    return t;
}
void lrhip_timer_destroy(lrhip_timer_t *t)
{
    if (!t) return;
    (void)hipEventDestroy(t->a);
    (void)hipEventDestroy(t->b);
    delete t;
}
int lrhip_timer_start(lrhip_timer_t *t)
{
    if (!t) return set_error("null timer");
    LR_HIP(hipEventRecord(t->a, ctx().stream));
    return 0;
}
int lrhip_timer_stop(lrhip_timer_t *t)
{
    if (!t) return set_error("null timer");
    LR_HIP(hipEventRecord(t->b, ctx().stream));
    return 0;
}
double lrhip_timer_elapsed_ms(lrhip_timer_t *t)
{
    if (!t) return (double)set_error("null timer");
    if (hipEventSynchronize(t->b) != hipSuccess) return (double)set_error("hipEventSynchronize failed");
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t->a, t->b) != hipSuccess) return (double)set_error("hipEventElapsedTime failed");
    return (double)ms;
}

}  // extern "C"
