// ab_fft64.hip - stand-alone driver of fir_fft64_kernel (luaradio_amd/csrc/kernels_firfft64.h): builds in seconds instead of the library's two minutes, checks
// the kernel against a double-precision direct form at spread positions and times it with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I luaradio_amd/csrc -I include -o gpurun_scratch/ab_fft64 tools/ab_fft64.hip
//   ab_fft64 <log2n> <ntaps> [iters] [xcd_map]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "common.h"
#include "kernels_elem.h"
#include "kernels_fft.h"
#include "kernels_fir.h"
#include "kernels_firwin.h"
#include "kernels_firfft64.h"

using namespace lrhip;

#define CK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #c, hipGetErrorString(e_)); exit(1); } } while (0)

static void build_tables(const std::vector<float> &taps, std::vector<float> &tab)
{
    const double PI2 = 6.283185307179586476925286766559;
    const int ntaps = (int)taps.size();
    tab.assign((size_t)F64_TABLE_ELEMS * 2, 0.f);
    auto put = [&](size_t o, double a) { tab[2 * o] = (float)std::cos(a); tab[2 * o + 1] = (float)std::sin(a); };
    for (int c = 0; c < 16; c++)
        for (int t = 0; t < 64; t++) put((size_t)c * 64 + t, -PI2 * (double)((c * t) % 1024) / 1024.0);
    for (int d = 0; d < 4; d++)
        for (int t = 0; t < 64; t++) put((size_t)F64_TAB_D + d * 64 + t, -PI2 * (double)(t * d) / F4K_N);
    std::vector<double> cs(F4K_N), sn(F4K_N);
    for (int k = 0; k < F4K_N; k++) { cs[k] = std::cos(-PI2 * k / F4K_N); sn[k] = std::sin(-PI2 * k / F4K_N); }
    for (int r = 0; r < 64; r++)
        for (int l = 0; l < 64; l++) {
            const int k = 64 * f64_index(r) + l;
            double sr = 0, si = 0;
            for (int m = 0; m < ntaps; m++) {
                const int a = (int)(((long)k * m) % F4K_N);
                sr += taps[m] * cs[a];
                si += taps[m] * sn[a];
            }
            tab[2 * ((size_t)F64_TAB_H + r * 64 + l)] = (float)(sr / F4K_N);
            tab[2 * ((size_t)F64_TAB_H + r * 64 + l) + 1] = (float)(si / F4K_N);
            if (l <= 32) {
                const size_t o = (size_t)F64_TAB_HSYM + (size_t)f64_index(r) * F64_HSYM_ROW + l;
                tab[2 * o] = (float)(sr / F4K_N);
                tab[2 * o + 1] = (float)(si / F4K_N);
            }
        }
}

#ifndef AB_WAVES
#define AB_WAVES 4
#endif
constexpr int F64_WAVES = AB_WAVES;
template <int V>
static void launch(const float *hist, const float *x, const float2 *tables, float *y, int M, long n, int grid, int xcd)
{
    const size_t lds = (size_t)f64_lds_elems(F64_WAVES) * sizeof(float2);
    static bool set = false;
    if (!set) { CK(hipFuncSetAttribute((const void *)fir_fft64_kernel<V, F64_WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); set = true; }
    const long nblocks = (n + (F4K_N - V) - 1) / (F4K_N - V);
    hipLaunchKernelGGL((fir_fft64_kernel<V, F64_WAVES>), dim3(grid), dim3(64 * F64_WAVES), lds, 0, hist, x, tables, y, M, n, n, nblocks, (float *)nullptr, xcd, 0L, 0);
}

int main(int argc, char **argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 24, ntaps = argc > 2 ? atoi(argv[2]) : 1276, S = 2;
    const int iters = argc > 3 ? atoi(argv[3]) : 20, xcd = argc > 4 ? atoi(argv[4]) : 1;
    const long run = 8;
    const long n = 1L << log2n;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<float> taps(ntaps), xh((size_t)n * S), histh((size_t)(ntaps - 1) * S);
    double g = 0;
    for (auto &t : taps) { t = U(rng); g += std::fabs(t); }
    for (auto &t : taps) t = (float)(t / g);                 // sum |h| = 1: |y| <= 1
    for (auto &v : xh) v = U(rng);
    for (auto &v : histh) v = U(rng);
    std::vector<float> tab;
    build_tables(taps, tab);
    const int V = ((ntaps - 1 + 255) / 256) * 256 < 768 ? 768 : ((ntaps - 1 + 255) / 256) * 256;
    float *x, *y, *hist;
    float2 *tables;
    CK(hipMalloc(&x, xh.size() * 4)); CK(hipMalloc(&y, xh.size() * 4)); CK(hipMalloc(&hist, histh.size() * 4 + 16)); CK(hipMalloc(&tables, tab.size() * 4));
    CK(hipMemcpy(x, xh.data(), xh.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(hist, histh.data(), histh.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(tables, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    int dev = 0, cus = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const long nblocks = (n + (F4K_N - V) - 1) / (F4K_N - V), nslots = (nblocks + F64_WAVES - 1) / F64_WAVES;
    const int grid = (int)std::min<long>(nslots, cus);
    auto go = [&]() {
        if (V == 768) launch<768>(hist, x, tables, y, ntaps, n, grid, xcd);
        else if (V == 1024) launch<1024>(hist, x, tables, y, ntaps, n, grid, xcd);
        else launch<1280>(hist, x, tables, y, ntaps, n, grid, xcd);
    };
#ifdef LRHIP_F64_TRACE
    unsigned long long *trace;
    const size_t trace_n = (size_t)8 * 8 * 16 * 16;
    CK(hipMalloc(&trace, trace_n * 8));
    CK(hipMemset(trace, 0, trace_n * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(lrhip_f64_trace), &trace, sizeof(trace)));
#endif
    go();
    CK(hipDeviceSynchronize());
#ifdef LRHIP_F64_TRACE
    {
        for (int i = 0; i < 3; i++) go();
        CK(hipMemset(trace, 0, trace_n * 8));
        go();
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> tr(trace_n);
        CK(hipMemcpy(tr.data(), trace, trace_n * 8, hipMemcpyDeviceToHost));
        // stamps: 0 loop top, 1 window loaded (issued), 2 first dft64 done (waits for the loads), 3 big twiddle, 4 transpose (both planes), 5 second dft64, 6 x H, 7 idft64,
        // 8 transpose back, 9 conjugate twiddle, 10 idft64, 11 stores issued
        static const char *names[11] = {"load issue", "dft64 (+wait)", "twiddle", "transpose", "dft64", "x H", "idft64", "transpose back", "twiddle", "idft64", "store issue"};
        double sum[11] = {0}, blk = 0;
        int cnt = 0;
        for (int w = 0; w < 64; w++)
            for (int t = 1; t < 14; t++) {
                const unsigned long long *s = &tr[((size_t)w * 16 + t) * 16], *nx = s + 16;
                if (!s[0] || !s[11] || !nx[0]) continue;
                for (int i = 0; i < 11; i++) sum[i] += (double)(s[i + 1] - s[i]);
                blk += (double)(nx[0] - s[0]);
                cnt++;
            }
        if (cnt) {
            printf("f64 trace (%d blocks, %d waves per CU):", cnt, F64_WAVES);
            for (int i = 0; i < 11; i++) printf("  %s %.0f", names[i], sum[i] / cnt);
            printf("  | block %.0f clocks\n", blk / cnt);
        }
    }
#endif
    std::vector<float> yh(xh.size());
    CK(hipMemcpy(yh.data(), y, yh.size() * 4, hipMemcpyDeviceToHost));
    // check: 4096 positions spread over the vector (and the first / last 700) against the direct form in double
    double worst = 0;
    auto sample = [&](long p, int c) -> double {       // stream position p >= -(M-1)
        if (p >= 0) return xh[(size_t)p * S + c];
        const long h = p + (ntaps - 1);
        return h >= 0 ? histh[(size_t)h * S + c] : 0.0;
    };
    std::vector<long> pos;
    for (long i = 0; i < 700 && i < n; i++) { pos.push_back(i); pos.push_back(n - 1 - i); }
    for (int i = 0; i < 4096; i++) pos.push_back((long)((double)i / 4096 * (n - 1)));
    for (long i = 0; i < 64; i++) for (long r = 1; r < 8 && r * run * 512 + i - 32 < n; r++) pos.push_back(r * run * 512 + i - 32);      // run seams
    for (long q : pos)
        for (int c = 0; c < S; c++) {
            double acc = 0;
            for (int m = 0; m < ntaps; m++) acc += (double)taps[m] * sample(q - m, c);
            worst = std::max(worst, std::fabs(acc - (double)yh[(size_t)q * S + c]));
        }
    if (getenv("AB_DEBUG")) {
        const long Lh = F4K_N - V;
        double wb[6] = {0, 0, 0, 0, 0, 0};
        for (long q : pos) {
            double acc = 0;
            for (int m = 0; m < ntaps; m++) acc += (double)taps[m] * sample(q - m, 0);
            const long blk = q / Lh, nb = (n + Lh - 1) / Lh;
            const int cls = blk == 0 ? 0 : blk == 1 ? 1 : blk == nb - 1 ? 5 : blk == nb - 2 ? 4 : blk == nb - 3 ? 3 : 2;
            wb[cls] = std::max(wb[cls], std::fabs(acc - (double)yh[(size_t)q * S]));
        }
        int shown = 0;
        for (long q : pos) {
            double acc = 0;
            for (int m = 0; m < ntaps; m++) acc += (double)taps[m] * sample(q - m, 0);
            const double e = std::fabs(acc - (double)yh[(size_t)q * S]);
            if (e > 1e-6 && shown < 40) { printf("  bad q=%ld block=%ld window pos=%ld (row %ld lane %ld) err %.3g\n", q, q / Lh, q % Lh + V, (q % Lh + V) / 64, (q % Lh + V) % 64, e); shown++; }
        }
        printf("  max err by block: first %.3g second %.3g interior %.3g third-last %.3g second-last %.3g last %.3g\n", wb[0], wb[1], wb[2], wb[3], wb[4], wb[5]);
        for (long q : {0L, 1L, 2L, 3000L, 3001L, 100000L, 100001L}) {
            double acc = 0, acc1 = 0;
            for (int m = 0; m < ntaps; m++) { acc += (double)taps[m] * sample(q - m, 0); acc1 += (double)taps[m] * sample(q - m, 1); }
            printf("  y[%ld] = (%.6f, %.6f)   want (%.6f, %.6f)\n", q, yh[2 * q], yh[2 * q + 1], acc, acc1);
        }
        printf("  last error: %s\n", hipGetErrorString(hipGetLastError()));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; i++) go();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) go();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    printf("fft64 S=%d taps=%d V=%d n=2^%d run=%ld grid=%d: %.4f ms  %.1f GS/s  %.2f TB/s algorithmic (%.3f of 8 TB/s)  max err %.3g %s\n", S, ntaps, V, log2n, run, grid,
           ms, n / ms / 1e6, 8.0 * S * n / ms / 1e9, 8.0 * S * n / ms / 1e9 / 8.0, worst, worst < 1e-6 ? "OK" : "FAIL");
    return worst < 1e-6 ? 0 : 1;
}
