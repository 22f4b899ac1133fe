// ab_pols.hip - stand-alone driver of fir_pols_kernel (luaradio_amd/csrc/kernels_firpols.h): builds in seconds instead of the library's two minutes, checks
// the kernel against a double-precision direct form at spread positions and times it with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I luaradio_amd/csrc -I include -o gpurun_scratch/ab_pols tools/ab_pols.hip
//   ab_pols <log2n> <ntaps> <S: 2 cf32 | 1 f32> [run_blocks] [iters]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "common.h"
#include "kernels_elem.h"
#include "kernels_fft.h"
#include "kernels_firpols.h"

using namespace lrhip;

#define CK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #c, hipGetErrorString(e_)); exit(1); } } while (0)

static void build_tables(const std::vector<float> &taps, std::vector<float> &tab, int &nparts)
{
    const double PI2 = 6.283185307179586476925286766559;
    const int ntaps = (int)taps.size();
    nparts = (ntaps + 511) / 512;
    tab.assign((size_t)nparts * FFT_TABLE_ELEMS * 2, 0.f);
    for (int part = 0; part < nparts; part++) {
        float *tp = tab.data() + (size_t)part * FFT_TABLE_ELEMS * 2;
        const int m0 = part * 512, m1 = std::min(ntaps, m0 + 512);
        for (int k1 = 0; k1 < 16; k1++)
            for (int t = 0; t < 64; t++) {
                double a = -PI2 * (double)((k1 * t) % FFTN) / FFTN;
                tp[2 * (k1 * 64 + t)] = (float)std::cos(a);
                tp[2 * (k1 * 64 + t) + 1] = (float)std::sin(a);
            }
        std::vector<double> Hr(FFTN, 0.0), Hi(FFTN, 0.0);
        for (int k = 0; k < FFTN; k++) {
            double sr = 0, si = 0;
            for (int m = m0; m < m1; m++) {
                double a = -PI2 * (double)((k * (long)(m - m0)) % FFTN) / FFTN;
                sr += taps[m] * std::cos(a);
                si += taps[m] * std::sin(a);
            }
            Hr[k] = sr / FFTN;
            Hi[k] = si / FFTN;
        }
        for (int j = 0; j < 4; j++)
            for (int k3 = 0; k3 < 4; k3++)
                for (int lane = 0; lane < 64; lane++) {
                    int qq = lane & 3, k1 = lane >> 2, k = k1 + 16 * (4 * j + qq) + 256 * k3;
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
}

template <int S, int P>
static void launch(const float *hist, const float *x, const float2 *tables, float *y, int M, long n, long run, int part0, int acc, int grid)
{
    const size_t lds = (size_t)pols_lds_elems(S, P) * sizeof(float2);
    static bool set = false;
    if (!set) { CK(hipFuncSetAttribute((const void *)fir_pols_kernel<S, P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); set = true; }
    const long nblocks = (n + POLS_HOP - 1) / POLS_HOP;
    hipLaunchKernelGGL((fir_pols_kernel<S, P>), dim3(grid), dim3(64 * pols_wpb(S)), lds, 0, hist, x, tables, y, M, n, n, nblocks, run, part0, acc, (float *)nullptr);
}

template <int S>
static void filter(const float *hist, const float *x, const float2 *tables, float *y, int M, int nparts, long n, long run, int grid)
{
    for (int p0 = 0; p0 < nparts; p0 += 3) {
        const int P = std::min(3, nparts - p0);
        if (P == 3) launch<S, 3>(hist, x, tables, y, M, n, run, p0, p0 > 0, grid);
        else if (P == 2) launch<S, 2>(hist, x, tables, y, M, n, run, p0, p0 > 0, grid);
        else launch<S, 1>(hist, x, tables, y, M, n, run, p0, p0 > 0, grid);
    }
}

int main(int argc, char **argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 24, ntaps = argc > 2 ? atoi(argv[2]) : 1276, S = argc > 3 ? atoi(argv[3]) : 2;
    long run = argc > 4 ? atol(argv[4]) : 0;
    const int iters = argc > 5 ? atoi(argv[5]) : 20;
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
    int nparts = 0;
    build_tables(taps, tab, nparts);
    float *x, *y, *hist;
    float2 *tables;
    CK(hipMalloc(&x, xh.size() * 4)); CK(hipMalloc(&y, xh.size() * 4)); CK(hipMalloc(&hist, histh.size() * 4 + 16)); CK(hipMalloc(&tables, tab.size() * 4));
    CK(hipMemcpy(x, xh.data(), xh.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(hist, histh.data(), histh.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(tables, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    int dev = 0, cus = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const long nblocks = (n + POLS_HOP - 1) / POLS_HOP, runs_per_round = (long)cus * pols_wpb(S) * (S == 2 ? 1 : 2);
    if (run <= 0) {
        long k = (nblocks + runs_per_round * 40 - 1) / (runs_per_round * 40);
        if (k < 1) k = 1;
        run = (nblocks + runs_per_round * k - 1) / (runs_per_round * k);
        if (run < 4) run = 4;
    }
    const long nruns = (nblocks + run - 1) / run, nslots = (nruns + pols_wpb(S) * (S == 2 ? 1 : 2) - 1) / (pols_wpb(S) * (S == 2 ? 1 : 2));
    const int grid = (int)std::min<long>(nslots, cus);
    auto go = [&]() { if (S == 2) filter<2>(hist, x, tables, y, ntaps, nparts, n, run, grid); else filter<1>(hist, x, tables, y, ntaps, nparts, n, run, grid); };
    go();
    CK(hipDeviceSynchronize());
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
    printf("pols S=%d taps=%d parts=%d n=2^%d run=%ld grid=%d: %.4f ms  %.1f GS/s  %.2f TB/s algorithmic (%.3f of 8 TB/s)  max err %.3g %s\n", S, ntaps, nparts, log2n, run, grid,
           ms, n / ms / 1e6, 8.0 * S * n / ms / 1e9, 8.0 * S * n / ms / 1e9 / 8.0, worst, worst < 1e-6 ? "OK" : "FAIL");
    return worst < 1e-6 ? 0 : 1;
}
