// ab_decim.hip - stand-alone driver of the LDS-staged decimator (fir_decim_lds_kernel, luaradio_amd/csrc/kernels_fir.h): builds in seconds, checks the
// kernel against a double-precision direct form at spread outputs, times it with HIP events and - built with -DLRHIP_DECIM_TRACE - prints where the
// cycles of a tile go (clock64 stamps by lane 0 of every wave of the first workgroups).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize [-DLRHIP_DECIM_TRACE] -I luaradio_amd/csrc -I include -o tools/ab_decim tools/ab_decim.hip
//   ab_decim <log2n> <D> <ntaps> <rot 0|1> [span_max] [iters]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "common.h"
#include "kernels_elem.h"
#include "kernels_fir.h"

using namespace lrhip;

#define CK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #c, hipGetErrorString(e_)); exit(1); } } while (0)

#ifdef AB_V2
#define AB_EXTRA , 0.0, (const float2 *)nullptr, (float2 *)nullptr, (getenv("AB_ONE_CHAIN") ? 1 : 0)
#include "kernels_firdecim.h"
#ifdef AB_PH
#define AB_KERNEL_ROT fir_decim_lds2_kernel<true, 0, true>
#define AB_KERNEL_PLAIN fir_decim_lds2_kernel<false, 0, true>
#else
#define AB_KERNEL_ROT fir_decim_lds2_kernel<true>
#define AB_KERNEL_PLAIN fir_decim_lds2_kernel<false>
#endif
#ifdef AB_PH
#define AB_LDS_BYTES(M, span) (((size_t)(((M) + 3) & ~3) + (size_t)2 * decim2_slots((int)(span), (long)D)) * sizeof(float))
#else
#define AB_LDS_BYTES(M, span) (((size_t)(((M) + 3) & ~3) + (size_t)2 * ((span) + DECIM2_PAD_SLOTS)) * sizeof(float))
#endif
#define AB_SPAN_MAX DECIM2_SPAN_MAX
#else
#define AB_SPAN_MAX DECIM_SPAN_MAX
#endif
#ifndef AB_EXTRA
#define AB_EXTRA
#endif
#ifndef AB_KERNEL_ROT
#define AB_KERNEL_ROT fir_decim_lds_kernel<2, true>
#define AB_KERNEL_PLAIN fir_decim_lds_kernel<2, false>
#define AB_LDS_BYTES(M, span) (((size_t)(((M) + 3) & ~3) + (size_t)2 * ((span) + ((span) >> 5) + 2)) * sizeof(float))
#endif

int main(int argc, char **argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 26, D = argc > 2 ? atoi(argv[2]) : 50, M = argc > 3 ? atoi(argv[3]) : 128, rot = argc > 4 ? atoi(argv[4]) : 1;
    const long span_req = argc > 5 ? atol(argv[5]) : 0;
    const int iters = argc > 6 ? atoi(argv[6]) : 20;
    const long n = 1L << log2n, n_out = (n + D - 1) / D;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<float> taps(M), trev(M), xh((size_t)n * 2), histh((size_t)(M - 1) * 2);
    double g = 0;
    for (auto &t : taps) { t = U(rng); g += std::fabs(t); }
    for (auto &t : taps) t = (float)(t / g);
    for (int i = 0; i < M; i++) trev[i] = taps[M - 1 - i];
    for (auto &v : xh) v = U(rng);
    for (auto &v : histh) v = U(rng);
    float *x, *y, *hist, *dt, *ho;
    CK(hipMalloc(&x, xh.size() * 4)); CK(hipMalloc(&y, (size_t)(n_out + 64) * 8)); CK(hipMalloc(&hist, histh.size() * 4 + 16)); CK(hipMalloc(&dt, M * 4 + 16));
    CK(hipMalloc(&ho, histh.size() * 4 + 16));
    CK(hipMemcpy(x, xh.data(), xh.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(hist, histh.data(), histh.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dt, trev.data(), M * 4, hipMemcpyHostToDevice));
    int dev = 0, cus = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const long span_max = span_req >= M + 64 && span_req <= AB_SPAN_MAX ? span_req : AB_SPAN_MAX;
    long ow = (span_max - M) / (long)D + 1;
    const int OW = (int)(ow > 256 ? 256 : ow < 1 ? 1 : ow);
    const long ntiles = (n_out + OW - 1) / OW, span = (long)(OW - 1) * D + M;
    const size_t lds_bytes = AB_LDS_BYTES(M, span);
    const uint64_t rot_step = rot ? (uint64_t)(-100e3 / 1102500.0 * 18446744073709551616.0) : 0, count0 = 12345678;
    const void *kern = rot ? (const void *)AB_KERNEL_ROT : (const void *)AB_KERNEL_PLAIN;
    CK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    int per_cu = 0;
    if (rot) CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, AB_KERNEL_ROT, 256, lds_bytes));
    else CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, AB_KERNEL_PLAIN, 256, lds_bytes));
    const long slots = (long)cus * per_cu;
    const unsigned grid = (unsigned)(ntiles < slots ? ntiles : slots);
    auto go = [&]() {
        if (rot) hipLaunchKernelGGL(AB_KERNEL_ROT, dim3(grid), dim3(256), lds_bytes, 0, hist, x, dt, y, M, n, n_out, 0L, (long)D, OW, ntiles, rot_step, count0, ho, 0, 0 AB_EXTRA);
        else hipLaunchKernelGGL(AB_KERNEL_PLAIN, dim3(grid), dim3(256), lds_bytes, 0, hist, x, dt, y, M, n, n_out, 0L, (long)D, OW, ntiles, (uint64_t)0, (uint64_t)0, ho, 0, 0 AB_EXTRA);
    };
#ifdef LRHIP_DECIM_TRACE
    unsigned long long *trace;
    const size_t trace_n = (size_t)8 * 4 * 32 * 8;
    CK(hipMalloc(&trace, trace_n * 8));
    CK(hipMemset(trace, 0, trace_n * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(lrhip_decim_trace), &trace, sizeof(trace)));
#endif
    go();
    CK(hipDeviceSynchronize());
    std::vector<float> yh((size_t)n_out * 2);
    CK(hipMemcpy(yh.data(), y, yh.size() * 4, hipMemcpyDeviceToHost));
    // check against the direct form in double: first / last outputs and 4096 spread ones
    const double PI2 = 6.283185307179586476925286766559;
    auto sample = [&](long p, int c) -> double {          // chunk index p >= -(M-1), rotated
        double re, im;
        if (p >= 0) { re = xh[(size_t)p * 2]; im = xh[(size_t)p * 2 + 1]; }
        else { re = histh[(size_t)(p + M - 1) * 2]; im = histh[(size_t)(p + M - 1) * 2 + 1]; }
        if (rot) {                                         // the carried history is the raw stream: rotated by its absolute index like the chunk
            const uint64_t turns = rot_step * (count0 + (uint64_t)p);
            const double a = PI2 * (double)turns / 18446744073709551616.0;
            const double cr = std::cos(a), ci = std::sin(a), r2 = re * cr - im * ci, i2 = re * ci + im * cr;
            re = r2; im = i2;
        }
        return c ? im : re;
    };
    std::vector<long> pos;
    for (long i = 0; i < 300 && i < n_out; i++) { pos.push_back(i); pos.push_back(n_out - 1 - i); }
    for (int i = 0; i < 4096; i++) pos.push_back((long)((double)i / 4096 * (n_out - 1)));
    double worst = 0, worst_head = 0;
    for (long k : pos)
        for (int c = 0; c < 2; c++) {
            double acc = 0;
            for (int m = 0; m < M; m++) {
                const long p = k * D - m;
                if (p < n) acc += (double)taps[m] * sample(p, c);
            }
            worst = std::max(worst, std::fabs(acc - (double)yh[(size_t)k * 2 + c]));
            if (k * D < M) worst_head = std::max(worst_head, std::fabs(acc - (double)yh[(size_t)k * 2 + c]));
        }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) go();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) go();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    const double bytes = (double)n * 8 + (double)n_out * 8;
    printf("D %d taps %d rot %d  2^%d: OW %d span %ld lds %zu B  %d wg/CU grid %u  max err %.3g (head %.3g)  %.4f ms  %.1f GS/s  %.0f GB/s  %.3f of 8 TB/s\n", D, M, rot, log2n, OW, span,
           lds_bytes, per_cu, grid, worst, worst_head, ms, n / ms * 1e-6, bytes / ms * 1e-6, bytes / ms * 1e-6 / 8000.0);
#ifdef LRHIP_DECIM_TRACE
    CK(hipMemset(trace, 0, trace_n * 8));
    go();
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> tr(trace_n);
    CK(hipMemcpy(tr.data(), trace, trace_n * 8, hipMemcpyDeviceToHost));
    // stamps: 0 loop top, 1 staged (before the barrier), 2 behind the barrier, 3 prefetch issued, 4 filtered, 5 behind the second barrier
    static const char *names[5] = {"stage", "barrier1", "prefetch", "filter", "barrier2"};
    for (int w = 0; w < 4; w++) {
        double sum[5] = {0, 0, 0, 0, 0}, tile = 0;
        int cnt = 0;
        for (int b = 0; b < 8; b++)
            for (int t = 2; t < 30; t++) {
                const unsigned long long *s = &tr[(((size_t)b * 4 + w) * 32 + t) * 8];
                if (!s[0] || !s[5]) continue;
                for (int i = 0; i < 5; i++) sum[i] += (double)(s[i + 1] - s[i]);
                const unsigned long long *nx = &tr[(((size_t)b * 4 + w) * 32 + t + 1) * 8];
                if (nx[0]) tile += (double)(nx[0] - s[0]);
                cnt++;
            }
        if (!cnt) continue;
        printf("  wave %d (%d tiles):", w, cnt);
        for (int i = 0; i < 5; i++) printf("  %s %.0f", names[i], sum[i] / cnt);
        printf("  | tile %.0f clocks\n", tile / cnt);
    }
#endif
    return 0;
}
