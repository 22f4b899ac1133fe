// host_path_driver.cpp - C++ driver replaying the call sequence a LuaRadio DeviceChainBlock makes (lua/radio/composites/devicechain.lua):
// lrhip_init, stage constructors, lrhip_chain_create, lrhip_chain_set_ring, then lrhip_chain_push() per process() call with the
// reference's chunk sizes and lrhip_chain_flush() at cleanup - through the C ABI only (include/lrhip.h), no Python in the loop.
// Prints the PCIe-inclusive rate per chunk size for the WBFM-mono receiver fed with raw u8 I/Q records (2 bytes per complex sample
// cross PCIe) and with ComplexFloat32 samples.   Build: g++ -O2 -o build/host_path_driver tools/host_path_driver.cpp -Iinclude -Lluaradio_amd -llrhip
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "lrhip.h"

static std::vector<float> firwin_lowpass(int M, double cutoff)      // radio/utilities/filter_utils.lua:21-33, :121-157 (hamming)
{
    std::vector<double> h(M);
    double sum = 0;
    for (int n = 0; n < M; n++) {
        double m = n - (M - 1) / 2.0, s = m == 0 ? cutoff : std::sin(M_PI * cutoff * m) / (M_PI * m);
        double w = 0.54 - 0.46 * std::cos(2 * M_PI * n / (M - 1));
        h[n] = s * w; sum += h[n];
    }
    std::vector<float> f(M);
    for (int n = 0; n < M; n++) f[n] = (float)(h[n] / sum);
    return f;
}
#define CHK(p) do { if (!(p)) { fprintf(stderr, "%s: %s\n", #p, lrhip_strerror()); return 1; } } while (0)

int main()
{
    if (lrhip_init(0)) { fprintf(stderr, "%s\n", lrhip_strerror()); return 1; }
    const double fs = 1102500.0;
    const unsigned long total = 1ul << 25;
    std::vector<unsigned char> raw(2 * total);
    for (size_t i = 0; i < raw.size(); i++) raw[i] = (unsigned char)(rand() & 255);
    std::vector<float> cf(2 * total);
    for (size_t i = 0; i < cf.size(); i++) cf[i] = ((float)raw[i] - 127.5f) / 127.5f;
    for (int u8 = 1; u8 >= 0; u8--) {
        for (unsigned long chunk : {8192ul, 32768ul, 131072ul, 1048576ul}) {
            std::vector<float> t1 = firwin_lowpass(128, 100e3 / (fs / 2)), t2 = firwin_lowpass(128, 15e3 / (fs / 5 / 2));
            const double tau = 75e-6, r = fs / 5, wc = 1 / tau, wca = 2 * r * std::tan(wc / (2 * r)), taua = 1 / wca;   // singlepolelowpassfilter.lua:55-67
            float b[2] = {(float)(1 / (1 + 2 * taua * r)), (float)(1 / (1 + 2 * taua * r))}, a[2] = {1.f, (float)((1 - 2 * taua * r) / (1 + 2 * taua * r))};
            lrhip_stage_t *st[8];
            unsigned ns = 0;
            if (u8) CHK(st[ns++] = lrhip_format_convert_create("u8", 1));
            CHK(st[ns++] = lrhip_rotator_create(2 * M_PI * (-250e3 / fs)));
            CHK(st[ns++] = lrhip_fir_create(t1.data(), 128, 0, 1, 1, 0));
            CHK(st[ns++] = lrhip_downsampler_create(5, 8));
            CHK(st[ns++] = lrhip_fmdiscrim_create(2 * M_PI * 1.25));
            CHK(st[ns++] = lrhip_fir_create(t2.data(), 128, 0, 0, 1, 2));
            CHK(st[ns++] = lrhip_iir_create(b, 2, a, 2, 0));
            CHK(st[ns++] = lrhip_downsampler_create(5, 4));
            lrhip_chain_t *c;
            CHK(c = lrhip_chain_create(st, ns));
            if (lrhip_chain_set_ring(c, 3, 1ul << 20)) { fprintf(stderr, "%s\n", lrhip_strerror()); return 1; }
            unsigned long cap = lrhip_chain_push_bound(c, chunk);
            std::vector<float> out(cap);
            const char *src = u8 ? (const char *)raw.data() : (const char *)cf.data();
            const size_t rec = u8 ? 2 : 8;
            long n_out = 0;
            for (int pass = 0; pass < 2; pass++) {           // pass 0: warm-up (allocations, clocks)
                auto t0 = std::chrono::steady_clock::now();
                n_out = 0;
                for (unsigned long a0 = 0; a0 < total; a0 += chunk) {
                    long got = lrhip_chain_push(c, src + a0 * rec, chunk, out.data(), cap);
                    if (got < 0) { fprintf(stderr, "%s\n", lrhip_strerror()); return 1; }
                    n_out += got;
                }
                long got = lrhip_chain_flush(c, out.data(), cap);
                if (got < 0) { fprintf(stderr, "%s\n", lrhip_strerror()); return 1; }
                n_out += got;
                double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (pass) printf("{\"input\": \"%s\", \"chunk_samples\": %lu, \"MS/s\": %.1f, \"GB/s_h2d\": %.2f, \"audio_samples\": %ld, \"launches_per_batch\": %d}\n",
                                 u8 ? "u8 records" : "ComplexFloat32", chunk, total / dt / 1e6, rec * total / dt / 1e9, n_out, lrhip_chain_last_launches(c));
            }
            lrhip_chain_destroy(c);
            for (unsigned i = 0; i < ns; i++) lrhip_stage_destroy(st[i]);
        }
    }
    return 0;
}
