// host_path_driver.cpp - the host path of the library measured THROUGH C, no interpreter in the loop: what a LuaJIT host pays per call is
// what this driver pays.  It replays the call sequences of the Lua glue (lua/radio/composites/devicechain.lua, lua/radio/core/lrhip.lua) through include/lrhip.h:
//
//   file (page cache) -> WBFM-mono receiver chain, u8 records (2 B / sample on the link) and f32le records (8 B):
//     push      read(2) of 8 192 / 131 072 records into a user buffer + lrhip_chain_push per read     (a DeviceChainBlock fed by a pipe: pipe.lua:495-533)
//     ring      read(2) of 2^20 records straight into lrhip_chain_ring_input + lrhip_chain_submit      (IQFileSource absorbed, FIFO / device path)
//     fd        lrhip_chain_submit_fd: the library preads the records itself on its copy threads       (IQFileSource absorbed, regular file)
//   stand-alone LowpassFilterBlock(128) ComplexFloat32 -> ComplexFloat32, 2^20-sample vectors, lrhip_stage_execute: staged and with registered vectors; 2^22-sample vectors registered.
//
// One JSON object per line on stdout; bench.py folds them into its host_path leg.  Every leg's output is checksummed (sum of the output samples in double) and
// the legs of one input format must agree with each other to the chain's stated rounding - bench.py checks that; bit-level verification of the same paths is
// done by the Python legs of bench.py and by tests/.
// Build: g++ -O2 -std=c++17 -o tools/host_path_driver tools/host_path_driver.cpp -Iinclude -Lluaradio_amd -llrhip -Wl,-rpath,'$ORIGIN/../luaradio_amd'
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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
#define CHK(p) do { if (!(p)) { fprintf(stderr, "%s: %s\n", #p, lrhip_strerror()); exit(1); } } while (0)
#define CHK0(p) do { if ((p) != 0) { fprintf(stderr, "%s: %s\n", #p, lrhip_strerror()); exit(1); } } while (0)

static const double FS = 1102500.0;

struct Receiver {
    std::vector<lrhip_stage_t *> st;
    lrhip_chain_t *c = nullptr;
    Receiver(const char *format, unsigned long ring_chunk = 1ul << 20)          // IQFileSource(format) -> Tuner(-250e3, 200e3, 5) -> FrequencyDiscriminator(1.25) -> Lowpass(128, 15e3) -> FMDeemphasis(75e-6) -> Downsampler(5)
    {
        std::vector<float> t1 = firwin_lowpass(128, 100e3 / (FS / 2)), t2 = firwin_lowpass(128, 15e3 / (FS / 5 / 2));
        const double tau = 75e-6, r = FS / 5, wc = 1 / tau, wca = 2 * r * std::tan(wc / (2 * r)), taua = 1 / wca;   // singlepolelowpassfilter.lua:55-67
        float b[2] = {(float)(1 / (1 + 2 * taua * r)), (float)(1 / (1 + 2 * taua * r))}, a[2] = {1.f, (float)((1 - 2 * taua * r) / (1 + 2 * taua * r))};
        lrhip_stage_t *s;
        CHK(s = lrhip_format_convert_create(format, 1)); st.push_back(s);
        CHK(s = lrhip_rotator_create(2 * M_PI * (-250e3 / FS))); st.push_back(s);
        CHK(s = lrhip_fir_create(t1.data(), 128, 0, 1, 1, 3)); st.push_back(s);
        CHK(s = lrhip_downsampler_create(5, 8)); st.push_back(s);
        CHK(s = lrhip_fmdiscrim_create(2 * M_PI * 1.25)); st.push_back(s);
        CHK(s = lrhip_fir_create(t2.data(), 128, 0, 0, 1, 3)); st.push_back(s);
        CHK(s = lrhip_iir_create(b, 2, a, 2, 0)); st.push_back(s);
        CHK(s = lrhip_downsampler_create(5, 4)); st.push_back(s);
        CHK(c = lrhip_chain_create(st.data(), (unsigned)st.size()));
        CHK0(lrhip_chain_set_ring(c, 3, ring_chunk));
    }
    ~Receiver()
    {
        lrhip_chain_destroy(c);
        for (auto s : st) lrhip_stage_destroy(s);
    }
};

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static double checksum(const float *y, long n)
{
    double s = 0;
    for (long i = 0; i < n; i++) s += y[i];
    return s;
}

int main(int argc, char **argv)
{
    const unsigned long total = argc > 1 ? strtoul(argv[1], nullptr, 10) : (1ul << 24);
    const char *dir = argc > 2 ? argv[2] : "/tmp";
    if (lrhip_init(0)) { fprintf(stderr, "%s\n", lrhip_strerror()); return 1; }
    // the recordings: u8 I/Q and the same samples as f32le, written once, read back through the page cache
    std::vector<unsigned char> raw(2 * total);
    unsigned lcg = 12345;
    for (auto &v : raw) { lcg = lcg * 1664525u + 1013904223u; v = (unsigned char)(lcg >> 24); }
    std::vector<float> cf(2 * total);
    for (size_t i = 0; i < cf.size(); i++) cf[i] = ((float)raw[i] - 127.5f) / 127.5f;
    struct Fmt { const char *name; const void *data; size_t rec; std::string path; } fmts[2] = {
        {"u8", raw.data(), 2, std::string(dir) + "/lrhip_hp.u8"}, {"f32le", cf.data(), 8, std::string(dir) + "/lrhip_hp.f32"}};
    for (auto &f : fmts) {
        int fd = open(f.path.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0600);
        if (fd < 0) { perror("open"); return 1; }
        size_t put = 0, bytes = f.rec * total;
        while (put < bytes) { ssize_t w = write(fd, (const char *)f.data + put, bytes - put); if (w <= 0) { perror("write"); return 1; } put += (size_t)w; }
        close(fd);
    }
    for (auto &f : fmts) {
        struct Leg { const char *mode; unsigned long chunk; } legs[] = {{"push", 8192}, {"push", 131072}, {"ring", 1ul << 20}, {"fd", 1ul << 20}, {"fd", 1ul << 22}, {"mmap", 1ul << 20}};
        for (auto &leg : legs) {
            double best = 0, sum = 0;
            long n_audio = 0;
            for (int pass = 0; pass < 3; pass++) {           // pass 0: warm-up (allocations, clocks, page cache); the better of the next two counts
                Receiver rx(f.name, leg.chunk > (1ul << 20) ? leg.chunk : (1ul << 20));
                int fd = open(f.path.c_str(), O_RDONLY);
                if (fd < 0) { perror("open"); return 1; }
                const unsigned long cap = lrhip_chain_push_bound(rx.c, 1ul << 20) + (3ul << 22);
                std::vector<float> out(cap), audio;
                audio.reserve(total / 25 + 64);
                std::vector<char> buf(leg.chunk * f.rec);
                const double t0 = now();
                if (!strcmp(leg.mode, "push")) {
                    for (;;) {
                        ssize_t got = read(fd, buf.data(), buf.size());
                        if (got <= 0) break;
                        long n = lrhip_chain_push(rx.c, buf.data(), (unsigned long)got / f.rec, out.data(), cap);
                        if (n < 0) { fprintf(stderr, "%s\n", lrhip_strerror()); return 1; }
                        audio.insert(audio.end(), out.begin(), out.begin() + n);
                    }
                    long n = lrhip_chain_flush(rx.c, out.data(), cap);
                    if (n < 0) { fprintf(stderr, "%s\n", lrhip_strerror()); return 1; }
                    audio.insert(audio.end(), out.begin(), out.begin() + n);
                } else {
                    unsigned long long off = 0;
                    bool eof = false;
                    // mmap: the recording mapped read-only and registered once (lrhip_host_register): lrhip_chain_submit() then DMAs each batch straight out
                    // of the page cache - no read(2), no staging copy.  EXPERIMENTAL: whether a read-only file mapping can be pinned is up to the driver.
                    void *map = nullptr;
                    const size_t map_bytes = f.rec * total;
                    if (!strcmp(leg.mode, "mmap")) {
                        map = mmap(nullptr, map_bytes, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0);
                        if (map == MAP_FAILED) { perror("mmap"); return 1; }
                        if (lrhip_host_register(map, map_bytes) != 0) {
                            if (pass == 0) printf("{\"leg\": \"file_to_receiver\", \"format\": \"%s\", \"mode\": \"mmap\", \"error\": \"lrhip_host_register of a read-only file mapping: %s\"}\n", f.name, lrhip_strerror());
                            munmap(map, map_bytes);
                            map = nullptr;
                            best = -1;
                        }
                    }
                    while (best >= 0 && (!eof || lrhip_chain_in_flight(rx.c) > 0)) {
                        void *slot = eof ? nullptr : lrhip_chain_ring_input(rx.c);
                        if (!slot) {
                            long n = lrhip_chain_collect(rx.c, out.data(), cap);
                            if (n < 0) { fprintf(stderr, "%s\n", lrhip_strerror()); return 1; }
                            audio.insert(audio.end(), out.begin(), out.begin() + n);
                            continue;
                        }
                        long got;
                        if (map) {
                            got = (long)(total - off / f.rec < leg.chunk ? total - off / f.rec : leg.chunk);
                            if (got && lrhip_chain_submit(rx.c, (const char *)map + off, (unsigned long)got) < 0) { fprintf(stderr, "%s\n", lrhip_strerror()); return 1; }
                        } else if (!strcmp(leg.mode, "fd")) {
                            got = lrhip_chain_submit_fd(rx.c, fd, off, leg.chunk);
                            if (got < 0) { fprintf(stderr, "%s\n", lrhip_strerror()); return 1; }
                        } else {
                            ssize_t r = read(fd, slot, leg.chunk * f.rec);
                            got = r > 0 ? (long)((size_t)r / f.rec) : 0;
                            if (got && lrhip_chain_submit(rx.c, slot, (unsigned long)got) < 0) { fprintf(stderr, "%s\n", lrhip_strerror()); return 1; }
                        }
                        if (got == 0) eof = true;
                        off += (unsigned long long)got * f.rec;
                    }
                    if (map) { lrhip_host_unregister(map); munmap(map, map_bytes); }
                }
                if (best < 0) break;                                   // the mapping could not be registered: reported above
                const double dt = now() - t0;
                // (map / unmap outside the figure would flatter it: they are inside)
                close(fd);
                if (pass && total / dt > best) best = total / dt;
                sum = checksum(audio.data(), (long)audio.size());
                n_audio = (long)audio.size();
            }
            if (best < 0) continue;
            printf("{\"leg\": \"file_to_receiver\", \"format\": \"%s\", \"mode\": \"%s\", \"chunk_samples\": %lu, \"MSamples/s\": %.1f, \"h2d_GB/s\": %.2f, "
                   "\"audio_samples\": %ld, \"checksum\": %.9g}\n", f.name, leg.mode, leg.chunk, best / 1e6, f.rec * best / 1e9, n_audio, sum);
            fflush(stdout);
        }
    }
    // stand-alone block, both directions over the link
    {
        std::vector<float> taps = firwin_lowpass(128, 15e3 / (220500.0 / 2));
        float *x = nullptr, *y = nullptr;
        if (posix_memalign((void **)&x, 4096, total * 8) || posix_memalign((void **)&y, 4096, total * 8)) return 1;
        memcpy(x, cf.data(), total * 8);
        memset(y, 0, total * 8);
        for (int leg = 0; leg < 3; leg++) {                    // 2^20-sample vectors staged / registered, 2^22-sample vectors registered
            const int registered = leg > 0;
            const unsigned long vec = leg == 2 && total >= (1ul << 22) ? (1ul << 22) : (1ul << 20);
            if (leg == 1) { CHK0(lrhip_host_register(x, total * 8)); CHK0(lrhip_host_register(y, total * 8)); }
            double best = 0;
            for (int pass = 0; pass < 3; pass++) {
                lrhip_stage_t *q;
                CHK(q = lrhip_fir_create(taps.data(), 128, 0, 1, 1, 2));
                const double t0 = now();
                for (unsigned long a = 0; a + vec <= total; a += vec)
                    if (lrhip_stage_execute(q, x + 2 * a, vec, y + 2 * a, vec) != (long)vec) { fprintf(stderr, "%s\n", lrhip_strerror()); return 1; }
                const double dt = now() - t0;
                if (pass && total / dt > best) best = total / dt;
                lrhip_stage_destroy(q);
            }
            printf("{\"leg\": \"standalone_lowpass_cf32\", \"mode\": \"%s\", \"vector_samples\": %lu, \"MSamples/s\": %.1f, \"each_direction_GB/s\": %.2f, \"checksum\": %.9g}\n",
                   registered ? "registered" : "staged", vec, best / 1e6, 8.0 * best / 1e9, checksum(y, (long)(2 * total)));
            fflush(stdout);
            if (leg == 2) { lrhip_host_unregister(x); lrhip_host_unregister(y); }
        }
        free(x); free(y);
    }
    for (auto &f : fmts) unlink(f.path.c_str());
    return 0;
}
