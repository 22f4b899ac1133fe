// host_fft_check.hip - the FFT building blocks of the overlap-save kernels (radix4 / dft16 / dft64 of kernels_firfft*.h, scalar pk_math) run ON THE CPU: the
// functions are __host__ __device__, so the index algebra of a kernel can be checked without a GPU.  Emulates fir_fft64_kernel's data flow for one block
// (64 lanes x 64 registers, the two half-buffer transposes, the D x C twiddle) against a double-precision DFT.
//   hipcc --offload-arch=gfx950 -O1 -std=c++17 -I luaradio_amd/csrc -I include -o /tmp/host_fft_check tools/host_fft_check.hip && /tmp/host_fft_check
#include <cmath>
#include <complex>
#include <cstdio>
#include <random>
#include <vector>
#include "common.h"
#include "kernels_elem.h"
#include "kernels_fft.h"
#include "kernels_fir.h"
#include "kernels_firwin.h"
#include "kernels_firfft64.h"
using namespace lrhip;
typedef std::complex<double> cd;

int main()
{
    const double PI2 = 6.283185307179586476925286766559;
    std::mt19937 rng(3);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    int bad = 0;
    // ---- dft16 / dft64 against the definition
    {
        cf v[64];
        std::vector<cd> x(64);
        for (int i = 0; i < 64; i++) { v[i] = cf{U(rng), U(rng)}; x[i] = cd(v[i].x, v[i].y); }
        cf w[64];
        for (int i = 0; i < 64; i++) w[i] = v[i];
        dft64<1>(w);
        double e = 0;
        for (int r = 0; r < 64; r++) {
            const int k = f64_index(r);
            cd s = 0;
            for (int n = 0; n < 64; n++) s += x[n] * std::polar(1.0, -PI2 * n * k / 64.0);
            e = std::max(e, std::abs(s - cd(w[r].x, w[r].y)));
        }
        printf("dft64 forward: max err %.3g\n", e);
        bad += e > 1e-4;
        dft64<-1>(w);
        e = 0;
        for (int i = 0; i < 64; i++) e = std::max(e, std::abs(cd(w[i].x, w[i].y) / 64.0 - x[i]));
        printf("dft64 round trip: max err %.3g\n", e);
        bad += e > 1e-5;
    }
    // ---- one 4096-point block through the kernel's flow
    {
        const int N = 4096;
        std::vector<cd> x(N), H(N);
        for (auto &s : x) s = cd(U(rng), U(rng));
        for (auto &s : H) s = cd(U(rng), U(rng)) / (double)N;
        static cf v[64][64], z[64][64], ex[64][65];          // [lane][register]
        for (int t = 0; t < 64; t++)
            for (int i = 0; i < 64; i++) v[t][i] = cf{(float)x[t + 64 * i].real(), (float)x[t + 64 * i].imag()};
        auto tw = [&](int t, int r) { const int k2 = f64_index(r); return std::polar(1.0, -PI2 * (double)t * k2 / N); };
        for (int t = 0; t < 64; t++) {
            dft64<1>(v[t]);
            for (int r = 1; r < 64; r++) { cd w = tw(t, r); v[t][r] = cmul(v[t][r], cf{(float)w.real(), (float)w.imag()}); }
        }
        // forward transpose in two half passes
        for (int half = 0; half < 2; half++) {
            for (int t = 0; t < 64; t++)
                for (int r = 0; r < 64; r++)
                    if ((f64_index(r) >= 32) == (half == 1)) ex[f64_index(r) - 32 * half][t] = v[t][r];
            for (int l = 32 * half; l < 32 * half + 32; l++)
                for (int t = 0; t < 64; t++) z[l][t] = ex[l - 32 * half][t];
        }
        double e = 0;
        for (int l = 0; l < 64; l++) {
            dft64<1>(z[l]);
            for (int r = 0; r < 64; r++) {
                const int k = 64 * f64_index(r) + l;
                cd s = 0;
                for (int n = 0; n < N; n++) s += x[n] * std::polar(1.0, -PI2 * (double)((long)n * k % N) / N);
                e = std::max(e, std::abs(s - cd(z[l][r].x, z[l][r].y)));
                cd h = H[k];
                z[l][r] = cmul(z[l][r], cf{(float)h.real(), (float)h.imag()});
            }
            dft64<-1>(z[l]);
        }
        printf("4096-point forward: max err %.3g\n", e);
        bad += e > 2e-3;
        for (int half = 0; half < 2; half++) {
            for (int l = 32 * half; l < 32 * half + 32; l++)
                for (int t = 0; t < 64; t++) ex[l - 32 * half][t] = z[l][t];
            for (int t = 0; t < 64; t++)
                for (int r = 0; r < 64; r++)
                    if ((f64_index(r) >= 32) == (half == 1)) v[t][r] = ex[f64_index(r) - 32 * half][t];
        }
        // reference: circular convolution y = IDFT(X H N) (H carries 1/N)
        std::vector<cd> X(N), Y(N);
        for (int k = 0; k < N; k++) { cd s = 0; for (int n = 0; n < N; n++) s += x[n] * std::polar(1.0, -PI2 * (double)((long)n * k % N) / N); X[k] = s * H[k]; }
        e = 0;
        for (int t = 0; t < 64; t++) {
            for (int r = 1; r < 64; r++) { cd w = tw(t, r); v[t][r] = cmulc(v[t][r], cf{(float)w.real(), (float)w.imag()}); }
            dft64<-1>(v[t]);
        }
        for (int n = 0; n < N; n += 37) {
            cd s = 0;
            for (int k = 0; k < N; k++) s += X[k] * std::polar(1.0, PI2 * (double)((long)n * k % N) / N);
            e = std::max(e, std::abs(s - cd(v[n % 64][n / 64].x, v[n % 64][n / 64].y)));
        }
        printf("4096-point block (forward, x H, inverse): max err %.3g\n", e);
        bad += e > 1e-4;
    }
    printf(bad ? "FAIL\n" : "OK\n");
    return bad;
}
