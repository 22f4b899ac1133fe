/*
 * lr_oracle.c - CPU restatement of LuaRadio's block arithmetic for the hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it (oracle/README.md).  The product path
 * (luaradio_amd/ + liblrhip.so) never links, imports or falls back to anything here.
 *
 * Parity status: PINNED.  Every function below is checked against the reference's own
 * committed golden vectors (tests/golden/<spec>.json.gz, converted verbatim from the
 * reference's tests/<dir>/<spec>.gen.lua by tests/golden/make_golden.py) in tests/test_oracle_golden.py,
 * at the reference's own epsilons, in whole-vector and one-sample-per-call modes
 * (the two modes of /root/reference/tests/jigs.lua:191-250).
 *
 * The reference (LuaJIT + un-vendored VOLK v2.1.0 / liquid-dsp v1.3.2 / FFTW3f) cannot run in
 * this image (no luajit, no libvolk/libliquid/libfftw3f), so there is no oracle/_ref build; the
 * reference's pure-Lua branch of each block - which its CI holds to the same golden vectors as
 * the VOLK/liquid branches (.github/workflows/tests.yml:99-106) - is what is restated here.
 *
 * Each function cites the reference file:line it follows (paths relative to /root/reference).
 *
 * Arithmetic modes (FIR / IIR):
 *   LRO_MODE_LUA  (0)  the pure-Lua op order: f32*f32 product rounded to f32, f32 add rounded to f32,
 *                      taps applied oldest-sample-first (firfilter.lua:272-280).
 *   LRO_MODE_FMA  (1)  the same order with each mul+add fused (fmaf chain) - this is bit-for-bit what
 *                      the HIP kernels compute (v_fma_f32 / v_mfma_f32_16x16x4_f32 are fmaf chains).
 *   LRO_MODE_SIMD (3)  16 interleaved partial sums over the taps, what a SIMD dot product (VOLK's
 *                      volk_32fc_32f_dot_prod_32fc / volk_32f_x2_dot_prod_32f, firfilter.lua:139-142,157-160) does;
 *                      used as the timed CPU baseline (single thread, or OpenMP over output ranges).
 *   LRO_MODE_F64  (2)  products and sums in double, one final rounding - what scipy.signal.lfilter
 *                      (the generator of the golden vectors, tests/blocks/signal/firfilter_spec.py:7-9) does.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#define LRO_MODE_LUA 0
#define LRO_MODE_FMA 1
#define LRO_MODE_F64 2
#define LRO_MODE_SIMD 3   /* VOLK-style: 16 partial sums across taps (order unspecified, as volk_32fc_32f_dot_prod_32fc) */

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------
 * Windows: radio/utilities/window_utils.lua:11-27 (functions), :39-50 (periodic => M+1)
 * ---------------------------------------------------------------------------------------- */
int lro_window(int M, const char *type, int periodic, double *w)
{
    double Mw = periodic ? (M + 1) : M;
    for (int n = 0; n < M; n++) {
        if (!strcmp(type, "rectangular")) w[n] = 1.0;
        else if (!strcmp(type, "hamming")) w[n] = 0.54 - 0.46 * cos((2 * M_PI * n) / (Mw - 1));
        else if (!strcmp(type, "hanning")) w[n] = 0.5 - 0.5 * cos((2 * M_PI * n) / (Mw - 1));
        else if (!strcmp(type, "bartlett")) w[n] = (2 / (Mw - 1)) * ((Mw - 1) / 2 - fabs(n - (Mw - 1) / 2));
        else if (!strcmp(type, "blackman"))
            w[n] = 0.42 - 0.5 * cos((2 * M_PI * n) / (Mw - 1)) + 0.08 * cos((4 * M_PI * n) / (Mw - 1));
        else return -1;
    }
    return 0;
}

/* radio/utilities/filter_utils.lua:121-141 (firwin: window, then scale to unity gain at scale_freq) */
static int firwin_apply(double *h, int M, const char *window, double scale_freq)
{
    double *w = (double *)malloc(sizeof(double) * M);
    if (lro_window(M, window ? window : "hamming", 0, w)) { free(w); return -1; }
    for (int n = 0; n < M; n++) h[n] = h[n] * w[n];
    double scale = 0;
    for (int n = 0; n < M; n++) scale = scale + h[n] * cos(M_PI * (n - (M - 1) / 2.0) * scale_freq);
    for (int n = 0; n < M; n++) h[n] = h[n] / scale;
    free(w);
    return 0;
}

/* filter_utils.lua:21-33 + :152-157 */
int lro_firwin_lowpass(int M, double cutoff, const char *window, double *h)
{
    for (int n = 0; n < M; n++) {
        double c = n - (M - 1) / 2.0;
        h[n] = (c == 0.0) ? cutoff : sin(M_PI * cutoff * c) / (M_PI * c);
    }
    return firwin_apply(h, M, window, 0.0);
}

/* filter_utils.lua:43-57 + :168-173 */
int lro_firwin_highpass(int M, double cutoff, const char *window, double *h)
{
    if ((M % 2) != 1) return -2;
    for (int n = 0; n < M; n++) {
        double c = n - (M - 1) / 2.0;
        h[n] = (c == 0.0) ? 1 - cutoff : -sin(M_PI * cutoff * c) / (M_PI * c);
    }
    return firwin_apply(h, M, window, 1.0);
}

/* filter_utils.lua:67-82 + :184-189 */
int lro_firwin_bandpass(int M, double c1, double c2, const char *window, double *h)
{
    if ((M % 2) != 1) return -2;
    for (int n = 0; n < M; n++) {
        double c = n - (M - 1) / 2.0;
        h[n] = (c == 0.0) ? (c2 - c1) : sin(M_PI * c2 * c) / (M_PI * c) - sin(M_PI * c1 * c) / (M_PI * c);
    }
    return firwin_apply(h, M, window, (c1 + c2) / 2);
}

/* filter_utils.lua:92-107 + :200-205 */
int lro_firwin_bandstop(int M, double c1, double c2, const char *window, double *h)
{
    if ((M % 2) != 1) return -2;
    for (int n = 0; n < M; n++) {
        double c = n - (M - 1) / 2.0;
        h[n] = (c == 0.0) ? 1 - (c2 - c1) : sin(M_PI * c1 * c) / (M_PI * c) - sin(M_PI * c2 * c) / (M_PI * c);
    }
    return firwin_apply(h, M, window, 0.0);
}

/* ------------------------------------------------------------------------------------------
 * FIRFilterBlock, dot-product form: radio/blocks/signal/firfilter.lua:230-305 (pure-Lua branch;
 * VOLK branch :90-163 has the same history handling).  state = [last M-1 inputs | chunk];
 * out[i] = sum_j state[i+j] * taps_reversed[j], j ascending (oldest sample first).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int ntaps, taps_complex, input_complex, mode;
    float *taps_rev;      /* reversed taps (firfilter.lua:234-238), 1 or 2 floats per tap */
    float *hist;          /* last ntaps-1 input samples, zero-initialised (:240) */
    float *state;         /* scratch: hist ++ chunk */
    long state_cap;
} lro_fir;

lro_fir *lro_fir_create(const float *taps, int ntaps, int taps_complex, int input_complex, int mode)
{
    if (ntaps < 1 || (taps_complex && !input_complex)) return NULL;
    lro_fir *q = (lro_fir *)calloc(1, sizeof(*q));
    int ts = taps_complex ? 2 : 1, es = input_complex ? 2 : 1;
    q->ntaps = ntaps; q->taps_complex = taps_complex; q->input_complex = input_complex; q->mode = mode;
    q->taps_rev = (float *)malloc(sizeof(float) * ts * ntaps);
    for (int i = 0; i < ntaps; i++)
        for (int c = 0; c < ts; c++) q->taps_rev[i * ts + c] = taps[(ntaps - 1 - i) * ts + c];
    q->hist = (float *)calloc((size_t)es * ntaps, sizeof(float));
    return q;
}

void lro_fir_destroy(lro_fir *q)
{
    if (!q) return;
    free(q->taps_rev); free(q->hist); free(q->state); free(q);
}

/* one output sample from window s[0..M) (interleaved if complex) */
static inline void fir_dot_cr(const float *s, const float *h, int M, int mode, float *out)
{
    if (mode == LRO_MODE_F64) {
        double re = 0, im = 0;
        for (int j = 0; j < M; j++) { re += (double)s[2 * j] * h[j]; im += (double)s[2 * j + 1] * h[j]; }
        out[0] = (float)re; out[1] = (float)im;
    } else if (mode == LRO_MODE_FMA) {
        float re = 0, im = 0;
        for (int j = 0; j < M; j++) { re = fmaf(s[2 * j], h[j], re); im = fmaf(s[2 * j + 1], h[j], im); }
        out[0] = re; out[1] = im;
    } else { /* firfilter.lua:277-280: scalar_mul (rounds to f32) then + (rounds to f32) */
        volatile float re = 0, im = 0;
        for (int j = 0; j < M; j++) {
            volatile float pr = s[2 * j] * h[j], pi = s[2 * j + 1] * h[j];
            re = re + pr; im = im + pi;
        }
        out[0] = re; out[1] = im;
    }
}

static inline void fir_dot_rr(const float *s, const float *h, int M, int mode, float *out)
{
    if (mode == LRO_MODE_F64) {
        double a = 0;
        for (int j = 0; j < M; j++) a += (double)s[j] * h[j];
        out[0] = (float)a;
    } else if (mode == LRO_MODE_FMA) {
        float a = 0;
        for (int j = 0; j < M; j++) a = fmaf(s[j], h[j], a);
        out[0] = a;
    } else { /* firfilter.lua:299-302 */
        volatile float a = 0;
        for (int j = 0; j < M; j++) { volatile float p = s[j] * h[j]; a = a + p; }
        out[0] = a;
    }
}

static inline void fir_dot_cc(const float *s, const float *h, int M, int mode, float *out)
{
    if (mode == LRO_MODE_F64) {
        double re = 0, im = 0;
        for (int j = 0; j < M; j++) {
            double xr = s[2 * j], xi = s[2 * j + 1], hr = h[2 * j], hi = h[2 * j + 1];
            re += xr * hr - xi * hi; im += xr * hi + xi * hr;
        }
        out[0] = (float)re; out[1] = (float)im;
    } else if (mode == LRO_MODE_FMA) {
        /* kernel order: per tap, re += xr*hr; re += xi*(-hi); im += xr*hi; im += xi*hr */
        float re = 0, im = 0;
        for (int j = 0; j < M; j++) {
            float xr = s[2 * j], xi = s[2 * j + 1], hr = h[2 * j], hi = h[2 * j + 1];
            re = fmaf(xr, hr, re); re = fmaf(xi, -hi, re);
            im = fmaf(xr, hi, im); im = fmaf(xi, hr, im);
        }
        out[0] = re; out[1] = im;
    } else { /* firfilter.lua:255-258 with complexfloat32.lua:79-81 (__mul rounds each component once) */
        volatile float re = 0, im = 0;
        for (int j = 0; j < M; j++) {
            double xr = s[2 * j], xi = s[2 * j + 1], hr = h[2 * j], hi = h[2 * j + 1];
            volatile float pr = (float)(xr * hr - xi * hi), pi = (float)(xr * hi + xi * hr);
            re = re + pr; im = im + pi;
        }
        out[0] = re; out[1] = im;
    }
}

/* firfilter.lua:244-305: shift history, append chunk, one dot product per output sample */
long lro_fir_process(lro_fir *q, const float *x, long n, float *y)
{
    int M = q->ntaps, es = q->input_complex ? 2 : 1;
    long need = (long)(M - 1 + n) * es;
    if (need > q->state_cap) { q->state = (float *)realloc(q->state, sizeof(float) * need); q->state_cap = need; }
    memcpy(q->state, q->hist, sizeof(float) * es * (M - 1));
    memcpy(q->state + (size_t)es * (M - 1), x, sizeof(float) * es * n);
    for (long i = 0; i < n; i++) {
        const float *s = q->state + (size_t)es * i;
        if (!q->input_complex) fir_dot_rr(s, q->taps_rev, M, q->mode, y + i);
        else if (!q->taps_complex) fir_dot_cr(s, q->taps_rev, M, q->mode, y + 2 * i);
        else fir_dot_cc(s, q->taps_rev, M, q->mode, y + 2 * i);
    }
    /* keep the last M-1 samples of state (firfilter.lua:248: memmove from state.length-(M-1)) */
    memmove(q->hist, q->state + (size_t)es * n, sizeof(float) * es * (M - 1));
    return n;
}

/* VOLK-style dot product over a float stream z[0..len) with (possibly duplicated) taps hd[0..len):
 * 16 independent partial sums so the compiler emits packed FMAs; even lanes / odd lanes are re / im when z is
 * an interleaved complex stream and hd holds every tap twice. */
static inline void dot16(const float *z, const float *hd, int len, float *even, float *odd)
{
    float acc[16] = {0};
    int k = 0;
    for (; k + 16 <= len; k += 16)
        for (int l = 0; l < 16; l++) acc[l] += z[k + l] * hd[k + l];
    for (; k < len; k++) acc[k & 15] += z[k] * hd[k];
    float e = 0, o = 0;
    for (int l = 0; l < 16; l += 2) { e += acc[l]; o += acc[l + 1]; }
    *even = e; *odd = o;
}

/* Timed CPU baseline: same history handling as lro_fir_process, dot products in LRO_MODE_SIMD, optionally
 * split over `nthreads` OpenMP threads (contiguous output ranges; the reference itself gives a block one core,
 * docs/5.architecture.md:62-68, so nthreads = 1 is the reference-equivalent figure). Real taps only. */
long lro_fir_process_simd(lro_fir *q, const float *x, long n, float *y, int nthreads)
{
    int M = q->ntaps, es = q->input_complex ? 2 : 1;
    if (q->taps_complex) return -1;
    long need = (long)(M - 1 + n) * es;
    if (need > q->state_cap) { q->state = (float *)realloc(q->state, sizeof(float) * need); q->state_cap = need; }
    memcpy(q->state, q->hist, sizeof(float) * es * (M - 1));
    memcpy(q->state + (size_t)es * (M - 1), x, sizeof(float) * es * n);
    float *hd = (float *)malloc(sizeof(float) * es * M);
    for (int j = 0; j < M; j++) for (int c = 0; c < es; c++) hd[j * es + c] = q->taps_rev[j];
    const float *state = q->state;
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
    for (long i = 0; i < n; i++) {
        float e, o;
        dot16(state + (size_t)es * i, hd, M * es, &e, &o);
        if (es == 2) { y[2 * i] = e; y[2 * i + 1] = o; } else y[i] = e + o;
    }
    free(hd);
    memmove(q->hist, q->state + (size_t)es * n, sizeof(float) * es * (M - 1));
    return n;
}

/* ------------------------------------------------------------------------------------------
 * Self-contained complex FFT (iterative radix-2, double precision twiddles and data) standing in for
 * FFTW (radio/utilities/spectrum_utils.lua:86-113) and for the pure-Lua O(N^2) DFT (:206-246):
 * both compute X_k = sum_n x_n e^{-2 pi i k n / N}.  Non-power-of-2 even N falls back to O(N^2).
 * ---------------------------------------------------------------------------------------- */
static void dft_double(double *re, double *im, int N, int inverse)
{
    if ((N & (N - 1)) == 0) {
        for (int i = 1, j = 0; i < N; i++) {
            int bit = N >> 1;
            for (; j & bit; bit >>= 1) j ^= bit;
            j ^= bit;
            if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
        }
        for (int len = 2; len <= N; len <<= 1) {
            double ang = (inverse ? 2 : -2) * M_PI / len;
            for (int i = 0; i < N; i += len)
                for (int k = 0; k < len / 2; k++) {
                    double wr = cos(ang * k), wi = sin(ang * k);
                    int a = i + k, b = i + k + len / 2;
                    double tr = re[b] * wr - im[b] * wi, ti = re[b] * wi + im[b] * wr;
                    re[b] = re[a] - tr; im[b] = im[a] - ti; re[a] += tr; im[a] += ti;
                }
        }
    } else {
        double *or_ = (double *)calloc(N, sizeof(double)), *oi = (double *)calloc(N, sizeof(double));
        for (int k = 0; k < N; k++)
            for (int n = 0; n < N; n++) {
                double ang = (inverse ? 2 : -2) * M_PI * ((long)k * n % N) / N;
                or_[k] += re[n] * cos(ang) - im[n] * sin(ang);
                oi[k] += re[n] * sin(ang) + im[n] * cos(ang);
            }
        memcpy(re, or_, sizeof(double) * N); memcpy(im, oi, sizeof(double) * N);
        free(or_); free(oi);
    }
}

/* spectrum_utils.lua:25-57 (DFT; real input => Hermitian fill :109-112, i.e. the full complex spectrum) */
int lro_dft(const float *in, int N, int in_complex, float *out)
{
    if (N % 2) return -1;
    double *re = (double *)malloc(sizeof(double) * N), *im = (double *)malloc(sizeof(double) * N);
    for (int i = 0; i < N; i++) { re[i] = in_complex ? in[2 * i] : in[i]; im[i] = in_complex ? in[2 * i + 1] : 0.0; }
    dft_double(re, im, N, 0);
    for (int i = 0; i < N; i++) { out[2 * i] = (float)re[i]; out[2 * i + 1] = (float)im[i]; }
    free(re); free(im);
    return 0;
}

/* spectrum_utils.lua:259-349 (IDFT with 1/N normalisation :335-338; real output takes the real part :499-503) */
int lro_idft(const float *in, int N, int out_complex, float *out)
{
    if (N % 2) return -1;
    double *re = (double *)malloc(sizeof(double) * N), *im = (double *)malloc(sizeof(double) * N);
    for (int i = 0; i < N; i++) { re[i] = in[2 * i]; im[i] = in[2 * i + 1]; }
    dft_double(re, im, N, 1);
    for (int i = 0; i < N; i++) {
        if (out_complex) { out[2 * i] = (float)(re[i] * (1.0 / N)); out[2 * i + 1] = (float)(im[i] * (1.0 / N)); }
        else out[i] = (float)(re[i] * (1.0 / N));
    }
    free(re); free(im);
    return 0;
}

/* spectrum_utils.lua:522-561 (constructor: periodic window as f32, window energy from the f32 window) and
 * :611-640 (compute: window multiply in f32, DFT, |X|^2/(fs*sum w^2), optional 10*log10) */
int lro_psd(const float *in, int N, int in_complex, const char *window, double sample_rate, int logarithmic, float *out)
{
    if (N % 2) return -1;
    double *wd = (double *)malloc(sizeof(double) * N);
    if (lro_window(N, window ? window : "hamming", 1, wd)) { free(wd); return -2; }
    float *w = (float *)malloc(sizeof(float) * N);
    double energy = 0;
    for (int i = 0; i < N; i++) { w[i] = (float)wd[i]; energy = energy + (double)w[i] * (double)w[i]; }
    int es = in_complex ? 2 : 1;
    float *xw = (float *)malloc(sizeof(float) * N * es), *X = (float *)malloc(sizeof(float) * 2 * N);
    for (int i = 0; i < N; i++)
        for (int c = 0; c < es; c++) xw[i * es + c] = in[i * es + c] * w[i];
    lro_dft(xw, N, in_complex, X);
    double scale = sample_rate * energy;
    for (int i = 0; i < N; i++) {
        double p = ((double)X[2 * i] * X[2 * i] + (double)X[2 * i + 1] * X[2 * i + 1]) / scale;
        out[i] = (float)(logarithmic ? 10 * log10(p) : p);
    }
    free(wd); free(w); free(xw); free(X);
    return 0;
}

/* spectrum_utils.lua:654-667 (swap halves in place) */
void lro_fftshift(float *x, int N, int is_complex)
{
    int es = is_complex ? 2 : 1, off = N / 2;
    for (int k = 0; k < N / 2; k++)
        for (int c = 0; c < es; c++) {
            float t = x[k * es + c]; x[k * es + c] = x[(k + off) * es + c]; x[(k + off) * es + c] = t;
        }
}

/* ------------------------------------------------------------------------------------------
 * FIRFilterBlock, FFT overlap-save form: firfilter.lua:406-486 (framing identical to the VOLK twin :320-398)
 * N = 2^floor(log2(8M)), L = N-M+1; emits floor((fill+n)/L)*L samples per call, tail retained.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int M, N, L, input_complex, fill;
    double *Hr, *Hi;       /* taps DFT (:427-428) */
    double *blk;           /* input block, N complex (re,im) interleaved doubles */
} lro_firfft;

lro_firfft *lro_firfft_create(const float *taps, int ntaps, int taps_complex, int input_complex)
{
    if (ntaps < 1 || (taps_complex && !input_complex)) return NULL;
    lro_firfft *q = (lro_firfft *)calloc(1, sizeof(*q));
    q->M = ntaps; q->input_complex = input_complex;
    q->N = 1 << (int)floor(log(8.0 * ntaps) / log(2.0));   /* :414 */
    q->L = q->N - ntaps + 1;                               /* :415 */
    q->Hr = (double *)calloc(q->N, sizeof(double)); q->Hi = (double *)calloc(q->N, sizeof(double));
    for (int i = 0; i < ntaps; i++) { q->Hr[i] = taps_complex ? taps[2 * i] : taps[i]; q->Hi[i] = taps_complex ? taps[2 * i + 1] : 0; }
    dft_double(q->Hr, q->Hi, q->N, 0);
    q->blk = (double *)calloc(2 * (size_t)q->N, sizeof(double));
    return q;
}

void lro_firfft_destroy(lro_firfft *q) { if (q) { free(q->Hr); free(q->Hi); free(q->blk); free(q); } }

int lro_firfft_block_length(const lro_firfft *q) { return q->L; }

long lro_firfft_process(lro_firfft *q, const float *x, long n, float *y, long cap)
{
    int N = q->N, M = q->M, L = q->L, es = q->input_complex ? 2 : 1;
    long want = ((q->fill + n) / L) * L, oi = 0, i = 0;     /* :451 */
    if (want > cap) return -1;
    double *re = (double *)malloc(sizeof(double) * N), *im = (double *)malloc(sizeof(double) * N);
    while (i < n) {
        long len = n - i < L - q->fill ? n - i : L - q->fill;     /* :457 */
        for (long k = 0; k < len; k++) {
            q->blk[2 * (M - 1 + q->fill + k)] = x[(i + k) * es];
            q->blk[2 * (M - 1 + q->fill + k) + 1] = es == 2 ? x[(i + k) * es + 1] : 0.0;
        }
        q->fill += (int)len; i += len;
        if (q->fill < L) break;                                   /* :463-465 */
        for (int k = 0; k < N; k++) { re[k] = q->blk[2 * k]; im[k] = q->blk[2 * k + 1]; }
        dft_double(re, im, N, 0);
        for (int k = 0; k < N; k++) {                             /* :471-473 */
            double a = re[k] * q->Hr[k] - im[k] * q->Hi[k], b = re[k] * q->Hi[k] + im[k] * q->Hr[k];
            re[k] = a; im[k] = b;
        }
        dft_double(re, im, N, 1);
        for (int k = 0; k < L; k++) {                             /* :479 copy output_block[M-1 ..] */
            if (es == 2) { y[2 * (oi + k)] = (float)(re[M - 1 + k] / N); y[2 * (oi + k) + 1] = (float)(im[M - 1 + k] / N); }
            else y[oi + k] = (float)(re[M - 1 + k] / N);
        }
        oi += L;
        memmove(q->blk, q->blk + 2 * (size_t)(N - (M - 1)), sizeof(double) * 2 * (M - 1));   /* :483 */
        q->fill = 0;
    }
    free(re); free(im);
    return oi;
}

/* ------------------------------------------------------------------------------------------
 * FrequencyTranslatorBlock: radio/blocks/signal/frequencytranslator.lua:93-110 (pure-Lua branch):
 * double phase accumulator, cos/sin in double rounded into a ComplexFloat32, complex multiply rounding
 * each component once (complexfloat32.lua:79-81), wrap when phase > 2*pi.
 * mode 0 = that; mode 2 = closed form x[n]*exp(j*omega*n) with n a 64-bit counter (what the device
 * kernel computes; long-run behaviour is unpinned in the reference - SURVEY.md 8c(iii)).
 * ---------------------------------------------------------------------------------------- */
typedef struct { double omega, phase; int mode; uint64_t count; } lro_rotator;

lro_rotator *lro_rotator_create(double omega, int mode)
{
    lro_rotator *q = (lro_rotator *)calloc(1, sizeof(*q));
    q->omega = omega; q->mode = mode;
    return q;
}
void lro_rotator_destroy(lro_rotator *q) { free(q); }

long lro_rotator_process(lro_rotator *q, const float *x, long n, float *y)
{
    for (long i = 0; i < n; i++) {
        double ph;
        if (q->mode == LRO_MODE_F64) {
            /* reduce omega*count mod 2*pi in extended precision */
            long double t = (long double)q->omega * (long double)q->count;
            t = fmodl(t, 2 * 3.14159265358979323846264338327950288L);
            ph = (double)t; q->count++;
        } else ph = q->phase;
        float cr = (float)cos(ph), ci = (float)sin(ph);
        double xr = x[2 * i], xi = x[2 * i + 1];
        y[2 * i] = (float)(xr * cr - xi * ci);
        y[2 * i + 1] = (float)(xr * ci + xi * cr);
        if (q->mode != LRO_MODE_F64) {
            q->phase = q->phase + q->omega;
            q->phase = (q->phase > 2 * M_PI) ? (q->phase - 2 * M_PI) : q->phase;
        }
    }
    return n;
}

/* ------------------------------------------------------------------------------------------
 * DownsamplerBlock: radio/blocks/signal/downsampler.lua:40-56.  Bit-exact copies.
 * ---------------------------------------------------------------------------------------- */
typedef struct { long factor, index; int elem_floats; } lro_downsampler;

lro_downsampler *lro_downsampler_create(long factor, int elem_floats)
{
    if (factor < 1) return NULL;
    lro_downsampler *q = (lro_downsampler *)calloc(1, sizeof(*q));
    q->factor = factor; q->elem_floats = elem_floats;
    return q;
}
void lro_downsampler_destroy(lro_downsampler *q) { free(q); }

long lro_downsampler_process(lro_downsampler *q, const float *x, long n, float *y)
{
    /* :46 out.length = ceil((x.length - index)/factor) */
    long len = (n - q->index + q->factor - 1) / q->factor;
    if (n - q->index <= 0) len = 0;
    for (long i = 0; i < len; i++) {
        memcpy(y + i * q->elem_floats, x + q->index * q->elem_floats, sizeof(float) * q->elem_floats);
        q->index += q->factor;
    }
    q->index -= n;     /* :53 */
    return len;
}

/* ------------------------------------------------------------------------------------------
 * FrequencyDiscriminatorBlock: radio/blocks/signal/frequencydiscriminator.lua:33-38, :68-88.
 * tmp = x[n]*conj(x[n-1]) (each component rounded once to f32), out = atan2f(im, re) * (1/gain).
 * ---------------------------------------------------------------------------------------- */
typedef struct { double gain; float prev_re, prev_im; } lro_fmdiscrim;

lro_fmdiscrim *lro_fmdiscrim_create(double modulation_index)
{
    lro_fmdiscrim *q = (lro_fmdiscrim *)calloc(1, sizeof(*q));
    q->gain = 2 * M_PI * modulation_index;     /* :28 */
    return q;
}
void lro_fmdiscrim_destroy(lro_fmdiscrim *q) { free(q); }

long lro_fmdiscrim_process(lro_fmdiscrim *q, const float *x, long n, float *y)
{
    for (long i = 0; i < n; i++) {
        double ar = x[2 * i], ai = x[2 * i + 1];
        double br = i ? x[2 * i - 2] : q->prev_re, bi = -(double)(i ? x[2 * i - 1] : q->prev_im);
        float tr = (float)(ar * br - ai * bi), ti = (float)(ar * bi + ai * br);
        y[i] = (float)((double)atan2f(ti, tr) * (1 / q->gain));     /* :81 */
    }
    if (n > 0) { q->prev_re = x[2 * n - 2]; q->prev_im = x[2 * n - 1]; }
    return n;
}

/* ------------------------------------------------------------------------------------------
 * IIRFilterBlock: radio/blocks/signal/iirfilter.lua:113-181 (pure-Lua branch).
 * y[n] = (sum_j b[j] x[n-j] - sum_{j>=1} a[j] y[n-j]) / a[0], each op rounded to f32 (mode LUA);
 * mode F64 keeps the accumulator in double (what scipy.signal.lfilter, the vector generator, does).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int nb, na, input_complex, mode;
    float *b, *a;
    float *xs, *ys;       /* input_state (nb), output_state (na-1); index 0 = newest (:126-128) */
    double *ysd;
} lro_iir;

lro_iir *lro_iir_create(const float *b, int nb, const float *a, int na, int input_complex, int mode)
{
    if (nb < 1 || na < 1) return NULL;
    lro_iir *q = (lro_iir *)calloc(1, sizeof(*q));
    int es = input_complex ? 2 : 1;
    q->nb = nb; q->na = na; q->input_complex = input_complex; q->mode = mode;
    q->b = (float *)malloc(sizeof(float) * nb); memcpy(q->b, b, sizeof(float) * nb);
    q->a = (float *)malloc(sizeof(float) * na); memcpy(q->a, a, sizeof(float) * na);
    q->xs = (float *)calloc((size_t)nb * es, sizeof(float));
    q->ys = (float *)calloc((size_t)na * es, sizeof(float));
    q->ysd = (double *)calloc((size_t)na * es, sizeof(double));
    return q;
}
void lro_iir_destroy(lro_iir *q) { if (q) { free(q->b); free(q->a); free(q->xs); free(q->ys); free(q->ysd); free(q); } }

long lro_iir_process(lro_iir *q, const float *x, long n, float *y)
{
    int es = q->input_complex ? 2 : 1, nb = q->nb, na = q->na;
    for (long i = 0; i < n; i++) {
        /* :122-125 shift input state down, insert x[i] at index 0 */
        memmove(q->xs + es, q->xs, sizeof(float) * es * (nb - 1));
        for (int c = 0; c < es; c++) q->xs[c] = x[i * es + c];
        double accd[2] = {0, 0};
        for (int c = 0; c < es; c++) {
            if (q->mode == LRO_MODE_F64) {
                double acc = 0;
                for (int j = 0; j < nb; j++) acc += (double)q->xs[j * es + c] * q->b[j];
                for (int j = 0; j < na - 1; j++) acc -= q->ysd[j * es + c] * q->a[j + 1];
                acc /= q->a[0];
                accd[c] = acc;
                y[i * es + c] = (float)acc;
            } else {   /* :127-138 every op rounds to f32 */
                volatile float acc = 0;
                for (int j = 0; j < nb; j++) { volatile float p = q->xs[j * es + c] * q->b[j]; acc = acc + p; }
                for (int j = 0; j < na - 1; j++) { volatile float p = q->ys[j * es + c] * q->a[j + 1]; acc = acc - p; }
                acc = acc / q->a[0];
                y[i * es + c] = acc;
            }
        }
        if (na > 1) {   /* :140-143 shift output state down, insert y[i] */
            memmove(q->ys + es, q->ys, sizeof(float) * es * (na - 2));
            memmove(q->ysd + es, q->ysd, sizeof(double) * es * (na - 2));
            for (int c = 0; c < es; c++) { q->ys[c] = y[i * es + c]; q->ysd[c] = accd[c]; }
        }
    }
    return n;
}

/* SinglepoleLowpassFilterBlock:initialize radio/blocks/signal/singlepolelowpassfilter.lua:55-67
 * (FMDeemphasisFilterBlock: cutoff = 1/(2*pi*tau), fmdeemphasisfilter.lua:24-27). Double math, cast once. */
void lro_singlepole_lowpass_taps(double cutoff, double rate, float b[2], float a[2])
{
    double tau = 1 / (2 * M_PI * cutoff);
    tau = 1 / (2 * rate * tan(1 / (2 * rate * tau)));
    b[0] = (float)(1 / (1 + 2 * tau * rate));
    b[1] = (float)(1 / (1 + 2 * tau * rate));
    a[0] = 1.0f;
    a[1] = (float)((1 - 2 * tau * rate) / (1 + 2 * tau * rate));
}

/* MultiplyConjugateBlock (pure Lua): radio/blocks/signal/multiplyconjugate.lua - out = a * conj(b) */
void lro_multiply_conjugate(const float *a, const float *b, long n, float *y)
{
    for (long i = 0; i < n; i++) {
        double ar = a[2 * i], ai = a[2 * i + 1], br = b[2 * i], bi = -(double)b[2 * i + 1];
        y[2 * i] = (float)(ar * br - ai * bi);
        y[2 * i + 1] = (float)(ar * bi + ai * br);
    }
}

/* ------------------------------------------------------------------------------------------
 * IQFileSource / RealFileSource conversion: radio/blocks/sources/iqfile.lua:99-113, realfile.lua:99-110 with the
 * formats table radio/utilities/format_utils.lua:82-97: optional byte swap, then (value - offset)/scale in double,
 * rounded to float on store.  `nscalars` raw scalars in, floats out.  Returns -1 for an unknown format.
 * ---------------------------------------------------------------------------------------- */
long lro_format_convert(const char *format, const unsigned char *raw, long nscalars, float *out)
{
    static const struct { const char *name; int bytes, cls, be; double offset, scale; } F[] = {
        {"u8", 1, 0, 0, 127.5, 127.5}, {"s8", 1, 1, 0, 0, 127.5},
        {"u16le", 2, 2, 0, 32767.5, 32767.5}, {"u16be", 2, 2, 1, 32767.5, 32767.5},
        {"s16le", 2, 3, 0, 0, 32767.5}, {"s16be", 2, 3, 1, 0, 32767.5},
        {"u32le", 4, 4, 0, 2147483647.5, 2147483647.5}, {"u32be", 4, 4, 1, 2147483647.5, 2147483647.5},
        {"s32le", 4, 5, 0, 0, 2147483647.5}, {"s32be", 4, 5, 1, 0, 2147483647.5},
        {"f32le", 4, 6, 0, 0, 1.0}, {"f32be", 4, 6, 1, 0, 1.0}, {"f64le", 8, 7, 0, 0, 1.0}, {"f64be", 8, 7, 1, 0, 1.0}};
    int k = -1;
    for (unsigned i = 0; i < sizeof(F) / sizeof(F[0]); i++) if (!strcmp(F[i].name, format)) k = (int)i;
    if (k < 0) return -1;
    for (long i = 0; i < nscalars; i++) {
        unsigned char b[8];
        for (int j = 0; j < F[k].bytes; j++) b[j] = raw[i * F[k].bytes + (F[k].be ? F[k].bytes - 1 - j : j)];   /* host is little-endian */
        double v;
        switch (F[k].cls) {
            case 0: v = b[0]; break;
            case 1: v = (int8_t)b[0]; break;
            case 2: { uint16_t t; memcpy(&t, b, 2); v = t; break; }
            case 3: { int16_t t; memcpy(&t, b, 2); v = t; break; }
            case 4: { uint32_t t; memcpy(&t, b, 4); v = t; break; }
            case 5: { int32_t t; memcpy(&t, b, 4); v = t; break; }
            case 6: { float t; memcpy(&t, b, 4); v = t; break; }
            default: { double t; memcpy(&t, b, 8); v = t; break; }
        }
        out[i] = (float)((v - F[k].offset) / F[k].scale);
    }
    return nscalars;
}

/* ------------------------------------------------------------------------------------------
 * FrequencyModulatorBlock: radio/blocks/signal/frequencymodulator.lua:71-90 (pure-Lua branch):
 * phase (a Lua double) = (phase + delta*x) % (2 pi); out = (cosf(phase), sinf(phase)) - the double is narrowed to
 * float at the cosf/sinf call (ffi.C.cosf takes a float).  Lua's % is a floored modulo.
 * ---------------------------------------------------------------------------------------- */
typedef struct { double phase, delta; } lro_fmmod;

lro_fmmod *lro_fmmod_create(double modulation_index)
{
    lro_fmmod *q = (lro_fmmod *)calloc(1, sizeof(*q));
    q->delta = 2 * M_PI * modulation_index;      /* :73 */
    return q;
}
void lro_fmmod_destroy(lro_fmmod *q) { free(q); }

long lro_fmmod_process(lro_fmmod *q, const float *x, long n, float *y)
{
    const double two_pi = 2 * M_PI;
    for (long i = 0; i < n; i++) {
        double p = q->phase + q->delta * (double)x[i];
        p = p - floor(p / two_pi) * two_pi;          /* Lua: a % b == a - floor(a/b)*b */
        q->phase = p;
        y[2 * i] = cosf((float)p);
        y[2 * i + 1] = sinf((float)p);
    }
    return n;
}

/* ------------------------------------------------------------------------------------------
 * AGCBlock: radio/blocks/signal/agc.lua:45-96.  average_power and gain are Lua numbers (double); the input sample is
 * Float32; out = sqrt(gain) * x narrowed to Float32 on store.
 * ---------------------------------------------------------------------------------------- */
typedef struct { double pa, ga, target, thr, power, gain; int cplx; } lro_agc;

lro_agc *lro_agc_create(double power_alpha, double gain_alpha, double target_lin, double threshold_lin, int input_complex)
{
    lro_agc *q = (lro_agc *)calloc(1, sizeof(*q));
    q->pa = power_alpha; q->ga = gain_alpha; q->target = target_lin; q->thr = threshold_lin; q->cplx = input_complex;
    return q;
}
void lro_agc_destroy(lro_agc *q) { free(q); }

long lro_agc_process(lro_agc *q, const float *x, long n, float *y)
{
    int S = q->cplx ? 2 : 1;
    for (long i = 0; i < n; i++) {
        double e = S == 1 ? (double)x[i] * (double)x[i] : (double)x[2 * i] * (double)x[2 * i] + (double)x[2 * i + 1] * (double)x[2 * i + 1];
        q->power = (1 - q->pa) * q->power + q->pa * e;                                   /* :50 / :70 */
        if (q->power >= q->thr) {
            q->gain = (1 - q->ga) * q->gain + q->ga * (q->target * (1 / q->power));      /* :54 */
            double g = sqrt(q->gain);
            for (int c = 0; c < S; c++) y[S * i + c] = (float)(g * (double)x[S * i + c]);
        } else {
            for (int c = 0; c < S; c++) y[S * i + c] = x[S * i + c];
        }
    }
    return n;
}
