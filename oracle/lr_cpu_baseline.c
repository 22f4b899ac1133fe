/*
 * lr_cpu_baseline.c - the TIMED CPU forms of FIRFilterBlock for bench.py's cpu_baseline leg.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE (see lr_oracle.c header and oracle/README.md).
 *
 * The reference runs a FIR in one of two forms (paths relative to /root/reference):
 *   (1) one SIMD dot product per output sample over a sliding state buffer
 *       (radio/blocks/signal/firfilter.lua:129-145, VOLK volk_32fc_32f_dot_prod_32fc), and
 *   (2) FFT overlap-save, its default whenever FFTW is present (firfilter.lua:57, :320-398): N = 2^floor(log2(8M)),
 *       L = N-M+1, per block one forward DFT, N complex multiplies by the taps' DFT, one inverse DFT, 1/N scale.
 * LuaJIT, VOLK and FFTW3f are absent from this image, so both are restated here as plain C built with
 * -O3 -march=x86-64-v3 (AVX2 + FMA, contraction allowed - these are the speed builds; the arithmetic-order-exact
 * restatements used for parity live in lr_oracle.c).  Form (2) carries its own Float32 FFT (Stockham autosort, radix 4 with a
 * final radix-2 pass, split re/im arrays so the butterflies vectorise); it is not FFTW, and the bench line says so.
 * Both are checked against lr_oracle.c's F64 mode in tests/test_oracle_golden.py.
 *
 * Threads: the reference gives a block one process = one core (docs/5.architecture.md:62-68), so nthreads = 1 is the
 * reference-equivalent figure; nthreads > 1 splits the output range (form 1) or the block list (form 2) with OpenMP.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---- form (1): dot product per output, 16 partial sums (what a SIMD dot product does) ----------------------- */
/* x: n input samples (es floats each), hist: M-1 samples before x[0] (NULL = zeros), taps_rev[j] = h[M-1-j] */
long lrb_fir_dot(const float *taps, int M, int input_complex, const float *x, long n, float *y, int nthreads)
{
    const int es = input_complex ? 2 : 1;
    /* the state vector [M-1 zeros | chunk] of firfilter.lua:131-137, kept between calls (the Lua block keeps its Vector too) */
    static float *state = NULL;
    static size_t state_cap = 0;
    const size_t need = (size_t)(M - 1 + n) * es + 16;
    if (need > state_cap) { free(state); state = (float *)malloc(need * sizeof(float)); state_cap = state ? need : 0; }
    float *hd = (float *)malloc(sizeof(float) * es * M);
    if (!state || !hd) { free(hd); return -1; }
    memset(state, 0, sizeof(float) * es * (M - 1));
    memcpy(state + (size_t)es * (M - 1), x, sizeof(float) * es * n);
    for (int j = 0; j < M; j++) for (int c = 0; c < es; c++) hd[j * es + c] = taps[M - 1 - j];
    const int len = M * es;
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
    for (long i = 0; i < n; i++) {
        const float *z = state + (size_t)es * i;
        float acc[16] = {0};
        int k = 0;
        for (; k + 16 <= len; k += 16)
            for (int l = 0; l < 16; l++) acc[l] += z[k + l] * hd[k + l];
        for (; k < len; k++) acc[k & 15] += z[k] * hd[k];
        float e = 0, o = 0;
        for (int l = 0; l < 16; l += 2) { e += acc[l]; o += acc[l + 1]; }
        if (es == 2) { y[2 * i] = e; y[2 * i + 1] = o; } else y[i] = e + o;
    }
    free(hd);
    return n;
}

/* ---- form (2): overlap-save on a Float32 Stockham FFT --------------------------------------------------------- */
typedef struct {
    int N, nstages;
    float *wr, *wi;      /* per radix-4 stage: 3 twiddle rows of n/4 entries each, concatenated */
} lrb_plan;

static lrb_plan *plan_create(int N)
{
    lrb_plan *p = (lrb_plan *)calloc(1, sizeof(*p));
    p->N = N;
    size_t total = 0;
    for (int n = N; n >= 4; n /= 4) total += 3 * (size_t)(n / 4);
    p->wr = (float *)malloc(sizeof(float) * (total + 1));
    p->wi = (float *)malloc(sizeof(float) * (total + 1));
    size_t o = 0;
    for (int n = N; n >= 4; n /= 4) {
        int n1 = n / 4;
        for (int k = 1; k <= 3; k++)
            for (int q = 0; q < n1; q++) {
                double a = -2.0 * M_PI * (double)k * q / n;
                p->wr[o] = (float)cos(a); p->wi[o] = (float)sin(a); o++;
            }
    }
    return p;
}
static void plan_destroy(lrb_plan *p) { if (p) { free(p->wr); free(p->wi); free(p); } }

/* forward DFT (sign = -1) of N points, split arrays; the result is left in (*pr, *pi), which point to either the input
 * or the scratch pair on return.  inverse: call with the twiddles conjugated (conj = 1); no 1/N. */
static void stockham(const lrb_plan *p, int conj, float **pr, float **pi, float **sr, float **si)
{
    float *xr = *pr, *xi = *pi, *yr = *sr, *yi = *si;
    const float cs = conj ? -1.f : 1.f;
    size_t o = 0;
    int n = p->N, s = 1;
    for (; n >= 4; n /= 4, s *= 4) {
        const int n1 = n / 4;
        const float *w1r = p->wr + o, *w1i = p->wi + o, *w2r = w1r + n1, *w2i = w1i + n1, *w3r = w2r + n1, *w3i = w2i + n1;
        o += 3 * (size_t)n1;
        for (int q = 0; q < n1; q++) {
            const float a1r = w1r[q], a1i = cs * w1i[q], a2r = w2r[q], a2i = cs * w2i[q], a3r = w3r[q], a3i = cs * w3i[q];
            const float *x0r = xr + (size_t)s * q, *x0i = xi + (size_t)s * q;
            const float *x1r = x0r + (size_t)s * n1, *x1i = x0i + (size_t)s * n1;
            const float *x2r = x1r + (size_t)s * n1, *x2i = x1i + (size_t)s * n1;
            const float *x3r = x2r + (size_t)s * n1, *x3i = x2i + (size_t)s * n1;
            float *y0r = yr + (size_t)s * 4 * q, *y0i = yi + (size_t)s * 4 * q;
            float *y1r = y0r + s, *y1i = y0i + s, *y2r = y1r + s, *y2i = y1i + s, *y3r = y2r + s, *y3i = y2i + s;
            for (int t = 0; t < s; t++) {
                const float ar = x0r[t], ai = x0i[t], br = x1r[t], bi = x1i[t], cr = x2r[t], ci = x2i[t], dr = x3r[t], di = x3i[t];
                const float apcr = ar + cr, apci = ai + ci, amcr = ar - cr, amci = ai - ci;
                const float bpdr = br + dr, bpdi = bi + di;
                /* -j*(b-d) for the forward transform, +j*(b-d) for the inverse */
                const float jr = cs * (bi - di), ji = -cs * (br - dr);
                y0r[t] = apcr + bpdr; y0i[t] = apci + bpdi;
                const float t1r = amcr + jr, t1i = amci + ji;
                y1r[t] = t1r * a1r - t1i * a1i; y1i[t] = t1r * a1i + t1i * a1r;
                const float t2r = apcr - bpdr, t2i = apci - bpdi;
                y2r[t] = t2r * a2r - t2i * a2i; y2i[t] = t2r * a2i + t2i * a2r;
                const float t3r = amcr - jr, t3i = amci - ji;
                y3r[t] = t3r * a3r - t3i * a3i; y3i[t] = t3r * a3i + t3i * a3r;
            }
        }
        float *tr = xr, *ti = xi; xr = yr; xi = yi; yr = tr; yi = ti;
    }
    if (n == 2) {
        for (int t = 0; t < s; t++) {
            const float ar = xr[t], ai = xi[t], br = xr[t + s], bi = xi[t + s];
            yr[t] = ar + br; yi[t] = ai + bi; yr[t + s] = ar - br; yi[t + s] = ai - bi;
        }
        float *tr = xr, *ti = xi; xr = yr; xi = yi; yr = tr; yi = ti;
    }
    *pr = xr; *pi = xi; *sr = yr; *si = yi;
}

/* y[i] = sum_k h[k] x[i-k], zero history, all n outputs (the reference's arithmetic per block, firfilter.lua:361-398,
 * without its emission framing: the last partial block is zero-padded instead of retained).  taps: M floats, or M {re,im}
 * pairs when taps_complex.  Complex taps need complex input. */
long lrb_fir_overlap_save(const float *taps, int M, int taps_complex, int input_complex, const float *x, long n, float *y, int nthreads)
{
    if (M < 1 || (taps_complex && !input_complex)) return -1;
    const int N = 1 << (int)floor(log(8.0 * M) / log(2.0));      /* firfilter.lua:329 */
    const int L = N - M + 1;                                     /* :330 */
    const int es = input_complex ? 2 : 1;
    lrb_plan *plan = plan_create(N);
    /* taps' DFT (:342-347), 1/N of the inverse transform folded in */
    float *Hr = (float *)calloc(N, sizeof(float)), *Hi = (float *)calloc(N, sizeof(float));
    {
        float *ar = (float *)calloc(N, sizeof(float)), *ai = (float *)calloc(N, sizeof(float));
        float *br = (float *)calloc(N, sizeof(float)), *bi = (float *)calloc(N, sizeof(float));
        for (int i = 0; i < M; i++) { ar[i] = taps_complex ? taps[2 * i] : taps[i]; ai[i] = taps_complex ? taps[2 * i + 1] : 0.f; }
        float *pr = ar, *pi = ai, *sr = br, *si = bi;
        stockham(plan, 0, &pr, &pi, &sr, &si);
        for (int k = 0; k < N; k++) { Hr[k] = pr[k] / N; Hi[k] = pi[k] / N; }
        free(ar); free(ai); free(br); free(bi);
    }
    const long nblocks = (n + L - 1) / L;
#pragma omp parallel num_threads(nthreads) if (nthreads > 1)
    {
        float *buf = (float *)malloc(sizeof(float) * 4 * (size_t)N);
#pragma omp for schedule(static)
        for (long b = 0; b < nblocks; b++) {
            float *pr = buf, *pi = buf + N, *sr = buf + 2 * N, *si = buf + 3 * N;
            const long lo = b * L - (M - 1);                     /* input index of window position 0 */
            for (int k = 0; k < N; k++) {
                long g = lo + k;
                int ok = g >= 0 && g < n;
                pr[k] = ok ? x[g * es] : 0.f;
                pi[k] = (ok && es == 2) ? x[g * es + 1] : 0.f;
            }
            stockham(plan, 0, &pr, &pi, &sr, &si);
            for (int k = 0; k < N; k++) {                        /* :374-376 */
                float a = pr[k] * Hr[k] - pi[k] * Hi[k], c = pr[k] * Hi[k] + pi[k] * Hr[k];
                pr[k] = a; pi[k] = c;
            }
            stockham(plan, 1, &pr, &pi, &sr, &si);
            const long o0 = b * L;
            const int cnt = (int)(o0 + L <= n ? L : n - o0);
            if (es == 2) for (int k = 0; k < cnt; k++) { y[2 * (o0 + k)] = pr[M - 1 + k]; y[2 * (o0 + k) + 1] = pi[M - 1 + k]; }   /* :379 */
            else for (int k = 0; k < cnt; k++) y[o0 + k] = pr[M - 1 + k];
        }
        free(buf);
    }
    free(Hr); free(Hi);
    plan_destroy(plan);
    return n;
}
