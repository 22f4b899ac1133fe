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
#include "kernels_firdecfft.h"
#include "kernels_firdecim.h"
#include "kernels_channelizer.h"
#include "kernels_iir.h"
#include "kernels_agc.h"
#include "kernels_firwin.h"
#include "kernels_firwin2.h"
#include "kernels_firfft4k.h"
#include "kernels_firpols.h"
#include "kernels_firfft64.h"
#include "kernels_interp.h"
#include "kernels_rx.h"

using namespace lrhip;

static int g_launches = 0;   // kernels enqueued since the counter was last cleared (chain diagnostics)
#define LR_LAUNCH_CHECK()                                                                              \
    do {                                                                                               \
        g_launches++;                                                                                  \
        hipError_t e__ = hipGetLastError();                                                            \
        if (e__ != hipSuccess) return set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

#include "stage.h"
#include "stage_fir.h"
#include "stage_elem.h"
#include "stage_iir.h"
#include "stage_firwin.h"
#include "stage_spectrum.h"
#include "stage_elem2.h"
#include "stage_resample.h"
#include "stage_elem3.h"
#include "stage_rx.h"
#include "chain.h"

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

int lrhip_device(void)
{
    const Context &c = ctx();
    return (c.ready && c.pid == (long)getpid()) ? c.device : -1;
}

int lrhip_device_count(void)
{
    if (ctx().ready && ctx().pid != (long)getpid()) return ensure_init();      // forked after the parent initialised the device: the message, not a hang
    int count = 0;
    LR_HIP(hipGetDeviceCount(&count));
    return count;
}

int lrhip_set_stream(void *hip_stream)
{
    if (ensure_init()) return -1;
    // (void *)1 = hipStreamLegacy: the default ("null") stream, whose handle inside the library is plain 0 - every HIP entry point takes that
    hipStream_t next = hip_stream == (void *)1 ? (hipStream_t) nullptr : hip_stream ? (hipStream_t)hip_stream : ctx().own_stream, prev = ctx().stream;
    if (next != prev) {
        // work already queued on the previous stream (reset() memsets, earlier chunks that carried state forward) is ordered
        // before everything that follows on the new one
        hipEvent_t ev = nullptr;
        LR_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t e1 = hipEventRecord(ev, prev), e2 = e1 == hipSuccess ? hipStreamWaitEvent(next, ev, 0) : e1;
        (void)hipEventDestroy(ev);
        if (e2 != hipSuccess) return set_error("set_stream: ordering the streams failed: %s", hipGetErrorString(e2));
    }
    ctx().stream = next;
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
        const double p16 = P == 1 ? T[0] : 0.0;                     // first order: A^LC is the scalar p^16
        for (int k = 0; k <= 8; k++) {
            for (int i = 0; i < P * P; i++) { tpow[(size_t)k * P * P + i] = (float)T[i]; tpow64[(size_t)k * P * P + i] = T[i]; }
            if (k == 8) q->Ttile = T;
            matmul(T, T, T, P);
        }
        if (P == 1 && std::fabs(p16) < 1.0 && p16 != 0.0) {
            // chunks of 16 samples after which a zero start has decayed to 1e-12: the partial warm-up of the one-shot launch
            const int wc = (int)std::ceil(std::log(1e-12) / std::log(std::fabs(p16)));
            if (wc >= 1 && wc <= 128) q->warm_chunks = wc;
        } else if (P == 1 && p16 == 0.0) {
            q->warm_chunks = 1;
        }
        // (orders 2-4 were given the same one-shot launch in round 3 - 4 chunks of warm-up instead of a whole tile per 8 - and lost: 0.417 against 0.372 ms
        // for the suite's 3-feedback-tap entry on ComplexFloat32, same box: every workgroup then runs the 18-barrier block scan twice per emitted tile)
        // Round 3, on top of the two-barrier wave scan: a workgroup per tile still loses (0.267 against 0.232 ms), but the partial warm-up tile itself pays
        // at the old 8 tiles per workgroup - 0.213 (stage_iir.h has the table).
        if (P >= 2 && P <= 4) {
            std::vector<double> A16(tpow64.begin(), tpow64.begin() + P * P), M = A16;
            for (int wc = 1; wc <= 128 && !q->warm_chunks; wc++) {
                double mx = 0.0;
                for (double v : M) mx = std::fabs(v) > mx ? std::fabs(v) : mx;
                if (std::isfinite(mx) && mx < 1e-12) q->warm_chunks = wc;
                matmul(M, A16, M, P);
            }
        }
        if (P == 1) {                                               // p^(16 (l + 1)), l < 64: the per-lane powers of the single-launch kernel's wave scan
            double acc = 1.0;
            for (int l = 0; l < 64; l++) { acc *= p16; tpow.push_back((float)acc); }
        } else if (P <= 4) {                                        // the same for orders 2-4: the matrices A^(16 (l + 1))
            std::vector<double> A16(tpow64.begin(), tpow64.begin() + P * P), M = A16;
            for (int l = 0; l < 64; l++) {
                for (int i = 0; i < P * P; i++) tpow.push_back((float)M[i]);
                matmul(M, A16, M, P);
            }
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

lrhip_stage_t *lrhip_format_pack_create(const char *format, int complex_in)
{
    FormatStage *q = (FormatStage *)lrhip_format_convert_create(format, complex_in);
    if (!q) return nullptr;
    q->pack = true;
    std::swap(q->in_size, q->out_size);
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
                        [&](const void *di, unsigned long n, void *dout, unsigned long cap) { return q->run(di, n, dout, cap); }, q->align(), q->direct_io_ok());
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
lrhip_chain_t *lrhip_chain_create(lrhip_stage_t **stages, unsigned nstages) { return lrhip_chain_create_ex(stages, nstages, 0u); }

lrhip_chain_t *lrhip_chain_create_ex(lrhip_stage_t **stages, unsigned nstages, unsigned flags)
{
    if (!stages || nstages < 1) { set_error("chain: need at least one stage"); return nullptr; }
    if (flags & ~(unsigned)(LRHIP_CHAIN_EXACT_ROTATOR | LRHIP_CHAIN_NO_POLYPHASE_TAIL | LRHIP_CHAIN_NO_FUSION | LRHIP_CHAIN_NO_SINGLE_LAUNCH)) {
        set_error("chain: unknown flag bits 0x%x", flags);
        return nullptr;
    }
    // the numerical contract is a property of the chain (flags); the environment variables of the same names stay as process-wide
    // overrides for A/B runs
    const bool exact_rotator = (flags & LRHIP_CHAIN_EXACT_ROTATOR) != 0;
    const bool no_fusion = (flags & LRHIP_CHAIN_NO_FUSION) != 0;
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
    c->flags = flags;
    // Downsampler(1) is the identity (downsampler.lua:45-56 with factor 1 copies every sample; its index stays 0): it gets no launch of its own
    // (a Tuner / Decimator built with decimation 1) - unless it is the whole chain
    std::vector<lrhip_stage_t *> kept;
    if (!no_fusion && nstages > 1) {
        for (unsigned k = 0; k < nstages; k++) {
            DownsamplerStage *d1 = dynamic_cast<DownsamplerStage *>(stages[k]);
            if (!(d1 && d1->factor == 1)) kept.push_back(stages[k]);
        }
        if (!kept.empty() && kept.size() < nstages) { stages = kept.data(); nstages = (unsigned)kept.size(); }
    }
    unsigned i = 0;
    while (i < nstages) {
        if (no_fusion) {                                    // every block runs its own kernels; edges stay on the device
            c->ops.push_back({stages[i], false});
            i++;
            continue;
        }
        // fusion: fir fir ... (ComplexFloat32 stream, every one of them on the overlap-save arithmetic: "fast", or left to the library)  ->  ONE filter with the convolved taps
        // while they fit the 4096-point kernel (kernels_firfft4k.h: 1 281 taps).  A chain of filters is a filter; overlap-save promises 1e-6 of the exact
        // result, not the bits of a particular block size, and the merged filter rounds once where the chain rounds per stage.  The reference suite's first
        // entry (five 256-tap filters back to back, luaradio_benchmark.lua:17-37) = 1 276 taps = one launch instead of five.
        {
            static const bool no_cascade = getenv("LRHIP_NO_FIR_CASCADE") != nullptr;      // A/B knob
            auto mergeable = [](lrhip_stage_t *st) -> FirStage * {
                FirStage *f = dynamic_cast<FirStage *>(st);
                return (f && f->S == 2 && f->D == 1 && !f->rot && !f->pre_disc && !f->post_disc && !f->use_fft && f->fft_arith && (f->mode_req == 2 || f->mode_req == 3)) ? f : nullptr;
            };
            // (LRHIP_CHAIN_NO_POLYPHASE_TAIL = "keep every block's own arithmetic": the same kind of identity, the same switch)
            FirStage *f0 = (no_cascade || (flags & LRHIP_CHAIN_NO_POLYPHASE_TAIL)) ? nullptr : mergeable(stages[i]);
            if (f0 && i + 1 < nstages && mergeable(stages[i + 1])) {
                // taps in natural order, double, complex (real taps: zero imaginary parts)
                auto natural = [](const FirStage *f) {
                    const int ts = f->taps_complex ? 2 : 1;
                    std::vector<double> h((size_t)2 * f->M, 0.0);
                    for (int t = 0; t < f->M; t++) {
                        h[2 * t] = f->taps_rev[(size_t)(f->M - 1 - t) * ts];
                        if (ts == 2) h[2 * t + 1] = f->taps_rev[(size_t)(f->M - 1 - t) * ts + 1];
                    }
                    return h;
                };
                std::vector<double> acc = natural(f0);
                bool cplx = f0->taps_complex;
                unsigned k = i + 1;
                while (k < nstages) {
                    FirStage *fk = mergeable(stages[k]);
                    if (!fk || acc.size() / 2 + (size_t)fk->M - 1 > 1281) break;
                    const std::vector<double> h = natural(fk);
                    std::vector<double> out(acc.size() + h.size() - 2, 0.0);
                    for (size_t a = 0; a < acc.size() / 2; a++)
                        for (size_t b = 0; b < h.size() / 2; b++) {
                            out[2 * (a + b)] += acc[2 * a] * h[2 * b] - acc[2 * a + 1] * h[2 * b + 1];
                            out[2 * (a + b) + 1] += acc[2 * a] * h[2 * b + 1] + acc[2 * a + 1] * h[2 * b];
                        }
                    acc.swap(out);
                    cplx = cplx || fk->taps_complex;
                    k++;
                }
                if (k > i + 1) {
                    const unsigned Mm = (unsigned)(acc.size() / 2);
                    std::vector<float> taps((size_t)Mm * (cplx ? 2 : 1));
                    for (unsigned t = 0; t < Mm; t++) {
                        if (cplx) { taps[2 * t] = (float)acc[2 * t]; taps[2 * t + 1] = (float)acc[2 * t + 1]; }
                        else taps[t] = (float)acc[2 * t];
                    }
                    FirStage *merged = fir_build(taps.data(), Mm, cplx ? 1 : 0, 1, 1, 2, false, 0.0);
                    if (!merged) return nullptr;
                    c->ops.push_back({merged, true});
                    i = k;
                    continue;
                }
            }
        }
        // fusion: [multiplyconstant(real)] upsampler fir(real taps, plain) [downsampler]  ->  one polyphase resampling launch
        {
            static const bool no_resample_fusion = getenv("LRHIP_NO_RESAMPLE_FUSION") != nullptr;      // A/B knob
            unsigned k = i;
            MulConstStage *mc = dynamic_cast<MulConstStage *>(stages[k]);
            if (mc && mc->mode <= 1) k++; else mc = nullptr;
            UpsamplerStage *up = k < nstages ? dynamic_cast<UpsamplerStage *>(stages[k]) : nullptr;
            FirStage *rf = (up && k + 1 < nstages) ? dynamic_cast<FirStage *>(stages[k + 1]) : nullptr;
            // (a filter whose arithmetic was left to the library - use_fft nil, mode 3 - may have picked overlap-save for itself: the polyphase form wins)
            if (!no_resample_fusion && up && rf && !rf->taps_complex && !rf->use_fft && (!rf->fft_arith || rf->mode_req == 3) && rf->D == 1 && !rf->rot && !rf->pre_disc) {
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
                    static const bool no_interp_win = getenv("LRHIP_NO_INTERP_WIN") != nullptr;      // A/B knob: one output per thread (fir_resample_kernel)
                    if (!no_interp_win && q->S == 2 && q->M == 128 && ((D == 1 && L >= 2 && L <= 5) || ResampleStage::rational_supported(L, D))) {
                        // tap table of fir_interp_kernel: ttab[s * LP + p] = h[p + (J - 1 - s) L], step 0 = the oldest sample
                        const int J = (q->M + L - 1) / L, LP = (L + 3) & ~3;
                        std::vector<float> tt((size_t)J * LP, 0.f);
                        for (int st = 0; st < J; st++)
                            for (int p = 0; p < L; p++) {
                                const int t = p + (J - 1 - st) * L;
                                if (t < q->M) tt[(size_t)st * LP + p] = taps[t];
                            }
                        if (upload(q->d_ttab, tt.data(), tt.size() * sizeof(float))) return nullptr;
                        q->interp_J = J;
                    }
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
        // a rotator in front of a filter that does NOT decimate stays a launch of its own: the rotating Toeplitz kernel at D = 1 has no register room left
        // (112-296 bytes of scratch) and runs 2^26 samples in 0.62 ms, against 0.17 (rotator) + 0.21 (overlap-save) / 0.39 (direct form) for the pair
        static const bool rot_fir_d1 = getenv("LRHIP_ROT_FIR_FUSE_D1") != nullptr;      // A/B knob: fuse it all the same
        if (rot && fusable_fir && !ds && !rot_fir_d1) {
            c->ops.push_back({stages[i], false});
            i++;
            continue;
        }
        // ... [discriminator]: runs as the epilogue of the persistent kernel (ComplexFloat32 outputs never reach HBM)
        static const bool no_disc_fusion = getenv("LRHIP_NO_DISC_FUSION") != nullptr;      // A/B knob
        unsigned after = j + 1 + (ds ? 1 : 0);
        // a filter that asked for overlap-save arithmetic keeps it when fused with a downsampler (polyphase FFT form, kernels_firdecfft.h)
        // (only when the caller asked for it explicitly, mode 2: an automatic filter takes the direct form when it decimates - faster here)
        const bool want_fft = fusable_fir && fir->fft_arith && fir->mode_req == 2 && ds && FirStage::decfft_supported((unsigned)ds->factor, fir->M, fir->S);
        // overlap-save arithmetic the caller pinned (mode 2) stays; an automatic filter (mode 3) that gets a rotator or a downsampler
        // fused takes the direct form, and then also the discriminator epilogue
        const bool fft_pinned = fusable_fir && fir->fft_arith && !(fir->mode_req == 3 && (rot || ds));
        FmDiscrimStage *dsc_after = (!no_disc_fusion && fusable_fir && fir->S == 2 && !fir->taps_complex && (!fft_pinned || want_fft) && after < nstages)
                                        ? dynamic_cast<FmDiscrimStage *>(stages[after]) : nullptr;
        if (fusable_fir && (rot || ds || dsc_after)) {
            unsigned D = ds ? (unsigned)ds->factor : 1;
            int ts = fir->taps_complex ? 2 : 1;
            std::vector<float> taps((size_t)fir->M * ts);
            for (int t = 0; t < fir->M; t++)
                for (int cc = 0; cc < ts; cc++) taps[(size_t)t * ts + cc] = fir->taps_rev[(size_t)(fir->M - 1 - t) * ts + cc];
            bool want_rot = rot && fir->S == 2;
            FirStage *fused = nullptr;
            if (want_fft || FirStage::mfma_supported_decim(D) || !rot || (fir->S == 2 && !fir->taps_complex && fir->M + 255 <= DECIM_SPAN_MAX))
                fused = fir_build(taps.data(), (unsigned)fir->M, fir->taps_complex, fir->S == 2, D, want_fft ? 2 : 0, want_rot, want_rot ? rot->omega : 0.0);
            if (fused && rot && !want_rot) { delete fused; fused = nullptr; }
            bool with_disc = fused && dsc_after && fused->can_post_disc();
            // Round 5 (the receivers OFF the stock shape): the Toeplitz kernel has its discriminator epilogue at decimation 5 / 128 taps only, so
            // an FM receiver at another input rate (Tuner /4, /8) ran its tuner, the discriminator and their fix-ups as separate launches.  The polyphase-FFT
            // decimator has the epilogue at every decimation it supports: an AUTOMATIC filter (use_fft nil / "auto": the caller left the arithmetic to the
            // library) that would otherwise lose the epilogue takes that form - tuner + discriminator stay one launch.  Pinned direct-form filters
            // (use_fft = false), exact chains and unsupported decimations keep what they had.
            static const bool no_auto_decfft = getenv("LRHIP_NO_AUTO_DECFFT") != nullptr;      // A/B knob
            // Measured on 2^26 RF samples, one box (profiles/r05_receiver_shapes.txt): Tuner /4 0.385 -> 0.236 ms (5 -> 2 launches); Tuner /8 0.291 -> 0.367 ms, the
            // polyphase-FFT form with eight branches is the slower tuner there - so decimation 4 only (2 is unmeasured and stays as it was).
            if (!no_auto_decfft && fused && dsc_after && !with_disc && !want_fft && fir->mode_req == 3 && ds && D == 4 && !exact_rotator && fir->S == 2 && !fir->taps_complex &&
                FirStage::decfft_supported(D, fir->M, fir->S)) {
                FirStage *alt = fir_build(taps.data(), (unsigned)fir->M, fir->taps_complex, true, D, 2, want_rot, want_rot ? rot->omega : 0.0);
                if (alt && alt->decfft && alt->can_post_disc()) {
                    delete fused;
                    fused = alt;
                    with_disc = true;
                } else {
                    delete alt;
                }
            }
            if (fused && !rot && !ds && !with_disc) { delete fused; fused = nullptr; }      // nothing was fused
            if (fused) {
                if (exact_rotator) fused->rel_rot = false;     // block-of-8 staging with the stand-alone rotator's phasors: fused == unfused bit for bit
                if (with_disc) {
                    fused->post_disc = true;
                    fused->disc_gain = dsc_after->gain;
                    fused->out_size = 4;                // ComplexFloat32 in, Float32 out
                    if (fused->reset()) { delete fused; return nullptr; }
                }
                // ... [complex -> real element-wise block] behind an LDS-staged decimator (decimations without a Toeplitz instantiation: the AM / SSB / NBFM
                // receivers' Tuner(…, 50)): ComplexMagnitude / ComplexPhase / ComplexToReal / ComplexToImag run on the accumulators, one launch less and the
                // ComplexFloat32 tuner output never reaches HBM; the same Float32 operation on the same Float32 filter outputs = the unfused bits
                static const bool no_unary_fold = getenv("LRHIP_NO_UNARY_FOLD") != nullptr;      // A/B knob
                UnaryStage *un = (!no_unary_fold && !with_disc && after < nstages) ? dynamic_cast<UnaryStage *>(stages[after]) : nullptr;
                const bool with_unary = un && fused->can_post_unary() && (un->op == UN_CMAG || un->op == UN_CPHASE || un->op == UN_CREAL || un->op == UN_CIMAG);
                if (with_unary) {
                    fused->post_unary = 1 + un->op;
                    fused->out_size = 4;
                }
                c->ops.push_back({fused, true});
                i = j + (ds ? 2 : 1) + (with_disc ? 1 : 0) + (with_unary ? 1 : 0);
                continue;
            }
        }
        // fusion: fir(Float32 stream, real taps) -> first-order iir -> [downsampler]: one launch on the register-window kernel, the
        // recurrence runs on the filter's accumulators (kernels_firwin.h).  A filter pinned to overlap-save (use_fft 1 / 2) keeps it.
        {
            static const bool no_fir_iir_fusion = getenv("LRHIP_NO_FIR_IIR_FUSION") != nullptr;      // A/B knob
            FirStage *f1 = dynamic_cast<FirStage *>(stages[i]);
            IirStage *ii = (f1 && i + 1 < nstages) ? dynamic_cast<IirStage *>(stages[i + 1]) : nullptr;
            if (!no_fir_iir_fusion && ii && f1->S == 1 && !f1->taps_complex && f1->D == 1 && !f1->rot && !f1->pre_disc && !f1->post_disc && !f1->use_fft &&
                (f1->mode_req == 0 || f1->mode_req == 3) && FirWinRealStage::supported_taps(f1->M) && ii->S == 1 && ii->scan && ii->P == 1 && ii->nb <= 2 &&
                ii->D == 1 && FirWinRealStage::warm_waves_for((double)ii->seq.a[1] / (double)ii->seq.a[0]) > 0) {
                DownsamplerStage *ds3 = i + 2 < nstages ? dynamic_cast<DownsamplerStage *>(stages[i + 2]) : nullptr;
                std::vector<float> taps((size_t)f1->M);
                for (int t = 0; t < f1->M; t++) taps[t] = f1->taps_rev[f1->M - 1 - t];
                // With a downsampler behind the recurrence only every D-th output of it is kept, and
                //     y[n] = u[n] + p y[n-1]   =>   y[n] = sum_{k<D} p^k u[n-k]  +  p^D y[n-D]           (exact identity, p = -a1/a0)
                // so the kept samples are a DECIMATING filter  g = h * b * (1, p, .., p^(D-1))  followed by the first-order recurrence with
                // pole p^D at the LOW rate: 1/D of the multiply-adds, and the high-rate audio is never computed.  Same transfer function,
                // different rounding: <= 1e-6 of the reference's arithmetic (the bar of the IIR blocks), not bit-identical to it.
                static const bool no_polyphase_tail_env = getenv("LRHIP_NO_POLYPHASE_TAIL") != nullptr;      // A/B knob
                const bool no_polyphase_tail = no_polyphase_tail_env || (flags & LRHIP_CHAIN_NO_POLYPHASE_TAIL);
                if (!no_polyphase_tail && ds3 && ds3->factor >= 2 && ds3->factor <= 16) {
                    const int D = (int)ds3->factor, nbb = ii->nb;
                    const double a0 = (double)ii->seq.a[0];
                    const double pp = -(double)(float)((double)ii->seq.a[1] / a0);      // the pole as the IIR kernels round it (IirCoeffs)
                    std::vector<double> cpoly((size_t)(nbb + D - 1), 0.0);
                    double pk = 1.0;
                    for (int k = 0; k < D; k++, pk *= pp)
                        for (int jb = 0; jb < nbb; jb++) cpoly[(size_t)(k + jb)] += pk * (double)(float)((double)ii->seq.b[jb] / a0);
                    const int Mg0 = f1->M + nbb + D - 2, Mg = (Mg0 + 3) & ~3;            // zero taps at the far end: a multiple of four for the window kernel
                    std::vector<float> g((size_t)Mg, 0.f);
                    for (int t = 0; t < Mg0; t++) {
                        double acc = 0.0;
                        for (int c2 = 0; c2 < (int)cpoly.size(); c2++)
                            if (t - c2 >= 0 && t - c2 < f1->M) acc += (double)taps[(size_t)(t - c2)] * cpoly[(size_t)c2];
                        g[(size_t)t] = (float)acc;
                    }
                    FirStage *fd = fir_build(g.data(), (unsigned)Mg, 0, 0, (unsigned)D, 0, false, 0.0);
                    const float b2[1] = {1.0f}, a2[2] = {1.0f, (float)(-pk)};          // pk = p^D after the loop
                    // the low-rate recurrence runs on the filter's accumulators when the window kernel takes this shape ...
                    if (fd && fd->fuse_iir1(1.0, pk) == 0) {
                        c->ops.push_back({fd, true});
                        i += 3;
                        continue;
                    }
                    // ... and as its own launch otherwise - if the pole p^D survives rounding to ONE Float32: that moves the DC gain by
                    // 2^-24 q / (1 - q), held below 3e-7 here (the fused form above carries the pole as a Float32 pair instead)
                    const bool pole_ok = std::fabs(pk) * 5.96e-8 <= 3e-7 * (1.0 - std::fabs(pk));
                    lrhip_stage_t *i2 = (fd && pole_ok) ? lrhip_iir_create(b2, 1, a2, 2, 0) : nullptr;
                    if (fd && i2) {
                        c->ops.push_back({fd, true});
                        c->ops.push_back({i2, true});
                        i += 3;
                        continue;
                    }
                    delete fd;
                }
                // without a (small) decimation the recurrence can run on the accumulators of the D = 1 window kernel: one launch, but slower on
                // MI355X than overlap-save filter + scan (0.068 against 0.060 ms on the WBFM audio tail) - opt-in, LRHIP_FIR_IIR_WIN=1
                static const bool fir_iir_win = getenv("LRHIP_FIR_IIR_WIN") != nullptr;
                FirWinRealStage *fw = fir_iir_win ? firwin_real_build(taps.data(), f1->M, ii->seq.b, ii->nb, ii->seq.a, ii->na, ds3 ? ds3->factor : 1) : nullptr;
                if (fw) {
                    c->ops.push_back({fw, true});
                    i += ds3 ? 3 : 2;
                    continue;
                }
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
    // a Toeplitz tuner + discriminator stage directly in front of the pair-mode window filter hands its wave-boundary fix-up to that filter's
    // staging (kernels_firwin2.h FwcParams::fix_edge): the WBFM receiver is then two launches
    static const bool no_fixup_fold = getenv("LRHIP_NO_FIXUP_FOLD") != nullptr;      // A/B knob
    for (size_t k = 0; k + 1 < c->ops.size() && !no_fixup_fold; k++) {
        FirStage *prod = c->ops[k].owned ? dynamic_cast<FirStage *>(c->ops[k].stage) : nullptr;
        FirStage *cons = c->ops[k + 1].owned ? dynamic_cast<FirStage *>(c->ops[k + 1].stage) : nullptr;
        if (prod && cons && prod->post_disc && prod->ksteps != 0 && !prod->decfft && !prod->win_cplx_ok() && cons->iir_fused && cons->win_pair_ok()) {
            prod->defer_fixup = true;
            cons->fix_src = prod;
        }
    }
    // ... and the two become ONE stage whose run() is a single launch (kernels_rx.h): the discriminator stream stays in LDS.  The stage keeps
    // both FirStages and their state formats, so the two-launch form remains its fallback (and what LRHIP_CHAIN_NO_SINGLE_LAUNCH selects)
    for (size_t k = 0; k + 1 < c->ops.size(); k++) {
        FirStage *prod = c->ops[k].owned ? dynamic_cast<FirStage *>(c->ops[k].stage) : nullptr;
        FirStage *cons = c->ops[k + 1].owned ? dynamic_cast<FirStage *>(c->ops[k + 1].stage) : nullptr;
        if (!RxStage::shapes_ok(prod, cons)) continue;
        std::unique_ptr<RxStage> rx(new (std::nothrow) RxStage());
        if (!rx) { set_error("out of memory"); return nullptr; }
        rx->A.reset(prod);
        rx->B.reset(cons);
        c->ops[k] = {nullptr, false};                        // ownership has moved: the unique_ptrs delete them
        c->ops.erase(c->ops.begin() + (long)k + 1);
        rx->single_launch = !(flags & LRHIP_CHAIN_NO_SINGLE_LAUNCH);
        // IQFileSource's u8 / s8 / s16le records directly in front of it: the receiver reads them itself (2 or 4 bytes per RF sample instead of a
        // conversion pass and an 8-byte read); every other format keeps its conversion launch
        static const bool no_u8_fold = getenv("LRHIP_RX_NO_U8_FOLD") != nullptr;      // A/B knob
        FormatStage *fs = (k > 0 && !no_u8_fold && !(flags & LRHIP_CHAIN_NO_FUSION)) ? dynamic_cast<FormatStage *>(c->ops[k - 1].stage) : nullptr;
        const int fcls = fs ? kFormats[fs->fmt].cls : -1;
        const bool fold = fs && !fs->pack && fs->scalars == 2 && !kFormats[fs->fmt].swap && (fcls == 0 || fcls == 1 || fcls == 3) && !c->ops[k - 1].owned;
        if (fold) { rx->in_u8 = true; rx->fmt = fs; rx->in_fmt = fcls == 0 ? RX_FMT_U8 : fcls == 1 ? RX_FMT_S8 : RX_FMT_S16LE; }
        if (rx->prepare()) { c->ops.erase(c->ops.begin() + (long)k); return nullptr; }
        c->ops[k] = {rx.release(), true};
        if (fold) { c->ops.erase(c->ops.begin() + (long)k - 1); k--; }
    }
    // ... and the same records in front of a plain fused Tuner or Decimator ([rotator +] 128-tap filter + decimation 5, no discriminator: the branch of a
    // fan-out, TunerBlock / DecimatorBlock on its own; or the LDS-staged decimators of the AM / SSB / NBFM receivers, decimation 9 .. 50): folded into the
    // kernel's staging (kernels_fir.h FMT)
    {
        static const bool no_tuner_fold = getenv("LRHIP_TUNER_NO_RAW_FOLD") != nullptr;      // A/B knob
        for (size_t k = 0; k + 1 < c->ops.size() && !no_tuner_fold && !(flags & LRHIP_CHAIN_NO_FUSION); k++) {
            FormatStage *fs = c->ops[k].owned ? nullptr : dynamic_cast<FormatStage *>(c->ops[k].stage);
            FirStage *tf = c->ops[k + 1].owned ? dynamic_cast<FirStage *>(c->ops[k + 1].stage) : nullptr;
            if (!fs || !tf || fs->pack || fs->scalars != 2 || kFormats[fs->fmt].swap) continue;
            const int fcls = kFormats[fs->fmt].cls;
            if (fcls != 0 && fcls != 1 && fcls != 3) continue;
            if (tf->post_disc || tf->pre_disc || tf->use_fft || tf->decfft || tf->fft_arith || tf->taps_complex || tf->S != 2 || tf->win_cplx_ok()) continue;
            if (!((tf->D == 5 && tf->ksteps == 51) || (tf->ksteps == 0 && tf->D > 1 && tf->decim_lds_ok()))) continue;
            tf->in_fmt = fcls == 0 ? RX_FMT_U8 : fcls == 1 ? RX_FMT_S8 : RX_FMT_S16LE;
            tf->fmt_stage = fs;
            tf->in_size = fs->in_size;
            c->ops.erase(c->ops.begin() + (long)k);
        }
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

int lrhip_chain_reset(lrhip_chain_t *c)
{
    if (!c) return set_error("null chain");
    if (c->inflight) return set_error("chain reset: %u chunks still in flight", c->inflight);
    c->fill = 0;
    c->discard_in = 0;
    if (ctx().ready) LR_HIP(hipStreamSynchronize(ctx().stream));
    for (auto &o : c->ops)
        if (o.stage->reset()) return -1;
    return 0;
}

unsigned long lrhip_chain_max_output(const lrhip_chain_t *c, unsigned long n_in)
{
    if (!c) return 0;
    unsigned long n = n_in;
    for (auto &o : c->ops) n = o.stage->max_output(n);
    return n;
}

static unsigned long chain_output_bound(const lrhip_chain *c, unsigned long n);

long lrhip_chain_execute_device(lrhip_chain_t *c, const void *in_dev, unsigned long n_in, void *out_dev, unsigned long out_capacity)
{
    if (!c) return set_error("null chain");
    g_launches = 0;
    if (c->discard_in && n_in) {
        // lrhip_chain_start_at(): the first samples after the seek are the partition's replayed halo - state only, their output is dropped
        const unsigned long k = c->discard_in < n_in ? (unsigned long)c->discard_in : n_in;
        const unsigned long bound = chain_output_bound(c, k);
        if (c->d_discard.reserve((size_t)bound * c->ops.back().stage->out_size + 16)) return -1;
        c->discard_in -= k;
        const unsigned long long left = c->discard_in;
        c->discard_in = 0;
        long rc = lrhip_chain_execute_device(c, in_dev, k, c->d_discard.p, bound);
        c->discard_in = left;
        if (rc < 0) return rc;
        in_dev = (const char *)in_dev + (size_t)k * c->ops.front().stage->in_size;
        n_in -= k;
        if (!n_in) return 0;
    }
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
                        [&](const void *di, unsigned long n, void *dout, unsigned long cap) { return lrhip_chain_execute_device(c, di, n, dout, cap); },
                        lrhip_chain_shard_align(c), c->ops.front().stage->direct_io_ok() && c->ops.back().stage->direct_io_ok());
}

int lrhip_chain_last_launches(const lrhip_chain_t *c) { return c ? c->last_launches : set_error("null chain"); }

int lrhip_chain_in_flight(const lrhip_chain_t *c) { return c ? (int)c->inflight : set_error("null chain"); }

// Upper bound on what a stage emits for n inputs in ANY carried state (max_output() is evaluated in the current state).
static unsigned long stage_output_bound(const lrhip_stage *s, unsigned long n)
{
    unsigned long m = s->max_output(n);
    if (const FirStage *f = dynamic_cast<const FirStage *>(s))
        if (f->use_fft) m = n + (unsigned long)f->L;            // block-emission framing: up to L - 1 retained samples come out as well
    return (m > n ? m : n) + 64;
}
static unsigned long chain_output_bound(const lrhip_chain *c, unsigned long n)
{
    for (auto &o : c->ops) n = stage_output_bound(o.stage, n);
    return n;
}

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
    // bound on the output of one chunk WHATEVER the carried state: rate-changing stages may emit one extra sample, a FIR with the
    // reference's block-emission framing up to a whole retained block (L - 1 samples) more than its stateless max_output() says
    unsigned long max_out = chain_output_bound(c, max_chunk);
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
        nmax = stage_output_bound(c->ops[k].stage, nmax);
        if (c->edges[k]->reserve((size_t)nmax * c->ops[k].stage->out_size + 16)) return -1;
    }
    c->ring_chunk = max_chunk;
    c->ring_out_cap = max_out;
    c->fill = 0;
    c->head = 0;
    c->inflight = 0;
    return 0;
}

static long chain_submit(lrhip_chain_t *c, const void *in_host, unsigned long n_in, bool allow_inplace);
long lrhip_chain_submit(lrhip_chain_t *c, const void *in_host, unsigned long n_in) { return chain_submit(c, in_host, n_in, true); }
static long chain_submit(lrhip_chain_t *c, const void *in_host, unsigned long n_in, bool allow_inplace)
{
    if (!c) return set_error("null chain");
    if (c->ring.empty()) return set_error("chain has no ring: call lrhip_chain_set_ring first");
    if (n_in > c->ring_chunk) return set_error("chunk of %lu samples exceeds the ring's max_chunk %lu", n_in, c->ring_chunk);
    if (c->inflight == c->ring.size()) return set_error("ring full: collect a chunk first (%u in flight)", c->inflight);
    if (c->fill) return set_error("chain has %lu pushed samples pending: flush before mixing submit() with push()", c->fill);
    lrhip_chain::Slot &sl = *c->ring[c->head];
    if (n_in && !in_host) return set_error("null input buffer");
    int in_size = c->ops.front().stage->in_size, out_size = c->ops.back().stage->out_size;
    size_t bytes = (size_t)n_in * in_size;
    const void *zin = nullptr;           // input read in place across the link
    // the slot was collected (ev_out waited) before it can be reused, so its buffers are free on host and device
    if (bytes) {
        // lrhip_chain_ring_input(): the caller filled the slot itself.  A vector inside a range registered with lrhip_host_register() is DMA'd from where it
        // lies (a recording mmap()ed and registered once: the page cache is the staging buffer); the caller keeps it valid until the batch is collected.
        const bool direct = in_host != sl.h_in.p && host_ranges().has(in_host, bytes);
        if (in_host != sl.h_in.p && !direct) host_copy(sl.h_in.p, in_host, bytes);
        // round 5: a chain whose first stage reads its input once takes the slot's pinned input (or the caller's registered vector) AS its input - the
        // kernel loads the records across the link itself; no copy engine, no event hop in front of the launch.  Same box, u8 records -> receiver: pushes of
        // 131 072 records 7.4 -> 9.2 GS/s, read(2) into the slot 9.6 -> 10.6; NOT for lrhip_chain_submit_fd, whose copy threads are the bound either way
        // (15.2 -> 14.3 GS/s: the copy engine costs the GPU nothing while they read).  LRHIP_RING_DIRECT=0: always the copy (A/B)
        static const bool ring_direct = !getenv("LRHIP_RING_DIRECT") || atoi(getenv("LRHIP_RING_DIRECT")) != 0;
        if (ring_direct && allow_inplace && c->ops.front().stage->direct_io_ok()) zin = direct ? host_ranges().device_ptr(in_host, bytes) : sl.h_in.p;
        if (!zin) LR_HIP(hipMemcpyAsync(sl.d_in.p, direct ? in_host : sl.h_in.p, bytes, hipMemcpyHostToDevice, c->s_in));
    }
    if (!zin) {
        LR_HIP(hipEventRecord(sl.ev_in, c->s_in));
        LR_HIP(hipStreamWaitEvent(ctx().stream, sl.ev_in, 0));
    }
    unsigned long cap = (unsigned long)(sl.d_out.cap / out_size);
    long n_out = lrhip_chain_execute_device(c, zin ? zin : sl.d_in.p, n_in, sl.d_out.p, cap);
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

// File-fed chains (radio/blocks/sources/iqfile.lua:82-96 without the interpreter in the data path): the raw records of a regular file go from the page cache
// straight into the pinned input of the next ring slot - positional reads split over the library's copy threads - and the slot is submitted.
long lrhip_chain_submit_fd(lrhip_chain_t *c, int fd, unsigned long long offset, unsigned long max_in)
{
    if (!c) return set_error("null chain");
    if (c->ring.empty()) return set_error("chain has no ring: call lrhip_chain_set_ring first");
    if (c->inflight == c->ring.size()) { set_error("ring full: collect a chunk first (%u in flight)", c->inflight); return -3; }
    if (c->fill) return set_error("chain has %lu pushed samples pending: flush before mixing submit() with push()", c->fill);
    struct stat st;
    if (fstat(fd, &st) != 0) return set_error("fstat(%d): %s", fd, strerror(errno));
    if (!S_ISREG(st.st_mode)) { set_error("lrhip_chain_submit_fd: descriptor %d is not a regular file (read() into lrhip_chain_ring_input() instead)", fd); return -4; }
    const int in_size = c->ops.front().stage->in_size;
    unsigned long long avail = (unsigned long long)st.st_size > offset ? ((unsigned long long)st.st_size - offset) / (unsigned)in_size : 0;
    unsigned long n = max_in < c->ring_chunk ? max_in : c->ring_chunk;
    if (avail < n) n = (unsigned long)avail;
    if (!n) return 0;                                        // end of the file (or less than one whole record left)
    lrhip_chain::Slot &sl = *c->ring[c->head];
    const int err = host_pread(sl.h_in.p, fd, (long long)offset, (size_t)n * in_size);
    if (err) return set_error("pread(%d): %s", fd, strerror(err));
    const long rc = chain_submit(c, sl.h_in.p, n, false);
    return rc < 0 ? rc : (long)n;
}

void *lrhip_chain_ring_input(lrhip_chain_t *c)
{
    if (!c) { set_error("null chain"); return nullptr; }
    if (c->ring.empty()) { set_error("chain has no ring: call lrhip_chain_set_ring first"); return nullptr; }
    if (c->inflight == c->ring.size()) { set_error("ring full: collect a chunk first (%u in flight)", c->inflight); return nullptr; }
    return c->ring[c->head]->h_in.p;
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
    if (sl.n_out) host_copy(out_host, sl.h_out.p, (size_t)sl.n_out * c->ops.back().stage->out_size);
    c->inflight--;
    return sl.n_out;
}

// ---- chunk coalescing on the ring ---------------------------------------------------------------------------------
// launch the head slot with its `fill` accumulated samples (they already sit in the slot's pinned input)
static long push_launch(lrhip_chain_t *c)
{
    unsigned long n = c->fill;
    c->fill = 0;                        // lrhip_chain_submit() refuses to run while pushed samples are pending
    long rc = lrhip_chain_submit(c, c->ring[c->head]->h_in.p, n);
    if (rc < 0) c->fill = n;            // nothing was launched: the samples are still in the slot, the caller may collect and retry
    else c->fill_t0 = 0.0;
    return rc;
}
static double monotonic_seconds()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int lrhip_chain_set_latency(lrhip_chain_t *c, double max_seconds)
{
    if (!c) return set_error("null chain");
    if (!(max_seconds >= 0.0)) return set_error("latency bound must be >= 0 (0 = batches run when full)");
    c->max_latency = max_seconds;
    return 0;
}
// copy the oldest slot's output out (waiting for it when `wait`); returns samples copied, -2 when it is not finished yet
static long push_collect(lrhip_chain_t *c, char *out, unsigned long cap, bool wait)
{
    unsigned tail = (c->head + (unsigned)c->ring.size() - c->inflight) % c->ring.size();
    lrhip_chain::Slot &sl = *c->ring[tail];
    if (!wait) {
        hipError_t q = hipEventQuery(sl.ev_out);
        if (q == hipErrorNotReady) return -2;
        if (q != hipSuccess) return set_error("hipEventQuery failed: %s", hipGetErrorString(q));
    }
    return lrhip_chain_collect(c, out, cap);
}

// the tail of push() and all of poll(): run the partial batch if its oldest sample has waited out the latency bound (and then wait for it: a live
// flow graph is far from the GPU's throughput, so the wait costs a launch, not the stream rate), then hand out whatever has finished, in stream order
static long push_drain(lrhip_chain_t *c, char *dst, unsigned long out_capacity, long total)
{
    const int out_size = c->ops.back().stage->out_size;
    if (c->fill && c->max_latency > 0.0 && monotonic_seconds() - c->fill_t0 >= c->max_latency) {
        if (c->inflight == c->ring.size()) {
            long got = push_collect(c, dst + (size_t)total * out_size, out_capacity - (unsigned long)total, true);
            if (got < 0) return got;
            total += got;
        }
        long rc = push_launch(c);
        if (rc < 0) return rc;
        while (c->inflight) {
            long got = push_collect(c, dst + (size_t)total * out_size, out_capacity - (unsigned long)total, true);
            if (got < 0) return got;
            total += got;
        }
    }
    while (c->inflight) {                                    // whatever has finished meanwhile, without waiting
        long got = push_collect(c, dst + (size_t)total * out_size, out_capacity - (unsigned long)total, false);
        if (got == -2) break;
        if (got < 0) return got;
        total += got;
    }
    return total;
}

unsigned long lrhip_chain_push_bound(const lrhip_chain_t *c, unsigned long n_in)
{
    if (!c || c->ring.empty()) return 0;
    // every slot in flight, the head slot, and the slots this call itself can fill
    return (unsigned long)(c->ring.size() + 1 + n_in / c->ring_chunk + 1) * c->ring_out_cap;
}

long lrhip_chain_push(lrhip_chain_t *c, const void *in_host, unsigned long n_in, void *out_host, unsigned long out_capacity)
{
    if (!c) return set_error("null chain");
    if (c->ring.empty()) return set_error("chain has no ring: call lrhip_chain_set_ring first");
    if (n_in && !in_host) return set_error("null input buffer");
    if (out_capacity < lrhip_chain_push_bound(c, n_in)) return set_error("output capacity %lu < lrhip_chain_push_bound() = %lu", out_capacity, lrhip_chain_push_bound(c, n_in));
    const int in_size = c->ops.front().stage->in_size, out_size = c->ops.back().stage->out_size;
    const char *src = (const char *)in_host;
    char *dst = (char *)out_host;
    long total = 0;
    while (n_in) {
        if (c->inflight == c->ring.size()) {                 // the head slot is still in flight: its turn to be collected
            long got = push_collect(c, dst + (size_t)total * out_size, out_capacity - (unsigned long)total, true);
            if (got < 0) return got;
            total += got;
        }
        unsigned long take = c->ring_chunk - c->fill;
        if (take > n_in) take = n_in;
        host_copy((char *)c->ring[c->head]->h_in.p + (size_t)c->fill * in_size, src, (size_t)take * in_size);
        if (!c->fill && c->max_latency > 0.0) c->fill_t0 = monotonic_seconds();
        c->fill += take; src += (size_t)take * in_size; n_in -= take;
        if (c->fill == c->ring_chunk) {
            long rc = push_launch(c);
            if (rc < 0) return rc;
        }
    }
    return push_drain(c, dst, out_capacity, total);
}

// The wall-clock side of the latency bound.  push() can only look at the clock when it is called; a live source that STALLS (an SDR that
// drops out, a squelched upstream block, a paused network source) would leave the partial batch on the device side of the host forever,
// where the reference streams every chunk through as it arrives (radio/core/block.lua:575-602).  The host calls poll() whenever its wait for
// input timed out (lua/radio/composites/devicechain.lua: DeviceChainBlock:run polls its input descriptors for lrhip_chain_poll_due()
// seconds instead of forever): a partial batch whose oldest sample has waited max_seconds is launched and its output returned from this
// call, like the same check at the end of push(); finished batches are handed out either way.  Nothing due: returns 0 at once.
long lrhip_chain_poll(lrhip_chain_t *c, void *out_host, unsigned long out_capacity)
{
    if (!c) return set_error("null chain");
    if (c->ring.empty()) return 0;
    if (out_capacity < lrhip_chain_push_bound(c, 0)) return set_error("output capacity %lu < lrhip_chain_push_bound() = %lu", out_capacity, lrhip_chain_push_bound(c, 0));
    return push_drain(c, (char *)out_host, out_capacity, 0);
}

double lrhip_chain_poll_due(const lrhip_chain_t *c)
{
    if (!c) { set_error("null chain"); return -1.0; }
    if (!(c->max_latency > 0.0)) return -1.0;                             // batches only run when full: wait for input forever
    if (!c->fill) {
        // nothing accumulating, but a batch launched by the last push() may still be in flight (batch == chunk, then the source stalls): its output
        // is handed out by the next poll(), so the wait for input stays bounded by the latency bound until the ring has drained
        return c->inflight ? c->max_latency : -1.0;
    }
    const double left = c->fill_t0 + c->max_latency - monotonic_seconds();
    return left > 0.0 ? left : 0.0;
}

long lrhip_chain_flush(lrhip_chain_t *c, void *out_host, unsigned long out_capacity)
{
    if (!c) return set_error("null chain");
    if (c->ring.empty()) return 0;                           // nothing is ever pending without a ring
    if (out_capacity < lrhip_chain_push_bound(c, 0)) return set_error("output capacity %lu < lrhip_chain_push_bound() = %lu", out_capacity, lrhip_chain_push_bound(c, 0));
    const int out_size = c->ops.back().stage->out_size;
    char *dst = (char *)out_host;
    long total = 0;
    if (c->fill) {
        if (c->inflight == c->ring.size()) {
            long got = push_collect(c, dst, out_capacity, true);
            if (got < 0) return got;
            total += got;
        }
        long rc = push_launch(c);
        if (rc < 0) return rc;
    }
    while (c->inflight) {
        long got = push_collect(c, dst + (size_t)total * out_size, out_capacity - (unsigned long)total, true);
        if (got < 0) return got;
        total += got;
    }
    return total;
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
int lrhip_host_register(void *ptr, unsigned long bytes)
{
    if (!ptr || !bytes) return set_error("host_register: null range");
    if (ensure_init()) return -1;
    HostRanges &h = host_ranges();
    {
        std::lock_guard<std::mutex> lk(h.m);
        if (h.pid != (long)getpid()) { h.r.clear(); h.dev.clear(); h.pid = (long)getpid(); }
        for (auto &e : h.r)
            if (e.first == (const char *)ptr) {
                if (e.second == bytes) return 0;
                return set_error("host_register: %p is already registered with another size", ptr);
            }
    }
    LR_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    void *d = nullptr;
    if (hipHostGetDevicePointer(&d, ptr, 0) != hipSuccess) { (void)hipGetLastError(); d = nullptr; }      // not mapped: the staged path only
    std::lock_guard<std::mutex> lk(h.m);
    h.r.emplace_back((const char *)ptr, (size_t)bytes);
    h.dev.push_back((char *)d);
    return 0;
}
int lrhip_host_unregister(void *ptr)
{
    if (!ptr) return 0;
    HostRanges &h = host_ranges();
    {
        std::lock_guard<std::mutex> lk(h.m);
        auto it = h.r.begin();
        for (; it != h.r.end(); ++it)
            if (it->first == (const char *)ptr) break;
        if (it == h.r.end()) return set_error("host_unregister: %p is not registered", ptr);
        if (h.dev.size() == h.r.size()) h.dev.erase(h.dev.begin() + (it - h.r.begin()));
        h.r.erase(it);
    }
    // nothing of this range may still be in flight: every host-pointer entry point has synchronised before it returned
    LR_HIP(hipHostUnregister(ptr));
    return 0;
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

// ---- time-axis sharding ---------------------------------------------------------------------------------------------
int lrhip_stage_seek(lrhip_stage_t *q, unsigned long long n0)
{
    if (!q) return set_error("null stage");
    unsigned long long out = 0;
    return q->seek(n0, &out);
}

int lrhip_chain_seek(lrhip_chain_t *c, unsigned long long n0)
{
    if (!c) return set_error("null chain");
    if (c->inflight || c->fill) return set_error("chain seek: chunks still in flight or pushed and not flushed");
    if (ctx().ready) LR_HIP(hipStreamSynchronize(ctx().stream));
    c->discard_in = 0;
    unsigned long long n = n0;
    for (auto &o : c->ops) {
        unsigned long long nn = 0;
        if (o.stage->seek(n, &nn)) return -1;
        n = nn;
    }
    return 0;
}

long lrhip_chain_halo(const lrhip_chain_t *c)
{
    if (!c) return set_error("null chain");
    // H = sum_k memory_k * (input samples of the chain per input sample of stage k), rounded up
    long double scale = 1.0L, h = 0.0L;
    for (auto &o : c->ops) {
        const long m = o.stage->memory();
        if (m < 0) return set_error("chain halo: stage '%s' has unbounded memory - this chain cannot be sharded in time", o.stage->kind());
        h += (long double)m * scale;
        unsigned long num = 1, den = 1;
        o.stage->rate(&num, &den);
        scale = scale * (long double)num / (long double)den;
    }
    return (long)ceill(h);
}

unsigned long lrhip_chain_shard_align(const lrhip_chain_t *c)
{
    if (!c) return 0;
    // a chain-input count `a` reaches stage k as a * den / num samples (num / den = product of the rates before it); the smallest a that is
    // a multiple of align_k there is align_k * num / gcd(align_k * num, den); the chain's alignment is the lcm over its stages
    auto gcd = [](unsigned long long x, unsigned long long y) { while (y) { unsigned long long t = x % y; x = y; y = t; } return x; };
    unsigned long long num = 1, den = 1, l = 1;
    for (auto &o : c->ops) {
        unsigned long long a = (unsigned long long)o.stage->align() * num;
        a /= gcd(a, den);
        l = l / gcd(l, a) * a;
        unsigned long rn = 1, rd = 1;
        o.stage->rate(&rn, &rd);
        num *= rn; den *= rd;
        unsigned long long g = gcd(num, den);
        num /= g; den /= g;
    }
    return (unsigned long)l;
}

int lrhip_chain_start_at(lrhip_chain_t *c, unsigned long long first_sample, unsigned long long *seek_sample)
{
    if (!c) return set_error("null chain");
    const long halo = lrhip_chain_halo(c);
    if (halo < 0) return -1;
    const unsigned long long align = lrhip_chain_shard_align(c);
    unsigned long long s = first_sample > (unsigned long long)halo ? first_sample - (unsigned long long)halo : 0ULL;
    s -= s % (align ? align : 1ULL);
    if (lrhip_chain_seek(c, s)) return -1;
    c->discard_in = first_sample - s;
    if (seek_sample) *seek_sample = s;
    return 0;
}

// ---- interprocess memory / events / peer copies --------------------------------------------------------------------
struct lrhip_ipc_event {
    hipEvent_t ev = nullptr;
};
static hipStream_t copy_stream()
{
    static hipStream_t s = nullptr;
    static long pid = 0;
    if (!s || pid != (long)getpid()) {
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) s = nullptr;
        pid = (long)getpid();
    }
    return s;
}

int lrhip_ipc_export(const void *dev_ptr, void *handle_out)
{
    static_assert(sizeof(hipIpcMemHandle_t) <= LRHIP_IPC_HANDLE_BYTES && sizeof(hipIpcEventHandle_t) <= LRHIP_IPC_HANDLE_BYTES, "handle size");
    if (!dev_ptr || !handle_out) return set_error("ipc_export: null argument");
    if (ensure_init()) return -1;
    hipIpcMemHandle_t h;
    LR_HIP(hipIpcGetMemHandle(&h, const_cast<void *>(dev_ptr)));
    memset(handle_out, 0, LRHIP_IPC_HANDLE_BYTES);
    memcpy(handle_out, &h, sizeof(h));
    return 0;
}

void *lrhip_ipc_open(const void *handle)
{
    if (!handle) { set_error("ipc_open: null handle"); return nullptr; }
    if (ensure_init()) return nullptr;
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    void *p = nullptr;
    LR_HIP_NULL(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    return p;
}

int lrhip_ipc_close(void *dev_ptr)
{
    if (!dev_ptr) return 0;
    LR_HIP(hipIpcCloseMemHandle(dev_ptr));
    return 0;
}

lrhip_ipc_event_t *lrhip_ipc_event_create(void *handle_out)
{
    if (!handle_out) { set_error("ipc_event_create: null handle buffer"); return nullptr; }
    if (ensure_init()) return nullptr;
    std::unique_ptr<lrhip_ipc_event> e(new (std::nothrow) lrhip_ipc_event());
    if (!e) { set_error("out of memory"); return nullptr; }
    LR_HIP_NULL(hipEventCreateWithFlags(&e->ev, hipEventDisableTiming | hipEventInterprocess));
    hipIpcEventHandle_t h;
    if (hipIpcGetEventHandle(&h, e->ev) != hipSuccess) {
        (void)hipEventDestroy(e->ev);
        set_error("hipIpcGetEventHandle failed");
        return nullptr;
    }
    memset(handle_out, 0, LRHIP_IPC_HANDLE_BYTES);
    memcpy(handle_out, &h, sizeof(h));
    return e.release();
}

lrhip_ipc_event_t *lrhip_ipc_event_open(const void *handle)
{
    if (!handle) { set_error("ipc_event_open: null handle"); return nullptr; }
    if (ensure_init()) return nullptr;
    std::unique_ptr<lrhip_ipc_event> e(new (std::nothrow) lrhip_ipc_event());
    if (!e) { set_error("out of memory"); return nullptr; }
    hipIpcEventHandle_t h;
    memcpy(&h, handle, sizeof(h));
    LR_HIP_NULL(hipIpcOpenEventHandle(&e->ev, h));
    return e.release();
}

void lrhip_ipc_event_destroy(lrhip_ipc_event_t *e)
{
    if (!e) return;
    if (e->ev) (void)hipEventDestroy(e->ev);
    delete e;
}

int lrhip_ipc_event_record(lrhip_ipc_event_t *e, int on_copy_stream)
{
    if (!e) return set_error("null event");
    hipStream_t s = on_copy_stream ? copy_stream() : ctx().stream;
    if (!s) return set_error("no stream");
    LR_HIP(hipEventRecord(e->ev, s));
    return 0;
}

int lrhip_ipc_event_wait(lrhip_ipc_event_t *e, int on_copy_stream)
{
    if (!e) return set_error("null event");
    hipStream_t s = on_copy_stream ? copy_stream() : ctx().stream;
    if (!s) return set_error("no stream");
    LR_HIP(hipStreamWaitEvent(s, e->ev, 0));
    return 0;
}

int lrhip_ipc_event_query(lrhip_ipc_event_t *e)
{
    if (!e) return set_error("null event");
    hipError_t r = hipEventQuery(e->ev);
    if (r == hipSuccess) return 1;
    if (r == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
    return set_error("hipEventQuery failed: %s", hipGetErrorString(r));
}

int lrhip_ipc_event_synchronize(lrhip_ipc_event_t *e)
{
    if (!e) return set_error("null event");
    LR_HIP(hipEventSynchronize(e->ev));
    return 0;
}

int lrhip_peer_copy(void *dst, int dst_device, const void *src, int src_device, unsigned long bytes)
{
    if (ensure_init()) return -1;
    if (!bytes) return 0;
    if (!dst || !src) return set_error("peer_copy: null pointer");
    hipStream_t s = copy_stream();
    if (!s) return set_error("peer_copy: no copy stream");
    if (dst_device != src_device) {
        int can = 0;
        LR_HIP(hipDeviceCanAccessPeer(&can, src_device, dst_device));
        if (!can) return set_error("peer_copy: device %d cannot reach device %d", src_device, dst_device);
        LR_HIP(hipMemcpyPeerAsync(dst, dst_device, src, src_device, bytes, s));
    } else {
        LR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
    }
    return 0;
}

int lrhip_copy_stream_synchronize(void)
{
    hipStream_t s = copy_stream();
    if (!s) return set_error("no copy stream");
    LR_HIP(hipStreamSynchronize(s));
    return 0;
}

}  // extern "C"
