// kernels_firpols.h - long overlap-save filters (radio/blocks/signal/firfilter.lua:320-398 for 513 taps and more) as a UNIFORMLY PARTITIONED
// convolution in the frequency domain, one launch per 1 536 taps (round 4).
//
// Why: round 3 ran 513 .. 1 281 taps on a 4096-point block per workgroup / per wave (kernels_firfft4k.h): 2.3-2.7 TB/s, bound by workgroup barriers
// in one form and by 390 registers (one wave per SIMD) in the other, and everything longer - or on a Float32 stream - as accumulating passes of the
// 1024-point kernel, 512 taps per pass, each pass re-reading and re-writing the output.  The 1024-point engine itself (kernels_firfft.h: one wave per
// block, no barrier, 12 waves per CU) is the fastest thing in the library, so the long filter is built from IT:
//
//     taps h = [h_0 | h_1 | ... | h_{P-1}], 512 taps each;  H_p = FFT_1024(h_p);  block b = outputs 512 b .. 512 b + 511
//     X_b = FFT_1024(x[512 (b-1) .. 512 (b+1)))                       one forward transform per block, hop 512
//     y_b = IFFT_1024( sum_p X_{b-p} H_p )[512 ..]                    one inverse transform per block, whatever P is
//
// A wave walks a RUN of consecutive blocks and keeps the last P-1 spectra in registers (a delay line of 32 registers per spectrum), so a block costs
// one forward and one inverse 1024-point transform plus P complex multiply-accumulates per bin - against P forward + P inverse transforms and P passes over
// the output for the accumulating form.  The window's old half is the previous block's new half, kept raw in 16 registers: every input sample is loaded
// ONCE (the 4096-point kernel re-read 31 % of them through L2), every output stored once.  A run starts with P-1 warm-up blocks (forward transform only,
// nothing stored): 2 / 43 blocks for 1 276 taps on 2^26 samples.
//
// P <= 3 in one launch (the delay line has to fit 168 registers: 3 waves per SIMD); longer filters take one launch per 1 536 taps, the later ones accumulating
// (`delay`, `accumulate`, as fir_fft_kernel).  Float32 streams with real taps: two runs ride as re / im of one complex transform (h real => the two
// convolutions stay apart), which is what fir_fft_kernel does with two consecutive blocks - here they have to be two RUNS, because a packed pair of
// consecutive blocks cannot feed a delay line whose step is one block.
//
// LDS: one 768-thread workgroup per CU = 12 waves x 8.7 KB of exchange buffer + tw1 8 KB + tw2 0.5 KB + P x 8 KB of H = 137 KB for P = 3.
// Accuracy: Float32 FFT arithmetic; the P products are summed in Float32 before the inverse transform; held to 1e-6 of the f64 oracle by the tests.
#pragma once
#include "kernels_firfft.h"

#ifndef LRHIP_POLS_WPB
#define LRHIP_POLS_WPB 12
#endif
// twiddle tables in registers instead of LDS reads per block (the kernel is LDS-bound: 55 % busy, section 4.7): 1 = tw1 (30 registers), 2 = tw1 + both tw2 sets
// (84 registers: needs LRHIP_POLS_WPB = 8, two waves per SIMD)
#ifndef LRHIP_POLS_TW_REG
#define LRHIP_POLS_TW_REG 0
#endif

namespace lrhip {

// waves per workgroup (= per CU: one workgroup).  ComplexFloat32 streams: 12 (three per SIMD, 168 registers).  Float32 streams carry two runs per wave and
// spilled up to 164 B per lane at that cap: 8 (two per SIMD, 220 registers, no scratch) measures 0.216 against 0.228 ms for 1 276 taps on 2^26 samples, while
// the ComplexFloat32 kernel LOSES 10 % at 8 (1.62 against 1.47 ms for 4 096 taps) - round 4, three alternations
#ifndef LRHIP_POLS_WPB_F32
#define LRHIP_POLS_WPB_F32 8
#endif
// (four partitions per launch - a delay line of three spectra, 96 registers - need the 256 registers of two waves per SIMD: 8 waves)
__host__ __device__ constexpr int pols_wpb(int S, int P = 1) { return S == 1 ? LRHIP_POLS_WPB_F32 : P >= 4 ? 8 : LRHIP_POLS_WPB; }
constexpr int POLS_HOP = 512;
// LDS map (float2 units): [WPB x exchange | tw1 16x64 | tw2 64 | H P x 1024]
__host__ __device__ constexpr int pols_lds_tw1(int S, int P = 1) { return pols_wpb(S, P) * FFT_EX_ELEMS; }
__host__ __device__ constexpr int pols_lds_tw2(int S, int P = 1) { return pols_lds_tw1(S, P) + 16 * 64; }
__host__ __device__ constexpr int pols_lds_h(int S, int P = 1) { return pols_lds_tw2(S, P) + 64; }
__host__ __device__ constexpr int pols_lds_elems(int S, int P) { return pols_lds_h(S, P) + P * 1024; }

// a + s * h on the packed VALU: two v_pk_fma_f32
__device__ __forceinline__ cf cmac(cf a, cf s, cf h)
{
    cf t = __builtin_elementwise_fma(__builtin_shufflevector(s, s, 0, 0), h, a), r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(s), "v"(h), "v"(t));
    return r;
}

// tables: the per-partition tables of fir_fft_kernel, FFT_TABLE_ELEMS float2 each ([tw1 16x64 | Hperm 16x64 | tw2 64]); this launch applies partitions
// part0 .. part0 + P - 1 (taps [512 part0, 512 (part0 + P)) of the Mh-tap filter) to the stream delayed by 512 part0 samples.
// nblocks = ceil(n_out / 512); a wave owns `run` consecutive blocks (S = 1: two runs, `run` blocks apart).
template <int S, int P>
__global__ __launch_bounds__(64 * pols_wpb(S, P), 1) void fir_pols_kernel(const float *__restrict__ hist, const float *__restrict__ x, const float2 *__restrict__ tables,
                                                                    float *__restrict__ y, int Mh, long n, long n_out, long nblocks, long run, int part0,
                                                                    int accumulate, float *__restrict__ hist_out)
{
    constexpr int POLS_WPB = pols_wpb(S, P), POLS_LDS_TW1 = pols_lds_tw1(S, P), POLS_LDS_TW2 = pols_lds_tw2(S, P), POLS_LDS_H = pols_lds_h(S, P);
    static_assert(P >= 1 && P <= 4, "the spectra delay line lives in registers");
    extern __shared__ __attribute__((aligned(16))) float2 fl[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (hist_out && blockIdx.x == 0)
        for (int i = tid; i < (Mh - 1) * S; i += 64 * POLS_WPB) hist_out[i] = stream_at<S>(hist, x, n + i / S, i % S, Mh, n);
    cf *flc = reinterpret_cast<cf *>(fl);
    cf *ex = flc + wave * FFT_EX_ELEMS;
    const cf *tw1 = flc + POLS_LDS_TW1, *tw2 = flc + POLS_LDS_TW2, *Hs = flc + POLS_LDS_H;
    {
        const float2 *t0 = tables + (size_t)part0 * FFT_TABLE_ELEMS;
        for (int i = tid; i < 16 * 64; i += 64 * POLS_WPB) fl[POLS_LDS_TW1 + i] = t0[i];
        for (int i = tid; i < 64; i += 64 * POLS_WPB) fl[POLS_LDS_TW2 + i] = t0[2 * 16 * 64 + i];
#if LRHIP_FFT_E2_SWAP
        for (int i = tid; i < P * 1024; i += 64 * POLS_WPB) fl[POLS_LDS_H + i] = tables[(size_t)(part0 + i / 1024) * FFT_TABLE_ELEMS + FFT_TABLE_HSW + (i & 1023)];
#else
        for (int i = tid; i < P * 1024; i += 64 * POLS_WPB) fl[POLS_LDS_H + i] = tables[(size_t)(part0 + i / 1024) * FFT_TABLE_ELEMS + 16 * 64 + (i & 1023)];
#endif
    }
    __syncthreads();
#if LRHIP_FFT_E2_SWAP
    const int sub = lane >> 4, k1s = lane & 15;       // kernels_firfft.h: the inner transposes as register <-> row swaps
#else
    const int sub = lane & 3, k1s = lane >> 2;
#endif
    const long delay = (long)part0 * POLS_HOP;
#if LRHIP_POLS_TW_REG >= 1
    cf tw1r[16];
#pragma unroll
    for (int q = 1; q < 16; q++) tw1r[q] = tw1[q * 64 + lane];
#define POLS_TW1(q) tw1r[q]
#else
#define POLS_TW1(q) tw1[(q) * 64 + lane]
#endif
#if LRHIP_POLS_TW_REG >= 2
    cf tw2f[16], tw2i[16];
#pragma unroll
    for (int q = 1; q < 16; q++) tw2f[q] = tw2[q * 4 + sub];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int t2 = 1; t2 < 4; t2++) tw2i[4 * j + t2] = tw2[(4 * j + sub) * 4 + t2];
#define POLS_TW2F(q) tw2f[q]
#define POLS_TW2I(j, t2) tw2i[4 * (j) + (t2)]
#else
#define POLS_TW2F(q) tw2[(q) * 4 + sub]
#define POLS_TW2I(j, t2) tw2[(4 * (j) + sub) * 4 + (t2)]
#endif

    // runs: slot s = the POLS_WPB (S = 1: 2 x POLS_WPB) adjacent runs of one workgroup pass; workgroup g walks slots g, g + gridDim.x, ...
    constexpr int RPW = S == 2 ? 1 : 2;                       // runs per wave
    const long nruns = (nblocks + run - 1) / run, nslots = (nruns + POLS_WPB * RPW - 1) / (POLS_WPB * RPW);
    for (long slot = blockIdx.x; slot < nslots; slot += gridDim.x) {
        const long r0 = (slot * POLS_WPB + wave) * RPW;       // this wave's (first) run
        if (r0 >= nruns) continue;                            // no workgroup barrier inside the loop: a wave may skip
        const long ba = r0 * run, bend_a = ba + run < nblocks ? ba + run : nblocks;
        const long bb = ba + run, bend_b = S == 1 ? (bb + run < nblocks ? bb + run : nblocks) : 0;      // second run (S = 1), may be empty
        cf keep[8], S1[16], S2[16], S3[16];
#pragma unroll
        for (int i = 0; i < 8; i++) keep[i] = cf{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; i++) { S1[i] = cf{0.f, 0.f}; S2[i] = cf{0.f, 0.f}; S3[i] = cf{0.f, 0.f}; }
        // k = -(P-1) .. run-1: block ba + k of run A (and bb + k of run B); k < 0 = warm-up (spectrum only)
        for (long k = -(long)(P - 1); k < run; k++) {
            const long b = ba + k;
            if (b >= bend_a) break;
            cf v[16];
            // ---- window of block b: x indices 512 (b - 1) - delay .. 512 (b + 1) - delay; rows 0..7 = the previous block's rows 8..15
            auto load_rows = [&](int first_row) {
                if (S == 2) {
                    const long xlo = POLS_HOP * (b - 1) - delay;
                    if (xlo + 64 * first_row >= 0 && xlo + FFTN <= n) {
                        const cf *srcu = reinterpret_cast<const cf *>(x) + xlo;
#pragma unroll
                        for (int i = 0; i < 16; i++)
                            if (i >= first_row) v[i] = (srcu + 64 * i)[(unsigned)lane];
                    } else {
                        const long p0 = xlo + (Mh - 1);
#pragma unroll
                        for (int i = 0; i < 16; i++)
                            if (i >= first_row) {
                                const long p = p0 + 64 * i + lane;
                                v[i] = cf{stream_at<2>(hist, x, p, 0, Mh, n), stream_at<2>(hist, x, p, 1, Mh, n)};
                            }
                    }
                } else {
                    const long xa = POLS_HOP * (b - 1) - delay, xb = xa + POLS_HOP * run;
                    const bool live_b = bb + k < bend_b;
                    if (xa + 64 * first_row >= 0 && xb + FFTN <= n && live_b) {
                        const float *sa = x + xa, *sb = x + xb;
#pragma unroll
                        for (int i = 0; i < 16; i++)
                            if (i >= first_row) v[i] = cf{(sa + 64 * i)[(unsigned)lane], (sb + 64 * i)[(unsigned)lane]};
                    } else {
                        const long pa = xa + (Mh - 1), pb = xb + (Mh - 1);
#pragma unroll
                        for (int i = 0; i < 16; i++)
                            if (i >= first_row)
                                v[i] = cf{stream_at<1>(hist, x, pa + 64 * i + lane, 0, Mh, n), live_b ? stream_at<1>(hist, x, pb + 64 * i + lane, 0, Mh, n) : 0.f};
                    }
                }
            };
            if (k == -(long)(P - 1)) load_rows(0);
            else {
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = keep[i];
                load_rows(8);
            }
#pragma unroll
            for (int i = 0; i < 8; i++) keep[i] = v[8 + i];
            // ---- forward 1024-point transform (kernels_firfft.h stages)
            dft16<1>(v);
#pragma unroll
            for (int q = 1; q < 16; q++) v[q] = cmul(v[q], POLS_TW1(q));
#if LRHIP_FFT_E2_SWAP
            exchange(ex, v, [&](int q) { return q * FFT_E1F_ROW_SW + lane; }, [&](int i) { return k1s * FFT_E1F_ROW_SW + 4 * i + sub; });
#else
            exchange(ex, v, [&](int q) { return q * FFT_E1_ROW + lane; }, [&](int i) { return k1s * FFT_E1_ROW + 4 * i + sub; });
#endif
            dft16<1>(v);
#pragma unroll
            for (int q = 1; q < 16; q++) v[q] = cmul(v[q], POLS_TW2F(q));
#if LRHIP_FFT_E2_SWAP
#pragma unroll
            for (int j = 0; j < 4; j++) fft_transpose_rows(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
#else
            exchange(ex, v, [&](int q) { return k1s * FFT_E2_ROW + 17 * sub + q; }, [&](int r) { return k1s * FFT_E2_ROW + 17 * (r & 3) + (r & 12) + sub; });
#endif
#pragma unroll
            for (int j = 0; j < 4; j++) radix4<1>(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
            if (k < 0) {
                // warm-up: the spectrum enters the delay line, nothing comes out
                if (P >= 4) {
#pragma unroll
                    for (int r = 0; r < 16; r++) S3[r] = S2[r];
                }
                if (P >= 3) {
#pragma unroll
                    for (int r = 0; r < 16; r++) S2[r] = S1[r];
                }
                if (P >= 2) {
#pragma unroll
                    for (int r = 0; r < 16; r++) S1[r] = v[r];
                }
                continue;
            }
            // ---- sum_p X_{b-p} H_p, the delay line moves on
#pragma unroll
            for (int r = 0; r < 16; r++) {
                cf a = cmul(v[r], Hs[r * 64 + lane]);
                if (P >= 2) a = cmac(a, S1[r], Hs[1024 + r * 64 + lane]);
                if (P >= 3) a = cmac(a, S2[r], Hs[2048 + r * 64 + lane]);
                if (P >= 4) a = cmac(a, S3[r], Hs[3072 + r * 64 + lane]);
                if (P >= 4) S3[r] = S2[r];
                if (P >= 3) S2[r] = S1[r];
                if (P >= 2) S1[r] = v[r];
                v[r] = a;
            }
            // ---- inverse transform
#pragma unroll
            for (int j = 0; j < 4; j++) {
                radix4<-1>(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
#pragma unroll
                for (int t2 = 1; t2 < 4; t2++) v[4 * j + t2] = cmulc(v[4 * j + t2], POLS_TW2I(j, t2));
            }
#if LRHIP_FFT_E2_SWAP
#pragma unroll
            for (int j = 0; j < 4; j++) fft_transpose_rows(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
            dft16<-1>(v);
            exchange(ex, v, [&](int i) { return k1s * FFT_E1I_ROW_SW + 4 * i + sub; }, [&](int q) { return q * FFT_E1I_ROW_SW + lane; });
#else
            exchange(ex, v, [&](int r) { return k1s * FFT_E2_ROW + 17 * (r & 3) + (r & 12) + sub; }, [&](int q) { return k1s * FFT_E2_ROW + 17 * sub + q; });
            dft16<-1>(v);
            exchange(ex, v, [&](int i) { return k1s * FFT_E1_ROW + 4 * i + sub; }, [&](int q) { return q * FFT_E1_ROW + lane; });
#endif
#pragma unroll
            for (int q = 1; q < 16; q++) v[q] = cmulc(v[q], POLS_TW1(q));
            dft16<-1>(v);
            // ---- rows 8..15 are the block's 512 outputs
            if (S == 2) {
                const long o0 = POLS_HOP * b;
                cf *dstu = reinterpret_cast<cf *>(y) + o0;
                if (o0 + POLS_HOP <= n_out) {
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        if (!accumulate) __builtin_nontemporal_store(v[8 + i], (dstu + 64 * i) + (unsigned)lane);
                        else (dstu + 64 * i)[(unsigned)lane] = (dstu + 64 * i)[(unsigned)lane] + v[8 + i];
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        if (o0 + 64 * i + lane < n_out) dstu[64 * i + lane] = accumulate ? dstu[64 * i + lane] + v[8 + i] : v[8 + i];
                }
            } else {
                const long oa = POLS_HOP * b, ob = oa + POLS_HOP * run;
                const bool live_b = bb + k < bend_b;
                if (ob + POLS_HOP <= n_out && live_b) {
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        float *da = y + oa + 64 * i, *db = y + ob + 64 * i;
                        if (!accumulate) {
                            __builtin_nontemporal_store(v[8 + i].x, da + (unsigned)lane);
                            __builtin_nontemporal_store(v[8 + i].y, db + (unsigned)lane);
                        } else {
                            da[lane] += v[8 + i].x;
                            db[lane] += v[8 + i].y;
                        }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const long pa = oa + 64 * i + lane, pb = ob + 64 * i + lane;
                        if (pa < n_out) y[pa] = accumulate ? y[pa] + v[8 + i].x : v[8 + i].x;
                        if (live_b && pb < n_out) y[pb] = accumulate ? y[pb] + v[8 + i].y : v[8 + i].y;
                    }
                }
            }
        }
    }
}

}  // namespace lrhip
