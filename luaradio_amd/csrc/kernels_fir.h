// kernels_fir.h - FIRFilterBlock on CDNA4.
//
// Reference semantics (radio/blocks/signal/firfilter.lua:230-305): with s = [last M-1 inputs | chunk] and
// taps_rev[j] = h[M-1-j], output k is  y[k] = sum_{j<M} taps_rev[j] * s[q_k + j]  (oldest sample first),
// q_k = first + k*D  (D = 1 for a plain FIRFilterBlock; D > 1 fuses the following DownsamplerBlock,
// radio/composites/decimator.lua:37-39, `first` being the downsampler's carried index).
//
// Two kernels:
//   fir_mfma_kernel   real taps (rr / cr - the headline): the filter is applied as a banded-Toeplitz
//                     matrix product on the f32 matrix cores (v_mfma_f32_16x16x4_f32, exact f32, = fmaf
//                     chain in ascending-k order), data staged once per tile into LDS.
//   fir_direct_kernel any taps/decimation (complex taps, very long filters, odd decimations): one output
//                     per thread, taps via scalar loads, inputs through L1/L2.
//
// Both produce BIT-IDENTICAL results to the fmaf chain  acc = fmaf(s[q+j], taps_rev[j], acc), j ascending
// (oracle LRO_MODE_FMA), for finite inputs.
#pragma once
#include "common.h"
#include "kernels_elem.h"

#ifndef LRHIP_FIR_PIPE
#define LRHIP_FIR_PIPE 2      /* explicit one-step software pipeline of the persistent kernel's MFMA loop - 0: never, 1: always, 2: decimating instantiations (WBFM receiver 0.167 -> 0.162 ms, same box; no gain at D = 1) */
#endif
#ifndef LRHIP_DISC_EPI_LDS
#define LRHIP_DISC_EPI_LDS 0      /* 1: discriminator epilogue of the persistent Toeplitz kernel through LDS (fewer VALU instructions, two more barriers per tile: 0.206-0.208 against 0.201-0.202 ms for the WBFM receiver, same box) */
#endif

namespace lrhip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// resident workgroups (= waves per SIMD, 256-thread workgroups) the persistent FIR kernel is register-budgeted for
#ifndef LRHIP_FIR_SETPRIO
#define LRHIP_FIR_SETPRIO 0
#endif
#ifndef LRHIP_FIR_SCHED
#define LRHIP_FIR_SCHED 0
#endif
#ifndef LRHIP_FIR_XCD_MAP
#define LRHIP_FIR_XCD_MAP 0       /* 1: XCD-contiguous tile order in the persistent kernel.  Measured (same box): HBM reads of the WBFM tuner 550.5 -> 537.7 MB (1.025x -> 1.001x of its
                                     input) and of the direct-form headline 2.215 -> 2.163 GB, but the receiver step 0.161 -> 0.164 ms and the headline unchanged (1.267 / 1.270 ms):
                                     neither kernel is HBM-bound, so the plain stride stays */
#endif
#ifndef LRHIP_FIR_WAVES_PER_SIMD
#define LRHIP_FIR_WAVES_PER_SIMD 3
#endif

// s(p): the "state" vector of firfilter.lua:244-250 without materialising it:
// p < M-1 -> hist[p] (carried), else x[p-(M-1)]; beyond the chunk reads as 0 (never used by a valid output).
template <int S>
__device__ __forceinline__ float stream_at(const float *__restrict__ hist, const float *__restrict__ x,
                                           long p, int c, int M, long n)
{
    if (p < 0) return 0.0f;
    if (p < M - 1) return hist[p * S + c];
    long xi = p - (M - 1);
    return xi < n ? x[xi * S + c] : 0.0f;
}

// Raw IQ-file records as a kernel's input (round 3; FMT > 0): the input is the raw record stream of an IQ file - (I, Q) pairs of unsigned 8-bit (RTL-SDR), signed 8-bit (HackRF) or little-endian
// signed 16-bit integers, 2 or 4 bytes per sample instead of 8 - and IQFileSource's conversion (radio/blocks/sources/iqfile.lua:99-113, format_utils.lua:82-88:
// (raw - offset) / scale, evaluated in double, stored as Float32) happens on the way into LDS.  x = raw - offset is exact in Float32, and
// fma(x, RH, x * RL) with RH + RL = 1 / scale to 48 bits gives the bits of the double-precision expression for every raw value of these formats
// (tests/test_gpu_rx.py: emulated for all values, and the kernel against the file-format kernel).
enum { RX_FMT_CF32 = 0, RX_FMT_U8 = 1, RX_FMT_S8 = 2, RX_FMT_S16LE = 3 };
template <int FMT> __host__ __device__ constexpr int rx_raw_bytes() { return FMT == RX_FMT_S16LE ? 2 : 1; }      // per scalar
template <int FMT>
__device__ __forceinline__ cf rx_raw_sample(unsigned i_raw, unsigned q_raw)
{
    constexpr double SC = FMT == RX_FMT_S16LE ? 32767.5 : 127.5;
    constexpr float RH = (float)(1.0 / SC), RL = (float)(1.0 / SC - (double)RH);
    cf x;
    if (FMT == RX_FMT_U8) x = cf{(float)i_raw, (float)q_raw} - cf{127.5f, 127.5f};
    else if (FMT == RX_FMT_S8) x = cf{(float)(int)(int8_t)i_raw, (float)(int)(int8_t)q_raw};
    else x = cf{(float)(int)(int16_t)i_raw, (float)(int)(int16_t)q_raw};
    return __builtin_elementwise_fma(x, cf{RH, RH}, x * cf{RL, RL});
}
// two samples = one 4-byte (8-bit formats: .x) or 8-byte word of records -> (re0, im0, re1, im1)
template <int FMT>
__device__ __forceinline__ float4 rx_raw_pair(uint2 w)
{
    cf s0, s1;
    if (FMT == RX_FMT_S16LE) {
        s0 = rx_raw_sample<FMT>(w.x & 0xffffu, w.x >> 16);
        s1 = rx_raw_sample<FMT>(w.y & 0xffffu, w.y >> 16);
    } else {
        s0 = rx_raw_sample<FMT>(w.x & 0xffu, (w.x >> 8) & 0xffu);
        s1 = rx_raw_sample<FMT>((w.x >> 16) & 0xffu, w.x >> 24);
    }
    return make_float4(s0.x, s0.y, s1.x, s1.y);
}
// stream_at<2> on a stream whose chunk is raw records (the carried history stays ComplexFloat32)
template <int FMT>
__device__ __forceinline__ float stream_at_raw(const float *__restrict__ hist, const float *__restrict__ x, long p, int c, int M, long n)
{
    if (FMT == RX_FMT_CF32) return stream_at<2>(hist, x, p, c, M, n);
    if (p < 0) return 0.0f;
    if (p < M - 1) return hist[p * 2 + c];
    const long xi = p - (M - 1);
    if (xi >= n) return 0.0f;
    const cf v = FMT == RX_FMT_S16LE ? rx_raw_sample<FMT>(reinterpret_cast<const uint16_t *>(x)[2 * xi], reinterpret_cast<const uint16_t *>(x)[2 * xi + 1])
                                     : rx_raw_sample<FMT>(reinterpret_cast<const uint8_t *>(x)[2 * xi], reinterpret_cast<const uint8_t *>(x)[2 * xi + 1]);
    return c ? v.y : v.x;
}

// ------------------------------------------------------------------------------------------------
// history carry: hist_out[i] = s[n + i], i < M-1   (firfilter.lua:248 memmove of the last M-1 state samples)
// ping-pong buffers, so it can run concurrently with nothing and after the filter kernel in stream order.
// ------------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void fir_history_kernel(const float *__restrict__ hist_in, const float *__restrict__ x,
                                                          float *__restrict__ hist_out, int M, long n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (M - 1) * S) return;
    int r = i / S, c = i % S;
    hist_out[i] = stream_at<S>(hist_in, x, n + r, c, M, n);
}

// ------------------------------------------------------------------------------------------------
// Direct form, one output per thread.  MODE: 0 = rr, 1 = cr, 2 = cc.
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void fir_direct_kernel(const float *__restrict__ hist, const float *__restrict__ x,
                                                         const float *__restrict__ taps_rev, float *__restrict__ y,
                                                         int M, long n, long n_out, long first, long D)
{
    constexpr int S = MODE == 0 ? 1 : 2;
    long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_out) return;
    long q = first + k * D;
    float re = 0.f, im = 0.f;
    for (int j = 0; j < M; j++) {
        float xr = stream_at<S>(hist, x, q + j, 0, M, n);
        if (MODE == 0) {
            re = fmaf(xr, taps_rev[j], re);
        } else if (MODE == 1) {
            float xi = stream_at<S>(hist, x, q + j, 1, M, n);
            float h = taps_rev[j];
            re = fmaf(xr, h, re);
            im = fmaf(xi, h, im);
        } else {
            float xi = stream_at<S>(hist, x, q + j, 1, M, n);
            float hr = taps_rev[2 * j], hi = taps_rev[2 * j + 1];
            re = fmaf(xr, hr, re);
            re = fmaf(xi, -hi, re);
            im = fmaf(xr, hi, im);
            im = fmaf(xi, hr, im);
        }
    }
    if (S == 1) y[k] = re;
    else reinterpret_cast<float2 *>(y)[k] = make_float2(re, im);
}

// ------------------------------------------------------------------------------------------------
// Banded-Toeplitz MFMA FIR (real taps).
//
// One MFMA accumulator = 16 columns x 16 rows.  A column is (block, component): `block` = 16 consecutive
// (decimated) outputs, component = re/im for ComplexFloat32 (S = 2) so an accumulator covers 16/S blocks =
// 256/S output samples.  Row m of block b is output k = 16*b + m.  With the tile's input samples r = 0.. in
// LDS (r = 0 is stream position first + tile_k0*D - e, `e` = 0..3 samples of slack that makes the global
// source 16-B aligned), block b's window starts at r0 = 16*D*b and
//        Y[m][col] = sum_t A[m][t] * X[t][col],   X[t][col] = lds[S*(r0 + t) + c],
//        A[m][t]   = taps_rev[t - e - m*D]  (0 outside [0, M))          <- banded Toeplitz, built on the host
// K = e + 15*D + M, issued as K/4 v_mfma_f32_16x16x4_f32 steps (A: lane -> A[m = lane&15][t = 4*step + (lane>>4)],
// B: lane -> X[t = 4*step + (lane>>4)][col = lane&15]).  t ascends, zero entries add exactly 0, so every output
// is the fmaf chain over its M taps in the reference's order.
//
// LDS layout: logical float address a = S*r + c; rows of ROW = 16*D*S floats are padded by PAD = 2*S floats
// (physical = a + PAD*(a/ROW)), which makes the 32 lanes of each ds_read_b32 lane group hit 32 distinct banks
// ((ROW+PAD)*b + S*k + c are all distinct mod 32 for the 16/S blocks x S components x 2 k's of a lane group).
// B-fragment address = lane_base + immediate: the row-crossing term floor((S*t+c)/ROW) = floor(step/(4*D))
// is lane-invariant.
//
// Work per workgroup (256 threads = 4 waves): 4*NACC accumulators = TILE_OUT = 1024*NACC/S output samples.
// ------------------------------------------------------------------------------------------------
template <int S, int D>
struct FirMfmaGeom {
    static constexpr int ROW = 16 * D * S;        // floats per LDS row (= one block's input stride)
    static constexpr int PAD = 2 * S;             // floats of padding per row
    static constexpr int BPA = 16 / S;            // blocks per accumulator
    static constexpr int GROUP = 4 * D;           // MFMA steps per LDS row of t
    // nw = waves per workgroup (4; 1: a wave stages its own window - no barrier couples it to others, fir_mfma_persistent_kernel NW)
    __host__ __device__ static constexpr int tile_out(int nacc, int nw = 4) { return nw * nacc * BPA * 16; }
    // samples staged per tile for `ksteps` MFMA steps
    __host__ __device__ static constexpr int span(int nacc, int ksteps, int nw = 4) { return 16 * D * (nw * nacc * BPA - 1) + 4 * ksteps; }
    __host__ __device__ static constexpr int phys(int a) { return a + PAD * (a / ROW); }
};

// ---- shared device pieces ---------------------------------------------------------------------------------

// (fused FrequencyTranslatorBlock: rotate_pair / rotate_sample of kernels_elem.h on the way into LDS)

// one float4 of staged samples (logical float index 4*i4) -> padded LDS rows
template <int S, int D>
__device__ __forceinline__ void lds_put4(float *ldsX, int i4, float4 v)
{
    using G = FirMfmaGeom<S, D>;
    int p = G::phys(4 * i4);
    if (S == 2) {
        *reinterpret_cast<float4 *>(ldsX + p) = v;
    } else {
        *reinterpret_cast<float2 *>(ldsX + p) = make_float2(v.x, v.y);
        *reinterpret_cast<float2 *>(ldsX + p + 2) = make_float2(v.z, v.w);
    }
}

// edge tiles (touch the carried history, the end of the chunk, or an unaligned source): per-sample staging
// the phasor of window position k under the relative rotator staging of fir_mfma_persistent_kernel (REL): the product the interior tiles keep in
// registers, float4 i4 = k / 2 = tid + NT u -> P(2 tid) * P(2 NT u + (k & 1)); edge tiles evaluate it per sample, same bits
template <int NT>
__device__ __forceinline__ cf rel_window_phasor(uint64_t step_fx, int k)
{
    const int i4 = k >> 1;
    return cmul(phasor_poly(step_fx * (uint64_t)(2 * (i4 % NT))), phasor_poly(step_fx * (uint64_t)(2 * NT * (i4 / NT) + (k & 1))));
}

template <int S, int D, bool ROT, bool REL = false, int NT = 256, int FMT = 0>
__device__ __forceinline__ void stage_edge(float *ldsX, const float *__restrict__ hist, const float *__restrict__ x,
                                           long base, int span, int M, long n, uint64_t rot_step_fx, uint64_t rot_count0)
{
    using G = FirMfmaGeom<S, D>;
    static_assert(FMT == 0 || S == 2, "raw records are (I, Q) pairs");
    for (int r = threadIdx.x; r < span; r += NT) {
        long p = base + r;
        float v0 = FMT ? stream_at_raw<FMT>(hist, x, p, 0, M, n) : stream_at<S>(hist, x, p, 0, M, n);
        float v1 = S == 2 ? (FMT ? stream_at_raw<FMT>(hist, x, p, 1, M, n) : stream_at<S>(hist, x, p, 1, M, n)) : 0.f;
        if (REL) {
            const cf o = cmul(cf{v0, v1}, rel_window_phasor<NT>(rot_step_fx, r));
            v0 = o.x;
            v1 = o.y;
        } else if (ROT) {
            // absolute sample index of stream position p is rot_count0 + p - (M-1); history before the
            // start of the stream is zero, so its phase is irrelevant
            float2 o = rotate_sample(make_float2(v0, v1), rot_step_fx, rot_count0 + (uint64_t)(p - (M - 1)));
            v0 = o.x;
            v1 = o.y;
        }
        int pa = G::phys(S * r);
        ldsX[pa] = v0;
        if (S == 2) ldsX[pa + 1] = v1;
    }
}

// Zero-padded reversed taps in LDS: [ZL zeros | taps_rev[0..M) | zeros], ZL = 3 + 15*D, length fir_taps_len().
// The Toeplitz entry A[m][t] = taps_rev[t - e - m*D] is then a plain read at (ZL + t - e - m*D): no table.
__host__ __device__ constexpr int fir_taps_zl(int D) { return 3 + 15 * D; }
__host__ __device__ constexpr int fir_taps_len(int D, int ksteps) { return ((fir_taps_zl(D) + 4 * ksteps + 3) / 4) * 4; }

// the MFMA main loop over one staged tile; KS > 0 => fully unrolled.
// A fragment of step s: lane (m = lane&15, kq = lane>>4) reads taps_pad[ZL + 4s + kq - e - m*D] (a broadcast-friendly
// LDS read; equal addresses across lanes are free).  NOUT = 2: two tap arrays (ldsT, ldsT + tlen) applied to the same B
// fragments -> two accumulator sets (the re and im outputs of a complex-taps filter over the interleaved float stream).
// TQS > 0: the tap array is kept in FOUR copies, TQS floats apart, and lane group kq reads copy kq.  At D = 5 the A-fragment address is
// kq - 5 col + const: the 32 lanes of a half-wave (kq in {0, 1} or {2, 3}) then hit 16 + 16 banks that overlap in three places (5 (c' - c) = 1
// mod 32 at c' = c + 13), a 2-way conflict on every tap read (SQ_LDS_BANK_CONFLICT: 14 M of 48 M LDS cycles of the receiver kernel).  With
// TQS = 15 mod 32 the second lane group of a half lands 16 banks away from the first - the complement of {-5 c mod 32} - and the read is conflict-free.
template <int S, int D, int NACC, int KS, int NOUT, int TQS = 0>
__device__ __forceinline__ void mfma_tile(const float *ldsT, int tlen, int e, const float *ldsX, int ksteps, f32x4 (&acc)[NOUT][NACC])
{
    using G = FirMfmaGeom<S, D>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int blk_in_acc = S == 2 ? (col >> 1) : col;
    const int comp = S == 2 ? (col & 1) : 0;
    constexpr int ACC_STRIDE = (G::ROW + G::PAD) * G::BPA;    // floats between consecutive accumulators' blocks
    const float *bptr = ldsX + (G::ROW + G::PAD) * ((wave * NACC) * G::BPA + blk_in_acc) + S * kq + comp;
    const float *aptr = ldsT + kq * TQS + fir_taps_zl(D) + kq - e - col * D;      // row m of the Toeplitz block = lane & 15
#pragma unroll
    for (int o = 0; o < NOUT; o++)
#pragma unroll
        for (int a = 0; a < NACC; a++) acc[o][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto step = [&](const float *ap, const float *bp, int j) {
        float av[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; o++) av[o] = ap[o * tlen + 4 * j];
#pragma unroll
        for (int a = 0; a < NACC; a++) {
            float bv = bp[a * ACC_STRIDE + j * 4 * S];
#pragma unroll
            for (int o = 0; o < NOUT; o++) acc[o][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[o], bv, acc[o][a], 0, 0, 0);
        }
    };
    if constexpr (KS > 0 && (LRHIP_FIR_PIPE == 1 || (LRHIP_FIR_PIPE == 2 && D > 1))) {
        // explicit one-step software pipeline: the fragments of step s+1 are in flight while step s is multiplied.  The
        // sched_barriers keep hipcc from sinking each ds_read next to its MFMA (where it waits for it at once).
        float av[2][NOUT], bv[2][NACC];
        auto fetch = [&](int buf, int s) {
            const int g = s / G::GROUP, j = s % G::GROUP;
            const float *ap = aptr + 4 * g * G::GROUP, *bp = bptr + g * (G::ROW + G::PAD);
#pragma unroll
            for (int o = 0; o < NOUT; o++) av[buf][o] = ap[o * tlen + 4 * j];
#pragma unroll
            for (int a = 0; a < NACC; a++) bv[buf][a] = bp[a * ACC_STRIDE + j * 4 * S];
        };
        fetch(0, 0);
#pragma unroll
        for (int s = 0; s < KS; s++) {
            if (s + 1 < KS) fetch((s + 1) & 1, s + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < NACC; a++)
#pragma unroll
                for (int o = 0; o < NOUT; o++) acc[o][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s & 1][o], bv[s & 1][a], acc[o][a], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if constexpr (KS > 0) {
#if LRHIP_FIR_SETPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int g = 0; g < (KS + G::GROUP - 1) / G::GROUP; g++)
#pragma unroll
            for (int j = 0; j < G::GROUP; j++)
                if (g * G::GROUP + j < KS) step(aptr + 4 * g * G::GROUP, bptr + g * (G::ROW + G::PAD), j);
#if LRHIP_FIR_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    } else {
        const int ngroups = ksteps / G::GROUP;
        for (int g = 0; g < ngroups; g++) {
#pragma unroll
            for (int j = 0; j < G::GROUP; j++) step(aptr, bptr, j);
            aptr += 4 * G::GROUP;
            bptr += G::ROW + G::PAD;
        }
        for (int j = 0; j < ksteps - ngroups * G::GROUP; j++) step(aptr, bptr, j);
    }
}

// The A fragments (taps) of a launch do not change from tile to tile: lane (m, kq) reads the same KS floats every time.  A persistent kernel with registers
// to spare keeps them (round 4, the FM receiver: KS = 51 registers): the MFMA loop then makes ONE LDS read per step instead of two - the counters of that kernel
// read LDS 55 % busy, like every kernel here whose LDS traffic turned out to matter.
// NR <= KS: the first NR steps take their A fragment from registers, the rest from LDS as before (the register file decides NR)
template <int S, int D, int KS, int TQS, int NR>
__device__ __forceinline__ void mfma_load_areg(const float *ldsT, int e, float (&areg)[NR])
{
    using G = FirMfmaGeom<S, D>;
    const int lane = threadIdx.x & 63;
    const int col = lane & 15, kq = lane >> 4;
    const float *aptr = ldsT + kq * TQS + fir_taps_zl(D) + kq - e - col * D;
#pragma unroll
    for (int s = 0; s < NR; s++) areg[s] = aptr[4 * (s / G::GROUP) * G::GROUP + 4 * (s % G::GROUP)];
}
// SPLIT = 2: even and odd steps accumulate in two registers sets that are added at the end.  A lone chain of v_mfma_f32_16x16x4_f32 on ONE accumulator is paced by
// the 40-cycle dependent latency plus the cliff an LDS read between two dependent MFMAs costs (MI355X_MICROARCH.md: +43 cycles), against 32 cycles of pipe time -
// two interleaved chains issue back to back.  The sum is no longer the single fmaf chain of the direct form: only for kernels whose contract is a tolerance
// (the FM receiver, include/lrhip.h's rounding exceptions)
#ifndef LRHIP_MFMA_AREG_DEPTH
#define LRHIP_MFMA_AREG_DEPTH 1
#endif
template <int S, int D, int NACC, int KS, int TQS, int NR, int SPLIT = 1>
__device__ __forceinline__ void mfma_tile_areg(const float (&areg)[NR], const float *ldsT, int e, const float *ldsX, f32x4 (&acc)[1][NACC])
{
    using G = FirMfmaGeom<S, D>;
    static_assert(NR <= KS, "register-resident steps");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int blk_in_acc = S == 2 ? (col >> 1) : col;
    const int comp = S == 2 ? (col & 1) : 0;
    constexpr int ACC_STRIDE = (G::ROW + G::PAD) * G::BPA;
    const float *bptr = ldsX + (G::ROW + G::PAD) * ((wave * NACC) * G::BPA + blk_in_acc) + S * kq + comp;
    const float *aptr = ldsT + kq * TQS + fir_taps_zl(D) + kq - e - col * D;
    f32x4 acc2[NACC];
#pragma unroll
    for (int a = 0; a < NACC; a++) acc[0][a] = acc2[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // software pipeline: the B (and late A) fragments of step s + DEPTH are requested while step s multiplies (DEPTH + 1 register slots).  DEPTH = 1 leaves an LDS
    // round trip between a read and the MFMA that needs it one 32-cycle step later - and 2 or 4 measure EQUAL on the receiver (0.1567 / 0.1591 / 0.1571 ms, round 4):
    // the other two waves of the SIMD fill it
    constexpr int DEPTH = LRHIP_MFMA_AREG_DEPTH, NB = DEPTH + 1;
    float bv[NB][NACC], av[NB];
    auto fetch = [&](int buf, int s) {
        const int g = s / G::GROUP, j = s % G::GROUP;
        const float *bp = bptr + g * (G::ROW + G::PAD);
#pragma unroll
        for (int a = 0; a < NACC; a++) bv[buf][a] = bp[a * ACC_STRIDE + j * 4 * S];
        if (s >= NR) av[buf] = aptr[4 * g * G::GROUP + 4 * j];
    };
#pragma unroll
    for (int s = 0; s < DEPTH && s < KS; s++) fetch(s % NB, s);
#pragma unroll
    for (int s = 0; s < KS; s++) {
        if (s + DEPTH < KS) fetch((s + DEPTH) % NB, s + DEPTH);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < NACC; a++) {
            const float af = s < NR ? areg[s < NR ? s : 0] : av[s % NB];
            if (SPLIT == 2 && (s & 1)) acc2[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bv[s % NB][a], acc2[a], 0, 0, 0);
            else acc[0][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bv[s % NB][a], acc[0][a], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (SPLIT == 2) {
#pragma unroll
        for (int a = 0; a < NACC; a++) acc[0][a] += acc2[a];
    }
}

// epilogue: accumulator lane (col, kq) holds rows 4*kq .. 4*kq+3 of column col
template <int S, int D, int NACC, int NOUT>
__device__ __forceinline__ void store_tile(float *__restrict__ y, long tile_k0, long n_out, int out_aligned, f32x4 (&acc)[NOUT][NACC])
{
    using G = FirMfmaGeom<S, D>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int blk_in_acc = S == 2 ? (col >> 1) : col;
#pragma unroll
    for (int a = 0; a < NACC; a++) {
        const long kblk = tile_k0 + 16 * (long)((wave * NACC + a) * G::BPA + blk_in_acc);
        const float a0 = acc[0][a][0], a1 = acc[0][a][1], a2 = acc[0][a][2], a3 = acc[0][a][3];
        if (NOUT == 2) {
            // complex taps over the float stream: set 0 = re, set 1 = im of the same four outputs
            const float b0 = acc[NOUT - 1][a][0], b1 = acc[NOUT - 1][a][1], b2 = acc[NOUT - 1][a][2], b3 = acc[NOUT - 1][a][3];
            long k = kblk + 4 * kq;
            if (out_aligned && k + 3 < n_out) {
                nt_store(reinterpret_cast<float4 *>(y + 2 * k), make_float4(a0, b0, a1, b1));
                nt_store(reinterpret_cast<float4 *>(y + 2 * k + 4), make_float4(a2, b2, a3, b3));
            } else {
                if (k < n_out) *reinterpret_cast<float2 *>(y + 2 * k) = make_float2(a0, b0);
                if (k + 1 < n_out) *reinterpret_cast<float2 *>(y + 2 * k + 2) = make_float2(a1, b1);
                if (k + 2 < n_out) *reinterpret_cast<float2 *>(y + 2 * k + 4) = make_float2(a2, b2);
                if (k + 3 < n_out) *reinterpret_cast<float2 *>(y + 2 * k + 6) = make_float2(a3, b3);
            }
        } else if (S == 1) {
            long k = kblk + 4 * kq;
            if (out_aligned && k + 3 < n_out) {
                nt_store(reinterpret_cast<float4 *>(y + k), make_float4(a0, a1, a2, a3));
            } else {
                if (k < n_out) y[k] = a0;
                if (k + 1 < n_out) y[k + 1] = a1;
                if (k + 2 < n_out) y[k + 2] = a2;
                if (k + 3 < n_out) y[k + 3] = a3;
            }
        } else {
            // even lane = re column, odd lane = im column of the same block: trade halves so each lane owns
            // two whole ComplexFloat32 outputs (rows 4kq+{0,1} on the even lane, 4kq+{2,3} on the odd lane)
            const bool odd = col & 1;
            float send0 = odd ? a0 : a2;
            float send1 = odd ? a1 : a3;
            float recv0 = __shfl_xor(send0, 1);
            float recv1 = __shfl_xor(send1, 1);
            float4 o = odd ? make_float4(recv0, a2, recv1, a3) : make_float4(a0, recv0, a1, recv1);
            long k = kblk + 4 * kq + (odd ? 2 : 0);
            if (out_aligned && k + 1 < n_out) {
                nt_store(reinterpret_cast<float4 *>(y + 2 * k), o);
            } else {
                if (k < n_out) *reinterpret_cast<float2 *>(y + 2 * k) = make_float2(o.x, o.y);
                if (k + 1 < n_out) *reinterpret_cast<float2 *>(y + 2 * k + 2) = make_float2(o.z, o.w);
            }
        }
    }
}

// discriminator epilogue, in registers: after the re/im exchange a lane owns two consecutive ComplexFloat32 outputs of
// its wave's WAVE_OUT = NACC*BPA*16 consecutive outputs; the output before the lane's first one lives in another lane
// (or, for lane 0, in the previous accumulator), one ds_bpermute away.  Only the first output of a WAVE needs another
// wave's data: each wave records (first, last) in `edge` and fir_disc_fixup_kernel rewrites those samples afterwards.
// The last valid output of the chunk is published as the carried previous sample.
template <int D, int NACC, bool REL = false>
__device__ __forceinline__ void disc_epilogue(float *__restrict__ y, long tile_k0, long n_out, int out_aligned, f32x4 (&acc)[1][NACC],
                                              float2 *__restrict__ edge_tile, float2 *__restrict__ prev_out, double inv_gain, cf pt = cf{1.f, 0.f},
                                              bool stream_start = false, float2 o_first = make_float2(0.f, 0.f))
{
    using G = FirMfmaGeom<2, D>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const bool odd = col & 1;
    // pt: the tile's outputs carry a common phasor the angles do not see (relative rotator staging); what LEAVES the tile as a ComplexFloat32
    // sample - the edge records, the carried previous sample - is multiplied by it
    auto leave = [&](float2 o) { return REL ? cf_to(cmul(cf_from(o), pt)) : o; };
    // lane that owns the output just before this lane's first one (lane 0: previous accumulator, handled below)
    const int src = odd ? lane - 1 : kq ? lane - 15 : col ? col + 47 : 63;
    const long wave_k0 = tile_k0 + (long)wave * (NACC * G::BPA * 16);
    float2 carry = make_float2(0.f, 0.f);          // last output of the previous accumulator
#pragma unroll
    for (int a = 0; a < NACC; a++) {
        const float a0 = acc[0][a][0], a1 = acc[0][a][1], a2 = acc[0][a][2], a3 = acc[0][a][3];
        const float recv0 = __shfl_xor(odd ? a0 : a2, 1);
        const float recv1 = __shfl_xor(odd ? a1 : a3, 1);
        const float2 o0 = odd ? make_float2(recv0, a2) : make_float2(a0, recv0);
        const float2 o1 = odd ? make_float2(recv1, a3) : make_float2(a1, recv1);
        float2 p = make_float2(__shfl(o1.x, src), __shfl(o1.y, src));
        if (lane == 0) p = carry;
        carry = make_float2(__shfl(o1.x, 63), __shfl(o1.y, 63));
        const float2 d = make_float2(discriminate(o0, p, inv_gain), discriminate(o1, o0, inv_gain));
        const long k = wave_k0 + 16 * (a * G::BPA + (col >> 1)) + 4 * kq + (odd ? 2 : 0);
        if (out_aligned && k + 1 < n_out) {
            nt_store(reinterpret_cast<float2 *>(y + k), d);
        } else {
            if (k < n_out) y[k] = d.x;
            if (k + 1 < n_out) y[k + 1] = d.y;
        }
        if (k == n_out - 1) *prev_out = leave(o0);
        if (k + 1 == n_out - 1) *prev_out = leave(o1);
        // (stream_start: the stream's first output meets the zero initial state - o[0] conj(0) is a zero product and the reference's angle is then decided by
        // the SIGNS of those zeros (discriminate()).  Under the window-relative staging o[0] is the exact output times two phasors that multiply to one: a
        // component that is zero or tiny in the unrotated arithmetic comes back as rounding noise of either sign.  That ONE record takes its direct-form
        // value, computed by the caller: h[0] x[0].)
        if (a == 0 && lane == 0) edge_tile[2 * wave] = (REL && stream_start && wave == 0) ? o_first : leave(o0);
        if (a == NACC - 1 && lane == 63) edge_tile[2 * wave + 1] = leave(o1);
    }
}

// Discriminator epilogue through LDS (LRHIP_DISC_EPI_LDS = 1; measured slower than the in-register form, kept as the A/B variant): after the re/im exchange every lane writes its two ComplexFloat32 outputs of
// each accumulator as ONE 16-byte word into the (now free) window area - tile-local output l at float 2 l, a wave writes 1 KB contiguous per
// accumulator - and after a barrier thread t turns outputs 4t .. 4t+3 (with 4t-1 from its neighbour's word) into four angles and one 16-byte store.
// 205 VALU instructions per wave and tile against 292 for the in-register form (its ds_bpermute shuffles, selects and 8-byte stores), and only the
// FIRST output of a TILE needs another workgroup's data: edge[2t] / edge[2t+1] = first / last output of tile t, fixed afterwards as before.
template <int D, int NACC>
__device__ __forceinline__ void disc_epilogue_lds(float *__restrict__ y, float *ldsO, long tile_k0, long n_out, f32x4 (&acc)[1][NACC],
                                                  float2 *__restrict__ edge_tile, float2 *__restrict__ prev_out, double inv_gain)
{
    using G = FirMfmaGeom<2, D>;
    constexpr int TILE_OUT = G::tile_out(NACC);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const bool odd = col & 1;
    __syncthreads();                                   // every wave is done reading the window: its first TILE_OUT * 8 bytes become the out-area
#pragma unroll
    for (int a = 0; a < NACC; a++) {
        const float a0 = acc[0][a][0], a1 = acc[0][a][1], a2 = acc[0][a][2], a3 = acc[0][a][3];
        const float recv0 = __shfl_xor(odd ? a0 : a2, 1);
        const float recv1 = __shfl_xor(odd ? a1 : a3, 1);
        const float4 o = odd ? make_float4(recv0, a2, recv1, a3) : make_float4(a0, recv0, a1, recv1);      // two consecutive outputs
        const int l = 16 * ((wave * NACC + a) * G::BPA + (col >> 1)) + 4 * kq + (odd ? 2 : 0);
        *reinterpret_cast<float4 *>(ldsO + 2 * l) = o;
    }
    __syncthreads();
#pragma unroll
    for (int l0 = 4 * tid; l0 < TILE_OUT; l0 += 1024) {
        const long k = tile_k0 + l0;
        if (k >= n_out) break;
        const float4 u = *reinterpret_cast<const float4 *>(ldsO + 2 * l0), v = *reinterpret_cast<const float4 *>(ldsO + 2 * l0 + 4);
        const float2 p = l0 ? *reinterpret_cast<const float2 *>(ldsO + 2 * l0 - 2) : make_float2(0.f, 0.f);       // tile-first: fixed afterwards
        const float2 o0 = make_float2(u.x, u.y), o1 = make_float2(u.z, u.w), o2 = make_float2(v.x, v.y), o3 = make_float2(v.z, v.w);
        const float4 d = make_float4(discriminate(o0, p, inv_gain), discriminate(o1, o0, inv_gain), discriminate(o2, o1, inv_gain), discriminate(o3, o2, inv_gain));
        if (k + 3 < n_out && (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
            nt_store(reinterpret_cast<float4 *>(y + k), d);
        } else {
            y[k] = d.x;
            if (k + 1 < n_out) y[k + 1] = d.y;
            if (k + 2 < n_out) y[k + 2] = d.z;
            if (k + 3 < n_out) y[k + 3] = d.w;
        }
        const long last = n_out - 1 - k;                // the chunk's last output, if it is one of these four
        if (last >= 0 && last < 4) *prev_out = last == 0 ? o0 : last == 1 ? o1 : last == 2 ? o2 : o3;
        if (l0 == 0) edge_tile[0] = o0;
        if (l0 == TILE_OUT - 4) edge_tile[1] = o3;
    }
}

// second half of the discriminator epilogue: the first output of every wave's range (wave_out outputs) needs the last
// output of the previous wave (or of the previous chunk)
__global__ __launch_bounds__(256) void fir_disc_fixup_kernel(const float2 *__restrict__ edge, long nwaves, int wave_out, float *__restrict__ y, long n_out,
                                                             const float2 *__restrict__ prev_in, double inv_gain)
{
    long w = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nwaves && w * wave_out < n_out) {
        float2 p = w ? edge[2 * (w - 1) + 1] : *prev_in;
        y[w * wave_out] = discriminate(edge[2 * w], p, inv_gain);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Decimating FIR for the decimations the Toeplitz kernel has no instantiation for (9, 25, 50, ...: the reference's own
// examples tune with TunerBlock(offset, bw, 50)), real taps, optional fused rotator.  One output per thread as an fmaf
// chain in the reference's tap order (bit-identical to the direct form), but from a tile staged ONCE in LDS - rotated once
// per sample, block-of-8 phasors - instead of 128 cached global reads per output.  OW outputs per tile (host: as many as
// DECIM_SPAN_MAX staged samples allow, at most 256), persistent over tiles.  LDS index = p + (p >> 5): the pad keeps the
// thread stride of D samples off the bank period for even D.  HBM bound: S*4 B in, S*4/D B out per input sample.
// ------------------------------------------------------------------------------------------------------------
constexpr int DECIM_SPAN_MAX = 6144;
#ifndef LRHIP_DECIM_EARLY_PREFETCH
#define LRHIP_DECIM_EARLY_PREFETCH 0
#endif
#ifndef LRHIP_DECIM_UNROLL16
#define LRHIP_DECIM_UNROLL16 1      /* tap loop of fir_decim_lds_kernel in groups of sixteen (0: four at a time, rounds 2-3) */
#endif

__device__ __forceinline__ int decim_phys(int p) { return p + (p >> 5); }

// tools/ab_decim.hip builds this header with -DLRHIP_DECIM_TRACE: lane 0 of every wave of the first workgroups stamps the phases of its first tiles
#ifdef LRHIP_DECIM_TRACE
__device__ unsigned long long *lrhip_decim_trace;       // [block][wave][tile][8]
#define DECIM_STAMP(i)                                                                                                                              \
    do {                                                                                                                                            \
        if (lrhip_decim_trace && blockIdx.x < 8 && trace_tile < 32 && (threadIdx.x & 63) == 0)                                                      \
            lrhip_decim_trace[(((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 + trace_tile) * 8 + (i)] = clock64();                               \
    } while (0)
#else
#define DECIM_STAMP(i) do { } while (0)
#endif

// FMT > 0 (round 3): x holds raw IQ-file records (RX_FMT_*), converted while the tile is staged
template <int S, bool ROT, bool CT = false, int FMT = 0>      // CT: ComplexFloat32 taps (S = 2), taps_rev as {re, im} pairs
__global__ __launch_bounds__(256) void fir_decim_lds_kernel(const float *__restrict__ hist, const float *__restrict__ x, const float *__restrict__ taps_rev,
                                                            float *__restrict__ y, int M, long n, long n_out, long first, long D, int OW, long ntiles,
                                                            uint64_t rot_step_fx, uint64_t rot_count0, float *__restrict__ hist_out, int post_op, int rounds)
{
    static_assert(FMT == 0 || (S == 2 && !CT), "raw records: (I, Q) pairs, real taps");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TS = CT ? 2 : 1;
    float *ldsT = lds;                              // M reversed taps
    float *ldsX = lds + ((TS * M + 3) & ~3);        // staged samples, S floats each, padded index
    const int tid = threadIdx.x;
    if (hist_out && blockIdx.x == 0)
        for (int i = tid; i < (M - 1) * S; i += 256) hist_out[i] = FMT ? stream_at_raw<FMT>(hist, x, n + i / S, i % S, M, n) : stream_at<S>(hist, x, n + i / S, i % S, M, n);
    for (int i = tid; i < TS * M; i += 256) ldsT[i] = taps_rev[i];
    RotTab rt;
    if (ROT) rt = rot_tab(rot_step_fx);
    const int span = (int)((OW - 1) * D) + M;
    // register prefetch (rotator form): the next tile's blocks are loaded while this tile is filtered
    constexpr int KB = (DECIM_SPAN_MAX + 14) / 8 / 256 + 1;         // blocks of 8 samples per thread, at most
    [[maybe_unused]] float4 raw[(ROT && !FMT) ? KB : 1][4];
    [[maybe_unused]] uint2 rawr[(ROT && FMT) ? KB : 1][4];       // raw records: two samples per 4- (.x) or 8-byte word
    bool have = false;
    auto prefetch = [&](long tt) {
        have = false;
        if constexpr (ROT) {
            if (tt >= ntiles) return;
            const long g0n = first + tt * OW * D - (M - 1);
            const int an = (int)((rot_count0 + (uint64_t)g0n) & 7), nblkn = (span + an + 7) >> 3;
            // blocks on 16-byte boundaries (an even absolute sample count at x[0]: every chunk the reference hands over), tile inside the chunk
            have = g0n >= 0 && g0n + span <= n && ((g0n - an) & 1) == 0 && (reinterpret_cast<uintptr_t>(x) & (FMT ? 7 : 15)) == 0 && nblkn <= KB * 256;
            if (!have) return;
#pragma unroll
            for (int k = 0; k < KB; k++) {
                const int b = tid + 256 * k, w0 = 8 * b - an;
                if (b < nblkn && w0 >= 0 && w0 + 8 <= span) {
                    if constexpr (FMT != 0) {
                        const uint8_t *src0 = reinterpret_cast<const uint8_t *>(x) + 2 * rx_raw_bytes<FMT>() * (g0n + w0);
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            if (FMT == RX_FMT_S16LE) rawr[k][q] = reinterpret_cast<const uint2 *>(src0)[q];
                            else rawr[k][q].x = reinterpret_cast<const unsigned *>(src0)[q];
                        }
                    } else {
                        const float4 *src = reinterpret_cast<const float4 *>(reinterpret_cast<const cf *>(x) + g0n + w0);
#pragma unroll
                        for (int q = 0; q < 4; q++) raw[(ROT && !FMT) ? k : 0][q] = src[q];
                    }
                }
            }
        }
    };
    // the same for the plain decimator on a ComplexFloat32 stream: lane-contiguous 16-byte words (two samples) of the next tile's window
    constexpr int KF = (DECIM_SPAN_MAX / 2 + 1 + 255) / 256;       // 16-byte words per thread, at most
    [[maybe_unused]] float4 rawp[(!ROT && S == 2 && !FMT) ? KF : 1];
    [[maybe_unused]] uint2 rawq[(!ROT && S == 2 && FMT) ? KF : 1];
    auto prefetch_plain = [&](long tt) {
        have = false;
        if constexpr (!ROT && S == 2) {
            if (tt >= ntiles) return;
            const long g0n = first + tt * OW * D - (M - 1), gb = g0n - (g0n & 1);
            const int nf = (span + (int)(g0n & 1) + 1) >> 1;
            have = gb >= 0 && gb + 2L * nf <= n && (reinterpret_cast<uintptr_t>(x) & (FMT ? 7 : 15)) == 0 && nf <= KF * 256;
            if (!have) return;
            if constexpr (FMT != 0) {
                const uint8_t *src0 = reinterpret_cast<const uint8_t *>(x) + 2 * rx_raw_bytes<FMT>() * gb;
#pragma unroll
                for (int k = 0; k < KF; k++) {
                    const int f = tid + 256 * k;
                    if (f < nf) {
                        if (FMT == RX_FMT_S16LE) rawq[k] = reinterpret_cast<const uint2 *>(src0)[f];
                        else rawq[k].x = reinterpret_cast<const unsigned *>(src0)[f];
                    }
                }
            } else {
                const float4 *src = reinterpret_cast<const float4 *>(reinterpret_cast<const cf *>(x) + gb);
#pragma unroll
                // (round 5: no branch around a load - a lane past the end repeats the last word.  A conditional load is a basic block of its own and hipcc puts
                // s_waitcnt vmcnt(0) in front of each: the prefetch went out one load at a time, kernels_firdecim.h)
                for (int k = 0; k < KF; k++) {
                    const int f = tid + 256 * k;
                    rawp[(!ROT && S == 2 && !FMT) ? k : 0] = src[f < nf ? f : nf - 1];
                }
            }
        }
    };
    // tile order: persistent grid stride (rounds == 0), or the `rounds` consecutive tiles from blockIdx.x * rounds on (workgroups in address order)
    const long t_first = rounds > 0 ? (long)blockIdx.x * rounds : (long)blockIdx.x, t_step = rounds > 0 ? 1 : (long)gridDim.x;
    const long t_end = rounds > 0 ? (t_first + rounds < ntiles ? t_first + rounds : ntiles) : ntiles;
    if (ROT) prefetch(t_first);
    else prefetch_plain(t_first);
    [[maybe_unused]] int trace_tile = 0;
    for (long t = t_first; t < t_end; t += t_step) {
        DECIM_STAMP(0);
        const long k0 = t * OW;                     // first output of the tile
        const long q0 = first + k0 * D;             // stream position of staged sample 0 (stream = [M-1 history | chunk])
        const long g0 = q0 - (M - 1);               // the same as an index into x (negative: history)
        const bool interior = g0 >= 0 && g0 + span <= n;
        if (ROT) {
            // aligned blocks of 8 absolute samples: block b covers window positions 8b - a .. 8b - a + 7
            const int a = (int)((rot_count0 + (uint64_t)g0) & 7);
            const int nblk = (span + a + 7) >> 3;
            // prefetched tiles: all of a thread's blocks were loaded as 16-byte words a tile ago (up to four blocks: one memory round trip per tile, hidden
            // behind the previous tile's filter phase, instead of one per block), and are only rotated here
            if (have) {
#pragma unroll
                for (int k = 0; k < KB; k++) {
                    const int b = tid + 256 * k, w0 = 8 * b - a;
                    if (b >= nblk) continue;
                    const cf pb = phasor_poly(rot_step_fx * (rot_count0 + (uint64_t)(g0 + w0)));
                    const bool whole = w0 >= 0 && w0 + 8 <= span;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int w = w0 + j;
                        if (w >= 0 && w < span) {
                            float4 rq;
                            if constexpr (FMT != 0) rq = rx_raw_pair<FMT>(rawr[k][j >> 1]);
                            else rq = raw[(ROT && !FMT) ? k : 0][j >> 1];
                            const cf v = whole ? ((j & 1) ? cf{rq.z, rq.w} : cf{rq.x, rq.y})
                                               : cf{stream_at_raw<FMT>(hist, x, q0 + w, 0, M, n), stream_at_raw<FMT>(hist, x, q0 + w, 1, M, n)};
                            const cf r = cmul(v, cmul(pb, rt.w[j]));
                            *reinterpret_cast<float2 *>(ldsX + 2 * decim_phys(w)) = make_float2(r.x, r.y);
                        }
                    }
                }
            } else
            for (int b = tid; b < nblk; b += 256) {
                const int w0 = 8 * b - a;
                const cf pb = phasor_poly(rot_step_fx * (rot_count0 + (uint64_t)(g0 + w0)));
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int w = w0 + j;
                    if (w >= 0 && w < span) {
                        const cf v = (interior && !FMT) ? reinterpret_cast<const cf *>(x)[g0 + w]
                                                        : cf{stream_at_raw<FMT>(hist, x, q0 + w, 0, M, n), stream_at_raw<FMT>(hist, x, q0 + w, 1, M, n)};
                        const cf r = cmul(v, cmul(pb, rt.w[j]));
                        *reinterpret_cast<float2 *>(ldsX + 2 * decim_phys(w)) = make_float2(r.x, r.y);
                    }
                }
            }
        } else if (S == 2 && have) {
            const int o = (int)(g0 & 1), nf = (span + o + 1) >> 1;
#pragma unroll
            for (int k = 0; k < KF; k++) {
                const int f = tid + 256 * k, w = 2 * f - o;
                if (f < nf) {
                    float4 v;
                    if constexpr (FMT != 0) v = rx_raw_pair<FMT>(rawq[(!ROT && S == 2) ? k : 0]);
                    else v = rawp[(!ROT && S == 2) ? k : 0];
                    if (w >= 0) *reinterpret_cast<float2 *>(ldsX + 2 * decim_phys(w)) = make_float2(v.x, v.y);
                    if (w + 1 < span) *reinterpret_cast<float2 *>(ldsX + 2 * decim_phys(w + 1)) = make_float2(v.z, v.w);
                }
            }
        } else {
            if (interior && !FMT) {
                for (int w = tid; w < span; w += 256) {
                    if (S == 2) *reinterpret_cast<float2 *>(ldsX + 2 * decim_phys(w)) = reinterpret_cast<const float2 *>(x)[g0 + w];
                    else ldsX[decim_phys(w)] = x[g0 + w];
                }
            } else {
                for (int w = tid; w < span; w += 256) {
#pragma unroll
                    for (int c = 0; c < S; c++) ldsX[S * decim_phys(w) + c] = FMT ? stream_at_raw<FMT>(hist, x, q0 + w, c, M, n) : stream_at<S>(hist, x, q0 + w, c, M, n);
                }
            }
        }
        // LRHIP_DECIM_EARLY_PREFETCH 1 (round 4, A/B): the next tile's samples requested in FRONT of the barrier, as soon as this tile's are out of the registers
        // (behind it the loads have only the tap loop to arrive in; a tile costs ~4.8 us whatever its size: time = 75 us + 430 000 us / span over
        // LRHIP_DECIM_SPAN = 1536 .. 6144).  Measured SLOWER - Tuner(.., 50) 0.1488 against 0.1434 ms, Decimator(25) 0.124 against 0.118, three alternations: off
        if (LRHIP_DECIM_EARLY_PREFETCH) {
            if (ROT) prefetch(t + t_step < t_end ? t + t_step : ntiles);
            else prefetch_plain(t + t_step < t_end ? t + t_step : ntiles);
        }
        DECIM_STAMP(1);
        __syncthreads();
        DECIM_STAMP(2);
        if (!LRHIP_DECIM_EARLY_PREFETCH) {
            if (ROT) prefetch(t + t_step < t_end ? t + t_step : ntiles);
            else prefetch_plain(t + t_step < t_end ? t + t_step : ntiles);
        }
        DECIM_STAMP(3);
        const long k = k0 + tid;
        if (tid < OW && k < n_out) {
            int p = tid * (int)D;
            float re = 0.f, im = 0.f;
            int tt = 0;
            if constexpr (CT) {
                // complex taps: the operation order of fir_direct_kernel<2> (re += xr hr; re += xi (-hi); im += xr hi; im += xi hr)
                for (; tt < M; tt++) {
                    const float2 h = *reinterpret_cast<const float2 *>(ldsT + 2 * tt);
                    const float2 xv = *reinterpret_cast<const float2 *>(ldsX + 2 * decim_phys(p + tt));
                    re = fmaf(xv.x, h.x, re);
                    re = fmaf(xv.y, -h.y, re);
                    im = fmaf(xv.x, h.y, im);
                    im = fmaf(xv.y, h.x, im);
                }
            }
            // sixteen taps at a time: all sixteen window reads and the four 16-byte tap reads are issued before the first FMA.  (Round 4: with four taps per
            // trip the loop ran at ~90 cycles per tap - one exposed LDS round trip per group, 11.7 k cycles per tile however few samples the tile held; the
            // fmaf chain itself is unchanged: ascending taps, one accumulator.)  Sixteen consecutive window samples cross at most one padded row of 32
            if constexpr (LRHIP_DECIM_UNROLL16) {
                for (; tt + 16 <= M; tt += 16) {
                    float hh[16];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float4 h4 = *reinterpret_cast<const float4 *>(ldsT + tt + 4 * q);
                        hh[4 * q] = h4.x; hh[4 * q + 1] = h4.y; hh[4 * q + 2] = h4.z; hh[4 * q + 3] = h4.w;
                    }
                    const int q0 = p + tt, ql = q0 & 31, b0 = q0 + (q0 >> 5);
                    float xr[16], xi[16];
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        const int ix = b0 + j + ((ql + j) >> 5);
                        if (S == 2) {
                            const float2 v = *reinterpret_cast<const float2 *>(ldsX + 2 * ix);
                            xr[j] = v.x; xi[j] = v.y;
                        } else {
                            xr[j] = ldsX[ix]; xi[j] = 0.f;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        re = fmaf(xr[j], hh[j], re);
                        if (S == 2) im = fmaf(xi[j], hh[j], im);
                    }
                }
            }
            for (; tt + 4 <= M; tt += 4) {
                const float h0 = ldsT[tt], h1 = ldsT[tt + 1], h2 = ldsT[tt + 2], h3 = ldsT[tt + 3];
                const int i0 = decim_phys(p + tt), i1 = decim_phys(p + tt + 1), i2 = decim_phys(p + tt + 2), i3 = decim_phys(p + tt + 3);
                if (S == 2) {
                    const float2 x0 = *reinterpret_cast<const float2 *>(ldsX + 2 * i0), x1 = *reinterpret_cast<const float2 *>(ldsX + 2 * i1);
                    const float2 x2 = *reinterpret_cast<const float2 *>(ldsX + 2 * i2), x3 = *reinterpret_cast<const float2 *>(ldsX + 2 * i3);
                    re = fmaf(x0.x, h0, re); im = fmaf(x0.y, h0, im);
                    re = fmaf(x1.x, h1, re); im = fmaf(x1.y, h1, im);
                    re = fmaf(x2.x, h2, re); im = fmaf(x2.y, h2, im);
                    re = fmaf(x3.x, h3, re); im = fmaf(x3.y, h3, im);
                } else {
                    re = fmaf(ldsX[i0], h0, re);
                    re = fmaf(ldsX[i1], h1, re);
                    re = fmaf(ldsX[i2], h2, re);
                    re = fmaf(ldsX[i3], h3, re);
                }
            }
            for (; tt < M; tt++) {
                const float h0 = ldsT[tt];
                const int i0 = decim_phys(p + tt);
                re = fmaf(ldsX[S * i0], h0, re);
                if (S == 2) im = fmaf(ldsX[2 * i0 + 1], h0, im);
            }
            // post_op = 1 + a complex -> real element-wise operation folded into the store (ComplexMagnitude behind the AM receiver's tuner ...): Float32 out
            if (S == 2 && post_op) y[k] = post_op == 1 + UN_CMAG ? unary_c2r<UN_CMAG>(re, im) : post_op == 1 + UN_CPHASE ? unary_c2r<UN_CPHASE>(re, im) : post_op == 1 + UN_CREAL ? re : im;
            else if (S == 2) nt_store(reinterpret_cast<float2 *>(y) + k, make_float2(re, im));
            else y[k] = re;
        }
        DECIM_STAMP(4);
        __syncthreads();
        DECIM_STAMP(5);
#ifdef LRHIP_DECIM_TRACE
        trace_tile++;
#endif
    }
}

// ------------------------------------------------------------------------------------------------------------
// Polyphase rational resampler: [MultiplyConstant(c)] -> Upsampler(L) -> FIR(real taps h, M) -> [Downsampler(D)]
// (radio/composites/interpolator.lua:31-34, rationalresampler.lua:37-44, upsampler.lua:45-53) without the zero-stuffed
// stream.  Output m sits at upsampled position n = m*D; with p = n mod L, q = n div L
//     y[m] = sum_{j >= 0, p + jL < M} h[p + jL] * (c * x[q - j])
// - exactly the nonzero terms of the zero-stuffed direct form, accumulated in the same (ascending time = descending j)
// order with fmaf, so the result is bit-identical to the unfused chain (a zero sample adds exactly +-0).
// One output per thread; a workgroup stages the input span of its 256 outputs and the taps in LDS.  q0 / Q0 are absolute
// sample counts, the history holds the HQ input samples before the chunk.
// ------------------------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void fir_resample_kernel(const float *__restrict__ hist, const float *__restrict__ x,
                                                           const float *__restrict__ taps, float *__restrict__ y, int M, int L, long D,
                                                           long n_in, long n_out, uint64_t m0, uint64_t Q0, int HQ, float c, int span_max,
                                                           float *__restrict__ hist_out)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int jmax_all = (M - 1) / L;
    const int TP = (jmax_all + 1) * L;     // taps, zero-padded to a whole number of phases
    float *ldsT = lds;
    float *ldsX = lds + ((TP + 3) & ~3);   // staged samples (scaled), S floats each, 16-B aligned
    const int tid = threadIdx.x;
    // history carry: last HQ raw input samples of [hist | x]
    if (hist_out && blockIdx.x == 0)
        for (int i = tid; i < HQ * S; i += 256) {
            long g = n_in - HQ + i / S;            // index into x (negative: history)
            hist_out[i] = g >= 0 ? x[g * S + i % S] : hist[(g + HQ) * S + i % S];
        }
    for (int i = tid; i < TP; i += 256) ldsT[i] = i < M ? taps[i] : 0.f;
    const long mb = (long)blockIdx.x * 256;                   // first output of this workgroup (chunk-relative)
    // absolute input index range needed: [qlo - jmax_all, qhi]
    const uint64_t nlo = (m0 + (uint64_t)mb) * (uint64_t)D;
    const long mlast = mb + 255 < n_out - 1 ? mb + 255 : n_out - 1;
    const uint64_t nhi = (m0 + (uint64_t)mlast) * (uint64_t)D;
    const long qbase = (long)(nlo / L) - jmax_all;            // absolute index of ldsX[0] (may be < Q0 - HQ: zeros)
    const int span = (int)((long)(nhi / L) - qbase + 1);
    for (int i = tid; i < span && i < span_max; i += 256) {
        long g = qbase + i - (long)Q0;                        // chunk-relative input index
#pragma unroll
        for (int cc = 0; cc < S; cc++) {
            float v = 0.f;
            if (g >= 0) { if (g < n_in) v = x[g * S + cc]; }
            else if (g + HQ >= 0) v = hist[(g + HQ) * S + cc];
            ldsX[i * S + cc] = v * c;                          // multiplyconstant.lua: Float32 product, rounded once
        }
    }
    __syncthreads();
    const long m = mb + tid;
    if (m >= n_out) return;
    const uint64_t n = (m0 + (uint64_t)m) * (uint64_t)D;
    const int p = (int)(n % L);
    const int qi = (int)((long)(n / L) - qbase);              // LDS index of x[q]
    // uniform trip count: the tap array is zero-padded to (jmax_all + 1) * L entries, and fmaf(x, 0, acc) == acc
    float re = 0.f, im = 0.f;
    const float *tp = ldsT + p + jmax_all * L;
    const float *xp = ldsX + (qi - jmax_all) * S;
    int j = jmax_all + 1;
    for (; j >= 4; j -= 4) {
        float h0 = tp[0], h1 = tp[-L], h2 = tp[-2 * L], h3 = tp[-3 * L];
        if (S == 2) {
            float2 x0 = *reinterpret_cast<const float2 *>(xp), x1 = *reinterpret_cast<const float2 *>(xp + 2);
            float2 x2 = *reinterpret_cast<const float2 *>(xp + 4), x3 = *reinterpret_cast<const float2 *>(xp + 6);
            re = fmaf(x0.x, h0, re); im = fmaf(x0.y, h0, im);
            re = fmaf(x1.x, h1, re); im = fmaf(x1.y, h1, im);
            re = fmaf(x2.x, h2, re); im = fmaf(x2.y, h2, im);
            re = fmaf(x3.x, h3, re); im = fmaf(x3.y, h3, im);
        } else {
            re = fmaf(xp[0], h0, re);
            re = fmaf(xp[1], h1, re);
            re = fmaf(xp[2], h2, re);
            re = fmaf(xp[3], h3, re);
        }
        tp -= 4 * L;
        xp += 4 * S;
    }
    for (; j >= 1; j--) {
        float h0 = tp[0];
        re = fmaf(xp[0], h0, re);
        if (S == 2) im = fmaf(xp[1], h0, im);
        tp -= L;
        xp += S;
    }
    if (S == 2) reinterpret_cast<float2 *>(y)[m] = make_float2(re, im);
    else y[m] = re;
}

// ---- generic kernel: run-time number of MFMA steps, one tile per workgroup ----------------------------------------
// NOUT = 2 (S = 1 geometry over the interleaved float stream, two Toeplitz tables): complex taps.
// HILB (S = 1, D = 1, real taps): HilbertTransformBlock in one launch (radio/blocks/signal/hilberttransform.lua:107-124: one loop writes the delayed
// input as the real part and the filtered input as the imaginary part): the output is ComplexFloat32, out[k] = (s[(M-1)/2 + k], fir[k]) with
// s = [M-1 history | chunk] - the delayed sample is already in the tile's staged window, (M-1)/2 + slack floats behind the output's first tap.
template <int S, int D, int NACC, bool ROT, int NOUT, bool HILB = false>
__global__ __launch_bounds__(256) void fir_mfma_kernel(
    const float *__restrict__ hist, const float *__restrict__ x, const float *__restrict__ taps_pad, float *__restrict__ y,
    int M, long n, long n_out, long first, int e, int ksteps, int out_aligned,
    uint64_t rot_step_fx, uint64_t rot_count0)
{
    using G = FirMfmaGeom<S, D>;
    constexpr int TILE_OUT = G::tile_out(NACC);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tlen = fir_taps_len(D, ksteps);
    float *ldsT = lds;                       // NOUT zero-padded tap arrays
    float *ldsX = lds + NOUT * tlen;         // staged samples (padded rows)
    const int tid = threadIdx.x;
    RotTab rot_t;
    if (ROT) rot_t = rot_tab(rot_step_fx);
    const long tile_k0 = (long)blockIdx.x * TILE_OUT;
    const long base = first + tile_k0 * D - e;               // stream position of r = 0
    const int span = G::span(NACC, ksteps);                   // samples to stage
    const int nf4 = span * S / 4;

    for (int i = tid; i < NOUT * tlen; i += 256) ldsT[i] = taps_pad[i];

    const long xlo = base - (M - 1);                          // x index of r = 0
    const bool interior = (xlo >= 0) && (xlo + span <= n);
    if (interior) {
        const float4 *src = reinterpret_cast<const float4 *>(x + xlo * S);   // 16-B aligned by choice of e
        constexpr int UX = 4;                                  // loads in flight per thread per batch
        for (int i0 = tid; i0 < nf4; i0 += 256 * UX) {
            float4 vv[UX];
#pragma unroll
            for (int u = 0; u < UX; u++) {
                int idx = i0 + u * 256;
                vv[u] = src[idx < nf4 ? idx : nf4 - 1];       // clamped, unconditional: no branch between loads
            }
#pragma unroll
            for (int u = 0; u < UX; u++) {
                const int i4 = i0 + u * 256;
                if (i4 < nf4) lds_put4<S, D>(ldsX, i4, ROT ? rotate_pair(vv[u], rot_step_fx, rot_count0 + (uint64_t)(xlo + 2 * (long)i4), rot_t) : vv[u]);
            }
        }
    } else {
        stage_edge<S, D, ROT>(ldsX, hist, x, base, span, M, n, rot_step_fx, rot_count0);
    }
    __syncthreads();
    f32x4 acc[NOUT][NACC];
    mfma_tile<S, D, NACC, 0, NOUT>(ldsT, tlen, e, ldsX, ksteps, acc);
    if constexpr (HILB) {
        static_assert(S == 1 && D == 1 && NOUT == 1 && !ROT, "Hilbert epilogue: Float32 stream, real taps");
        const int lane = tid & 63, wave = tid >> 6, col = lane & 15, kq = lane >> 4;
        const int half = (M - 1) / 2;
#pragma unroll
        for (int a = 0; a < NACC; a++) {
            const int l = 16 * ((wave * NACC + a) * G::BPA + col) + 4 * kq;      // tile-local index of the lane's first output
            const long k = tile_k0 + l;
            float dl[4];
#pragma unroll
            for (int i = 0; i < 4; i++) dl[i] = ldsX[G::phys(half + e + l + i)];
            float *o = y + 2 * k;
            if (out_aligned && k + 3 < n_out) {
                *reinterpret_cast<float4 *>(o) = make_float4(dl[0], acc[0][a][0], dl[1], acc[0][a][1]);
                *reinterpret_cast<float4 *>(o + 4) = make_float4(dl[2], acc[0][a][2], dl[3], acc[0][a][3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (k + i < n_out) *reinterpret_cast<float2 *>(o + 2 * i) = make_float2(dl[i], acc[0][a][i]);
            }
        }
    } else {
        store_tile<S, D, NACC, NOUT>(y, tile_k0, n_out, out_aligned, acc);
    }
}

// ---- persistent kernel: compile-time number of MFMA steps KS -----------------------------------------------------------
// One workgroup per resident slot walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...  The Toeplitz table is
// loaded into LDS once.  The NEXT tile's samples are fetched into registers (UX float4 per thread) right before
// the current tile's MFMA loop, so HBM latency hides under ~9k cycles of matrix work; they are written to LDS
// after the loop.  Edge tiles (first tile: history; last tiles: end of chunk) are staged synchronously.
// EPI = 1 (S = 2, real taps): fused FrequencyDiscriminatorBlock BEHIND the filter (frequencydiscriminator.lua:68-88): the
// ComplexFloat32 outputs never leave the registers, y receives arg(o[k] conj(o[k-1])) / gain as Float32 (disc_epilogue).
// FMT > 0 (round 3): x holds raw IQ-file records (u8 / s8 / s16le, RX_FMT_*) converted on the way into LDS - the plain Tuner / Decimator (no discriminator) only
template <int S, int D, int NACC, bool ROT, int KS, int EPI = 0, bool REL = false, int NW = 4, int FMT = 0>
__global__ __launch_bounds__(64 * NW, LRHIP_FIR_WAVES_PER_SIMD) void fir_mfma_persistent_kernel(
    const float *__restrict__ hist, const float *__restrict__ x, const float *__restrict__ taps_pad, float *__restrict__ y,
    int M, long n, long n_out, long first, int e, long ntiles, int out_aligned,
    uint64_t rot_step_fx, uint64_t rot_count0, float2 *__restrict__ edge, float2 *__restrict__ prev_out, double inv_gain, float *__restrict__ hist_out, int rounds)
{
    using G = FirMfmaGeom<S, D>;
    // NW = 1: one wave per workgroup, three such workgroups per SIMD - every wave stages the window of its own 256 outputs (7 % more samples staged
    // than a quarter of the four-wave window) and no barrier couples it to other waves
    constexpr int NT = 64 * NW;
    static_assert(NW == 4 || (NW == 1 && REL), "one-wave workgroups: relative rotator staging + discriminator epilogue only");
    constexpr int TILE_OUT = G::tile_out(NACC, NW);
    constexpr int SPAN = G::span(NACC, KS, NW);
    static_assert(FMT == 0 || (S == 2 && !REL && EPI == 0 && NW == 4), "raw records: the plain Tuner / Decimator instantiations");
    // history carry (fir_history_kernel's job, saved launch): the other ping-pong buffer, raw (unrotated) samples
    if (hist_out && blockIdx.x == 0)
        for (int i = threadIdx.x; i < (M - 1) * S; i += NT) hist_out[i] = FMT ? stream_at_raw<FMT>(hist, x, n + i / S, i % S, M, n) : stream_at<S>(hist, x, n + i / S, i % S, M, n);
    constexpr int NF4 = SPAN * S / 4;
    constexpr int UX = (NF4 + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TLEN = fir_taps_len(D, KS);
    float *ldsT = lds;
    float *ldsX = lds + TLEN;
    const int tid = threadIdx.x;

    for (int i = tid; i < TLEN; i += NT) ldsT[i] = taps_pad[i];
    // (round 4, measured EQUAL and dropped: the plain decimator's A fragments in registers as in the FM receiver's tuner loop - 0.1240 against 0.1230 ms for
    // Decimator(5) on 2^26 samples, three alternations; this kernel is not LDS-bound)

    auto xlo_of = [&](long t) { return first + t * (long)TILE_OUT * D - e - (M - 1); };
    auto interior = [&](long t) { long lo = xlo_of(t); return t < ntiles && lo >= 0 && lo + SPAN <= n; };

    // Thread -> float4 mapping of the staged window (NF4 float4).  Plain: i4 = tid + 256 u.  With the fused rotator a thread
    // owns whole ALIGNED blocks of 8 samples (4 consecutive float4, block b = tid + 256 v starting at absolute sample index
    // = 0 mod 8), so one phasor polynomial serves 8 samples; the block grid is shifted by a4 float4 against the window, the
    // same for every tile because the tile advance TILE_OUT * D is a multiple of 8 samples.
    static_assert(!ROT || (S == 2 && (TILE_OUT * D) % 8 == 0), "rotator staging: complex stream, tile advance = 0 mod 8 samples");
    // Tuner with the discriminator epilogue (REL): the window is rotated RELATIVE to its first sample, W(k) = exp(j omega k), k = 2 tid + 512 u + j
    // for float4 tid + 256 u - the same for every tile, so a thread keeps its 2 UX phasors in registers (the block-of-8 staging needs four more
    // prefetch registers and two phasor tables): ONE packed complex multiply per sample, no polynomial per tile.  The filter is linear, so every output of the tile
    // carries the common phasor P(window start); an angle arg(o[k] conj(o[k-1])) does not see it, and the samples that leave the tile (edge records,
    // carried previous sample) are multiplied by it in the epilogue.  The rounding now depends on where the tiles fall: chunkings agree to Float32
    // rounding of the filter outputs instead of bit for bit (FirStage::align() names the grid for time partitions; LRHIP_TUNER_EXACT=1 in the
    // environment keeps the block-of-8 staging, whose phasors are those of the stand-alone FrequencyTranslatorBlock bit for bit).
    static_assert(!REL || (ROT && EPI != 0 && !LRHIP_DISC_EPI_LDS), "relative rotator staging: rotator + in-register discriminator epilogue");
    constexpr int NB = ROT && !REL ? NF4 / 4 + 2 : 0;         // blocks that can touch the window
    constexpr int UB = (NB + NT - 1) / NT;
    constexpr int NPRE = ROT && !REL ? 4 * UB : UX;
    RotTab rot_t;
    int a4 = 0;
    // odd absolute offset (an odd number of samples consumed so far): float4 pairs straddle the blocks; same thread mapping
    // with a4 = 0 and the general rotate_pair (correct, slower - the reference's own chunk sizes are even)
    const bool rot_blocks = ROT && (((rot_count0 + (uint64_t)xlo_of(0)) & 1) == 0);
    if (ROT && !REL) {
        rot_t = rot_tab(rot_step_fx);
        if (rot_blocks) a4 = (int)(((rot_count0 + (uint64_t)xlo_of(0)) & 7) >> 1);      // window float4 0 is float4 a4 of its block
    }
    auto i4_of = [&](int u) { return ROT && !REL ? 4 * (tid + NT * (u >> 2)) + (u & 3) - a4 : tid + NT * u; };
    cf rel_w[REL ? UX : 1][2];
    if constexpr (REL) {
        // rel_window_phasor(2 (tid + NT u) + j) = P(2 tid) * P(2 NT u + j): the second factor is wave-uniform - lane 2u + j evaluates it once and the
        // others fetch it (2 polynomials per thread instead of 2 + 2 UX; same operands, same bits as the per-sample form of the edge tiles)
        static_assert(2 * UX <= 64, "one lane per uniform phasor");
        const int l = tid & 63;
        const cf pt = phasor_poly(rot_step_fx * (uint64_t)(2 * tid)), pu = phasor_poly(rot_step_fx * (uint64_t)(2 * NT * (l >> 1) + (l & 1)));
#pragma unroll
        for (int u = 0; u < UX; u++)
#pragma unroll
            for (int j = 0; j < 2; j++) rel_w[u][j] = cmul(pt, cf{__shfl(pu.x, 2 * u + j), __shfl(pu.y, 2 * u + j)});
    }

    // Tile order: plain stride, or (LRHIP_FIR_XCD_MAP) one contiguous eighth of the tiles per XCD - workgroup b is observed to run on XCD b % 8
    // (nothing promises it; only speed depends on it), its workgroups stride through their eighth side by side, and the M - 1 samples two
    // neighbouring tiles share come from HBM once and from that XCD's L2 the second time.
    const bool xmap = LRHIP_FIR_XCD_MAP && (gridDim.x & 7) == 0;
    const long xt8 = (ntiles + 7) >> 3, xper = gridDim.x >> 3;
    const long xbase = (long)(blockIdx.x & 7) * xt8, xj = blockIdx.x >> 3;
    auto tile_of = [&](long k) -> long {
        // rounds > 0 (round 4): one-shot order - workgroup b owns the `rounds` consecutive tiles from b * rounds on, workgroups handed out in address order
        if (rounds > 0) return k < rounds ? (long)blockIdx.x * rounds + k : ntiles;
        if (!xmap) return blockIdx.x + k * (long)gridDim.x;
        const long tl = xj + k * xper;
        return tl < xt8 ? xbase + tl : ntiles;
    };
    float4 pre[FMT ? 1 : NPRE];
    uint2 praw[FMT ? NPRE : 1];                             // raw records: the same two samples per load are 4 (.x) or 8 bytes
    long tk = 0;
    long t = tile_of(0);
    bool have = false;
    auto prefetch = [&](long tt) {
        have = interior(tt);
        if (have && FMT) {
            const uint8_t *src0 = reinterpret_cast<const uint8_t *>(x) + 2 * rx_raw_bytes<FMT>() * xlo_of(tt);
#pragma unroll
            for (int u = 0; u < NPRE; u++) {
                const int idx = i4_of(u), ic = idx < 0 ? 0 : idx < NF4 ? idx : NF4 - 1;
                if (FMT == RX_FMT_S16LE) praw[FMT ? u : 0] = reinterpret_cast<const uint2 *>(src0)[ic];
                else praw[FMT ? u : 0].x = reinterpret_cast<const unsigned *>(src0)[ic];
            }
        } else if (have) {
            const float4 *src = reinterpret_cast<const float4 *>(x + xlo_of(tt) * S);
#pragma unroll
            for (int u = 0; u < NPRE; u++) {
                int idx = i4_of(u);
                pre[FMT ? 0 : u] = src[idx < 0 ? 0 : idx < NF4 ? idx : NF4 - 1];        // clamped, unconditional: no branch between loads
            }
        }
    };
    // staged value u of the thread as ComplexFloat32 pairs
    auto pre_of = [&](int u) -> float4 { if constexpr (FMT != 0) return rx_raw_pair<FMT>(praw[u]); else return pre[u]; };
    prefetch(t);
    for (; t < ntiles; t = tile_of(++tk)) {
        const long tile_k0 = t * (long)TILE_OUT;
        if (have) {
            const long xlo = xlo_of(t);
            if constexpr (REL) {
#pragma unroll
                for (int u = 0; u < UX; u++) {
                    const int i4 = tid + u * NT;
                    const cf a = cmul(cf{pre[u].x, pre[u].y}, rel_w[u][0]), b = cmul(cf{pre[u].z, pre[u].w}, rel_w[u][1]);
                    if (i4 < NF4) lds_put4<S, D>(ldsX, i4, make_float4(a.x, a.y, b.x, b.y));
                }
            } else if constexpr (ROT) {
#pragma unroll
                for (int v = 0; v < UB; v++) {
                    const int i40 = 4 * (tid + NT * v) - a4;                 // first float4 of this thread's block
                    if (!rot_blocks) {
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (i40 + j < NF4)
                                lds_put4<S, D>(ldsX, i40 + j, rotate_pair(pre_of(4 * v + j), rot_step_fx, rot_count0 + (uint64_t)(xlo + 2 * (long)(i40 + j)), rot_t));
                    } else if (i40 + 3 >= 0 && i40 < NF4) {
                        const cf p = phasor_poly(rot_step_fx * (rot_count0 + (uint64_t)(xlo + 2 * (long)i40)));
                        if (i40 >= 0) lds_put4<S, D>(ldsX, i40, rotate_in_block<0>(pre_of(4 * v), p, rot_t));
                        if (i40 + 1 >= 0 && i40 + 1 < NF4) lds_put4<S, D>(ldsX, i40 + 1, rotate_in_block<1>(pre_of(4 * v + 1), p, rot_t));
                        if (i40 + 2 >= 0 && i40 + 2 < NF4) lds_put4<S, D>(ldsX, i40 + 2, rotate_in_block<2>(pre_of(4 * v + 2), p, rot_t));
                        if (i40 + 3 < NF4) lds_put4<S, D>(ldsX, i40 + 3, rotate_in_block<3>(pre_of(4 * v + 3), p, rot_t));
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < UX; u++) {
                    const int i4 = tid + u * NT;
                    if (i4 < NF4) lds_put4<S, D>(ldsX, i4, pre_of(u));
                }
            }
        } else {
            stage_edge<S, D, ROT, REL, NT, FMT>(ldsX, hist, x, first + tile_k0 * D - e, SPAN, M, n, rot_step_fx, rot_count0);
        }
        __syncthreads();
        // prefetch the next tile while this one is multiplied
        prefetch(tile_of(tk + 1));
        f32x4 acc[1][NACC];
        mfma_tile<S, D, NACC, KS, 1>(ldsT, TLEN, e, ldsX, KS, acc);
        if constexpr (EPI == 0) {
            store_tile<S, D, NACC, 1>(y, tile_k0, n_out, out_aligned, acc);
        } else {
            static_assert(S == 2, "discriminator epilogue: complex stream");
#if LRHIP_DISC_EPI_LDS
            disc_epilogue_lds<D, NACC>(y, ldsX, tile_k0, n_out, acc, edge + 2 * t, prev_out, inv_gain);
#else
            cf pt = cf{1.f, 0.f};
            if constexpr (REL) pt = phasor_poly(rot_step_fx * (rot_count0 + (uint64_t)xlo_of(t)));
            // the stream's first output in direct form (history is zero there: one product per component, the fmaf chain's value)
            const bool stream_start = REL && t == 0 && rot_count0 == 0 && first == 0 && n > 0;
            float2 o_first = make_float2(0.f, 0.f);
            if constexpr (REL) {
                if (stream_start) {
                    const float h0 = ldsT[fir_taps_zl(D) + M - 1];
                    const float2 x0 = rotate_sample(*reinterpret_cast<const float2 *>(x), rot_step_fx, (uint64_t)0);
                    o_first = make_float2(fmaf(h0, x0.x, 0.f), fmaf(h0, x0.y, 0.f));
                }
            }
            disc_epilogue<D, NACC, REL>(y, tile_k0, n_out, out_aligned, acc, edge + 2 * NW * t, prev_out, inv_gain, pt, stream_start, o_first);
#endif
        }
        __syncthreads();      // everyone is done reading ldsX before it is overwritten
    }
}

// Host: number of MFMA steps, K = emax + 15*D + M rounded up to a multiple of 4 (the zero padding of the tap array
// supplies the zero rows; fma(0, b, acc) == acc), and the zero-padded reversed taps the kernels read.
inline int fir_mfma_ksteps(int M, int D, int S, int emax_override = -1)
{
    int emax = emax_override >= 0 ? emax_override : 4 / S - 1;
    return (emax + 15 * D + M + 3) / 4;
}
inline void fir_mfma_build_taps(const float *taps_rev, int M, int D, int ksteps, std::vector<float> &out)
{
    out.assign((size_t)fir_taps_len(D, ksteps), 0.f);
    for (int j = 0; j < M; j++) out[(size_t)fir_taps_zl(D) + j] = taps_rev[j];
}

}  // namespace lrhip
