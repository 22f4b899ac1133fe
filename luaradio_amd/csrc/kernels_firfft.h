// kernels_firfft.h - FIRFilterBlock by FFT overlap-save (the reference's production default,
// radio/blocks/signal/firfilter.lua:320-398: N = 2^floor(log2(8M)), L = N-M+1, per block: DFT, multiply by the
// taps' DFT, IDFT, keep the last L outputs) as ONE fused kernel: load -> 1024-point FFT -> x H -> inverse FFT ->
// store, everything between the global load and the global store in registers and LDS.
//
// Why: direct form is 4M flop per cf32 sample (512 at M = 128; matrix-pipe bound at ~39 % of the HBM roof);
// overlap-save is ~125 flop per sample, which makes the block HBM-bound.
//
// One wave (64 lanes x 16 points) transforms one N = 1024 block; a 256-thread workgroup runs four blocks at a
// time and walks the block list persistently.  1024 = 16 x 16 x 4:
//   forward (decimation in frequency), lane t holds x[64*n1 + t]:
//     radix-16 over n1 -> k1;  x W_1024^(t*k1);   exchange E1;   lane (t2,k1) holds [4*t1 + t2]
//     radix-16 over t1 -> k2;  x W_64^(t2*k2);    exchange E2;   lane (q,k1) holds k2 = 4j+q, t2 = 0..3
//     radix-4  over t2 -> k3:  X[k1 + 16*k2 + 256*k3]
//   multiply by H (host-permuted to this register/lane order, 1/N folded in)
//   inverse = the mirror image with conjugated twiddles; the result lands as y[64*n1 + t] in natural order,
//   so loads and stores are both coalesced and no bit-reversal pass exists.
// Exchanges go through a per-wave LDS buffer (no workgroup barrier: a wave's DS operations execute in order).
// Stages 2 and 3 number their lanes (k1 = lane >> 2, t2 or q = lane & 3).  LDS layouts are padded so that every
// ds_read_b64 (32-lane groups, 32 bank pairs) AND every ds_write_b64 (16-lane groups, 16 bank pairs) is
// conflict-free:
//     E1 element(k1, t)      = 68*k1 + t
//     E2 element(k1, k2, t2) = 68*k1 + 17*t2 + k2
// Twiddles and H live in LDS tables indexed [register][lane] (or broadcast), loaded once per workgroup.
//
// Accuracy: f32 FFT arithmetic, not the fmaf chain of the direct form; error vs the f64 oracle is ~3e-7 for
// |x| <= 1 and unity-gain taps (tests hold it to the reference's 1e-6).
#pragma once
#ifndef LRHIP_FFT_NT
#define LRHIP_FFT_NT 1      /* 1 = non-temporal output stores (same-box A/B at 2^28 samples: 0.849 against 0.855 ms), 2 = + non-temporal loads (0.871: the 12.5 % block overlap is re-read from L2) */
#endif
#include "common.h"
#include "kernels_fir.h"
#include "pk_math.h"

// 1: exchange re and im planes one after the other through a half-size per-wave buffer (34 KB of LDS per
//    workgroup -> 4 workgroups = 16 waves per CU); 0: one ds_*_b64 pass (52 KB -> 12 waves per CU)
#ifndef LRHIP_FFT_SPLIT
#define LRHIP_FFT_SPLIT 0
#endif
// waves per workgroup of fir_fft_kernel: the 17 KB of twiddle / H tables are per workgroup, so larger workgroups fit more
// waves per CU (4: 3 x 4 = 12 waves, 16: one 1024-thread workgroup = 16 waves)
#ifndef LRHIP_FFT_WPB
#define LRHIP_FFT_WPB 4
#endif
// 1: the next block's 16 loads are issued into spare registers before this block's arithmetic
#ifndef LRHIP_FFT_PREFETCH
#define LRHIP_FFT_PREFETCH 0
#endif

// 1: the 15 stage-1 twiddles W_1024^(lane k1) of a lane stay in registers for the whole launch (30 registers; the kernel has 131 of the 168 that three waves
// per SIMD allow) instead of being read from the LDS table twice per block: 30 of a block's 201 eight-byte LDS operations.  Round 4: the counters of this kernel
// say LDS 50 % + VALU 41 % busy, the pattern of every overlap-save kernel here (section 4.7) - LDS traffic is worth removing
#ifndef LRHIP_FFT_STRAIGHT_STORES
#define LRHIP_FFT_STRAIGHT_STORES 0      /* round 4 A/B, measured EQUAL with settled clocks (below): off */
#endif
#ifndef LRHIP_FFT_TW_REG
#define LRHIP_FFT_TW_REG 1
#endif

// 2 (A/B, round 4): ALL of a lane's table values in registers - tw1 (15), tw2 forward (15), tw2 inverse (12), H (16): 116 registers, which needs two waves
// per SIMD instead of three (256 registers each); a block then moves 128 eight-byte LDS operations (the four exchanges) instead of 201
#if LRHIP_FFT_TW_REG == 2
#define LRHIP_FFT_ALLREG 1
#else
#define LRHIP_FFT_ALLREG 0
#endif

// blocks per wave and iteration (round 3 A/B): 2 = two independent 1024-point pipelines interleaved stage by stage in one wave, each with its own
// exchange buffer, so that the LDS exchanges and global loads of one block can overlap the butterflies of the other ("dependent phases" hypothesis); needs
// LRHIP_FFT_WPB = 8 (one 512-thread workgroup = 8 waves per CU: 8 x 2 x 8.7 KB of exchange buffers + 17 KB of tables)
#ifndef LRHIP_FFT_SETPRIO
#define LRHIP_FFT_SETPRIO 0      /* A/B (round 6): 1..3 = s_setprio around a block's loads and around its stores (+4: the priority stays up from the stores through the next loads).
                                    Measured EQUAL: 0.8495 / 0.8522 / 0.8528 ms (off) against 0.8514 / 0.8543 / 0.8498 (2) and 0.8463 / 0.8507 / 0.8483 (6), one box */
#endif
#ifndef LRHIP_FFT_NB
#define LRHIP_FFT_NB 1
#endif

// Round 5 ("one structural try on the headline"): 128 of a block's 171 LDS operations are the four register <-> LDS transposes, and the two
// inner ones (E2 and its inverse) only move data between FOUR lanes that share k1.  1: stages 2 and 3 number their lanes (t2 or q = lane >> 4, k1 = lane & 15) -
// the four lanes of a group then sit in the four ROWS of 16 lanes of the wave, at the same position - and E2 / its inverse become 4 x 4 register <-> row
// transposes in the VALU (v_permlane32_swap + v_permlane16_swap: two instructions per register PAIR, no selects, no LDS): 64 swaps replace 64 ds_write_b64 +
// 64 ds_read_b64 per block.  E1 and its inverse keep their LDS transposes with row strides re-derived for the new numbering (below).  H is host-permuted to
// the new (register, lane) order (an extra section of the tables: FFT_TABLE_HSW).  fir_fft_kernel and the partitioned kernel (kernels_firpols.h); the 4096-point
// kernels keep the round-2 numbering.
// Same-box A/B, three alternations (profiles/r05_ab_e2swap.txt): 2^28 samples 0.8705 / 0.8701 / 0.8836 -> 0.8482 / 0.8453 / 0.8493 ms (-2.6 .. -3.9 %), 2^26 samples
// 0.2581 / 0.2586 / 0.2591 -> 0.2461 / 0.2521 / 0.2473 ms (-2.5 .. -4.6 %); outputs bit-identical to the LDS form on 2^24 samples (same arithmetic, other wires).
#ifndef LRHIP_FFT_E2_SWAP
#define LRHIP_FFT_E2_SWAP 1
#endif

namespace lrhip {

constexpr int FFTN = 1024;
constexpr int FFT_E1_ROW = 68;
constexpr int FFT_E2_ROW = 68;
// LRHIP_FFT_E2_SWAP row stride.  hipcc pairs neighbouring accesses into ds_read2_b64 / ds_read2st64_b64 / ds_write2_b64 (the ISA of this kernel has 43 paired
// reads, 15 paired writes and only 4 + 3 single ones), and a paired access is served in FOUR groups of 16 contiguous lanes with bank pair = (8-byte index)
// mod 16 (MI355X_MICROARCH.md, LDS table) - reads too, not only writes.  E1 forward: write element (k, lane) at k R + lane (consecutive lanes: fine for any R);
// read element (k1s, 4 i + sub) at k1s R + 4 i + sub - a 16-lane group is one ROW (one sub), k1s = 0..15, bank pair (R k1s) mod 16: R odd.  E1 inverse: the
// same two patterns with read and write exchanged.  R = 65 serves all four.  (The first cut used 66 for the forward transpose - derived for single
// ds_read_b64s, two groups of 32 lanes over 32 bank pairs - and the counters showed it: SQ_LDS_BANK_CONFLICT 4.19 M cycles per 2^26-sample launch, 12 % of
// the LDS-active cycles, where the round-4 layout had 0; profiles/r05_streaming_rows_counters.txt.)
#ifndef LRHIP_FFT_E1F_ROW
#define LRHIP_FFT_E1F_ROW 65      /* A/B: 66 = the first cut */
#endif
constexpr int FFT_E1F_ROW_SW = LRHIP_FFT_E1F_ROW;
constexpr int FFT_E1I_ROW_SW = 65;
constexpr int FFT_EX_ELEMS = LRHIP_FFT_SPLIT ? 16 * FFT_E2_ROW / 2 : 16 * FFT_E2_ROW;   // per-wave exchange buffer (float2 units)
constexpr int FFT_WPB = LRHIP_FFT_WPB;
constexpr int FFT_NB = LRHIP_FFT_NB;
constexpr int FFT_WAVES_PER_SIMD = (FFT_NB == 2 || LRHIP_FFT_ALLREG) ? 2 : FFT_WPB == 16 ? 4 : LRHIP_FFT_SPLIT ? 4 : 3;
// LDS map (float2 units): [FFT_WPB waves x FFT_NB x FFT_EX_ELEMS | tw1 16x64 | H 16x64 | tw2 64]
constexpr int FFT_LDS_TW1 = FFT_WPB * FFT_NB * FFT_EX_ELEMS;
constexpr int FFT_LDS_H = FFT_LDS_TW1 + 16 * 64;
constexpr int FFT_LDS_TW2 = FFT_LDS_H + 16 * 64;
constexpr int FFT_LDS_ELEMS = FFT_LDS_TW2 + 64;
// tw1 | Hperm | tw2, as uploaded by the host (one set per partition); LRHIP_FFT_E2_SWAP appends H in fir_fft_kernel's (register, lane) order - the partitioned
// kernel reads the same tables and keeps the round-2 order
constexpr int FFT_TABLE_HSW = 16 * 64 + 16 * 64 + 64;
constexpr int FFT_TABLE_ELEMS = 16 * 64 + 16 * 64 + 64 + (LRHIP_FFT_E2_SWAP ? 16 * 64 : 0);

// DIR = +1: forward (kernel e^{-j...}), -1: inverse.  In-place 4-point DFT, natural order out.
// A2J: a2 carries a pending factor W_16^(4*DIR) = -j*DIR (dft16's only trivial twiddle), folded into the first butterfly.
template <int DIR, bool A2J = false>
__host__ __device__ __forceinline__ void radix4(cf &a0, cf &a1, cf &a2, cf &a3)
{
    cf b0, b1;
    if (!A2J) { b0 = cadd(a0, a2); b1 = csub(a0, a2); }
    else if (DIR > 0) { b0 = sub_j(a0, a2); b1 = add_j(a0, a2); }
    else { b0 = add_j(a0, a2); b1 = sub_j(a0, a2); }
    cf b2 = cadd(a1, a3), d = csub(a1, a3);
    a0 = cadd(b0, b2);
    a2 = csub(b0, b2);
    a1 = DIR > 0 ? sub_j(b1, d) : add_j(b1, d);     // b1 + d * (-j) (forward) or d * (+j) (inverse)
    a3 = DIR > 0 ? add_j(b1, d) : sub_j(b1, d);
}

// multiply by W_16^(DIR*p), p in {1,2,3,6,9}
template <int DIR, int P>
__host__ __device__ __forceinline__ cf mul_w16(cf a)
{
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R = 0.70710678118654752f;
    constexpr float D = DIR > 0 ? -1.f : 1.f;          // sign of the imaginary part
    if constexpr (P == 1) return cmul_const(a, C1, D * S1);
    if constexpr (P == 2) return cmul_const(a, R, D * R);
    if constexpr (P == 3) return cmul_const(a, S1, D * C1);
    if constexpr (P == 6) return cmul_const(a, -R, D * R);
    if constexpr (P == 9) return cmul_const(a, -C1, -D * S1);
    return a;
}

// 16-point DFT of v[0..15] (index n = 4a + b), result in natural order: v[k] = sum_n v[n] W_16^(DIR*n*k)
template <int DIR>
__host__ __device__ __forceinline__ void dft16(cf (&v)[16])
{
    // radix-4 over a for every b: afterwards position 4c+b holds u[b][c]
#pragma unroll
    for (int b = 0; b < 4; b++) radix4<DIR>(v[b], v[4 + b], v[8 + b], v[12 + b]);
    // twiddle W_16^(b*c); W_16^4 of position 4*2+2 is folded into the radix-4 below
    v[4 * 1 + 1] = mul_w16<DIR, 1>(v[4 * 1 + 1]);
    v[4 * 1 + 2] = mul_w16<DIR, 2>(v[4 * 1 + 2]);
    v[4 * 1 + 3] = mul_w16<DIR, 3>(v[4 * 1 + 3]);
    v[4 * 2 + 1] = mul_w16<DIR, 2>(v[4 * 2 + 1]);
    v[4 * 2 + 3] = mul_w16<DIR, 6>(v[4 * 2 + 3]);
    v[4 * 3 + 1] = mul_w16<DIR, 3>(v[4 * 3 + 1]);
    v[4 * 3 + 2] = mul_w16<DIR, 6>(v[4 * 3 + 2]);
    v[4 * 3 + 3] = mul_w16<DIR, 9>(v[4 * 3 + 3]);
    // radix-4 over b for every c: position 4c+d holds X[c + 4d]
    radix4<DIR>(v[0], v[1], v[2], v[3]);
    radix4<DIR>(v[4], v[5], v[6], v[7]);
    radix4<DIR, true>(v[8], v[9], v[10], v[11]);
    radix4<DIR>(v[12], v[13], v[14], v[15]);
    // un-permute (compile-time renaming): natural[c + 4d] = v[4c + d]
    cf t;
    t = v[1]; v[1] = v[4]; v[4] = t;
    t = v[2]; v[2] = v[8]; v[8] = t;
    t = v[3]; v[3] = v[12]; v[12] = t;
    t = v[6]; v[6] = v[9]; v[9] = t;
    t = v[7]; v[7] = v[13]; v[13] = t;
    t = v[11]; v[11] = v[14]; v[14] = t;
}

// register <-> LDS transpose: lane writes v[k] to element widx(k), then reads element ridx(i) into v[i]
template <typename WI, typename RI>
__device__ __forceinline__ void exchange(cf *ex, cf (&v)[16], WI widx, RI ridx)
{
#if LRHIP_FFT_SPLIT
    float *exf = reinterpret_cast<float *>(ex);
    float re[16];
#pragma unroll
    for (int k = 0; k < 16; k++) exf[widx(k)] = v[k].x;
#pragma unroll
    for (int i = 0; i < 16; i++) re[i] = exf[ridx(i)];
#pragma unroll
    for (int k = 0; k < 16; k++) exf[widx(k)] = v[k].y;      // in-order DS queue: the re reads above are already issued
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = cf{re[i], exf[ridx(i)]};
#else
#pragma unroll
    for (int k = 0; k < 16; k++) ex[widx(k)] = v[k];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = ex[ridx(i)];
#endif
}

// 4 x 4 transpose register <-> row of 16 lanes: row b of p_s = row s of the old p_b (two swaps of 32-lane halves, two of 16-lane rows)
__device__ __forceinline__ void fft_swap32(float &a, float &b)
{
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r.x);
    b = __uint_as_float(r.y);
}
__device__ __forceinline__ void fft_swap16(float &a, float &b)
{
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r.x);
    b = __uint_as_float(r.y);
}
__device__ __forceinline__ void fft_transpose_rows(cf &p0, cf &p1, cf &p2, cf &p3)
{
    float r0 = p0.x, r1 = p1.x, r2 = p2.x, r3 = p3.x, i0 = p0.y, i1 = p1.y, i2 = p2.y, i3 = p3.y;
    fft_swap32(r0, r2); fft_swap32(i0, i2);
    fft_swap32(r1, r3); fft_swap32(i1, i3);
    fft_swap16(r0, r1); fft_swap16(i0, i1);
    fft_swap16(r2, r3); fft_swap16(i2, i3);
    p0 = cf{r0, i0}; p1 = cf{r1, i1}; p2 = cf{r2, i2}; p3 = cf{r3, i3};
}

// One overlap-save block per wave.  S = 2: ComplexFloat32 stream.  S = 1: Float32 stream with REAL taps, two
// consecutive blocks packed as re/im of one complex FFT (h real => IFFT(H*(Xa + jXb)) = h*xa + j h*xb).
// PRE = 1 (S = 1 only): fused FrequencyDiscriminatorBlock in front of the filter (frequencydiscriminator.lua:68-88): x is
// the ComplexFloat32 stream c, the filtered real stream is r[i] = arg(c[i] conj(c[i-1])) / gain with c[-1] = *disc_prev;
// `hist` then holds the last M-1 values of r.
template <int PRE>
__device__ __forceinline__ float fft_real_sample(const float *__restrict__ hist, const float *__restrict__ x, long p, int M, long n,
                                                 double inv_gain, const float2 *__restrict__ disc_prev)
{
    if (p < 0) return 0.f;
    if (p < M - 1) return hist[p];
    long xi = p - (M - 1);
    if (xi >= n) return 0.f;
    if (PRE == 0) return x[xi];
    const float2 *c = reinterpret_cast<const float2 *>(x);
    return discriminate(c[xi], xi ? c[xi - 1] : *disc_prev, inv_gain);
}

// history carry of the fused discriminator + FIR stage: last M-1 values of r, and the last complex sample
__global__ __launch_bounds__(256) void fir_fft_pre_history_kernel(const float *__restrict__ hist_in, const float *__restrict__ x,
                                                                  float *__restrict__ hist_out, int M, long n, double inv_gain,
                                                                  const float2 *__restrict__ prev_in, float2 *__restrict__ prev_out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M - 1) hist_out[i] = fft_real_sample<1>(hist_in, x, n + i, M, n, inv_gain, prev_in);
    if (i == 0 && n > 0) *prev_out = reinterpret_cast<const float2 *>(x)[n - 1];
}

// -DLRHIP_FFT_TRACE (a variant library, tools/fft_trace.sh): lane 0 of the waves of the first workgroups stamps the phases of its first blocks with clock64()
#ifdef LRHIP_FFT_TRACE
__device__ unsigned long long *lrhip_fft_trace;         // [block 8][wave FFT_WPB][iteration 32][12]
#define FFT_STAMP(i)                                                                                                                                \
    do {                                                                                                                                            \
        if (lrhip_fft_trace && blockIdx.x < 8 && trace_it < 32 && (threadIdx.x & 63) == 0)                                                          \
            lrhip_fft_trace[(((size_t)blockIdx.x * FFT_WPB + (threadIdx.x >> 6)) * 32 + trace_it) * 12 + (i)] = clock64();                            \
    } while (0)
#else
#define FFT_STAMP(i) do { } while (0)
#endif

template <int S, int PRE>
__global__ __launch_bounds__(64 * FFT_WPB, FFT_WAVES_PER_SIMD) void fir_fft_kernel(const float *__restrict__ hist, const float *__restrict__ x,
                                                          const float2 *__restrict__ tables, float *__restrict__ y,
                                                          int M, long n, long n_out, long nblocks,
                                                          double inv_gain, const float2 *__restrict__ disc_prev, float *__restrict__ hist_out,
                                                          int Mh, long delay, int accumulate, int rounds, int n_full, int taper)
{
    // Long filters are PARTITIONED: this launch applies taps [delay, delay + M) of an Mh-tap filter - the M-tap overlap-save
    // on the stream delayed by `delay` samples (history = Mh - 1 samples) - and adds to y when accumulate != 0.
    // A plain filter is the one-partition case Mh = M, delay = 0.
    extern __shared__ __attribute__((aligned(16))) float2 fl[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: block addresses and bounds stay on the SALU
    // history carry (PRE = 0; saves the fir_history_kernel launch): last M-1 stream samples into the other ping-pong buffer
    if (PRE == 0 && hist_out && blockIdx.x == 0)
        for (int i = tid; i < (Mh - 1) * S; i += 64 * FFT_WPB) hist_out[i] = stream_at<S>(hist, x, n + i / S, i % S, Mh, n);
    cf *flc = reinterpret_cast<cf *>(fl);
    cf *ex0 = flc + wave * FFT_NB * FFT_EX_ELEMS;
    const cf *tw1 = flc + FFT_LDS_TW1, *Hp = flc + FFT_LDS_H, *tw2 = flc + FFT_LDS_TW2;

    // block advance: the overlap is rounded up to a multiple of 64 samples (V >= M-1) so that every block's load window
    // AND its stored rows start on 512-B boundaries relative to x / y (L = 897 would misalign every row)
    const int V = ((M - 1 + 63) / 64) * 64;
    const int L = FFTN - V;
#if LRHIP_FFT_E2_SWAP
    const int sub = lane >> 4, k1s = lane & 15;      // stages 2 and 3: sub = t2 or q = the lane's ROW of 16, k1s = k1 = its position in the row
#else
    const int sub = lane & 3, k1s = lane >> 2;       // stages 2 and 3: sub = t2 or q, k1s = k1
#endif
    constexpr int BPW = S == 2 ? 1 : 2;               // stream blocks per FFT

    // Block order.  rounds == 0: persistent, workgroup g walks batches g, g + gridDim.x, ...  rounds > 0: one-shot, workgroup g
    // owns the `rounds` consecutive batches starting at g * rounds (a batch = FFT_WPB transforms, one per wave) and the
    // dispatcher hands workgroups out in address order - the shape that reaches 6.3 TB/s in tools/mb_stream.hip where the
    // persistent stride reaches 4.4-5.4.
    // LRHIP_FFT_XCD_MAP: workgroup g (observed on XCD g % 8) takes slot (g % 8) * gridDim.x / 8 + g / 8, so that neighbours in the stream share an L2
    // (measured, same box, 2^28 samples: 0.840 against 0.835 ms for the plain address order - eight compact windows instead of one; off)
#ifndef LRHIP_FFT_XCD_MAP
#define LRHIP_FFT_XCD_MAP 0
#endif
    const long bid = (LRHIP_FFT_XCD_MAP && (gridDim.x & 7) == 0) ? (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : (long)blockIdx.x;
    // One-shot order with a TAPERED tail (taper > 0): the first n_full workgroups own `rounds` batches each, then `taper` workgroups (one per resident
    // slot) own rounds / 2 each, the next `taper` rounds / 4, the rest rounds / 8 (at least one) - the dispatcher hands workgroups out in order, so the
    // launch ends with short workgroups instead of a last wave of long ones finishing at different times (a 2^28-sample launch is 12 waves of 69 us
    // workgroups: its fixed cost of ~60 us, read off the size sweep 2^24 .. 2^28, is mostly that tail).
    long b0 = bid * rounds, bn = rounds;
    if (rounds > 0 && taper > 0 && bid >= n_full) {
        const long r1 = rounds / 2 > 0 ? rounds / 2 : 1, r2 = rounds / 4 > 0 ? rounds / 4 : 1, r3 = rounds / 8 > 0 ? rounds / 8 : 1;
        const long k = bid - n_full, base = (long)n_full * rounds;
        if (k < taper) { bn = r1; b0 = base + k * r1; }
        else if (k < 2L * taper) { bn = r2; b0 = base + taper * r1 + (k - taper) * r2; }
        else { bn = r3; b0 = base + taper * (r1 + r2) + (k - 2L * taper) * r3; }
    }
    // LRHIP_FFT_CARRY (A/B, round 3): in the one-shot order a wave takes `bn` ADJACENT blocks instead of every FFT_WPB-th one, and keeps the last V = 128
    // samples of a block's window (two of its sixteen rows, raw) in registers as the first two rows of the next: 14 loads per block instead of 16 - the
    // 12.5 % overlap is then not even re-read from L2
#ifndef LRHIP_FFT_CARRY
#define LRHIP_FFT_CARRY 0
#endif
    const bool wave_major = LRHIP_FFT_CARRY && S == 2 && PRE == 0 && FFT_NB == 1 && rounds > 0 && V == 128 && delay == 0;
    const long fstep = wave_major ? 1 : rounds > 0 ? FFT_WPB : (long)gridDim.x * FFT_WPB;
    const long ffirst = wave_major ? b0 * FFT_WPB + (long)wave * bn : (rounds > 0 ? b0 : bid) * FFT_WPB + wave;
    const long fend = wave_major ? ffirst + bn : rounds > 0 ? (b0 + bn) * FFT_WPB : (nblocks + 1);
    cf keep0 = cf{0.f, 0.f}, keep1 = cf{0.f, 0.f};
    bool kept = false;
#if LRHIP_FFT_PREFETCH
    cf pre[16];
    bool have = false;
    auto prefetch = [&](long b) {
        long lo = b * L - V;
        have = S == 2 && b < nblocks && lo >= 0 && lo + FFTN <= n;
        if (have) {
            const cf *src = reinterpret_cast<const cf *>(x) + lo + lane;
#pragma unroll
            for (int i = 0; i < 16; i++) pre[i] = src[64 * i];
        }
    };
    prefetch(ffirst);
    // LRHIP_FFT_PREFETCH == 3 (round 6, A/B): the next block's sixteen loads go out FOUR AT A TIME between the phases of this block instead of as one burst - a
    // burst of sixteen wave-wide loads stalls in issue against the memory system's back-pressure (31 % of a block's clocks in the phase trace) and the wave, in
    // order, issues nothing else meanwhile.  Measured 1.8 % SLOWER than no prefetch (0.826 against 0.812 ms, three alternations, profiles/r06_ab_fft_prefetch3.txt)
    [[maybe_unused]] const cf *pf_src = nullptr;
    [[maybe_unused]] auto prefetch_begin = [&](long b) {
        const long lo = b * L - V;
        have = S == 2 && b < nblocks && lo >= 0 && lo + FFTN <= n;
        pf_src = reinterpret_cast<const cf *>(x) + (have ? lo : 0);
    };
    [[maybe_unused]] auto prefetch_part = [&](int q) {
        if (have) {
#pragma unroll
            for (int i = 4 * q; i < 4 * q + 4; i++) pre[i] = (pf_src + 64 * i)[(unsigned)lane];
        }
        __builtin_amdgcn_sched_barrier(0);
    };
#endif
    // NB blocks per iteration (LRHIP_FFT_NB): block b of an iteration is transform fbase + b * fstep
    auto load_block = [&](long fb, cf (&v)[16], [[maybe_unused]] int b) {
        // ---- load: window position 64*i + lane  (stream = [M-1 history | chunk])
        // (measured and dropped: 16-B accesses through an LDS transpose - no gain)
        if (S == 2) {
            const long xlo = fb * L - V - delay;          // x index of window position 0
            const long p0 = xlo + (Mh - 1);               // the same in stream coordinates
#if LRHIP_FFT_PREFETCH
            if (have && b == 0) {
#pragma unroll
                for (int i = 0; i < 16; i++) v[i] = pre[i];
            } else
#endif
            if (xlo >= 0 && xlo + FFTN <= n) {
                const cf *src = reinterpret_cast<const cf *>(x) + xlo + lane;
                [[maybe_unused]] const cf *srcu = reinterpret_cast<const cf *>(x) + xlo;
                if (wave_major && kept) {
                    v[0] = keep0; v[1] = keep1;
#pragma unroll
                    for (int i = 2; i < 16; i++) v[i] = src[64 * i];
                } else {
#pragma unroll
#if LRHIP_FFT_NT >= 2
                    for (int i = 0; i < 16; i++) v[i] = __builtin_nontemporal_load((srcu + 64 * i) + (unsigned)lane);
#else
                    for (int i = 0; i < 16; i++) v[i] = (srcu + 64 * i)[(unsigned)lane];      // wave-uniform row pointer + the lane's 32-bit index
#endif
                }
                if (wave_major) { keep0 = v[14]; keep1 = v[15]; kept = true; }
            } else {
                kept = false;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    long p = p0 + 64 * i + lane;
                    v[i] = cf{stream_at<2>(hist, x, p, 0, Mh, n), stream_at<2>(hist, x, p, 1, Mh, n)};
                }
            }
        } else {
            const long pa = (fb * 2) * L - V - delay + (Mh - 1), pb = pa + L;      // stream positions of the two packed blocks
            const long xa = pa - (Mh - 1), xb = pb - (Mh - 1);                     // their x indices
            if (xa >= PRE && xb + FFTN <= n) {
                // both windows inside the chunk: coalesced loads, no history
                if (PRE == 0) {
                    const float *sa = x + xa + lane, *sb = x + xb + lane;
#pragma unroll
                    for (int i = 0; i < 16; i++) v[i] = cf{sa[64 * i], sb[64 * i]};
                } else {
                    const float2 *ca = reinterpret_cast<const float2 *>(x) + xa + lane, *cb = reinterpret_cast<const float2 *>(x) + xb + lane;
#pragma unroll
                    for (int i = 0; i < 16; i++)
                        v[i] = cf{discriminate(ca[64 * i], ca[64 * i - 1], inv_gain), discriminate(cb[64 * i], cb[64 * i - 1], inv_gain)};
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; i++)
                    v[i] = cf{fft_real_sample<PRE>(hist, x, pa + 64 * i + lane, Mh, n, inv_gain, disc_prev),
                              fft_real_sample<PRE>(hist, x, pb + 64 * i + lane, Mh, n, inv_gain, disc_prev)};
            }
        }

    };
    auto store_block = [&](long fb, cf (&v)[16]) {
        // ---- store: window positions V .. N-1 are this block's L outputs (positions < M-1 are the circular wrap,
        // firfilter.lua:379 copies output_block[M-1 ..]; we drop up to 63 more so the rows stay aligned)
        if (S == 2) {
            const long o0 = fb * L - V;
            cf *dst = reinterpret_cast<cf *>(y) + o0 + lane;
            [[maybe_unused]] cf *dstu = reinterpret_cast<cf *>(y) + o0;
            if (LRHIP_FFT_STRAIGHT_STORES && LRHIP_FFT_NT >= 1 && o0 + FFTN <= n_out && V == 128 && !accumulate) {
                // the common case - 128 taps or fewer, no accumulation - as fourteen stores in a row off one base address.  (Round 4, read off the ISA: with the
                // overlap and the accumulate flag tested per row every store sat in a basic block of its own behind two scalar branches and a recomputed
                // 64-bit address: ~10 instructions per row, 4 of them vector.)  Behind bench.py's 250 ms clock ramp the two forms measure the same at 2^26 / 2^27 / 2^28
                // samples (0.2056 / 0.4156 / 0.8414 against 0.2085 / 0.4062 / 0.8439 ms); the 5-7 % this form gains in a cold 20-launch loop is the clock ramp
#pragma unroll
                for (int i = 2; i < 16; i++) __builtin_nontemporal_store(v[i], (dstu + 64 * i) + (unsigned)lane);
            } else if (o0 + FFTN <= n_out) {
#pragma unroll
                for (int i = 0; i < 16; i++)
                    if (64 * i >= V) {                                                                // wave-uniform: whole rows only
#if LRHIP_FFT_NT >= 1
                        if (!accumulate) __builtin_nontemporal_store(v[i], (dstu + 64 * i) + (unsigned)lane);
                        else dst[64 * i] = dst[64 * i] + v[i];
#else
                        dst[64 * i] = accumulate ? dst[64 * i] + v[i] : v[i];
#endif
                    }
            } else {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    int nn = 64 * i + lane;
                    if (nn >= V && o0 + nn < n_out) dst[64 * i] = accumulate ? dst[64 * i] + v[i] : v[i];
                }
            }
        } else {
            const long oa = (fb * 2) * L - V, ob = oa + L;
            if (LRHIP_FFT_STRAIGHT_STORES && LRHIP_FFT_NT >= 1 && ob + FFTN <= n_out && V == 128 && !accumulate) {
                float *da = y + oa + lane, *db = y + ob + lane;
#pragma unroll
                for (int i = 2; i < 16; i++) {
                    __builtin_nontemporal_store(v[i].x, da + 64 * i);
                    __builtin_nontemporal_store(v[i].y, db + 64 * i);
                }
            } else if (ob + FFTN <= n_out) {
                float *da = y + oa + lane, *db = y + ob + lane;
#pragma unroll
                for (int i = 0; i < 16; i++)
                    if (64 * i >= V) {
#if LRHIP_FFT_NT >= 1
                        if (!accumulate) {
                            __builtin_nontemporal_store(v[i].x, da + 64 * i);
                            __builtin_nontemporal_store(v[i].y, db + 64 * i);
                        } else
#endif
                        {
                            da[64 * i] = accumulate ? da[64 * i] + v[i].x : v[i].x;
                            db[64 * i] = accumulate ? db[64 * i] + v[i].y : v[i].y;
                        }
                    }
            } else {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    int nn = 64 * i + lane;
                    if (nn >= V) {
                        if (oa + nn < n_out) y[oa + nn] = accumulate ? y[oa + nn] + v[i].x : v[i].x;
                        if (ob + nn < n_out) y[ob + nn] = accumulate ? y[ob + nn] + v[i].y : v[i].y;
                    }
                }
            }
        }
    };
    // Start-up: the one-shot block order pays it once per WORKGROUP, and a 2^28-sample launch is 12 rounds of workgroups per resident slot (the ~45 us
    // per launch that do not scale with n, profiles/r03_fir_fft_ab_counters.txt): the first block's 16 loads are issued BEFORE the 17 KB of tables are
    // staged, so the two latencies overlap instead of adding (LRHIP_FFT_EARLY = 0: tables first, as in round 2)
#ifndef LRHIP_FFT_EARLY
#define LRHIP_FFT_EARLY 1
#endif
    cf v_first[16];
    const bool early = LRHIP_FFT_EARLY && FFT_NB == 1 && !LRHIP_FFT_PREFETCH && ffirst * BPW < nblocks && ffirst < fend;
    if (early) load_block(ffirst, v_first, 0);
#if LRHIP_FFT_E2_SWAP
    for (int i = tid; i < 16 * 64 + 16 * 64 + 64; i += 64 * FFT_WPB) fl[FFT_LDS_TW1 + i] = tables[(i >= 16 * 64 && i < 2 * 16 * 64) ? i - 16 * 64 + FFT_TABLE_HSW : i];
#else
    for (int i = tid; i < FFT_TABLE_ELEMS; i += 64 * FFT_WPB) fl[FFT_LDS_TW1 + i] = tables[i];
#endif
    __syncthreads();
#if LRHIP_FFT_TW_REG
    // (the Float32-stream instantiation sits at the 168-register cap and spills six dwords per lane with all fifteen twiddles in registers; keeping only the first
    // 12 or 8 of them - LRHIP_FFT_TW_REG_F32, no spill - measured EQUAL, 0.1188 / 0.1188 / 0.123 ms on 2^26 samples: the spill is not in the block loop)
#ifndef LRHIP_FFT_TW_REG_F32
#define LRHIP_FFT_TW_REG_F32 16
#endif
    constexpr int TWR = S == 1 ? LRHIP_FFT_TW_REG_F32 : 16;
    cf tw1r[16];
#pragma unroll
    for (int k = 1; k < TWR; k++) tw1r[k] = tw1[k * 64 + lane];
#define FFT_TW1(k) ((k) < TWR ? tw1r[(k) < TWR ? (k) : 0] : tw1[(k) * 64 + lane])
#if LRHIP_FFT_ALLREG
    cf tw2f[16], tw2i[16], Hr[16];
#pragma unroll
    for (int k = 1; k < 16; k++) tw2f[k] = tw2[k * 4 + (lane & 3)];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int t2 = 1; t2 < 4; t2++) tw2i[4 * j + t2] = tw2[(4 * j + (lane & 3)) * 4 + t2];
#pragma unroll
    for (int r = 0; r < 16; r++) Hr[r] = Hp[r * 64 + lane];
#endif
#else
#define FFT_TW1(k) tw1[(k) * 64 + lane]
#endif

    [[maybe_unused]] int trace_it = 0;
    for (long fbase = ffirst; fbase * BPW < nblocks && fbase < fend; fbase += FFT_NB * fstep) {
        FFT_STAMP(0);
        cf v[FFT_NB][16];
        bool live[FFT_NB];
#pragma unroll
        for (int b = 0; b < FFT_NB; b++) {
            const long fb = fbase + b * fstep;
            live[b] = fb * BPW < nblocks && fb < fend;      // wave-uniform
            if (live[b] && early && b == 0 && fbase == ffirst) {
#pragma unroll
                for (int i = 0; i < 16; i++) v[b][i] = v_first[i];
            } else if (live[b]) {
#if LRHIP_FFT_SETPRIO
                __builtin_amdgcn_s_setprio(LRHIP_FFT_SETPRIO & 3);        // (A/B, round 6) a wave in its memory phase issues ahead of the waves that are computing
#endif
                load_block(fb, v[b], b);
#if LRHIP_FFT_SETPRIO
                if (!(LRHIP_FFT_SETPRIO & 4)) __builtin_amdgcn_s_setprio(0);
#endif
            }
            else {
#pragma unroll
                for (int i = 0; i < 16; i++) v[b][i] = cf{0.f, 0.f};
            }
        }
        FFT_STAMP(1);
        // ---- forward stage 1: radix-16 over n1, twiddle W_1024^(t*k1)
#pragma unroll
        for (int b = 0; b < FFT_NB; b++) {
            dft16<1>(v[b]);
#if LRHIP_FFT_PREFETCH == 1
            if (S == 2 && b == 0) prefetch(fbase + FFT_NB * fstep);
#endif
#if LRHIP_FFT_PREFETCH == 3
            if (S == 2 && b == 0) { prefetch_begin(fbase + FFT_NB * fstep < fend ? fbase + FFT_NB * fstep : nblocks); prefetch_part(0); }
#endif
#pragma unroll
            for (int k = 1; k < 16; k++) v[b][k] = cmul(v[b][k], FFT_TW1(k));
        }
        FFT_STAMP(2);
        // E1: write (k1, t), read (k1 = k1s, 4*t1 + t2), t2 = sub
#pragma unroll
        for (int b = 0; b < FFT_NB; b++)
#if LRHIP_FFT_E2_SWAP
            exchange(ex0 + b * FFT_EX_ELEMS, v[b], [&](int k) { return k * FFT_E1F_ROW_SW + lane; }, [&](int i) { return k1s * FFT_E1F_ROW_SW + 4 * i + sub; });
#else
            exchange(ex0 + b * FFT_EX_ELEMS, v[b], [&](int k) { return k * FFT_E1_ROW + lane; }, [&](int i) { return k1s * FFT_E1_ROW + 4 * i + sub; });
#endif
        FFT_STAMP(3);
#if LRHIP_FFT_PREFETCH == 3
        if (S == 2) prefetch_part(1);
#endif
        // ---- forward stage 2: radix-16 over t1, twiddle W_64^(t2*k2)
#pragma unroll
        for (int b = 0; b < FFT_NB; b++) {
            dft16<1>(v[b]);
#pragma unroll
#if LRHIP_FFT_ALLREG
            for (int k = 1; k < 16; k++) v[b][k] = cmul(v[b][k], tw2f[k]);
#else
            for (int k = 1; k < 16; k++) v[b][k] = cmul(v[b][k], tw2[k * 4 + sub]);
#endif
        }
        // E2: write (k1 = k1s, k2, t2 = sub), read (k1 = k1s, k2 = 4j + q, t2 = 0..3), q = sub; register 4j + t2
#pragma unroll
        for (int b = 0; b < FFT_NB; b++)
#if LRHIP_FFT_E2_SWAP
#pragma unroll
            for (int j = 0; j < 4; j++) fft_transpose_rows(v[b][4 * j], v[b][4 * j + 1], v[b][4 * j + 2], v[b][4 * j + 3]);     // row q, register 4j + t2 = old row t2, register 4j + q
#else
            exchange(ex0 + b * FFT_EX_ELEMS, v[b], [&](int k) { return k1s * FFT_E2_ROW + 17 * sub + k; },
                     [&](int r) { return k1s * FFT_E2_ROW + 17 * (r & 3) + (r & 12) + sub; });       // r = 4j + t2
#endif
        FFT_STAMP(4);
#if LRHIP_FFT_PREFETCH == 3
        if (S == 2) prefetch_part(2);
#endif
        // ---- forward stage 3: radix-4 over t2 -> k3; multiply by H; inverse stage 3: radix-4 over k3 -> t2
#pragma unroll
        for (int b = 0; b < FFT_NB; b++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                radix4<1>(v[b][4 * j], v[b][4 * j + 1], v[b][4 * j + 2], v[b][4 * j + 3]);
#pragma unroll
#if LRHIP_FFT_ALLREG
                for (int k3 = 0; k3 < 4; k3++) v[b][4 * j + k3] = cmul(v[b][4 * j + k3], Hr[4 * j + k3]);
#else
                for (int k3 = 0; k3 < 4; k3++) v[b][4 * j + k3] = cmul(v[b][4 * j + k3], Hp[(4 * j + k3) * 64 + lane]);
#endif
                radix4<-1>(v[b][4 * j], v[b][4 * j + 1], v[b][4 * j + 2], v[b][4 * j + 3]);
                // conj twiddle W_64^(-t2*k2), k2 = 4j + q
#pragma unroll
#if LRHIP_FFT_ALLREG
                for (int t2 = 1; t2 < 4; t2++) v[b][4 * j + t2] = cmulc(v[b][4 * j + t2], tw2i[4 * j + t2]);
#else
                for (int t2 = 1; t2 < 4; t2++) v[b][4 * j + t2] = cmulc(v[b][4 * j + t2], tw2[(4 * j + sub) * 4 + t2]);
#endif
            }
        }
        // E2 back: write (k1, k2 = 4j + q, t2), read (k1 = k1s, k2 = 0..15, t2 = sub)
#pragma unroll
        for (int b = 0; b < FFT_NB; b++)
#if LRHIP_FFT_E2_SWAP
#pragma unroll
            for (int j = 0; j < 4; j++) fft_transpose_rows(v[b][4 * j], v[b][4 * j + 1], v[b][4 * j + 2], v[b][4 * j + 3]);     // the transpose is its own inverse
#else
            exchange(ex0 + b * FFT_EX_ELEMS, v[b], [&](int r) { return k1s * FFT_E2_ROW + 17 * (r & 3) + (r & 12) + sub; },
                     [&](int k) { return k1s * FFT_E2_ROW + 17 * sub + k; });
#endif
        FFT_STAMP(5);
#if LRHIP_FFT_PREFETCH == 3
        if (S == 2) prefetch_part(3);
#endif
        // ---- inverse stage 2: radix-16 over k2 -> t1
#pragma unroll
        for (int b = 0; b < FFT_NB; b++) dft16<-1>(v[b]);
        FFT_STAMP(6);
        // E1 back: write (k1 = k1s, 4*t1 + t2), read (k1, t = lane)
#pragma unroll
        for (int b = 0; b < FFT_NB; b++)
#if LRHIP_FFT_E2_SWAP
            exchange(ex0 + b * FFT_EX_ELEMS, v[b], [&](int i) { return k1s * FFT_E1I_ROW_SW + 4 * i + sub; }, [&](int k) { return k * FFT_E1I_ROW_SW + lane; });
#else
            exchange(ex0 + b * FFT_EX_ELEMS, v[b], [&](int i) { return k1s * FFT_E1_ROW + 4 * i + sub; }, [&](int k) { return k * FFT_E1_ROW + lane; });
#endif
        FFT_STAMP(7);
        // ---- inverse stage 1: conj twiddle, radix-16 over k1 -> n1
#pragma unroll
        for (int b = 0; b < FFT_NB; b++) {
#pragma unroll
            for (int k = 1; k < 16; k++) v[b][k] = cmulc(v[b][k], FFT_TW1(k));
            dft16<-1>(v[b]);
        }
        FFT_STAMP(8);
#if LRHIP_FFT_PREFETCH == 2
        // LATE prefetch (round 5, A/B): the next block's sixteen loads go out in front of this block's stores - the only point of the loop where nothing but
        // the sixteen outputs is live, so the 32 registers cost no occupancy - and land while the stores are issued
        if (S == 2) prefetch(fbase + FFT_NB * fstep < fend ? fbase + FFT_NB * fstep : nblocks);
#endif
#pragma unroll
        for (int b = 0; b < FFT_NB; b++)
            if (live[b]) {
#if LRHIP_FFT_SETPRIO
                __builtin_amdgcn_s_setprio(LRHIP_FFT_SETPRIO & 3);
#endif
                store_block(fbase + b * fstep, v[b]);
#if LRHIP_FFT_SETPRIO
                if (!(LRHIP_FFT_SETPRIO & 4)) __builtin_amdgcn_s_setprio(0);
#endif
            }
        FFT_STAMP(9);
#ifdef LRHIP_FFT_TRACE
        trace_it++;
#endif
    }
#undef FFT_TW1
}

// ------------------------------------------------------------------------------------------------------------
// spectrum_utils.DFT / IDFT / PSD for N = 1024 frames on the same one-wave-per-frame engine
// (radio/utilities/spectrum_utils.lua:25-113, :259-349, :522-640; fftshift :654-667 folded into the store index).
// Forward: natural-order load (x window) -> 3 stages -> lane (k1 = lane>>2, q = lane&3), register 4j + k3 holds
// X[k1 + 16*(4j + q) + 256*k3]; for one register the 64 lanes cover 64 consecutive bins (permuted), so the
// scattered store is still one full segment per instruction.  Inverse: gather-load in that layout, mirror stages,
// natural-order store.  LDS: [4 waves x exchange | tw1 16x64 | tw2 64] float2.
// ------------------------------------------------------------------------------------------------------------
#ifndef LRHIP_PSD_SCALAR_STORES
#define LRHIP_PSD_SCALAR_STORES 0      /* 1 = the round-2 stores, 4 bytes per lane (A/B) */
#endif
constexpr int SPEC_LDS_TW1 = 4 * FFT_EX_ELEMS;
constexpr int SPEC_LDS_TW2 = SPEC_LDS_TW1 + 16 * 64;
constexpr int SPEC_LDS_ELEMS = SPEC_LDS_TW2 + 64;
constexpr int SPEC_TABLE_ELEMS = 16 * 64 + 64;        // tw1 | tw2 as uploaded by the host

enum { SPEC_FWD_COMPLEX = 0, SPEC_FWD_PSD = 1, SPEC_FWD_PSD_LOG = 2, SPEC_INV_COMPLEX = 3, SPEC_INV_REAL = 4 };

template <bool IN_REAL>
__global__ __launch_bounds__(256, 3) void spectrum1024_kernel(const float *__restrict__ x, float *__restrict__ y, long nframes,
                                                               const float2 *__restrict__ tables, const float *__restrict__ window,
                                                               int mode, float out_scale, int shift, int rounds)
{
    extern __shared__ __attribute__((aligned(16))) float2 fl[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    cf *flc = reinterpret_cast<cf *>(fl);
    cf *ex = flc + wave * FFT_EX_ELEMS;
    const cf *tw1 = flc + SPEC_LDS_TW1, *tw2 = flc + SPEC_LDS_TW2;
    for (int i = tid; i < SPEC_TABLE_ELEMS; i += 256) fl[SPEC_LDS_TW1 + i] = tables[i];
    __syncthreads();
    const int sub = lane & 3, k1s = lane >> 2;
    // (round 4, measured and dropped: the lane's twiddles in registers as in fir_fft_kernel - this kernel is at its 168 registers already: with both tables 0.295
    // against 0.154 ms, with the first one alone 20 registers still spill)

    // rounds > 0: one-shot order - workgroup g owns the `rounds` consecutive batches of four frames from g * rounds on and the dispatcher hands workgroups out in
    // address order (what took the overlap-save filter kernel from 263-314 to 316 GS/s); 0: persistent stride
    const long fstep = rounds > 0 ? 4 : (long)gridDim.x * 4;
    const long ffirst = (rounds > 0 ? (long)blockIdx.x * rounds : (long)blockIdx.x) * 4 + wave;
    const long fend = rounds > 0 ? ((long)blockIdx.x + 1) * rounds * 4 : nframes;
    for (long f = ffirst; f < nframes && f < fend; f += fstep) {
        cf v[16];
        if (mode <= SPEC_FWD_PSD_LOG) {
            // ---- forward
            if (IN_REAL) {
                const float *src = x + f * FFTN + lane;
#pragma unroll
                for (int i = 0; i < 16; i++) v[i] = cf{src[64 * i], 0.f};
            } else {
                const cf *src = reinterpret_cast<const cf *>(x) + f * FFTN + lane;
#pragma unroll
                for (int i = 0; i < 16; i++) v[i] = src[64 * i];
            }
            if (window) {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    float w = window[64 * i + lane];
                    v[i].x *= w;
                    v[i].y *= w;
                }
            }
            dft16<1>(v);
#pragma unroll
            for (int k = 1; k < 16; k++) v[k] = cmul(v[k], tw1[k * 64 + lane]);
            exchange(ex, v, [&](int k) { return k * FFT_E1_ROW + lane; }, [&](int i) { return k1s * FFT_E1_ROW + 4 * i + sub; });
            dft16<1>(v);
#pragma unroll
            for (int k = 1; k < 16; k++) v[k] = cmul(v[k], tw2[k * 4 + sub]);
            exchange(ex, v, [&](int k) { return k1s * FFT_E2_ROW + 17 * sub + k; },
                     [&](int r) { return k1s * FFT_E2_ROW + 17 * (r & 3) + (r & 12) + sub; });
#pragma unroll
            for (int j = 0; j < 4; j++) radix4<1>(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
            // ---- store bin k = k1 + 16*(4j + q) + 256*k3 at position (k + N/2) mod N when shifting
            if (mode != SPEC_FWD_COMPLEX && ((reinterpret_cast<uintptr_t>(y) & 15) == 0) && !LRHIP_PSD_SCALAR_STORES) {
                // power spectra leave as 16-byte non-temporal stores (round 3): the 1024 values of the frame go through the wave's exchange buffer in
                // natural order - 16 four-byte LDS writes, 4 sixteen-byte reads - instead of 16 stores of 4 bytes per lane
                float *exf = reinterpret_cast<float *>(ex);
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int k3 = 0; k3 < 4; k3++) {
                        const int k = k1s + 16 * (4 * j + sub) + 256 * k3;
                        const int pos = shift ? ((k + FFTN / 2) & (FFTN - 1)) : k;
                        const cf X = v[4 * j + k3];
                        const float p = fmaf(X.x, X.x, X.y * X.y) * out_scale;       // spectrum_utils.lua:631-638
                        exf[pos] = mode == SPEC_FWD_PSD_LOG ? psd_db(p) : p;
                    }
                float4 *dst = reinterpret_cast<float4 *>(y + f * FFTN);
#pragma unroll
                for (int i = 0; i < 4; i++) nt_store(dst + 64 * i + lane, reinterpret_cast<const float4 *>(exf)[64 * i + lane]);
                continue;
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int k3 = 0; k3 < 4; k3++) {
                    int k = k1s + 16 * (4 * j + sub) + 256 * k3;
                    int pos = shift ? ((k + FFTN / 2) & (FFTN - 1)) : k;
                    cf X = v[4 * j + k3];
                    if (mode == SPEC_FWD_COMPLEX) {
                        reinterpret_cast<cf *>(y)[f * FFTN + pos] = X * cf{out_scale, out_scale};
                    } else {
                        float p = fmaf(X.x, X.x, X.y * X.y) * out_scale;       // spectrum_utils.lua:631-638
                        y[f * FFTN + pos] = mode == SPEC_FWD_PSD_LOG ? psd_db(p) : p;
                    }
                }
        } else {
            // ---- inverse: gather the spectrum in the stage-3 layout, mirror the stages
            const cf *src = reinterpret_cast<const cf *>(x) + f * FFTN;
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int k3 = 0; k3 < 4; k3++) v[4 * j + k3] = src[k1s + 16 * (4 * j + sub) + 256 * k3];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                radix4<-1>(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
#pragma unroll
                for (int t2 = 1; t2 < 4; t2++) v[4 * j + t2] = cmulc(v[4 * j + t2], tw2[(4 * j + sub) * 4 + t2]);
            }
            exchange(ex, v, [&](int r) { return k1s * FFT_E2_ROW + 17 * (r & 3) + (r & 12) + sub; },
                     [&](int k) { return k1s * FFT_E2_ROW + 17 * sub + k; });
            dft16<-1>(v);
            exchange(ex, v, [&](int i) { return k1s * FFT_E1_ROW + 4 * i + sub; }, [&](int k) { return k * FFT_E1_ROW + lane; });
#pragma unroll
            for (int k = 1; k < 16; k++) v[k] = cmulc(v[k], tw1[k * 64 + lane]);
            dft16<-1>(v);
            if (mode == SPEC_INV_COMPLEX) {
                cf *dst = reinterpret_cast<cf *>(y) + f * FFTN + lane;
#pragma unroll
                for (int i = 0; i < 16; i++) dst[64 * i] = v[i] * cf{out_scale, out_scale};
            } else {
                float *dst = y + f * FFTN + lane;
#pragma unroll
                for (int i = 0; i < 16; i++) dst[64 * i] = v[i].x * out_scale;    // real part (spectrum_utils.lua:499-503)
            }
        }
    }
}

}  // namespace lrhip
