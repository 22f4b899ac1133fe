// kernels_firdecim.h - round 5: second form of the LDS-staged decimating FIR (fir_decim_lds_kernel of kernels_fir.h stays for Float32 streams and ComplexFloat32
// taps) for a ComplexFloat32 stream - or the raw records of an IQ file - and real taps: TunerBlock(offset, bw, 50) of radio/composites/tuner.lua:34-44 as
// rtlsdr_am_envelope.lua / rtlsdr_ssb.lua / rtlsdr_nbfm.lua use it, TunerBlock(.., 80) of rtlsdr_pocsag.lua / rtlsdr_ax25.lua (a multiple of four: the
// phase-array layout below), DecimatorBlock(25), ...
//
// Same result as the first form, bit for bit, wherever the taps of an output stay one chain (one fmaf chain per output in the reference's tap order; the
// rotator's phasor is the library-wide P(n & ~7) W[n & 7] of kernels_elem.h).  What changed is where a tile's ~19 000 clocks went (clock64 stamps of lane 0
// of every wave, tools/ab_decim.hip -DLRHIP_DECIM_TRACE; decimation 50, 128 taps, rotator):
//   filter 8 200   two of the four waves held all 121 outputs of the tile, 64 clocks per tap: five address instructions per window read (the padded
//                  layout) and two scalar fmaf per tap.  Now: an unpadded window (a lane's sixteen reads are ONE address + immediate offsets; a thread
//                  stride of D complex samples is conflict-free for odd D and two-way for D = 2 mod 4), one v_pk_fma_f32 per tap on (re, im), and tiles of
//                  <= 128 outputs spread over all four waves (32 outputs per wave).  With the rotator (whose output is compared with the oracle to a
//                  tolerance anyway) the upper half-wave takes the second half of the taps of the same 32 outputs and the halves meet in one cross-half
//                  add: 64 taps per lane (0.117 -> 0.106 ms) - not in an exact chain (`one_chain`), which keeps one chain per output and with it the bits;
//   stage 6-8 000  a partial first / last block of eight sent ONE lane down the per-sample path and the whole wave with it, and a lane owned whole blocks
//                  (64 bytes: every load instruction touched 64 half-lines, eight-way bank conflicts on the way into LDS without the padding).  Now the
//                  window is staged from the aligned block below it to the one above (slot = window position + a: whole blocks only on interior tiles)
//                  as lane-contiguous 16-byte words - coalesced loads, conflict-free ds_write_b128 - and the four lanes of a quad, which share a block,
//                  share its phasor polynomial through quad broadcasts (three polynomials per thread and tile, as before);
//   prefetch       every `if (in range) load` was a basic block of its own and hipcc put s_waitcnt vmcnt(0) in front of each: the next tile's loads went
//                  out one at a time.  Lanes past the end repeat the last word instead - no branch, twelve loads back to back;
//   two barriers per tile: unchanged.  What is left is the pace of the memory system: 0.62-0.65 of 8 TB/s = 80 % of the copy yardstick.
// Same-box A/B (tools/ab_decim.hip, 2^26 samples): Tuner(.., 50) 0.152 -> 0.106 ms, Decimator(50) 0.116 -> 0.110, Decimator(25) 0.124 -> 0.110,
// rotator + decimation 25 0.171 -> 0.113.
#pragma once
#include "kernels_fir.h"

namespace lrhip {

constexpr int DECIM2_SPAN_MAX = 6128;                         // (span + 14) / 8 <= 768 blocks of eight = twelve 16-byte words per thread
constexpr int DECIM2_PAD_SLOTS = 16;                          // slots in front of / behind the window that the whole-block staging may write
#ifndef LRHIP_DECIM2_SPLIT
#define LRHIP_DECIM2_SPLIT 1      /* rotator form, tiles of <= 128 outputs: 1 - the upper half-wave takes the second half of the taps; 0 - it idles as in the plain form */
#endif

// Decimations that are a multiple of four: a thread stride of D samples would put 4 .. 32 lanes of a half-wave on one bank pair, so the window is dealt out
// over E = 4 / 8 / 16 phase arrays (sample p -> array p mod E, index p / E).  D is a multiple of E: a thread's tap r lies in array (a + r) mod E for EVERY
// lane, at index oi D / E + (a + r) / E - a lane stride of D / E samples, odd for D = E x odd (TunerBlock(.., 80) of rtlsdr_pocsag.lua / rtlsdr_ax25.lua: 5).
// The array stride is 16 / E mod 16 so that the staging writes of sixteen consecutive lanes (which walk through the arrays) fall on sixteen bank pairs.
__host__ __device__ __forceinline__ int decim2_esh(long D) { return (D & 3) ? 0 : (D & 7) ? 2 : (D & 15) ? 3 : 4; }
__host__ __device__ __forceinline__ int decim2_arr(int span, int esh) { return ((((span + DECIM2_PAD_SLOTS) >> esh) + 1 + 15) & ~15) + (16 >> esh); }
__host__ __device__ __forceinline__ int decim2_slots(int span, long D)
{
    const int esh = decim2_esh(D);
    return esh ? (decim2_arr(span, esh) << esh) : span + DECIM2_PAD_SLOTS;
}

// acc += x * h for a (re, im) pair and ONE real tap: HI = 0 takes the tap from the low half of `hp`, 1 from the high half (the taps arrive as float4 = two
// pairs, no register moves).  The two lanes of v_pk_fma_f32 are IEEE fma: the bits of fmaf(x.x, h, acc.x), fmaf(x.y, h, acc.y).
template <int HI>
__device__ __forceinline__ void pk_fma_tap(cf &acc, cf xv, cf hp)
{
    if (HI) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(xv), "v"(hp));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(xv), "v"(hp));
}

// the value of lane j of the caller's quad (lanes 4m .. 4m + 3)
__device__ __forceinline__ float quad_bcast(float v, int j)       // j: a constant after unrolling
{
    const int i = __builtin_bit_cast(int, v);
    switch (j & 3) {
        case 0: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(i, 0x00, 0xf, 0xf, true));
        case 1: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(i, 0x55, 0xf, 0xf, true));
        case 2: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(i, 0xaa, 0xf, 0xf, true));
        default: return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(i, 0xff, 0xf, 0xf, true));
    }
}

// PH: the phase-array layout (decimations that are a multiple of four); without it `esh` is the constant 0 and none of its code exists
template <bool ROT, int FMT = 0, bool PH = false>
__global__ __launch_bounds__(256, 3) void fir_decim_lds2_kernel(const float *__restrict__ hist, const float *__restrict__ x, const float *__restrict__ taps_rev,
                                                             float *__restrict__ y, int M, long n, long n_out, long first, long D, int OW, long ntiles,
                                                             uint64_t rot_step_fx, uint64_t rot_count0, float *__restrict__ hist_out, int post_op, int rounds,
                                                             double inv_gain, const float2 *__restrict__ disc_prev_in, float2 *__restrict__ disc_prev_out,
                                                             int one_chain)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *ldsT = lds;                                  // M reversed taps
    float *ldsX = lds + ((M + 3) & ~3);                 // staged samples as (re, im), slot = window position + a
    // FrequencyDiscriminatorBlock behind the filter (frequencydiscriminator.lua:62-78) as an epilogue: OW counts the STORED outputs of a tile; lane 0 of every
    // wave recomputes the output in front of its wave's first (the tile's first wave: in front of the tile), so no angle needs another wave's or another
    // workgroup's output - one redundant output in 31 (63), no exchange, no third barrier.  The stream's first angle takes the carried output *disc_prev_in.
    const int disc = disc_prev_out != nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (hist_out && blockIdx.x == 0)
        for (int i = tid; i < (M - 1) * 2; i += 256) hist_out[i] = stream_at_raw<FMT>(hist, x, n + i / 2, i % 2, M, n);
    for (int i = tid; i < M; i += 256) ldsT[i] = taps_rev[i];
    const int span = (int)((OW - 1 + disc) * D) + M;
    const int esh = PH ? decim2_esh(D) : 0, emask = (1 << esh) - 1, arr = PH ? decim2_arr(span, esh) : 0;
    auto slot_of = [&](int p) { return esh ? (p & emask) * arr + (p >> esh) : p; };
    // Rotator form: a thread takes the 16-byte words tid + 256 k of the window (lane-contiguous loads and LDS writes): always pair q = tid & 3 of an aligned
    // block of eight - it needs W[2q], W[2q + 1] only - and block (tid >> 2) + 64 k.  The block phasors P() are the expensive part (one polynomial each);
    // the four lanes of a quad share a block, so lane i of the quad evaluates the polynomials of the loads k = i, i + 4, i + 8 and the others fetch them with
    // a quad broadcast (v_mov_b32_dpp): three polynomials per thread and tile, as with a block per lane.
    [[maybe_unused]] cf wq0, wq1;
    if constexpr (ROT) {
        const int q = tid & 3;
        wq0 = q ? phasor_poly(rot_step_fx * (uint64_t)(2 * q)) : cf{1.f, 0.f};        // W[0] = 1 exactly (rot_tab of kernels_elem.h)
        wq1 = phasor_poly(rot_step_fx * (uint64_t)(2 * q + 1));
    }
    // register prefetch of the next tile: ROT - the lane's blocks as 16-byte words in its pair order; plain - lane-contiguous 16-byte words
    constexpr int KF = (DECIM2_SPAN_MAX + 2 + 511) / 512;             // 16-byte words (two samples) per thread
    static_assert(KF % 4 == 0, "the quad shares the block phasors of four consecutive loads");
    [[maybe_unused]] float4 raw[(ROT && !FMT) ? KF : 1];
    [[maybe_unused]] uint2 rawr[(ROT && FMT) ? KF : 1];
    [[maybe_unused]] float4 rawp[(!ROT && !FMT) ? KF : 1];
    [[maybe_unused]] uint2 rawq[(!ROT && FMT) ? KF : 1];
    bool have = false;
    const bool x_aligned = (reinterpret_cast<uintptr_t>(x) & (FMT ? (FMT == RX_FMT_S16LE ? 7 : 3) : 15)) == 0;
    auto prefetch = [&](long tt) {
        have = false;
        if (tt >= ntiles) return;
        const long g0n = first + (tt * OW - disc) * D - (M - 1);
        if constexpr (ROT) {
            const int an = (int)((rot_count0 + (uint64_t)g0n) & 7), nblkn = (span + an + 7) >> 3, nfn = 4 * nblkn;
            const long lo = g0n - an;                   // chunk index of slot 0
            have = lo >= 0 && lo + 8L * nblkn <= n && (lo & 1) == 0 && x_aligned && nfn <= KF * 256;
            if (!have) return;
            // no branch around a load: a lane past the last word repeats it (a conditional load is a basic block of its own, and hipcc then puts
            // s_waitcnt vmcnt(0) in front of every one of them - the registers might still be the target of the previous tile's loads on the path
            // that did not stage them - which serialises the prefetch; seen in the ISA of the first cut, 0.155 against 0.117 ms)
#pragma unroll
            for (int k = 0; k < KF; k++) {
                const int f = min(tid + 256 * k, nfn - 1);
                if constexpr (FMT != 0) {
                    const uint8_t *src0 = reinterpret_cast<const uint8_t *>(x) + 2 * rx_raw_bytes<FMT>() * lo;
                    if (FMT == RX_FMT_S16LE) rawr[k] = reinterpret_cast<const uint2 *>(src0)[f];
                    else rawr[k].x = reinterpret_cast<const unsigned *>(src0)[f];
                } else
                    raw[(ROT && !FMT) ? k : 0] = reinterpret_cast<const float4 *>(reinterpret_cast<const cf *>(x) + lo)[f];
            }
        } else {
            const int an = (int)(g0n & 1), nf = (span + an + 1) >> 1;
            const long lo = g0n - an;
            have = lo >= 0 && lo + 2L * nf <= n && x_aligned && nf <= KF * 256;
            if (!have) return;
#pragma unroll
            for (int k = 0; k < KF; k++) {
                const int f = min(tid + 256 * k, nf - 1);
                if constexpr (FMT != 0) {
                    const uint8_t *src0 = reinterpret_cast<const uint8_t *>(x) + 2 * rx_raw_bytes<FMT>() * lo;
                    if (FMT == RX_FMT_S16LE) rawq[k] = reinterpret_cast<const uint2 *>(src0)[f];
                    else rawq[k].x = reinterpret_cast<const unsigned *>(src0)[f];
                } else
                    rawp[(!ROT && !FMT) ? k : 0] = reinterpret_cast<const float4 *>(reinterpret_cast<const cf *>(x) + lo)[f];
            }
        }
    };
    // outputs of a tile over the threads: up to 128 -> 32 per wave in the lower half-wave (all four SIMDs filter), the upper half-wave idle (plain) or on the
    // second half of the taps of the same outputs (rotator form); more than 128 -> one per thread
    const bool spread = OW <= (disc ? 124 : 128);
    const int lane_l = spread ? (lane & 31) : lane, LWQ = (spread ? 32 : 64) - disc;      // output lanes of a wave; stored outputs per wave
    const int oi = wave * LWQ + lane_l - disc;          // tile-local output (-1: the one in front of the tile)
    const int part = spread ? lane >> 5 : 0;
    // one_chain (LRHIP_CHAIN_EXACT_ROTATOR / LRHIP_TUNER_EXACT: FirStage::rel_rot == false): every output stays ONE fmaf chain in the reference's tap order - the
    // bits of FrequencyTranslatorBlock and the filter run one by one (include/lrhip.h, LRHIP_CHAIN_EXACT); the upper half-wave then idles as in the plain form
    const bool split = spread && ROT && LRHIP_DECIM2_SPLIT && !one_chain;
    const int Mh = split ? (((M >> 1) + 15) & ~15) : M;           // taps [0, Mh) in part 0, [Mh, M) in part 1
    const int t_lo = part ? (Mh < M ? Mh : M) : 0, t_hi = part ? (split ? M : 0) : (Mh < M ? Mh : M);
    const long t_first = rounds > 0 ? (long)blockIdx.x * rounds : (long)blockIdx.x, t_step = rounds > 0 ? 1 : (long)gridDim.x;
    const long t_end = rounds > 0 ? (t_first + rounds < ntiles ? t_first + rounds : ntiles) : ntiles;
    // phase arrays: E | D, so the window's alignment modulo E is the launch's (tiles start OW D samples apart)
    [[maybe_unused]] int eoff[PH ? 16 : 1];
    if constexpr (PH) {
        const long g00 = first - (M - 1);
        const int a_lo = (ROT ? (int)((rot_count0 + (uint64_t)g00) & 7) : (int)(g00 & 1)) & emask;
#pragma unroll
        for (int j = 0; j < 16; j++) eoff[j] = ((a_lo + j) & emask) * arr + ((a_lo + j) >> esh);
    }
    prefetch(t_first);
    [[maybe_unused]] int trace_tile = 0;
    for (long t = t_first; t < t_end; t += t_step) {
        DECIM_STAMP(0);
        const long k0 = t * OW;                         // first output of the tile
        const long q0 = first + (k0 - disc) * D;        // stream position of window sample 0 (stream = [M-1 history | chunk])
        const long g0 = q0 - (M - 1);                   // the same as an index into x (negative: history)
        const int a = ROT ? (int)((rot_count0 + (uint64_t)g0) & 7) : (int)(g0 & 1);
        if (have) {
            if constexpr (ROT) {
                const int nf = 4 * ((span + a + 7) >> 3);
                cf pm[KF / 4];                          // this lane's share of the quad's block phasors
#pragma unroll
                for (int kk = 0; kk < KF / 4; kk++)
                    pm[kk] = phasor_poly(rot_step_fx * (rot_count0 + (uint64_t)(g0 - a + 8L * ((tid >> 2) + 64 * (4 * kk + (tid & 3))))));
                // (the layout test outside the unrolled loop: a branch per load would cost the loop its staggered s_waitcnt vmcnt(11 - k))
                auto rotated = [&](int k, cf &r0, cf &r1) {
                    cf pb;
                    pb.x = quad_bcast(pm[k >> 2].x, k);
                    pb.y = quad_bcast(pm[k >> 2].y, k);
                    float4 rq;
                    if constexpr (FMT != 0) rq = rx_raw_pair<FMT>(rawr[k]);
                    else rq = raw[(ROT && !FMT) ? k : 0];
                    r0 = cmul(cf{rq.x, rq.y}, cmul(pb, wq0));
                    r1 = cmul(cf{rq.z, rq.w}, cmul(pb, wq1));
                };
                if (!PH || esh == 0) {
#pragma unroll
                    for (int k = 0; k < KF; k++) {
                        const int f = tid + 256 * k;
                        cf r0, r1;
                        rotated(k, r0, r1);
                        if (f < nf) reinterpret_cast<float4 *>(ldsX)[f] = make_float4(r0.x, r0.y, r1.x, r1.y);
                    }
                } else {
                    // sample 2 tid + 512 k: array (2 tid) mod E - the thread's own, for every k - at index (2 tid) / E + 512 k / E; the odd sample one array on
                    cf *dst = reinterpret_cast<cf *>(ldsX) + slot_of(2 * tid);
                    const int kstep = 512 >> esh;
#pragma unroll
                    for (int k = 0; k < KF; k++) {
                        const int f = tid + 256 * k;
                        cf r0, r1;
                        rotated(k, r0, r1);
                        if (f < nf) {
                            dst[k * kstep] = r0;
                            dst[k * kstep + arr] = r1;
                        }
                    }
                }
            } else {
                const int nf = (span + a + 1) >> 1;
                if (!PH || esh == 0) {
#pragma unroll
                    for (int k = 0; k < KF; k++) {
                        const int f = min(tid + 256 * k, nf - 1);
                        float4 v;
                        if constexpr (FMT != 0) v = rx_raw_pair<FMT>(rawq[k]);
                        else v = rawp[(!ROT && !FMT) ? k : 0];
                        reinterpret_cast<float4 *>(ldsX)[f] = v;
                    }
                } else {
                    cf *dst = reinterpret_cast<cf *>(ldsX) + slot_of(2 * tid);
                    const int kstep = 512 >> esh;
#pragma unroll
                    for (int k = 0; k < KF; k++) {
                        float4 v;
                        if constexpr (FMT != 0) v = rx_raw_pair<FMT>(rawq[k]);
                        else v = rawp[(!ROT && !FMT) ? k : 0];
                        if (tid + 256 * k < nf) {
                            dst[k * kstep] = cf{v.x, v.y};
                            dst[k * kstep + arr] = cf{v.z, v.w};
                        }
                    }
                }
            }
        } else {
            // edge tiles (history in front, the end of the chunk behind) and unaligned chunks: one sample at a time, the same bits
            for (int w = tid; w < span; w += 256) {
                float2 v = make_float2(stream_at_raw<FMT>(hist, x, q0 + w, 0, M, n), stream_at_raw<FMT>(hist, x, q0 + w, 1, M, n));
                if constexpr (ROT) v = rotate_sample(v, rot_step_fx, rot_count0 + (uint64_t)(g0 + w));
                *reinterpret_cast<float2 *>(ldsX + 2 * slot_of(w + a)) = v;
            }
        }
        DECIM_STAMP(1);
        __syncthreads();
        DECIM_STAMP(2);
        prefetch(t + t_step < t_end ? t + t_step : ntiles);
        DECIM_STAMP(3);
        const long k = k0 + oi;
        const bool active = oi < OW && k >= 0 && k < n_out && t_lo < t_hi;
        cf acc = cf{0.f, 0.f};
        if (active && (!PH || esh == 0)) {
            const cf *xs = reinterpret_cast<const cf *>(ldsX) + (oi + disc) * (int)D + a;
            int tt = t_lo;
            for (; tt + 16 <= t_hi; tt += 16) {
                cf hp[8], xv[16];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float4 h4 = *reinterpret_cast<const float4 *>(ldsT + tt + 4 * q);
                    hp[2 * q] = cf{h4.x, h4.y};
                    hp[2 * q + 1] = cf{h4.z, h4.w};
                }
#pragma unroll
                for (int j = 0; j < 16; j++) xv[j] = xs[tt + j];
#pragma unroll
                for (int j = 0; j < 16; j += 2) {
                    pk_fma_tap<0>(acc, xv[j], hp[j >> 1]);
                    pk_fma_tap<1>(acc, xv[j + 1], hp[j >> 1]);
                }
            }
            for (; tt < t_hi; tt++) {
                const float h0 = ldsT[tt];
                acc = __builtin_elementwise_fma(xs[tt], cf{h0, h0}, acc);
            }
        } else if (active) {
            // phase arrays: the same chain; tap r of every lane lies in array (a + r) mod E at the lane's base + (a + r) / E.  With r = tt + j, tt a multiple
            // of 16 and E | 16: array (a_lo + j) mod E, index (a_lo + j) / E + a / E + tt / E - the sixteen offsets eoff[j] are the same for every tile of the launch
            const cf *xl = reinterpret_cast<const cf *>(ldsX) + (oi + disc) * (int)(D >> esh) + (a >> esh);
            int tt = t_lo;
            for (; tt + 16 <= t_hi; tt += 16) {
                cf hp[8], xv[16];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float4 h4 = *reinterpret_cast<const float4 *>(ldsT + tt + 4 * q);
                    hp[2 * q] = cf{h4.x, h4.y};
                    hp[2 * q + 1] = cf{h4.z, h4.w};
                }
                const cf *xg = xl + (tt >> esh);
#pragma unroll
                for (int j = 0; j < 16; j++) xv[j] = xg[eoff[PH ? j : 0]];
#pragma unroll
                for (int j = 0; j < 16; j += 2) {
                    pk_fma_tap<0>(acc, xv[j], hp[j >> 1]);
                    pk_fma_tap<1>(acc, xv[j + 1], hp[j >> 1]);
                }
            }
            for (; tt < t_hi; tt++) {
                const float h0 = ldsT[tt];
                const int al = (a & emask) + tt;
                acc = __builtin_elementwise_fma(xl[(al & emask) * arr + (al >> esh)], cf{h0, h0}, acc);
            }
        }
        if (split) {
            // the halves meet: part 1 hands its sum to part 0 of the same output (lane - 32)
            acc.x += __shfl_xor(acc.x, 32);
            acc.y += __shfl_xor(acc.y, 32);
        }
        if (disc) {
            // the output in front of this lane's: lane - 1 of the same wave (lane 0 of a wave holds only that predecessor)
            float2 pv = make_float2(__shfl_up(acc.x, 1), __shfl_up(acc.y, 1));
            if (k == 0) pv = *disc_prev_in;                                // the stream's first output: the carried one (zero at the very start)
            if (oi >= 0 && oi < OW && k < n_out && part == 0 && lane_l >= 1) {
                y[k] = discriminate(make_float2(acc.x, acc.y), pv, inv_gain);
                if (k == n_out - 1) *disc_prev_out = make_float2(acc.x, acc.y);
            }
        } else if (oi < OW && k < n_out && part == 0) {
            const float re = acc.x, im = acc.y;
            // post_op = 1 + a complex -> real element-wise operation folded into the store (ComplexMagnitude behind the AM receiver's tuner ...): Float32 out
            if (post_op) y[k] = post_op == 1 + UN_CMAG ? unary_c2r<UN_CMAG>(re, im) : post_op == 1 + UN_CPHASE ? unary_c2r<UN_CPHASE>(re, im) : post_op == 1 + UN_CREAL ? re : im;
            else nt_store(reinterpret_cast<float2 *>(y) + k, make_float2(re, im));
        }
        DECIM_STAMP(4);
        __syncthreads();
        DECIM_STAMP(5);
#ifdef LRHIP_DECIM_TRACE
        trace_tile++;
#endif
    }
}

}  // namespace lrhip
