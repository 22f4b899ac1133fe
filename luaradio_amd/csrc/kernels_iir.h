// kernels_iir.h - IIRFilterBlock (radio/blocks/signal/iirfilter.lua:113-181) on the GPU.
//
//   y[n] = ( sum_{j<nb} b[j] x[n-j]  -  sum_{1<=i<na} a[i] y[n-i] ) / a[0]
//
// (rewritten in round 1: direct 16-B global access per thread instead of an LDS transposition; optional fused downsampler)
// The WBFM chain needs the first-order case (FMDeemphasisFilterBlock = SinglepoleLowpassFilterBlock,
// singlepolelowpassfilter.lua:55-67: nb = na = 2).  A linear recurrence is a scan over affine maps of the
// P = na-1 element output state, so it is done in three data-parallel passes:
//   pass 1 (per 4096-sample tile): every thread runs the recurrence over its 16-sample chunk from ZERO state;
//           a Kogge-Stone scan over the 256 chunk end-states with the precomputed transition powers
//           A^(16*2^k) gives the tile's zero-state end state.
//   pass 2 (one thread): true tile start states  s_t = E_{t-1} + A^4096 s_{t-1}  from the carried state.
//   pass 3 (per tile): same as pass 1 but seeded with the true tile start state; every thread then re-runs
//           its chunk from its TRUE start state and writes y.  Inside a chunk the arithmetic is the plain
//           sequential f32 recurrence.
// Traffic: 2 reads + 1 write of the stream (12 B per Float32 sample against an 8 B algorithmic minimum).
// Orders 1-8 take the scan paths; above that the sequential kernel (one thread per component) - correct, not fast.
#pragma once
#include "common.h"

namespace lrhip {

constexpr int IIR_LC = 16;                    // samples per thread chunk
constexpr int IIR_TILE = 256 * IIR_LC;        // samples per workgroup tile
constexpr int IIR_MAX_NB = 16;
constexpr int IIR_MAX_P = 8;                  // scan path; above this -> sequential kernel
constexpr int IIR_SEQ_MAX = 32;

// The transition powers A^(LC*2^k), k = 0..8 (k = 8 is the whole-tile transition), row-major PxP each, live in device
// memory (`tpow`, 9*P*P floats; uniform addresses -> scalar loads): by value they would not fit the kernel-argument limit.
struct IirCoeffs {
    int nb, P;
    float b[IIR_MAX_NB];                 // b[j]/a0
    float a[IIR_MAX_P];                  // a[i+1]/a0
};

// state convention: st[i] = y[n-1-i] (st[0] newest).  One homogeneous step: y = -sum a[i] st[i].
// The scan over chunk / tile states (S_c = z_c + A^k S_{c-1}) runs in Float32 up to order 4 and in double above: the powers
// of a companion matrix of order 5-8 have large, cancelling entries, and Float32 there loses up to 50x against the plain
// sequential Float32 recurrence (measured on four pole pairs at radius 0.9995).  The per-sample recurrence stays Float32.
template <int P> struct IirScanT { using T = float; };
template <> struct IirScanT<5> { using T = double; };
template <> struct IirScanT<6> { using T = double; };
template <> struct IirScanT<7> { using T = double; };
template <> struct IirScanT<8> { using T = double; };

template <int P, typename T>
__device__ __forceinline__ void mat_apply(const T *M, const T *v, T *out)
{
#pragma unroll
    for (int r = 0; r < P; r++) {
        T acc = 0;
#pragma unroll
        for (int c = 0; c < P; c++) acc = fma(M[r * P + c], v[c], acc);
        out[r] = acc;
    }
}

// inclusive Kogge-Stone scan over the 256 chunk states already in sst (and synchronised): S_c = z_c + A^(LC) S_{c-1}
template <int S, int P, typename T>
__device__ __forceinline__ void iir_block_scan(T (*sst)[256][P], const T *__restrict__ tpow)
{
    const int tid = threadIdx.x;
    for (int lvl = 0; lvl < 8; lvl++) {
        const int off = 1 << lvl;
        T nv[S][P];
#pragma unroll
        for (int c = 0; c < S; c++) {
#pragma unroll
            for (int k = 0; k < P; k++) nv[c][k] = sst[c][tid][k];
            if (tid >= off) {
                T prev[P], tmp[P];
#pragma unroll
                for (int k = 0; k < P; k++) prev[k] = sst[c][tid - off][k];
                mat_apply<P, T>(tpow + lvl * P * P, prev, tmp);
#pragma unroll
                for (int k = 0; k < P; k++) nv[c][k] += tmp[k];
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < S; c++)
#pragma unroll
            for (int k = 0; k < P; k++) sst[c][tid][k] = nv[c][k];
        __syncthreads();
    }
}

// Each thread owns LC = 16 consecutive samples and reads them straight from global memory as 16-B vectors (a wave
// covers 64 x 64 B = 4 KB contiguous; the four loads of a thread reuse its L1 lines), plus the NBT-1 samples before its
// chunk for the feed-forward taps; LDS is used only for the scan over chunk end states.
// NBT = compile-time bound on the number of feed-forward taps (2: single-pole filters; 16: general).
// FINAL pass only: `dec`/`dfirst` fuse a following DownsamplerBlock (radio/blocks/signal/downsampler.lua:45-56):
// only samples with (index - dfirst) % dec == 0 are stored, at (index - dfirst) / dec; it also publishes the last P
// outputs of the chunk as the carried state.
template <int S, int P, bool FINAL, int NBT>
__global__ __launch_bounds__(256) void iir_scan_kernel(const float *__restrict__ x, float *__restrict__ y, long n,
                                                       const float *__restrict__ xhist,      // nb-1 samples before x[0]
                                                       const typename IirScanT<P>::T *__restrict__ tile_start, // FINAL: [tile][S][P]
                                                       typename IirScanT<P>::T *__restrict__ tile_end,         // !FINAL: [tile][S][P]
                                                       const float *__restrict__ state_in, float *__restrict__ state_out,
                                                       long dec, long dfirst, IirCoeffs co, const typename IirScanT<P>::T *__restrict__ tpow)
{
    using ST = typename IirScanT<P>::T;
    constexpr int LC = IIR_LC, TILE = IIR_TILE, PV = NBT - 1;
    __shared__ ST sst[S][256][P];            // chunk end states

    const int tid = threadIdx.x;
    const long t0 = (long)blockIdx.x * TILE;
    const long c0 = t0 + (long)tid * LC;     // first sample of this thread's chunk
    const int nb = co.nb;

    // ---- load: xs[c][PV + i] = x[c0 + i], xs[c][PV - j] = x[c0 - j]
    float xs[S][PV + LC];
    const bool vec = (c0 + LC <= n) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    if (vec) {
        const float4 *src = reinterpret_cast<const float4 *>(x + c0 * S);
#pragma unroll
        for (int q = 0; q < LC * S / 4; q++) {
            float4 v = src[q];
            if (S == 1) {
                xs[0][PV + 4 * q] = v.x; xs[0][PV + 4 * q + 1] = v.y; xs[0][PV + 4 * q + 2] = v.z; xs[0][PV + 4 * q + 3] = v.w;
            } else {
                xs[0][PV + 2 * q] = v.x; xs[S - 1][PV + 2 * q] = v.y; xs[0][PV + 2 * q + 1] = v.z; xs[S - 1][PV + 2 * q + 1] = v.w;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < LC; i++)
#pragma unroll
            for (int c = 0; c < S; c++) xs[c][PV + i] = (c0 + i < n) ? x[(c0 + i) * S + c] : 0.f;
    }
#pragma unroll
    for (int j = 1; j <= PV; j++)
#pragma unroll
        for (int c = 0; c < S; c++) {
            long g = c0 - j;
            float v = 0.f;
            if (j < nb && c0 < n) v = g >= 0 ? x[g * S + c] : xhist[(g + (nb - 1)) * S + c];
            xs[c][PV - j] = v;
        }

    // ---- feed-forward part u[i] = sum_j b[j] x[i-j]
    float u[S][LC];
#pragma unroll
    for (int c = 0; c < S; c++)
#pragma unroll
        for (int i = 0; i < LC; i++) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < NBT; j++)
                if (j < nb) acc = fmaf(co.b[j], xs[c][PV + i - j], acc);
            u[c][i] = acc;
        }

    // ---- zero-state run over the chunk -> chunk end state
    float st[S][P];
#pragma unroll
    for (int c = 0; c < S; c++) {
#pragma unroll
        for (int k = 0; k < P; k++) st[c][k] = 0.f;
#pragma unroll
        for (int i = 0; i < LC; i++) {
            float v = u[c][i];
#pragma unroll
            for (int k = 0; k < P; k++) v = fmaf(-co.a[k], st[c][k], v);
#pragma unroll
            for (int k = P - 1; k > 0; k--) st[c][k] = st[c][k - 1];
            st[c][0] = v;
        }
    }

    // ---- inclusive Kogge-Stone scan of chunk end states: S_c = z_c + A^LC S_{c-1}  (S_{-1} = tile start state)
#pragma unroll
    for (int c = 0; c < S; c++) {
        ST z[P];
#pragma unroll
        for (int k = 0; k < P; k++) z[k] = (ST)st[c][k];
        if (FINAL && tid == 0) {
            ST ts[P], tmp[P];
#pragma unroll
            for (int k = 0; k < P; k++) ts[k] = tile_start[((long)blockIdx.x * S + c) * P + k];
            mat_apply<P, ST>(tpow, ts, tmp);
#pragma unroll
            for (int k = 0; k < P; k++) z[k] += tmp[k];
        }
#pragma unroll
        for (int k = 0; k < P; k++) sst[c][tid][k] = z[k];
    }
    __syncthreads();
    iir_block_scan<S, P, ST>(sst, tpow);

    if (!FINAL) {
        if (tid == 255)
#pragma unroll
            for (int c = 0; c < S; c++)
#pragma unroll
                for (int k = 0; k < P; k++) tile_end[((long)blockIdx.x * S + c) * P + k] = sst[c][255][k];
        return;
    }

    // ---- true start state of this chunk = scanned end state of the previous chunk (or the tile start state); re-run
#pragma unroll
    for (int c = 0; c < S; c++) {
#pragma unroll
        for (int k = 0; k < P; k++)
            st[c][k] = (float)(tid ? sst[c][tid - 1][k] : tile_start[((long)blockIdx.x * S + c) * P + k]);
#pragma unroll
        for (int i = 0; i < LC; i++) {
            float v = u[c][i];
#pragma unroll
            for (int k = 0; k < P; k++) v = fmaf(-co.a[k], st[c][k], v);
#pragma unroll
            for (int k = P - 1; k > 0; k--) st[c][k] = st[c][k - 1];
            st[c][0] = v;
            u[c][i] = v;                       // u now holds y
            long g = c0 + i;
            if (g < n && g >= n - P) state_out[c * P + (int)(n - 1 - g)] = v;       // carried state y[n-1-k]
        }
    }
    if (blockIdx.x == 0 && tid == 0 && n < P)       // tiny chunk: older state entries shift down
#pragma unroll
        for (int c = 0; c < S; c++)
            for (int k = (int)n; k < P; k++) state_out[c * P + k] = state_in[c * P + k - (int)n];

    // ---- store
    if (dec == 1) {
        if (vec && (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
            float4 *dst = reinterpret_cast<float4 *>(y + c0 * S);
#pragma unroll
            for (int q = 0; q < LC * S / 4; q++)
                dst[q] = S == 1 ? make_float4(u[0][4 * q], u[0][4 * q + 1], u[0][4 * q + 2], u[0][4 * q + 3])
                                : make_float4(u[0][2 * q], u[S - 1][2 * q], u[0][2 * q + 1], u[S - 1][2 * q + 1]);
        } else {
#pragma unroll
            for (int i = 0; i < LC; i++)
#pragma unroll
                for (int c = 0; c < S; c++)
                    if (c0 + i < n) y[(c0 + i) * S + c] = u[c][i];
        }
    } else {
        // first kept index >= c0:  dfirst + ceil((c0 - dfirst)/dec)*dec
        long k0 = c0 <= dfirst ? 0 : (c0 - dfirst + dec - 1) / dec;
        long g0 = dfirst + k0 * dec;
#pragma unroll
        for (int i = 0; i < LC; i++) {
            long g = c0 + i;
            if (g == g0 && g < n) {
#pragma unroll
                for (int c = 0; c < S; c++) y[k0 * S + c] = u[c][i];
                k0++;
                g0 += dec;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Single-launch variant for filters whose memory is shorter than a tile in Float32 terms (host check: every entry of
// A^(warm*TILE) is below 1e-46, i.e. underflows to zero - true for every single-pole audio filter of the reference's
// receivers).  A workgroup emits `run` consecutive tiles, carrying the state from tile to tile in LDS, after running
// `warm` tiles in front of them from ZERO state with the output discarded: by the time the first emitted tile starts,
// the zero start has decayed out of the Float32 state.  The workgroups whose warm-up would reach before the chunk start
// there with the true carried state instead.  One launch, (1 + warm/run) reads + 1 write of the stream, no inter-
// workgroup dependency.  Same per-sample arithmetic as the three-pass form.
// ------------------------------------------------------------------------------------------------------------
// LRHIP_IIR_CF_WGS (round 6, A/B): resident workgroups per CU the ComplexFloat32 instantiations are compiled for (0 = no bound: 159 registers, three per CU).
// Measured on one box, three alternations, 2^26 samples: 4 (128 registers, 72 bytes of scratch per lane) is SLOWER - 5 ff / 3 fb 0.230 -> 0.299 ms, biquad 0.220 ->
// 0.273; 1 (no register pressure from the bound at all) is equal.  The ComplexFloat32 recurrence holds 16 complex samples per thread; it stays at three per CU.
#ifndef LRHIP_IIR_CF_WGS
#define LRHIP_IIR_CF_WGS 0
#endif
template <int S, int P, int NBT>
__global__ __launch_bounds__(256, (S == 2 && LRHIP_IIR_CF_WGS > 0) ? LRHIP_IIR_CF_WGS : 1) void iir_stream_kernel(const float *__restrict__ x, float *__restrict__ y, long n,
                                                         const float *__restrict__ xhist, const float *__restrict__ state_in,
                                                         float *__restrict__ state_out, long dec, long dfirst, int run, int warm, int warm_chunks, IirCoeffs co,
                                                         float *__restrict__ xhist_out, const typename IirScanT<P>::T *__restrict__ tpow, int no_coal)
{
    using ST = typename IirScanT<P>::T;
    constexpr int LC = IIR_LC, TILE = IIR_TILE, PV = NBT - 1;
    // orders 1-4 scan inside the waves (shuffles): LDS only holds the four wave totals and, in the last row, the tile end state
    constexpr int NS = P <= 4 ? 8 : 256, LAST = NS - 1;
    __shared__ ST sst[S][NS][P];
    __shared__ ST carry[S][P];
    // Coalesced tile I/O (round 3).  A thread's zero-state run wants LC = 16 CONSECUTIVE samples - 64 B (Float32) or 128 B (ComplexFloat32) per thread - but
    // a load instruction whose 64 lanes are 64 / 128 B apart moves 4.1 / 2.2 TB/s where consecutive lanes on consecutive 16 B move 5.3-6.2 (tools/mb_chunk.hip,
    // 512 MiB in + out): that mapping, not the scan, was what held the recurrences at 38-53 % of the roof.  Whole tiles inside the chunk are now loaded and
    // stored lane-contiguously and transposed through LDS: linear float4 i of the tile lives at i + i / F4 (one float4 of padding per thread chunk: the chunk
    // stride F4 + 1 is odd in 16-byte units, both directions conflict-free).  Edge tiles and partial warm-up tiles keep the per-thread path; same arithmetic.
    constexpr int F4 = IIR_LC * S / 4;
    __shared__ float4 tr[256 * (F4 + 1)];
    const int tid = threadIdx.x;
    const int nb = co.nb;
    // carried feed-forward history (iir_state_kernel's job): the last nb-1 inputs
    if (blockIdx.x == 0 && tid < (nb - 1) * S) {
        int r = tid / S, c = tid % S;
        long g = n - (nb - 1) + r;
        xhist_out[tid] = g >= 0 ? x[g * S + c] : xhist[(g + (nb - 1)) * S + c];
    }
    const long first_tile = (long)blockIdx.x * run;
    long tb = first_tile - warm;
    const bool from_true_state = tb <= 0;
    if (tb < 0) tb = 0;
    if (tid < S * P) carry[tid / P][tid % P] = from_true_state ? (ST)state_in[tid] : (ST)0;
    // first order: every thread keeps the carried state in a register (it computes it anyway), no LDS word to race on
    ST creg[S];
#pragma unroll
    for (int c = 0; c < S; c++) creg[c] = (P == 1 && from_true_state) ? (ST)state_in[c] : (ST)0;
    __syncthreads();
    // LRHIP_IIR_TPRE 1 (round 4, A/B): the scan's uniform tables (orders 1-2: 7 P^2 floats) requested HERE, ahead of the tile's loads, instead of where they are
    // used (where each level's s_load sits directly in front of an `s_waitcnt lgkmcnt(0)`: 7 scalar-cache latencies in the dependent chain of a tile).  Measured
    // EQUAL (0.2283 against 0.2278 ms for the 5 / 3-tap ComplexFloat32 entry, three alternations; 26 more SGPRs spill to lanes): off
#ifndef LRHIP_IIR_TPRE
#define LRHIP_IIR_TPRE 0
#endif
    constexpr bool TPRE = LRHIP_IIR_TPRE && P <= 2;
    ST tpr[TPRE ? 7 * P * P : 1];
    if constexpr (TPRE) {
#pragma unroll
        for (int i = 0; i < 7 * P * P; i++) tpr[i] = tpow[i];
        __builtin_amdgcn_sched_barrier(0);
    }
    const ST *tps = TPRE ? tpr : tpow;       // levels 0 .. 6 of the wave scan

    for (long tt = tb; tt < first_tile + run && tt * TILE < n; tt++) {
        const bool emit = tt >= first_tile;
        const long c0 = tt * TILE + (long)tid * LC;
        // a warm-up tile only has to cover the recurrence's memory: with warm_chunks > 0 just its last warm_chunks chunks are read and run (the
        // others count as zero input), so a workgroup that owns ONE tile re-reads 16 * warm_chunks samples instead of 4096 - which is what lets the
        // launch be one-shot, a workgroup per tile in address order (the DRAM pages in flight stay compact: common.h grid_for)
        const bool skip = !emit && !from_true_state && warm_chunks > 0 && tid < 256 - warm_chunks;
        // ---- load (as iir_scan_kernel)
        float xs[S][PV + LC];
        const bool vec = (c0 + LC <= n) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
        // workgroup-uniform: the whole tile lies inside the chunk, 16-byte aligned, and nobody skips its chunk
        const bool coal = !(no_coal & 1) && (tt + 1) * TILE <= n && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && !(!emit && !from_true_state && warm_chunks > 0);
        if (coal) {
            const float4 *src = reinterpret_cast<const float4 *>(x + tt * (long)TILE * S);
#pragma unroll
            for (int j = 0; j < F4; j++) {
                const int idx = 256 * j + tid;
                tr[idx + idx / F4] = src[idx];
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < F4; q++) {
                const float4 v = tr[tid * (F4 + 1) + q];
                if (S == 1) {
                    xs[0][PV + 4 * q] = v.x; xs[0][PV + 4 * q + 1] = v.y; xs[0][PV + 4 * q + 2] = v.z; xs[0][PV + 4 * q + 3] = v.w;
                } else {
                    xs[0][PV + 2 * q] = v.x; xs[S - 1][PV + 2 * q] = v.y; xs[0][PV + 2 * q + 1] = v.z; xs[S - 1][PV + 2 * q + 1] = v.w;
                }
            }
        } else if (skip) {
#pragma unroll
            for (int i = 0; i < PV + LC; i++)
#pragma unroll
                for (int c = 0; c < S; c++) xs[c][i] = 0.f;
        } else if (vec) {
            const float4 *src = reinterpret_cast<const float4 *>(x + c0 * S);
#pragma unroll
            for (int q = 0; q < LC * S / 4; q++) {
                float4 v = src[q];
                if (S == 1) {
                    xs[0][PV + 4 * q] = v.x; xs[0][PV + 4 * q + 1] = v.y; xs[0][PV + 4 * q + 2] = v.z; xs[0][PV + 4 * q + 3] = v.w;
                } else {
                    xs[0][PV + 2 * q] = v.x; xs[S - 1][PV + 2 * q] = v.y; xs[0][PV + 2 * q + 1] = v.z; xs[S - 1][PV + 2 * q + 1] = v.w;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < LC; i++)
#pragma unroll
                for (int c = 0; c < S; c++) xs[c][PV + i] = (c0 + i < n) ? x[(c0 + i) * S + c] : 0.f;
        }
        if (coal && tid > 0) {
            // the previous thread's chunk is in LDS: its last PV samples are the tail of chunk tid - 1 (PV < LC), read as whole 16-byte words
            constexpr int NQ = (PV * S + 3) / 4;
#pragma unroll
            for (int q = F4 - NQ; q < F4; q++) {
                const float4 v = tr[(tid - 1) * (F4 + 1) + q];
                const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int f = 4 * q + r, i = f / S, c = f % S, j = LC - i;      // float f of the chunk = sample i, component c = j samples back
                    if (j >= 1 && j <= PV) xs[c][PV - j] = e[r];
                }
            }
        } else {
#pragma unroll
            for (int j = 1; j <= PV; j++)
#pragma unroll
                for (int c = 0; c < S; c++) {
                    long g = c0 - j;
                    float v = 0.f;
                    if (j < nb && c0 < n && !skip) v = g >= 0 ? x[g * S + c] : xhist[(g + (nb - 1)) * S + c];
                    xs[c][PV - j] = v;
                }
        }
        float u[S][LC];
        // feed-forward part, with the tap count a compile-time constant where that is cheap (nb in (NBT - 4, NBT]: the predicated form below costs a
        // v_cndmask per term, 262 of the 815 vector instructions of this section for the suite's 5-tap entry); same terms, same order
        auto ff = [&](auto nbc) {
            constexpr int NB = decltype(nbc)::value;
#pragma unroll
            for (int c = 0; c < S; c++)
#pragma unroll
                for (int i = 0; i < LC; i++) {
                    float acc = 0.f;
#pragma unroll
                    for (int j = 0; j < NBT; j++)
                        if (NB ? j < NB : j < nb) acc = fmaf(co.b[j], xs[c][PV + i - j], acc);
                    u[c][i] = acc;
                }
        };
        if (nb == NBT) ff(std::integral_constant<int, NBT>());
        else if (NBT > 1 && nb == NBT - 1) ff(std::integral_constant<int, (NBT > 1 ? NBT - 1 : 0)>());
        else if (NBT > 2 && NBT <= 8 && nb == NBT - 2) ff(std::integral_constant<int, (NBT > 2 ? NBT - 2 : 0)>());
        else if (NBT > 3 && NBT <= 8 && nb == NBT - 3) ff(std::integral_constant<int, (NBT > 3 ? NBT - 3 : 0)>());
        else ff(std::integral_constant<int, 0>());
        // ---- zero-state run, scan of the chunk end states seeded with the carried tile start state
        float st[S][P];
#pragma unroll
        for (int c = 0; c < S; c++) {
#pragma unroll
            for (int k = 0; k < P; k++) st[c][k] = 0.f;
#pragma unroll
            for (int i = 0; i < LC; i++) {
                float v = u[c][i];
#pragma unroll
                for (int k = 0; k < P; k++) v = fmaf(-co.a[k], st[c][k], v);
#pragma unroll
                for (int k = P - 1; k > 0; k--) st[c][k] = st[c][k - 1];
                st[c][0] = v;
            }
        }
        if constexpr (P == 1) {
            // first-order recurrences (every single-pole filter of the reference's receivers): the scan over the 256 chunk end states runs inside
            // the waves with shuffles and across the four waves through four LDS words - 2 barriers per tile instead of 18.  tpow[l] = p^(16 * 2^l),
            // tpow[9 + l] = p^(16 (l + 1)) (host, from double).
            const int lane = tid & 63, wave = tid >> 6;
            ST zs[S], Sx[S], Ew[S];
#pragma unroll
            for (int c = 0; c < S; c++) {
                zs[c] = (ST)st[c][0];
                if (tid == 0) zs[c] += fma(tps[0], creg[c], (ST)0);
#pragma unroll
                for (int l = 0; l < 6; l++) {
                    const ST prev = __shfl_up(zs[c], 1 << l);
                    if (lane >= (1 << l)) zs[c] += fma(tps[l], prev, (ST)0);
                }
                if (lane == 63) sst[c][wave][0] = zs[c];
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < S; c++) {
                ST E = 0;
                Ew[c] = 0;
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    if (w == wave) Ew[c] = E;
                    E = sst[c][w][0] + fma(tps[6], E, (ST)0);
                }
                Sx[c] = zs[c] + fma(tpow[9 + lane], Ew[c], (ST)0);          // true end state of this chunk
                const ST up = __shfl_up(Sx[c], 1);
                st[c][0] = (float)(tid == 0 ? creg[c] : lane ? up : Ew[c]);
                creg[c] = E;                                                // state at the end of the tile = the next tile's carry
            }
            __syncthreads();                                                // the wave totals are read: the next tile may overwrite them
            if (tid == 255)
#pragma unroll
                for (int c = 0; c < S; c++) sst[c][LAST][0] = creg[c];       // read below (by this thread) when the chunk ends with this tile
        } else if constexpr (P <= 4) {
            // orders 2-4 (round 3): the same two-barrier wave scan on P-vectors.  tpow + l P^2 = A^(16 * 2^l) (l <= 8), tpow + (9 + l) P^2 = A^(16 (l + 1))
            // (l < 64), host, from double.  The LDS Kogge-Stone below (16 barriers and 8 x 3 P S LDS words per thread and tile) is left to orders 5-8.
            const int lane = tid & 63, wave = tid >> 6;
            ST zs[S][P], tmp[P];
#pragma unroll
            for (int c = 0; c < S; c++) {
#pragma unroll
                for (int k = 0; k < P; k++) zs[c][k] = (ST)st[c][k];
                if (tid == 0) {
                    ST ts[P];
#pragma unroll
                    for (int k = 0; k < P; k++) ts[k] = carry[c][k];
                    mat_apply<P, ST>(tps, ts, tmp);
#pragma unroll
                    for (int k = 0; k < P; k++) zs[c][k] += tmp[k];
                }
#pragma unroll
                for (int l = 0; l < 6; l++) {
                    ST prev[P];
#pragma unroll
                    for (int k = 0; k < P; k++) prev[k] = __shfl_up(zs[c][k], 1 << l);
                    mat_apply<P, ST>(tps + l * P * P, prev, tmp);
                    if (lane >= (1 << l))
#pragma unroll
                        for (int k = 0; k < P; k++) zs[c][k] += tmp[k];
                }
                if (lane == 63)
#pragma unroll
                    for (int k = 0; k < P; k++) sst[c][wave][k] = zs[c][k];
            }
            __syncthreads();
            ST Mp[P * P];                                                   // A^(16 (lane + 1))
#pragma unroll
            for (int i = 0; i < P * P; i++) Mp[i] = tpow[(9 + lane) * P * P + i];
#pragma unroll
            for (int c = 0; c < S; c++) {
                ST E[P], Ew[P], Sx[P];
#pragma unroll
                for (int k = 0; k < P; k++) E[k] = Ew[k] = 0;
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    if (w == wave)
#pragma unroll
                        for (int k = 0; k < P; k++) Ew[k] = E[k];
                    mat_apply<P, ST>(tps + 6 * P * P, E, tmp);
#pragma unroll
                    for (int k = 0; k < P; k++) E[k] = sst[c][w][k] + tmp[k];
                }
                mat_apply<P, ST>(Mp, Ew, tmp);
#pragma unroll
                for (int k = 0; k < P; k++) {
                    Sx[k] = zs[c][k] + tmp[k];                              // true end state of this chunk
                    const ST up = __shfl_up(Sx[k], 1);
                    st[c][k] = (float)(tid == 0 ? carry[c][k] : lane ? up : Ew[k]);
                }
#pragma unroll
                for (int k = 0; k < P; k++) zs[c][k] = E[k];                // state at the end of the tile
            }
            __syncthreads();                                                // wave totals and carry are read
            if (tid == 255)
#pragma unroll
                for (int c = 0; c < S; c++)
#pragma unroll
                    for (int k = 0; k < P; k++) { carry[c][k] = zs[c][k]; sst[c][LAST][k] = zs[c][k]; }
        } else {
#pragma unroll
        for (int c = 0; c < S; c++) {
            ST z[P];
#pragma unroll
            for (int k = 0; k < P; k++) z[k] = (ST)st[c][k];
            if (tid == 0) {
                ST ts[P], tmp[P];
#pragma unroll
                for (int k = 0; k < P; k++) ts[k] = carry[c][k];
                mat_apply<P, ST>(tpow, ts, tmp);
#pragma unroll
                for (int k = 0; k < P; k++) z[k] += tmp[k];
            }
#pragma unroll
            for (int k = 0; k < P; k++) sst[c][tid][k] = z[k];
        }
        __syncthreads();
        iir_block_scan<S, P, ST>(sst, tpow);
        // ---- true start state of this chunk; the tile end state becomes the next tile's carried state
#pragma unroll
        for (int c = 0; c < S; c++)
#pragma unroll
            for (int k = 0; k < P; k++) st[c][k] = (float)(tid ? sst[c][tid - 1][k] : carry[c][k]);
        __syncthreads();
        if (tid == 255)
#pragma unroll
            for (int c = 0; c < S; c++)
#pragma unroll
                for (int k = 0; k < P; k++) carry[c][k] = sst[c][255][k];
        }
        if (emit) {
#pragma unroll
            for (int c = 0; c < S; c++)
#pragma unroll
                for (int i = 0; i < LC; i++) {
                    float v = u[c][i];
#pragma unroll
                    for (int k = 0; k < P; k++) v = fmaf(-co.a[k], st[c][k], v);
#pragma unroll
                    for (int k = P - 1; k > 0; k--) st[c][k] = st[c][k - 1];
                    st[c][0] = v;
                    u[c][i] = v;
                }
            // carried state = the chunk's last P outputs: only the tile(s) holding them look for them
            // (a chunk that ends WITH this tile hands over the scanned tile end state instead - below - which is what the next tile
            // of an uninterrupted run is given: the scan and the re-run round differently, and time partitions must not see that)
            if ((tt + 1) * TILE + P > n && n != (tt + 1) * TILE)
#pragma unroll
                for (int c = 0; c < S; c++)
#pragma unroll
                    for (int i = 0; i < LC; i++) {
                        const long g = c0 + i;
                        if (g < n && g >= n - P) state_out[c * P + (int)(n - 1 - g)] = u[c][i];
                    }
            if (tid == 255 && n == (tt + 1) * TILE)
#pragma unroll
                for (int c = 0; c < S; c++)
#pragma unroll
                    for (int k = 0; k < P; k++) state_out[c * P + k] = (float)sst[c][LAST][k];
            if (tt == 0 && tid == 0 && n < P)
#pragma unroll
                for (int c = 0; c < S; c++)
                    for (int k = (int)n; k < P; k++) state_out[c * P + k] = state_in[c * P + k - (int)n];
            if (dec == 1 && coal && (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
                // (every thread read its inputs out of `tr` before the scan's barriers: the buffer is free)
#pragma unroll
                for (int q = 0; q < F4; q++)
                    tr[tid * (F4 + 1) + q] = S == 1 ? make_float4(u[0][4 * q], u[0][4 * q + 1], u[0][4 * q + 2], u[0][4 * q + 3])
                                                    : make_float4(u[0][2 * q], u[S - 1][2 * q], u[0][2 * q + 1], u[S - 1][2 * q + 1]);
                __syncthreads();
                float4 *dst = reinterpret_cast<float4 *>(y + tt * (long)TILE * S);
#pragma unroll
                for (int j = 0; j < F4; j++) {
                    const int idx = 256 * j + tid;
                    if (no_coal & 2) dst[idx] = tr[idx + idx / F4];
                    else nt_store(dst + idx, tr[idx + idx / F4]);
                }
            } else if (dec == 1) {
                if (vec && (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
                    float4 *dst = reinterpret_cast<float4 *>(y + c0 * S);
#pragma unroll
                    for (int q = 0; q < LC * S / 4; q++)
                        dst[q] = S == 1 ? make_float4(u[0][4 * q], u[0][4 * q + 1], u[0][4 * q + 2], u[0][4 * q + 3])
                                        : make_float4(u[0][2 * q], u[S - 1][2 * q], u[0][2 * q + 1], u[S - 1][2 * q + 1]);
                } else {
#pragma unroll
                    for (int i = 0; i < LC; i++)
#pragma unroll
                        for (int c = 0; c < S; c++)
                            if (c0 + i < n) y[(c0 + i) * S + c] = u[c][i];
                }
            } else {
                long k0 = c0 <= dfirst ? 0 : (c0 - dfirst + dec - 1) / dec;
                long g0 = dfirst + k0 * dec;
#pragma unroll
                for (int i = 0; i < LC; i++) {
                    long g = c0 + i;
                    if (g == g0 && g < n) {
#pragma unroll
                        for (int c = 0; c < S; c++) y[k0 * S + c] = u[c][i];
                        k0++;
                        g0 += dec;
                    }
                }
            }
        }
        __syncthreads();      // carry is published, sst may be reused
    }
}

// pass 2: carry across tiles, s_t = E_{t-1} + A^TILE s_{t-1}, as a 256-thread scan: every thread owns a
// contiguous segment of `seg` tiles (sequential inside the segment, twice), segment end states are combined with a
// Kogge-Stone scan using tseg[k] = A^(TILE*seg*2^k), which iir_tseg_kernel computes in double from A^TILE per launch.
template <int P, typename T>
__global__ void iir_tseg_kernel(const double *__restrict__ ttile, long seg, T *__restrict__ tseg)
{
    if (threadIdx.x || blockIdx.x) return;
    double R[P * P], B[P * P], Tm[P * P];
    for (int i = 0; i < P * P; i++) { B[i] = ttile[i]; R[i] = (i / P == i % P) ? 1.0 : 0.0; }
    auto mul = [&](const double *X, const double *Y, double *Z) {          // Z = X * Y (Z may alias neither)
        for (int r = 0; r < P; r++)
            for (int c = 0; c < P; c++) {
                double acc = 0;
                for (int k = 0; k < P; k++) acc += X[r * P + k] * Y[k * P + c];
                Z[r * P + c] = acc;
            }
    };
    for (long e = seg; e > 0; e >>= 1) {                                   // R = ttile^seg by repeated squaring
        if (e & 1) { mul(R, B, Tm); for (int i = 0; i < P * P; i++) R[i] = Tm[i]; }
        mul(B, B, Tm);
        for (int i = 0; i < P * P; i++) B[i] = Tm[i];
    }
    for (int k = 0; k < 8; k++) {
        for (int i = 0; i < P * P; i++) tseg[k * P * P + i] = (T)R[i];
        mul(R, R, Tm);
        for (int i = 0; i < P * P; i++) R[i] = Tm[i];
    }
}

template <int S, int P>
__global__ __launch_bounds__(256) void iir_carry_kernel(const typename IirScanT<P>::T *__restrict__ tile_end, typename IirScanT<P>::T *__restrict__ tile_start,
                                                        long ntiles, long seg, const float *__restrict__ state_in,
                                                        const typename IirScanT<P>::T *__restrict__ tseg, const typename IirScanT<P>::T *__restrict__ tpow)
{
    using ST = typename IirScanT<P>::T;
    __shared__ ST sst[S][256][P];
    const int tid = threadIdx.x;
    const long t0 = (long)tid * seg, t1 = (t0 + seg < ntiles) ? t0 + seg : ntiles;
    const ST *ttile = tpow + 8 * P * P;
    ST tmp[P];
#pragma unroll
    for (int c = 0; c < S; c++) {
        ST z[P];
#pragma unroll
        for (int k = 0; k < P; k++) z[k] = 0;
        for (long t = t0; t < t1; t++) {
            mat_apply<P, ST>(ttile, z, tmp);
#pragma unroll
            for (int k = 0; k < P; k++) z[k] = tile_end[(t * S + c) * P + k] + tmp[k];
        }
        if (tid == 0) {      // fold the carried state into segment 0: S_0 = z_0 + A^(TILE*seg) * carried
            ST ci[P];
#pragma unroll
            for (int k = 0; k < P; k++) ci[k] = (ST)state_in[c * P + k];
            mat_apply<P, ST>(tseg, ci, tmp);
#pragma unroll
            for (int k = 0; k < P; k++) z[k] += tmp[k];
        }
#pragma unroll
        for (int k = 0; k < P; k++) sst[c][tid][k] = z[k];
    }
    __syncthreads();
    iir_block_scan<S, P, ST>(sst, tseg);
#pragma unroll
    for (int c = 0; c < S; c++) {
        ST s[P];
#pragma unroll
        for (int k = 0; k < P; k++) s[k] = tid ? sst[c][tid - 1][k] : (ST)state_in[c * P + k];
        for (long t = t0; t < t1; t++) {
#pragma unroll
            for (int k = 0; k < P; k++) tile_start[(t * S + c) * P + k] = s[k];
            mat_apply<P, ST>(ttile, s, tmp);
#pragma unroll
            for (int k = 0; k < P; k++) s[k] = tile_end[(t * S + c) * P + k] + tmp[k];
        }
    }
}

// carried feed-forward history after the chunk: the last nb-1 inputs (the output state is published by the final pass)
template <int S>
__global__ void iir_state_kernel(const float *__restrict__ x, long n, int nb, const float *__restrict__ xhist_in,
                                 float *__restrict__ xhist_out)
{
    for (int i = threadIdx.x; i < (nb - 1) * S; i += blockDim.x) {
        int r = i / S, c = i % S;                 // r-th oldest of the nb-1 retained inputs
        long g = n - (nb - 1) + r;
        xhist_out[i] = g >= 0 ? x[g * S + c] : xhist_in[(g + (nb - 1)) * S + c];
    }
}

// Sequential fallback for high orders: one thread per component, state in registers/scratch.
struct IirSeqCoeffs {
    int nb, na;
    float b[IIR_SEQ_MAX], a[IIR_SEQ_MAX];   // raw taps, a[0] divides (iirfilter.lua:160-170 op order)
};
template <int S>
__global__ void iir_seq_kernel(const float *__restrict__ x, float *__restrict__ y, long n, IirSeqCoeffs co,
                               float *__restrict__ xs_state, float *__restrict__ ys_state)
{
    int c = threadIdx.x;
    if (c >= S) return;
    float xs[IIR_SEQ_MAX], ys[IIR_SEQ_MAX];
    for (int j = 0; j < co.nb; j++) xs[j] = xs_state[c * IIR_SEQ_MAX + j];
    for (int j = 0; j < co.na - 1; j++) ys[j] = ys_state[c * IIR_SEQ_MAX + j];
    for (long i = 0; i < n; i++) {
        for (int j = co.nb - 1; j > 0; j--) xs[j] = xs[j - 1];
        xs[0] = x[i * S + c];
        float acc = 0.f;
        for (int j = 0; j < co.nb; j++) acc = acc + xs[j] * co.b[j];
        for (int j = 0; j < co.na - 1; j++) acc = acc - ys[j] * co.a[j + 1];
        acc = acc / co.a[0];
        for (int j = co.na - 2; j > 0; j--) ys[j] = ys[j - 1];
        if (co.na > 1) ys[0] = acc;
        y[i * S + c] = acc;
    }
    for (int j = 0; j < co.nb; j++) xs_state[c * IIR_SEQ_MAX + j] = xs[j];
    for (int j = 0; j < co.na - 1; j++) ys_state[c * IIR_SEQ_MAX + j] = ys[j];
}

}  // namespace lrhip
