// kernels_interp.h - polyphase interpolator on the register-window scheme of kernels_firwin2.h
// (part of liblrhip.so; included by lrhip.hip, one translation unit)
//
// radio/composites/interpolator.lua:31-34: [MultiplyConstant(L)] -> Upsampler(L) -> LowpassFilter(128, 1/L) on a ComplexFloat32 stream.
// Output n = q L + p of input position q is
//     y[q L + p] = sum_{j = J-1 .. 0} h[p + j L] * (c x[q - j]),        J = ceil(M / L), h zero-padded to J L taps
// - the nonzero terms of the zero-stuffed direct form in its own order (oldest sample first), so the bits are those of the unfused chain
// and of fir_resample_kernel (a zero tap adds exactly +-0 to a finite sum).
//
// fir_resample_kernel computes one output per thread and reads a tap and a sample from LDS for every two FMAs (Interpolator(5): 2.2 ms
// for 2^26 input samples, 18 % of the HBM roof).  Here a lane owns R = 5 consecutive input positions and all L phases of them: L R
// accumulators (ComplexFloat32 in a register pair), one 8-byte window read per tap step feeds L R packed FMAs whose L taps are scalar
// operands (fw_step5<HI, SG>: an SGPR pair loaded by s_load from the tap table, the half picked by op_sel; FW_TAPS_SGPR 0 = LDS broadcasts, the
// round-2 form; the window is a compile-time indexed register ring).  The 5 L outputs of a lane are
// contiguous in y; they go through LDS (the window's space, reused) so that the stores are 16 bytes per lane, consecutive lanes
// consecutive addresses.  Persistent workgroups over tiles of 1280 input positions.
#pragma once
#include "kernels_firwin2.h"

// threads per workgroup of fir_interp_kernel (a tile = FIP_NT x 5 input positions, its out-area FIP_NT x 5 L x 8 B of LDS)
#ifndef LRHIP_INTERP_NT
#define LRHIP_INTERP_NT 256
#endif

namespace lrhip {

#ifndef LRHIP_INTERP_HALVES
#define LRHIP_INTERP_HALVES 1
#endif
#ifndef LRHIP_INTERP_PIN
#define LRHIP_INTERP_PIN 1       /* pinned wait / load / multiply order of the tap loop's scalar loads (0: hipcc's own order) */
#endif
#ifndef LRHIP_RATIONAL_PIN
#define LRHIP_RATIONAL_PIN 0     /* the same for fir_rational_kernel: measured EQUAL on all six shapes (its s_loads are merged four steps at a time), off */
#endif
#ifndef LRHIP_INTERP_WAVES
#define LRHIP_INTERP_WAVES 3     /* waves per SIMD the register allocation aims at */
#endif
constexpr int FIP_NT = LRHIP_INTERP_NT, FIP_HALVES = LRHIP_INTERP_HALVES;

template <int L, int J>
struct FipGeom {
    static constexpr int R = 5;                                // input positions per lane (odd: lane stride of 5 samples = 40 B, conflict-free 8-byte reads)
    static constexpr int TQ = FIP_NT * R;                         // input positions per tile
    static constexpr int LP = (L + 3) & ~3;                    // taps per step in LDS, padded to float4s
    static constexpr int NQ = LP / 4;
    static constexpr int WN = TQ + J - 1;                      // window samples: position i <-> input q = qb - (J - 1) + i
    static constexpr int XN = WN + FWC_LA + 8;
    static constexpr int ON = TQ * L;                          // outputs per tile
    static constexpr int BUF = (XN > ON / FIP_HALVES ? XN : ON / FIP_HALVES) * 2;        // floats: window, then (after the tap loop) the tile's outputs
    static constexpr int LDS_FLOATS = BUF + J * LP;
};

// ttab[s * LP + p] = h[p + (J - 1 - s) L] (0 beyond the filter or the phase count): step s = 0 is the oldest sample
// dbg (LRHIP_INTERP_DBG in the environment, ablation only): 1 = no tap loop, 2 = no stores, 4 = plain instead of non-temporal stores.  At L = 5, 2^26 input
// samples: 0.71 ms whole, 0.60 without the tap loop, 0.48 without the stores - the 2.7 GB of output set the pace, the arithmetic hides under them in part.
template <int L, int J>
__global__ __launch_bounds__(FIP_NT, LRHIP_INTERP_WAVES) void fir_interp_kernel(const float *__restrict__ hist, const float *__restrict__ x, const float *__restrict__ ttab,
                                                             float *__restrict__ y, long n_in, int HQ, float c, float *__restrict__ hist_out, int dbg, int rounds)
{
    using G = FipGeom<L, J>;
    constexpr int R = G::R, LA = FWC_LA, C = R + LA;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *ldsX = lds;
    float *ldsT = lds + G::BUF;
    const int tid = threadIdx.x;
    // history carry: the last HQ raw input samples of [hist | x] (fir_resample_kernel's layout)
    if (hist_out && blockIdx.x == 0)
        for (int i = tid; i < HQ * 2; i += FIP_NT) {
            const long g = n_in - HQ + i / 2;
            hist_out[i] = g >= 0 ? x[g * 2 + i % 2] : hist[(g + HQ) * 2 + i % 2];
        }
    if (!FW_TAPS_SGPR)
        for (int i = tid; i < J * G::LP; i += FIP_NT) ldsT[i] = ttab[i];

    // Persistent workgroups, register prefetch: the loads of the NEXT tile are issued before this tile's tap loop - ahead of its 51 KB of output
    // stores in the CU's memory queue - and the stores of tile t drain while tile t + 1 is filtered (one tile per workgroup measured 0.98 ms for
    // 2^26 input samples at L = 5: 0.54 ms of stores + 0.27 ms of arithmetic + 0.3 ms of load latency and launch, almost nothing overlapped).
    const long ntiles = (n_in + G::TQ - 1) / G::TQ;
    constexpr int NPRE = (G::WN + FIP_NT - 1) / FIP_NT;
    cf pre[NPRE];
    bool have = false;
    auto prefetch = [&](long tt) {
        const long lo = tt * G::TQ - (J - 1);
        have = tt < ntiles && lo >= 0 && lo + G::WN <= n_in;
        if (have) {
            const cf *src = reinterpret_cast<const cf *>(x) + lo;
#pragma unroll
            for (int u = 0; u < NPRE; u++) {
                const int idx = tid + FIP_NT * u;
                pre[u] = src[idx < G::WN ? idx : G::WN - 1];      // clamped, unconditional
            }
        }
    };
    // Tile order.  rounds == 0: persistent, workgroup g walks tiles g, g + gridDim.x, ...  rounds > 0 (round 4): one-shot, workgroup g owns the `rounds`
    // CONSECUTIVE tiles from g * rounds on and the dispatcher hands workgroups out in address order - the order of the overlap-save kernels
    // (kernels_firfft.h), whose stream microbenchmark reaches 6.3 TB/s where the persistent stride stops at 4.4-5.4
    long t = rounds > 0 ? (long)blockIdx.x * rounds : (long)blockIdx.x;
    const long tstep = rounds > 0 ? 1 : (long)gridDim.x;
    const long tend = rounds > 0 ? (t + rounds < ntiles ? t + rounds : ntiles) : ntiles;
    prefetch(t);
    for (; t < tend; t += tstep) {
        const long qb = t * G::TQ;                              // first input position of the tile (chunk-relative)
        if (have) {
#pragma unroll
            for (int u = 0; u < NPRE; u++) {
                asm volatile("" : "+v"(pre[u]));                // the loaded registers are touched here and not earlier (hipcc would wait for them before the tap loop)
                const int i = tid + FIP_NT * u;
                if (i < G::WN) *reinterpret_cast<cf *>(ldsX + 2 * i) = pre[u] * c;      // multiplyconstant.lua: Float32 product, rounded once
            }
        } else {
            for (int i = tid; i < G::WN; i += FIP_NT) {
                const long g = qb - (J - 1) + i;
                cf v = cf{0.f, 0.f};
                if (g >= 0) { if (g < n_in) v = reinterpret_cast<const cf *>(x)[g]; }
                else if (g + HQ >= 0) v = cf{hist[(g + HQ) * 2], hist[(g + HQ) * 2 + 1]};
                *reinterpret_cast<cf *>(ldsX + 2 * i) = v * c;
            }
        }
        __syncthreads();
        prefetch(t + tstep < tend ? t + tstep : ntiles);

        cf acc[L][R];
#pragma unroll
        for (int p = 0; p < L; p++)
#pragma unroll
            for (int i = 0; i < R; i++) acc[p][i] = cf{0.f, 0.f};
        if (!(dbg & 1)) {
            const float *base = ldsX + 2 * (R * tid);           // lane's window: sample r at base + 2 r, r = i + s for position i at step s
            cf W[C];
            float4 T[2][G::NQ];
            auto ld = [&](int r) { return *reinterpret_cast<const cf *>(base + 2 * r); };
            static_for<R - 1 + LA>([&](auto I) { constexpr int r = decltype(I)::value; W[r % C] = ld(r); });
#pragma unroll
            for (int k = 0; k < G::NQ; k++) T[0][k] = (FW_TAPS_SGPR ? uniform_load4(ttab + 4 * k) : *reinterpret_cast<const float4 *>(ldsT + 4 * k));
            static_for<J>([&](auto Sx) {
                constexpr int s = decltype(Sx)::value, rn = R - 1 + s + LA;
                // Scalar loads return out of order: the only wait for them is lgkmcnt(0), which also drains whatever was issued since.  Left to itself hipcc
                // sinks the s_loads of step s + 1 below the FMAs of step s and waits right behind them - the full scalar-cache latency every other step
                // (round 4, read off the ISA: 13 `s_waitcnt lgkmcnt(0)` directly behind an s_load; the tap loop alone ran at 60 % of the VALU rate).  The order
                // is pinned instead: touch this step's taps (the wait lands HERE, on loads that are a whole step old), then issue the next step's, then multiply
                if constexpr (FW_TAPS_SGPR && LRHIP_INTERP_PIN) {
#pragma unroll
                    for (int k = 0; k < G::NQ; k++) asm volatile("" ::"s"(T[s & 1][k].x), "s"(T[s & 1][k].y), "s"(T[s & 1][k].z), "s"(T[s & 1][k].w));
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (s + 1 < J) {
#pragma unroll
                    for (int k = 0; k < G::NQ; k++) T[(s + 1) & 1][k] = (FW_TAPS_SGPR ? uniform_load4(ttab + (s + 1) * G::LP + 4 * k) : *reinterpret_cast<const float4 *>(ldsT + (s + 1) * G::LP + 4 * k));
                }
                if constexpr (rn <= R - 1 + J - 1) W[rn % C] = ld(rn);
                if constexpr (FW_TAPS_SGPR && LRHIP_INTERP_PIN) __builtin_amdgcn_sched_barrier(0);
                static_for<L>([&](auto Px) {
                    constexpr int p = decltype(Px)::value;
                    const float4 tq = T[s & 1][p >> 2];
                    const cf tp = (p & 2) ? cf{tq.z, tq.w} : cf{tq.x, tq.y};
                    fw_step5<(p & 1), FW_TAPS_SGPR>(acc[p], tp, W[s % C], W[(s + 1) % C], W[(s + 2) % C], W[(s + 3) % C], W[(s + 4) % C]);
                });
            });
        }
        __syncthreads();                                        // every wave is done with the window: its space becomes the out-area
        const long o0 = qb * L, n_out = n_in * L;
        const long cnt_all = n_out - o0 < G::ON ? n_out - o0 : G::ON;   // outputs of this tile
        // FIP_HALVES = 2: the out-area holds half a tile - the lower half of the threads park their outputs and everybody stores them, then the upper half
#pragma unroll
        for (int hh = 0; hh < FIP_HALVES; hh++) {
            constexpr int HT = FIP_NT / FIP_HALVES, HON = G::ON / FIP_HALVES;
            if (hh) __syncthreads();
            if (FIP_HALVES == 1 || tid / HT == hh) {
                const int tl = tid - hh * HT;
#pragma unroll
                for (int i = 0; i < R; i++)
#pragma unroll
                    for (int p = 0; p < L; p++) *reinterpret_cast<cf *>(ldsX + 2 * ((R * tl + i) * L + p)) = acc[p][i];
            }
            __syncthreads();
            const long cnt = cnt_all - (long)hh * HON < HON ? cnt_all - (long)hh * HON : HON;
            float *yo = y + 2 * (o0 + (long)hh * HON);
            if (!(dbg & 2)) {
                constexpr int NK = (HON / 2 + FIP_NT - 1) / FIP_NT;     // 16-byte stores per thread of a whole (half-)tile
                if ((reinterpret_cast<uintptr_t>(yo) & 15) == 0 && cnt == HON && !(dbg & 4)) {
                    // whole tile: every LDS read is issued before the first store.  (Round 4: the rolled loop below - one ds_read_b128, a wait, one store and two
                    // branches per trip - was the "stores behind a barrier drain a quarter slower than from registers" of rounds 2-3: a wave had ONE store in flight)
                    f32x4 v[NK];
#pragma unroll
                    for (int kk = 0; kk < NK; kk++) {
                        const int k = tid + FIP_NT * kk;
                        if (kk < NK - 1 || (HON / 2) % FIP_NT == 0 || k < HON / 2) v[kk] = *reinterpret_cast<const f32x4 *>(ldsX + 4 * k);
                    }
#pragma unroll
                    for (int kk = 0; kk < NK; kk++) {
                        const int k = tid + FIP_NT * kk;
                        if (kk < NK - 1 || (HON / 2) % FIP_NT == 0 || k < HON / 2) __builtin_nontemporal_store(v[kk], reinterpret_cast<f32x4 *>(yo + 4 * k));
                    }
                } else if ((reinterpret_cast<uintptr_t>(yo) & 15) == 0) {
                    for (int k = tid; 2 * k < cnt; k += FIP_NT) {
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(ldsX + 4 * k);
                        if (2 * k + 1 < cnt) {
                            if (dbg & 4) *reinterpret_cast<f32x4 *>(yo + 4 * k) = v;
                            else __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(yo + 4 * k));      // written once, never re-read here: 0.710 against 0.724 ms
                        } else {
                            *reinterpret_cast<float2 *>(yo + 4 * k) = make_float2(v[0], v[1]);
                        }
                    }
                } else {
                    for (int k = tid; k < cnt; k += FIP_NT) *reinterpret_cast<float2 *>(yo + 2 * k) = *reinterpret_cast<const float2 *>(ldsX + 2 * k);
                }
            }
        }
        __syncthreads();                                        // the out-area is read: the next tile's window may be staged over it
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// RationalResamplerBlock (radio/composites/rationalresampler.lua:37-44): [MultiplyConstant(L)] -> Upsampler(L) -> LowpassFilter(128) ->
// Downsampler(D) on a ComplexFloat32 stream, gcd(L, D) = 1.  Output m sits at the upsampled position n' = m D (absolute); with q = n' div L,
// p = n' mod L it is the polyphase sum above.  fir_interp_kernel with every phase computed and only the kept outputs stored was measured at
// 1.11 ms for (3, 2) on 2^26 samples against 0.86-0.92 ms for the one-output-per-thread fir_resample_kernel (18 % of the HBM roof, 12 % of the
// vector peak: bound by neither).  This kernel computes ONLY the kept (position, phase) pairs: a group of D input positions that starts at
// an absolute position = 0 mod D yields L outputs, always the same pairs (i, p) with (i L + p) mod D = 0 - a compile-time pattern.  A lane
// owns R = G D positions (G groups, R L / D outputs that are contiguous in y); per tap step one new 8-byte window sample and R L / D packed
// FMAs, every one of them useful.  For even R the window is padded by one sample per R (lane stride R + 1: odd, conflict-free 8-byte reads;
// the pad moves a lane's sample r by the compile-time amount r / R).  Same oldest-first order of the nonzero terms as the unfused chain:
// bit-identical to it and to fir_resample_kernel.
// ------------------------------------------------------------------------------------------------------------------------------------
template <int L, int D, int J, int R>
struct FrrGeom {
    static_assert(R % D == 0, "a lane owns whole groups of D input positions");
    static constexpr int TQ = 256 * R;                         // input positions per tile
    static constexpr int OUTL = R * L / D;                     // outputs per lane
    static constexpr int LP = (L + 3) & ~3, NQ = LP / 4;
    static constexpr bool PADW = (R % 2) == 0;
    static constexpr int WN = TQ + J - 1;                      // window samples: position w <-> absolute input A + w - (J - 1)
    __host__ __device__ static constexpr int phys(int w) { return PADW ? w + w / R : w; }
    static constexpr int XN = phys(WN) + FWC_LA + 10;
    static constexpr int ON = TQ * L / D;                      // outputs per tile
    static constexpr int BUF = (XN > ON ? XN : ON) * 2;
    static constexpr int LDS_FLOATS = BUF + J * LP;
};

template <int L, int D, int J, int R>
__global__ __launch_bounds__(256, 3) void fir_rational_kernel(const float *__restrict__ hist, const float *__restrict__ x, const float *__restrict__ ttab,
                                                               float *__restrict__ y, long n_in, long n_out, uint64_t m0, uint64_t Q0, int HQ, float c,
                                                               float *__restrict__ hist_out, int rounds)
{
    using G = FrrGeom<L, D, J, R>;
    constexpr int LA = FWC_LA, C = R + LA, OUTL = G::OUTL;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *ldsX = lds;
    float *ldsT = lds + G::BUF;
    const int tid = threadIdx.x;
    if (hist_out && blockIdx.x == 0)
        for (int i = tid; i < HQ * 2; i += 256) {
            const long g = n_in - HQ + i / 2;
            hist_out[i] = g >= 0 ? x[g * 2 + i % 2] : hist[(g + HQ) * 2 + i % 2];
        }
    if (!FW_TAPS_SGPR)
        for (int i = tid; i < J * G::LP; i += 256) ldsT[i] = ttab[i];

    // tiles are anchored at the absolute input position A0 = Q0 rounded down to a multiple of D (the up to D - 1 positions in front of the chunk
    // are history, their outputs were emitted by the previous chunk: m < m0 is skipped)
    const long lead = (long)(Q0 % (uint64_t)D);                // chunk-relative index of A0 is -lead
    const long ntiles = (n_in + lead + G::TQ - 1) / G::TQ;
    const uint64_t mt0 = (Q0 - (uint64_t)lead) * (uint64_t)L / (uint64_t)D;      // output index of tile 0's first kept pair (exact: A0 = 0 mod D)
    constexpr int NPRE = (G::WN + 255) / 256;
    cf pre[NPRE];
    bool have = false;
    auto prefetch = [&](long tt) {
        const long lo = tt * G::TQ - lead - (J - 1);            // chunk-relative index of window sample 0
        have = tt < ntiles && lo >= 0 && lo + G::WN <= n_in;
        if (have) {
            const cf *src = reinterpret_cast<const cf *>(x) + lo;
#pragma unroll
            for (int u = 0; u < NPRE; u++) {
                const int idx = tid + 256 * u;
                pre[u] = src[idx < G::WN ? idx : G::WN - 1];
            }
        }
    };
    // tile order: fir_interp_kernel's (rounds > 0: a run of consecutive tiles per workgroup, workgroups in address order)
    long t = rounds > 0 ? (long)blockIdx.x * rounds : (long)blockIdx.x;
    const long tstep = rounds > 0 ? 1 : (long)gridDim.x;
    const long tend = rounds > 0 ? (t + rounds < ntiles ? t + rounds : ntiles) : ntiles;
    prefetch(t);
    for (; t < tend; t += tstep) {
        const long qb = t * G::TQ - lead;                       // chunk-relative index of the tile's first input position
        if (have) {
#pragma unroll
            for (int u = 0; u < NPRE; u++) {
                asm volatile("" : "+v"(pre[u]));
                const int i = tid + 256 * u;
                if (i < G::WN) *reinterpret_cast<cf *>(ldsX + 2 * G::phys(i)) = pre[u] * c;
            }
        } else {
            for (int i = tid; i < G::WN; i += 256) {
                const long g = qb - (J - 1) + i;
                cf v = cf{0.f, 0.f};
                if (g >= 0) { if (g < n_in) v = reinterpret_cast<const cf *>(x)[g]; }
                else if (g + HQ >= 0) v = cf{hist[(g + HQ) * 2], hist[(g + HQ) * 2 + 1]};
                *reinterpret_cast<cf *>(ldsX + 2 * G::phys(i)) = v * c;
            }
        }
        __syncthreads();
        prefetch(t + tstep < tend ? t + tstep : ntiles);

        // accumulator a <-> the a-th kept pair (i, p) of the lane in output order: n' = i L + p = a D
        cf acc[OUTL];
#pragma unroll
        for (int a = 0; a < OUTL; a++) acc[a] = cf{0.f, 0.f};
        {
            // lane's window: sample r (position i at step s: r = i + s) at base + 2 (r + r / R) when padded
            const float *base = ldsX + 2 * ((G::PADW ? R + 1 : R) * tid);
            cf W[C];
            float4 T[2][G::NQ];
            auto ld = [&](int r) { return *reinterpret_cast<const cf *>(base + 2 * (G::PADW ? r + r / R : r)); };
            static_for<R - 1 + LA>([&](auto I) { constexpr int r = decltype(I)::value; W[r % C] = ld(r); });
#pragma unroll
            for (int k = 0; k < G::NQ; k++) T[0][k] = (FW_TAPS_SGPR ? uniform_load4(ttab + 4 * k) : *reinterpret_cast<const float4 *>(ldsT + 4 * k));
            static_for<J>([&](auto Sx) {
                constexpr int s = decltype(Sx)::value, rn = R - 1 + s + LA;
                // the scalar loads' wait / load / multiply order is pinned as in fir_interp_kernel
                if constexpr (FW_TAPS_SGPR && LRHIP_RATIONAL_PIN) {
#pragma unroll
                    for (int k = 0; k < G::NQ; k++) asm volatile("" ::"s"(T[s & 1][k].x), "s"(T[s & 1][k].y), "s"(T[s & 1][k].z), "s"(T[s & 1][k].w));
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (s + 1 < J) {
#pragma unroll
                    for (int k = 0; k < G::NQ; k++) T[(s + 1) & 1][k] = (FW_TAPS_SGPR ? uniform_load4(ttab + (s + 1) * G::LP + 4 * k) : *reinterpret_cast<const float4 *>(ldsT + (s + 1) * G::LP + 4 * k));
                }
                if constexpr (rn <= R - 1 + J - 1) W[rn % C] = ld(rn);
                if constexpr (FW_TAPS_SGPR && LRHIP_RATIONAL_PIN) __builtin_amdgcn_sched_barrier(0);
                static_for<OUTL>([&](auto Ax) {
                    constexpr int a = decltype(Ax)::value, np = a * D, i = np / L, p = np % L;
                    const float4 tq = T[s & 1][p >> 2];
                    const float tp = (p & 3) == 0 ? tq.x : (p & 3) == 1 ? tq.y : (p & 3) == 2 ? tq.z : tq.w;
                    acc[a] = __builtin_elementwise_fma(W[(s + i) % C], cf{tp, tp}, acc[a]);
                });
            });
        }
        __syncthreads();                                        // every wave is done with the window: its space becomes the out-area
#pragma unroll
        for (int a = 0; a < OUTL; a++) *reinterpret_cast<cf *>(ldsX + 2 * (OUTL * tid + a)) = acc[a];
        __syncthreads();
        // tile outputs: absolute index mt0 + t ON + k, chunk-relative o = that - m0 (negative in front of the chunk: skipped)
        const long ob = (long)(mt0 - m0) + t * (long)G::ON;     // chunk-relative index of the tile's output 0 (mt0 <= m0: may be negative for t = 0)
        const int k0 = ob < 0 ? (int)(-ob) : 0;
        const long cnt = n_out - ob < G::ON ? n_out - ob : G::ON;
        float *yo = y + 2 * ob;
        constexpr int NK = (G::ON / 2 + 255) / 256;
        if (k0 == 0 && (reinterpret_cast<uintptr_t>(yo) & 15) == 0 && cnt == G::ON && G::ON % 2 == 0) {
            // whole tile: all LDS reads, then all stores (fir_interp_kernel)
            f32x4 v[NK];
#pragma unroll
            for (int kk = 0; kk < NK; kk++) {
                const int k = tid + 256 * kk;
                if (kk < NK - 1 || (G::ON / 2) % 256 == 0 || k < G::ON / 2) v[kk] = *reinterpret_cast<const f32x4 *>(ldsX + 4 * k);
            }
#pragma unroll
            for (int kk = 0; kk < NK; kk++) {
                const int k = tid + 256 * kk;
                if (kk < NK - 1 || (G::ON / 2) % 256 == 0 || k < G::ON / 2) __builtin_nontemporal_store(v[kk], reinterpret_cast<f32x4 *>(yo + 4 * k));
            }
        } else if (k0 == 0 && (reinterpret_cast<uintptr_t>(yo) & 15) == 0) {
            for (int k = tid; 2 * k < cnt; k += 256) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(ldsX + 4 * k);
                if (2 * k + 1 < cnt) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(yo + 4 * k));
                else *reinterpret_cast<float2 *>(yo + 4 * k) = make_float2(v[0], v[1]);
            }
        } else {
            for (int k = k0 + tid; k < cnt; k += 256) *reinterpret_cast<float2 *>(yo + 2 * k) = *reinterpret_cast<const float2 *>(ldsX + 2 * k);
        }
        __syncthreads();
    }
}

}  // namespace lrhip
