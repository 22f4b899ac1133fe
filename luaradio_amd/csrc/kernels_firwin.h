// kernels_firwin.h - direct-form FIR on the packed-f32 VALU with a sliding register window ("window" kernels).
//
// Reference semantics as kernels_fir.h (radio/blocks/signal/firfilter.lua:230-305): with s = [last M-1 inputs | chunk] and
// taps_rev[j] = h[M-1-j], output k is the chain  acc = fmaf(s[q_k + j], taps_rev[j], acc), j ascending, q_k = first + k*D.
//
// Why a second direct form next to the Toeplitz-MFMA one: the f32 matrix pipe and the VALU are the same datapath on gfx950
// (DESIGN.md 4.3), and the banded-Toeplitz product spends 11 % (D = 1) to 37 % (D = 5) of its MACs on structural zeros.
// Here every issued v_pk_fma_f32 is a useful pair of MACs:
//   * a lane owns R consecutive outputs; at tap step j it needs the R samples s[q + D*i + j], i < R.  Going from step j to
//     j + 1 the window slides by one sample, so ONE new 8-byte LDS read per step feeds R packed FMAs (the other operands are
//     already in registers: a ring of D*(R-1)+1 samples, indexed at compile time because the tap loop is fully unrolled);
//   * the taps sit in LDS; one wave-uniform ds_read_b128 (a broadcast) brings four of them, and a tap enters the packed FMA
//     through op_sel (low or high half of its register pair for both result halves);
//   * the R FMAs of a step are ONE inline-asm block: left to itself hipcc clusters the FMAs of one accumulator back to back
//     (register pressure heuristic) and the dependent issue stalls cost 40 % of the VALU time (rocprofv3: SQ_WAIT_INST_ANY);
//     the blocks keep the R independent chains interleaved, and the compiler still places the LDS reads between them;
//   * ComplexFloat32 samples are the packed pairs as they are (re, im); for a Float32 stream a pair is two ADJACENT outputs
//     (n, n+1) and the operand pair (s[n+j], s[n+j+1]) must be an aligned 8-byte LDS word for even and for odd j, so the
//     tile is staged twice, the second copy shifted by one sample.
// Every output is the same fmaf chain as the oracle's LRO_MODE_FMA: bit-exact, and no zero-padding products, so a
// non-finite input sample reaches exactly the M outputs whose window holds it.
//
// fir_win_real_kernel<M, IIR>: Float32 stream, D = 1.  IIR = true fuses a first-order IIRFilterBlock (FMDeemphasisFilterBlock,
// singlepolelowpassfilter.lua:55-67 / iirfilter.lua:113-181) and a DownsamplerBlock (downsampler.lua:45-56) behind the filter:
// the lane's 16 consecutive filter outputs are exactly one chunk of the scan formulation of kernels_iir.h, so the recurrence
// runs on the accumulators and the 220.5 kHz audio of the WBFM receiver never reaches HBM.
#pragma once
#include "common.h"
#include "kernels_fir.h"
#include "kernels_iir.h"
#include "pk_math.h"
#include <utility>

namespace lrhip {

// ------------------------------------------------------------------------------------------------------------
// Float32 stream, D = 1
// ------------------------------------------------------------------------------------------------------------
constexpr int FWR_TILE = 4096;                                   // outputs per tile: 256 lanes x 16
// LDS index of tile coordinate i: 2 floats of padding per 16, so the lane stride is 18 dwords and the 32 lanes of a
// ds_read_b64 group hit 32 distinct even banks
__host__ __device__ constexpr int fwr_phys(int i) { return i + 2 * (i >> 4); }

template <int M>
struct FwrGeom {
    static constexpr int HALO = ((M - 1 + 3) / 4) * 4;           // staged samples in front of the tile's first output (16-B aligned start)
    static constexpr int E0 = HALO - (M - 1);                    // tile coordinate of s[q_0] of local output 0
    static constexpr int SPAN = FWR_TILE + HALO;                 // staged floats (multiple of 4)
    static constexpr int NF4 = SPAN / 4;
    static constexpr int NPRE = (NF4 + 255) / 256;
    static constexpr int LDSA = fwr_phys(SPAN) + 16;             // floats per copy
    static constexpr int LDS_FLOATS = 2 * LDSA + 16 + M;             // two copies, the exchange words, the taps
};

struct FwrParams {
    const float *hist;         // M-1 inputs before x[0]
    const float *x;
    long n;
    const float *taps_rev;
    float *y;
    float *hist_out;           // the other history buffer (block 0 writes it), or null
    long run;                  // tiles per workgroup (contiguous)
    // ---- fused first-order IIR + downsampler (IIR = true)
    float b0, b1, na1;         // b[0]/a0, b[1]/a0 (nb = 1: unused), -a[1]/a0
    int nb;
    const float *ptab;         // ptab[l] = (-a1)^(16 (l+1)), l < 64, rounded from double
    const float *vhist;        // the filter output before the chunk (the IIR's x[n-1])
    const float *state_in;     // y[-1]
    float *state_out, *vhist_out;
    long dec, dfirst;
    int warm_waves;            // waves of the warm-up tile that compute (the zero start must have decayed: |a1|^(1024 warm_waves) < 1e-12)
};

// one float4 of staged samples (tile coordinates a .. a+3, a = 0 mod 4) into both copies: E[i] = s[i], O[i] = s[i+1]
__device__ __forceinline__ void fwr_put4(float *ldsE, float *ldsO, int a, float4 v)
{
    const int p = fwr_phys(a);
    *reinterpret_cast<float2 *>(ldsE + p) = make_float2(v.x, v.y);
    *reinterpret_cast<float2 *>(ldsE + p + 2) = make_float2(v.z, v.w);
    if (a) ldsO[fwr_phys(a - 1)] = v.x;
    *reinterpret_cast<float2 *>(ldsO + p) = make_float2(v.y, v.z);
    ldsO[p + 2] = v.w;
}

template <typename F, int... Js>
__host__ __device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Js...>) { (f(std::integral_constant<int, Js>{}), ...); }
// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): an unrolled loop whose index is a constant expression in the body
template <int N, typename F>
__host__ __device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// Real taps are wave-uniform: read from the global table at compile-time offsets (uniform_load*, common.h) they arrive by scalar loads in SGPR
// pairs and feed the packed FMAs as a scalar operand (SG = true) - no LDS tap reads at all.  Where the LDS pipe is the bound that is a large step
// (RationalResampler(3, 4): 0.317 -> 0.185 ms, (2, 3): 0.26 -> 0.195, (3, 2): 0.33 -> 0.28; Interpolator(5) 0.726 -> 0.696; Hilbert(129) +1-5 %);
// the short window kernels lose by it (16 real taps cf32: 0.172 -> 0.180 ms - a scalar load drains lgkmcnt, and with it the window reads in
// flight, every four taps) and keep their taps in the LDS.  FW_TAPS_SGPR 0 = LDS-staged taps everywhere (A/B).
#ifndef FWR_SGPR_MIN_M
#define FWR_SGPR_MIN_M 128       /* fir_win_real_kernel: tap counts from which the taps come by scalar loads */
#endif
#ifndef FW_TAPS_SGPR
#define FW_TAPS_SGPR 1
#endif
// acc[i] = fma(tap, w_i, acc[i]) for the eight accumulators; the tap is the low (HI = 0) or high (HI = 1) half of t
template <int HI, bool SG = false>
__device__ __forceinline__ void fw_step8(cf (&a)[8], cf t, cf w0, cf w1, cf w2, cf w3, cf w4, cf w5, cf w6, cf w7)
{
    if constexpr (SG) {
        if constexpr (HI == 0)
            asm("v_pk_fma_f32 %0, %8, %9, %0 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %1, %8, %10, %1 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %2, %8, %11, %2 op_sel_hi:[0,1,1]\n\t"
                "v_pk_fma_f32 %3, %8, %12, %3 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %4, %8, %13, %4 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %5, %8, %14, %5 op_sel_hi:[0,1,1]\n\t"
                "v_pk_fma_f32 %6, %8, %15, %6 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %7, %8, %16, %7 op_sel_hi:[0,1,1]"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                : "s"(t), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "v"(w5), "v"(w6), "v"(w7));
        else
            asm("v_pk_fma_f32 %0, %8, %9, %0 op_sel:[1,0,0]\n\tv_pk_fma_f32 %1, %8, %10, %1 op_sel:[1,0,0]\n\tv_pk_fma_f32 %2, %8, %11, %2 op_sel:[1,0,0]\n\t"
                "v_pk_fma_f32 %3, %8, %12, %3 op_sel:[1,0,0]\n\tv_pk_fma_f32 %4, %8, %13, %4 op_sel:[1,0,0]\n\tv_pk_fma_f32 %5, %8, %14, %5 op_sel:[1,0,0]\n\t"
                "v_pk_fma_f32 %6, %8, %15, %6 op_sel:[1,0,0]\n\tv_pk_fma_f32 %7, %8, %16, %7 op_sel:[1,0,0]"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                : "s"(t), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "v"(w5), "v"(w6), "v"(w7));
    } else {
        if constexpr (HI == 0)
            asm("v_pk_fma_f32 %0, %8, %9, %0 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %1, %8, %10, %1 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %2, %8, %11, %2 op_sel_hi:[0,1,1]\n\t"
                "v_pk_fma_f32 %3, %8, %12, %3 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %4, %8, %13, %4 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %5, %8, %14, %5 op_sel_hi:[0,1,1]\n\t"
                "v_pk_fma_f32 %6, %8, %15, %6 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %7, %8, %16, %7 op_sel_hi:[0,1,1]"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                : "v"(t), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "v"(w5), "v"(w6), "v"(w7));
        else
            asm("v_pk_fma_f32 %0, %8, %9, %0 op_sel:[1,0,0]\n\tv_pk_fma_f32 %1, %8, %10, %1 op_sel:[1,0,0]\n\tv_pk_fma_f32 %2, %8, %11, %2 op_sel:[1,0,0]\n\t"
                "v_pk_fma_f32 %3, %8, %12, %3 op_sel:[1,0,0]\n\tv_pk_fma_f32 %4, %8, %13, %4 op_sel:[1,0,0]\n\tv_pk_fma_f32 %5, %8, %14, %5 op_sel:[1,0,0]\n\t"
                "v_pk_fma_f32 %6, %8, %15, %6 op_sel:[1,0,0]\n\tv_pk_fma_f32 %7, %8, %16, %7 op_sel:[1,0,0]"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                : "v"(t), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "v"(w5), "v"(w6), "v"(w7));
    }
}

// The tap loop.  Pair i of the lane = outputs 16 L + 2i, 16 L + 2i + 1; at step j its operand starts at tile coordinate
// 16 L + 2i + q, q = j + E0: an aligned 8-byte word of copy (q & 1) at pair index m = i + (q >> 1).  Same-parity steps slide
// the ring of pairs by one.  ldsT: the M reversed taps (M = 0 mod 4).
template <int M, int E0>
__device__ __forceinline__ void fwr_taps(const float *ldsT, const float *baseE, const float *baseO, cf (&acc)[8])
{
    static_assert(M % 4 == 0, "taps are read four at a time");
    constexpr int R = 8, LA = 2, NS = R + LA;
    cf W[2][NS];
    float4 T[2];
    auto ld = [&](int par, int m) { return *reinterpret_cast<const cf *>((par ? baseO : baseE) + 2 * m + 2 * (m >> 3)); };
#pragma unroll
    for (int i = 0; i < R; i++) acc[i] = cf{0.f, 0.f};
    static_for<2>([&](auto JJ) {
        constexpr int q = decltype(JJ)::value + E0, par = q & 1, h = q >> 1;
        static_for<R - 1 + LA>([&](auto I) { constexpr int i = decltype(I)::value; W[par][(i + h) % NS] = ld(par, i + h); });
    });
    constexpr bool SG = M >= FWR_SGPR_MIN_M;       // long tap loops: taps by scalar loads (ldsT is then the global table) - section 4.3 fact 5
    T[0] = SG ? uniform_load4(ldsT) : *reinterpret_cast<const float4 *>(ldsT);
    static_for<M>([&](auto J) {
        constexpr int j = decltype(J)::value, q = j + E0, par = q & 1, h = q >> 1;
        if constexpr ((j & 3) == 0 && j + 4 < M) T[((j >> 2) + 1) & 1] = SG ? uniform_load4(ldsT + j + 4) : *reinterpret_cast<const float4 *>(ldsT + j + 4);
        if constexpr (j + 2 * LA < M) W[par][(R - 1 + h + LA) % NS] = ld(par, R - 1 + h + LA);      // needed LA same-parity steps ahead
        const float4 tq = T[(j >> 2) & 1];
        const cf tp = (j & 2) ? cf{tq.z, tq.w} : cf{tq.x, tq.y};
        fw_step8<(j & 1), SG>(acc, tp, W[par][(0 + h) % NS], W[par][(1 + h) % NS], W[par][(2 + h) % NS], W[par][(3 + h) % NS], W[par][(4 + h) % NS],
                          W[par][(5 + h) % NS], W[par][(6 + h) % NS], W[par][(7 + h) % NS]);
    });
}

template <int M, bool IIR>
__global__ __launch_bounds__(256, 2) void fir_win_real_kernel(const FwrParams pr)
{
    using G = FwrGeom<M>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *ldsE = lds, *ldsO = lds + G::LDSA, *xch = lds + 2 * G::LDSA;      // xch[0..3]: last filter output of each wave; [4..7]: scan totals
    float *ldsT = xch + 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long n = pr.n;
    const float *__restrict__ x = pr.x;
    const float *__restrict__ hist = pr.hist;
    if (pr.hist_out && blockIdx.x == 0)
        for (int i = tid; i < M - 1; i += 256) pr.hist_out[i] = stream_at<1>(hist, x, n + i, 0, M, n);
    for (int i = tid; i < M; i += 256) ldsT[i] = pr.taps_rev[i];

    const long ntiles = (n + FWR_TILE - 1) / FWR_TILE;
    const long first_tile = (long)blockIdx.x * pr.run;
    const long t_end = first_tile + pr.run < ntiles ? first_tile + pr.run : ntiles;
    long tb = first_tile;
    float carry = 0.f, vprev = 0.f;                  // y[-1] and v[-1] of the tile about to be processed
    if (IIR) {
        if (first_tile == 0) { carry = pr.state_in[0]; vprev = pr.vhist[0]; }
        else tb = first_tile - 1;                    // warm-up tile: zero start, output discarded
    }
    const bool aligned = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    auto interior = [&](long tt) { const long lo = tt * FWR_TILE - G::HALO; return aligned && lo >= 0 && lo + G::SPAN <= n; };
    float4 pre[G::NPRE];
    bool have = false;
    auto prefetch = [&](long tt) {
        have = interior(tt);
        if (have) {
            const float4 *src = reinterpret_cast<const float4 *>(x + (tt * FWR_TILE - G::HALO));
#pragma unroll
            for (int u = 0; u < G::NPRE; u++) {
                const int idx = tid + 256 * u;
                pre[u] = src[idx < G::NF4 ? idx : G::NF4 - 1];
            }
        }
    };
    float ptl = 0.f, p1024 = 0.f, tp[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (IIR) {
        ptl = pr.ptab[lane];
        p1024 = pr.ptab[63];
#pragma unroll
        for (int l = 0; l < 6; l++) tp[l] = pr.ptab[(1 << l) - 1];
    }
    if (tb < t_end) prefetch(tb);
    for (long tt = tb; tt < t_end; tt++) {
        const bool emit = tt >= first_tile;
        if (have) {
#pragma unroll
            for (int u = 0; u < G::NPRE; u++) {
                const int idx = tid + 256 * u;
                if (idx < G::NF4) fwr_put4(ldsE, ldsO, 4 * idx, pre[u]);
            }
        } else {
            const long p0 = tt * FWR_TILE - G::HALO + (M - 1);          // stream position of tile coordinate 0
            for (int c = tid; c < G::SPAN; c += 256) {
                const float v = stream_at<1>(hist, x, p0 + c, 0, M, n);
                ldsE[fwr_phys(c)] = v;
                if (c) ldsO[fwr_phys(c - 1)] = v;
            }
        }
        __syncthreads();
        if (tt + 1 < t_end) prefetch(tt + 1);
        else have = false;

        cf acc[8];
        const bool active = !IIR || emit || wave >= 4 - pr.warm_waves;
        if (active) {
            fwr_taps<M, G::E0>(M >= FWR_SGPR_MIN_M ? pr.taps_rev : ldsT, ldsE + 18 * tid, ldsO + 18 * tid, acc);
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = cf{0.f, 0.f};
        }
        const long c0 = tt * FWR_TILE + 16L * tid;                      // first output of this lane
        if constexpr (!IIR) {
            float *yo = pr.y + c0;
            if (c0 + 16 <= n && (reinterpret_cast<uintptr_t>(pr.y) & 15) == 0) {
#pragma unroll
                for (int k = 0; k < 4; k++) nt_store(reinterpret_cast<float4 *>(yo) + k, make_float4(acc[2 * k].x, acc[2 * k].y, acc[2 * k + 1].x, acc[2 * k + 1].y));
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (c0 + 2 * i < n) yo[2 * i] = acc[i].x;
                    if (c0 + 2 * i + 1 < n) yo[2 * i + 1] = acc[i].y;
                }
            }
            __syncthreads();                                            // every wave is done reading the tile before it is overwritten
        } else {
            float v[16];
#pragma unroll
            for (int i = 0; i < 8; i++) { v[2 * i] = acc[i].x; v[2 * i + 1] = acc[i].y; }
            if (lane == 63) xch[wave] = v[15];
            __syncthreads();                                            // (also: every wave is done reading the tile)
            float vm1 = __shfl_up(v[15], 1);
            if (lane == 0) vm1 = wave ? xch[wave - 1] : vprev;
            const float vnext = xch[3];
            // feed-forward part, zero-state run over the chunk (the per-sample arithmetic of iir_stream_kernel)
            float u[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                float a = fmaf(pr.b0, v[i], 0.f);
                if (pr.nb > 1) a = fmaf(pr.b1, i ? v[i - 1] : vm1, a);
                u[i] = a;
            }
            float z = 0.f;
#pragma unroll
            for (int i = 0; i < 16; i++) z = fmaf(pr.na1, z, u[i]);
            // inclusive scan of the chunk end states inside the wave ...
#pragma unroll
            for (int l = 0; l < 6; l++) {
                const float prev = __shfl_up(z, 1 << l);
                if (lane >= (1 << l)) z = z + fmaf(tp[l], prev, 0.f);
            }
            if (lane == 63) xch[4 + wave] = z;
            __syncthreads();
            // ... and across the four waves: C = state entering the wave
            float C = carry, Cw = carry;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                if (w == wave) Cw = C;
                C = xch[4 + w] + fmaf(p1024, C, 0.f);
            }
            const float S = z + fmaf(ptl, Cw, 0.f);                     // true end state of this chunk
            float st = __shfl_up(S, 1);
            if (lane == 0) st = Cw;
            if (emit) {
                const long dec = pr.dec, dfirst = pr.dfirst;
                long k0 = 0, g0 = 0;
                if (dec > 1) {
                    k0 = c0 <= dfirst ? 0 : (c0 - dfirst + dec - 1) / dec;
                    g0 = dfirst + k0 * dec;
                }
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    st = fmaf(pr.na1, st, u[i]);
                    u[i] = st;
                    const long g = c0 + i;
                    // (a chunk that ends with the tile hands over C, the value the next tile of an uninterrupted run is given: below)
                    if (g == n - 1) { if (n != (tt + 1) * FWR_TILE) pr.state_out[0] = st; pr.vhist_out[0] = v[i]; }
                    if (dec > 1 && g == g0 && g < n) { pr.y[k0] = st; k0++; g0 += dec; }
                }
                if (dec == 1) {
                    float *yo = pr.y + c0;
                    if (c0 + 16 <= n && (reinterpret_cast<uintptr_t>(pr.y) & 15) == 0) {
#pragma unroll
                        for (int k = 0; k < 4; k++) nt_store(reinterpret_cast<float4 *>(yo) + k, make_float4(u[4 * k], u[4 * k + 1], u[4 * k + 2], u[4 * k + 3]));
                    } else {
#pragma unroll
                        for (int i = 0; i < 16; i++)
                            if (c0 + i < n) yo[i] = u[i];
                    }
                }
            }
            if (emit && tid == 0 && n == (tt + 1) * FWR_TILE) pr.state_out[0] = C;
            carry = C;
            vprev = vnext;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Short filters on the Float32 stream, D = 1 (the reference suite's "16 Real taps, Real input"): a streaming problem.  One thread = four
// consecutive outputs = one 16-byte store; its M + 3 input samples are HALO / 4 + 1 aligned 16-byte loads whose lines the neighbouring lanes
// share through L1; taps from scalar loads; one-shot grid.  The same fmaf chain per output as every other direct form (bit-exact).
// ------------------------------------------------------------------------------------------------------------
template <int M>
__global__ __launch_bounds__(256) void fir_short_real_kernel(const float *__restrict__ hist, const float *__restrict__ x, const float *__restrict__ taps_rev,
                                                             float *__restrict__ y, long n, float *__restrict__ hist_out)
{
    constexpr int HALO = ((M - 1 + 3) / 4) * 4, NV = HALO / 4 + 1;
    const long t = (long)blockIdx.x * 256 + threadIdx.x, n0 = 4 * t;
    if (hist_out && blockIdx.x == 0)
        for (int i = threadIdx.x; i < M - 1; i += 256) hist_out[i] = stream_at<1>(hist, x, n + i, 0, M, n);
    if (n0 >= n) return;
    float w[4 * NV];                                   // w[i] = x[n0 - HALO + i]
    const bool vec = n0 - HALO >= 0 && n0 + 4 <= n && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    if (vec) {
        const float4 *src = reinterpret_cast<const float4 *>(x + n0 - HALO);
#pragma unroll
        for (int q = 0; q < NV; q++) {
            const float4 v = src[q];
            w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4 * NV; i++) w[i] = stream_at<1>(hist, x, n0 - HALO + i + (M - 1), 0, M, n);
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < M; j++) {
        const float h = taps_rev[j];
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = fmaf(w[HALO - (M - 1) + i + j], h, acc[i]);      // s[n0 + i + j], s = [history | chunk]
    }
    if (n0 + 4 <= n && (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
        nt_store(reinterpret_cast<float4 *>(y + n0), make_float4(acc[0], acc[1], acc[2], acc[3]));
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (n0 + i < n) y[n0 + i] = acc[i];
    }
}

// ------------------------------------------------------------------------------------------------------------
// HilbertTransformBlock (radio/blocks/signal/hilberttransform.lua:107-124: out[i] = {delayed input, dot(state[i ..], reversed taps)}) on the window
// engine, for the tap counts the reference uses (65: its benchmark suite, 129: its examples).  A Hilbert transformer's taps at even distance from the
// centre are exactly zero (filter_utils.fir_hilbert_transform), so
//   * only the (M - 1) / 2 steps of one parity run - half the packed FMAs of the plain filter, and none of the Toeplitz product's structural zeros
//     (the matrix-core form of round 3, 0.21 ms for 2^26 samples, spent 3 of every 4 products on zeros);
//   * all operand pairs have the same parity, so ONE staged copy of the tile serves (C[i] = s[i + 1]);
//   * the lane's 16 outputs become (re, im) pairs = 128 consecutive bytes per lane, the mapping tools/mb_chunk.hip prices at 2.2 TB/s: the imaginary
//     parts go through an LDS out-area instead (lane stride 20 dwords, conflict-free both ways) and leave as lane-contiguous 16-byte stores, the
//     real parts are read back from the staged tile at the centre-tap offset.
// The chain is the oracle's fmaf chain without its zero-tap terms (those add +-0: equal bits for finite input).  The host checks the zero pattern.
// ------------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int fwh_ophys(int o) { return o + 4 * (o >> 4); }

template <int M>
struct FwhGeom {
    static_assert(M % 4 == 1, "centre tap at an even index: the non-zero reversed taps are the odd ones");
    static constexpr int HALO = M - 1, SPAN = FWR_TILE + HALO, NF4 = SPAN / 4, NPRE = (NF4 + 255) / 256;
    static constexpr int NZ = (M - 1) / 2;                          // non-zero taps: reversed index 2 k + 1
    static constexpr int CEN = (M - 1) / 2 - 1;                     // C index of output o's delayed input sample is o + CEN
    static constexpr int LDSC = fwr_phys(SPAN) + 16;
    static constexpr int LDSO = fwh_ophys(FWR_TILE);
    static constexpr int LDS_FLOATS = LDSC + LDSO + NZ;
};

template <int M>
__global__ __launch_bounds__(256, 4) void hilbert_win_kernel(const float *__restrict__ hist, const float *__restrict__ x, const float *__restrict__ taps_rev,
                                                             float *__restrict__ y, long n, long run, float *__restrict__ hist_out)
{
    using G = FwhGeom<M>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *ldsC = lds, *ldsOut = lds + G::LDSC, *ldsT = ldsOut + G::LDSO;
    const unsigned tid = threadIdx.x;
    if (hist_out && blockIdx.x == 0)
        for (int i = tid; i < M - 1; i += 256) hist_out[i] = stream_at<1>(hist, x, n + i, 0, M, n);
    if (!FW_TAPS_SGPR)
        for (int i = tid; i < G::NZ; i += 256) ldsT[i] = taps_rev[2 * i + 1];

    const long ntiles = (n + FWR_TILE - 1) / FWR_TILE;
    const long first_tile = (long)blockIdx.x * run;
    const long t_end = first_tile + run < ntiles ? first_tile + run : ntiles;
    const bool aligned = (reinterpret_cast<uintptr_t>(x) & 15) == 0, yal = (reinterpret_cast<uintptr_t>(y) & 15) == 0;
    auto interior = [&](long tt) { const long lo = tt * FWR_TILE - G::HALO; return aligned && lo >= 0 && lo + G::SPAN <= n; };
    // every address below is one per-lane base plus a compile-time constant (LDS: immediate offsets; global: a wave-uniform pointer plus the lane's
    // 32-bit offset), so no address lives in a vector register across the tap loop: with an address pair per access the kernel spilled (72 bytes of
    // scratch, 17 reloads per tile) at the 128 registers four workgroups per CU leave it
    constexpr int NFULL = G::NF4 / 256, NTAIL = G::NF4 - 256 * NFULL;       // full rounds of 256 16-byte loads, and the lanes of the last one
    static_assert(NTAIL > 0 && NTAIL < 256 && G::NPRE == NFULL + 1, "tile geometry");
    float4 pre[G::NPRE];
    bool have = false;
    auto prefetch = [&](long tt) {
        have = interior(tt);
        if (have) {
            const float4 *src = reinterpret_cast<const float4 *>(x + (tt * FWR_TILE - G::HALO));      // wave-uniform
#pragma unroll
            for (int u = 0; u < NFULL; u++) pre[u] = (src + 256 * u)[tid];
            if (tid < NTAIL) pre[NFULL] = (src + 256 * NFULL)[tid];
        }
    };
    float *const stC = ldsC + fwr_phys(4 * (int)tid), *const stC1 = ldsC + fwr_phys(4 * (int)tid - 1);      // staging: coordinates 4 tid (+ 1024 u) and the one before
    auto stage = [&](int u, float4 v) {
        constexpr int K = 1024 + 128;                                   // fwr_phys(a + 1024 u) = fwr_phys(a) + 1152 u
        if (u || tid) stC1[K * u] = v.x;
        *reinterpret_cast<float2 *>(stC + K * u) = make_float2(v.y, v.z);
        stC[K * u + 2] = v.w;
    };
    const float *const re0 = ldsC + fwr_phys(2 * (int)tid + G::CEN), *const re1 = ldsC + fwr_phys(2 * (int)tid + 1 + G::CEN);
    const float *const imp = ldsOut + fwh_ophys(2 * (int)tid);
    if (first_tile < t_end) prefetch(first_tile);
    for (long tt = first_tile; tt < t_end; tt++) {
        if (have) {
#pragma unroll
            for (int u = 0; u < NFULL; u++) stage(u, pre[u]);
            if (tid < NTAIL) stage(NFULL, pre[NFULL]);
        } else {
            const long p0 = tt * FWR_TILE - G::HALO + (M - 1);          // stream position of tile coordinate 0
            for (int c = tid + 1; c < G::SPAN; c += 256) ldsC[fwr_phys(c - 1)] = stream_at<1>(hist, x, p0 + c, 0, M, n);
        }
        __syncthreads();

        // ---- the tap loop of fwr_taps over the odd steps: step j = 2 k + 1 reads pair index m = i + k of the copy
        cf acc[8];
        {
            constexpr int R = 8, LA = 2, NS = R + LA;
            const float *base = ldsC + 18 * tid;
            cf W[NS];
            float4 T[2];
            auto ld = [&](int m) { return *reinterpret_cast<const cf *>(base + 2 * m + 2 * (m >> 3)); };
#pragma unroll
            for (int i = 0; i < R; i++) acc[i] = cf{0.f, 0.f};
            static_for<R - 1 + LA>([&](auto I) { constexpr int i = decltype(I)::value; W[i % NS] = ld(i); });
            if (!FW_TAPS_SGPR) T[0] = *reinterpret_cast<const float4 *>(ldsT);
            static_for<G::NZ>([&](auto K) {
                constexpr int k = decltype(K)::value;
                if constexpr (!FW_TAPS_SGPR && (k & 3) == 0 && k + 4 < G::NZ) T[((k >> 2) + 1) & 1] = *reinterpret_cast<const float4 *>(ldsT + k + 4);
                if constexpr (k + LA < G::NZ) W[(R - 1 + k + LA) % NS] = ld(R - 1 + k + LA);
                const float4 tq = T[(k >> 2) & 1];
                // scalar form: the pair (h[2k], h[2k+1]) of the full reversed tap list, the odd one is the high half
                const cf tp = FW_TAPS_SGPR ? uniform_load2(taps_rev + 2 * k) : (k & 2) ? cf{tq.z, tq.w} : cf{tq.x, tq.y};
                fw_step8<FW_TAPS_SGPR ? 1 : (k & 1), FW_TAPS_SGPR>(acc, tp, W[(0 + k) % NS], W[(1 + k) % NS], W[(2 + k) % NS], W[(3 + k) % NS], W[(4 + k) % NS], W[(5 + k) % NS],
                                  W[(6 + k) % NS], W[(7 + k) % NS]);
            });
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            *reinterpret_cast<float4 *>(ldsOut + 20 * tid + 4 * q) = make_float4(acc[2 * q].x, acc[2 * q].y, acc[2 * q + 1].x, acc[2 * q + 1].y);
        // the next tile's loads fly during the store phase only, not across the tap loop (20 registers)
        if (tt + 1 < t_end) prefetch(tt + 1);
        else have = false;
        __syncthreads();
        const long ob = tt * FWR_TILE;
        if (ob + FWR_TILE <= n && yal) {
            f32x4 *dst = reinterpret_cast<f32x4 *>(y + 2 * ob);         // wave-uniform
#pragma unroll
            for (int u = 0; u < FWR_TILE / 512; u++) {
                // outputs o = 2 tid + 512 u, o + 1:  fwr_phys(o + c) = fwr_phys(2 tid + c) + 576 u, fwh_ophys(o) = fwh_ophys(2 tid) + 640 u
                const float2 im = *reinterpret_cast<const float2 *>(imp + 640 * u);
                const f32x4 v = {re0[576 * u], im.x, re1[576 * u], im.y};
                __builtin_nontemporal_store(v, (dst + 256 * u) + tid);
            }
        } else {
            for (int o = tid; o < FWR_TILE && ob + o < n; o += 256)
                *reinterpret_cast<float2 *>(y + 2 * (ob + o)) = make_float2(ldsC[fwr_phys(o + G::CEN)], ldsOut[fwh_ophys(o)]);
        }
        __syncthreads();                                                // the tile and the out-area are read: the next tile may be staged
    }
}

}  // namespace lrhip
