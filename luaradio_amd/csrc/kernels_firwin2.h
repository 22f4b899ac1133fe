// kernels_firwin2.h - register-window FIR (kernels_firwin.h) for the ComplexFloat32 stream with decimation:
// Decimator / Tuner (+ FrequencyDiscriminator) and, with two halves of a Float32 stream riding as (re, im), a decimating
// Float32 filter with a fused first-order recurrence.
//
//   fir_win_cplx_kernel<D, R, M, MODE>      a lane owns R consecutive decimated outputs, 256 R per tile
//     MODE & FWC_ROT   FrequencyTranslatorBlock in front (frequencytranslator.lua:93-110): samples are rotated on their way into LDS
//                      with the block-of-8 phasors of kernels_elem.h - the bits of the standalone rotator
//     MODE & FWC_DISC  FrequencyDiscriminatorBlock behind (frequencydiscriminator.lua:68-88): the ComplexFloat32 outputs go to an
//                      LDS out-area, every thread then turns R of them (lane stride 1) into Float32 angles.  Tiles overlap by OV = 8
//                      outputs, so the output in front of a tile's first one is recomputed locally (same fmaf chain, same bits):
//                      no edge buffer, no fix-up launch
//     MODE & FWC_PAIR  Float32 stream: packed pair = (s[c], s[c + D * 256 R]), i.e. the tile's two halves are filtered as the two
//                      components of one "complex" stream.  With FWC_IIR the first-order recurrence y[k] = b0 v[k] - a1 y[k-1]
//                      (IIRFilterBlock with one feed-forward tap) runs on the accumulators: chunk end states scanned inside the
//                      wave (DPP shuffles) and across the eight half-waves through LDS, as in fir_win_real_kernel
// Tap loop: step j reads ONE new 8-byte window sample per lane (coordinate D (R-1) + j + LA) and issues R packed FMAs on the ring
// of D (R-1) + 1 + LA samples held in registers.  R is chosen so that the lane stride D R (window samples) is odd: the 32 lanes of
// a ds_read_b64 group then hit 32 distinct bank pairs without any padding, the window in LDS is a straight copy of the stream and
// staging is one aligned 16-byte LDS write per loaded float4 (no index arithmetic, no bounds checks: guard samples absorb the
// alignment slack and the rotator's block overhang).  Everything around the tap loop is VALU time the filter cannot hide
// (rocprofv3 on the first cut, R = 6 with padded rows: 1 100 of 1 870 VALU instructions per wave and tile were staging and epilogue).
#pragma once
#include "kernels_firwin.h"

namespace lrhip {

enum { FWC_ROT = 1, FWC_DISC = 2, FWC_PAIR = 4, FWC_IIR = 8, FWC_CTAPS = 16 };      // CTAPS: ComplexFloat32 taps (taps_rev = (re, im, -im, re) per tap)
#ifndef FWC_LA
#define FWC_LA 6
#endif

template <int D, int R, int M, int MODE>
struct FwcGeom {
    static constexpr bool ROT = (MODE & FWC_ROT) != 0, DISC = (MODE & FWC_DISC) != 0, PAIR = (MODE & FWC_PAIR) != 0, IIR = (MODE & FWC_IIR) != 0;
    static constexpr bool CTAPS = (MODE & FWC_CTAPS) != 0;
    static constexpr int TF = CTAPS ? 4 : 1;                   // floats per tap in LDS
    static_assert(!(PAIR && (ROT || DISC)) && (!IIR || PAIR) && !(CTAPS && (PAIR || ROT || DISC)), "mode combination");
    static_assert(M % 4 == 0, "taps are read four at a time");
    static constexpr int DR = D * R, PADS = (DR & 1) ? 0 : 1, LS = DR + PADS;      // lane stride in window samples (odd)
    static_assert(PADS == 0, "choose R so that D R is odd: the window then needs no padding and staging is a straight 16-byte copy");
    static constexpr int TO = 256 * R;                         // window outputs per tile (pair mode: per half)
    static constexpr int OV = DISC ? 8 : 0;                    // outputs recomputed in front of a tile
    static constexpr int TA = TO - OV;                         // tile advance in outputs
    static constexpr int SPAN = D * (TO - 1) + M;              // window samples per tile
    // staging: float4 loads.  Complex: 2 samples per float4, e <= 1 samples of slack in front; pair: 4 floats per float4 and source
    // half, e <= 3.  The staged float4 i lands at LDS sample GUARD + SPF * i, so window coordinate c sits at GUARD + e + c.
    static constexpr int SPF = PAIR ? 4 : 2;                   // window samples per staged float4
    static constexpr int NF4 = PAIR ? (SPAN + 3 + 3) / 4 : (SPAN + 1 + 1) / 2;
    static constexpr int NB = NF4 / 4 + 2;                     // aligned blocks of 8 samples that can touch the window (ROT)
    static constexpr int UB = (NB + 255) / 256;
    static constexpr int NPRE = ROT ? 4 * UB : PAIR ? 2 * ((NF4 + 255) / 256) : (NF4 + 255) / 256;
    static constexpr int GUARD = 8;                            // samples in front of staged float4 0 (a rotator block may start up to 3 float4 early)
    static constexpr int XN = GUARD + SPF * (NF4 + 4) + 8;     // window samples in LDS (8 B each)
    static constexpr int ON = PAIR ? 0 : TO + 2;               // out-area samples (complex modes)
    static constexpr int LDS_FLOATS = 2 * XN + 2 * ON + 32 + TF * M;
};

struct FwcParams {
    const float *hist;           // M-1 input samples before x[0]
    const float *x;
    long n;                      // input samples (ComplexFloat32, or Float32 in pair mode)
    const float *taps_rev;
    float *y;
    long n_out, first;           // outputs of this chunk; stream position of output 0 (the carried downsampler index)
    float *hist_out;
    long ntiles;
    uint64_t rot_step_fx, rot_count0;
    const float2 *prev_in;       // discriminator: the output before the chunk
    float2 *prev_out;
    double inv_gain;
    // pair mode + recurrence
    float b0, na1, na1_lo;       // y[k] = b0 v[k] + (na1 + na1_lo) y[k-1]: the pole as a Float32 pair (see fuse_iir1)
    const float *ptab;           // ptab[l] = (-a1)^(R (l+1)), l < 64
    const float *state_in;
    float *state_out;
    int warm_waves;
    long run;                    // tiles per workgroup (contiguous; recurrence mode)
    // pair mode behind the Toeplitz tuner + discriminator kernel (fir_mfma_persistent_kernel, EPI = 1): that kernel leaves the first output of
    // every wave's 256-output range to be fixed from its edge records (disc_epilogue).  Instead of a fix-up launch, the consumer patches those
    // samples as it stages them: x[256 w] = discriminate(edge[2w], w ? edge[2(w-1)+1] : *fix_prev).  Null = nothing to fix.
    const float2 *fix_edge;      // ALWAYS a readable address (the kernel loads from it unconditionally, see prefetch()); fix_on says whether it means anything
    const float2 *fix_prev;
    double fix_inv_gain;
    int fix_on;
    int fix_shift;               // a record pair per 2^fix_shift outputs of the producer (8: per wave, 10: per tile)
};

// sample x[g] of a Float32 stream whose wave-first samples are still to be fixed (see FwcParams::fix_edge)
__device__ __forceinline__ float fwc_fix_value(const FwcParams &pr, long g)
{
    const long w = g >> pr.fix_shift;
    const float2 e0 = pr.fix_edge[2 * w], ep = w ? pr.fix_edge[2 * (w - 1) + 1] : *pr.fix_prev;
    return discriminate(e0, ep, pr.fix_inv_gain);
}
__device__ __forceinline__ float fwc_stream_at_fixed(const FwcParams &pr, const float *__restrict__ hist, const float *__restrict__ x, long p, int M, long n)
{
    const long g = p - (M - 1);
    if (pr.fix_on && g >= 0 && g < n && (g & ((1L << pr.fix_shift) - 1)) == 0) return fwc_fix_value(pr, g);
    return stream_at<1>(hist, x, p, 0, M, n);
}

template <int HI, bool SG = false>
__device__ __forceinline__ void fw_step5(cf (&a)[5], cf t, cf w0, cf w1, cf w2, cf w3, cf w4)
{
    if constexpr (SG) {
        if constexpr (HI == 0)
            asm("v_pk_fma_f32 %0, %5, %6, %0 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %1, %5, %7, %1 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %2, %5, %8, %2 op_sel_hi:[0,1,1]\n\t"
                "v_pk_fma_f32 %3, %5, %9, %3 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %4, %5, %10, %4 op_sel_hi:[0,1,1]"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4])
                : "s"(t), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4));
        else
            asm("v_pk_fma_f32 %0, %5, %6, %0 op_sel:[1,0,0]\n\tv_pk_fma_f32 %1, %5, %7, %1 op_sel:[1,0,0]\n\tv_pk_fma_f32 %2, %5, %8, %2 op_sel:[1,0,0]\n\t"
                "v_pk_fma_f32 %3, %5, %9, %3 op_sel:[1,0,0]\n\tv_pk_fma_f32 %4, %5, %10, %4 op_sel:[1,0,0]"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4])
                : "s"(t), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4));
    } else {
        if constexpr (HI == 0)
            asm("v_pk_fma_f32 %0, %5, %6, %0 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %1, %5, %7, %1 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %2, %5, %8, %2 op_sel_hi:[0,1,1]\n\t"
                "v_pk_fma_f32 %3, %5, %9, %3 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %4, %5, %10, %4 op_sel_hi:[0,1,1]"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4])
                : "v"(t), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4));
        else
            asm("v_pk_fma_f32 %0, %5, %6, %0 op_sel:[1,0,0]\n\tv_pk_fma_f32 %1, %5, %7, %1 op_sel:[1,0,0]\n\tv_pk_fma_f32 %2, %5, %8, %2 op_sel:[1,0,0]\n\t"
                "v_pk_fma_f32 %3, %5, %9, %3 op_sel:[1,0,0]\n\tv_pk_fma_f32 %4, %5, %10, %4 op_sel:[1,0,0]"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4])
                : "v"(t), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4));
    }
}

// the tap loop over one staged tile: window coordinate r of the lane sits at base + 2 r floats (base = window + 2 D R L)
template <int D, int R, int M>
__device__ __forceinline__ void fwc_taps(const float *ldsT, const float *base, cf (&acc)[R])
{
    static_assert(R == 5, "accumulators per lane");
    constexpr int LA = FWC_LA, C = D * (R - 1) + 1 + LA;      // LA: window samples requested ahead of their first use (LDS latency / one step of R FMAs)
    cf W[C];
    float4 T[2];
    auto ld = [&](int r) { return *reinterpret_cast<const cf *>(base + 2 * r); };
#pragma unroll
    for (int i = 0; i < R; i++) acc[i] = cf{0.f, 0.f};
    static_for<D *(R - 1) + LA>([&](auto I) { constexpr int r = decltype(I)::value; W[r % C] = ld(r); });
    T[0] = *reinterpret_cast<const float4 *>(ldsT);
    static_for<M>([&](auto J) {
        constexpr int j = decltype(J)::value, rn = D * (R - 1) + j + LA;
        if constexpr ((j & 3) == 0 && j + 4 < M) T[((j >> 2) + 1) & 1] = *reinterpret_cast<const float4 *>(ldsT + j + 4);
        if constexpr (rn <= D * (R - 1) + M - 1) W[rn % C] = ld(rn);
        const float4 tq = T[(j >> 2) & 1];
        const cf tp = (j & 2) ? cf{tq.z, tq.w} : cf{tq.x, tq.y};
        fw_step5<(j & 1)>(acc, tp, W[j % C], W[(D + j) % C], W[(2 * D + j) % C], W[(3 * D + j) % C], W[(4 * D + j) % C]);
    });
}

// complex taps: acc_i += w_i.re * (h.re, h.im);  acc_i += w_i.im * (-h.im, h.re) - per component the order of fir_direct_kernel<2>
// (re += xr hr; re += xi (-hi); im += xr hi; im += xi hr), so the bits are those of the other complex-taps paths
__device__ __forceinline__ void fw_cstep5(cf (&a)[5], cf tlo, cf thi, cf w0, cf w1, cf w2, cf w3, cf w4)
{
    asm("v_pk_fma_f32 %0, %7, %5, %0 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %1, %8, %5, %1 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %2, %9, %5, %2 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %3, %10, %5, %3 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %4, %11, %5, %4 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %7, %6, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\tv_pk_fma_f32 %1, %8, %6, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %2, %9, %6, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\tv_pk_fma_f32 %3, %10, %6, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %4, %11, %6, %4 op_sel:[1,0,0] op_sel_hi:[1,1,1]"
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4])
        : "v"(tlo), "v"(thi), "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4));
}

template <int D, int R, int M>
__device__ __forceinline__ void fwc_ctaps(const float *ldsT, const float *base, cf (&acc)[R])
{
    static_assert(R == 5, "accumulators per lane");
    constexpr int LA = FWC_LA, C = D * (R - 1) + 1 + LA;
    cf W[C];
    float4 T[2];
    auto ld = [&](int r) { return *reinterpret_cast<const cf *>(base + 2 * r); };
#pragma unroll
    for (int i = 0; i < R; i++) acc[i] = cf{0.f, 0.f};
    static_for<D *(R - 1) + LA>([&](auto I) { constexpr int r = decltype(I)::value; W[r % C] = ld(r); });
    T[0] = *reinterpret_cast<const float4 *>(ldsT);
    static_for<M>([&](auto J) {
        constexpr int j = decltype(J)::value, rn = D * (R - 1) + j + LA;
        if constexpr (j + 1 < M) T[(j + 1) & 1] = *reinterpret_cast<const float4 *>(ldsT + 4 * (j + 1));
        if constexpr (rn <= D * (R - 1) + M - 1) W[rn % C] = ld(rn);
        const float4 tq = T[j & 1];
        fw_cstep5(acc, cf{tq.x, tq.y}, cf{tq.z, tq.w}, W[j % C], W[(D + j) % C], W[(2 * D + j) % C], W[(3 * D + j) % C], W[(4 * D + j) % C]);
    });
}

__device__ __forceinline__ float4 f4(f32x4 v) { return make_float4(v[0], v[1], v[2], v[3]); }

#ifndef FWC_PAIR_WAVES
#define FWC_PAIR_WAVES 2      /* A/B: resident workgroups per CU the pair-mode instantiation is register-budgeted for */
#endif
template <int D, int R, int M, int MODE>
__global__ __launch_bounds__(256, (MODE & FWC_PAIR) ? FWC_PAIR_WAVES : 2) void fir_win_cplx_kernel(const FwcParams pr)
{
    using G = FwcGeom<D, R, M, MODE>;
    constexpr bool ROT = G::ROT, DISC = G::DISC, PAIR = G::PAIR, IIR = G::IIR;
    constexpr int S = PAIR ? 1 : 2;                                  // floats per input sample
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *ldsX = lds;                                               // window samples, 2 floats each
    float *ldsOut = lds + 2 * G::XN;                                 // complex modes: the tile's outputs
    float *xch = ldsOut + 2 * G::ON;                                 // scan totals of the eight half-waves
    float *ldsT = xch + 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long n = pr.n, n_out = pr.n_out;
    const float *__restrict__ x = pr.x;
    const float *__restrict__ hist = pr.hist;
    if (pr.hist_out && blockIdx.x == 0)
        for (int i = tid; i < (M - 1) * S; i += 256)
            pr.hist_out[i] = PAIR ? fwc_stream_at_fixed(pr, hist, x, n + i, M, n) : stream_at<S>(hist, x, n + i / S, i % S, M, n);
    for (int i = tid; i < G::TF * M; i += 256) ldsT[i] = pr.taps_rev[i];

    // tile t: first window output kb(t) (negative for tile 0 with the discriminator overlap), stream position of window coordinate 0
    auto kb_of = [&](long t) { return PAIR ? t * 2L * G::TO : t * (long)G::TA - G::OV; };
    auto qb_of = [&](long t) { return pr.first + kb_of(t) * D; };
    // alignment slack: the staged float4s start e samples in front of window coordinate 0 so that their global address is 16-B aligned
    constexpr int EMASK = PAIR ? 3 : 1;
    const long samp0 = (long)(reinterpret_cast<uintptr_t>(x) / (4 * S));
    const bool xal = (reinterpret_cast<uintptr_t>(x) % (4 * S)) == 0;
    auto e_of = [&](long t) { return (int)((samp0 + qb_of(t) - (M - 1)) & EMASK); };
    auto interior = [&](long t) {
        const long lo = qb_of(t) - (M - 1) - e_of(t);
        const long hi = lo + (long)G::SPF * G::NF4 + (PAIR ? (long)D * G::TO : 0L);
        return xal && t < pr.ntiles && lo >= 0 && hi <= n;
    };

    RotTab rot_t;
    if constexpr (ROT) rot_t = rot_tab(pr.rot_step_fx);
    f32x4 pre[G::NPRE];                                              // (ext_vector type: usable as an inline-asm operand)
    bool have = false, prot_blocks = false;
    int pe = 0, pa4 = 0;                                             // of the prefetched tile
    int fpos = -1, fhalf = 0;                                        // pair mode with fix_edge: the staged sample this thread patches, and its records
    cf fe0 = cf{0.f, 0.f}, fep = cf{0.f, 0.f};
    // first tile this workgroup emits (recurrence mode: the tile before it is the warm-up tile)
    const long first_emit = IIR ? (long)blockIdx.x * pr.run : 0;
    auto prefetch = [&](long t) {
        have = interior(t);
        pe = 0;
        if (!have) return;
        pe = e_of(t);
        const long lo = qb_of(t) - (M - 1) - pe;                      // x index of staged float4 0
        if constexpr (PAIR) {
            const f32x4 *sa = reinterpret_cast<const f32x4 *>(x + lo), *sb = reinterpret_cast<const f32x4 *>(x + lo + (long)D * G::TO);
            // warm-up tile: only the last warm_waves waves of half B run, and they read the window from float4 D R 64 (4 - warm_waves) / 4 on.  Everything
            // in front of it (and all of half A) is loaded from that float4 instead - finite samples of the right magnitude, which is all the discarded
            // lanes need - so that a warm-up tile re-reads an eighth of a tile from HBM instead of a whole one (the chain's traffic: 1.26x -> 1.05x algorithmic)
            int lo4 = 0;
            if (IIR && t < first_emit) { lo4 = (D * R * 64 / 4) * (4 - pr.warm_waves); sa = sb; }
#pragma unroll
            for (int u = 0; u < G::NPRE / 2; u++) {
                int idx = tid + 256 * u;
                idx = idx < G::NF4 ? idx : G::NF4 - 1;
                idx = idx < lo4 ? lo4 : idx;
                pre[2 * u] = sa[idx];
                pre[2 * u + 1] = sb[idx];
            }
            // the edge records of the (at most 26 per half) wave-first samples in this window: threads 0..63, one sample each.  The two loads
            // are UNCONDITIONAL (clamped address) and their registers are touched only at the staging point, like pre[]: behind a branch the
            // compiler merges them with the "no fix" values right here and waits for the whole prefetch before the filter starts (+18 us per step)
            {
                fhalf = (tid >> 5) & 1;
                const long lo_h = lo + (fhalf ? (long)D * G::TO : 0L);
                const int sh = pr.fix_shift;
                const long g = ((lo_h <= 0 ? 0 : (lo_h + (1L << sh) - 1) >> sh) + (tid & 31)) << sh;
                const bool ok = pr.fix_on && tid < 64 && g < lo_h + 4L * G::NF4 && g < n;
                const long w = ok ? g >> sh : 0;
                fpos = ok ? (int)(g - lo_h) : -1;
                fe0 = *reinterpret_cast<const cf *>(pr.fix_edge + 2 * w);
                fep = *reinterpret_cast<const cf *>(w ? pr.fix_edge + 2 * (w - 1) + 1 : pr.fix_prev);
            }
        } else if constexpr (ROT) {
            // a thread owns whole ALIGNED blocks of 8 samples (absolute index = 0 mod 8): one phasor polynomial serves 8 samples
            const uint64_t abs0 = pr.rot_count0 + (uint64_t)lo;
            prot_blocks = (abs0 & 1) == 0;
            pa4 = prot_blocks ? (int)((abs0 & 7) >> 1) : 0;
            const f32x4 *src = reinterpret_cast<const f32x4 *>(x + 2 * lo);
#pragma unroll
            for (int u = 0; u < G::NPRE; u++) {
                const int idx = 4 * (tid + 256 * (u >> 2)) + (u & 3) - pa4;
                pre[u] = src[idx < 0 ? 0 : idx < G::NF4 ? idx : G::NF4 - 1];
            }
        } else {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(x + 2 * lo);
#pragma unroll
            for (int u = 0; u < G::NPRE; u++) {
                const int idx = tid + 256 * u;
                pre[u] = src[idx < G::NF4 ? idx : G::NF4 - 1];
            }
        }
    };
    // staged float4 i (complex: 2 window samples; pair: half of a quad) -> LDS, 16-byte aligned by construction
    float4 *ldsX4 = reinterpret_cast<float4 *>(ldsX + 2 * G::GUARD);

    // tiles of this workgroup: a contiguous run with the recurrence (its state is carried from tile to tile), strided otherwise
    long t0, t1, tstep;
    float carry = 0.f;
    float ptl = 0.f, pw64 = 0.f, tp[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (IIR) {
        const long ft = (long)blockIdx.x * pr.run;
        t0 = ft;
        t1 = ft + pr.run < pr.ntiles ? ft + pr.run : pr.ntiles;
        tstep = 1;
        if (ft == 0) carry = pr.state_in[0];
        else t0 = ft - 1;                                            // warm-up tile: zero start, output discarded
        ptl = pr.ptab[lane];
        pw64 = pr.ptab[63];
#pragma unroll
        for (int l = 0; l < 6; l++) tp[l] = pr.ptab[(1 << l) - 1];
    } else {
        t0 = blockIdx.x; t1 = pr.ntiles; tstep = gridDim.x;
    }

    if (t0 < t1) prefetch(t0);
    for (long t = t0; t < t1; t += tstep) {
        const long kb = kb_of(t), qb = qb_of(t);
        const int ce = pe;                                           // slack of THIS tile (prefetch() below moves on to the next)
        // ---- stage the window
        if (have) {
            // the loaded registers are touched here and not earlier: hipcc would otherwise hoist the shuffles of the staging code up
            // to the loads and wait for HBM right after issuing them
#pragma unroll
            for (int u = 0; u < G::NPRE; u++) asm volatile("" : "+v"(pre[u]));
            if constexpr (PAIR) {
#pragma unroll
                for (int u = 0; u < G::NPRE / 2; u++) {
                    const int idx = tid + 256 * u;
                    if (idx < G::NF4) {
                        const f32x4 a = pre[2 * u], b = pre[2 * u + 1];
                        ldsX4[2 * idx] = make_float4(a.x, b.x, a.y, b.y);
                        ldsX4[2 * idx + 1] = make_float4(a.z, b.z, a.w, b.w);
                    }
                }
                if (pr.fix_on) {                                      // kernel-uniform
                    asm volatile("" : "+v"(fe0), "+v"(fep));
                    __syncthreads();                                  // the unfixed values are in place: patch over them
                    if (fpos >= 0) ldsX[2 * (G::GUARD + fpos) + fhalf] = discriminate(make_float2(fe0.x, fe0.y), make_float2(fep.x, fep.y), pr.fix_inv_gain);
                }
            } else if constexpr (ROT) {
                const long lo = qb - (M - 1) - ce;
#pragma unroll
                for (int v = 0; v < G::UB; v++) {
                    const int i40 = 4 * (tid + 256 * v) - pa4;        // first float4 of this thread's block (>= -3: the guard in front)
                    if (i40 < G::NF4) {
                        if (!prot_blocks) {
#pragma unroll
                            for (int j = 0; j < 4; j++)
                                ldsX4[i40 + j] = rotate_pair(f4(pre[4 * v + j]), pr.rot_step_fx, pr.rot_count0 + (uint64_t)(lo + 2 * (long)(i40 + j)), rot_t);
                        } else {
                            const cf p = phasor_poly(pr.rot_step_fx * (pr.rot_count0 + (uint64_t)(lo + 2 * (long)i40)));
                            ldsX4[i40] = rotate_in_block<0>(f4(pre[4 * v]), p, rot_t);
                            ldsX4[i40 + 1] = rotate_in_block<1>(f4(pre[4 * v + 1]), p, rot_t);
                            ldsX4[i40 + 2] = rotate_in_block<2>(f4(pre[4 * v + 2]), p, rot_t);
                            ldsX4[i40 + 3] = rotate_in_block<3>(f4(pre[4 * v + 3]), p, rot_t);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < G::NPRE; u++) {
                    const int idx = tid + 256 * u;
                    if (idx < G::NF4) ldsX4[idx] = f4(pre[u]);
                }
            }
        } else {
            // edge tiles (carried history, end of the chunk, unaligned source): one window sample at a time, slack 0
            for (int c = tid; c < G::SPAN; c += 256) {
                float a, b;
                if constexpr (PAIR) {
                    a = stream_at<1>(hist, x, qb + c, 0, M, n);          // (wave-first samples are patched below, not here: a dependent
                    b = stream_at<1>(hist, x, qb + (long)D * G::TO + c, 0, M, n);      // load inside this loop costs ~2 us per hit, 13 hits per wave)
                } else {
                    a = stream_at<2>(hist, x, qb + c, 0, M, n);
                    b = stream_at<2>(hist, x, qb + c, 1, M, n);
                    if constexpr (ROT) {
                        const float2 o = rotate_sample(make_float2(a, b), pr.rot_step_fx, pr.rot_count0 + (uint64_t)(qb + c - (M - 1)));
                        a = o.x; b = o.y;
                    }
                }
                *reinterpret_cast<float2 *>(ldsX + 2 * (G::GUARD + c)) = make_float2(a, b);
            }
            if constexpr (PAIR) {
                if (pr.fix_on) {                                      // the same patch as on the prefetched path, records loaded here
                    __syncthreads();
                    const int h = (tid >> 5) & 1;
                    const long lo_h = qb - (M - 1) + (h ? (long)D * G::TO : 0L);       // x index of window coordinate 0 of this half (slack 0)
                    const int sh = pr.fix_shift;
                    const long g = ((lo_h <= 0 ? 0 : (lo_h + (1L << sh) - 1) >> sh) + (tid & 31)) << sh;
                    if (tid < 64 && g < lo_h + G::SPAN && g < n) ldsX[2 * (G::GUARD + (int)(g - lo_h)) + h] = fwc_fix_value(pr, g);
                }
            }
        }
        __syncthreads();
        if (t + tstep < t1) prefetch(t + tstep);
        else { have = false; pe = 0; }

        // ---- filter
        cf acc[R];
        const bool emit = !IIR || t >= first_emit;
        const bool active = emit || wave >= 4 - pr.warm_waves;          // warm-up tile: only the last waves (of half B) matter
        if (active) {
            if constexpr (G::CTAPS) fwc_ctaps<D, R, M>(ldsT, ldsX + 2 * (G::GUARD + ce + G::DR * tid), acc);
            else fwc_taps<D, R, M>(ldsT, ldsX + 2 * (G::GUARD + ce + G::DR * tid), acc);
        } else {
#pragma unroll
            for (int i = 0; i < R; i++) acc[i] = cf{0.f, 0.f};
        }

        // ---- epilogue
        if constexpr (!PAIR) {
            // outputs -> LDS out-area (lane L owns local outputs R L .. R L + R - 1), then coalesced stores
#pragma unroll
            for (int i = 0; i < R; i++) {
                cf o = acc[i];
                if constexpr (DISC) {
                    if (kb + R * tid + i == -1) { const float2 pv = *pr.prev_in; o = cf{pv.x, pv.y}; }      // the output before the chunk
                    if (kb + R * tid + i == n_out - 1) *pr.prev_out = make_float2(o.x, o.y);
                }
                *reinterpret_cast<float2 *>(ldsOut + 2 * (R * tid + i)) = make_float2(o.x, o.y);
            }
            __syncthreads();          // (also: every wave is done reading the window)
            if constexpr (DISC) {
                // local output l = tid + 256 i (consecutive lanes, consecutive outputs): global index k = t TA + l - OV
#pragma unroll
                for (int i = 0; i < R; i++) {
                    const int l = tid + 256 * i;
                    const long k = t * (long)G::TA + l - G::OV;
                    if (l >= G::OV && k < n_out) {
                        const float2 cur = *reinterpret_cast<const float2 *>(ldsOut + 2 * l), prv = *reinterpret_cast<const float2 *>(ldsOut + 2 * (l - 1));
                        pr.y[k] = discriminate(cur, prv, pr.inv_gain);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < (R + 1) / 2; i++) {
                    const int m = tid + 256 * i;                    // outputs 2m, 2m + 1
                    const long k = kb + 2 * m;
                    if (2 * m < G::TO && k < n_out) {
                        const float4 o = *reinterpret_cast<const float4 *>(ldsOut + 4 * m);
                        if (k + 1 < n_out && (reinterpret_cast<uintptr_t>(pr.y) & 15) == 0) {
                            nt_store(reinterpret_cast<float4 *>(pr.y + 2 * k), o);
                        } else {
                            *reinterpret_cast<float2 *>(pr.y + 2 * k) = make_float2(o.x, o.y);
                            if (k + 1 < n_out) *reinterpret_cast<float2 *>(pr.y + 2 * k + 2) = make_float2(o.z, o.w);
                        }
                    }
                }
            }
            // the next tile's out-area writes come after its staging barrier: no third barrier needed
        } else {
            const long ka = kb + (long)R * tid, kbb = ka + G::TO;       // first output of the lane in half A / half B
            if constexpr (!IIR) {
#pragma unroll
                for (int i = 0; i < R; i++) {
                    if (ka + i < n_out) pr.y[ka + i] = acc[i].x;
                    if (kbb + i < n_out) pr.y[kbb + i] = acc[i].y;
                }
                __syncthreads();      // every wave is done reading the window before it is overwritten
            } else {
                // y[k] = b0 v[k] + p y[k-1], p = -a1: zero-state run over the lane's chunk of each half ...
                cf u[R], z = cf{0.f, 0.f};
                const cf pp = cf{pr.na1, pr.na1}, pl = cf{pr.na1_lo, pr.na1_lo}, bb = cf{pr.b0, pr.b0};
#pragma unroll
                for (int i = 0; i < R; i++) {
                    u[i] = __builtin_elementwise_fma(bb, acc[i], cf{0.f, 0.f});
                    z = __builtin_elementwise_fma(pp, z, __builtin_elementwise_fma(pl, z, u[i]));
                }
                // ... inclusive scan of the chunk end states inside the wave (both halves at once) ...
#pragma unroll
                for (int l = 0; l < 6; l++) {
                    const cf prev = cf{__shfl_up(z.x, 1 << l), __shfl_up(z.y, 1 << l)};
                    if (lane >= (1 << l)) z = z + __builtin_elementwise_fma(cf{tp[l], tp[l]}, prev, cf{0.f, 0.f});
                }
                if (lane == 63) { xch[wave] = z.x; xch[4 + wave] = z.y; }
                __syncthreads();      // (also: every wave is done reading the window)
                // ... and across the eight half-waves in stream order A0..A3, B0..B3: C = state entering the half-wave
                float C = carry, Ca = carry, Cb = carry;
#pragma unroll
                for (int w = 0; w < 8; w++) {
                    if (w == wave) Ca = C;
                    if (w == 4 + wave) Cb = C;
                    C = xch[w] + fmaf(pw64, C, 0.f);
                }
                const cf Sx = z + __builtin_elementwise_fma(cf{ptl, ptl}, cf{Ca, Cb}, cf{0.f, 0.f});      // true end states of the two chunks
                cf st = cf{__shfl_up(Sx.x, 1), __shfl_up(Sx.y, 1)};
                if (lane == 0) st = cf{Ca, Cb};
                if (emit) {
#pragma unroll
                    for (int i = 0; i < R; i++) {
                        st = __builtin_elementwise_fma(pp, st, __builtin_elementwise_fma(pl, st, u[i]));
                        if (ka + i < n_out) pr.y[ka + i] = st.x;
                        if (kbb + i < n_out) pr.y[kbb + i] = st.y;
                        // the carried state of a chunk that ends inside this tile; a chunk that ends WITH the tile hands over C below -
                        // the value the next tile of an uninterrupted run would have been given (the scan and the re-run round differently)
                        if (ka + i == n_out - 1 && n_out != kb + 2L * G::TO) pr.state_out[0] = st.x;
                        if (kbb + i == n_out - 1 && n_out != kb + 2L * G::TO) pr.state_out[0] = st.y;
                    }
                    if (tid == 0 && n_out == kb + 2L * G::TO) pr.state_out[0] = C;
                }
                carry = C;
            }
        }
    }
}

}  // namespace lrhip
