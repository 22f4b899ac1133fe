// kernels_rx.h - the FM receiver's device path as ONE launch (examples/rtlsdr_wbfm_mono.lua:12-17 collapsed by lrhip_chain_create):
//
//     FrequencyTranslator -> LowpassFilter(128) -> Downsampler(5) -> FrequencyDiscriminator          (the tuner tile: Toeplitz MFMA, kernels_fir.h)
//       -> LowpassFilter(128) -> FMDeemphasis -> Downsampler(5)                                      (the audio tail: ONE decimating filter of 136 taps
//                                                                                                    + the recurrence at the low rate)
//
// Round 2 ran this as two launches with the 220.5 kHz discriminator stream crossing HBM both ways (107 of the 673 MB moved per 2^26
// samples).  Here the discriminator output of a tuner tile never leaves the workgroup:
//
//   * a workgroup owns a contiguous RUN of tuner tiles (512 discriminator samples each: one accumulator per wave).  Ten tiles are a BATCH:
//     5 120 discriminator samples = 1 024 audio samples.  The discriminator epilogue (in registers, disc_epilogue of kernels_fir.h) writes its
//     angles straight into the batch window P in LDS, in the padded-row layout of the Float32 Toeplitz product (FirMfmaGeom<1, 5>);
//     the first output of every wave needs the previous wave's last filter output - the waves leave it in LDS in front of the barrier that
//     frees the RF window, and lane 0 picks it up behind it, before the angles are computed.  No edge records in HBM, no fix-up launch, nothing for a next kernel to patch;
//   * after the tenth tile the 136-tap decimating filter runs on P as ONE MORE Toeplitz product on the matrix cores (54 steps, an accumulator
//     of 256 audio samples per wave; the same fmaf chain in ascending tap order as every direct-form filter of the library).  A first cut
//     did this on the packed VALU, one output per lane and pair of batch halves: 544 B of LDS reads per audio sample made the pass
//     LDS-bound (25 of 190 us, measured by ablation); the matrix form reads 106 B per sample.  Then the low-rate recurrence
//     y[m] = b0 v[m] + q y[m-1]: zero-state run over a lane's four outputs, wave scan, carry across the four waves
//     (fir_win_cplx_kernel's scheme), and 1 024 audio samples leave as one coalesced 4 KB row;
//   * NO carry between workgroups: a run that does not start the chunk first computes the tile in front of it, which gives the filter its 135
//     samples of history and the recurrence a zero-state warm-up over 75 audio outputs (q^75 = 1.4e-10: below half an ulp of the audio) - the
//     trick lrhip_chain_halo plays across GPUs, here between workgroups.  One extra tile (and one wave's audio product) per run: 1 / (10
//     batches-per-run) of the tuner work; the batches are dealt out evenly over one round of workgroups.
//
// What the launch carries from chunk to chunk is what the two stages carry on their own (raw tuner history, the discriminator's previous
// sample, 135 discriminator samples, the recurrence state, the two decimation indices), in the same buffers: the host may alternate between
// this kernel and the two-launch form from one chunk to the next.
// Algorithmic traffic: 8 B in + 4/25 B out per RF sample (8.16 B); the overlap tile of a run is re-read from L2 / HBM (+3 %).
#pragma once
#include "kernels_fir.h"

// N > 0: the tuner's Toeplitz A fragments of the first N of its 51 MFMA steps stay in registers across the tiles of a launch (kernels_fir.h mfma_tile_areg):
// one LDS read per such step instead of two.  51 would spill (123 + 51 registers against the 168 of three waves per SIMD); 0 = the round-3 loop
#ifndef LRHIP_RX_AREG
#define LRHIP_RX_AREG 40
#endif
// 2: the tuner's 51 MFMA steps as two interleaved chains (even / odd steps) summed at the end (mfma_tile_areg SPLIT); 1: one chain.  Measured EQUAL (0.1535 against
// 0.1555 ms, three alternations on one box, profiles/r04_twiddle_registers_ab.txt): three waves per SIMD hide the dependent-accumulator latency already
#ifndef LRHIP_RX_SPLITK
#define LRHIP_RX_SPLITK 1
#endif

namespace lrhip {

struct RxParams {
    // ---- tuner: rotator + 128 real taps + decimation 5 + discriminator (FirStage A)
    const float *hist;           // 127 raw ComplexFloat32 samples before x[0]
    const float *x;
    long n;                      // RF samples in this chunk
    const float *taps_pad;       // zero-padded reversed taps (fir_mfma_build_taps)
    long n_out_a, first_a;       // tuner outputs of this chunk; stream position of output 0 (carried downsampler index)
    int e;                       // alignment slack of the staged window
    long ntiles;
    uint64_t rot_step_fx, rot_count0;
    const float2 *prev_in;       // the tuner output before the chunk (absolute phase)
    float2 *prev_out;
    double inv_gain;
    float *hist_out;
    // ---- audio tail: 136-tap decimating filter + first-order recurrence at the low rate (FirStage B)
    const float *g_pad;          // zero-padded reversed taps of the 136-tap filter (fir_mfma_build_taps, decimation 5)
    const float *thist_in;       // 135 discriminator samples before the chunk
    float *thist_out;
    long first_b, n_out_b;       // carried index of the tail's downsampler (0..4); audio samples this chunk emits
    float *y;
    float b0, na1, na1_lo;       // y[m] = b0 v[m] + (na1 + na1_lo) y[m-1]
    const float *ptab4;          // ptab4[l] = q^(4 (l+1)), l < 64
    const float *state_in;
    float *state_out;
    // ---- runs
    int min_tiles;               // (host) a run is at least this many tiles
    int dbg;                     // ablation bits (LRHIP_RX_DBG; wrong results): 1 no audio product, 2 no discriminator arithmetic, 4 no tuner MFMA loop, 8 no HBM reads
};

constexpr int RX_D = 5, RX_KS = 51, RX_M = 128, RX_MT = 136;
constexpr int RX_KST = 54;                                            // audio filter: slack (0..4) + 15 x 5 + 136 <= 4 x 54
// LRHIP_RX_NACC 2 (round 4, VERDICT r03 next 7 - the halved skeleton): two accumulators = 256 outputs per wave, 1 024 per tile: one staging pass and one pair of
// barriers per 1 024 outputs, 5 120 + 203 staged samples instead of 2 x (2 560 + 203); the window doubles to 43 KB, so two workgroups per CU instead of three
#ifndef LRHIP_RX_NACC
#define LRHIP_RX_NACC 1
#endif
constexpr int RX_NACC = LRHIP_RX_NACC;
constexpr int RX_TILE = FirMfmaGeom<2, RX_D>::tile_out(RX_NACC);      // 512 (1 024) tuner outputs per tile
#ifndef LRHIP_RX_TPB
#define LRHIP_RX_TPB (LRHIP_RX_NACC == 2 ? 5 : 10)      /* tiles per batch: 10 = 1 024 audio outputs, an accumulator for each of the four waves, 46 KB of LDS (3 workgroups per CU); 5 = 512 audio outputs on waves 0-1, 36 KB (4 per CU) */
#endif
constexpr int RX_TPB = LRHIP_RX_TPB;                                  // tiles per batch
constexpr int RX_BATCH = RX_TILE * RX_TPB;                            // 5 120 discriminator samples
constexpr int RX_AUDIO = RX_BATCH / 5;                                // 1 024 audio outputs per batch: one accumulator (256) per wave
constexpr int RX_AW = RX_AUDIO / 256;                                 // waves that run the audio product
constexpr int RX_TH = RX_MT - 1;                                      // 135 samples of tail history
constexpr int RX_SPAN = FirMfmaGeom<2, RX_D>::span(RX_NACC, RX_KS);
constexpr int RX_TLEN = fir_taps_len(RX_D, RX_KS);
#ifndef LRHIP_RX_TAP_COPIES
#define LRHIP_RX_TAP_COPIES 0      /* 1: four copies of the tuner tap array, 303 floats apart (conflict-free A-fragment reads, mfma_tile TQS) - measured equal on the receiver (0.1515 against 0.1518 ms, same box): the conflicts are not on its critical path; 0: one copy */
#endif
constexpr int RX_TQS = LRHIP_RX_TAP_COPIES ? 303 : 0;                 // 303 = 15 mod 32, >= TLEN
constexpr int RX_TFLOATS = LRHIP_RX_TAP_COPIES ? ((3 * RX_TQS + RX_TLEN + 3) / 4) * 4 : RX_TLEN;
constexpr int RX_XF = FirMfmaGeom<2, RX_D>::phys(2 * RX_SPAN) + FirMfmaGeom<2, RX_D>::PAD + 8;
// audio window: logical float a = (discriminator sample of the batch) + 135, padded rows of the Float32 Toeplitz product
constexpr int RX_PSPAN = FirMfmaGeom<1, RX_D>::span(1, RX_KST, RX_AW);       // 5 256 = 5 120 + 135 + 1
constexpr int RX_PF = ((FirMfmaGeom<1, RX_D>::phys(RX_PSPAN) + FirMfmaGeom<1, RX_D>::PAD + 3) / 4) * 4;
constexpr int RX_GZ = 4;                                              // extra leading zeros of the audio tap table: slack up to 4 (fir_taps_zl covers 3)
constexpr int RX_GLEN = RX_GZ + fir_taps_len(RX_D, RX_KST);
// LDS map (floats): [taps_pad TLEN | window XF (the audio pass reuses its head as the 1 024-float output row) | P | audio taps GLEN | ptab4 64 | xch 8 | eo 16 | 4]
constexpr int RX_LDS_X = RX_TFLOATS;
constexpr int RX_LDS_P = RX_LDS_X + ((RX_XF + 3) / 4) * 4;
constexpr int RX_LDS_G = RX_LDS_P + RX_PF;
constexpr int RX_LDS_PT = RX_LDS_G + RX_GLEN;
constexpr int RX_LDS_XCH = RX_LDS_PT + 64;
constexpr int RX_LDS_EO = RX_LDS_XCH + 8;
constexpr int RX_LDS_PREV = RX_LDS_EO + 16;
constexpr int RX_LDS_FLOATS = RX_LDS_PREV + 4;
static_assert(RX_TQS == 0 || (RX_TQS >= RX_TLEN && RX_TQS % 32 == 15), "tap copies: 16 banks apart");
static_assert(RX_TILE == 512 * RX_NACC && (RX_TPB == 10 || RX_TPB == 5) && RX_AUDIO == 256 * RX_AW && RX_TLEN % 4 == 0 && RX_GLEN % 4 == 0 && RX_PSPAN == RX_BATCH + RX_TH + 1, "receiver geometry");
static_assert(RX_XF >= RX_AUDIO, "the audio output row lives in the RF window area");

#ifndef LRHIP_RX_WAVES_PER_SIMD
#define LRHIP_RX_WAVES_PER_SIMD (LRHIP_RX_NACC == 2 ? 2 : LRHIP_RX_TPB == 5 ? 4 : 3)
#endif

// discriminator sample b of the batch (-135 .. 5119: negative = the history in front of it) -> its float in the padded audio window
__device__ __forceinline__ int rx_pos(int b) { return FirMfmaGeom<1, RX_D>::phys(b + RX_TH); }

// stream = [127 ComplexFloat32 history samples | chunk], the chunk as ComplexFloat32 or as raw records
template <int FMT>
__device__ __forceinline__ cf rx_stream_at(const float *__restrict__ hist, const float *__restrict__ x, long p, int M, long n)
{
    if (p < 0) return cf{0.f, 0.f};
    if (p < M - 1) return cf{hist[2 * p], hist[2 * p + 1]};
    const long xi = p - (M - 1);
    if (xi >= n) return cf{0.f, 0.f};
    if (FMT == RX_FMT_CF32) return cf{x[2 * xi], x[2 * xi + 1]};
    if (FMT == RX_FMT_S16LE) {
        const uint16_t *b = reinterpret_cast<const uint16_t *>(x) + 2 * xi;
        return rx_raw_sample<FMT>(b[0], b[1]);
    }
    const uint8_t *b = reinterpret_cast<const uint8_t *>(x) + 2 * xi;
    return rx_raw_sample<FMT>(b[0], b[1]);
}

// -DLRHIP_RX_TRACE (a variant library, tools/rx_trace.sh): lane 0 of every wave of the first workgroups stamps the phases of its first tiles with clock64()
#ifdef LRHIP_RX_TRACE
__device__ unsigned long long *lrhip_rx_trace;          // [block 8][wave 4][tile 64][8]
#define RX_STAMP(i)                                                                                                                                 \
    do {                                                                                                                                            \
        if (lrhip_rx_trace && blockIdx.x < 8 && trace_tile < 64 && (threadIdx.x & 63) == 0)                                                         \
            lrhip_rx_trace[(((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + trace_tile) * 8 + (i)] = clock64();                                  \
    } while (0)
#else
#define RX_STAMP(i) do { } while (0)
#endif

template <int FMT>
__global__ __launch_bounds__(256, LRHIP_RX_WAVES_PER_SIMD) void rx_fused_kernel(const RxParams pr)
{
    constexpr bool U8 = FMT != RX_FMT_CF32;      // raw records (the name of the first format folded in)
    constexpr int NT = 256, D = RX_D, S = 2, M = RX_M;
    constexpr int NF4 = RX_SPAN * S / 4;
    constexpr int UX = (NF4 + NT - 1) / NT;
    static_assert(2 * UX <= 64, "one lane per uniform phasor");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *ldsT = lds, *ldsX = lds + RX_LDS_X, *P = lds + RX_LDS_P, *ldsGT = lds + RX_LDS_G, *ldsPt = lds + RX_LDS_PT, *xch = lds + RX_LDS_XCH;
    float2 *eo = reinterpret_cast<float2 *>(lds + RX_LDS_EO);         // eo[4 (t & 1) + w] = last filter output of wave w in tile t (tile basis)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long n = pr.n;
    const float *__restrict__ x = pr.x;
    const float *__restrict__ hist = pr.hist;

    // raw tuner history for the next chunk (the other ping-pong buffer)
    if (pr.hist_out && blockIdx.x == 0)
        for (int i = tid; i < M - 1; i += NT) {
            const cf v = rx_stream_at<FMT>(hist, x, n + i, M, n);
            pr.hist_out[2 * i] = v.x;
            pr.hist_out[2 * i + 1] = v.y;
        }
    for (int i = tid; i < RX_TLEN; i += NT) {
        const float v = pr.taps_pad[i];
        ldsT[i] = v;
        if (RX_TQS) { ldsT[RX_TQS + i] = v; ldsT[2 * RX_TQS + i] = v; ldsT[3 * RX_TQS + i] = v; }
    }
    for (int i = tid; i < RX_GLEN; i += NT) ldsGT[i] = i < RX_GZ ? 0.f : pr.g_pad[i - RX_GZ];
    if (tid < 64) ldsPt[tid] = pr.ptab4[tid];
    // Every float of the audio window meets taps in the Toeplitz product - zero taps outside an output's own 136 - so what a run has not
    // written yet (the batch in front of it, the tail of the chunk's last batch, the last window float) must be FINITE: 0 * NaN poisons a block
    for (int i = tid; i < RX_PF; i += NT) P[i] = 0.f;
    __syncthreads();
#if LRHIP_RX_AREG
    float areg[LRHIP_RX_AREG];                                        // the tuner's A fragments: the same floats for every tile of the launch
    mfma_load_areg<S, D, RX_KS, RX_TQS, LRHIP_RX_AREG>(ldsT, pr.e, areg);
#endif

    // ---- this workgroup's run
    // the chunk's TILES are dealt out evenly (the first ntiles mod grid workgroups take one more); a run's batches count from its own first tile,
    // so the audio outputs of a run are those whose window ends inside its tiles: m0 <= m < m1, and `phi` is the position of output m0's last
    // sample in the run's first tile (0 .. 4)
    const long tbase = pr.ntiles / gridDim.x, trem = pr.ntiles % gridDim.x;
    const long t0 = (long)blockIdx.x * tbase + ((long)blockIdx.x < trem ? (long)blockIdx.x : trem);
    const long tend = t0 + tbase + ((long)blockIdx.x < trem ? 1 : 0);
    const long m0 = (t0 * RX_TILE - pr.first_b + 4) / 5, m1 = (tend * RX_TILE - pr.first_b + 4) / 5;
    const int phi = (int)(pr.first_b + 5 * m0 - t0 * RX_TILE);
    const bool chunk_start = t0 == 0;
    long t = chunk_start ? 0 : t0 - 1;                                // the tile in front of the run: history + warm-up, output discarded
    const long tfirst = t;
    float carry = 0.f;
    if (chunk_start) {
        if (tid < RX_TH) P[rx_pos(tid - RX_TH)] = pr.thist_in[tid];
        carry = pr.state_in[0];
    }
    const cf tileR = phasor_poly(pr.rot_step_fx * (uint64_t)(RX_TILE * RX_D));      // phase basis of tile t+1 over that of tile t

    auto xlo_of = [&](long tt) { return pr.first_a + tt * (long)RX_TILE * D - pr.e - (M - 1); };
    auto interior = [&](long tt) { const long lo = xlo_of(tt); return tt < tend && lo >= 0 && lo + RX_SPAN <= n; };

    // window-relative rotator phasors (fir_mfma_persistent_kernel, REL): float4 tid + 256 u holds window samples 2 tid + 512 u + {0, 1}
    cf rel_w[UX][2];
    {
        const cf pt = phasor_poly(pr.rot_step_fx * (uint64_t)(2 * tid)), pu = phasor_poly(pr.rot_step_fx * (uint64_t)(2 * NT * (lane >> 1) + (lane & 1)));
#pragma unroll
        for (int u = 0; u < UX; u++)
#pragma unroll
            for (int j = 0; j < 2; j++) rel_w[u][j] = cmul(pt, cf{__shfl(pu.x, 2 * u + j), __shfl(pu.y, 2 * u + j)});
    }

    float4 pre[U8 ? 1 : UX];
    uint2 pre8[U8 ? UX : 1];                                         // raw records: the same two samples per lane and load are 4 (.x) or 8 bytes
    bool have = false;
    auto prefetch = [&](long tt) {
        have = interior(tt);
        if (U8) {
            if (have) {
                const uint8_t *src0 = reinterpret_cast<const uint8_t *>(x) + 2 * rx_raw_bytes<FMT>() * xlo_of(tt);
#pragma unroll
                for (int u = 0; u < UX; u++) {
                    const int idx = tid + NT * u, ic = idx < NF4 ? idx : NF4 - 1;
                    if (FMT == RX_FMT_S16LE) pre8[u] = reinterpret_cast<const uint2 *>(src0)[ic];
                    else pre8[u].x = reinterpret_cast<const unsigned *>(src0)[ic];
                }
            }
        } else if (have && (pr.dbg & 8)) {                            // ablation: no HBM reads
#pragma unroll
            for (int u = 0; u < UX; u++) pre[u] = make_float4(0.5f, 0.25f, -0.5f, 0.125f);
        } else if (have) {
            const float4 *src = reinterpret_cast<const float4 *>(x + xlo_of(tt) * S);
#pragma unroll
            for (int u = 0; u < UX; u++) {
                const int idx = tid + NT * u;
                pre[u] = src[idx < NF4 ? idx : NF4 - 1];
            }
        }
    };
    prefetch(t);

    [[maybe_unused]] int trace_tile = 0;
    for (; t < tend; t++) {
        RX_STAMP(0);
        const bool warm = t < t0;
        const int tau = warm ? RX_TPB - 1 : (int)((t - t0) % RX_TPB);  // place of the tile in its batch (the warm-up tile: the last of the batch in front)
        const long bidx = warm ? 0 : (t - t0) / RX_TPB;               // batch of the run
        const long tile_k0 = t * (long)RX_TILE;
        // ---- stage the RF window, rotated relative to its first sample
        if (have) {
#pragma unroll
            for (int u = 0; u < UX; u++) {
                const int i4 = tid + u * NT;
                cf s0, s1;
                if (U8) {
                    if (FMT == RX_FMT_S16LE) {
                        s0 = rx_raw_sample<FMT>(pre8[u].x & 0xffffu, pre8[u].x >> 16);
                        s1 = rx_raw_sample<FMT>(pre8[u].y & 0xffffu, pre8[u].y >> 16);
                    } else {
                        const unsigned w = pre8[u].x;
                        s0 = rx_raw_sample<FMT>(w & 0xffu, (w >> 8) & 0xffu);
                        s1 = rx_raw_sample<FMT>((w >> 16) & 0xffu, w >> 24);
                    }
                } else {
                    s0 = cf{pre[U8 ? 0 : u].x, pre[U8 ? 0 : u].y};
                    s1 = cf{pre[U8 ? 0 : u].z, pre[U8 ? 0 : u].w};
                }
                const cf a = cmul(s0, rel_w[u][0]), b = cmul(s1, rel_w[u][1]);
                if (i4 < NF4) lds_put4<S, D>(ldsX, i4, make_float4(a.x, a.y, b.x, b.y));
            }
        } else if (U8) {
            // stage_edge<S, D, true, true, NT> on the record stream
            using GE = FirMfmaGeom<S, D>;
            const long base = pr.first_a + tile_k0 * D - pr.e;
            for (int r = tid; r < RX_SPAN; r += NT) {
                const cf o = cmul(rx_stream_at<FMT>(hist, x, base + r, M, n), rel_window_phasor<NT>(pr.rot_step_fx, r));
                const int pa = GE::phys(S * r);
                ldsX[pa] = o.x;
                ldsX[pa + 1] = o.y;
            }
        } else {
            stage_edge<S, D, true, true, NT>(ldsX, hist, x, pr.first_a + tile_k0 * D - pr.e, RX_SPAN, M, n, pr.rot_step_fx, pr.rot_count0);
        }
        RX_STAMP(1);
        __syncthreads();                                              // (A) window staged; the previous tile's patches and history copy are visible
        RX_STAMP(2);
        prefetch(t + 1 < tend ? t + 1 : tend);
        RX_STAMP(3);

        // ---- filter: banded-Toeplitz product on the f32 matrix cores, one accumulator (128 outputs) per wave
        f32x4 acc[1][RX_NACC];
        if (pr.dbg & 4) {
#pragma unroll
            for (int a = 0; a < RX_NACC; a++) acc[0][a] = (f32x4){ldsX[tid], ldsX[tid + 256], ldsX[tid + 512], ldsX[tid + 768]};
        }
#if LRHIP_RX_AREG
        else mfma_tile_areg<S, D, RX_NACC, RX_KS, RX_TQS, LRHIP_RX_AREG, LRHIP_RX_SPLITK>(areg, ldsT, pr.e, ldsX, acc);
#else
        else mfma_tile<S, D, RX_NACC, RX_KS, 1, RX_TQS>(ldsT, RX_TLEN, pr.e, ldsX, RX_KS, acc);
#endif

        // ---- discriminator on the accumulators -> P.  After the re/im exchange a lane owns two consecutive filter outputs; the one in front of them is
        // one shuffle away - except for lane 0, whose predecessor is the previous wave's last output (through LDS, after the barrier that also
        // frees the window) or, for wave 0, the previous TILE's last output, which lives in that tile's phase basis: bases of consecutive tiles
        // differ by the constant phasor R = exp(j omega TILE D), so lane 0 takes prev * conj(R).  Only a run's very first sample meets a
        // predecessor in absolute phase (the carried one) and pays for the tile's own phasor.
        RX_STAMP(4);
#if LRHIP_RX_NACC == 1
        // (one accumulator per wave: the round-3 epilogue, kept verbatim - the generalised form below compiles 1.6 % slower at one accumulator)
        {
            const int col = lane & 15, kq = lane >> 4;
            const bool odd = col & 1;
            const int src = odd ? lane - 1 : kq ? lane - 15 : col ? col + 47 : 63;      // the lane that owns the output before this lane's first one
            const float a0 = acc[0][0][0], a1 = acc[0][0][1], a2 = acc[0][0][2], a3 = acc[0][0][3];
            const float recv0 = __shfl_xor(odd ? a0 : a2, 1);
            const float recv1 = __shfl_xor(odd ? a1 : a3, 1);
            const float2 o0 = odd ? make_float2(recv0, a2) : make_float2(a0, recv0);
            const float2 o1 = odd ? make_float2(recv1, a3) : make_float2(a1, recv1);
            float2 p = make_float2(__shfl(o1.x, src), __shfl(o1.y, src));
            float2 *eo_t = eo + 4 * (int)(t & 1), *eo_p = eo + 4 * (int)((t & 1) ^ 1);     // last outputs of the four waves: this tile's, the previous tile's
            if (lane == 63) eo_t[wave] = o1;
            RX_STAMP(5);
            __syncthreads();                                          // (B) window free; the waves' last outputs are visible
            RX_STAMP(6);
            // (+ 0: a silent stretch gives exactly +0 filter outputs, but +0 times a phasor with negative parts is -0, and the angle of a zero product
            // is decided by the signs of the zeros - frequencydiscriminator.lua:74 via discriminate(): keep what the reference's own operands would be)
            if (lane == 0) p = wave ? eo_t[wave - 1] : cf_to(cmulc(cf_from(eo_p[3]), tileR) + cf{0.f, 0.f});
            float2 d = (pr.dbg & 2) ? make_float2(o0.x + p.x, o1.y) : make_float2(discriminate(o0, p, pr.inv_gain), discriminate(o1, o0, pr.inv_gain));
            const int lk = wave * 128 + 16 * (col >> 1) + 4 * kq + (odd ? 2 : 0);          // tile-local index of o0
            const long k = tile_k0 + lk;
            if (t == tfirst && tid == 0) {
                // the run's first sample: the carried output is in absolute phase (zero in front of a warm-up tile, whose first angle is never used)
                const cf pt = phasor_poly(pr.rot_step_fx * (pr.rot_count0 + (uint64_t)xlo_of(t)));
                d.x = discriminate(cf_to(cmul(cf_from(o0), pt) + cf{0.f, 0.f}), chunk_start ? *pr.prev_in : make_float2(0.f, 0.f), pr.inv_gain);
            }
            const int b = tau * RX_TILE + lk;
            P[rx_pos(b)] = d.x;
            P[rx_pos(b + 1)] = d.y;
            // the chunk's last tuner output, in absolute phase, for the next chunk
            if (k == pr.n_out_a - 1 || k + 1 == pr.n_out_a - 1) {
                const cf pt = phasor_poly(pr.rot_step_fx * (pr.rot_count0 + (uint64_t)xlo_of(t)));
                *pr.prev_out = cf_to(cmul(cf_from(k == pr.n_out_a - 1 ? o0 : o1), pt) + cf{0.f, 0.f});
            }
        }
#else
        {
            const int col = lane & 15, kq = lane >> 4;
            const bool odd = col & 1;
            const int src = odd ? lane - 1 : kq ? lane - 15 : col ? col + 47 : 63;      // the lane that owns the output before this lane's first one
            float2 o0s[RX_NACC], o1s[RX_NACC], ps[RX_NACC];
#pragma unroll
            for (int a = 0; a < RX_NACC; a++) {
                const float a0 = acc[0][a][0], a1 = acc[0][a][1], a2 = acc[0][a][2], a3 = acc[0][a][3];
                const float recv0 = __shfl_xor(odd ? a0 : a2, 1);
                const float recv1 = __shfl_xor(odd ? a1 : a3, 1);
                o0s[a] = odd ? make_float2(recv0, a2) : make_float2(a0, recv0);
                o1s[a] = odd ? make_float2(recv1, a3) : make_float2(a1, recv1);
                ps[a] = make_float2(__shfl(o1s[a].x, src), __shfl(o1s[a].y, src));
            }
            float2 *eo_t = eo + 4 * (int)(t & 1), *eo_p = eo + 4 * (int)((t & 1) ^ 1);     // last outputs of the four waves: this tile's, the previous tile's
            if (lane == 63) eo_t[wave] = o1s[RX_NACC - 1];
            __syncthreads();                                          // (B) window free; the waves' last outputs are visible
#pragma unroll
            for (int a = 0; a < RX_NACC; a++) {
                const float2 o0 = o0s[a], o1 = o1s[a];
                float2 p = ps[a];
                // (+ 0: a silent stretch gives exactly +0 filter outputs, but +0 times a phasor with negative parts is -0, and the angle of a zero product
                // is decided by the signs of the zeros - frequencydiscriminator.lua:74 via discriminate(): keep what the reference's own operands would be)
                if (a == 0) {
                    if (lane == 0) p = wave ? eo_t[wave - 1] : cf_to(cmulc(cf_from(eo_p[3]), tileR) + cf{0.f, 0.f});
                } else {
                    // the second accumulator's first output follows the first accumulator's last one: lane 63 of the same wave (ps[a] of lane 0 read lane 63 of
                    // accumulator a; the right one is accumulator a - 1)
                    const float2 q = make_float2(__shfl(o1s[a > 0 ? a - 1 : 0].x, 63), __shfl(o1s[a > 0 ? a - 1 : 0].y, 63));
                    if (lane == 0) p = q;
                }
                float2 d = (pr.dbg & 2) ? make_float2(o0.x + p.x, o1.y) : make_float2(discriminate(o0, p, pr.inv_gain), discriminate(o1, o0, pr.inv_gain));
                const int lk = (wave * RX_NACC + a) * 128 + 16 * (col >> 1) + 4 * kq + (odd ? 2 : 0);          // tile-local index of o0
                const long k = tile_k0 + lk;
                if (a == 0 && t == tfirst && tid == 0) {
                    // the run's first sample: the carried output is in absolute phase (zero in front of a warm-up tile, whose first angle is never used)
                    const cf pt = phasor_poly(pr.rot_step_fx * (pr.rot_count0 + (uint64_t)xlo_of(t)));
                    d.x = discriminate(cf_to(cmul(cf_from(o0), pt) + cf{0.f, 0.f}), chunk_start ? *pr.prev_in : make_float2(0.f, 0.f), pr.inv_gain);
                }
                const int b = tau * RX_TILE + lk;
                P[rx_pos(b)] = d.x;
                P[rx_pos(b + 1)] = d.y;
                // the chunk's last tuner output, in absolute phase, for the next chunk
                if (k == pr.n_out_a - 1 || k + 1 == pr.n_out_a - 1) {
                    const cf pt = phasor_poly(pr.rot_step_fx * (pr.rot_count0 + (uint64_t)xlo_of(t)));
                    *pr.prev_out = cf_to(cmul(cf_from(k == pr.n_out_a - 1 ? o0 : o1), pt) + cf{0.f, 0.f});
                }
            }
        }
#endif

        RX_STAMP(7);
#ifdef LRHIP_RX_TRACE
        trace_tile++;
#endif
        const bool last_tile = t == pr.ntiles - 1;
        if (tau == RX_TPB - 1 || t == tend - 1) {
            __syncthreads();                                          // (P) the batch's angles are all in P
            // ---- audio: the 136-tap decimating filter as a Toeplitz product on P (slack = the tail's carried downsampler index), 256 outputs per wave.
            // In front of a run only the last tile of the batch is real: its 75 whole windows are all in wave 3's accumulator
            f32x4 acct[1][1];
            acct[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (!(pr.dbg & 1) && wave < RX_AW && (!warm || wave == RX_AW - 1)) mfma_tile<1, D, 1, RX_KST, 1>(ldsGT + RX_GZ, RX_GLEN - RX_GZ, phi, P, RX_KST, acct);
            float *vrow = ldsX;                                       // the RF window is free between barrier (B) and the next staging
            {
                const int col = lane & 15, kq = lane >> 4;
                if (wave < RX_AW) *reinterpret_cast<float4 *>(vrow + 16 * (wave * 16 + col) + 4 * kq) = make_float4(acct[0][0][0], acct[0][0][1], acct[0][0][2], acct[0][0][3]);
            }
            __syncthreads();
            const bool scan_lane = tid < 64 * RX_AW;                  // the lanes that own four audio outputs of this batch
            float4 v4 = scan_lane ? *reinterpret_cast<const float4 *>(vrow + 4 * tid) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (warm) {
                // the first window that lies inside the tile in front of the run (its first sample, whose own predecessor was not computed, excluded)
                const int kmin = (RX_TH + (RX_TPB - 1) * RX_TILE + 1 - phi + 4) / 5;
                const int k = 4 * tid;
                v4 = make_float4(k >= kmin ? v4.x : 0.f, k + 1 >= kmin ? v4.y : 0.f, k + 2 >= kmin ? v4.z : 0.f, k + 3 >= kmin ? v4.w : 0.f);
            }
            // ---- recurrence at the low rate: zero-state run over the lane's four outputs, inclusive scan inside the wave, carry across the waves
            const float u0 = pr.b0 * v4.x, u1 = pr.b0 * v4.y, u2 = pr.b0 * v4.z, u3 = pr.b0 * v4.w;
            auto stepq = [&](float stv, float uv) { return fmaf(pr.na1, stv, fmaf(pr.na1_lo, stv, uv)); };
            float z = stepq(stepq(stepq(u0, u1), u2), u3);
#pragma unroll
            for (int l = 0; l < 6; l++) {
                const float prev = __shfl_up(z, 1 << l);
                if (lane >= (1 << l)) z = z + fmaf(ldsPt[(1 << l) - 1], prev, 0.f);
            }
            if (lane == 63 && scan_lane) xch[wave] = z;
            __syncthreads();
            const float pw256 = ldsPt[63];
            float C = carry, Cw = carry;
#pragma unroll
            for (int w = 0; w < RX_AW; w++) {
                if (w == wave) Cw = C;
                C = xch[w] + fmaf(pw256, C, 0.f);
            }
            const float Sx = z + fmaf(ldsPt[lane], Cw, 0.f);          // true state after this lane's four outputs
            float st = __shfl_up(Sx, 1);
            if (lane == 0) st = Cw;
            const float y0 = stepq(st, u0), y1 = stepq(y0, u1), y2 = stepq(y1, u2), y3 = stepq(y2, u3);
            if (!warm && scan_lane) {
                const long m = m0 + bidx * RX_AUDIO + 4 * tid;
                const long mend = m1 < pr.n_out_b ? m1 : pr.n_out_b;  // outputs past the run's last tile belong to the next run
                if (m + 3 < mend && (reinterpret_cast<uintptr_t>(pr.y + m) & 15) == 0) {
                    *reinterpret_cast<float4 *>(pr.y + m) = make_float4(y0, y1, y2, y3);
                } else {
                    if (m < mend) pr.y[m] = y0;
                    if (m + 1 < mend) pr.y[m + 1] = y1;
                    if (m + 2 < mend) pr.y[m + 2] = y2;
                    if (m + 3 < mend) pr.y[m + 3] = y3;
                }
                // the carried state: a chunk that ends inside the batch hands over the value at its last output; one that ends WITH the batch
                // the scanned end state C - what the next batch of an uninterrupted run is given (fir_win_cplx_kernel's rule)
                const bool ends_with_batch = pr.n_out_b == m0 + (bidx + 1) * RX_AUDIO;
                const long last = pr.n_out_b - 1 - m;
                if (!ends_with_batch && last >= 0 && last < 4) pr.state_out[0] = last == 0 ? y0 : last == 1 ? y1 : last == 2 ? y2 : y3;
                if (ends_with_batch && tid == 0) pr.state_out[0] = C;
            }
            carry = C;
            __syncthreads();                                          // (Q) every wave is done reading P
            if (last_tile) {
                // the chunk's last 135 discriminator samples for the next chunk
                const int ev = (int)(pr.n_out_a - (t0 * RX_TILE + bidx * RX_BATCH));   // valid samples of this batch, 1 .. 5120
                if (tid < RX_TH) pr.thist_out[tid] = P[rx_pos(ev - RX_TH + tid)];
            } else if (tid < RX_TH) {
                P[rx_pos(tid - RX_TH)] = P[rx_pos(RX_BATCH - RX_TH + tid)];      // history of the next batch = the end of this one (visible after barrier A)
            }
        }
    }
}

}  // namespace lrhip
