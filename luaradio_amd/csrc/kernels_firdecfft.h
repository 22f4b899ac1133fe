// kernels_firdecfft.h - decimating FIR (Decimator / Tuner: [FrequencyTranslator ->] FIRFilter -> Downsampler, optionally
// -> FrequencyDiscriminator) by POLYPHASE FFT overlap-save, one fused kernel.
//
// Reference semantics (radio/composites/tuner.lua:32-48, decimator.lua:28-42, radio/blocks/signal/firfilter.lua:230-305,
// downsampler.lua:45-56, frequencytranslator.lua:93-110): with the chunk x[0..n), the carried downsampler index `first`
// and c0 = absolute index of x[0], output k sits at input index n_k = first + k*D and is
//     y[k] = sum_i h[i] * x[n_k - i] * e^{j w (c0 + n_k - i)}  =  e^{j w (c0 + n_k)} * z[k],
//     z[k] = sum_i g[i] x[n_k - i],    g[i] = h[i] e^{-j w i}            (w = 0: plain Decimator, g = h)
// The reference computes every full-rate output and throws D-1 of D away; the direct-form kernel here (kernels_fir.h)
// computes only the kept ones but pays 15*D + M Toeplitz columns for M useful ones on the f32 matrix pipe, plus a rotator
// per INPUT sample on the VALU that shares that pipe.  This kernel uses the reference's own production algorithm (FFT
// overlap-save, firfilter.lua:320-398) in its polyphase form instead:
//     i = D q + r:   z[k] = sum_r (g_r * u_r)[k],   u_r[m] = x[first + D m - r],   g_r[q] = g[D q + r]
// D branch convolutions at the LOW rate, each done as a 256-point FFT overlap-save (branch filters have ceil(M/D) <= 32
// taps -> overlap V = 32, 224 new outputs per block) and summed in the frequency domain:
//     Z = sum_r FFT256(u_r) . G_r ;   z = IFFT256(Z)            -> D forward FFTs + 1 inverse per 224 outputs
// about 62 flop per input sample at D = 5 against 163 issued by the Toeplitz form, the rotation moves from every input sample
// to the OUTPUT (1/D of the rate) - and disappears altogether behind a discriminator: arg(y[k] conj(y[k-1])) =
// arg(z[k] conj(z[k-1]) e^{j w D}), one constant.  Not bit-identical to the fmaf-chain direct form: Float32 FFT arithmetic,
// <= 1e-6 of the f64 oracle for |x| <= 1 (same bar as the overlap-save kernel of kernels_firfft.h; tuner_spec holds 1e-5).
//
// Geometry.  A 256-point FFT = 16 lanes x 16 registers (dft16 in registers, twiddle, 16x16 transpose through LDS, dft16),
// so a wave runs FOUR transforms at a time, one per row of 16 lanes.  D = 4A + C phases:
//   * per block, A batches: rows 0..3 transform phases 4t..4t+3 of THIS block.  The block's 256 D samples are loaded
//     coalesced (8 B per lane, natural order), staged in LDS and read back in polyphase order, window position
//     D (16 i + u) + 4t + g into register i of lane (row g, u).  (Measured and dropped: loading in polyphase order straight
//     from global memory - every wave-load then covers one contiguous 128 D byte span, but 64 separate 8-byte pieces of it,
//     and the address pipeline, not HBM, became the limit: 0.18 ms against 0.15 ms for 2^26 samples.)  The products
//     with G are summed over t in registers and then across the four rows with v_permlane32_swap / v_permlane16_swap
//     (gfx950), which leaves 8 floats per lane;
//   * the C left-over phases of four consecutive blocks (a "quad") wait in registers and are transformed together, row b =
//     block b, after a 4x4 register/row transpose with the same swap instructions (which also brings every block's partial
//     sum to its row);
//   * one inverse batch per quad, row b = block b; outputs land as z[16 i + u] in register i of lane u: consecutive lanes =
//     consecutive outputs.
// No cross-wave exchange at all: the discriminator's previous output z[k-1] of a block's first output is position V-1 of the
// same transform (valid, because V - 1 >= ceil(M/D) - 1), so neither edge buffer nor fix-up launch exists.
// Also measured and dropped: the window by LDS-DMA (global_load_lds_dwordx4, 10 KB per wave in flight, awaited with vmcnt(0) right
// before its read-back; one 8-wave workgroup per CU): the arithmetic alone got faster (0.122 ms, no prefetch registers), but the
// input arrived at 2.9 TB/s against 4.6 TB/s for plain 8-byte loads in the same skeleton, whatever the block order - 0.24 ms.
// The next block's samples are requested before the current block's arithmetic (register prefetch); the 15 stage twiddles
// W_256^(u k) depend on the lane only and stay in registers; the G rows are read from LDS one dft16 ahead of their use.
// LDS: per wave the staged window (256 D complex; the 4 x 16 x 17 transpose area is laid over it when a block has one full
// batch), per workgroup G (D x 256) and the 256 output phasors (D = 5: 52 KB -> 2 workgroups = 8 waves per CU).
#pragma once
#include "common.h"
#include "kernels_fir.h"
#include "kernels_firfft.h"
#include "pk_math.h"

namespace lrhip {

constexpr int DF_N = 256;                 // transform length (decimated samples per block window)
constexpr int DF_V = 32;                  // overlap in decimated samples
constexpr int DF_LO = DF_N - DF_V;        // new outputs per block
constexpr int DF_ROW = 17;                // padded row of the 16x16 transpose (complex elements)
constexpr int DF_GRP = 16 * DF_ROW;       // transpose area per 16-lane row
constexpr int DF_EX = 4 * DF_GRP;         // transpose area per wave
// per-wave LDS (complex elements): the staged window, and the transpose area - aliased onto the window when every staged sample
// is read before the first transpose (at most one full batch per block: D < 8), behind it otherwise
__host__ __device__ constexpr int df_ex_offset(int D) { return D / 4 >= 2 ? DF_N * D : 0; }
__host__ __device__ constexpr int df_wave_elems(int D) { return D / 4 >= 2 ? DF_N * D + DF_EX : (DF_N * D > DF_EX ? DF_N * D : DF_EX); }

// table layout (complex elements): twA[16][16] | G full batches [A][16 k2][64 lanes] | G left-over [C][16 k2][16 k1] | rot[256]
__host__ __device__ constexpr int df_table_elems(int D) { return 256 + (D / 4) * 1024 + (D % 4) * 256 + 256; }
// LDS (complex elements): 4 waves x per-wave area | the tables without twA
__host__ __device__ constexpr int df_lds_elems(int D) { return 4 * df_wave_elems(D) + df_table_elems(D) - 256; }

__device__ __forceinline__ void swap32(float &a, float &b)      // a = [a.lo32 | b.lo32], b = [a.hi32 | b.hi32]
{
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r.x);
    b = __uint_as_float(r.y);
}
__device__ __forceinline__ void swap16(float &a, float &b)      // a = [a.r0 b.r0 a.r2 b.r2], b = [a.r1 b.r1 a.r3 b.r3]  (rows of 16 lanes)
{
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r.x);
    b = __uint_as_float(r.y);
}
// four registers (one per block), four rows each  ->  row b of p_s = row s of the old p_b   (4 x 4 transpose register <-> row)
__device__ __forceinline__ void transpose_rows(float &p0, float &p1, float &p2, float &p3)
{
    swap32(p0, p2);      // p0 = [p0.r0 p0.r1 p2.r0 p2.r1], p2 = [p0.r2 p0.r3 p2.r2 p2.r3]
    swap32(p1, p3);
    swap16(p0, p1);      // p0 = [p0.r0 p1.r0 p2.r0 p3.r0], p1 = [p0.r1 p1.r1 p2.r1 p3.r1]
    swap16(p2, p3);      // p2 = rows 2, p3 = rows 3
}

// 16 x 16 transpose inside every row of 16 lanes: lane a register b -> lane b register a
__device__ __forceinline__ void df_transpose(cf *ex, cf (&v)[16], int g, int u)
{
    cf *base = ex + g * DF_GRP;
#pragma unroll
    for (int k = 0; k < 16; k++) base[u * DF_ROW + k] = v[k];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = base[k * DF_ROW + u];
}

// forward: lane u holds in[16 i + u] in register i  ->  lane k1 holds X[k1 + 16 k2] in register k2.  tw[k] = W_256^(u k).
__device__ __forceinline__ void df_fft_fwd(cf *ex, const cf (&tw)[16], cf (&v)[16], int g, int u)
{
    dft16<1>(v);
#pragma unroll
    for (int k = 1; k < 16; k++) v[k] = cmul(v[k], tw[k]);
    df_transpose(ex, v, g, u);
    dft16<1>(v);
}
// inverse (no 1/N): lane k1 holds Z[k1 + 16 k2] in register k2  ->  lane u holds z[16 i + u] in register i
__device__ __forceinline__ void df_fft_inv(cf *ex, const cf (&tw)[16], cf (&v)[16], int g, int u)
{
    dft16<-1>(v);
#pragma unroll
    for (int k = 1; k < 16; k++) v[k] = cmulc(v[k], tw[k]);
    df_transpose(ex, v, g, u);
    dft16<-1>(v);
}

// previous element along a row of 16 lanes: lane u gets `cur` of lane u-1, lane 0 gets `before` of lane 15
__device__ __forceinline__ float row_prev(float cur, float before)
{
    const int wrapped = __builtin_amdgcn_update_dpp(0, __float_as_int(before), 0x121 /* row_ror:1 */, 0xf, 0xf, false);
    return __int_as_float(__builtin_amdgcn_update_dpp(wrapped, __float_as_int(cur), 0x111 /* row_shr:1 */, 0xf, 0xf, false));
}

struct DfParams {
    int M;
    long n, n_out, first;          // chunk length, outputs, carried downsampler index
    long nblocks;
    int rounds;                    // quads per wave (one-shot order)
    uint64_t rot_step_fx, rot_count0;   // rot_step_fx = 0: no rotation
    float2 cD;                     // e^{j w D}
    double inv_gain;
    const float *taps_rev;         // reversed taps (direct-form order), M floats or M {re, im} pairs: exact re-evaluation of single outputs
    int taps_complex;
    int dbg;                       // ablation bits for tools/ab_decfft.py (0 in production): 1 no global loads, 2 no staging, 4 no forward batches, 8 no inverse, 16 no epilogue arithmetic, 32 no stores
};

// discriminator on UNROTATED filter outputs: arg(a conj(b) cD) / gain, cD = e^{j w D}.
// A product that is exactly zero (the first output of a stream: zero previous sample; or exact silence) is decided in the reference by
// the SIGNS of the zeros (frequencydiscriminator.lua:74 with complexfloat32.lua:79-81), i.e. by the signs of the filter output's
// components - which sit at the 1e-6 level of the FFT arithmetic when the filter has just started.  That one output is therefore
// re-evaluated in direct form (rotated samples, fmaf chain in the reference's tap order: bit-identical to kernels_fir.h), out of line.
__device__ __noinline__ float df_discriminate_zero(float2 b, const DfParams &p, const float *__restrict__ hist, const float *__restrict__ x, long k, unsigned D)
{
    const long q = p.first + k * (long)D;                  // stream position ([M-1 history | chunk]) of the first tap's sample
    float re = 0.f, im = 0.f;
    for (int j = 0; j < p.M; j++) {
        float2 sv = make_float2(stream_at<2>(hist, x, q + j, 0, p.M, p.n), stream_at<2>(hist, x, q + j, 1, p.M, p.n));
        if (p.rot_step_fx) sv = rotate_sample(sv, p.rot_step_fx, p.rot_count0 + (uint64_t)(q + j - (p.M - 1)));
        if (p.taps_complex) {
            const float hr = p.taps_rev[2 * j], hi = p.taps_rev[2 * j + 1];
            re = fmaf(sv.x, hr, re); re = fmaf(sv.y, -hi, re);
            im = fmaf(sv.x, hi, im); im = fmaf(sv.y, hr, im);
        } else {
            const float h = p.taps_rev[j];
            re = fmaf(sv.x, h, re); im = fmaf(sv.y, h, im);
        }
    }
    float2 br = b;
    if (p.rot_step_fx && (b.x != 0.f || b.y != 0.f)) br = rotate_sample(b, p.rot_step_fx, p.rot_count0 + (uint64_t)(q - D));
    return discriminate(make_float2(re, im), br, p.inv_gain);
}
__device__ __forceinline__ float df_discriminate(float2 a, float2 b, const DfParams &p, const float *__restrict__ hist, const float *__restrict__ x, long k, unsigned D)
{
    const float tr = fmaf(a.x, b.x, a.y * b.y), ti = fmaf(a.y, b.x, -a.x * b.y);
    if (__builtin_expect(tr == 0.f && ti == 0.f, 0)) return df_discriminate_zero(b, p, hist, x, k, D);
    const float ry = fmaf(tr, p.cD.y, ti * p.cD.x), rx = fmaf(tr, p.cD.x, -ti * p.cD.y);
    if (__builtin_expect(fmaxf(fabsf(rx), fabsf(ry)) < 0x1p-60f, 0)) return fast_atan2f_tiny(ry, rx) * (float)p.inv_gain;      // v_rcp_f32 flushes denormals
    return fast_atan2f(ry, rx) * (float)p.inv_gain;
}

// one sample of the stream [.. zeros | M-1 history | chunk | zeros ..] by its x index, branch-free (edge blocks only)
__device__ __forceinline__ cf df_sample_edge(const float2 *__restrict__ hc, const float2 *__restrict__ xc, long xi, int M, long n)
{
    const bool inx = xi >= 0 && xi < n, inh = xi < 0 && xi >= -(long)(M - 1);
    const float2 *src = inx ? xc + xi : hc + (inh ? xi + (M - 1) : 0);
    const float2 v = *src;
    return (inx || inh) ? cf{v.x, v.y} : cf{0.f, 0.f};
}


// EPI 0: ComplexFloat32 out (rotated when p.rot_step_fx != 0).  EPI 1: Float32 out = discriminator of the rotated outputs.
template <int D, int EPI>
__global__ __launch_bounds__(256, 2) void fir_decfft_kernel(const float *__restrict__ hist, const float *__restrict__ x, const float2 *__restrict__ tables,
                                                            float *__restrict__ y, DfParams p, const float2 *__restrict__ disc_prev_in,
                                                            float2 *__restrict__ disc_prev_out, float *__restrict__ hist_out)
{
    constexpr int A = D / 4, C = D % 4;
    constexpr int NV = A > 0 ? A : 1, NC = C > 0 ? C : 1;
    extern __shared__ __attribute__((aligned(16))) float2 fl[];
    cf *flc = reinterpret_cast<cf *>(fl);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, u = lane & 15;
    constexpr int WV = df_wave_elems(D);
    cf *stg = flc + wave * WV, *ex = stg + df_ex_offset(D);
    const cf *Gf = flc + 4 * WV, *Gl = Gf + A * 1024, *rotT = Gl + C * 256;
    const int M = p.M;

    for (int i = tid; i < df_table_elems(D) - 256; i += 256) fl[4 * WV + i] = tables[256 + i];
    cf tw[16];                                               // W_256^(u k): lane constants
#pragma unroll
    for (int k = 0; k < 16; k++) tw[k] = cf_from(tables[k * 16 + u]);
    // history carry: last M-1 raw input samples into the other ping-pong buffer
    if (hist_out && blockIdx.x == 0)
        for (int i = tid; i < (M - 1) * 2; i += 256) hist_out[i] = stream_at<2>(hist, x, p.n + i / 2, i % 2, M, p.n);
    __syncthreads();

    const long nquads = (p.nblocks + 3) / 4;
    const float2 *xc = reinterpret_cast<const float2 *>(x), *hc = reinterpret_cast<const float2 *>(hist);
    // window position j of block b <-> x index block_xs(b) + j;  phase rho sample mm sits at j = D mm + rho
    auto block_xs = [&](long b) { return p.first + (long)D * (b * DF_LO - DF_V) - (D - 1); };
    // ---- register prefetch of one block: pre[c] = window[64 c + lane], coalesced
    constexpr int NL = 4 * D;                              // 8-byte loads per lane per block (256 D samples / 64 lanes)
#ifndef LRHIP_DECFFT_NPRE_DIV
#define LRHIP_DECFFT_NPRE_DIV 2      /* half of a block's loads are requested a block ahead, the rest when it is staged.  Round 4: with all 20 (D = 5) the kernel
                                        needed 40 registers for the prefetch alone and SPILLED 13 dwords per lane at its 256-register cap: 0.1596 against 0.1456 ms
                                        for Decimator(5) on 2^26 samples, three alternations (1: the round-3 kernel; 4: equal to 2) */
#endif
    constexpr int NPRE = (NL <= 24 ? NL : NL / 2) / LRHIP_DECFFT_NPRE_DIV;           // registers spent on the prefetch; the rest of a long window loads late
    cf pre[NPRE];
    bool have = false;               // pre holds the next block (interior blocks only; edge blocks are staged late, below)
    auto prefetch = [&](long b) {
        const long xs = block_xs(b);
        have = b < p.nblocks && xs >= 0 && xs + DF_N * D <= p.n;
        if (have && !(p.dbg & 1)) {
            const float2 *src = xc + xs + lane;
#pragma unroll
            for (int c = 0; c < NPRE; c++) pre[c] = cf_from(src[64 * c]);
        }
    };
    // Which four blocks a wave takes in round r.  Adjacent (LRHIP_DECFFT_INTERLEAVE 0, rounds 1-3): blocks 4 q .. 4 q + 3 of its own quad q, one after the other -
    // the V D = 160 samples two neighbours share are then read a whole block time apart, and with 8 waves x 10 KB per CU in flight (2.6 MB per XCD against a 4 MB L2)
    // they came from HBM twice: FETCH_SIZE 1.13x the algorithmic bytes.  Interleaved (1): the workgroup's sixteen blocks of a round go round-robin over the waves,
    // block 16 R + 4 bq + wave at step bq - neighbours are in flight at the same time, on the same CU (the order of the 1024-point kernel, whose overlap costs 1.03x)
#ifndef LRHIP_DECFFT_INTERLEAVE
#define LRHIP_DECFFT_INTERLEAVE 1
#endif
    auto blk = [&](int r, int bq) {
        const long R = (long)blockIdx.x * p.rounds + r;
        return LRHIP_DECFFT_INTERLEAVE ? R * 16 + 4 * bq + wave : (R * 4 + wave) * 4 + bq;
    };
    (void)nquads;
    prefetch(blk(0, 0));
    for (int r = 0; r < p.rounds; r++) {
        if (blk(r, 0) >= p.nblocks) break;
        float T[4][8];                   // per block: partial sums over the full batches, 8 floats per lane after the row reduction
        cf left[NC][4][4];               // left-over phases: [c][block][cc] = sample mm = 16 (4 g + cc) + u of phase 4A + c
#pragma unroll
        for (int bq = 0; bq < 4; bq++) {
            const long b = blk(r, bq);
            // ---- stage the block's window in natural order.  Blocks past the end of a ragged last quad compute on whatever the
            // window holds: rows never mix blocks and nothing of theirs is stored.
            if (have && !(p.dbg & 2)) {
#pragma unroll
                for (int c = 0; c < NPRE; c++) stg[64 * c + lane] = pre[c];
                if (NPRE < NL) {
                    const float2 *src = xc + block_xs(b) + lane;
#pragma unroll
                    for (int c = NPRE; c < NL; c++) stg[64 * c + lane] = cf_from(src[64 * c]);
                }
            } else if (b < p.nblocks) {
                // edge blocks (touch the carried history, the zeros before it, or the end of the chunk): sample by sample, rolled
                const long xs = block_xs(b);
#pragma nounroll
                for (int c = 0; c < NL; c++) stg[64 * c + lane] = df_sample_edge(hc, xc, xs + 64 * c + lane, M, p.n);
            }
            // ---- polyphase read-back: row g takes phase 4t + g, window position D mm + phase, mm = 16 i + u
            cf cur[NV][16];
            if (!(p.dbg & 2)) {
#pragma unroll
                for (int t = 0; t < A; t++)
#pragma unroll
                    for (int i = 0; i < 16; i++) cur[t][i] = stg[D * (16 * i + u) + 4 * t + g];
#pragma unroll
                for (int c = 0; c < C; c++)
#pragma unroll
                    for (int cc = 0; cc < 4; cc++) left[c][bq][cc] = stg[D * (16 * (4 * g + cc) + u) + 4 * A + c];
            } else {
#pragma unroll
                for (int t = 0; t < A; t++)
#pragma unroll
                    for (int i = 0; i < 16; i++) cur[t][i] = pre[(i + t) % NPRE];
#pragma unroll
                for (int c = 0; c < C; c++)
#pragma unroll
                    for (int cc = 0; cc < 4; cc++) left[c][bq][cc] = pre[(16 + cc + c) % NPRE];
            }
            prefetch(bq < 3 ? blk(r, bq + 1) : (r + 1 < p.rounds ? blk(r + 1, 0) : p.nblocks));
            if (A > 0 && (p.dbg & 4)) {
#pragma unroll
                for (int t2 = 0; t2 < 8; t2++) T[bq][t2] = cur[0][t2].x + cur[0][t2 + 8].y;
            } else if (A > 0) {
                // ---- full batches: row g transforms phase 4t + g
                cf acc[16];
#pragma unroll
                for (int t = 0; t < A; t++) {
                    dft16<1>(cur[t]);
#pragma unroll
                    for (int k = 1; k < 16; k++) cur[t][k] = cmul(cur[t][k], tw[k]);
                    df_transpose(ex, cur[t], g, u);
                    cf G[16];                               // requested before the second dft16, needed after it
#pragma unroll
                    for (int k = 0; k < 16; k++) G[k] = Gf[t * 1024 + k * 64 + lane];
                    __builtin_amdgcn_sched_barrier(0);
                    dft16<1>(cur[t]);
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        const cf pr = cmul(cur[t][k], G[k]);
                        acc[k] = t == 0 ? pr : acc[k] + pr;
                    }
                }
                // ---- sum the four rows (phases): 32 floats -> 8 per lane; row g keeps (Re Z[2t], Re Z[2t+1], Im Z[2t], Im Z[2t+1])[g]
#pragma unroll
                for (int t2 = 0; t2 < 8; t2++) {
                    float a0 = acc[2 * t2].x, c0 = acc[2 * t2].y, a1 = acc[2 * t2 + 1].x, c1 = acc[2 * t2 + 1].y;
                    swap32(a0, c0);
                    swap32(a1, c1);
                    float s0 = a0 + c0, s1 = a1 + c1;
                    swap16(s0, s1);
                    T[bq][t2] = s0 + s1;
                }
            }
        }
        // ---- bring block b's sums to row b
        cf z[16];
        if (A > 0) {
#pragma unroll
            for (int t2 = 0; t2 < 8; t2++) {
                transpose_rows(T[0][t2], T[1][t2], T[2][t2], T[3][t2]);
                z[2 * t2] = cf{T[0][t2], T[2][t2]};
                z[2 * t2 + 1] = cf{T[1][t2], T[3][t2]};
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) z[k] = cf{0.f, 0.f};
        }
        // ---- left-over phases of the four blocks, row b = block b
#pragma unroll
        for (int c = 0; c < C; c++) {
            cf v[16];
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                float r0 = left[c][0][cc].x, r1 = left[c][1][cc].x, r2 = left[c][2][cc].x, r3 = left[c][3][cc].x;
                float i0 = left[c][0][cc].y, i1 = left[c][1][cc].y, i2 = left[c][2][cc].y, i3 = left[c][3][cc].y;
                transpose_rows(r0, r1, r2, r3);
                transpose_rows(i0, i1, i2, i3);
                v[cc] = cf{r0, i0};           // register i = 4 s + cc <- old row s
                v[4 + cc] = cf{r1, i1};
                v[8 + cc] = cf{r2, i2};
                v[12 + cc] = cf{r3, i3};
            }
            dft16<1>(v);
#pragma unroll
            for (int k = 1; k < 16; k++) v[k] = cmul(v[k], tw[k]);
            df_transpose(ex, v, g, u);
            cf G[16];
#pragma unroll
            for (int k = 0; k < 16; k++) G[k] = Gl[c * 256 + k * 16 + u];
            __builtin_amdgcn_sched_barrier(0);
            dft16<1>(v);
#pragma unroll
            for (int k = 0; k < 16; k++) z[k] = z[k] + cmul(v[k], G[k]);
        }
        // ---- inverse: row g = the wave's block bq = g of this round; z[i] = output window position w = 16 i + u
        if (!(p.dbg & 8)) df_fft_inv(ex, tw, z, g, u);
        const long b = blk(r, g);
        const long k0 = b * DF_LO - DF_V;            // output index of window position 0
        if (b < p.nblocks) {
            if (EPI == 0) {
                cf base = cf{1.f, 0.f};
                if (p.rot_step_fx) base = phasor_poly(p.rot_step_fx * (p.rot_count0 + (uint64_t)(p.first + (long)D * k0)));
                float2 *yo = reinterpret_cast<float2 *>(y);
#pragma unroll
                for (int i = DF_V / 16; i < 16; i++) {
                    const long k = k0 + 16 * i + u;
                    cf o = z[i];
                    if (p.rot_step_fx) o = cmul(o, cmul(base, rotT[16 * i + u]));
                    if (k < p.n_out && (!(p.dbg & 32) || o.x == 12345.678f)) nt_store(yo + k, cf_to(o));
                }
            } else {
#pragma unroll
                for (int i = DF_V / 16; i < 16; i++) {
                    const long k = k0 + 16 * i + u;
                    float2 prev = make_float2(row_prev(z[i].x, z[i - 1].x), row_prev(z[i].y, z[i - 1].y));
                    if (k == 0) prev = *disc_prev_in;
                    const float2 cur = cf_to(z[i]);
                    if (k < p.n_out) {
                        const float dv = (p.dbg & 16) ? cur.x + prev.y : df_discriminate(cur, prev, p, hist, x, k, D);
                        if (!(p.dbg & 32) || dv == 12345.678f) y[k] = dv;
                        if (k == p.n_out - 1) *disc_prev_out = cur;
                    }
                }
            }
        }
    }
}

}  // namespace lrhip
