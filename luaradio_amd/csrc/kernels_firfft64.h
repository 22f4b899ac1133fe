// kernels_firfft64.h - overlap-save FIRFilterBlock (radio/blocks/signal/firfilter.lua:320-398) for 513 .. 1 281 taps on a ComplexFloat32 stream:
// a 4096-point block per WAVE as 64 x 64, both 64-point transforms IN REGISTERS and ONE transpose through LDS per direction (round 4).
//
// Why: the two 4096-point kernels of round 3 (kernels_firfft4k.h) build the block from four 1024-point pipelines - 8 LDS exchanges per pipeline, 36 864
// 8-byte LDS operations per block - and sit at 2.3-2.7 TB/s: one form pays five workgroup barriers per block, the other needs 390 registers (one wave per SIMD)
// and still makes 32 LDS round trips per block with nothing to hide them behind.  The counters of every overlap-save kernel of this library say the same
// thing: LDS traffic and VALU work do not overlap well (VALU 27-42 %, LDS 42-55 % busy), so the lever is fewer LDS operations per output, not more waves.
//
// 4096 = 64 x 64 with a lane holding 64 points (position t + 64 i in register i of lane t - the natural coalesced load):
//   forward : 64-point DFT over i IN REGISTERS (radix 4 x radix 16) -> k2;  x W_4096^(t k2);  TRANSPOSE (lane k2 gets t = 0..63);
//             64-point DFT over t in registers -> k1:  X[64 k1 + k2] in register k1 of lane k2;   x H
//   inverse : the mirror image; the result lands as y[t + 64 i] in register i of lane t: coalesced stores, no bit reversal anywhere.
// LDS operations per block: 2 transposes x 4096 x (write + read) + 4096 H reads = 20 480 (the 4 x 1024 form: 36 864; the partitioned form of
// kernels_firpols.h for the same filter: 119 KB per 512 outputs = 5 x as many bytes per output).  No workgroup barrier: a wave's DS operations execute in order.
//
// The transpose goes through a per-wave buffer of 64 x 65 FLOATS, the real parts first and the imaginary parts behind them through the same 16.6 KB (a wave's
// DS operations execute in order, so the second write pass cannot overtake the first read pass): rows padded to 65 words make the row writes and the column
// reads conflict-free with plain base + immediate addresses.  Four waves (66 KB) + H (32 KB) + the small twiddle tables = 108 KB: one 256-thread workgroup per
// CU, one wave per SIMD with up to 512 registers.  Latency is hidden by instruction-level parallelism inside the wave (64 independent butterflies per stage).
// Two cuts that did not survive, for the record: a half-size ComplexFloat32 buffer in two passes with the reads of a pass under `if (lane < 32)` - the
// inactive half came back with stale registers in a few rows (register traffic to the accumulation registers inside a divergent region); and a full 64 x 64
// ComplexFloat32 buffer with an XOR swizzle instead of padding (exactly 160 KB with H) - correct, but the swizzled addresses are 128 per-lane values that
// lived in the accumulation registers and were fetched back 331 times per block.
#pragma once
#include "kernels_firfft4k.h"

// WAVES per workgroup (= per CU, template parameter): 4 = one per SIMD, H whole in LDS (real or complex taps); 8 = two per SIMD - with ONE wave per SIMD every
// scalar, LDS and memory instruction takes an issue slot the vector ALU could have used (counters: VALU 36 %, nothing else to run) - which needs 8 transpose
// buffers + H + C in 160 KB: REAL taps only, H conjugate-symmetric, columns 0..32 stored (lane l > 32 reads conj H[63 - k1][64 - l])
#ifndef LRHIP_F64_PREFETCH
#define LRHIP_F64_PREFETCH 0      /* measured equal at one wave per SIMD (0.385 against 0.381 ms), impossible at two (128 more registers) */
#endif
#ifndef LRHIP_F64_CARRY
#define LRHIP_F64_CARRY 1      /* two partitions: the half window two consecutive blocks share stays in registers (0: re-read through L2, A/B) */
#endif
#ifndef LRHIP_F64_HGROUP
#define LRHIP_F64_HGROUP 16
#endif

#ifndef LRHIP_F64_NOMEM
#define LRHIP_F64_NOMEM 0      /* 1 (stand-alone driver only, wrong results): no global loads / stores in interior blocks - what the arithmetic alone costs */
#endif
#ifndef LRHIP_F64_STAGGER
#define LRHIP_F64_STAGGER 0
#endif
#ifndef LRHIP_F64_NOARITH
#define LRHIP_F64_NOARITH 0      /* 1 (stand-alone driver only, wrong results): no transform - what the block's access shape alone costs */
#endif
#ifndef LRHIP_F64_FENCES
#define LRHIP_F64_FENCES 0
#endif
#if LRHIP_F64_FENCES
#define F64_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define F64_FENCE() ((void)0)
#endif

namespace lrhip {

constexpr int F64_ROW = 65;                                   // transpose buffer row in FLOATS (one plane at a time): odd -> conflict-free columns
constexpr int F64_EX = (64 * F64_ROW + 1) / 2;                // per wave, in float2 units
// LDS map (float2 units): [4 x transpose buffer | H 64x64 | C 16x64 = W_1024^(t c)]
constexpr int F64_HSYM_ROW = 33;                              // symmetric H storage: [k1][l = 0..32]
__host__ __device__ constexpr int f64_lds_h(int waves) { return waves * F64_EX; }
__host__ __device__ constexpr int f64_lds_c(int waves, int np = 1) { return f64_lds_h(waves) + np * (waves <= 4 ? F4K_N : 64 * F64_HSYM_ROW); }
__host__ __device__ constexpr int f64_lds_elems(int waves, int np = 1) { return f64_lds_c(waves, np) + 16 * 64; }
// host table (float2 units): [C 16x64 | D 4x64 | H[r][l] = H(64 k1(r) + l) / 4096, r = 16 d + c <-> k1 = d + 4 c | Hsym[k1][l = 0..32] = H(64 k1 + l) / 4096]
constexpr int F64_TAB_D = 16 * 64;
constexpr int F64_TAB_H = F64_TAB_D + 4 * 64;
constexpr int F64_TAB_HSYM = F64_TAB_H + F4K_N;
constexpr int F64_TABLE_ELEMS = F64_TAB_HSYM + 64 * F64_HSYM_ROW;
// round 5, two partitions (1 282 .. 4 097 taps at an overlap of 2 048: y_b = IFFT(X_b H_0 + X_(b-1) H_1)): the full H of partition 1 behind the table
constexpr int F64_TAB_H1 = F64_TABLE_ELEMS;
constexpr int F64_TABLE_ELEMS2 = F64_TAB_H1 + F4K_N;      // (round 6: 4 098 .. 8 193 taps = two such sets, partitions (0, 1) and (2, 3), one launch each)

// cos / sin of 2 pi m / 64
constexpr float F64_COS[64] = {
    1.000000000e+00f, 9.951847267e-01f, 9.807852804e-01f, 9.569403357e-01f, 9.238795325e-01f, 8.819212643e-01f, 8.314696123e-01f, 7.730104534e-01f,
    7.071067812e-01f, 6.343932842e-01f, 5.555702330e-01f, 4.713967368e-01f, 3.826834324e-01f, 2.902846773e-01f, 1.950903220e-01f, 9.801714033e-02f,
    0.0f, -9.801714033e-02f, -1.950903220e-01f, -2.902846773e-01f, -3.826834324e-01f, -4.713967368e-01f, -5.555702330e-01f, -6.343932842e-01f,
    -7.071067812e-01f, -7.730104534e-01f, -8.314696123e-01f, -8.819212643e-01f, -9.238795325e-01f, -9.569403357e-01f, -9.807852804e-01f, -9.951847267e-01f,
    -1.000000000e+00f, -9.951847267e-01f, -9.807852804e-01f, -9.569403357e-01f, -9.238795325e-01f, -8.819212643e-01f, -8.314696123e-01f, -7.730104534e-01f,
    -7.071067812e-01f, -6.343932842e-01f, -5.555702330e-01f, -4.713967368e-01f, -3.826834324e-01f, -2.902846773e-01f, -1.950903220e-01f, -9.801714033e-02f,
    0.0f, 9.801714033e-02f, 1.950903220e-01f, 2.902846773e-01f, 3.826834324e-01f, 4.713967368e-01f, 5.555702330e-01f, 6.343932842e-01f,
    7.071067812e-01f, 7.730104534e-01f, 8.314696123e-01f, 8.819212643e-01f, 9.238795325e-01f, 9.569403357e-01f, 9.807852804e-01f, 9.951847267e-01f};

template <int M> __host__ __device__ constexpr float f64_cos() { return F64_COS[M & 63]; }
template <int M> __host__ __device__ constexpr float f64_sin() { return F64_COS[(M + 48) & 63]; }      // sin(x) = cos(x - pi/2)

// 64-point DFT in registers.  Forward (DIR = 1): input index n = 16 a + b in register 16 a + b, output X[d + 4 c] in register 16 d + c.
// Inverse (DIR = -1): input X[d + 4 c] in register 16 d + c, output index n = 16 a + b in register 16 a + b.  No scaling.
template <int DIR>
__host__ __device__ __forceinline__ void dft64(cf (&v)[64])
{
    if constexpr (DIR > 0) {
#pragma unroll
        for (int b = 0; b < 16; b++) radix4<1>(v[b], v[16 + b], v[32 + b], v[48 + b]);      // over a -> d: register 16 d + b
        static_for<3>([&](auto DD) {
            constexpr int d = decltype(DD)::value + 1;
            static_for<15>([&](auto BB) {
                constexpr int b = decltype(BB)::value + 1, m = b * d;                         // W_64^(b d) = cos - j sin
                v[16 * d + b] = cmul_const(v[16 * d + b], f64_cos<m>(), -f64_sin<m>());
            });
        });
#pragma unroll
        for (int d = 0; d < 4; d++) dft16<1>(*reinterpret_cast<cf(*)[16]>(&v[16 * d]));      // over b -> c: register 16 d + c
    } else {
#pragma unroll
        for (int d = 0; d < 4; d++) dft16<-1>(*reinterpret_cast<cf(*)[16]>(&v[16 * d]));     // over c -> b
        static_for<3>([&](auto DD) {
            constexpr int d = decltype(DD)::value + 1;
            static_for<15>([&](auto BB) {
                constexpr int b = decltype(BB)::value + 1, m = b * d;                         // W_64^(-b d) = cos + j sin
                v[16 * d + b] = cmul_const(v[16 * d + b], f64_cos<m>(), f64_sin<m>());
            });
        });
#pragma unroll
        for (int b = 0; b < 16; b++) radix4<-1>(v[b], v[16 + b], v[32 + b], v[48 + b]);     // over d -> a
    }
}

// register 16 d + c of a dft64 spectrum holds index d + 4 c
__host__ __device__ constexpr int f64_index(int r) { return (r >> 4) + 4 * (r & 15); }

// tools/ab_fft64.hip builds this header with -DLRHIP_F64_TRACE: lane 0 of every wave of the first workgroups stamps the phases of its first blocks
#ifdef LRHIP_F64_TRACE
__device__ unsigned long long *lrhip_f64_trace;         // [block 8][wave 8][iteration 16][16]
#define F64_STAMP(i)                                                                                                                                \
    do {                                                                                                                                            \
        if (lrhip_f64_trace && blockIdx.x < 8 && trace_it < 16 && (threadIdx.x & 63) == 0)                                                          \
            lrhip_f64_trace[(((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + trace_it) * 16 + (i)] = clock64();                                  \
    } while (0)
#else
#define F64_STAMP(i) do { } while (0)
#endif

// NP = 2 (round 5): uniformly partitioned overlap-save with TWO partitions of 2 048 taps on the same engine - V = 2 048, hop 2 048, four waves per CU with both
// partitions' H whole in LDS.  A wave walks a RUN of consecutive blocks and keeps the previous block's spectrum in 128 registers (the wave has 512); the
// block in front of its run is transformed once to fill them (one forward transform in ~33 at 2^26 samples).  One launch, every sample read once (+ the
// overlap through L2): 4 096 taps on 2^26 samples in one pass instead of two passes of the 1024-point partitioned kernel.
//
// S = 1 (round 6): a Float32 stream with REAL taps - two stream blocks ride as the real and the imaginary plane of one transform (H of real taps is
// conjugate-symmetric: the planes come back separately), as in fir_fft_kernel<1, .>.  NP = 1: planes = the adjacent stream blocks 2 f, 2 f + 1 and `nblocks`
// counts TRANSFORMS.  NP = 2: a wave walks TWO runs of consecutive blocks at once, run A in the real plane and the run behind it in the imaginary plane, so the
// delayed spectrum in its registers is the previous block's in BOTH planes (adjacent blocks in one transform would need the spectrum of a pair shifted by one
// block); `nblocks` counts stream blocks.  Replaces the partitioned 1024-point kernel for 513 .. 4 097 real taps on Float32 streams (0.33 of the roof).
// HG (round 6, A/B only - measured 8-10 % slower, stage_fir.h): H read from the global table (32 KB, resident in L2) instead of the LDS - COMPLEX taps at eight waves
// per CU: their H has no symmetry to halve it, and the full table next to eight transpose buffers does not fit the LDS (the shipped form: four waves per CU)
template <int V, int F64_WAVES, int NP = 1, int S = 2, bool HG = false>
__global__ __launch_bounds__(64 * F64_WAVES, 1) void fir_fft64_kernel(const float *__restrict__ hist, const float *__restrict__ x, const float2 *__restrict__ tables,
                                                           float *__restrict__ y, int M, long n, long n_out, long nblocks, float *__restrict__ hist_out, int xcd_map,
                                                           long delay, int accumulate)
{
    // delay / accumulate (round 6): 4 098 .. 8 193 taps as TWO launches of the two-partition form - the second applies partitions 2 and 3 to the stream delayed
    // by 4 096 samples (M stays the whole filter's length: the carried history holds M - 1 samples) and adds to y, as fir_fft_kernel's partitions do
    static_assert(V % 64 == 0 && V >= 64 && V < F4K_N, "the overlap is a whole number of 64-sample rows");
    static_assert(NP == 1 || (NP == 2 && F64_WAVES == 4 && 2 * V == F4K_N), "two partitions: hop = overlap = 2 048, four waves, full H");
    static_assert(S == 2 || S == 1, "ComplexFloat32 or Float32 stream");
    constexpr int L = F4K_N - V;
    extern __shared__ __attribute__((aligned(16))) float2 fl[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (hist_out && blockIdx.x == 0)
        for (int i = tid; i < (M - 1) * S; i += 64 * F64_WAVES) hist_out[i] = stream_at<S>(hist, x, n + i / S, i % S, M, n);
    cf *flc = reinterpret_cast<cf *>(fl);
    float *ex = reinterpret_cast<float *>(flc + wave * F64_EX);
    constexpr bool HSYM = F64_WAVES > 4 && !HG;
    static_assert(!HG || (NP == 1 && S == 2), "H from the global table: one partition, ComplexFloat32 stream");
    constexpr int F64_LDS_H = f64_lds_h(F64_WAVES), F64_LDS_C = f64_lds_c(F64_WAVES, NP);
    const cf *Ct = flc + F64_LDS_C, *Hs = flc + F64_LDS_H;
    if (HSYM)
        for (int i = tid; i < 64 * F64_HSYM_ROW; i += 64 * F64_WAVES) fl[F64_LDS_H + i] = tables[F64_TAB_HSYM + i];
    else if (!HG)
        for (int i = tid; i < F4K_N; i += 64 * F64_WAVES) fl[F64_LDS_H + i] = tables[F64_TAB_H + i];
    if (NP == 2)
        for (int i = tid; i < F4K_N; i += 64 * F64_WAVES) fl[F64_LDS_H + F4K_N + i] = tables[F64_TAB_H1 + i];
    for (int i = tid; i < 16 * 64; i += 64 * F64_WAVES) fl[F64_LDS_C + i] = tables[i];
    // symmetric H: element index of H[k1][lane] = hs_a + k1 * hs_s, imaginary part times hs_sgn
    const int hs_a = lane <= 32 ? lane : 63 * F64_HSYM_ROW + (64 - lane), hs_s0 = lane <= 32 ? F64_HSYM_ROW : -F64_HSYM_ROW;
    const cf hs_sgn = cf{1.f, lane <= 32 ? 1.f : -1.f};
    __syncthreads();
    // LRHIP_F64_STAGGER (round 6, A/B of the stand-alone driver; 0 in the library): are the launch's waves in lock step - everybody loads, everybody transforms,
    // everybody stores, so that the memory system and the vector ALUs take turns?  Wave w of a workgroup starts w x STAGGER x 8 128 clocks late (a block is ~60 000
    // clocks at eight waves per CU).  Measured: NO effect at 1, slower at 2 and 4 (profiles/r06_fft64_ablation.txt) - the waves are out of phase by themselves.
    if (LRHIP_F64_STAGGER > 0)
        for (int i = 0; i < wave * LRHIP_F64_STAGGER; i++) __builtin_amdgcn_s_sleep(127);
    const cf *tb = reinterpret_cast<const cf *>(tables);
    const cf D1 = tb[F64_TAB_D + 64 + lane], D2 = tb[F64_TAB_D + 128 + lane], D3 = tb[F64_TAB_D + 192 + lane];      // W_4096^(lane d)

    // big twiddle W_4096^(t k2), k2 = d + 4 c in register 16 d + c: D[d][t] * C[c][t]
    auto twiddle = [&](cf (&v)[64], auto conj) {
        constexpr bool CJ = decltype(conj)::value;
        asm volatile("" ::: "memory");                       // the table reads stay here (hoisted out of the block loop they cost 30 registers)
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const cf cc = Ct[c * 64 + lane];
#pragma unroll
            for (int d = 0; d < 4; d++) {
                if (c == 0 && d == 0) continue;
                cf w = d == 0 ? cc : d == 1 ? D1 : d == 2 ? D2 : D3;
                if (c != 0 && d != 0) w = cmul(w, cc);
                v[16 * d + c] = CJ ? cmulc(v[16 * d + c], w) : cmul(v[16 * d + c], w);
            }
        }
    };

    const long nslots = (nblocks + F64_WAVES - 1) / F64_WAVES;
    long slot0 = blockIdx.x, sstep = gridDim.x, send = nslots;
    if (xcd_map && (gridDim.x & 7) == 0) {
        const long per = (nslots + 7) / 8;
        slot0 = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
        sstep = gridDim.x >> 3;
        send = (long)((blockIdx.x & 7) + 1) * per < nslots ? (long)((blockIdx.x & 7) + 1) * per : nslots;
    }
    // the next block's window is loaded while this block is between its two transposes (one wave per SIMD: nobody else hides the 2-3 us of a round trip to
    // HBM, and the stores of this block hold on to its registers): 128 more registers, which is what the accumulation half of the register file is for here
    cf pre[64];
    bool have = false;
    auto prefetch = [&](long nb) {
        const long plo = nb * L - V - delay;
        have = LRHIP_F64_PREFETCH && S == 2 && nb < nblocks && plo >= 0 && plo + F4K_N <= n;
        if (have) {
            const cf *src = reinterpret_cast<const cf *>(x) + plo;
#pragma unroll
            for (int i = 0; i < 64; i++) pre[i] = (src + 64 * i)[(unsigned)lane];
        }
    };
    // two partitions: this wave's run of consecutive blocks [r0, r1), entered one block early (warm-up: forward transform only)
    [[maybe_unused]] cf zp[NP == 2 ? 64 : 1];
    [[maybe_unused]] cf keep[(NP == 2 && LRHIP_F64_CARRY) ? 32 : 1];
    [[maybe_unused]] bool kept = false;
    // (S = 1: a wave's run is 2 x `run` stream blocks - [r0, r0 + run) in the real plane, [r0 + run, r0 + 2 run) in the imaginary plane)
    const long nwaves = (long)gridDim.x * F64_WAVES, run = (nblocks + nwaves * (3 - S) - 1) / (nwaves * (3 - S));
    const long r0 = ((long)blockIdx.x * F64_WAVES + wave) * run * (3 - S), r1 = r0 + run < nblocks ? r0 + run : nblocks;
    if (NP == 2) { slot0 = r0 - 1; sstep = 1; send = r1; }
    [[maybe_unused]] int trace_it = 0;
    for (long slot = slot0; slot < send; slot += sstep) {
        const long fb = NP == 2 ? slot : slot * F64_WAVES + wave;
        if (fb >= nblocks) continue;                         // no workgroup barrier inside the loop: a wave may skip
        F64_STAMP(0);
        [[maybe_unused]] const bool warm = NP == 2 && fb < r0;
        // stream block(s) of this transform: S = 2 - block fb; S = 1 - blocks (2 fb, 2 fb + 1) or, two partitions, (fb, fb + run)
        const long ba = (S == 1 && NP == 1) ? 2 * fb : fb;
        [[maybe_unused]] const long bb = NP == 1 ? ba + 1 : ba + run;
        const long xlo = ba * L - V - delay;
        [[maybe_unused]] const long xlob = bb * L - V - delay;
        cf v[64];
        if constexpr (S == 1) {
            const long xhi = xlo > xlob ? xlo : xlob;
            if (xlo >= 0 && xlob >= 0 && xhi + F4K_N <= n) {
                const float *sa = x + xlo, *sb = x + xlob;
                if constexpr (NP == 2 && LRHIP_F64_CARRY) {
                    if (kept) {
#pragma unroll
                        for (int i = 0; i < 32; i++) v[i] = keep[i];
                    } else {
#pragma unroll
                        for (int i = 0; i < 32; i++) v[i] = cf{(sa + 64 * i)[(unsigned)lane], (sb + 64 * i)[(unsigned)lane]};
                    }
#pragma unroll
                    for (int i = 32; i < 64; i++) v[i] = cf{(sa + 64 * i)[(unsigned)lane], (sb + 64 * i)[(unsigned)lane]};
                } else {
#pragma unroll
                    for (int i = 0; i < 64; i++) v[i] = cf{(sa + 64 * i)[(unsigned)lane], (sb + 64 * i)[(unsigned)lane]};
                }
            } else {
                // edge transforms (a window reaches into the carried history or past the chunk - the second plane of the last pair): rolled, through the buffer
#pragma unroll 1
                for (int i = 0; i < 64; i++) ex[i * 64 + lane] = stream_at<1>(hist, x, xlo + 64 * i + lane + (M - 1), 0, M, n);
#pragma unroll
                for (int i = 0; i < 64; i++) v[i].x = ex[i * 64 + lane];
#pragma unroll 1
                for (int i = 0; i < 64; i++) ex[i * 64 + lane] = stream_at<1>(hist, x, xlob + 64 * i + lane + (M - 1), 0, M, n);
#pragma unroll
                for (int i = 0; i < 64; i++) v[i].y = ex[i * 64 + lane];
            }
        } else
        if (have) {
#pragma unroll
            for (int i = 0; i < 64; i++) v[i] = pre[i];
        } else if (xlo >= 0 && xlo + F4K_N <= n) {
            const cf *src = reinterpret_cast<const cf *>(x) + xlo;
            if constexpr (NP == 2 && LRHIP_F64_CARRY) {
                // a run of consecutive blocks at a hop of half a window: the first half of this window IS the second half of the previous one - kept in
                // 64 registers (the wave has 512), so every sample of the stream is loaded exactly once
                if (kept) {
#pragma unroll
                    for (int i = 0; i < 32; i++) v[i] = keep[i];
                } else {
#pragma unroll
                    for (int i = 0; i < 32; i++) v[i] = (src + 64 * i)[(unsigned)lane];
                }
#pragma unroll
                for (int i = 32; i < 64; i++) v[i] = (src + 64 * i)[(unsigned)lane];
            } else if constexpr (LRHIP_F64_NOMEM != 0) {
#pragma unroll
                for (int i = 0; i < 64; i++) v[i] = cf{(float)(lane + i) * 1e-3f, (float)(fb & 7)};      // (ablation: no loads)
            } else {
#pragma unroll
                for (int i = 0; i < 64; i++) v[i] = (src + 64 * i)[(unsigned)lane];
            }
        } else {
            // edge blocks (the window reaches into the carried history or past the chunk): staged through the transpose buffer by a ROLLED loop, so that the
            // 64 x 2 guarded loads do not compete for registers with the main path
#pragma unroll 1
            for (int i = 0; i < 64; i++) ex[i * 64 + lane] = stream_at<2>(hist, x, xlo + 64 * i + lane + (M - 1), 0, M, n);
#pragma unroll
            for (int i = 0; i < 64; i++) v[i].x = ex[i * 64 + lane];
#pragma unroll 1
            for (int i = 0; i < 64; i++) ex[i * 64 + lane] = stream_at<2>(hist, x, xlo + 64 * i + lane + (M - 1), 1, M, n);
#pragma unroll
            for (int i = 0; i < 64; i++) v[i].y = ex[i * 64 + lane];
        }
        if constexpr (NP == 2 && LRHIP_F64_CARRY) {
#pragma unroll
            for (int i = 0; i < 32; i++) keep[i] = v[32 + i];
            kept = true;
        }
        // ---- forward: DFT over i, twiddle, transpose, DFT over t
        // (F64_FENCE = scheduling fence between phases: left alone, the scheduler pulls the next phase's 64 loads up to cover their latency and the wave
        // holds 128 + 128 values at the seams; with two waves per SIMD the other wave covers the latency and the registers are worth more)
        if constexpr (!LRHIP_F64_NOARITH) {              // (ablation build of tools/ab_fft64.hip: the block's loads and stores with nothing in between)
        F64_FENCE();
        F64_STAMP(1);
        dft64<1>(v);
        F64_STAMP(2);
        twiddle(v, std::false_type{});
        F64_FENCE();
        F64_STAMP(3);
        cf z[64];
        // transpose: lane t writes row k2 of register 16 d + c (k2 = d + 4 c), lane k2 reads its row (t = 0..63); real parts, then imaginary parts
#pragma unroll
        for (int r = 0; r < 64; r++) ex[f64_index(r) * F64_ROW + lane] = v[r].x;
#pragma unroll
        for (int t = 0; t < 64; t++) z[t].x = ex[lane * F64_ROW + t];
        F64_FENCE();
#pragma unroll
        for (int r = 0; r < 64; r++) ex[f64_index(r) * F64_ROW + lane] = v[r].y;
#pragma unroll
        for (int t = 0; t < 64; t++) z[t].y = ex[lane * F64_ROW + t];
        F64_FENCE();
        F64_STAMP(4);
        if (NP == 1) prefetch((slot + sstep) * F64_WAVES + wave);     // v is dead until the inverse transpose
        dft64<1>(z);
        F64_STAMP(5);
        if constexpr (NP == 2) {
            if (warm) {                                               // wave-uniform
#pragma unroll
                for (int r = 0; r < 64; r++) zp[r] = z[r];
                continue;
            }
        }
        // ---- x H (1 / N folded in): register r of lane l holds X[64 k1(r) + l]
        // (scheduling fences: without them the 64 H reads are hoisted above the transform to cover their latency - 128 more live registers at the point
        // where the wave already holds 128, and the allocator falls back to the accumulation registers and to scratch)
#pragma unroll
        for (int g = 0; g < 64; g += LRHIP_F64_HGROUP) {
            cf h[LRHIP_F64_HGROUP];
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (HSYM) {
                int hs_s = hs_s0;
                asm volatile("" : "+v"(hs_s));              // opaque per group: the 64 lane-dependent addresses are recomputed (one v_mad each), not hoisted out
                                                            // of the block loop into 64 registers the wave does not have
#pragma unroll
                for (int r = 0; r < LRHIP_F64_HGROUP; r++) h[r] = Hs[hs_a + f64_index(g + r) * hs_s] * hs_sgn;
            } else if constexpr (HG) {
                const cf *Hg = reinterpret_cast<const cf *>(tables) + F64_TAB_H;
#pragma unroll
                for (int r = 0; r < LRHIP_F64_HGROUP; r++) h[r] = (Hg + (g + r) * 64)[(unsigned)lane];
            } else {
#pragma unroll
                for (int r = 0; r < LRHIP_F64_HGROUP; r++) h[r] = (Hs + (g + r) * 64)[(unsigned)lane];      // row pointer + the lane's index
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NP == 2) {
                cf h1[LRHIP_F64_HGROUP];
#pragma unroll
                for (int r = 0; r < LRHIP_F64_HGROUP; r++) h1[r] = (Hs + F4K_N + (g + r) * 64)[(unsigned)lane];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < LRHIP_F64_HGROUP; r++) {
                    const cf xb = z[g + r];
                    z[g + r] = cmul(xb, h[r]) + cmul(zp[g + r], h1[r]);
                    zp[g + r] = xb;
                }
            } else {
#pragma unroll
                for (int r = 0; r < LRHIP_F64_HGROUP; r++) z[g + r] = cmul(z[g + r], h[r]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        F64_STAMP(6);
        // ---- inverse: IDFT over k1 -> t, transpose back, conjugate twiddle, IDFT over k2 -> i
        dft64<-1>(z);
        F64_FENCE();
        F64_STAMP(7);
#pragma unroll
        for (int t = 0; t < 64; t++) ex[lane * F64_ROW + t] = z[t].x;
#pragma unroll
        for (int r = 0; r < 64; r++) v[r].x = ex[f64_index(r) * F64_ROW + lane];
        F64_FENCE();
#pragma unroll
        for (int t = 0; t < 64; t++) ex[lane * F64_ROW + t] = z[t].y;
#pragma unroll
        for (int r = 0; r < 64; r++) v[r].y = ex[f64_index(r) * F64_ROW + lane];
        F64_FENCE();
        F64_STAMP(8);
        twiddle(v, std::true_type{});
        F64_STAMP(9);
        dft64<-1>(v);
        }
        F64_STAMP(10);
        // ---- rows at or behind the overlap are this block's outputs
        if constexpr (S == 1) {
            const long oa = ba * L - V, ob1 = bb * L - V;
            float *da = y + oa, *db = y + ob1;
            const long ohi = oa > ob1 ? oa : ob1;
            if (accumulate && V % 1024 == 0 && ohi + F4K_N <= n_out) {
                // groups of eight rows per plane: sixteen loads in flight, then the adds and stores (a load-add-store per row would be a memory round trip per row)
#pragma unroll
                for (int g = V / 64; g + 8 <= 64; g += 8) {
                    float ta[8], tc[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        ta[i] = (da + 64 * (g + i))[(unsigned)lane];
                        tc[i] = (db + 64 * (g + i))[(unsigned)lane];
                    }
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        (da + 64 * (g + i))[(unsigned)lane] = ta[i] + v[g + i].x;
                        (db + 64 * (g + i))[(unsigned)lane] = tc[i] + v[g + i].y;
                    }
                }
            } else if (accumulate) {
#pragma unroll
                for (int i = V / 64; i < 64; i++) {
                    if (oa + 64 * i + lane < n_out) (da + 64 * i)[(unsigned)lane] += v[i].x;
                    if (ob1 + 64 * i + lane < n_out) (db + 64 * i)[(unsigned)lane] += v[i].y;
                }
            } else if (ohi + F4K_N <= n_out) {
#pragma unroll
                for (int i = V / 64; i < 64; i++) {
                    __builtin_nontemporal_store(v[i].x, (da + 64 * i) + (unsigned)lane);
                    __builtin_nontemporal_store(v[i].y, (db + 64 * i) + (unsigned)lane);
                }
            } else {
#pragma unroll
                for (int i = V / 64; i < 64; i++) {
                    if (oa + 64 * i + lane < n_out) __builtin_nontemporal_store(v[i].x, (da + 64 * i) + (unsigned)lane);
                    if (ob1 + 64 * i + lane < n_out) __builtin_nontemporal_store(v[i].y, (db + 64 * i) + (unsigned)lane);
                }
            }
            F64_STAMP(11);
#ifdef LRHIP_F64_TRACE
            trace_it++;
#endif
            continue;
        }
        const long ob = fb * L - V;
        cf *dst = reinterpret_cast<cf *>(y) + ob;
        if (accumulate && V % 1024 == 0 && ob + F4K_N <= n_out) {
#pragma unroll
            for (int g = V / 64; g + 16 <= 64; g += 16) {
                cf t[16];
#pragma unroll
                for (int i = 0; i < 16; i++) t[i] = (dst + 64 * (g + i))[(unsigned)lane];
#pragma unroll
                for (int i = 0; i < 16; i++) (dst + 64 * (g + i))[(unsigned)lane] = t[i] + v[g + i];
            }
        } else if (accumulate) {
#pragma unroll
            for (int i = V / 64; i < 64; i++)
                if (ob + 64 * i + lane < n_out) (dst + 64 * i)[(unsigned)lane] += v[i];
        } else if (LRHIP_F64_NOMEM != 0 && ob + F4K_N <= n_out) {
#pragma unroll
            for (int i = V / 64; i < 64; i++)
                if (v[i].x == 12345.678f) __builtin_nontemporal_store(v[i], (dst + 64 * i) + (unsigned)lane);      // (ablation: no stores, the arithmetic stays alive)
        } else if (ob + F4K_N <= n_out) {
#pragma unroll
            for (int i = V / 64; i < 64; i++) __builtin_nontemporal_store(v[i], (dst + 64 * i) + (unsigned)lane);
        } else {
#pragma unroll
            for (int i = V / 64; i < 64; i++)
                if (ob + 64 * i + lane < n_out) __builtin_nontemporal_store(v[i], (dst + 64 * i) + (unsigned)lane);
        }
        F64_STAMP(11);
#ifdef LRHIP_F64_TRACE
        trace_it++;
#endif
    }
}

}  // namespace lrhip
