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
// The transpose goes through a HALF-size buffer (32 rows x 65: 16.6 KB per wave instead of 33 KB) in two passes: all lanes write the 32 registers of one half,
// lanes 0..31 (then 32..63) read their whole row - reads at half the lanes, the LDS pipe moves the same bytes - so that four waves + H (32 KB) + tables fit the
// 160 KB of a CU: one 256-thread workgroup per CU, one wave per SIMD with up to 512 registers; latency is hidden by instruction-level parallelism inside the
// wave (64 independent butterflies per stage) and by loading the next block's window while this one is transformed.
#pragma once
#include "kernels_firfft4k.h"

namespace lrhip {

constexpr int F64_ROW = 65;                                   // transpose buffer row (float2 units): odd -> the 32 lanes of a row read hit 32 bank pairs
constexpr int F64_EX = 32 * F64_ROW;                          // per wave
// LDS map (float2 units): [4 x transpose buffer | C 16x64 = W_1024^(t c) | D 4x64 = W_4096^(t d) | H 64x64]
constexpr int F64_LDS_C = 4 * F64_EX;
constexpr int F64_LDS_D = F64_LDS_C + 16 * 64;
constexpr int F64_LDS_H = F64_LDS_D + 4 * 64;
constexpr int F64_LDS_ELEMS = F64_LDS_H + F4K_N;
// host table (float2 units): [C 16x64 | D 4x64 | H[r][l] = H(64 k1(r) + l) / 4096, r = 16 d + c <-> k1 = d + 4 c]
constexpr int F64_TAB_D = 16 * 64;
constexpr int F64_TAB_H = F64_TAB_D + 4 * 64;
constexpr int F64_TABLE_ELEMS = F64_TAB_H + F4K_N;

// cos / sin of 2 pi m / 64
__device__ constexpr float F64_COS[64] = {
    1.000000000e+00f, 9.951847267e-01f, 9.807852804e-01f, 9.569403357e-01f, 9.238795325e-01f, 8.819212643e-01f, 8.314696123e-01f, 7.730104534e-01f,
    7.071067812e-01f, 6.343932842e-01f, 5.555702330e-01f, 4.713967368e-01f, 3.826834324e-01f, 2.902846773e-01f, 1.950903220e-01f, 9.801714033e-02f,
    0.0f, -9.801714033e-02f, -1.950903220e-01f, -2.902846773e-01f, -3.826834324e-01f, -4.713967368e-01f, -5.555702330e-01f, -6.343932842e-01f,
    -7.071067812e-01f, -7.730104534e-01f, -8.314696123e-01f, -8.819212643e-01f, -9.238795325e-01f, -9.569403357e-01f, -9.807852804e-01f, -9.951847267e-01f,
    -1.000000000e+00f, -9.951847267e-01f, -9.807852804e-01f, -9.569403357e-01f, -9.238795325e-01f, -8.819212643e-01f, -8.314696123e-01f, -7.730104534e-01f,
    -7.071067812e-01f, -6.343932842e-01f, -5.555702330e-01f, -4.713967368e-01f, -3.826834324e-01f, -2.902846773e-01f, -1.950903220e-01f, -9.801714033e-02f,
    0.0f, 9.801714033e-02f, 1.950903220e-01f, 2.902846773e-01f, 3.826834324e-01f, 4.713967368e-01f, 5.555702330e-01f, 6.343932842e-01f,
    7.071067812e-01f, 7.730104534e-01f, 8.314696123e-01f, 8.819212643e-01f, 9.238795325e-01f, 9.569403357e-01f, 9.807852804e-01f, 9.951847267e-01f};

template <int M> __device__ constexpr float f64_cos() { return F64_COS[M & 63]; }
template <int M> __device__ constexpr float f64_sin() { return F64_COS[(M + 48) & 63]; }      // sin(x) = cos(x - pi/2)

// 64-point DFT in registers.  Forward (DIR = 1): input index n = 16 a + b in register 16 a + b, output X[d + 4 c] in register 16 d + c.
// Inverse (DIR = -1): input X[d + 4 c] in register 16 d + c, output index n = 16 a + b in register 16 a + b.  No scaling.
template <int DIR>
__device__ __forceinline__ void dft64(cf (&v)[64])
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

template <int V>
__global__ __launch_bounds__(256, 1) void fir_fft64_kernel(const float *__restrict__ hist, const float *__restrict__ x, const float2 *__restrict__ tables,
                                                           float *__restrict__ y, int M, long n, long n_out, long nblocks, float *__restrict__ hist_out, int xcd_map)
{
    static_assert(V % 64 == 0 && V >= 64 && V < F4K_N, "the overlap is a whole number of 64-sample rows");
    constexpr int L = F4K_N - V;
    extern __shared__ __attribute__((aligned(16))) float2 fl[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (hist_out && blockIdx.x == 0)
        for (int i = tid; i < (M - 1) * 2; i += 256) hist_out[i] = stream_at<2>(hist, x, n + i / 2, i % 2, M, n);
    cf *flc = reinterpret_cast<cf *>(fl);
    cf *ex = flc + wave * F64_EX;
    const cf *Ct = flc + F64_LDS_C, *Hs = flc + F64_LDS_H;
    for (int i = tid; i < F64_TABLE_ELEMS; i += 256) fl[F64_LDS_C + i] = tables[i];
    __syncthreads();
    const cf D1 = flc[F64_LDS_D + 64 + lane], D2 = flc[F64_LDS_D + 128 + lane], D3 = flc[F64_LDS_D + 192 + lane];      // W_4096^(lane d)
    const bool lo_half = lane < 32;

    // big twiddle W_4096^(t k2), k2 = d + 4 c in register 16 d + c: D[d][t] * C[c][t]
    auto twiddle = [&](cf (&v)[64], auto conj) {
        constexpr bool CJ = decltype(conj)::value;
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

    const long nslots = (nblocks + 3) / 4;
    long slot0 = blockIdx.x, sstep = gridDim.x, send = nslots;
    if (xcd_map && (gridDim.x & 7) == 0) {
        const long per = (nslots + 7) / 8;
        slot0 = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
        sstep = gridDim.x >> 3;
        send = (long)((blockIdx.x & 7) + 1) * per < nslots ? (long)((blockIdx.x & 7) + 1) * per : nslots;
    }
    for (long slot = slot0; slot < send; slot += sstep) {
        const long fb = slot * 4 + wave;
        if (fb >= nblocks) continue;                         // no workgroup barrier inside the loop: a wave may skip
        const long xlo = fb * L - V;
        cf v[64];
        if (xlo >= 0 && xlo + F4K_N <= n) {
            const cf *src = reinterpret_cast<const cf *>(x) + xlo;
#pragma unroll
            for (int i = 0; i < 64; i++) v[i] = (src + 64 * i)[(unsigned)lane];
        } else {
#pragma unroll
            for (int i = 0; i < 64; i++) {
                const long p = xlo + 64 * i + lane + (M - 1);
                v[i] = cf{stream_at<2>(hist, x, p, 0, M, n), stream_at<2>(hist, x, p, 1, M, n)};
            }
        }
        // ---- forward: DFT over i, twiddle, transpose, DFT over t
        dft64<1>(v);
        twiddle(v, std::false_type{});
        cf z[64];
        // pass A: rows k2 = 0..31 (registers with c < 8), read by lanes 0..31; pass B: rows 32..63, lanes 32..63
#pragma unroll
        for (int r = 0; r < 64; r++)
            if (f64_index(r) < 32) ex[f64_index(r) * F64_ROW + lane] = v[r];
        if (lo_half) {
#pragma unroll
            for (int t = 0; t < 64; t++) z[t] = ex[lane * F64_ROW + t];
        }
#pragma unroll
        for (int r = 0; r < 64; r++)
            if (f64_index(r) >= 32) ex[(f64_index(r) - 32) * F64_ROW + lane] = v[r];
        if (!lo_half) {
#pragma unroll
            for (int t = 0; t < 64; t++) z[t] = ex[(lane - 32) * F64_ROW + t];
        }
        dft64<1>(z);
        // ---- x H (1 / N folded in): register r of lane l holds X[64 k1(r) + l]
#pragma unroll
        for (int r = 0; r < 64; r++) z[r] = cmul(z[r], Hs[r * 64 + lane]);
        // ---- inverse: IDFT over k1 -> t, transpose back, conjugate twiddle, IDFT over k2 -> i
        dft64<-1>(z);
        if (lo_half) {
#pragma unroll
            for (int t = 0; t < 64; t++) ex[lane * F64_ROW + t] = z[t];
        }
#pragma unroll
        for (int r = 0; r < 64; r++)
            if (f64_index(r) < 32) v[r] = ex[f64_index(r) * F64_ROW + lane];
        if (!lo_half) {
#pragma unroll
            for (int t = 0; t < 64; t++) ex[(lane - 32) * F64_ROW + t] = z[t];
        }
#pragma unroll
        for (int r = 0; r < 64; r++)
            if (f64_index(r) >= 32) v[r] = ex[(f64_index(r) - 32) * F64_ROW + lane];
        twiddle(v, std::true_type{});
        dft64<-1>(v);
        // ---- rows at or behind the overlap are this block's outputs
        const long ob = fb * L - V;
        cf *dst = reinterpret_cast<cf *>(y) + ob;
#pragma unroll
        for (int i = 0; i < 64; i++) {
            constexpr int dummy = 0;
            (void)dummy;
            const int p = 64 * i;
            if (p >= V && ob + p + lane < n_out) __builtin_nontemporal_store(v[i], (dst + p) + (unsigned)lane);
        }
    }
}

}  // namespace lrhip
