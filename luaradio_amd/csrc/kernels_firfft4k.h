// kernels_firfft4k.h - overlap-save FIRFilterBlock (radio/blocks/signal/firfilter.lua:320-398) for 513 .. 1281 taps on a ComplexFloat32 stream as ONE launch:
// a 4096-point block per WORKGROUP (round 3).
//
// Why: the 1024-point kernel of kernels_firfft.h takes at most 512 taps per pass, so a longer filter was 2-3 partition launches that re-read and re-write
// the output (24 B per sample and pass) - and the reference suite's first entry, five 256-tap filters back to back (benchmarks/luaradio_benchmark.lua:17-37),
// was five launches moving 80 B per sample.  A chain of filters is one filter (h1 * h2 * ...: 5 x 255 + 1 = 1 276 taps, lrhip_chain_create merges them when
// every one of them asked for the overlap-save arithmetic), and 1 276 taps fit a 4096-point block with a hop of 2 816.
//
// 4096 = 4 x 1024: the four waves of a workgroup each run the 1024-point register / LDS pipeline of kernels_firfft.h on one residue class of the spectrum,
//   X[w + 4 k'] = FFT1024( y_w )[k'],   y_w[n0] = W_4096^(n0 w) * sum_m x[n0 + 1024 m] W_4^(m w)        (decimation in frequency, radix 4 ACROSS the waves)
// so the only new code is the radix-4 stage through LDS in front (every wave loads one quarter of the window, 16 coalesced rows, and reads the four
// quarters of its own positions back) and its mirror image behind the inverse pipelines, where the 2 816 kept outputs are dealt evenly to the 256 threads
// (11 each; the quarter a position falls in is a compile-time constant per register).  H (1/N folded in) sits in 16 register pairs per lane - a lane multiplies
// the same 16 bins in every block - and the cross-stage twiddle W_4096^(n0 w) = W_4096^(t w) * W_64^(i w) is one lane constant times a wave-uniform table.
// The cross-stage buffer and the four per-wave exchange buffers are the same 35 KB of LDS (five barriers per block); 44 KB per workgroup = 3 per CU.
//
// Accuracy: as the 1024-point kernel (Float32 FFT arithmetic), two more butterfly levels; tests hold it to the reference's 1e-6 against the f64 oracle.
#pragma once
#include "kernels_firfft.h"

#ifndef LRHIP_F4K_PREFETCH
#define LRHIP_F4K_PREFETCH 1
#endif

namespace lrhip {

static_assert(LRHIP_FFT_SPLIT == 0, "the 4096-point kernel reuses the one-pass exchange buffers");
constexpr int F4K_N = 4096;
// LDS (float2 units): [NG x (cross-stage buffer (4096) = 4 per-wave exchange buffers) | tw1 16x64 | tw2 64 | c 4x16], NG = blocks per workgroup (4 waves each)
__host__ __device__ constexpr int f4k_lds_tw1(int ng) { return ng * 4 * FFT_EX_ELEMS; }
__host__ __device__ constexpr int f4k_lds_elems(int ng) { return f4k_lds_tw1(ng) + 16 * 64 + 64 + 64; }
static_assert(4 * FFT_EX_ELEMS >= F4K_N, "cross-stage buffer");
// host tables (float2 units): tw1 16x64 | tw2 64 | c[w][i] = W_64^(i w) (4x16) | b[w][t] = W_4096^(t w) (4x64) | H[w][r * 64 + lane] (4x1024)
constexpr int F4K_TAB_LDS = 16 * 64 + 64 + 64;
constexpr int F4K_TAB_B = F4K_TAB_LDS;
constexpr int F4K_TAB_H = F4K_TAB_B + 4 * 64;
constexpr int F4K_TABLE_ELEMS = F4K_TAB_H + F4K_N;

// NG = 2: two blocks per 512-thread workgroup share the tables: 2 x 78 KB = 16 waves per CU instead of 3 x 44 KB = 12.  xcd_map: workgroup g (on XCD g % 8)
// takes the blocks of the (g % 8)-th eighth of the stream, so that the 31 % window overlap of neighbouring blocks is an L2 hit instead of an HBM read.
template <int V, int NG>
__global__ __launch_bounds__(256 * NG, NG == 2 ? 2 : 3) void fir_fft4k_kernel(const float *__restrict__ hist, const float *__restrict__ x,
                                                                              const float2 *__restrict__ tables, float *__restrict__ y, int M, long n, long n_out,
                                                                              long nblocks, float *__restrict__ hist_out, int xcd_map)
{
    constexpr int F4K_LDS_TW1 = f4k_lds_tw1(NG), F4K_LDS_TW2 = F4K_LDS_TW1 + 16 * 64, F4K_LDS_C = F4K_LDS_TW2 + 64;
    static_assert(V % 256 == 0 && V >= 256 && V < F4K_N, "the overlap is a whole number of 256-sample rows");
    constexpr int L = F4K_N - V, NJ = L / 256;
    extern __shared__ __attribute__((aligned(16))) float2 fl[];
    const int lane = threadIdx.x & 63, tid = threadIdx.x & 255;      // tid: inside the block's four waves
    const int wave = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3), grp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    if (hist_out && blockIdx.x == 0)
        for (int i = threadIdx.x; i < (M - 1) * 2; i += 256 * NG) hist_out[i] = stream_at<2>(hist, x, n + i / 2, i % 2, M, n);
    cf *flc = reinterpret_cast<cf *>(fl);
    cf *xb = flc + grp * 4 * FFT_EX_ELEMS, *ex = xb + wave * FFT_EX_ELEMS;
    const cf *tw1 = flc + F4K_LDS_TW1, *tw2 = flc + F4K_LDS_TW2, *ctab = flc + F4K_LDS_C + 16 * wave;
    for (int i = threadIdx.x; i < F4K_TAB_LDS; i += 256 * NG) fl[F4K_LDS_TW1 + i] = tables[i];
    const cf *tb = reinterpret_cast<const cf *>(tables);
    const cf bw = tb[F4K_TAB_B + 64 * wave + lane];
    cf Hreg[16];
#pragma unroll
    for (int r = 0; r < 16; r++) Hreg[r] = tb[F4K_TAB_H + 1024 * wave + 64 * r + lane];
    __syncthreads();
    const int sub = lane & 3, k1s = lane >> 2;

    // the next block's 16 loads are issued while this block is transformed (108 registers without them: room for 32 more at 3 workgroups per CU)
    cf pre[16];
    bool have = false;
    auto prefetch = [&](long fb) {
        const long xlo = fb * L - V;
        have = fb < nblocks && xlo >= 0 && xlo + F4K_N <= n;
        if (have) {
            const cf *src = reinterpret_cast<const cf *>(x) + xlo + 1024 * wave + lane;
#pragma unroll
            for (int i = 0; i < 16; i++) pre[i] = src[64 * i];
        }
    };
    // block order: slot = workgroup-major (plain) or XCD-major; a slot is NG adjacent blocks
    const long nslots = (nblocks + NG - 1) / NG;
    long slot0 = blockIdx.x, sstep = gridDim.x, send = nslots;
    if (xcd_map && (gridDim.x & 7) == 0) {
        const long per = (nslots + 7) / 8;
        slot0 = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
        sstep = gridDim.x >> 3;
        send = (long)((blockIdx.x & 7) + 1) * per < nslots ? (long)((blockIdx.x & 7) + 1) * per : nslots;
    }
    prefetch(slot0 < send ? slot0 * NG + grp : nblocks);
    // (every wave of the workgroup runs the same number of iterations - the barriers are workgroup-wide; a group without a block computes on zeros)
    for (long slot = slot0; slot < send; slot += sstep) {
        const long fb = slot * NG + grp;
        const bool live = fb < nblocks;
        const long xlo = fb * L - V;                      // x index of window position 0
        cf v[16];
        // ---- this wave's quarter of the window: positions 1024 wave + 64 i + lane
        if (have) {
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = pre[i];
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const long p = xlo + 1024 * wave + 64 * i + lane + (M - 1);
                v[i] = cf{stream_at<2>(hist, x, p, 0, M, n), stream_at<2>(hist, x, p, 1, M, n)};
            }
        }
        // ---- forward radix 4 across the waves: y_w[n0] = W_4096^(n0 w) sum_m x[n0 + 1024 m] (-j)^(m w)
#pragma unroll
        for (int i = 0; i < 16; i++) xb[1024 * wave + 64 * i + lane] = v[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int n0 = 64 * i + lane;
            const cf x0 = xb[n0], x1 = xb[n0 + 1024], x2 = xb[n0 + 2048], x3 = xb[n0 + 3072];
            if (wave & 1) {
                const cf e = csub(x0, x2), o = csub(x1, x3);
                v[i] = wave == 1 ? sub_j(e, o) : add_j(e, o);
            } else {
                const cf e = cadd(x0, x2), o = cadd(x1, x3);
                v[i] = wave == 0 ? cadd(e, o) : csub(e, o);
            }
        }
        if (wave) {
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = cmul(v[i], cmul(bw, ctab[i]));
        }
        __syncthreads();                                  // the window is read: its space becomes the four exchange buffers
        if (LRHIP_F4K_PREFETCH && slot + sstep < send) prefetch((slot + sstep) * NG + grp);
        else have = false;
        // ---- the 1024-point pipeline of fir_fft_kernel on y_w (lane t holds y_w[64 i + t])
        dft16<1>(v);
#pragma unroll
        for (int k = 1; k < 16; k++) v[k] = cmul(v[k], tw1[k * 64 + lane]);
        exchange(ex, v, [&](int k) { return k * FFT_E1_ROW + lane; }, [&](int i) { return k1s * FFT_E1_ROW + 4 * i + sub; });
        dft16<1>(v);
#pragma unroll
        for (int k = 1; k < 16; k++) v[k] = cmul(v[k], tw2[k * 4 + sub]);
        exchange(ex, v, [&](int k) { return k1s * FFT_E2_ROW + 17 * sub + k; }, [&](int r) { return k1s * FFT_E2_ROW + 17 * (r & 3) + (r & 12) + sub; });
#pragma unroll
        for (int j = 0; j < 4; j++) {
            radix4<1>(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
#pragma unroll
            for (int k3 = 0; k3 < 4; k3++) v[4 * j + k3] = cmul(v[4 * j + k3], Hreg[4 * j + k3]);
            radix4<-1>(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
#pragma unroll
            for (int t2 = 1; t2 < 4; t2++) v[4 * j + t2] = cmulc(v[4 * j + t2], tw2[(4 * j + sub) * 4 + t2]);
        }
        exchange(ex, v, [&](int r) { return k1s * FFT_E2_ROW + 17 * (r & 3) + (r & 12) + sub; }, [&](int k) { return k1s * FFT_E2_ROW + 17 * sub + k; });
        dft16<-1>(v);
        exchange(ex, v, [&](int i) { return k1s * FFT_E1_ROW + 4 * i + sub; }, [&](int k) { return k * FFT_E1_ROW + lane; });
#pragma unroll
        for (int k = 1; k < 16; k++) v[k] = cmulc(v[k], tw1[k * 64 + lane]);
        dft16<-1>(v);
        // ---- inverse radix 4 across the waves: out[n0 + 1024 m] = sum_w z_w[n0] W_4096^(-n0 w) (+j)^(m w)
        if (wave) {
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = cmulc(v[i], cmul(bw, ctab[i]));
        }
        __syncthreads();                                  // every wave is done with its exchange buffer
#pragma unroll
        for (int i = 0; i < 16; i++) xb[1024 * wave + 64 * i + lane] = v[i];
        __syncthreads();
        // the L kept positions V .. 4095, 256 per pass: position V + 256 j + tid lies in quarter m = (V + 256 j) >> 10 for every thread
        const long ob = fb * L + tid;
        static_for<NJ>([&](auto J) {
            constexpr int j = decltype(J)::value, p0 = V + 256 * j, m = p0 >> 10;
            const int n0 = (p0 & 1023) + tid;
            const cf z0 = xb[n0], z1 = xb[n0 + 1024], z2 = xb[n0 + 2048], z3 = xb[n0 + 3072];
            cf o;
            if constexpr (m & 1) {
                const cf e = csub(z0, z2), d = csub(z1, z3);
                o = m == 1 ? add_j(e, d) : sub_j(e, d);
            } else {
                const cf e = cadd(z0, z2), d = cadd(z1, z3);
                o = m == 0 ? cadd(e, d) : csub(e, d);
            }
            if (live && ob + 256 * j < n_out) __builtin_nontemporal_store(o, reinterpret_cast<cf *>(y) + ob + 256 * j);
        });
        __syncthreads();                                  // the outputs are read: the next window may be written
    }
}

// (Round 3 also had a ONE-WAVE-per-block form of this decomposition, fir_fft4kw_kernel: four 1024-point pipelines one after the other in the same wave, 390
// registers, 0-8 % ahead of the form above.  Round 4 replaced it by the 64 x 64 decomposition of kernels_firfft64.h - two in-register 64-point transforms
// and ONE transpose per direction instead of eight exchanges per pipeline - which is what large launches now run; the workgroup-per-block form above stays
// for launches too small to fill the chip with one workgroup per CU.)

}  // namespace lrhip
