// kernels_channelizer.h - critically sampled K-channel analysis filterbank as ONE dense GEMM on the f32 matrix cores
// (BASELINE.json configs[4]; SURVEY.md section 8d "C5").
//
// Not a block of the reference: it is defined here as K parallel reference chains
//     FrequencyTranslatorBlock(-c*fs/K) -> FIRFilterBlock(h) -> DownsamplerBlock(K),   c = 0..K-1
// (frequencytranslator.lua:93-110, firfilter.lua:230-305, downsampler.lua:45-56), which is also its oracle.
// Because exp(-j*2*pi*c*(mK)/K) = 1, output frame m is
//     y_c[m] = sum_{i<M} g_c[i] * s[q_m + i],   g_c[i] = h[M-1-i] * exp(+j*2*pi*c*(M-1-i)/K),   q_m = first + m*K
// i.e.  Y[T x 2K] = Z[T x 2M] * W[2M x 2K]  over the interleaved float stream z (row m of Z is the 2M-float window of
// frame m, rows overlap by 2(M-K) floats - an implicit Hankel operand read straight out of LDS), with
//     W[2i][2c] = Re g, W[2i+1][2c] = -Im g, W[2i][2c+1] = Im g, W[2i+1][2c+1] = Re g.
// 8*M flop per input sample (8192 for M = 1024): MFMA bound by construction; the point of this kernel is matrix-core
// utilisation, an FFT polyphase form would be ~50x cheaper.
//
// Workgroup = CHAN_MT frames x all 2K columns; wave w owns CHAN_RT row tiles of 16 frames and CHAN_RT * 2K/16 accumulators
// (v_mfma_f32_16x16x4_f32: A = data rows, B = W).  K dimension in slabs of 64: the next W slab is fetched from L2 into
// registers while the current one is multiplied, then written to the other LDS buffer (W is 2M x 2K floats = 1 MiB
// for M = 1024, K = 64: L2 resident).  LDS: data rows of 2K floats padded by 2 (bank = 2*frame + k), W rows padded
// from 2K to 2K+16 floats (bank = col + 16*k): both operand reads are conflict-free.
#pragma once
#include "common.h"
#include "kernels_fir.h"

#ifndef LRHIP_CHAN_EXP
#define LRHIP_CHAN_EXP 0
#endif

namespace lrhip {

// Round 5: one row tile per wave and W slabs of 32 k values - 41 KB of data + 2 x 18.5 KB of W = 78 KB, TWO workgroups per CU (two waves per SIMD), where rounds
// 1-4 ran two row tiles per wave and slabs of 64 (147 KB, one workgroup per CU).  Same box, alternating: 1.317 / 1.310 -> 1.272 / 1.272 ms (0.665 -> 0.687 of the
// f32 matrix peak); LRHIP_CHAN_RT=2 LRHIP_CHAN_KSLAB=64 is the old shape.
#ifndef LRHIP_CHAN_INTERLEAVE
#define LRHIP_CHAN_INTERLEAVE 1      /* same box, three alternations: 1.239 / 1.251 / 1.244 -> 1.209 / 1.216 / 1.190 ms */
#endif
#ifndef LRHIP_CHAN_RT
#define LRHIP_CHAN_RT 1
#endif
#ifndef LRHIP_CHAN_KSLAB
#define LRHIP_CHAN_KSLAB 32
#endif
constexpr int CHAN_RT = LRHIP_CHAN_RT;         // 16-frame row tiles per wave
constexpr int CHAN_MT = 64 * CHAN_RT;   // frames per workgroup
constexpr int CHAN_KSLAB = LRHIP_CHAN_KSLAB;     // k values (floats of the window) per W slab

template <int NCT>      // number of 16-column tiles: 2K = 16*NCT
__global__ __launch_bounds__(256, CHAN_RT == 1 ? 2 : 1) void channelizer_kernel(const float *__restrict__ hist, const float *__restrict__ x,
                                                             const float *__restrict__ W, float *__restrict__ y,
                                                             int M, long n, long nframes, long first)
{
    constexpr int K2 = 16 * NCT;                 // floats per frame hop (= 2K)
    constexpr int K = K2 / 2;
    constexpr int DROW = K2 + 2;                 // padded data row
    constexpr int WROW = K2 + 16;                // padded W row
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const long f0 = (long)blockIdx.x * CHAN_MT;
    const int nsamp = (CHAN_MT - 1) * K + M;     // stream samples of this tile
    const int nflt = 2 * nsamp;
    float *ldsD = lds;
    const int dsize = ((nflt + 2 * (nflt / K2) + 2 + 3) / 4) * 4;
    float *ldsW = lds + dsize;                   // two slabs of CHAN_KSLAB x WROW

    // ---- stage the tile's samples (stream position p0 .. p0 + nsamp), padded rows
    const long p0 = first + f0 * K;
    const long xlo = p0 - (M - 1);
    if (xlo >= 0 && xlo + nsamp <= n && ((reinterpret_cast<uintptr_t>(x + 2 * xlo) & 15) == 0)) {
        const float4 *src = reinterpret_cast<const float4 *>(x + 2 * xlo);
        for (int i4 = tid; i4 < nflt / 4; i4 += 256) {
            float4 v = src[i4];
            int a = 4 * i4, p = a + 2 * (a / K2);
            *reinterpret_cast<float2 *>(ldsD + p) = make_float2(v.x, v.y);
            *reinterpret_cast<float2 *>(ldsD + p + 2) = make_float2(v.z, v.w);
        }
        for (int a = (nflt / 4) * 4 + tid; a < nflt; a += 256) ldsD[a + 2 * (a / K2)] = x[2 * xlo + a];
    } else {
        for (int r = tid; r < nsamp; r += 256) {
            int a = 2 * r, p = a + 2 * (a / K2);
            ldsD[p] = stream_at<2>(hist, x, p0 + r, 0, M, n);
            ldsD[p + 1] = stream_at<2>(hist, x, p0 + r, 1, M, n);
        }
    }

    // ---- first W slab
    constexpr int SLAB_F4 = CHAN_KSLAB * K2 / 4;           // float4 per slab
    constexpr int UW = (SLAB_F4 + 255) / 256;
    const float4 *W4 = reinterpret_cast<const float4 *>(W);
    // The prefetched slab lives in eight NAMED float4 registers: as an array (`float4 wreg[UW]`, however it was indexed) hipcc
    // kept it in scratch memory and waited for the whole prefetch at the top of every slab - the matrix pipe idled 48 %.
    static_assert(UW <= 8, "W slab prefetch: at most 8 float4 per thread");
    float4 w0, w1, w2, w3, w4, w5, w6, w7;
#define LR_CHAN_FOR8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define LR_CHAN_LOAD(U)                                                                          \
    if (U < UW) {                                                                                \
        int i4 = tid + U * 256;                                                                  \
        w##U = W4[wbase + (i4 < SLAB_F4 ? i4 : SLAB_F4 - 1)];                                    \
    }
#define LR_CHAN_PUT(U)                                                                           \
    if (U < UW) {                                                                                \
        int i4 = tid + U * 256;                                                                  \
        if (i4 < SLAB_F4) *reinterpret_cast<float4 *>(wdst + ((4 * i4) / K2) * WROW + (4 * i4) % K2) = w##U; \
    }
    {
        const size_t wbase = 0;
        float *wdst = ldsW;
        LR_CHAN_FOR8(LR_CHAN_LOAD)
        LR_CHAN_FOR8(LR_CHAN_PUT)
    }
    __syncthreads();

    f32x4 acc[CHAN_RT][NCT];
#pragma unroll
    for (int r = 0; r < CHAN_RT; r++)
#pragma unroll
        for (int c = 0; c < NCT; c++) acc[r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float *abase = ldsD + DROW * (wave * 16 * CHAN_RT + col) + kq;     // row (frame) = lane & 15 of this wave's first tile
    const int nslabs = 2 * M / CHAN_KSLAB;
    for (int s = 0; s < nslabs; s++) {
        const float *wcur = ldsW + (s & 1) * (CHAN_KSLAB * WROW);
#if LRHIP_CHAN_EXP == 1
        const bool more = false;      /* experiment: no W streaming at all (wrong results) */
#else
        const bool more = s + 1 < nslabs;
#endif
        if (more) {
            const size_t wbase = (size_t)(s + 1) * SLAB_F4;
            LR_CHAN_FOR8(LR_CHAN_LOAD)
        }
        // window float j = 64*s + 4*st + kq lives at padded offset j + 2*(j / K2); j / K2 is constant inside a slab
        const float *ap = abase + CHAN_KSLAB * s + 2 * ((CHAN_KSLAB * s) / K2);
        const float *bp = wcur + kq * WROW + col;
        // explicit one-step software pipeline: the fragments of step st+1 are loaded (and fenced with sched_barrier, or
        // hipcc sinks each ds_read next to its MFMA and waits on it at once) while step st is multiplied
        constexpr int NST = CHAN_KSLAB / 4;
        float av[2][CHAN_RT], bv[2][NCT];
        auto fetch = [&](int buf, int st) {
#pragma unroll
            for (int r = 0; r < CHAN_RT; r++) av[buf][r] = ap[16 * r * DROW + 4 * st];
#pragma unroll
            for (int c = 0; c < NCT; c++) bv[buf][c] = bp[4 * st * WROW + 16 * c];
        };
        fetch(0, 0);
#pragma unroll
        for (int st = 0; st < NST; st++) {
#if LRHIP_CHAN_INTERLEAVE
            // A/B (round 5): the next step's fragment reads issued BETWEEN this step's MFMAs (one DS read behind each) instead of in front of them
            if (st + 1 < NST) fetch((st + 1) & 1, st + 1);
#pragma unroll
            for (int c = 0; c < NCT; c++)
#pragma unroll
                for (int r = 0; r < CHAN_RT; r++)
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st & 1][r], bv[st & 1][c], acc[r][c], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NCT * CHAN_RT; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one DS read
            }
            __builtin_amdgcn_sched_barrier(0);
#else
            if (st + 1 < NST) fetch((st + 1) & 1, st + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < NCT; c++)
#pragma unroll
                for (int r = 0; r < CHAN_RT; r++)
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st & 1][r], bv[st & 1][c], acc[r][c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        if (more) {
            float *wdst = ldsW + ((s + 1) & 1) * (CHAN_KSLAB * WROW);
            LR_CHAN_FOR8(LR_CHAN_PUT)
        }
#if LRHIP_CHAN_EXP != 1
        __syncthreads();
#endif
    }

    // ---- store: lane (col, kq) holds frames 4kq..4kq+3 of column 16c + col; output layout [frame][channel] cf32
#pragma unroll
    for (int rt = 0; rt < CHAN_RT; rt++)
#pragma unroll
        for (int c = 0; c < NCT; c++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                long f = f0 + (wave * CHAN_RT + rt) * 16 + 4 * kq + r;
                if (f < nframes) y[f * K2 + 16 * c + col] = acc[rt][c][r];
            }
}

#undef LR_CHAN_FOR8
#undef LR_CHAN_LOAD
#undef LR_CHAN_PUT

}  // namespace lrhip
