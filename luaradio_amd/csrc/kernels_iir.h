// kernels_iir.h - IIRFilterBlock (radio/blocks/signal/iirfilter.lua:113-181) on the GPU.
//
//   y[n] = ( sum_{j<nb} b[j] x[n-j]  -  sum_{1<=i<na} a[i] y[n-i] ) / a[0]
//
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
// Orders above 4 use the sequential kernel (one thread per component) - correct, not fast.
#pragma once
#include "common.h"

namespace lrhip {

constexpr int IIR_LC = 16;                    // samples per thread chunk
constexpr int IIR_TILE = 256 * IIR_LC;        // samples per workgroup tile
constexpr int IIR_MAX_NB = 16;
constexpr int IIR_MAX_P = 4;                  // scan path; above this -> sequential kernel
constexpr int IIR_SEQ_MAX = 32;

struct IirCoeffs {
    int nb, P;
    float b[IIR_MAX_NB];                 // b[j]/a0
    float a[IIR_MAX_P];                  // a[i+1]/a0
    float Tpow[9][IIR_MAX_P * IIR_MAX_P];  // A^(LC*2^k), k = 0..8 (k = 8 is the whole-tile transition), row-major PxP
};

// state convention: st[i] = y[n-1-i] (st[0] newest).  One homogeneous step: y = -sum a[i] st[i].
template <int P>
__device__ __forceinline__ void mat_apply(const float *T, const float *v, float *out)
{
#pragma unroll
    for (int r = 0; r < P; r++) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < P; c++) acc = fmaf(T[r * P + c], v[c], acc);
        out[r] = acc;
    }
}

// LDS layout: component plane cpl, chunk c, offset i -> cpl*PLANE + c*(LC+1) + i   (+1: conflict-free column walks)
template <int S, int P, bool FINAL>
__global__ __launch_bounds__(256) void iir_scan_kernel(const float *__restrict__ x, float *__restrict__ y, long n,
                                                       const float *__restrict__ xhist,      // nb-1 samples before x[0]
                                                       const float *__restrict__ tile_start, // FINAL: [tile][S][P]
                                                       float *__restrict__ tile_end,         // !FINAL: [tile][S][P]
                                                       IirCoeffs co)
{
    constexpr int LC = IIR_LC, TILE = IIR_TILE, HALO = IIR_MAX_NB - 1;
    constexpr int PLANE = 256 * (LC + 1) + HALO + 1;
    __shared__ float sx[S * PLANE];          // inputs: [HALO history | tile]; reused for outputs
    __shared__ float sst[2][S][256][P];      // chunk end states (double-buffered scan)

    const int tid = threadIdx.x;
    const long t0 = (long)blockIdx.x * TILE;
    const int nb = co.nb;
    const long cnt = (n - t0) < TILE ? (n - t0) : TILE;

    // stage: history (nb-1 samples before the tile) at plane offsets [HALO-(nb-1), HALO), tile after it
    for (int i = tid; i < (nb - 1) * S; i += 256) {
        int r = i / S, c = i % S;                 // r-th history sample, oldest first
        long g = t0 - (nb - 1) + r;               // global sample index
        float v = g >= 0 ? x[g * S + c] : xhist[(g + (nb - 1)) * S + c];
        sx[c * PLANE + HALO - (nb - 1) + r] = v;
    }
    for (int i = tid; i < TILE * S; i += 256) {
        int r = i / S, c = i % S;
        float v = r < cnt ? x[(t0 + r) * S + c] : 0.f;
        sx[c * PLANE + HALO + (r / LC) * (LC + 1) + (r % LC)] = v;
    }
    __syncthreads();

    // feed-forward part for this thread's chunk, u[i] = sum_j b[j] x[n-j]; previous chunk's tail is at
    // (c-1)*(LC+1) + LC-1-..., i.e. not contiguous because of the +1 pad: fetch through a helper.
    float u[S][LC];
#pragma unroll
    for (int c = 0; c < S; c++) {
        const float *pl = sx + c * PLANE + HALO;
#pragma unroll
        for (int i = 0; i < LC; i++) {
            float acc = 0.f;
            for (int j = 0; j < nb; j++) {
                int r = tid * LC + i - j;          // tile-relative sample index, may be negative (history)
                float xv = r >= 0 ? pl[(r / LC) * (LC + 1) + (r % LC)] : pl[r];
                acc = fmaf(co.b[j], xv, acc);
            }
            u[c][i] = acc;
        }
    }

    // zero-state run over the chunk -> chunk end state
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

    // inclusive Kogge-Stone scan of chunk end states: S_c = z_c + A^LC S_{c-1}  (S_{-1} = tile start state)
    int buf = 0;
#pragma unroll
    for (int c = 0; c < S; c++) {
        if (FINAL && tid == 0) {
            float ts[P], tmp[P];
#pragma unroll
            for (int k = 0; k < P; k++) ts[k] = tile_start[((long)blockIdx.x * S + c) * P + k];
            mat_apply<P>(co.Tpow[0], ts, tmp);
#pragma unroll
            for (int k = 0; k < P; k++) st[c][k] += tmp[k];
        }
#pragma unroll
        for (int k = 0; k < P; k++) sst[0][c][tid][k] = st[c][k];
    }
    __syncthreads();
    for (int lvl = 0; lvl < 8; lvl++) {
        int off = 1 << lvl;
#pragma unroll
        for (int c = 0; c < S; c++) {
            float cur[P];
#pragma unroll
            for (int k = 0; k < P; k++) cur[k] = sst[buf][c][tid][k];
            if (tid >= off) {
                float prev[P], tmp[P];
#pragma unroll
                for (int k = 0; k < P; k++) prev[k] = sst[buf][c][tid - off][k];
                mat_apply<P>(co.Tpow[lvl], prev, tmp);
#pragma unroll
                for (int k = 0; k < P; k++) cur[k] += tmp[k];
            }
#pragma unroll
            for (int k = 0; k < P; k++) sst[buf ^ 1][c][tid][k] = cur[k];
        }
        buf ^= 1;
        __syncthreads();
    }

    if (!FINAL) {
        if (tid == 255)
#pragma unroll
            for (int c = 0; c < S; c++)
#pragma unroll
                for (int k = 0; k < P; k++) tile_end[((long)blockIdx.x * S + c) * P + k] = sst[buf][c][255][k];
        return;
    }

    // true start state of this chunk = scanned end state of the previous chunk (or the tile start state)
#pragma unroll
    for (int c = 0; c < S; c++) {
#pragma unroll
        for (int k = 0; k < P; k++)
            st[c][k] = tid ? sst[buf][c][tid - 1][k] : tile_start[((long)blockIdx.x * S + c) * P + k];
        float *pl = sx + c * PLANE + HALO + tid * (LC + 1);
#pragma unroll
        for (int i = 0; i < LC; i++) {
            float v = u[c][i];
#pragma unroll
            for (int k = 0; k < P; k++) v = fmaf(-co.a[k], st[c][k], v);
#pragma unroll
            for (int k = P - 1; k > 0; k--) st[c][k] = st[c][k - 1];
            st[c][0] = v;
            pl[i] = v;       // every thread only overwrites its own chunk; u[] already holds what it needed
        }
    }
    __syncthreads();
    for (int i = tid; i < TILE * S; i += 256) {
        int r = i / S, c = i % S;
        if (r < cnt) y[(t0 + r) * S + c] = sx[c * PLANE + HALO + (r / LC) * (LC + 1) + (r % LC)];
    }
}

// pass 2: carry across tiles, s_t = E_{t-1} + A^TILE s_{t-1}, as a 256-thread scan: every thread owns a
// contiguous segment of `seg` tiles (sequential inside the segment, twice), segment end states are combined with a
// Kogge-Stone scan using A^(TILE*seg*2^k) (computed on the host per launch in double).
struct IirCarryPowers {
    float Tseg[8][IIR_MAX_P * IIR_MAX_P];
};
template <int S, int P>
__global__ __launch_bounds__(256) void iir_carry_kernel(const float *__restrict__ tile_end, float *__restrict__ tile_start, long ntiles,
                                                        long seg, const float *__restrict__ state_in, IirCoeffs co, IirCarryPowers pw)
{
    __shared__ float sst[2][S][256][P];
    const int tid = threadIdx.x;
    const long t0 = (long)tid * seg, t1 = (t0 + seg < ntiles) ? t0 + seg : ntiles;
    float z[S][P], tmp[P];
#pragma unroll
    for (int c = 0; c < S; c++) {
#pragma unroll
        for (int k = 0; k < P; k++) z[c][k] = 0.f;
        for (long t = t0; t < t1; t++) {
            mat_apply<P>(co.Tpow[8], z[c], tmp);
#pragma unroll
            for (int k = 0; k < P; k++) z[c][k] = tile_end[(t * S + c) * P + k] + tmp[k];
        }
        if (tid == 0) {      // fold the carried state into segment 0: S_0 = z_0 + A^(TILE*seg) * carried
            float ci[P];
#pragma unroll
            for (int k = 0; k < P; k++) ci[k] = state_in[c * P + k];
            mat_apply<P>(pw.Tseg[0], ci, tmp);
#pragma unroll
            for (int k = 0; k < P; k++) z[c][k] += tmp[k];
        }
#pragma unroll
        for (int k = 0; k < P; k++) sst[0][c][tid][k] = z[c][k];
    }
    __syncthreads();
    int buf = 0;
    for (int lvl = 0; lvl < 8; lvl++) {
        int off = 1 << lvl;
#pragma unroll
        for (int c = 0; c < S; c++) {
            float cur[P];
#pragma unroll
            for (int k = 0; k < P; k++) cur[k] = sst[buf][c][tid][k];
            if (tid >= off) {
                float prev[P];
#pragma unroll
                for (int k = 0; k < P; k++) prev[k] = sst[buf][c][tid - off][k];
                mat_apply<P>(pw.Tseg[lvl], prev, tmp);
#pragma unroll
                for (int k = 0; k < P; k++) cur[k] += tmp[k];
            }
#pragma unroll
            for (int k = 0; k < P; k++) sst[buf ^ 1][c][tid][k] = cur[k];
        }
        buf ^= 1;
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < S; c++) {
        float s[P];
#pragma unroll
        for (int k = 0; k < P; k++) s[k] = tid ? sst[buf][c][tid - 1][k] : state_in[c * P + k];
        for (long t = t0; t < t1; t++) {
#pragma unroll
            for (int k = 0; k < P; k++) tile_start[(t * S + c) * P + k] = s[k];
            mat_apply<P>(co.Tpow[8], s, tmp);
#pragma unroll
            for (int k = 0; k < P; k++) s[k] = tile_end[(t * S + c) * P + k] + tmp[k];
        }
    }
}

// carried state after the chunk: y[n-1-i] (zero-extended by the previous state when n < P) and the last nb-1 inputs
template <int S>
__global__ void iir_state_kernel(const float *__restrict__ x, const float *__restrict__ y, long n, int nb, int P,
                                 const float *__restrict__ xhist_in, float *__restrict__ xhist_out,
                                 const float *__restrict__ state_in, float *__restrict__ state_out)
{
    int tid = threadIdx.x;
    for (int i = tid; i < (nb - 1) * S; i += blockDim.x) {
        int r = i / S, c = i % S;                 // r-th oldest of the nb-1 retained inputs
        long g = n - (nb - 1) + r;
        xhist_out[i] = g >= 0 ? x[g * S + c] : xhist_in[(g + (nb - 1)) * S + c];
    }
    for (int i = tid; i < P * S; i += blockDim.x) {
        int c = i / P, k = i % P;                 // state[c][k] = y[n-1-k]
        long g = n - 1 - k;
        state_out[i] = g >= 0 ? y[g * S + c] : state_in[c * P + (int)(-g - 1)];
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
