// kernels_agc.h - AGCBlock (radio/blocks/signal/agc.lua:45-96) as parallel scans.
//
//   P[n] = (1 - ap) P[n-1] + ap |x[n]|^2                                   average power (a Lua double)
//   G[n] = P[n] >= thr ? (1 - ag) G[n-1] + ag * target / P[n] : G[n-1]     filtered gain, frozen below the threshold
//   y[n] = P[n] >= thr ? sqrt(G[n]) * x[n] : x[n]
//
// It reads like a feedback loop but both recurrences are first-order LINEAR in their state with coefficients that depend
// only on the input (G's on P, which is known once P's scan is done): state' = A[n] * state + B[n].  Affine maps compose
// associatively, (A2, B2) o (A1, B1) = (A2 A1, A2 B1 + B2), so each recurrence is a prefix scan.  With the default
// power_tau = 1 s the power estimator remembers ~10^6 samples: no warm-up trick, a real three-level scan
// (thread chunk -> workgroup tile -> tile carries), in double like the reference's Lua numbers:
//   pass 0: tile maps of P           -> carry kernel -> P at every tile start
//   pass 1: true P, tile maps of G   -> carry kernel -> G at every tile start
//   pass 2: true P, true G, output, final (P, G) for the next call.
// PowerSquelchBlock is the first scan alone followed by the gate (pass 3).
// Traffic: 3 reads + 1 write of the stream.
#pragma once
#include "common.h"

namespace lrhip {

constexpr int AGC_LC = 8, AGC_TILE = 256 * AGC_LC;

struct AgcParams { double ap, ag, target, thr; };

// exclusive scan of affine maps over the 256 threads of a workgroup (Hillis-Steele in LDS): on return (A, B) is the
// composition of the maps of threads 0 .. tid-1 (identity for thread 0) and (At, Bt) the composition of all 256.
__device__ __forceinline__ void scan_affine_excl(double &A, double &B, double &At, double &Bt, double (*sh)[256][2])
{
    const int tid = threadIdx.x;
    int buf = 0;
    sh[0][tid][0] = A; sh[0][tid][1] = B;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        double a = sh[buf][tid][0], b = sh[buf][tid][1];
        if (tid >= off) {
            const double pa = sh[buf][tid - off][0], pb = sh[buf][tid - off][1];     // earlier maps apply first
            b = fma(a, pb, b);
            a = a * pa;
        }
        sh[buf ^ 1][tid][0] = a; sh[buf ^ 1][tid][1] = b;
        buf ^= 1;
        __syncthreads();
    }
    At = sh[buf][255][0]; Bt = sh[buf][255][1];
    if (tid) { A = sh[buf][tid - 1][0]; B = sh[buf][tid - 1][1]; }
    else { A = 1.0; B = 0.0; }
    __syncthreads();
}

template <int S, int PASS>
__global__ __launch_bounds__(256) void agc_pass_kernel(const float *__restrict__ x, float *__restrict__ y, unsigned long n, AgcParams p,
                                                       double *__restrict__ mapsP, double *__restrict__ mapsG,
                                                       const double *__restrict__ startP, const double *__restrict__ startG,
                                                       double *__restrict__ state_out)
{
    __shared__ double sh[2][256][2];
    const int tid = threadIdx.x;
    const unsigned long c0 = (unsigned long)blockIdx.x * AGC_TILE + (unsigned long)tid * AGC_LC;
    float xv[AGC_LC][S];
    double e[AGC_LC];
#pragma unroll
    for (int i = 0; i < AGC_LC; i++) {
        const bool in = c0 + i < n;
#pragma unroll
        for (int c = 0; c < S; c++) xv[i][c] = in ? x[(c0 + i) * S + c] : 0.f;
        // real: value*value in double (agc.lua:50); complex: abs_squared() of the Float32 pair in double (complexfloat32.lua)
        e[i] = S == 1 ? (double)xv[i][0] * (double)xv[i][0] : (double)xv[i][0] * (double)xv[i][0] + (double)xv[i][S - 1] * (double)xv[i][S - 1];
    }
    const double cp = 1.0 - p.ap, cg = 1.0 - p.ag;
    // ---- P: chunk map (cp^LC over the samples inside the stream, zero-state end), scan, chunk start state
    double Ap = 1.0, Bp = 0.0;
#pragma unroll
    for (int i = 0; i < AGC_LC; i++)
        if (c0 + i < n) { Bp = cp * Bp + p.ap * e[i]; Ap *= cp; }
    double Apt, Bpt;
    scan_affine_excl(Ap, Bp, Apt, Bpt, sh);
    if (PASS == 0) {
        if (tid == 0) { mapsP[2 * blockIdx.x] = Apt; mapsP[2 * blockIdx.x + 1] = Bpt; }
        return;
    }
    double P = fma(Ap, startP[blockIdx.x], Bp);          // P just before this thread's chunk
    if (PASS == 3) {
        // PowerSquelchBlock (radio/blocks/signal/powersquelch.lua:45-80): the same power estimator, the sample passes or is zeroed
#pragma unroll
        for (int i = 0; i < AGC_LC; i++)
            if (c0 + i < n) {
                P = cp * P + p.ap * e[i];
#pragma unroll
                for (int c = 0; c < S; c++) y[(c0 + i) * S + c] = P >= p.thr ? xv[i][c] : 0.f;
                if (c0 + i == n - 1) { state_out[0] = P; state_out[1] = 0.0; }
            }
        return;
    }
    // ---- G: per-sample maps from the true P
    double Pn[AGC_LC], Ag = 1.0, Bg = 0.0;
#pragma unroll
    for (int i = 0; i < AGC_LC; i++) {
        if (c0 + i < n) {
            P = cp * P + p.ap * e[i];                    // agc.lua:50 operation order
            if (P >= p.thr) { Bg = cg * Bg + p.ag * (p.target * (1.0 / P)); Ag *= cg; }
        }
        Pn[i] = P;
    }
    double Agt, Bgt;
    scan_affine_excl(Ag, Bg, Agt, Bgt, sh);
    if (PASS == 1) {
        if (tid == 0) { mapsG[2 * blockIdx.x] = Agt; mapsG[2 * blockIdx.x + 1] = Bgt; }
        return;
    }
    double Gs = fma(Ag, startG[blockIdx.x], Bg);         // G just before this thread's chunk
#pragma unroll
    for (int i = 0; i < AGC_LC; i++) {
        if (c0 + i < n) {
            const bool on = Pn[i] >= p.thr;
            if (on) Gs = cg * Gs + p.ag * (p.target * (1.0 / Pn[i]));      // agc.lua:54
            const double g = on ? sqrt(Gs) : 1.0;
#pragma unroll
            for (int c = 0; c < S; c++) y[(c0 + i) * S + c] = on ? (float)(g * (double)xv[i][c]) : xv[i][c];     // :56 / :59
            if (c0 + i == n - 1) { state_out[0] = Pn[i]; state_out[1] = Gs; }
        }
    }
}

// tile carries: s[t+1] = A_t s[t] + B_t from s[0] = *s0; one workgroup, threads own contiguous segments of tiles
__global__ __launch_bounds__(256) void agc_carry_kernel(const double *__restrict__ maps, unsigned long ntiles, const double *__restrict__ s0,
                                                        double *__restrict__ start)
{
    __shared__ double sh[2][256][2];
    const int tid = threadIdx.x;
    const unsigned long seg = (ntiles + 255) / 256, t0 = tid * seg, t1 = t0 + seg < ntiles ? t0 + seg : ntiles;
    double A = 1.0, B = 0.0;
    for (unsigned long t = t0; t < t1; t++) { B = fma(maps[2 * t], B, maps[2 * t + 1]); A *= maps[2 * t]; }
    double At, Bt;
    scan_affine_excl(A, B, At, Bt, sh);
    double s = fma(A, *s0, B);
    for (unsigned long t = t0; t < t1; t++) {
        start[t] = s;
        s = fma(maps[2 * t], s, maps[2 * t + 1]);
    }
}

}  // namespace lrhip
