// ab_split.hip - PROTOTYPE (round 6, stand-alone): a decimating FIR (DecimatorBlock(5): 128 real taps, ComplexFloat32 stream, decimation 5 -
// radio/composites/decimator.lua:28-42) as a banded-Toeplitz product on the bf16 matrix instruction with every Float32 operand SPLIT into three bf16 pieces
// (x = x1 + x2 + x3 exactly: 3 x 8 mantissa bits) and the six products x1h1, x1h2, x2h1, x1h3, x2h2, x3h1 accumulated in Float32 - the dropped terms are below
// 2^-24 of the product, i.e. Float32 rounding.  v_mfma_f32_16x16x32_bf16 runs at ~15 x the rate of the exact-f32 v_mfma_f32_16x16x4_f32 the direct form uses
// (tools/mb_mfma_f16_coexec.hip), so six of them per product still cut the matrix time 2.3 x; the question this driver answers is whether the kernel then
// reaches the memory system's pace.   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gpurun_scratch/ab_split tools/ab_split.hip ; run: ab_split [log2n]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(c) do { hipError_t e_ = (c); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #c, hipGetErrorString(e_)); exit(1); } } while (0)

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int D = 5, M = 128, OW = 512, KS = 7;            // outputs per workgroup tile (4 waves x 8 blocks x 16), K steps of 32 window positions
constexpr int WIN = D * OW + 32 * KS + 8;                   // staged window positions per tile (2 560 new + the Toeplitz span + alignment slack)
constexpr int PLANE = ((WIN + 15) / 16) * 16 + 8;           // bf16 elements per plane; (PLANE * 2) % 32 == 16: the im plane's 16-byte words fall between the re plane's
static_assert((PLANE * 2) % 32 == 16, "plane stride");
constexpr int NF4 = (WIN / 2 + 255) / 256;                  // float4 (two samples) per thread and tile

__device__ __forceinline__ unsigned hi16(float v) { return __float_as_uint(v) & 0xffff0000u; }

// three bf16 pieces of two Float32 values packed as (a | b << 16) per piece
__device__ __forceinline__ void split2(float a, float b, unsigned &p1, unsigned &p2, unsigned &p3)
{
    const unsigned a1 = hi16(a), b1 = hi16(b);
    const float ar = a - __uint_as_float(a1), br = b - __uint_as_float(b1);
    const unsigned a2 = hi16(ar), b2 = hi16(br);
    const float ar2 = ar - __uint_as_float(a2), br2 = br - __uint_as_float(b2);
    p1 = (a1 >> 16) | b1;
    p2 = (a2 >> 16) | b2;
    p3 = (hi16(ar2) >> 16) | hi16(br2);
}

__global__ __launch_bounds__(256, 3) void fir_split_decim_kernel(const float *__restrict__ x, const float *__restrict__ taps_rev, float *__restrict__ y, long n, long n_out,
                                                              long ntiles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];      // planes [piece 3][component 2][PLANE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = lane & 15, kg = lane >> 4;
    // output k of the chunk reads inputs 5 k - 127 .. 5 k; a tile's window starts at the aligned position P0 = 8 floor((2560 t - 127) / 8): slack e = (2560 t - 127) - P0 = 1
    // for every tile (2 560 is a multiple of 8)
    constexpr int E = 1;
    // Toeplitz fragments of this lane, all steps, three pieces: A_s[m][k] = taps_rev[32 s + k - E - 5 m], m = row, k = 8 kg + j
    bf16x8 A[KS][3];
#pragma unroll
    for (int s = 0; s < KS; s++) {
        unsigned p[3][4];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const int i0 = 32 * s + 8 * kg + j - E - D * row, i1 = i0 + 1;
            const float t0 = (i0 >= 0 && i0 < M) ? taps_rev[i0] : 0.f, t1 = (i1 >= 0 && i1 < M) ? taps_rev[i1] : 0.f;
            split2(t0, t1, p[0][j / 2], p[1][j / 2], p[2][j / 2]);
        }
#pragma unroll
        for (int q = 0; q < 3; q++) {
            u32x4 v = {p[q][0], p[q][1], p[q][2], p[q][3]};
            A[s][q] = __builtin_bit_cast(bf16x8, v);
        }
    }
    float4 pre[NF4];
    auto prefetch = [&](long t) {
        if (t >= ntiles) return;
        const long p0 = 2560L * t - 128;                   // position of LDS element 0 (= 8 floor((2560 t - 127) / 8))
#pragma unroll
        for (int k = 0; k < NF4; k++) {
            const long f = tid + 256L * k, pos = p0 + 2 * f;
            pre[k] = (pos >= 0 && pos + 1 < n && 2 * f < WIN) ? reinterpret_cast<const float4 *>(x)[pos / 2] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    prefetch(blockIdx.x);
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        // ---- stage: two samples per thread and word -> six packed bf16 words (piece x component)
#pragma unroll
        for (int k = 0; k < NF4; k++) {
            const int f = tid + 256 * k;
            if (2 * f < WIN) {
                unsigned r1, r2, r3, i1, i2, i3;
                split2(pre[k].x, pre[k].z, r1, r2, r3);
                split2(pre[k].y, pre[k].w, i1, i2, i3);
                unsigned *w = reinterpret_cast<unsigned *>(lds);
                w[(0 * PLANE) / 2 + f] = r1; w[(1 * PLANE) / 2 + f] = i1;
                w[(2 * PLANE) / 2 + f] = r2; w[(3 * PLANE) / 2 + f] = i2;
                w[(4 * PLANE) / 2 + f] = r3; w[(5 * PLANE) / 2 + f] = i3;
            }
        }
        __syncthreads();
        prefetch(t + gridDim.x);
        // ---- 6 x 7 matrix instructions: column n = (block b = n >> 1, component c = n & 1) of this wave's 8 blocks of 16 outputs
        const int b = row >> 1, c = row & 1;
        const unsigned short *base = lds + c * PLANE + 640 * wave + 80 * b + 8 * kg;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const bf16x8 B1 = *reinterpret_cast<const bf16x8 *>(base + 0 * PLANE + 32 * s);
            const bf16x8 B2 = *reinterpret_cast<const bf16x8 *>(base + 2 * PLANE + 32 * s);
            const bf16x8 B3 = *reinterpret_cast<const bf16x8 *>(base + 4 * PLANE + 32 * s);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][0], B1, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][1], B1, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][2], B1, acc2, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][0], B2, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][1], B2, acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][0], B3, acc2, 0, 0, 0);
        }
        f32x4 acc = acc0 + (acc1 + acc2);
        // ---- store: the lane holds rows 4 kg .. 4 kg + 3 of column (b, c); the neighbour lane (c ^ 1) holds the other component: trade two values, store 16 bytes
        const float s0 = c ? acc[0] : acc[2], s1 = c ? acc[1] : acc[3];
        const float g0 = __shfl_xor(s0, 1), g1 = __shfl_xor(s1, 1);
        const long o = OW * t + 128 * wave + 16 * b + 4 * kg + (c ? 2 : 0);
        const float4 out = c ? make_float4(g0, acc[2], g1, acc[3]) : make_float4(acc[0], g0, acc[1], g1);
        if (o + 1 < n_out) { const f32x4 ov = {out.x, out.y, out.z, out.w}; __builtin_nontemporal_store(ov, reinterpret_cast<f32x4 *>(y) + o / 2); }
        else if (o < n_out) { y[2 * o] = out.x; y[2 * o + 1] = out.y; }
        __syncthreads();
    }
}

int main(int argc, char **argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 26, iters = argc > 2 ? atoi(argv[2]) : 20;
    const long n = 1L << log2n, n_out = (n + D - 1) / D;
    std::mt19937 rng(5);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<float> taps(M), xh((size_t)n * 2);
    double g = 0;
    for (int i = 0; i < M; i++) { double m = i - (M - 1) / 2.0, s = m == 0 ? 0.2 : std::sin(M_PI * 0.2 * m) / (M_PI * m); taps[i] = (float)(s * (0.54 - 0.46 * std::cos(2 * M_PI * i / (M - 1)))); g += taps[i]; }
    for (auto &t : taps) t = (float)(t / g);
    for (auto &v : xh) v = U(rng);
    std::vector<float> trev(M);
    for (int i = 0; i < M; i++) trev[i] = taps[M - 1 - i];
    float *x, *y, *tr;
    CK(hipMalloc(&x, xh.size() * 4)); CK(hipMalloc(&y, (size_t)n_out * 8 + 64)); CK(hipMalloc(&tr, M * 4));
    CK(hipMemcpy(x, xh.data(), xh.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(tr, trev.data(), M * 4, hipMemcpyHostToDevice));
    const size_t lds = (size_t)6 * PLANE * 2;
    CK(hipFuncSetAttribute((const void *)fir_split_decim_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int dev = 0, cus = 0, per = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, fir_split_decim_kernel, 256, lds));
    const long ntiles = (n_out + OW - 1) / OW;
    const unsigned grid = (unsigned)std::min<long>(ntiles, (long)cus * per);
    auto go = [&]() { hipLaunchKernelGGL(fir_split_decim_kernel, dim3(grid), dim3(256), lds, 0, x, tr, y, n, n_out, ntiles); };
    for (int i = 0; i < 30; i++) go();
    CK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; i++) go();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= iters;
    std::vector<float> yh((size_t)n_out * 2);
    CK(hipMemcpy(yh.data(), y, yh.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, worst_f32 = 0;
    std::vector<long> pos;
    for (long i = 0; i < 600 && i < n_out; i++) { pos.push_back(i); pos.push_back(n_out - 1 - i); }
    for (int i = 0; i < 6000; i++) pos.push_back((long)((double)i / 6000 * (n_out - 1)));
    for (long k : pos)
        for (int c = 0; c < 2; c++) {
            double acc = 0; float accf = 0.f;
            for (int m = 0; m < M; m++) { const long p = D * k - m; if (p >= 0) { acc += (double)taps[m] * xh[(size_t)p * 2 + c]; } }
            for (int m = M - 1; m >= 0; m--) { const long p = D * k - m; if (p >= 0) accf = fmaf(taps[m], xh[(size_t)p * 2 + c], accf); }
            worst = std::max(worst, std::fabs(acc - (double)yh[(size_t)k * 2 + c]));
            worst_f32 = std::max(worst_f32, std::fabs(acc - (double)accf));
        }
    printf("split-bf16 Decimator(5), 128 taps, n=2^%d: %.4f ms  %.1f GS/s in  %.2f TB/s algorithmic (%.3f of 8 TB/s)  workgroups/CU %d  max err vs f64 %.3g (an f32 fmaf chain: %.3g) %s\n",
           log2n, ms, n / ms / 1e6, 9.6 * n / ms / 1e9, 9.6 * n / ms / 1e9 / 8.0, per, worst, worst_f32, worst < 1e-6 ? "OK" : "FAIL");
    return worst < 1e-6 ? 0 : 1;
}
