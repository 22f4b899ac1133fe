// micro-benchmark (round 6): does the f16 matrix instruction of gfx950 (v_mfma_f32_16x16x32_f16) run CONCURRENTLY with f32 vector work issued by other waves of the
// same SIMD - which the f32 matrix instruction (v_mfma_f32_16x16x4_f32) does not?  Eight waves per workgroup: waves 0-3 issue only MFMAs,
// waves 4-7 only v_pk_fma_f32; each half alone, then both.   hipcc --offload-arch=gfx950 -O3 -o mb_mfma_f16_coexec tools/mb_mfma_f16_coexec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int KIND>      // 0: f16 MFMA 16x16x32, 1: f32 MFMA 16x16x4
__global__ __launch_bounds__(512) void k(float *out, int iters, int mode)      // mode bit 0: the MFMA half works, bit 1: the VALU half works
{
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            f32x4 acc[4];
            for (int i = 0; i < 4; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            f16x8 a, b;
            for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x * 1e-3f + i); b[i] = (_Float16)(1.0f + threadIdx.x * 1e-4f); }
            const float af = threadIdx.x * 1e-3f, bf = 1.0f + threadIdx.x * 1e-4f;
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
                    else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[i], 0, 0, 0);
                }
            }
            for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        }
    } else if (mode & 2) {
        f32x2 acc[8], x = {1.0001f, 0.9999f}, y = {threadIdx.x * 1e-6f, 1e-7f};
        for (int i = 0; i < 8; i++) acc[i] = (f32x2){(float)i, 1.f};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(x), "v"(y));
        }
        for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int KIND>
static void run(const char *name, float *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2048, grid = 1024;
    for (int mode = 1; mode <= 3; mode++) {
        float best = 1e9;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(512), 0, 0, out, iters, mode);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        const double mf = (double)grid * 4 * iters * 4 * (KIND == 0 ? 16384.0 : 2048.0), vf = (double)grid * 4 * 64 * iters * 32 * 4.0;
        printf("%s mode %d (%s): %.3f ms  matrix %.1f TFLOP/s  vector %.1f TFLOP/s\n", name, mode, mode == 1 ? "MFMA waves only" : mode == 2 ? "VALU waves only" : "both", best,
               (mode & 1) ? mf / best / 1e9 : 0.0, (mode & 2) ? vf / best / 1e9 : 0.0);
    }
}
int main()
{
    float *out;
    hipMalloc(&out, 1024 * 512 * sizeof(float));
    run<0>("f16 16x16x32", out);
    run<1>("f32 16x16x4 ", out);
    return 0;
}
