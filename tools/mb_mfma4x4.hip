// micro-benchmark: sustained rate of v_mfma_f32_4x4x1_16B_f32 against v_mfma_f32_16x16x4_f32 (both 64 flop/clk/SIMD on paper), 8 independent
// accumulator chains per wave, 1024 waves x 4096 iterations.   hipcc --offload-arch=gfx950 -O3 -o mb_mfma4x4 tools/mb_mfma4x4.hip && ./mb_mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    f32x4 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    float *out;
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, grid = 2048;
    for (int kind = 0; kind < 2; kind++) {
        for (int rep = 0; rep < 3; rep++) {
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flops = (double)grid * 4 * iters * 8 * (kind == 0 ? 512.0 : 2048.0);
            printf("%s: %.3f ms  %.1f TFLOP/s\n", kind == 0 ? "4x4x1_16B" : "16x16x4  ", ms, flops / ms / 1e9);
        }
    }
    return 0;
}
