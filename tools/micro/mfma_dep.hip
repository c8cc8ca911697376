// Issue rate of v_mfma_f32_32x32x16_bf16 for dependent chains: 24 MFMAs per iteration as (a) 4 rotating accumulators,
// (b) groups of 6 on the same accumulator (the order of gemm_split_kernel), (c) one accumulator; 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(512) void k(int n, float* out) {
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    bf16x8 x = {}, y = {};
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(float)(threadIdx.x + i); y[i] = (__bf16)(float)(i + 1); }
    for (int it = 0; it < n; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 6; ++u) a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 6; ++u) a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 6; ++u) a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 6; ++u) a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 24; ++u) a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
template <int MODE>
static float run(int threads, int n, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, n, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, n, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int n = 2000;
    for (int r = 0; r < 2; ++r) {
        printf("1 wave/SIMD : rotating x4 %.1f us | groups of 6 dependent %.1f us | 24 dependent %.1f us\n", run<0>(256, n, out), run<1>(256, n, out), run<2>(256, n, out));
        printf("2 waves/SIMD: rotating x4 %.1f us | groups of 6 dependent %.1f us | 24 dependent %.1f us  (twice the MFMAs)\n", run<0>(512, n, out), run<1>(512, n, out), run<2>(512, n, out));
    }
    return 0;
}
