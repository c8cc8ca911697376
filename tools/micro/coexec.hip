// Do MFMA-only waves and VALU-only waves on the same SIMD overlap?  One workgroup per CU of 8 waves (2 per SIMD):
// waves 0-3 run `nm` dependent-free MFMA batches, waves 4-7 run `nv` VALU batches.  Timed alone and together.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(int nm, int nv, float* out) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
        bf16x8 x = {}, y = {};
        for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(float)(threadIdx.x + i); y[i] = (__bf16)(float)(i + 1); }
        for (int it = 0; it < nm; ++it) {
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
            }
        }
        if (nm) out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    } else {
        float v0 = threadIdx.x, v1 = 1.5f, v2 = 2.5f, v3 = 3.5f;
        unsigned u0 = threadIdx.x, u1 = 7;
        for (int it = 0; it < nv; ++it) {
#pragma unroll
            for (int u = 0; u < 22; ++u) {        // the split's mix: and / sub / perm
                u0 = __float_as_uint(v0) & 0xFFFF0000u;
                v1 = v0 - __uint_as_float(u0);
                u1 = __builtin_amdgcn_perm(__float_as_uint(v1), u0, 0x07060302u);
                v0 = v1 + __uint_as_float(u1 & 0x3F800000u) + v2;
                v2 = v3 - v1; v3 = v0 * 0.5f;
            }
        }
        if (nv) out[blockIdx.x * 512 + threadIdx.x] = v0 + v1 + v2 + v3 + (float)u1;
    }
}

static float run(int nm, int nv, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, nm, nv, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, nm, nv, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int nm = 2000, nv = 2000;
    for (int r = 0; r < 3; ++r)
        printf("MFMA only %.1f us | VALU only %.1f us | both %.1f us   (24 MFMA x %d per wave; ~132 VALU x %d per wave)\n",
               run(nm, 0, out), run(0, nv, out), run(nm, nv, out), nm, nv);
    return 0;
}
