// round 6 probe: does the legacy K = 8 16-bit MFMA (v_mfma_f32_32x32x8_bf16_1k / _f16) cost fewer matrix-pipe passes than the K = 16 form on gfx950?
// The scale tile of the W4A4 loop uses 2 of its 16 k-slots; a K = 8 form with the same two non-zero slots computes the same S = 2 ws as.
// One workgroup of 256 threads per CU (one wave per SIMD), `iters` x 64 back-to-back independent MFMAs per wave; wall clock via HIP events.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v16f __attribute__((ext_vector_type(16)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int MODE> __global__ __launch_bounds__(256, 1) void k(float *out, int iters) {
    v16f acc[4];
    for (int t = 0; t < 4; t++) for (int i = 0; i < 16; i++) acc[t][i] = 0.f;
    bf16x8 a8, b8; v4s a4, b4; f16x4 h4a, h4b; f16x8 h8a, h8b;
    for (int j = 0; j < 8; j++) { a8[j] = (__bf16)0.f; b8[j] = (__bf16)0.f; h8a[j] = (_Float16)0.f; h8b[j] = (_Float16)0.f; }
    for (int j = 0; j < 4; j++) { a4[j] = 0; b4[j] = 0; h4a[j] = (_Float16)0.f; h4b[j] = (_Float16)0.f; }
    a8[0] = (__bf16)(1.0f + threadIdx.x * 0.0078125f); b8[0] = (__bf16)1.5f; a4[0] = 0x3f80 + (threadIdx.x & 63); b4[0] = 0x3fc0;
    h4a[0] = (_Float16)(1.0f + threadIdx.x * 0.001f); h4b[0] = (_Float16)1.5f; h8a[0] = h4a[0]; h8b[0] = h4b[0];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if constexpr (MODE == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[t], 0, 0, 0);
                else if constexpr (MODE == 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, acc[t], 0, 0, 0);
                else if constexpr (MODE == 2) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h8a, h8b, acc[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(h4a, h4b, acc[t], 0, 0, 0);
            }
    }
    float s = 0; for (int t = 0; t < 4; t++) for (int i = 0; i < 16; i++) s += acc[t][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
typedef float v32f __attribute__((ext_vector_type(32)));
// two-block forms: D[b] = A[b] B[b]^T for b = 0, 1 (lanes 0-31 feed block 0, lanes 32-63 block 1): 32 output registers
template <int MODE> __global__ __launch_bounds__(256, 1) void k2(float *out, int iters) {
    v32f acc[2];
    for (int t = 0; t < 2; t++) for (int i = 0; i < 32; i++) acc[t][i] = 0.f;
    v4s a4, b4; f16x4 h4a, h4b;
    for (int j = 0; j < 4; j++) { a4[j] = 0; b4[j] = 0; h4a[j] = (_Float16)0.f; h4b[j] = (_Float16)0.f; }
    a4[0] = 0x3f80 + (threadIdx.x & 63); b4[0] = 0x3fc0; h4a[0] = (_Float16)(1.0f + threadIdx.x * 0.001f); h4b[0] = (_Float16)1.5f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 32; u++)
#pragma unroll
            for (int t = 0; t < 2; t++) {
                if constexpr (MODE == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x4bf16_1k(a4, b4, acc[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f32_32x32x4f16(h4a, h4b, acc[t], 0, 0, 0);
            }
    }
    float s = 0; for (int t = 0; t < 2; t++) for (int i = 0; i < 32; i++) s += acc[t][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run2(const char *name, float *out, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k2<MODE>), dim3(256), dim3(256), 0, 0, out, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL((k2<MODE>), dim3(256), dim3(256), 0, 0, out, iters); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double n = (double)iters * 64;
    printf("{\"mfma\":\"%s\",\"ns_per_mfma_per_simd\":%.3f,\"cycles_at_2.4GHz\":%.2f}\n", name, ms * 1e6 / n, ms * 1e6 / n * 2.4);
}
template <int MODE> void run(const char *name, float *out, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, out, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, out, iters); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double n = (double)iters * 64;
    float h[4]; CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    printf("{\"mfma\":\"%s\",\"ns_per_mfma_per_simd\":%.3f,\"cycles_at_2.4GHz\":%.2f,\"check\":%.4f}\n", name, ms * 1e6 / n, ms * 1e6 / n * 2.4, h[1]);
}
int main() {
    float *out; CK(hipMalloc(&out, 256 * 256 * 4));
    const int iters = 20000;
    run<0>("v_mfma_f32_32x32x16_bf16", out, iters);
    run<1>("v_mfma_f32_32x32x8_bf16_1k", out, iters);
    run<2>("v_mfma_f32_32x32x16_f16", out, iters);
    run<3>("v_mfma_f32_32x32x8_f16", out, iters);
    run<0>("v_mfma_f32_32x32x16_bf16", out, iters);
    run<1>("v_mfma_f32_32x32x8_bf16_1k", out, iters);
    run2<0>("v_mfma_f32_32x32x4_2b_bf16 (two blocks)", out, iters);
    run2<1>("v_mfma_f32_32x32x4_2b_f16 (two blocks)", out, iters);
    return 0;
}
