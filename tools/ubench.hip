// Micro-benchmarks that decide the GEMM's dequantisation strategy on gfx950 (DESIGN.md section
// "VALU budget").  Register-resident, no HBM traffic.
//
//  A) asm-pinned issue test: per iteration NM x v_mfma_i32_16x16x64_i8 (independent accumulators)
//     each followed by NV independent VALU ops of one kind -> cycles per MFMA vs NV.
//  B) compiler-scheduled main-loop body: 16 MFMAs + per-group dequant in several formulations
//     (scales from LDS, as in the real kernel) -> TOPS at 1/2 waves per SIMD.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip ; run: tools/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define CK(x)                                                                                    \
    do {                                                                                         \
        hipError_t e = (x);                                                                      \
        if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } \
    } while (0)

// ------------------------------------------------------------------------------------------ A
// KIND: 0 none, 1 v_fma_f32, 2 v_cvt_f32_i32, 3 v_pk_fma_f32, 4 v_mul_f32, 5 v_and_b32, 6 v_pk_mul_f32
template <int KIND> __device__ __forceinline__ void valu_op(float &x, float y, v2f &px, v2f py) {
    if constexpr (KIND == 1) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(x) : "v"(y));
    if constexpr (KIND == 2) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(x) : "v"(y));
    if constexpr (KIND == 3) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(px) : "v"(py));
    if constexpr (KIND == 4) asm volatile("v_mul_f32 %0, %1, %1" : "=v"(x) : "v"(y));
    if constexpr (KIND == 5) asm volatile("v_and_b32 %0, %1, %1" : "=v"(x) : "v"(y));
    if constexpr (KIND == 6) asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(px) : "v"(py));
}

template <int KIND, int NV, int NM>
__global__ void issue_kernel(int iters, float *out, long long *cycles) {
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
    v4i c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float x[8];
    v2f px[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { x[i] = threadIdx.x * 0.001f + i; px[i] = v2f{x[i], x[i] + 1.f}; }
    float y = 1.0001f;
    v2f py = {1.0001f, 0.9999f};
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < NM; m++) {
            if constexpr (NM > 0)
                asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c[m & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < NV; v++) valu_op<KIND>(x[v & 7], y, px[v & 7], py);
        }
        if constexpr (NM == 0) {
#pragma unroll
            for (int v = 0; v < NV * 4; v++) valu_op<KIND>(x[v & 7], y, px[v & 7], py);
        }
    }
    asm volatile("s_nop 15\n s_nop 15");
    long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i] + px[i][0] + px[i][1];
#pragma unroll
    for (int i = 0; i < 4; i++) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND, int NV, int NM> void run_issue(const char *name, int block, float *dout, long long *dcyc) {
    const int iters = 2000, grid = 256;
    issue_kernel<KIND, NV, NM><<<grid, block>>>(10, dout, dcyc);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    issue_kernel<KIND, NV, NM><<<grid, block>>>(iters, dout, dcyc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(grid);
    CK(hipMemcpy(h.data(), dcyc, grid * sizeof(long long), hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += v; avg /= grid;
    const int nm = NM > 0 ? NM : 4;
    // clock64 ticks at a fixed 100 MHz on gfx9; report time-derived numbers too
    double ns_per_slot = ms * 1e6 / iters / nm;
    printf("{\"exp\":\"A\",\"kind\":\"%s\",\"nv\":%d,\"nm\":%d,\"waves_per_simd\":%d,\"ns_per_mfma_slot\":%.3f,\"clock64_ticks_per_slot\":%.3f}\n",
           name, NV, NM, block / 256, ns_per_slot, avg / iters / nm);
}

// ------------------------------------------------------------------------------------------ B
// MODE 0: MFMA only (psums summed as int)       1: exact  cvt + mul + fma
//      2: cvt + fma with precomputed scale      3: fma only on denormal-trick bits + mul
//      4: fma only (scale precomputed)
template <int MODE>
__global__ __launch_bounds__(512) void loop_kernel(int groups, const float *scales, float *out) {
    __shared__ float sc[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) sc[i] = scales[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    v4i af[4], wf[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        af[i] = v4i{lane * 3 + i, lane + 7, i * 5 + 1, lane ^ 0x55};
        wf[i] = v4i{lane * 5 + i, lane + 3, i * 7 + 1, lane ^ 0x33};
    }
    v4f acc[4][4];
    v4i iacc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) { acc[i][j] = v4f{0, 0, 0, 0}; iacc[i][j] = v4i{0, 0, 0, 0}; }

    for (int g = 0; g < groups; g++) {
        const float *s = sc + (g & 15) * 128;
        float asv[4];
        v4f wsv[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            asv[i] = s[i * 16 + (lane & 15)];
            wsv[i] = *reinterpret_cast<const v4f *>(s + 64 + i * 16 + (lane >> 4) * 4);
        }
        // vary the operands a little so nothing is loop-invariant
        af[g & 3][0] ^= g; wf[(g + 1) & 3][1] ^= g;
#pragma unroll
        for (int mt = 0; mt < 4; mt++) {
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                if constexpr (MODE == 0) {
                    iacc[mt][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[nt], af[mt], iacc[mt][nt], 0, 0, 0);
                } else if constexpr (MODE == 3 || MODE == 4) {
                    v4i off = {1 << 20, 1 << 20, 1 << 20, 1 << 20};
                    v4i ps = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[nt], af[mt], off, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        float f = __builtin_bit_cast(float, ps[r]);
                        float sv = MODE == 3 ? asv[mt] * wsv[nt][r] : wsv[nt][r];
                        acc[mt][nt][r] = __builtin_fmaf(f, sv, acc[mt][nt][r]);
                    }
                } else {
                    v4i ps = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[nt], af[mt], v4i{0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        float sv = MODE == 1 ? asv[mt] * wsv[nt][r] : wsv[nt][r];
                        acc[mt][nt][r] = __builtin_fmaf((float)ps[r], sv, acc[mt][nt][r]);
                    }
                }
            }
        }
    }
    float t = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) t += acc[i][j][r] + (float)iacc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int MODE> void run_loop(const char *name, int block, int blocks_per_cu, const float *dsc, float *dout) {
    const int groups = 4096, grid = 256 * blocks_per_cu;
    loop_kernel<MODE><<<grid, block>>>(16, dsc, dout);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        loop_kernel<MODE><<<grid, block>>>(groups, dsc, dout);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    double waves = (double)grid * block / 64;
    double ops = waves * groups * 16.0 * (2.0 * 16 * 16 * 64);
    printf("{\"exp\":\"B\",\"mode\":\"%s\",\"block\":%d,\"blocks_per_cu\":%d,\"waves_per_simd\":%.1f,\"ms\":%.3f,\"TOPS\":%.1f}\n", name,
           block, blocks_per_cu, block / 256.0 * blocks_per_cu, best, ops / best / 1e9);
}

int main() {
    float *dout; long long *dcyc; float *dsc;
    CK(hipMalloc(&dout, 256 * 8 * 1024 * sizeof(float)));
    CK(hipMalloc(&dcyc, 4096 * sizeof(long long)));
    std::vector<float> hs(2048);
    for (int i = 0; i < 2048; i++) hs[i] = 0.001f + 1e-5f * (i % 97);
    CK(hipMalloc(&dsc, 2048 * sizeof(float)));
    CK(hipMemcpy(dsc, hs.data(), 2048 * sizeof(float), hipMemcpyHostToDevice));

    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("{\"device\":\"%s\",\"cus\":%d,\"clock_mhz\":%d,\"lds_per_block\":%zu}\n", prop.name, prop.multiProcessorCount,
           prop.clockRate / 1000, prop.sharedMemPerBlock);

    // A: 1 wave per SIMD (block 256) and 2 waves per SIMD (block 512)
#define A_ROW(KIND, NAME)                                                     \
    run_issue<KIND, 0, 4>(NAME, 256, dout, dcyc);                              \
    run_issue<KIND, 2, 4>(NAME, 256, dout, dcyc);                              \
    run_issue<KIND, 4, 4>(NAME, 256, dout, dcyc);                              \
    run_issue<KIND, 6, 4>(NAME, 256, dout, dcyc);                              \
    run_issue<KIND, 8, 4>(NAME, 256, dout, dcyc);                              \
    run_issue<KIND, 12, 4>(NAME, 256, dout, dcyc);                             \
    run_issue<KIND, 8, 0>(NAME, 256, dout, dcyc);                              \
    run_issue<KIND, 4, 4>(NAME, 512, dout, dcyc);                              \
    run_issue<KIND, 8, 4>(NAME, 512, dout, dcyc);                              \
    run_issue<KIND, 12, 4>(NAME, 512, dout, dcyc);
    A_ROW(1, "v_fma_f32")
    A_ROW(2, "v_cvt_f32_i32")
    A_ROW(3, "v_pk_fma_f32")
    A_ROW(4, "v_mul_f32")
    A_ROW(5, "v_and_b32")
    A_ROW(6, "v_pk_mul_f32")

#define B_ROW(MODE, NAME)                         \
    run_loop<MODE>(NAME, 256, 1, dsc, dout);      \
    run_loop<MODE>(NAME, 256, 2, dsc, dout);      \
    run_loop<MODE>(NAME, 512, 1, dsc, dout);
    B_ROW(0, "mfma_only")
    B_ROW(1, "cvt_mul_fma")
    B_ROW(2, "cvt_fma")
    B_ROW(3, "denorm_mul_fma")
    B_ROW(4, "denorm_fma")
    return 0;
}
