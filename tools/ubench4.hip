// VALU issue-rate probe (gfx950): 32 independent ops per loop iteration, fixed registers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
#define R4(M, i) M(i) M(i + 1) M(i + 2) M(i + 3)
#define R16(M, i) R4(M, i) R4(M, i + 4) R4(M, i + 8) R4(M, i + 12)
#define R32(M) R16(M, 0) R16(M, 16)
#define S_(x) #x
#define OP_A(i) "v_fma_f32 v[" S_(i) "], v[32+" S_(i) "], v[64+" S_(i) "], v[" S_(i) "]\n"
#define OP_B(i) "v_fma_f32 v[" S_(i) "], v32, v65, v[" S_(i) "]\n"
#define OP_C(i) "v_mul_f32 v[" S_(i) "], v[32+" S_(i) "], v[64+" S_(i) "]\n"
#define OP_D(i) "v_pk_fma_f32 v[2*(" S_(i) "):2*(" S_(i) ")+1], v[64+2*(" S_(i) "):64+2*(" S_(i) ")+1], v[128+2*(" S_(i) "):128+2*(" S_(i) ")+1], v[2*(" S_(i) "):2*(" S_(i) ")+1]\n"
#define OP_E(i) "v_mov_b32 v[" S_(i) "], v[32+" S_(i) "]\n"
#define OP_F(i) "v_fmac_f32 v[" S_(i) "], v[32+" S_(i) "], v[64+" S_(i) "]\n"
#define OP_G(i) "v_pk_mul_f32 v[2*(" S_(i) "):2*(" S_(i) ")+1], v[64+2*(" S_(i) "):64+2*(" S_(i) ")+1], v[128+2*(" S_(i) "):128+2*(" S_(i) ")+1]\n"
#define OP_H(i) "v_fma_f32 v[" S_(i) "], v[32+" S_(i) "], s4, v[" S_(i) "]\n"
#define OP_I(i) "v_pk_fma_f32 v[2*(" S_(i) "):2*(" S_(i) ")+1], v[64+2*(" S_(i) "):64+2*(" S_(i) ")+1], v[128:129], v[2*(" S_(i) "):2*(" S_(i) ")+1]\n"
#define OP_J(i) "v_add_f32 v[" S_(i) "], v[32+" S_(i) "], v[" S_(i) "]\n"

template <int K> __global__ __launch_bounds__(1024) void k(int iters, float *out, long long *cyc) {
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if constexpr (K == 0) asm volatile(R32(OP_A));
        if constexpr (K == 1) asm volatile(R32(OP_B));
        if constexpr (K == 2) asm volatile(R32(OP_C));
        if constexpr (K == 3) asm volatile(R32(OP_D));
        if constexpr (K == 4) asm volatile(R32(OP_E));
        if constexpr (K == 5) asm volatile(R32(OP_F));
        if constexpr (K == 6) asm volatile(R32(OP_G));
        if constexpr (K == 7) asm volatile(R32(OP_H));
        if constexpr (K == 8) asm volatile(R32(OP_I));
        if constexpr (K == 9) asm volatile(R32(OP_J));
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (iters < 0) out[threadIdx.x] = 1.f;
}
template <int K> void run(const char *name, float *dout, long long *dcyc) {
    for (int block = 256; block <= 1024; block *= 2) {
        const int iters = 4000, grid = 256;
        k<K><<<grid, block>>>(10, dout, dcyc);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        k<K><<<grid, block>>>(iters, dout, dcyc);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<long long> h(grid);
        CK(hipMemcpy(h.data(), dcyc, grid * sizeof(long long), hipMemcpyDeviceToHost));
        double avg = 0; for (auto v : h) avg += v; avg /= grid;
        printf("{\"exp\":\"V\",\"op\":\"%s\",\"waves_per_simd\":%d,\"ns_per_op_per_wave\":%.3f,\"ticks_per_op_per_wave\":%.2f,\"ns_per_op_per_simd\":%.3f}\n",
               name, block / 256, ms * 1e6 / iters / 32, avg / iters / 32, ms * 1e6 / iters / 32 / (block / 256));
    }
}
int main() {
    float *dout; long long *dcyc;
    CK(hipMalloc(&dout, 4096 * sizeof(float))); CK(hipMalloc(&dcyc, 256 * sizeof(long long)));
    run<0>("fma 3 distinct vgpr", dout, dcyc);
    run<1>("fma shared srcs", dout, dcyc);
    run<7>("fma vgpr,sgpr,vgpr", dout, dcyc);
    run<5>("fmac (VOP2)", dout, dcyc);
    run<2>("mul 2 distinct", dout, dcyc);
    run<9>("add (VOP2)", dout, dcyc);
    run<4>("mov", dout, dcyc);
    run<3>("pk_fma 3 distinct", dout, dcyc);
    run<8>("pk_fma shared src1", dout, dcyc);
    run<6>("pk_mul", dout, dcyc);
    return 0;
}
