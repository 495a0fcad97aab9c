// round 6 probe: the W4A4 loop's arithmetic with everything in registers (per 32 x 32 x 64 tile-group: one FP6 product MFMA, one scale-tile MFMA, 16 v_fmac that read
// the PREVIOUS tile-group's P and S), by MFMA form and by where the MFMA operands live (VGPRs or AGPRs), at one and two waves per SIMD.  Wall clock (HIP events), 256
// workgroups.  Question behind it: the K = 8 scale tile / unscaled product MFMA gained more on the two-waves-per-SIMD loop (operands in VGPRs) than on the wave-tile loop
// (operands in AGPRs) -- do MFMA operand reads from VGPRs compete with the VALU's?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

#define FM8(P, S, C0)                                                                                                                       \
    "v_fmac_f32 %[" #C0 "0], %[" #P "0], %[" #S "0]\n v_fmac_f32 %[" #C0 "1], %[" #P "1], %[" #S "1]\n v_fmac_f32 %[" #C0 "2], %[" #P "2], %[" #S "2]\n"   \
    "v_fmac_f32 %[" #C0 "3], %[" #P "3], %[" #S "3]\n v_fmac_f32 %[" #C0 "4], %[" #P "4], %[" #S "4]\n v_fmac_f32 %[" #C0 "5], %[" #P "5], %[" #S "5]\n"   \
    "v_fmac_f32 %[" #C0 "6], %[" #P "6], %[" #S "6]\n v_fmac_f32 %[" #C0 "7], %[" #P "7], %[" #S "7]\n"

// MODE: 0 = scaled FP6 pair + K = 16 scale tile (rounds 1-5); 1 = unscaled FP6 + K = 8 scale tile (round 6); OPS: 0 = MFMA operands in VGPRs, 1 = in AGPRs
template <int MODE, int OPS, int THREADS> __global__ __launch_bounds__(THREADS) void k(float *out, int iters) {
    v16f p0 = {}, p1 = {}, s0 = {}, s1 = {};
    float c[16];
    for (int i = 0; i < 16; i++) c[i] = (float)(threadIdx.x + i);
    v6i w6 = {0x11111111, 0x22222222, 0x12121212, 0x21212121, 0x11221122, 0x22112211}, a6 = {0x21212121, 0x11111111, 0x22222222, 0x12121212, 0x22112211, 0x11221122};
    v4i x4 = {0x3f80, 0, 0, 0}, y4 = {0x3fc0, 0, 0, 0};
    v2i x2 = {0x3f80, 0}, y2 = {0x3fc0, 0};
    int mx = 0x7f7f7f7f;
#define FMAC_A_LO "v_fmac_f32 %[c0], v96, v112\n" "v_fmac_f32 %[c1], v97, v113\n" "v_fmac_f32 %[c2], v98, v114\n" "v_fmac_f32 %[c3], v99, v115\n" "v_fmac_f32 %[c4], v100, v116\n" "v_fmac_f32 %[c5], v101, v117\n" "v_fmac_f32 %[c6], v102, v118\n" "v_fmac_f32 %[c7], v103, v119\n"
#define FMAC_A_HI "v_fmac_f32 %[c8], v104, v120\n" "v_fmac_f32 %[c9], v105, v121\n" "v_fmac_f32 %[c10], v106, v122\n" "v_fmac_f32 %[c11], v107, v123\n" "v_fmac_f32 %[c12], v108, v124\n" "v_fmac_f32 %[c13], v109, v125\n" "v_fmac_f32 %[c14], v110, v126\n" "v_fmac_f32 %[c15], v111, v127\n"
#define FMAC_B_LO "v_fmac_f32 %[c0], v64, v80\n" "v_fmac_f32 %[c1], v65, v81\n" "v_fmac_f32 %[c2], v66, v82\n" "v_fmac_f32 %[c3], v67, v83\n" "v_fmac_f32 %[c4], v68, v84\n" "v_fmac_f32 %[c5], v69, v85\n" "v_fmac_f32 %[c6], v70, v86\n" "v_fmac_f32 %[c7], v71, v87\n"
#define FMAC_B_HI "v_fmac_f32 %[c8], v72, v88\n" "v_fmac_f32 %[c9], v73, v89\n" "v_fmac_f32 %[c10], v74, v90\n" "v_fmac_f32 %[c11], v75, v91\n" "v_fmac_f32 %[c12], v76, v92\n" "v_fmac_f32 %[c13], v77, v93\n" "v_fmac_f32 %[c14], v78, v94\n" "v_fmac_f32 %[c15], v79, v95\n"
// the P / S tiles are pinned (v[64:79] v[80:95] v[96:111] v[112:127]): two tile-groups per statement, each one's v_fmac read the other one's tiles
#define PROBE_BODY(PM, SM)                                                                                                                  \
    for (int it = 0; it < iters; it += 2) {                                                                                                 \
        asm volatile(PM("v[64:79]") FMAC_A_LO SM("v[80:95]") FMAC_A_HI PM("v[96:111]") FMAC_B_LO SM("v[112:127]") FMAC_B_HI    \
                     : [c0] "+v"(c[0]), [c1] "+v"(c[1]), [c2] "+v"(c[2]), [c3] "+v"(c[3]), [c4] "+v"(c[4]), [c5] "+v"(c[5]),                 \
                       [c6] "+v"(c[6]), [c7] "+v"(c[7]), [c8] "+v"(c[8]), [c9] "+v"(c[9]), [c10] "+v"(c[10]), [c11] "+v"(c[11]), [c12] "+v"(c[12]), \
                       [c13] "+v"(c[13]), [c14] "+v"(c[14]), [c15] "+v"(c[15]), "+{v[64:79]}"(p0), "+{v[80:95]}"(s0), "+{v[96:111]}"(p1), "+{v[112:127]}"(s1) \
                     : OPERANDS, [mx] "v"(mx));                                                                                             \
    }
#define PM_SCALED(P) "v_mfma_scale_f32_32x32x64_f8f6f4 " P ", %[w6], %[a6], 0, %[mx], %[mx] op_sel_hi:[0,0,0] cbsz:2 blgp:2\n"
#define PM_PLAIN(P) "v_mfma_f32_32x32x64_f8f6f4 " P ", %[w6], %[a6], 0 cbsz:2 blgp:2\n"
#define SM_K16(S) "v_mfma_f32_32x32x16_bf16 " S ", %[x], %[y], 0\n"
#define SM_K8(S) "v_mfma_f32_32x32x8bf16_1k " S ", %[x], %[y], 0\n"
    if constexpr (MODE == 0 && OPS == 0) {
#define OPERANDS [w6] "v"(w6), [a6] "v"(a6), [x] "v"(x4), [y] "v"(y4)
        PROBE_BODY(PM_SCALED, SM_K16)
#undef OPERANDS
    } else if constexpr (MODE == 0 && OPS == 1) {
#define OPERANDS [w6] "a"(w6), [a6] "a"(a6), [x] "a"(x4), [y] "a"(y4)
        PROBE_BODY(PM_SCALED, SM_K16)
#undef OPERANDS
    } else if constexpr (MODE == 1 && OPS == 0) {
#define OPERANDS [w6] "v"(w6), [a6] "v"(a6), [x] "v"(x2), [y] "v"(y2)
        PROBE_BODY(PM_PLAIN, SM_K8)
#undef OPERANDS
    } else if constexpr (MODE == 1 && OPS == 1) {
#define OPERANDS [w6] "a"(w6), [a6] "a"(a6), [x] "a"(x2), [y] "a"(y2)
        PROBE_BODY(PM_PLAIN, SM_K8)
#undef OPERANDS
    } else if constexpr (MODE == 1 && OPS == 2) {   // first MFMA operand (weights side) in VGPRs, second in AGPRs
#define OPERANDS [w6] "v"(w6), [a6] "a"(a6), [x] "v"(x2), [y] "a"(y2)
        PROBE_BODY(PM_PLAIN, SM_K8)
#undef OPERANDS
    } else if constexpr (MODE == 1 && OPS == 3) {   // the other way round
#define OPERANDS [w6] "a"(w6), [a6] "v"(a6), [x] "a"(x2), [y] "v"(y2)
        PROBE_BODY(PM_PLAIN, SM_K8)
#undef OPERANDS
    } else if constexpr (MODE == 1 && OPS == 4) {   // product operands in AGPRs, scale tuples in VGPRs
#define OPERANDS [w6] "a"(w6), [a6] "a"(a6), [x] "v"(x2), [y] "v"(y2)
        PROBE_BODY(PM_PLAIN, SM_K8)
#undef OPERANDS
    } else {                                          // product operands in VGPRs, scale tuples in AGPRs
#define OPERANDS [w6] "v"(w6), [a6] "v"(a6), [x] "a"(x2), [y] "a"(y2)
        PROBE_BODY(PM_PLAIN, SM_K8)
#undef OPERANDS
    }
    float acc = p0[0] + p1[0] + s0[0] + s1[0];
    for (int i = 0; i < 16; i++) acc += c[i];
    out[blockIdx.x * THREADS + threadIdx.x] = acc;
}
template <int MODE, int OPS, int THREADS> void run(const char *name, float *out, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE, OPS, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); hipLaunchKernelGGL((k<MODE, OPS, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, iters); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double tg_per_simd = (double)iters * (THREADS / 256);
    printf("{\"case\":\"%s\",\"waves_per_simd\":%d,\"ns_per_tile_group_per_simd\":%.2f,\"cycles_at_2.4GHz\":%.1f}\n", name, THREADS / 256, ms * 1e6 / tg_per_simd, ms * 1e6 / tg_per_simd * 2.4);
}
int main() {
    float *out; CK(hipMalloc(&out, 256 * 1024 * 4));
    const int iters = 200000;
    for (int rep = 0; rep < 2; rep++) {
        run<0, 0, 512>("scaled pair + K=16 tile, VGPR operands", out, iters);
        run<0, 1, 512>("scaled pair + K=16 tile, AGPR operands", out, iters);
        run<1, 0, 512>("unscaled + K=8 tile, VGPR operands", out, iters);
        run<1, 1, 512>("unscaled + K=8 tile, AGPR operands", out, iters);
        run<0, 0, 256>("scaled pair + K=16 tile, VGPR operands", out, iters);
        run<0, 1, 256>("scaled pair + K=16 tile, AGPR operands", out, iters);
        run<1, 0, 256>("unscaled + K=8 tile, VGPR operands", out, iters);
        run<1, 1, 256>("unscaled + K=8 tile, AGPR operands", out, iters);
        run<1, 2, 256>("unscaled + K=8 tile, first operands VGPR / second AGPR", out, iters);
        run<1, 3, 256>("unscaled + K=8 tile, first operands AGPR / second VGPR", out, iters);
        run<1, 4, 256>("unscaled + K=8 tile, product operands AGPR / scale tuples VGPR", out, iters);
        run<1, 5, 256>("unscaled + K=8 tile, product operands VGPR / scale tuples AGPR", out, iters);
    }
    return 0;
}
