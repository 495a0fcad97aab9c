// Issue-level micro-benchmark of the FP6 main-loop body on gfx950, hand-scheduled in inline asm
// with fixed registers (the compiler cannot be steered into this interleave).
// One "tile-group" = P = fp6 MFMA(32x32x64), S = scale-tile MFMA, 16 fp32 fma (as 8 v_pk_fma_f32 or
// 16 v_fma_f32) reading the P/S pair produced one tile-group earlier.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench3 tools/ubench3.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// registers: acc v[0:127] (8 tiles), P0 v[128:143] S0 v[144:159] P1 v[160:175] S1 v[176:191]
// operands: a[0:5] a[6:11] (fp6), a[16:19] a[20:23] (scale frags), v192/v193 = MX scales
#define P_MFMA(dst) "v_mfma_scale_f32_32x32x64_f8f6f4 " dst ", a[0:5], a[6:11], 0, v196, v197 op_sel_hi:[0,0,0] cbsz:2 blgp:2\n"
#define S16(dst) "v_mfma_f32_32x32x16_bf16 " dst ", a[16:19], a[20:23], 0\n"
#define S8(dst) "v_mfma_f32_32x32x8_bf16 " dst ", a[16:17], a[20:21], 0\n"
#define S8H(dst) "v_mfma_f32_32x32x8_f16 " dst ", a[16:17], a[20:21], 0\n"
#define PK(t, r, pb, sb) "v_pk_fma_f32 v[" #t "+" #r ":" #t "+" #r "+1], v[" #pb "+" #r ":" #pb "+" #r "+1], v[" #sb "+" #r ":" #sb "+" #r "+1], v[" #t "+" #r ":" #t "+" #r "+1]\n"
#define FM(t, r, pb, sb) "v_fma_f32 v[" #t "+" #r "], v[" #pb "+" #r "], v[" #sb "+" #r "], v[" #t "+" #r "]\n"
#define PK4A(t, pb, sb) PK(t, 0, pb, sb) PK(t, 2, pb, sb) PK(t, 4, pb, sb) PK(t, 6, pb, sb)
#define PK4B(t, pb, sb) PK(t, 8, pb, sb) PK(t, 10, pb, sb) PK(t, 12, pb, sb) PK(t, 14, pb, sb)
#define FM8A(t, pb, sb) FM(t, 0, pb, sb) FM(t, 1, pb, sb) FM(t, 2, pb, sb) FM(t, 3, pb, sb) FM(t, 4, pb, sb) FM(t, 5, pb, sb) FM(t, 6, pb, sb) FM(t, 7, pb, sb)
#define FM8B(t, pb, sb) FM(t, 8, pb, sb) FM(t, 9, pb, sb) FM(t, 10, pb, sb) FM(t, 11, pb, sb) FM(t, 12, pb, sb) FM(t, 13, pb, sb) FM(t, 14, pb, sb) FM(t, 15, pb, sb)

#define CLOBBERS                                                                                                   \
    "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19", \
    "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37",   \
    "v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55",   \
    "v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v128","v129","v130","v131","v132","v133","v134","v135",       \
    "v136","v137","v138","v139","v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150",      \
    "v151","v152","v153","v154","v155","v156","v157","v158","v159","v160","v161","v162","v163","v164","v165",      \
    "v166","v167","v168","v169","v170","v171","v172","v173","v174","v175","v176","v177","v178","v179","v180",      \
    "v181","v182","v183","v184","v185","v186","v187","v188","v189","v190","v191","v192","v193","v194","v195","v196","v197","a0","a1","a2",     \
    "a3","a4","a5","a6","a7","a8","a9","a10","a11","a16","a17","a18","a19","a20","a21","a22","a23"

// BODY variants (two tile-groups per iteration, buffers 0/1):
//  0: P only            1: P + S16            2: P + S8(bf16)        3: P + S8(f16)
//  4: P,4pk,S16,4pk     5: P,4pk,S8,4pk       6: P,8fma,S16,8fma     7: P,8fma,S8,8fma
//  8: 8pk only          9: 16 fma only       10: P, 8pk (no S)      11: S16 only   12: S8 only
template <int BODY> __global__ __launch_bounds__(512) void issue_kernel(int iters, float *out, long long *cyc) {
    asm volatile(
        "v_mov_b32 v192, 0x1020304\n v_accvgpr_write_b32 a0, v192\n"
        "v_mov_b32 v192, 0x11121314\n v_accvgpr_write_b32 a1, v192\n"
        "v_mov_b32 v192, 0x5030107\n v_accvgpr_write_b32 a2, v192\n"
        "v_mov_b32 v192, 0x1020304\n v_accvgpr_write_b32 a3, v192\n"
        "v_mov_b32 v192, 0x11121314\n v_accvgpr_write_b32 a4, v192\n"
        "v_mov_b32 v192, 0x5030107\n v_accvgpr_write_b32 a5, v192\n"
        "v_mov_b32 v192, 0x1020304\n v_accvgpr_write_b32 a6, v192\n"
        "v_mov_b32 v192, 0x11121314\n v_accvgpr_write_b32 a7, v192\n"
        "v_mov_b32 v192, 0x5030107\n v_accvgpr_write_b32 a8, v192\n"
        "v_mov_b32 v192, 0x1020304\n v_accvgpr_write_b32 a9, v192\n"
        "v_mov_b32 v192, 0x11121314\n v_accvgpr_write_b32 a10, v192\n"
        "v_mov_b32 v192, 0x5030107\n v_accvgpr_write_b32 a11, v192\n"
        "v_mov_b32 v192, 0x3f80\n v_accvgpr_write_b32 a16, v192\n"
        "v_mov_b32 v192, 0x0\n v_accvgpr_write_b32 a17, v192\n"
        "v_mov_b32 v192, 0x0\n v_accvgpr_write_b32 a18, v192\n"
        "v_mov_b32 v192, 0x0\n v_accvgpr_write_b32 a19, v192\n"
        "v_mov_b32 v192, 0x3f00\n v_accvgpr_write_b32 a20, v192\n"
        "v_mov_b32 v192, 0x0\n v_accvgpr_write_b32 a21, v192\n"
        "v_mov_b32 v192, 0x0\n v_accvgpr_write_b32 a22, v192\n"
        "v_mov_b32 v192, 0x0\n v_accvgpr_write_b32 a23, v192\n"
        "v_mov_b32 v196, 0x82828282\n v_mov_b32 v197, 0x82828282\n" ::: CLOBBERS);
#define ZERO16(b) "v_mov_b32 v[" #b "+0], 0\n v_mov_b32 v[" #b "+1], 0\n v_mov_b32 v[" #b "+2], 0\n v_mov_b32 v[" #b "+3], 0\n v_mov_b32 v[" #b "+4], 0\n v_mov_b32 v[" #b "+5], 0\n v_mov_b32 v[" #b "+6], 0\n v_mov_b32 v[" #b "+7], 0\n v_mov_b32 v[" #b "+8], 0\n v_mov_b32 v[" #b "+9], 0\n v_mov_b32 v[" #b "+10], 0\n v_mov_b32 v[" #b "+11], 0\n v_mov_b32 v[" #b "+12], 0\n v_mov_b32 v[" #b "+13], 0\n v_mov_b32 v[" #b "+14], 0\n v_mov_b32 v[" #b "+15], 0\n"
    asm volatile(ZERO16(0) ZERO16(16) ZERO16(32) ZERO16(48) ZERO16(128) ZERO16(144) ZERO16(160) ZERO16(176) ::: CLOBBERS);
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if constexpr (BODY == 0) asm volatile(P_MFMA("v[128:143]") P_MFMA("v[160:175]") ::: CLOBBERS);
        if constexpr (BODY == 1) asm volatile(P_MFMA("v[128:143]") S16("v[144:159]") P_MFMA("v[160:175]") S16("v[176:191]") ::: CLOBBERS);
        if constexpr (BODY == 2) asm volatile(P_MFMA("v[128:143]") S8("v[144:159]") P_MFMA("v[160:175]") S8("v[176:191]") ::: CLOBBERS);
        if constexpr (BODY == 3) asm volatile(P_MFMA("v[128:143]") S8H("v[144:159]") P_MFMA("v[160:175]") S8H("v[176:191]") ::: CLOBBERS);
        if constexpr (BODY == 4) asm volatile(P_MFMA("v[128:143]") PK4A(0, 160, 176) S16("v[144:159]") PK4B(0, 160, 176)
                                              P_MFMA("v[160:175]") PK4A(16, 128, 144) S16("v[176:191]") PK4B(16, 128, 144) ::: CLOBBERS);
        if constexpr (BODY == 5) asm volatile(P_MFMA("v[128:143]") PK4A(0, 160, 176) S8("v[144:159]") PK4B(0, 160, 176)
                                              P_MFMA("v[160:175]") PK4A(16, 128, 144) S8("v[176:191]") PK4B(16, 128, 144) ::: CLOBBERS);
        if constexpr (BODY == 6) asm volatile(P_MFMA("v[128:143]") FM8A(0, 160, 176) S16("v[144:159]") FM8B(0, 160, 176)
                                              P_MFMA("v[160:175]") FM8A(16, 128, 144) S16("v[176:191]") FM8B(16, 128, 144) ::: CLOBBERS);
        if constexpr (BODY == 7) asm volatile(P_MFMA("v[128:143]") FM8A(0, 160, 176) S8("v[144:159]") FM8B(0, 160, 176)
                                              P_MFMA("v[160:175]") FM8A(16, 128, 144) S8("v[176:191]") FM8B(16, 128, 144) ::: CLOBBERS);
        if constexpr (BODY == 8) asm volatile(PK4A(0, 160, 176) PK4B(0, 160, 176) PK4A(16, 128, 144) PK4B(16, 128, 144) ::: CLOBBERS);
        if constexpr (BODY == 9) asm volatile(FM8A(0, 160, 176) FM8B(0, 160, 176) FM8A(16, 128, 144) FM8B(16, 128, 144) ::: CLOBBERS);
        if constexpr (BODY == 10) asm volatile(P_MFMA("v[128:143]") PK4A(0, 160, 176) PK4B(0, 160, 176)
                                               P_MFMA("v[160:175]") PK4A(16, 128, 144) PK4B(16, 128, 144) ::: CLOBBERS);
        // bank-aware variants: P base = 0 mod 4, S base = 2 mod 4, acc base = 1 mod 4 (scalar fma) / 2-offset pk
        if constexpr (BODY == 13) asm volatile(FM8A(1, 160, 178) FM8B(1, 160, 178) FM8A(17, 128, 146) FM8B(17, 128, 146) ::: CLOBBERS);
        if constexpr (BODY == 14) asm volatile(P_MFMA("v[128:143]") FM8A(1, 160, 178) S16("v[146:161]") FM8B(1, 160, 178)
                                               P_MFMA("v[160:175]") FM8A(17, 128, 146) S16("v[178:193]") FM8B(17, 128, 146) ::: CLOBBERS);
        if constexpr (BODY == 15) asm volatile(PK4A(0, 160, 178) PK4B(0, 160, 178) PK4A(16, 128, 146) PK4B(16, 128, 146) ::: CLOBBERS);
        if constexpr (BODY == 16) asm volatile(PK4A(2, 160, 178) PK4B(2, 160, 178) PK4A(18, 128, 146) PK4B(18, 128, 146) ::: CLOBBERS);
        if constexpr (BODY == 17) asm volatile(FM8A(3, 160, 178) FM8B(3, 160, 178) FM8A(19, 128, 146) FM8B(19, 128, 146) ::: CLOBBERS);
        if constexpr (BODY == 18) asm volatile(FM8A(1, 160, 177) FM8B(1, 160, 177) FM8A(17, 128, 145) FM8B(17, 128, 145) ::: CLOBBERS);
        if constexpr (BODY == 19) asm volatile(P_MFMA("v[128:143]") PK4A(0, 160, 178) S16("v[146:161]") PK4B(0, 160, 178)
                                               P_MFMA("v[160:175]") PK4A(16, 128, 146) S16("v[178:193]") PK4B(16, 128, 146) ::: CLOBBERS);
        if constexpr (BODY == 11) asm volatile(S16("v[144:159]") S16("v[176:191]") ::: CLOBBERS);
        if constexpr (BODY == 12) asm volatile(S8("v[144:159]") S8("v[176:191]") ::: CLOBBERS);
    }
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
    long long t1 = clock64();
    float s;
    asm volatile("v_add_f32 %0, v0, v16\n v_add_f32 %0, %0, v128\n v_add_f32 %0, %0, v176" : "=v"(s) :: CLOBBERS);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int BODY> void run(const char *name, float *dout, long long *dcyc, int block = 256) {
    const int iters = 4000, grid = 256;
    issue_kernel<BODY><<<grid, block>>>(10, dout, dcyc);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    issue_kernel<BODY><<<grid, block>>>(iters, dout, dcyc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(grid);
    CK(hipMemcpy(h.data(), dcyc, grid * sizeof(long long), hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += v; avg /= grid;
    double ns_tg = ms * 1e6 / iters / 2 / (block / 256);  // per tile-group per SIMD
    // TOPS if a tile-group (2*32*32*64 ops) took this long on all 1024 SIMDs
    printf("{\"exp\":\"I\",\"waves_per_simd\":%d,\"body\":\"%s\",\"ns_per_tilegroup\":%.2f,\"ticks_per_tilegroup\":%.1f,\"equiv_TOPS\":%.0f}\n", block / 256, name, ns_tg,
           avg / iters / 2, 1024.0 * 131072 / ns_tg / 1e3);
    fflush(stdout);
}

int main() {
    float *dout; long long *dcyc;
    CK(hipMalloc(&dout, 256 * 512 * sizeof(float)));
    CK(hipMalloc(&dcyc, 256 * sizeof(long long)));
    run<0>("P", dout, dcyc);
    run<11>("S16", dout, dcyc);
    run<12>("S8", dout, dcyc);
    run<1>("P+S16", dout, dcyc);
    run<2>("P+S8bf16", dout, dcyc);
    run<3>("P+S8f16", dout, dcyc);
    run<8>("8pk", dout, dcyc);
    run<9>("16fma", dout, dcyc);
    run<10>("P+8pk", dout, dcyc);
    run<13>("16fma P0 S2 acc1", dout, dcyc);
    run<17>("16fma P0 S2 acc3", dout, dcyc);
    run<18>("16fma P0 S1 acc1(conflict S/acc)", dout, dcyc);
    run<15>("8pk P0 S2 acc0", dout, dcyc);
    run<16>("8pk P0 S2 acc2", dout, dcyc);
    run<14>("P,8fma,S16,8fma banked", dout, dcyc);
    run<19>("P,4pk,S16,4pk banked", dout, dcyc);
    run<4>("P,4pk,S16,4pk", dout, dcyc);
    run<5>("P,4pk,S8,4pk", dout, dcyc);
    run<6>("P,8fma,S16,8fma", dout, dcyc);
    run<7>("P,8fma,S8,8fma", dout, dcyc);
    printf("--- 2 waves per SIMD (ns per tile-group per SIMD)\n");
    run<0>("P", dout, dcyc, 512);
    run<1>("P+S16", dout, dcyc, 512);
    run<8>("8pk", dout, dcyc, 512);
    run<9>("16fma", dout, dcyc, 512);
    run<10>("P+8pk", dout, dcyc, 512);
    run<4>("P,4pk,S16,4pk", dout, dcyc, 512);
    run<6>("P,8fma,S16,8fma", dout, dcyc, 512);
    return 0;
}
