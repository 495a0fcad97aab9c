// What does one more instruction between two MFMAs cost on a gfx950 SIMD that hosts ONE wave?  (geometry-2 attention kernel,
// DESIGN.md 6b.)  Loop body: 8 x { v_mfma_f32_32x32x16_bf16 (8 independent accumulators) ; N fillers of one kind (independent
// registers) }, 4 waves per workgroup = one per SIMD, one workgroup per CU.  Prints shader cycles per MFMA (s_memtime) for every
// (kind, N).   hipcc --offload-arch=gfx950 -O2 -o filler_probe filler_probe.hip && ./filler_probe
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdlib>
#include <cstdio>
#include <vector>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

// filler r of a slot works on its own register g<r> (no dependence between the fillers of a slot, nor on the MFMAs)
#define FILL_fma(r)   "v_fma_f32 %[g" #r "], %[f1], %[f2], %[g" #r "]\n"
#define FILL_exp(r)   "v_exp_f32 %[g" #r "], %[f1]\n"
#define FILL_cvt(r)   "v_cvt_pk_bf16_f32 %[g" #r "], %[f1], %[f2]\n"
#define FILL_dot(r)   "v_dot2c_f32_bf16 %[g" #r "], %[f1], %[f2]\n"
#define FILL_max3(r)  "v_max3_f32 %[g" #r "], %[f1], %[f2], %[g" #r "]\n"
#define FILL_add(r)   "v_add_f32 %[g" #r "], %[f1], %[g" #r "]\n"
#define FILL_perm(r)  "v_permlane32_swap_b32 %[g" #r "], %[h" #r "]\n"
#define FILL_mov(r)   "v_mov_b32 %[g" #r "], %[f1]\n"
#define FILL_pkmul(r) "v_pk_mul_f32 %[p0], %[p1], %[p1]\n"
#define FILL_dsr(r)   "ds_read_b128 %[d0], %[la]\n"
#define FILL_snop(r)  "s_nop 0\n"
#define FILL_salu(r)  "s_add_u32 %[s0], %[s0], 1\n"
#define FILL_dep(r)   "v_fma_f32 %[g0], %[f1], %[f2], %[g0]\n" /* a DEPENDENT chain: every filler on the same register */
#define FILL_accr(r)  "v_accvgpr_read_b32 %[g" #r "], %[a7]\n"

#define MF(i) "v_mfma_f32_32x32x16_bf16 %[a" #i "], %[x], %[y], %[a" #i "]\n"

#define BODY(F) MF(0) F MF(1) F MF(2) F MF(3) F MF(4) F MF(5) F MF(6) F MF(7) F

#define KERNEL(name, F)                                                                                                        \
    __global__ __launch_bounds__(256, 1) void name(long long *out, int iters) {                                               \
        __shared__ v4f lds[1024];                                                                                              \
        v16f a0 = {}, a1 = {}, a2 = {}, a3 = {}, a4 = {}, a5 = {}, a6 = {}, a7 = {};                                           \
        v4f x = {1.f, 2.f, 3.f, 4.f}, y = {1.f, 1.f, 1.f, 1.f}, d0 = {};                                                       \
        float f1 = 0.5f, f2 = 0.25f;                                                                                           \
        float g0 = threadIdx.x, g1 = 1, g2 = 2, g3 = 3, g4 = 4, g5 = 5, g6 = 6, g7 = 7;                                        \
        float h0 = 1, h1 = 1, h2 = 2, h3 = 3, h4 = 4, h5 = 5, h6 = 6, h7 = 7;                                                  \
        double p0 = 0, p1 = 1;                                                                                                 \
        unsigned la = (threadIdx.x & 63) * 16, s0 = 0;                                                                         \
        lds[threadIdx.x] = x;                                                                                                  \
        __syncthreads();                                                                                                       \
        long long t0 = __builtin_readcyclecounter();                                                                           \
        for (int i = 0; i < iters; i++)                                                                                        \
            asm volatile(BODY(F)                                                                                               \
                         : [a0] "+a"(a0), [a1] "+a"(a1), [a2] "+a"(a2), [a3] "+a"(a3), [a4] "+a"(a4), [a5] "+a"(a5), [a6] "+a"(a6),  \
                           [a7] "+a"(a7), [d0] "+v"(d0), [p0] "+v"(p0), [s0] "+s"(s0), [g0] "+v"(g0), [g1] "+v"(g1),            \
                           [g2] "+v"(g2), [g3] "+v"(g3), [g4] "+v"(g4), [g5] "+v"(g5), [g6] "+v"(g6), [g7] "+v"(g7),            \
                           [h0] "+v"(h0), [h1] "+v"(h1), [h2] "+v"(h2), [h3] "+v"(h3), [h4] "+v"(h4), [h5] "+v"(h5),            \
                           [h6] "+v"(h6), [h7] "+v"(h7)                                                                        \
                         : [x] "v"(x), [y] "v"(y), [f1] "v"(f1), [f2] "v"(f2), [la] "v"(la), [p1] "v"(p1) : "scc");                    \
        asm volatile("s_waitcnt lgkmcnt(0)\ns_nop 15\ns_nop 15" ::: "memory");                                                 \
        long long t1 = __builtin_readcyclecounter();                                                                           \
        float acc = a0[0] + a1[0] + a2[0] + a3[0] + a4[0] + a5[0] + a6[0] + a7[0] + g0 + g1 + g2 + g3 + g4 + g5 + g6 + g7 + h0 + h1 + h2 + h3 + h4 + h5 + h6 + h7 + d0[0] + (float)p0 + s0;          \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                                       \
        if (acc == 12345.678f) out[0] = 0;                                                                                     \
    }

#define K1(k) KERNEL(k_##k##_1, FILL_##k(0))
#define K2(k) KERNEL(k_##k##_2, FILL_##k(0) FILL_##k(1))
#define K4(k) KERNEL(k_##k##_4, FILL_##k(0) FILL_##k(1) FILL_##k(2) FILL_##k(3))
#define K5(k) KERNEL(k_##k##_5, FILL_##k(0) FILL_##k(1) FILL_##k(2) FILL_##k(3) FILL_##k(4))
#define K6(k) KERNEL(k_##k##_6, FILL_##k(0) FILL_##k(1) FILL_##k(2) FILL_##k(3) FILL_##k(4) FILL_##k(5))
#define K8(k) KERNEL(k_##k##_8, FILL_##k(0) FILL_##k(1) FILL_##k(2) FILL_##k(3) FILL_##k(4) FILL_##k(5) FILL_##k(6) FILL_##k(7))
#define KALL(k) K1(k) K2(k) K4(k) K5(k) K6(k) K8(k)

KERNEL(k_none, "")
KALL(fma) KALL(exp) KALL(cvt) KALL(dot) KALL(max3) KALL(add) KALL(perm) KALL(mov) KALL(pkmul) KALL(dsr) KALL(snop) KALL(salu) KALL(dep)
// mixes as in the attention iteration
KERNEL(k_mix_a, FILL_fma(0) FILL_fma(1) FILL_exp(2) FILL_exp(3) FILL_cvt(4))
KERNEL(k_mix_b, FILL_fma(0) FILL_fma(1) FILL_exp(2) FILL_exp(3))
KERNEL(k_mix_c, FILL_fma(0) FILL_exp(1) FILL_cvt(2) FILL_dot(3))
KERNEL(k_mix_d, FILL_fma(0) FILL_fma(1) FILL_exp(2) FILL_exp(3) FILL_cvt(4) FILL_dsr(5))
KERNEL(k_mix_e, FILL_fma(0) FILL_exp(0) FILL_fma(1) FILL_exp(1))   /* exp reads the fma before it */
KERNEL(k_mix_f, FILL_fma(0) FILL_fma(1) FILL_exp(0) FILL_exp(1))


// the MFMA CONSUMES what a ds_read_b128 delivered: A operand = fragment register r[i], read from LDS `lead` slots before its MFMA
// (two MFMAs per fragment, as in the attention kernel: reads per MFMA = 0.5), counted lgkmcnt wait in front of the first user
#define RD(r)  "ds_read_b128 %[r" #r "], %[la]\n"
#define MU(i, r) "v_mfma_f32_32x32x16_bf16 %[a" #i "], %[r" #r "], %[y], %[a" #i "]\n"
#define W(n) "s_waitcnt lgkmcnt(" #n ")\n"
#define KERNEL_USE(name, BODYSTR)                                                                                              \
    __global__ __launch_bounds__(256, 1) void name(long long *out, int iters) {                                               \
        __shared__ v4f lds[1024];                                                                                              \
        v16f a0 = {}, a1 = {}, a2 = {}, a3 = {}, a4 = {}, a5 = {}, a6 = {}, a7 = {};                                           \
        v4f y = {1.f, 1.f, 1.f, 1.f}, r0 = {}, r1 = {}, r2 = {}, r3 = {};                                                      \
        float f1 = 0.5f, f2 = 0.25f, g0 = 0, g1 = 1, g2 = 2, g3 = 3;                                                           \
        unsigned la = (threadIdx.x & 63) * 16;                                                                                 \
        lds[threadIdx.x] = y;                                                                                                  \
        __syncthreads();                                                                                                       \
        long long t0 = __builtin_readcyclecounter();                                                                           \
        for (int i = 0; i < iters; i++)                                                                                        \
            asm volatile(BODYSTR                                                                                               \
                         : [a0] "+a"(a0), [a1] "+a"(a1), [a2] "+a"(a2), [a3] "+a"(a3), [a4] "+a"(a4), [a5] "+a"(a5), [a6] "+a"(a6),  \
                           [a7] "+a"(a7), [r0] "+v"(r0), [r1] "+v"(r1), [r2] "+v"(r2), [r3] "+v"(r3), [g0] "+v"(g0), [g1] "+v"(g1),  \
                           [g2] "+v"(g2), [g3] "+v"(g3)                                                                        \
                         : [y] "v"(y), [f1] "v"(f1), [f2] "v"(f2), [la] "v"(la) : "scc");                                      \
        asm volatile("s_waitcnt lgkmcnt(0)\ns_nop 15\ns_nop 15" ::: "memory");                                                 \
        long long t1 = __builtin_readcyclecounter();                                                                           \
        float acc = a0[0] + a1[0] + a2[0] + a3[0] + a4[0] + a5[0] + a6[0] + a7[0] + r0[0] + r1[0] + r2[0] + r3[0] + g0 + g1 + g2 + g3; \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                                       \
        if (acc == 12345.678f) out[0] = 0;                                                                                     \
    }
// fragment ring of 4, a read every second MFMA, issued 4 MFMAs (two reads) before its first user: wait lgkmcnt(1)
#define V4 FILL_fma(0) FILL_fma(1) FILL_fma(2) FILL_fma(3)
KERNEL_USE(k_use_lead4, W(1) MU(0, 0) RD(2) MU(1, 0) W(1) MU(2, 1) RD(3) MU(3, 1) W(1) MU(4, 2) RD(0) MU(5, 2) W(1) MU(6, 3) RD(1) MU(7, 3))
KERNEL_USE(k_use_lead4_valu, W(1) MU(0, 0) RD(2) V4 MU(1, 0) V4 W(1) MU(2, 1) RD(3) V4 MU(3, 1) V4 W(1) MU(4, 2) RD(0) V4 MU(5, 2) V4 W(1) MU(6, 3) RD(1) V4 MU(7, 3) V4)
// the same reads, but the MFMAs do not use them (operand = a register nothing writes)
KERNEL_USE(k_nouse_lead4, W(1) MU(0, 0) "ds_read_b128 %[r2], %[la]\n" MU(1, 0) W(1) MU(2, 0) "ds_read_b128 %[r3], %[la]\n" MU(3, 0) W(1) MU(4, 0) "ds_read_b128 %[r2], %[la]\n" MU(5, 0) W(1) MU(6, 0) "ds_read_b128 %[r3], %[la]\n" MU(7, 0))
// lead 2 (one read ahead): wait lgkmcnt(0)
KERNEL_USE(k_use_lead2, W(0) MU(0, 0) RD(1) MU(1, 0) W(0) MU(2, 1) RD(2) MU(3, 1) W(0) MU(4, 2) RD(3) MU(5, 2) W(0) MU(6, 3) RD(0) MU(7, 3))

// ---- the same with TWO waves per SIMD (512-thread workgroup: the GEMM's regime): the partner wave's fillers can use the shadow too
#undef KERNEL
#define KERNEL(name, F)                                                                                                        \
    __global__ __launch_bounds__(512, 2) void name(long long *out, int iters) {                                               \
        __shared__ v4f lds[1024];                                                                                              \
        v16f a0 = {}, a1 = {}, a2 = {}, a3 = {}, a4 = {}, a5 = {}, a6 = {}, a7 = {};                                           \
        v4f x = {1.f, 2.f, 3.f, 4.f}, y = {1.f, 1.f, 1.f, 1.f}, d0 = {};                                                       \
        float f1 = 0.5f, f2 = 0.25f;                                                                                           \
        float g0 = threadIdx.x, g1 = 1, g2 = 2, g3 = 3, g4 = 4, g5 = 5, g6 = 6, g7 = 7;                                        \
        float h0 = 1, h1 = 1, h2 = 2, h3 = 3, h4 = 4, h5 = 5, h6 = 6, h7 = 7;                                                  \
        double p0 = 0, p1 = 1;                                                                                                 \
        unsigned la = (threadIdx.x & 63) * 16, s0 = 0;                                                                         \
        lds[threadIdx.x & 1023] = x;                                                                                           \
        __syncthreads();                                                                                                       \
        long long t0 = __builtin_readcyclecounter();                                                                           \
        for (int i = 0; i < iters; i++)                                                                                        \
            asm volatile(BODY(F)                                                                                               \
                         : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [a4] "+v"(a4), [a5] "+v"(a5), [a6] "+v"(a6),  \
                           [a7] "+v"(a7), [d0] "+v"(d0), [p0] "+v"(p0), [s0] "+s"(s0), [g0] "+v"(g0), [g1] "+v"(g1),            \
                           [g2] "+v"(g2), [g3] "+v"(g3), [g4] "+v"(g4), [g5] "+v"(g5), [g6] "+v"(g6), [g7] "+v"(g7),            \
                           [h0] "+v"(h0), [h1] "+v"(h1), [h2] "+v"(h2), [h3] "+v"(h3), [h4] "+v"(h4), [h5] "+v"(h5),            \
                           [h6] "+v"(h6), [h7] "+v"(h7)                                                                        \
                         : [x] "v"(x), [y] "v"(y), [f1] "v"(f1), [f2] "v"(f2), [la] "v"(la), [p1] "v"(p1) : "scc");            \
        asm volatile("s_waitcnt lgkmcnt(0)\ns_nop 15\ns_nop 15" ::: "memory");                                                 \
        long long t1 = __builtin_readcyclecounter();                                                                           \
        float acc = a0[0] + a1[0] + a2[0] + a3[0] + a4[0] + a5[0] + a6[0] + a7[0] + g0 + g1 + g2 + g3 + g4 + g5 + g6 + g7 + h0 + h1 + h2 + h3 + h4 + h5 + h6 + h7 + d0[0] + (float)p0 + s0; \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                                       \
        if (acc == 12345.678f) out[0] = 0;                                                                                     \
    }
KERNEL(w2_none, "")
KERNEL(w2_fma_2, FILL_fma(0) FILL_fma(1))
KERNEL(w2_fma_4, FILL_fma(0) FILL_fma(1) FILL_fma(2) FILL_fma(3))
KERNEL(w2_fma_6, FILL_fma(0) FILL_fma(1) FILL_fma(2) FILL_fma(3) FILL_fma(4) FILL_fma(5))
KERNEL(w2_fma_8, FILL_fma(0) FILL_fma(1) FILL_fma(2) FILL_fma(3) FILL_fma(4) FILL_fma(5) FILL_fma(6) FILL_fma(7))
KERNEL(w2_fma_8_dsr, FILL_fma(0) FILL_fma(1) FILL_fma(2) FILL_fma(3) FILL_fma(4) FILL_fma(5) FILL_fma(6) FILL_fma(7) FILL_dsr(0))
KERNEL(w2_fma_8_2dsr_salu, FILL_fma(0) FILL_fma(1) FILL_fma(2) FILL_fma(3) FILL_dsr(0) FILL_fma(4) FILL_fma(5) FILL_fma(6) FILL_fma(7) FILL_dsr(0) FILL_salu(0))

// ---- the GEMM's own arithmetic, registers only, two waves per SIMD: per 32x32x64 tile-group one FP6 product MFMA (v_mfma_scale_f32_32x32x64_f8f6f4,
//      fp6 x fp6), one 16-bit scale-tile MFMA and 16 v_fmac (acc += P * S) -- wall-clock rate at the chip's power limit, no memory at all
typedef int v6i_ __attribute__((ext_vector_type(6)));
#define TG(P, S, PN, SN) \
    "v_mfma_scale_f32_32x32x64_f8f6f4 %[" #PN "], %[w6], %[a6], 0, %[mx], %[mx] op_sel_hi:[0,0,0] cbsz:2 blgp:2\n" \
    "v_fmac_f32 %[c0], %[f1], %[f2]\n v_fmac_f32 %[c1], %[f1], %[f2]\n v_fmac_f32 %[c2], %[f1], %[f2]\n v_fmac_f32 %[c3], %[f1], %[f2]\n" \
    "v_fmac_f32 %[c4], %[f1], %[f2]\n v_fmac_f32 %[c5], %[f1], %[f2]\n v_fmac_f32 %[c6], %[f1], %[f2]\n v_fmac_f32 %[c7], %[f1], %[f2]\n" \
    "v_mfma_f32_32x32x16_bf16 %[" #SN "], %[x], %[y], 0\n" \
    "v_fmac_f32 %[c0], %[f1], %[f2]\n v_fmac_f32 %[c1], %[f1], %[f2]\n v_fmac_f32 %[c2], %[f1], %[f2]\n v_fmac_f32 %[c3], %[f1], %[f2]\n" \
    "v_fmac_f32 %[c4], %[f1], %[f2]\n v_fmac_f32 %[c5], %[f1], %[f2]\n v_fmac_f32 %[c6], %[f1], %[f2]\n v_fmac_f32 %[c7], %[f1], %[f2]\n"
#define TG_NOFMA(PN, SN) \
    "v_mfma_scale_f32_32x32x64_f8f6f4 %[" #PN "], %[w6], %[a6], 0, %[mx], %[mx] op_sel_hi:[0,0,0] cbsz:2 blgp:2\n" \
    "v_mfma_f32_32x32x16_bf16 %[" #SN "], %[x], %[y], 0\n"
#define GEMM_KERNEL(name, BODYSTR) GEMM_KERNEL_T(name, BODYSTR, 512)
#define GEMM_KERNEL_T(name, BODYSTR, THREADS)                                                                                 \
    __global__ __launch_bounds__(THREADS) void name(long long *out, int iters) {                                               \
        v16f p0 = {}, p1 = {}, s0 = {}, s1 = {};                                                                               \
        v4f x = {1.f, 2.f, 3.f, 4.f}, y = {1.f, 1.f, 1.f, 1.f};                                                                \
        v6i_ w6 = {0x11111111, 0x22222222, 0x12121212, 0x21212121, 0x11221122, 0x22112211}, a6 = w6;                           \
        int mx = 0x7f7f7f7f;                                                                                                   \
        float c0 = threadIdx.x, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7, f1 = 1.0001f, f2 = 0.5f;  /* (the fmac sources are plain registers: round 2 measured the same time with the MFMA results as sources) */ \
        long long t0 = __builtin_readcyclecounter();                                                                           \
        for (int i = 0; i < iters; i++)                                                                                        \
            asm volatile(BODYSTR                                                                                               \
                         : [p0] "+v"(p0), [p1] "+v"(p1), [s0] "+v"(s0), [s1] "+v"(s1), [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2),  \
                           [c3] "+v"(c3), [c4] "+v"(c4), [c5] "+v"(c5), [c6] "+v"(c6), [c7] "+v"(c7)                          \
                         : [x] "v"(x), [y] "v"(y), [w6] "v"(w6), [a6] "v"(a6), [mx] "v"(mx), [f1] "v"(f1), [f2] "v"(f2));              \
        long long t1 = __builtin_readcyclecounter();                                                                           \
        float acc = p0[0] + p1[0] + s0[0] + s1[0] + c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;                                     \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                                       \
        if (acc == 12345.678f) out[0] = 0;                                                                                     \
    }
GEMM_KERNEL(g_full, TG(p0, s0, p1, s1) TG(p1, s1, p0, s0) TG(p0, s0, p1, s1) TG(p1, s1, p0, s0))
GEMM_KERNEL(g_mfma_only, TG_NOFMA(p1, s1) TG_NOFMA(p0, s0) TG_NOFMA(p1, s1) TG_NOFMA(p0, s0))
// the same arithmetic at one, three and four waves per SIMD (round 4: does more thread-level parallelism fill the issue port better than two waves do?)
GEMM_KERNEL_T(g_full_w1, TG(p0, s0, p1, s1) TG(p1, s1, p0, s0) TG(p0, s0, p1, s1) TG(p1, s1, p0, s0), 256)
GEMM_KERNEL_T(g_full_w3, TG(p0, s0, p1, s1) TG(p1, s1, p0, s0) TG(p0, s0, p1, s1) TG(p1, s1, p0, s0), 768)
GEMM_KERNEL_T(g_full_w4, TG(p0, s0, p1, s1) TG(p1, s1, p0, s0) TG(p0, s0, p1, s1) TG(p1, s1, p0, s0), 1024)

struct Entry { const char *name; void (*fn)(long long *, int); };
#define E1(k) {#k "_1", k_##k##_1}, {#k "_2", k_##k##_2}, {#k "_4", k_##k##_4}, {#k "_5", k_##k##_5}, {#k "_6", k_##k##_6}, {#k "_8", k_##k##_8},

int main(int argc, char **argv) {
    if (argc >= 3 && !strcmp(argv[1], "sustain")) {
        // filler_probe sustain <seconds> [mfma|gemm]: keep the chip on the pure-MFMA loop (two waves per SIMD) or on the GEMM's arithmetic for that long
        // (rocm-smi clock / power sampling from outside) and print the in-kernel counter rate against the wall clock
        const double secs = atof(argv[2]);
        const bool gemm = argc >= 4 && !strcmp(argv[3], "gemm");
        long long *d;
        hipMalloc(&d, 256 * sizeof(long long));
        hipEvent_t ev0, ev1;
        hipEventCreate(&ev0); hipEventCreate(&ev1);
        const int it = 20000;
        double total = 0;
        float ms = 0;
        while (total < secs * 1e3) {
            hipEventRecord(ev0, 0);
            if (gemm) hipLaunchKernelGGL(g_full, dim3(256), dim3(512), 0, 0, d, it);
            else hipLaunchKernelGGL(w2_none, dim3(256), dim3(512), 0, 0, d, it);
            hipEventRecord(ev1, 0);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms, ev0, ev1);
            total += ms;
        }
        std::vector<long long> h(256);
        hipMemcpy(h.data(), d, 256 * sizeof(long long), hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += v;
        const double mfma_per_simd = gemm ? (double)it * 4 * 2 * 2 : (double)it * 8 * 2;  // both waves of a SIMD
        printf("{\"sustain\":\"%s\",\"launch_ms\":%.3f,\"s_memtime_GHz\":%.3f,\"ns_per_simd_mfma\":%.3f,\"GHz_if_32_cycles_per_mfma\":%.3f}\n", gemm ? "gemm arithmetic" : "pure mfma",
               ms, s / 256 / (ms * 1e-3) / 1e9, ms * 1e6 / mfma_per_simd, 32.0 * mfma_per_simd / (ms * 1e6));
        return 0;
    }
    std::vector<Entry> es = {{"none", k_none}, E1(fma) E1(exp) E1(cvt) E1(dot) E1(max3) E1(add) E1(perm) E1(mov) E1(pkmul) E1(dsr) E1(snop) E1(salu) E1(dep)
                             {"mix_2fma_2exp_cvt", k_mix_a}, {"mix_2fma_2exp", k_mix_b}, {"mix_fma_exp_cvt_dot", k_mix_c},
                             {"mix_2fma_2exp_cvt_dsr", k_mix_d}, {"mix_fma_exp_fma_exp_dependent", k_mix_e}, {"mix_fma_fma_exp_exp_dependent", k_mix_f},
                             {"use_lead4 (MFMA reads ds data)", k_use_lead4}, {"use_lead4 + 4 fma per MFMA", k_use_lead4_valu},
                             {"nouse_lead4 (same reads, unused)", k_nouse_lead4}, {"use_lead2", k_use_lead2}};
    std::vector<Entry> es2 = {{"2 waves/SIMD: none", w2_none}, {"2 waves/SIMD: fma_2", w2_fma_2}, {"2 waves/SIMD: fma_4", w2_fma_4}, {"2 waves/SIMD: fma_6", w2_fma_6},
                              {"2 waves/SIMD: fma_8", w2_fma_8}, {"2 waves/SIMD: fma_8 + ds_read", w2_fma_8_dsr}, {"2 waves/SIMD: fma_8 + 2 ds_read + salu", w2_fma_8_2dsr_salu}};
    setvbuf(stdout, nullptr, _IONBF, 0);
    if (argc >= 2 && !strcmp(argv[1], "gemm")) { es.clear(); es2.clear(); }  // only the GEMM-arithmetic kernels
    long long *d;
    const int G = 256, iters = 2000;
    hipMalloc(&d, G * sizeof(long long));
    std::vector<long long> h(G);
    for (auto &e : es) {
        printf("%-32s ...\r", e.name);
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(e.fn, dim3(G), dim3(256), 0, 0, d, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), d, G * sizeof(long long), hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += v;
        printf("%-32s %7.2f cycles / MFMA   (%s)\n", e.name, s / G / (iters * 8.0), hipGetErrorString(hipGetLastError()));
    }
    for (auto &e : es2) {  // 512 threads: the cycles are per MFMA of ONE wave; the SIMD retires two of them in that time
        hipEvent_t ev0, ev1;
        hipEventCreate(&ev0); hipEventCreate(&ev1);
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(ev0, 0);
            hipLaunchKernelGGL(e.fn, dim3(G), dim3(512), 0, 0, d, iters * 10);
            hipEventRecord(ev1, 0);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms, ev0, ev1);
        }
        printf("%-40s wall %.3f ms for %d x 8 waves x %d MFMAs = %.0f TFLOP/s\n", e.name, ms, G, iters * 10 * 8, (double)G * 8 * iters * 10 * 8 * 32768 / (ms * 1e-3) / 1e12);
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(e.fn, dim3(G), dim3(512), 0, 0, d, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), d, G * sizeof(long long), hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += v;
        printf("%-40s %7.2f cycles / MFMA of one wave = %6.2f per MFMA of the SIMD   (%s)\n", e.name, s / G / (iters * 8.0), s / G / (iters * 16.0), hipGetErrorString(hipGetLastError()));
    }
    struct { const char *name; void (*fn)(long long *, int); int threads; } gs[] = {{"GEMM arithmetic: P + S + 16 v_fmac per tile-group", g_full, 512}, {"GEMM arithmetic: P + S only", g_mfma_only, 512},
        {"GEMM arithmetic, 1 wave per SIMD", g_full_w1, 256}, {"GEMM arithmetic, 3 waves per SIMD", g_full_w3, 768}, {"GEMM arithmetic, 4 waves per SIMD", g_full_w4, 1024}};
    for (auto &e : gs) {
        hipEvent_t ev0, ev1;
        hipEventCreate(&ev0); hipEventCreate(&ev1);
        float ms = 0;
        const int it = iters * 5;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(ev0, 0);
            hipLaunchKernelGGL(e.fn, dim3(G), dim3(e.threads), 0, 0, d, it);
            hipEventRecord(ev1, 0);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms, ev0, ev1);
        }
        hipMemcpy(h.data(), d, G * sizeof(long long), hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += v;
        const double tg = (double)G * (e.threads / 64) * it * 4;  // tile-groups (32 x 32 x 64) of the launch
        printf("%-52s %6.1f cycles per tile-group of one wave, wall %.3f ms = %.0f TOP/s (2 x 32 x 32 x 64 per tile-group), clock %.2f GHz  (%s)\n", e.name,
               s / G / (it * 4.0), ms, tg * 2 * 32 * 32 * 64 / (ms * 1e-3) / 1e12, s / G / (ms * 1e-3) / 1e9, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
