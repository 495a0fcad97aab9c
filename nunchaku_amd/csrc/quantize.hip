// svdq_quantize_w4a4_act_fuse_lora: 16-bit activations -> FP6 operand image of the 4-bit codes +
// per-(row, group) scales, fused with the rank-R low-rank down projection.
//
// Replaces the reference's quantize_w4a4_fuse_lora_kernel (gemm_w4a4.cuh:1097-1184; launch
// gemm_w4a4_launch_impl.cuh:451-521).  Arithmetic (DESIGN.md "Quantiser"):
//   lora_act[m, r] = sum_k x[m,k] * lora_down[k,r]        16-bit MFMA, fp32 accumulate, on raw x
//   x_hat = round16(x / smooth)                           IEEE fp32 divide (reference: __fdividef)
//   amax  = max_{k in group} |x_hat|;  scale = amax * (1/7)  (fp32);  ascales = round16(scale)
//   q     = clamp(rne(x_hat * (1/scale)), -8, 7)          IEEE reciprocal (reference: rcp.approx)
//
// MI355X design: the op is HBM-bound (reads M*K*2 B, writes M*K*3/4 B).  The unit of work is one
// F6 chunk = 32 rows x 128 channels (svdq_common.h), owned by ONE wave: lane (r, h) loads exactly the
// channels its lane record holds, the group maximum needs a single cross-lane step (lane ^ 32) and
// the three 16-byte stores of a lane are three fully coalesced 1 KiB wave stores.  A workgroup =
// 4 waves on one 32-row tile and a slice of K; the low-rank partial sums of the slice are combined
// in LDS in a fixed order and added to lora_act with fp32 atomics when K is split over several
// workgroups (the reference reduces K/128 CTAs the same way, lora.cuh:253-339).
#include "svdq_common.h"
#include <stdlib.h>

namespace svdq {

// grouped launch: rows >= split_rows read a second input and use a second parameter set (svdq_quantize_args.x2 ...)
struct QuantSecond {
    const void *x, *smooth, *lora_down, *mod_scale, *mod_shift;
    const float *ln_stats;
    int M, ldx, split_rows; // M: valid rows of the second stream; x == nullptr: off
};

template <int DT, int RT32 /* 32-rank tiles held in registers */, int OCC /* workgroups per CU the register budget allows */>
__global__ __launch_bounds__(256, OCC) void quantize_kernel(const typename Half<DT>::T *__restrict__ x,
                                                       const typename Half<DT>::T *__restrict__ smooth,
                                                       const typename Half<DT>::T *__restrict__ lora_down, // [R][K]
                                                       uint8_t *__restrict__ act,
                                                       typename Half<DT>::T *__restrict__ ascales,
                                                       float *__restrict__ lora_act, int M, int K, int R, int ldx,
                                                       int chunks_per_wg, int use_atomics,
                                                       const float *__restrict__ ln_stats,
                                                       const typename Half<DT>::T *__restrict__ mod_scale,
                                                       const typename Half<DT>::T *__restrict__ mod_shift, const QuantSecond s2) {
    using T = typename Half<DT>::T;
    using V8 = typename Half<DT>::V8;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int KP = K / 128;
    const int slices = (KP + chunks_per_wg - 1) / chunks_per_wg;
    const int rt = blockIdx.x / slices, slice = blockIdx.x % slices;
    // grouped launch: this row tile's stream (block-uniform); outputs stay addressed by the joint row tile index rt
    const bool second = s2.x != nullptr && rt * 32 >= s2.split_rows;
    if (second) {
        x = (const T *)s2.x; smooth = (const T *)s2.smooth; lora_down = (const T *)s2.lora_down;
        ln_stats = s2.ln_stats; mod_scale = (const T *)s2.mod_scale; mod_shift = (const T *)s2.mod_shift;
        ldx = s2.ldx; M = s2.M;
    }
    const int row = rt * 32 + r - (second ? s2.split_rows : 0); // row inside this stream's input
    const bool valid = row < M;
    const int kp_end = min(KP, (slice + 1) * chunks_per_wg);

    constexpr int NT = RT32 > 0 ? RT32 : 1;
    v16f accL[NT];
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) accL[i][j] = 0.f;

    const T *xrow = x + (size_t)row * ldx;
    // fused AdaLayerNormZero front end: per-row statistics of the LayerNorm this projection follows
    float ln_mean = 0.f, ln_rstd = 0.f;
    if (ln_stats && valid) {
        const float2 st = *reinterpret_cast<const float2 *>(ln_stats + 2 * (size_t)row);
        ln_mean = st.x;
        ln_rstd = st.y;
    }

    for (int kp = slice * chunks_per_wg + wave; kp < kp_end; kp += 4) {
        uint32_t rec[12];
#pragma unroll
        for (int i = 0; i < 12; i++) rec[i] = 0;
        T sc16[2];
        // ---- every global load of the chunk is issued before anything is consumed (the kernel is
        //      latency-bound otherwise: two dependent round trips per chunk): activations, smoothing factors
        //      and the first 32 ranks of lora_down for both groups
        u16x4 xv[2][8], sv[2][8], bv[2][4][2], msv[2][8], mhv[2][8];
#pragma unroll
        for (int grp = 0; grp < 2; grp++) {
            const int kbase = kp * 128 + grp * 64 + 4 * h; // + 32t + 8c + e
#pragma unroll
            for (int tc = 0; tc < 8; tc++) {
                if (valid) xv[grp][tc] = *reinterpret_cast<const u16x4 *>(xrow + kbase + 8 * tc);
                else xv[grp][tc] = u16x4{0, 0, 0, 0};
                if (smooth) sv[grp][tc] = *reinterpret_cast<const u16x4 *>(smooth + kbase + 8 * tc);
            }
            if (ln_stats) { // modulation vectors of this group: requested with everything else, consumed below
#pragma unroll
                for (int tc = 0; tc < 8; tc++) {
                    msv[grp][tc] = *reinterpret_cast<const u16x4 *>(mod_scale + kbase + 8 * tc);
                    mhv[grp][tc] = *reinterpret_cast<const u16x4 *>(mod_shift + kbase + 8 * tc);
                }
            }
            if constexpr (RT32 > 0) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (r < R) {
                        const T *ld = lora_down + (size_t)r * K + kbase + 16 * q;
                        bv[grp][q][0] = *reinterpret_cast<const u16x4 *>(ld);
                        bv[grp][q][1] = *reinterpret_cast<const u16x4 *>(ld + 8);
                    } else {
                        bv[grp][q][0] = u16x4{0, 0, 0, 0};
                        bv[grp][q][1] = u16x4{0, 0, 0, 0};
                    }
                }
            }
        }
        if (ln_stats) { // x <- round16(round16(round16((x - mean) * rstd) * scale) + shift); padded rows stay 0
#pragma unroll
            for (int grp = 0; grp < 2; grp++)
#pragma unroll
                for (int tc = 0; tc < 8; tc++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float ln = round16<T>((h2f(hfrom<T>(xv[grp][tc][e])) - ln_mean) * ln_rstd);
                        const float y = round16<T>(ln * h2f(hfrom<T>(msv[grp][tc][e]))) + h2f(hfrom<T>(mhv[grp][tc][e]));
                        if (valid) xv[grp][tc][e] = hbits(f2h<T>(y));
                    }
        }
#pragma unroll
        for (int grp = 0; grp < 2; grp++) {
            const int kbase = kp * 128 + grp * 64 + 4 * h;
            if constexpr (RT32 > 0) {
                // D[m][rank] += x[m][k] * lora_down[k][rank]; MFMA q consumes pieces tc = 2q, 2q+1 of every
                // lane as k-slots 8h .. 8h+7 (any k order works as long as both operands agree)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    V8 a;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        a[e] = hfrom<T>(xv[grp][2 * q][e]);
                        a[4 + e] = hfrom<T>(xv[grp][2 * q + 1][e]);
                    }
#pragma unroll
                    for (int t32 = 0; t32 < RT32; t32++) {
                        if (t32 * 32 < R) { // wave-uniform
                            const int rank = t32 * 32 + r;
                            V8 b;
                            if (t32 == 0) {
#pragma unroll
                                for (int e = 0; e < 4; e++) {
                                    b[e] = hfrom<T>(bv[grp][q][0][e]);
                                    b[4 + e] = hfrom<T>(bv[grp][q][1][e]);
                                }
                            } else if (rank < R) {
                                const T *ld = lora_down + (size_t)rank * K + kbase + 16 * q;
                                u16x4 b0 = *reinterpret_cast<const u16x4 *>(ld);
                                u16x4 b1 = *reinterpret_cast<const u16x4 *>(ld + 8);
#pragma unroll
                                for (int e = 0; e < 4; e++) {
                                    b[e] = hfrom<T>(b0[e]);
                                    b[4 + e] = hfrom<T>(b1[e]);
                                }
                            } else {
#pragma unroll
                                for (int e = 0; e < 8; e++) b[e] = (T)0.f;
                            }
                            accL[t32] = Half<DT>::mfma32(a, b, accL[t32]);
                        }
                    }
                }
            }

            float xh[32];
#pragma unroll
            for (int tc = 0; tc < 8; tc++) {
                if (smooth) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float sm = h2f(hfrom<T>(sv[grp][tc][e]));
                        xh[4 * tc + e] = round16<T>(div_rn(h2f(hfrom<T>(xv[grp][tc][e])), sm, __builtin_amdgcn_rcpf(sm)));
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) xh[4 * tc + e] = h2f(hfrom<T>(xv[grp][tc][e]));
                }
            }
            float amax = 0.f;
#pragma unroll
            for (int j = 0; j < 32; j++) amax = fmaxf(amax, fabsf(xh[j]));
            amax = fmaxf(amax, __shfl_xor(amax, 32));

            const float scale = amax * (1.0f / 7.0f);
            const float rscale = scale == 0.f ? 0.f : 1.0f / scale;
            sc16[grp] = f2h<T>(scale);
            // q = rne(x_hat / scale) as FP6 e2m3 = q/8, packed 32 x 6 bits: ONE v_cvt_scalef32_2xpk16_fp6_f32 (scale 8 =
            // divide by 2^3; element 2i from the first source, 2i+1 from the second; RNE; |x_hat/scale| <= 7 by
            // construction so the -8..7 clamp never fires; probed on gfx950 with tools/ubench5.hip) instead of
            // rint / clamp / sign-magnitude encode / shift / or per element
            v16f ev, od;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                ev[i] = xh[2 * i] * rscale;
                od[i] = xh[2 * i + 1] * rscale;
            }
            const v6i pk = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(ev, od, 8.0f);
#pragma unroll
            for (int i = 0; i < 6; i++) rec[6 * grp + i] = (uint32_t)pk[i];
        }
        uint8_t *dst = act + ((size_t)rt * KP + kp) * F6_CHUNK + (size_t)lane * 16;
        *reinterpret_cast<uint4 *>(dst) = make_uint4(rec[0], rec[1], rec[2], rec[3]);
        *reinterpret_cast<uint4 *>(dst + F6_PLANE) = make_uint4(rec[4], rec[5], rec[6], rec[7]);
        *reinterpret_cast<uint4 *>(dst + 2 * F6_PLANE) = make_uint4(rec[8], rec[9], rec[10], rec[11]);
        // S image: [rt][kp][grp][32]; lane (r, h) writes group h
        ascales[(((size_t)rt * KP + kp) * 2 + h) * 32 + r] = sc16[h];
    }

    if constexpr (RT32 > 0) {
        // combine the four waves' partial sums in a fixed order
        __shared__ v16f red[4][64];
#pragma unroll
        for (int t32 = 0; t32 < RT32; t32++) {
            if (t32 * 32 < R) { // block-uniform
                red[wave][lane] = accL[t32];
                __syncthreads();
                if (wave == (t32 & 3)) {
                    v16f s = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
                    const int rank = t32 * 32 + r; // C layout: col = lane & 31, row = (i&3) + 8*(i>>2) + 4*(lane>>5)
                    if (rank < R) {
#pragma unroll
                        for (int i = 0; i < 16; i++) {
                            const int m = rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                            float *dst = lora_act + (size_t)m * R + rank;
                            if (use_atomics) unsafeAtomicAdd(dst, s[i]);
                            else *dst = s[i];
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
}

template <int DT>
static int launch_quantize(const svdq_quantize_args *a, hipStream_t st) {
    using T = typename Half<DT>::T;
    const int KP = a->K / 128, tiles = a->M_pad / 32;
    // enough workgroups to fill 256 CUs several times over, but at least one chunk per wave
#ifdef SVDQ_ABLATE
    static const int cpw_env = getenv("SVDQ_QUANT_CPW") ? atoi(getenv("SVDQ_QUANT_CPW")) : 0; // experiment knob
#else
    constexpr int cpw_env = 0;
#endif
    int cpw = cpw_env > 0 ? cpw_env : 4; // chunks per workgroup = 4 waves x 1 chunk: many short waves hide the HBM round trip
    while ((long)tiles * ((KP + cpw - 1) / cpw) > 8192) cpw *= 2;
    if (cpw > KP) cpw = ((KP + 3) / 4) * 4;
    const int slices = (KP + cpw - 1) / cpw;
    const int atomics = slices > 1;
    if (a->R > 0 && atomics && !a->lora_act_zeroed) {
        // the reference zeroes the buffer inside the op as well (launch_impl.cuh:487)
        int rc = hip_check(hipMemsetAsync(a->lora_act, 0, (size_t)a->M_pad * a->R * sizeof(float), st), "svdq_quantize memset");
        if (rc) return rc;
    }
    dim3 grid(tiles * slices), block(256);
    const int rt32 = (a->R + 31) / 32;
    QuantSecond s2{a->x2, a->smooth2, a->lora_down2, a->mod_scale2, a->mod_shift2, a->ln_stats2, a->M2, a->ldx2, a->split_rows};
#ifdef SVDQ_ABLATE
    static const int occ_env = getenv("SVDQ_QUANT_OCC") ? atoi(getenv("SVDQ_QUANT_OCC")) : 0; // experiment knob
#else
    constexpr int occ_env = 0;
#endif
#ifdef SVDQ_ABLATE
#define SVDQ_LAUNCH_Q_OCC4(RT)                                                                                       \
    if (RT <= 2 && occ_env == 4)                                                                                     \
    hipLaunchKernelGGL((quantize_kernel<DT, RT, (RT <= 2 ? 4 : 1)>), grid, block, 0, st, (const T *)a->x, (const T *)a->smooth, \
                       (const T *)a->lora_down, (uint8_t *)a->act, (T *)a->ascales, a->lora_act, a->M, a->K, a->R,    \
                       a->ldx, cpw, atomics, a->ln_stats, (const T *)a->mod_scale, (const T *)a->mod_shift, s2);          \
    else
#else
#define SVDQ_LAUNCH_Q_OCC4(RT)
    (void)occ_env;
#endif
#define SVDQ_LAUNCH_Q(RT)                                                                                            \
    SVDQ_LAUNCH_Q_OCC4(RT)                                                                                           \
    hipLaunchKernelGGL((quantize_kernel<DT, RT, 1>), grid, block, 0, st, (const T *)a->x, (const T *)a->smooth,          \
                       (const T *)a->lora_down, (uint8_t *)a->act, (T *)a->ascales, a->lora_act, a->M, a->K, a->R,    \
                       a->ldx, cpw, atomics, a->ln_stats, (const T *)a->mod_scale, (const T *)a->mod_shift, s2)
    if (rt32 == 0) SVDQ_LAUNCH_Q(0);
    else if (rt32 <= 1) SVDQ_LAUNCH_Q(1);
    else if (rt32 <= 2) SVDQ_LAUNCH_Q(2);
    else if (rt32 <= 4) SVDQ_LAUNCH_Q(4);
    else SVDQ_LAUNCH_Q(8);
#undef SVDQ_LAUNCH_Q
#undef SVDQ_LAUNCH_Q_OCC4
    return hip_check(hipGetLastError(), "svdq_quantize_w4a4_act_fuse_lora launch");
}

} // namespace svdq

using namespace svdq;

extern "C" int svdq_quantize_w4a4_act_fuse_lora(const svdq_quantize_args *a, void *stream) {
    if (!a) { set_error("svdq_quantize: args is NULL"); return SVDQ_E_INVALID; }
    if (a->fp4) { set_error("svdq_quantize: fp4 (NVFP4) is not supported on gfx950"); return SVDQ_E_UNSUPPORTED; }
    if (a->fuse_glu) { set_error("svdq_quantize: fuse_glu is not supported"); return SVDQ_E_UNSUPPORTED; }
    if (!a->x || !a->act || !a->ascales) { set_error("svdq_quantize: x, act and ascales are required"); return SVDQ_E_INVALID; }
    if (a->M <= 0 || a->M_pad < a->M || a->M_pad % 256) {
        set_error("svdq_quantize: need 0 < M=%d <= M_pad=%d and M_pad %% 256 == 0", a->M, a->M_pad);
        return SVDQ_E_INVALID;
    }
    if (a->K <= 0 || a->K % 128) { set_error("svdq_quantize: K=%d must be a positive multiple of 128", a->K); return SVDQ_E_INVALID; }
    if (a->ldx < a->K || a->ldx % 4) { set_error("svdq_quantize: ldx=%d must be >= K and a multiple of 4", a->ldx); return SVDQ_E_INVALID; }
    if (a->R < 0 || a->R % 16 || a->R > 256) { set_error("svdq_quantize: R=%d must be a multiple of 16 in [0, 256]", a->R); return SVDQ_E_INVALID; }
    if (a->R > 0 && (!a->lora_down || !a->lora_act)) { set_error("svdq_quantize: R > 0 needs lora_down and lora_act"); return SVDQ_E_INVALID; }
    if (((uintptr_t)a->act) & 15) { set_error("svdq_quantize: act must be 16-byte aligned"); return SVDQ_E_INVALID; }
    if (((uintptr_t)a->x | (uintptr_t)a->lora_down | (uintptr_t)a->smooth) & 7) {
        set_error("svdq_quantize: x, lora_down and smooth must be 8-byte aligned");
        return SVDQ_E_INVALID;
    }
    if (a->dtype != SVDQ_BF16 && a->dtype != SVDQ_FP16) {
        set_error("svdq_quantize: unknown dtype %d", a->dtype);
        return SVDQ_E_INVALID;
    }
    if ((a->ln_stats != nullptr) != (a->mod_scale != nullptr) || (a->ln_stats != nullptr) != (a->mod_shift != nullptr)) {
        set_error("svdq_quantize: ln_stats, mod_scale and mod_shift must be given together");
        return SVDQ_E_INVALID;
    }
    if (((uintptr_t)a->ln_stats | (uintptr_t)a->mod_scale | (uintptr_t)a->mod_shift) & 7) {
        set_error("svdq_quantize: ln_stats, mod_scale and mod_shift must be 8-byte aligned");
        return SVDQ_E_INVALID;
    }
    if (a->x2) { // grouped launch
        if (a->split_rows <= 0 || a->split_rows % 256 || a->M != a->split_rows || a->M2 <= 0 || a->split_rows + a->M2 > a->M_pad ||
            a->ldx2 < a->K || a->ldx2 % 4) {
            set_error("svdq_quantize: grouped launch needs M == split_rows (a multiple of 256), 0 < M2, split_rows + M2 <= M_pad, ldx2 >= K");
            return SVDQ_E_INVALID;
        }
        if ((a->smooth != nullptr) != (a->smooth2 != nullptr) || (a->R > 0 && !a->lora_down2) ||
            (a->ln_stats != nullptr) != (a->ln_stats2 != nullptr) || (a->ln_stats2 && (!a->mod_scale2 || !a->mod_shift2))) {
            set_error("svdq_quantize: grouped launch: the second parameter set must mirror the first");
            return SVDQ_E_INVALID;
        }
        if (((uintptr_t)a->x2 | (uintptr_t)a->lora_down2 | (uintptr_t)a->smooth2 | (uintptr_t)a->ln_stats2 | (uintptr_t)a->mod_scale2 |
             (uintptr_t)a->mod_shift2) & 7) {
            set_error("svdq_quantize: second parameter set must be 8-byte aligned");
            return SVDQ_E_INVALID;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    // algorithmic bytes: x in, codes + scales + lora_act out, lora_down in (once)
    const double bytes = (double)(a->M + (a->x2 ? a->M2 : 0)) * a->K * 2 + (double)a->M_pad * a->K * 3 / 4 + (double)a->M_pad * (a->K / 64) * 2 +
                         (double)a->M_pad * a->R * 4 + (double)a->K * a->R * 2;
    const int prof = prof_begin(1, bytes, st);
    int rc = a->dtype == SVDQ_BF16 ? launch_quantize<SVDQ_BF16>(a, st) : launch_quantize<SVDQ_FP16>(a, st);
    prof_end(prof, st);
    return rc;
}
