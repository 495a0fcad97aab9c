// svdq_quantize_w4a4_act_fuse_lora: 16-bit activations -> packed int4 codes + per-(row, group)
// scales, fused with the rank-R low-rank down projection.
//
// Replaces the reference's quantize_w4a4_fuse_lora_kernel (gemm_w4a4.cuh:1097-1184; launch
// gemm_w4a4_launch_impl.cuh:451-521).  Arithmetic (DESIGN.md "Quantiser"):
//   lora_act[m, r] = sum_k x[m,k] * lora_down[k,r]        16-bit MFMA, fp32 accumulate, on raw x
//   x_hat = round16(x / smooth)                           IEEE fp32 divide (reference: __fdividef)
//   amax  = max_{k in group} |x_hat|;  scale = amax * (1/7)  (fp32);  ascales = round16(scale)
//   q     = clamp(rne(x_hat * (1/scale)), -8, 7)          IEEE reciprocal (reference: rcp.approx)
//
// MI355X design: this op is HBM-bound (reads M*K*2 B, writes M*K/2 B).  One workgroup owns a
// 16-row tile across ALL of K, so the low-rank projection needs no atomics and lora_act is
// bit-deterministic (the reference reduces K/128 CTAs with fp32 red.add and is not).  The four
// waves of a workgroup stride over the 64-channel groups; their partial low-rank sums are
// combined in LDS in a fixed order.  Codes are written in the T16 tile order (svdq_common.h):
// each wave store instruction writes one contiguous 512-byte MFMA operand tile.
#include "svdq_common.h"

namespace svdq {

template <int DT, int RT /* 16-rank tiles held in registers */>
__global__ __launch_bounds__(256) void quantize_kernel(const typename Half<DT>::T *__restrict__ x,
                                                       const typename Half<DT>::T *__restrict__ smooth,
                                                       const typename Half<DT>::T *__restrict__ lora_down, // [R][K]
                                                       uint8_t *__restrict__ act,
                                                       typename Half<DT>::T *__restrict__ ascales,
                                                       float *__restrict__ lora_act, int M, int M_pad, int K, int R,
                                                       int ldx) {
    using T = typename Half<DT>::T;
    using V8 = typename Half<DT>::V8;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int rl = lane & 15, ks = lane >> 4;
    const int tile = blockIdx.x;
    const int row = tile * 16 + rl;
    const bool valid = row < M;
    const int G = K / GROUP;
    const int rtiles = R / 16;

    v4f accL[RT > 0 ? RT : 1];
#pragma unroll
    for (int i = 0; i < (RT > 0 ? RT : 1); i++) accL[i] = v4f{0.f, 0.f, 0.f, 0.f};

    const T *xrow = x + (size_t)row * ldx;
    uint8_t *act_tile = act + (((size_t)(tile >> 3) * G) * 8 + (tile & 7)) * 512 + (size_t)lane * 8;

    for (int g = wave; g < G; g += 4) {
        const int kbase = g * GROUP + ks * 16;
        V8 xa, xb;
        if (valid) {
            xa = *reinterpret_cast<const V8 *>(xrow + kbase);
            xb = *reinterpret_cast<const V8 *>(xrow + kbase + 8);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) { xa[j] = (T)0.f; xb[j] = (T)0.f; }
        }

        if constexpr (RT > 0) {
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                if (rt < rtiles) { // wave-uniform
                    const T *ld = lora_down + (size_t)(rt * 16 + rl) * K + kbase;
                    V8 b0 = *reinterpret_cast<const V8 *>(ld);
                    V8 b1 = *reinterpret_cast<const V8 *>(ld + 8);
                    accL[rt] = Half<DT>::mfma(xa, b0, accL[rt]);
                    accL[rt] = Half<DT>::mfma(xb, b1, accL[rt]);
                }
            }
        }

        float xh[16];
        if (smooth) {
            V8 sa = *reinterpret_cast<const V8 *>(smooth + kbase);
            V8 sb = *reinterpret_cast<const V8 *>(smooth + kbase + 8);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                xh[j] = round16<T>(h2f(xa[j]) / h2f(sa[j]));
                xh[8 + j] = round16<T>(h2f(xb[j]) / h2f(sb[j]));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                xh[j] = h2f(xa[j]);
                xh[8 + j] = h2f(xb[j]);
            }
        }
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 16; j++) amax = fmaxf(amax, fabsf(xh[j]));
        amax = fmaxf(amax, __shfl_xor(amax, 16));
        amax = fmaxf(amax, __shfl_xor(amax, 32));

        const float scale = amax * (1.0f / 7.0f);
        const float rscale = scale == 0.f ? 0.f : 1.0f / scale;
        uint32_t w0 = 0, w1 = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int q0 = (int)fminf(fmaxf(rintf(xh[j] * rscale), -8.f), 7.f);
            int q1 = (int)fminf(fmaxf(rintf(xh[8 + j] * rscale), -8.f), 7.f);
            w0 |= (uint32_t)(q0 & 15) << (4 * j);
            w1 |= (uint32_t)(q1 & 15) << (4 * j);
        }
        *reinterpret_cast<uint2 *>(act_tile + (size_t)g * 8 * 512) = make_uint2(w0, w1);
        if (ks == 0) ascales[(size_t)g * M_pad + row] = f2h<T>(scale);
    }

    if constexpr (RT > 0) {
        // combine the four waves' partial sums in a fixed order (deterministic)
        __shared__ v4f red[4][64];
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            if (rt < rtiles) { // block-uniform
                red[wave][lane] = accL[rt];
                __syncthreads();
                if (wave == (rt & 3)) {
                    v4f s = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        int m = tile * 16 + (lane >> 4) * 4 + i; // MFMA C layout: row=(lane>>4)*4+i, col=lane&15
                        lora_act[(size_t)m * R + rt * 16 + (lane & 15)] = s[i];
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
    dim3 grid(a->M_pad / 16), block(256);
    const int rtiles = a->R / 16;
#define SVDQ_LAUNCH_Q(RT)                                                                                             \
    hipLaunchKernelGGL((quantize_kernel<DT, RT>), grid, block, 0, st, (const T *)a->x, (const T *)a->smooth,           \
                       (const T *)a->lora_down, (uint8_t *)a->act, (T *)a->ascales, a->lora_act, a->M, a->M_pad, a->K, \
                       a->R, a->ldx)
    if (rtiles == 0) SVDQ_LAUNCH_Q(0);
    else if (rtiles <= 2) SVDQ_LAUNCH_Q(2);
    else if (rtiles <= 4) SVDQ_LAUNCH_Q(4);
    else if (rtiles <= 8) SVDQ_LAUNCH_Q(8);
    else SVDQ_LAUNCH_Q(16);
#undef SVDQ_LAUNCH_Q
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
    if (a->ldx < a->K || a->ldx % 8) { set_error("svdq_quantize: ldx=%d must be >= K and a multiple of 8", a->ldx); return SVDQ_E_INVALID; }
    if (a->R < 0 || a->R % 16 || a->R > 256) { set_error("svdq_quantize: R=%d must be a multiple of 16 in [0, 256]", a->R); return SVDQ_E_INVALID; }
    if (a->R > 0 && (!a->lora_down || !a->lora_act)) { set_error("svdq_quantize: R > 0 needs lora_down and lora_act"); return SVDQ_E_INVALID; }
    if (((uintptr_t)a->x | (uintptr_t)a->act | (uintptr_t)a->lora_down | (uintptr_t)a->smooth) & 15) {
        set_error("svdq_quantize: x, act, lora_down and smooth must be 16-byte aligned");
        return SVDQ_E_INVALID;
    }
    if (a->dtype != SVDQ_BF16 && a->dtype != SVDQ_FP16) {
        set_error("svdq_quantize: unknown dtype %d", a->dtype);
        return SVDQ_E_INVALID;
    }
    hipStream_t st = (hipStream_t)stream;
    // algorithmic bytes: x in, codes + scales + lora_act out, lora_down in (once)
    const double bytes = (double)a->M * a->K * 2 + (double)a->M_pad * a->K / 2 + (double)a->M_pad * (a->K / 64) * 2 +
                         (double)a->M_pad * a->R * 4 + (double)a->K * a->R * 2;
    const int prof = prof_begin(1, bytes, st);
    int rc = a->dtype == SVDQ_BF16 ? launch_quantize<SVDQ_BF16>(a, st) : launch_quantize<SVDQ_FP16>(a, st);
    prof_end(prof, st);
    return rc;
}
