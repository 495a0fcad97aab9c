// svdq_residual_gate_stats: the element-wise glue between the operators of a FLUX block, with the statistics of the
// NEXT LayerNorm produced in the same pass (extension; reference blocks: transformer_flux_v2.py:118-342 use
// torch ops: hidden + gate * attn_out, LayerNorm, (1 + scale) * x + shift).
//
//   t = b ? round16(a + b) : a;   y = a ? round16(res + round16(gate * t)) : res;   out = y;   stats = (mean, rstd) of y
// (the rounding points of the reference's 16-bit torch ops: `attn + mlp`, `gate * x`, `residual + x`,
//  transformer_flux_v2.py:230-251,332-335)
//
// HBM-bound: one pass over the row (3 reads + 1 write of 2 bytes per element); one wave per row keeps the whole
// row in registers (C <= 16384), so mean and variance are the exact two-pass formulas on the stored 16-bit values
// and the LayerNorm itself disappears into the quantiser (svdq_quantize_args.ln_stats).
#include "svdq_common.h"

namespace svdq {

template <int DT, int NV /* 16-byte pieces per lane */>
__global__ __launch_bounds__(256) void residual_kernel(const uint16_t *__restrict__ res, const uint16_t *__restrict__ a,
                                                        const uint16_t *__restrict__ b, const uint16_t *__restrict__ gate,
                                                        uint16_t *__restrict__ out, float *__restrict__ stats, int M, int C, int ld,
                                                        float eps, v4i *__restrict__ zero_ptr, long long zero_vec,
                                                        const uint16_t *__restrict__ res2, const uint16_t *__restrict__ a2,
                                                        const uint16_t *__restrict__ b2, const uint16_t *__restrict__ gate2,
                                                        uint16_t *__restrict__ out2, float *__restrict__ stats2, int M2, int clamp) {
    using T = typename Half<DT>::T;
    // side job: clear the scratch buffer, 16 bytes per thread, grid-strided (before any early return)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < zero_vec; i += (long long)gridDim.x * 256) zero_ptr[i] = v4i{0, 0, 0, 0};
    const int lane = threadIdx.x & 63;
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) { // grouped launch: rows beyond the first problem belong to the second one (wave-uniform)
        row -= M;
        if (!res2 || row >= M2) return;
        res = res2; a = a2; b = b2; gate = gate2; out = out2; stats = stats2;
        clamp >>= 1;
    }
    // fp16 only: y = clip(y, +-65504) like the reference's blocks (transformer_flux_v2.py:254-255, 339-340): an overflowed
    // sum is stored as the largest finite value and the statistics see that value, not inf
    const bool clip = DT == SVDQ_FP16 && (clamp & 1);
    const size_t base = (size_t)row * ld;
    float y[NV][8];
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < NV; v++) {
        const int c = (v * 64 + lane) * 8; // a wave instruction covers 1 KiB of the row
        if (c >= C) { // ragged tail of the last pass (C is a multiple of 8, not necessarily of 512)
#pragma unroll
            for (int e = 0; e < 8; e++) y[v][e] = 0.f;
            continue;
        }
        const u16x8 rv = *reinterpret_cast<const u16x8 *>(res + base + c);
        if (a) {
            const u16x8 av = *reinterpret_cast<const u16x8 *>(a + base + c);
            u16x8 bv, gv;
            if (b) bv = *reinterpret_cast<const u16x8 *>(b + base + c);
            if (gate) gv = *reinterpret_cast<const u16x8 *>(gate + c);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                float t = h2f(hfrom<T>(av[e]));
                if (b) t = round16<T>(t + h2f(hfrom<T>(bv[e])));
                if (gate) t = round16<T>(h2f(hfrom<T>(gv[e])) * t);
                y[v][e] = round16<T>(h2f(hfrom<T>(rv[e])) + t);
                if (clip) y[v][e] = fminf(fmaxf(y[v][e], -65504.f), 65504.f);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; e++) y[v][e] = h2f(hfrom<T>(rv[e]));
        }
        if (out && a) {
            u16x8 ov;
#pragma unroll
            for (int e = 0; e < 8; e++) ov[e] = hbits(f2h<T>(y[v][e]));
            *reinterpret_cast<u16x8 *>(out + base + c) = ov;
        }
#pragma unroll
        for (int e = 0; e < 8; e++) sum += y[v][e];
    }
    if (!stats) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int v = 0; v < NV; v++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float d = (v * 64 + lane) * 8 < C ? y[v][e] - mean : 0.f;
            sq = __builtin_fmaf(d, d, sq);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if (lane == 0) {
        stats[2 * (size_t)row] = mean;
        stats[2 * (size_t)row + 1] = 1.0f / sqrtf(sq / (float)C + eps);
    }
}

template <int DT> static int launch_residual(const svdq_residual_args *p, hipStream_t st) {
    dim3 grid((p->M + (p->res2 ? p->M2 : 0) + 3) / 4), block(256);
#define SVDQ_RES_CASE(NV)                                                                                                       \
    case NV:                                                                                                                    \
        hipLaunchKernelGGL((residual_kernel<DT, NV>), grid, block, 0, st, (const uint16_t *)p->res, (const uint16_t *)p->a,      \
                           (const uint16_t *)p->b, (const uint16_t *)p->gate, (uint16_t *)p->out, p->stats, p->M, p->C, p->ld,   \
                           p->eps, (v4i *)p->zero_ptr, (long long)(p->zero_ptr ? p->zero_bytes / 16 : 0),                          \
                           (const uint16_t *)p->res2, (const uint16_t *)p->a2, (const uint16_t *)p->b2, (const uint16_t *)p->gate2, \
                           (uint16_t *)p->out2, p->stats2, p->M2, p->clamp_fp16);                                                                                           \
        return 0;
    switch ((p->C + 511) / 512) {
        SVDQ_RES_CASE(1) SVDQ_RES_CASE(2) SVDQ_RES_CASE(3) SVDQ_RES_CASE(4) SVDQ_RES_CASE(5) SVDQ_RES_CASE(6) SVDQ_RES_CASE(7) SVDQ_RES_CASE(8)
        SVDQ_RES_CASE(12) SVDQ_RES_CASE(16) SVDQ_RES_CASE(24) SVDQ_RES_CASE(32)
    default: return -1;
    }
#undef SVDQ_RES_CASE
}

} // namespace svdq

using namespace svdq;

extern "C" int svdq_residual_gate_stats(const svdq_residual_args *a, void *stream) {
    if (!a) { set_error("svdq_residual_gate_stats: args is NULL"); return SVDQ_E_INVALID; }
    if (!a->res || (!a->out && !a->stats)) { set_error("svdq_residual_gate_stats: res and one of out / stats are required"); return SVDQ_E_INVALID; }
    if ((a->b || a->gate || a->out) && !a->a) { set_error("svdq_residual_gate_stats: b, gate and out need a"); return SVDQ_E_INVALID; }
    if (a->M <= 0 || a->C <= 0 || a->C % 8 || a->ld < a->C || a->ld % 8) {
        set_error("svdq_residual_gate_stats: need M=%d > 0, C=%d a multiple of 8, ld=%d >= C and a multiple of 8", a->M, a->C, a->ld);
        return SVDQ_E_INVALID;
    }
    if (((uintptr_t)a->res | (uintptr_t)a->a | (uintptr_t)a->b | (uintptr_t)a->gate | (uintptr_t)a->out) & 15 || ((uintptr_t)a->stats & 7)) {
        set_error("svdq_residual_gate_stats: tensors must be 16-byte aligned (stats 8-byte)");
        return SVDQ_E_INVALID;
    }
    if (a->dtype != SVDQ_BF16 && a->dtype != SVDQ_FP16) { set_error("svdq_residual_gate_stats: unknown dtype %d", a->dtype); return SVDQ_E_INVALID; }
    if (a->zero_ptr && (a->zero_bytes < 0 || a->zero_bytes % 16 || ((uintptr_t)a->zero_ptr & 15))) {
        set_error("svdq_residual_gate_stats: zero_ptr must be 16-byte aligned and zero_bytes a non-negative multiple of 16");
        return SVDQ_E_INVALID;
    }
    if (a->res2) {
        if (a->M2 <= 0 || (a->a != nullptr) != (a->a2 != nullptr) || (a->b != nullptr) != (a->b2 != nullptr) ||
            (a->gate != nullptr) != (a->gate2 != nullptr) || (a->out != nullptr) != (a->out2 != nullptr) ||
            (a->stats != nullptr) != (a->stats2 != nullptr)) {
            set_error("svdq_residual_gate_stats: the second problem must mirror the first (and M2 > 0)");
            return SVDQ_E_INVALID;
        }
        if (((uintptr_t)a->res2 | (uintptr_t)a->a2 | (uintptr_t)a->b2 | (uintptr_t)a->gate2 | (uintptr_t)a->out2) & 15 || ((uintptr_t)a->stats2 & 7)) {
            set_error("svdq_residual_gate_stats: second problem: tensors must be 16-byte aligned (stats 8-byte)");
            return SVDQ_E_INVALID;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    const int rc = a->dtype == SVDQ_BF16 ? launch_residual<SVDQ_BF16>(a, st) : launch_residual<SVDQ_FP16>(a, st);
    if (rc) { set_error("svdq_residual_gate_stats: C=%d: ceil(C/512) must be one of {1..8, 12, 16, 24, 32}", a->C); return SVDQ_E_UNSUPPORTED; }
    return hip_check(hipGetLastError(), "svdq_residual_gate_stats launch");
}
