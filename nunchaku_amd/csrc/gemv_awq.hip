// svdq_gemv_awq: AWQ W4A16 GEMV for the AdaLayerNormZero modulation projections.
// Reference: src/kernels/awq/gemv_awq.cu:100-286 (kernel + launcher), nunchaku/csrc/ops.h:123-145,
// nunchaku/models/linear.py:277-414 (AWQW4A16Linear), weight format text_encoders/tinychat_utils.py:76-107.
//
// HBM-bound: N*K/2 bytes of 4-bit codes are read once (28 MB for 3072 -> 18432), everything else is noise.
// The checkpoint layout is consumed as stored -- no load-time repack: a row group (4 output channels) is one
// contiguous run of 2*K bytes in which every 16-byte piece is 32 input channels of one channel, so a wave
// reads 1 KiB contiguous per instruction (8 chunks of 64 input channels x 4 channels x 2 halves) and one wave
// owns one row group.  4608 waves for N = 18432 keep ~4.7 MB in flight, enough to cover HBM latency without
// an explicit pipeline.  Arithmetic follows the reference's rounding points (DESIGN.md "AWQ GEMV"):
//   w16 = round16(fma(q, scale, scaled_zero));  p = round16(w16 * x);  y = round16(sum fp32 p) (+ bias, 16-bit add)
#include "svdq_common.h"

namespace svdq {

constexpr int AWQ_GROUP = 64;

template <int DT, int M>
__device__ __forceinline__ void gemv_awq_rowgroup(const uint16_t *__restrict__ x, const uint8_t *__restrict__ qw,
                                                  const uint16_t *__restrict__ scales, const uint16_t *__restrict__ zeros,
                                                  const uint16_t *__restrict__ bias, uint16_t *__restrict__ out, int K, int N,
                                                  int ldx, int ochunks, int rg) {
    using T = typename Half<DT>::T;
    const int lane = threadIdx.x & 63;
    if (rg * 4 >= N) return;
    const int row = (lane >> 1) & 3, half = lane & 1, cl = lane >> 3; // this lane's channel, 32-channel half, chunk in the wave-load
    const int n = rg * 4 + row;
    const int chunks = K / AWQ_GROUP;
    const uint8_t *wbase = qw + (size_t)rg * K * 2; // a row group holds 4*K nibbles

    float acc[M];
#pragma unroll
    for (int m = 0; m < M; m++) acc[m] = 0.f;

    // batches of U wave-loads (U KiB of codes per wave) issued back to back before any arithmetic: a wave's
    // HBM round trips overlap each other instead of adding up (K = 3072 is two batches)
    constexpr int U = 4;
    for (int c0 = 0; c0 < chunks; c0 += 8 * U) {
        v4i w[U];
        float s[U], z[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int c = c0 + 8 * u + cl;
            const bool live = c < chunks;
            w[u] = live ? __builtin_nontemporal_load(reinterpret_cast<const v4i *>(wbase + (size_t)(c0 + 8 * u) * 128 + lane * 16)) : v4i{0, 0, 0, 0};
            s[u] = live ? h2f(hfrom<T>(scales[(size_t)c * N + n])) : 0.f;
            z[u] = live ? h2f(hfrom<T>(zeros[(size_t)c * N + n])) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int c = c0 + 8 * u + cl;
            if (c0 + 8 * u >= chunks) break; // wave-uniform
            const int k0 = (c < chunks ? c : 0) * AWQ_GROUP + half * 32; // dead lanes read chunk 0 and add w = 0
            // 8 int16 = 32 channels: int16 j, nibble e <-> channel 8*e + j of this half (tinychat_utils.py:97-105)
            float wd[32]; // dequantised weights of channels k0 .. k0 + 31, already rounded to 16 bits
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const unsigned word = (unsigned)w[u][i];
                    const float qlo = (float)((word >> (4 * e)) & 15u), qhi = (float)((word >> (16 + 4 * e)) & 15u);
                    wd[8 * e + 2 * i] = round16<T>(__builtin_fmaf(qlo, s[u], z[u]));
                    wd[8 * e + 2 * i + 1] = round16<T>(__builtin_fmaf(qhi, s[u], z[u]));
                }
#pragma unroll
            for (int m = 0; m < M; m++) {
                const uint16_t *xp = x + (size_t)m * ldx + k0;
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const u16x8 xv = *reinterpret_cast<const u16x8 *>(xp + 8 * v);
#pragma unroll
                    for (int t = 0; t < 8; t++) acc[m] += round16<T>(wd[8 * v + t] * h2f(hfrom<T>(xv[t])));
                }
            }
        }
    }
    // lanes with the same channel: lane bit 0 (half) and bits 3..5 (chunk)
#pragma unroll
    for (int m = 0; m < M; m++) {
        float a = acc[m];
        a += __shfl_xor(a, 1);
        a += __shfl_xor(a, 8);
        a += __shfl_xor(a, 16);
        a += __shfl_xor(a, 32);
        if (half == 0 && cl == 0) {
            float y = round16<T>(a);
            if (bias) y = round16<T>(y + h2f(hfrom<T>(bias[n]))); // AWQW4A16Linear.forward: output.add_(bias), 16-bit
            const int no = ochunks > 1 ? (n % ochunks) * (N / ochunks) + n / ochunks : n; // de-interleave the modulation vectors
            out[(size_t)m * N + no] = hbits(f2h<T>(y));
        }
    }
}

template <int DT, int M>
__global__ __launch_bounds__(256) void gemv_awq_kernel(const uint16_t *__restrict__ x, const uint8_t *__restrict__ qw,
                                                        const uint16_t *__restrict__ scales, const uint16_t *__restrict__ zeros,
                                                        const uint16_t *__restrict__ bias, uint16_t *__restrict__ out, int K, int N,
                                                        int ldx, int ochunks) {
    gemv_awq_rowgroup<DT, M>(x, qw, scales, zeros, bias, out, K, N, ldx, ochunks, blockIdx.x * 4 + (threadIdx.x >> 6));
}

// batched form: the descriptors travel in the kernel arguments (80 x 48 B), a block finds its entry by a scalar scan
struct GemvEntry { const uint8_t *qw; const uint16_t *scales, *zeros, *bias; uint16_t *out; int N, ochunks; };
struct GemvBatch { GemvEntry e[SVDQ_GEMV_BATCH_MAX]; int count; }; // 80 x 48 B + 4: under the 4 KiB kernel-argument limit

template <int DT>
__global__ __launch_bounds__(256) void gemv_awq_batched_kernel(const uint16_t *__restrict__ x, const GemvBatch b, int K, int ldx) {
    int i = 0, first = 0; // block-uniform scan: entry i owns blocks [first, first + ceil(N_i / 16))
    while (i + 1 < b.count && (int)blockIdx.x >= first + (b.e[i].N / 4 + 3) / 4) { first += (b.e[i].N / 4 + 3) / 4; i++; }
    const GemvEntry &e = b.e[i];
    gemv_awq_rowgroup<DT, 1>(x, e.qw, e.scales, e.zeros, e.bias, e.out, K, e.N, ldx, e.ochunks,
                             ((int)blockIdx.x - first) * 4 + (threadIdx.x >> 6));
}

template <int DT> static int launch_gemv(const svdq_gemv_awq_args *a, hipStream_t st) {
    dim3 grid((a->N / 4 + 3) / 4), block(256);
#define SVDQ_GEMV_CASE(MM)                                                                                                          \
    case MM:                                                                                                                        \
        hipLaunchKernelGGL((gemv_awq_kernel<DT, MM>), grid, block, 0, st, (const uint16_t *)a->x, (const uint8_t *)a->qweight,        \
                           (const uint16_t *)a->scales, (const uint16_t *)a->zeros, (const uint16_t *)a->bias, (uint16_t *)a->out,  \
                           a->K, a->N, a->ldx, a->out_chunks);                                                                                   \
        break;
    switch (a->M) {
        SVDQ_GEMV_CASE(1) SVDQ_GEMV_CASE(2) SVDQ_GEMV_CASE(3) SVDQ_GEMV_CASE(4)
        SVDQ_GEMV_CASE(5) SVDQ_GEMV_CASE(6) SVDQ_GEMV_CASE(7) SVDQ_GEMV_CASE(8)
    default: return -1;
    }
#undef SVDQ_GEMV_CASE
    return 0;
}

} // namespace svdq

using namespace svdq;

static int validate_gemv(const svdq_gemv_awq_args *a) {
    if (!a->x || !a->qweight || !a->scales || !a->zeros || !a->out) { set_error("svdq_gemv_awq: x, qweight, scales, zeros and out are required"); return SVDQ_E_INVALID; }
    if (a->M < 1 || a->M > 8) { set_error("svdq_gemv_awq: M=%d must be in [1, 8] (gemv_awq.cu:280)", a->M); return SVDQ_E_INVALID; }
    if (a->group_size != AWQ_GROUP) { set_error("svdq_gemv_awq: group_size=%d (only 64 is implemented, gemv_awq.cu:281)", a->group_size); return SVDQ_E_UNSUPPORTED; }
    if (a->N <= 0 || a->N % 4 || a->K <= 0 || a->K % AWQ_GROUP) { set_error("svdq_gemv_awq: need N=%d %% 4 == 0 and K=%d %% 64 == 0", a->N, a->K); return SVDQ_E_INVALID; }
    if (a->ldx < a->K || a->ldx % 8) { set_error("svdq_gemv_awq: ldx=%d must be >= K and a multiple of 8", a->ldx); return SVDQ_E_INVALID; }
    if (((uintptr_t)a->x | (uintptr_t)a->qweight) & 15) { set_error("svdq_gemv_awq: x and qweight must be 16-byte aligned"); return SVDQ_E_INVALID; }
    if (a->dtype != SVDQ_BF16 && a->dtype != SVDQ_FP16) { set_error("svdq_gemv_awq: unknown dtype %d", a->dtype); return SVDQ_E_INVALID; }
    if (a->out_chunks < 0 || (a->out_chunks > 1 && a->N % a->out_chunks)) { set_error("svdq_gemv_awq: out_chunks=%d must divide N=%d", a->out_chunks, a->N); return SVDQ_E_INVALID; }
    return SVDQ_OK;
}

extern "C" int svdq_gemv_awq(const svdq_gemv_awq_args *a, void *stream) {
    if (!a) { set_error("svdq_gemv_awq: args is NULL"); return SVDQ_E_INVALID; }
    if (int rc = validate_gemv(a)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int prof = prof_begin(3, (double)a->N * a->K / 2 + 4.0 * (a->K / AWQ_GROUP) * a->N, st);
    if (a->dtype == SVDQ_BF16) launch_gemv<SVDQ_BF16>(a, st);
    else launch_gemv<SVDQ_FP16>(a, st);
    prof_end(prof, st);
    return hip_check(hipGetLastError(), "svdq_gemv_awq launch");
}

extern "C" int svdq_gemv_awq_batched(const svdq_gemv_awq_args *a, int32_t count, void *stream) {
    if (!a || count < 1 || count > SVDQ_GEMV_BATCH_MAX) { set_error("svdq_gemv_awq_batched: need 1 <= count=%d <= %d entries", count, SVDQ_GEMV_BATCH_MAX); return SVDQ_E_INVALID; }
    GemvBatch b;
    int blocks = 0;
    double bytes = 0;
    for (int i = 0; i < count; i++) {
        if (int rc = validate_gemv(a + i)) return rc;
        if (a[i].x != a[0].x || a[i].M != 1 || a[i].K != a[0].K || a[i].ldx != a[0].ldx || a[i].dtype != a[0].dtype) {
            set_error("svdq_gemv_awq_batched: entry %d must share x, M = 1, K, ldx and dtype with entry 0", i);
            return SVDQ_E_INVALID;
        }
        b.e[i] = GemvEntry{(const uint8_t *)a[i].qweight, (const uint16_t *)a[i].scales, (const uint16_t *)a[i].zeros,
                           (const uint16_t *)a[i].bias, (uint16_t *)a[i].out, a[i].N, a[i].out_chunks};
        blocks += (a[i].N / 4 + 3) / 4;
        bytes += (double)a[i].N * a[i].K / 2 + 4.0 * (a[i].K / AWQ_GROUP) * a[i].N;
    }
    b.count = count;
    hipStream_t st = (hipStream_t)stream;
    const int prof = prof_begin(3, bytes, st);
    if (a[0].dtype == SVDQ_BF16) hipLaunchKernelGGL((gemv_awq_batched_kernel<SVDQ_BF16>), dim3(blocks), dim3(256), 0, st, (const uint16_t *)a[0].x, b, a[0].K, a[0].ldx);
    else hipLaunchKernelGGL((gemv_awq_batched_kernel<SVDQ_FP16>), dim3(blocks), dim3(256), 0, st, (const uint16_t *)a[0].x, b, a[0].K, a[0].ldx);
    prof_end(prof, st);
    return hip_check(hipGetLastError(), "svdq_gemv_awq_batched launch");
}
