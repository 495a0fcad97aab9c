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

typedef float gv2f __attribute__((ext_vector_type(2)));
typedef __bf16 gbf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 gf16x2 __attribute__((ext_vector_type(2)));
// acc + round16(a) + round16(b): one v_cvt_pk (the two roundings) and one v_dot2c_f32_* against (1, 1)
template <int DT> __device__ __forceinline__ float add2_rounded(float a, float b, float acc) {
    if constexpr (DT == SVDQ_BF16) {
        const gbf16x2 pk = __builtin_convertvector((gv2f){a, b}, gbf16x2);
        return __builtin_amdgcn_fdot2_f32_bf16(pk, __builtin_bit_cast(gbf16x2, 0x3f803f80u), acc, false);
    } else {
        const gf16x2 pk = __builtin_convertvector((gv2f){a, b}, gf16x2);
        return __builtin_amdgcn_fdot2(pk, __builtin_bit_cast(gf16x2, 0x3c003c00u), acc, false);
    }
}

constexpr int AWQ_XLDS_MAX_K = 8192; // M = 1: the activation vector lives in LDS as fp32 (32 KiB at most)

// M = 1 (the modulation projections of a denoise step: 1.6 GB of codes, one activation vector).  At 1.7-1.9 TB/s the kernel
// is nowhere near HBM: it pays ~10 VALU instructions per weight for the reference's two 16-bit roundings per product, and a
// wave's life is a chain of dependent round trips.  This path trims both:
//   * the wave's first batch of codes (6 KiB: ALL of a K = 3072 row group) and its scales / zeros are requested BEFORE the
//     workgroup stages x in LDS, so the two round trips overlap instead of following each other;
//   * x is converted ONCE per workgroup (bf16: fp32 in LDS, broadcast ds_read_b128 instead of an unpack per weight and lane;
//     fp16: as stored), the accumulation takes two rounded products per v_dot2c against (1, 1);
//   * bf16: nibbles become floats through v_cvt_f32_ubyteN on two masked copies of the word, the two roundings go through
//     v_cvt_pk_bf16_f32 (~6 instructions per weight);
//   * fp16: the packed 16-bit pipe.  A dword of the checkpoint holds int16 2i | int16 2i+1 and nibble e of int16 j is channel
//     8e + j, so (word >> 4e) & 0x000f000f is the code PAIR of channels (8e + 2i, 8e + 2i + 1) -- adjacent channels, i.e. one
//     dword of x as stored.  1024 + q (0x6400 | q) is exact in fp16 and so is the step back to q (e even: - 1024; e odd, where
//     the nibble sits four bits up: * 1/16 - 64); then v_pk_fma_f16 (q, scale, zero) IS the reference's __hfma2 and
//     v_pk_mul_f16 its __hmul2: 5 instructions per two weights.
// Same rounding points either way; the fp32 sum is formed in a different order (as between any two launches of the reference).
constexpr int AWQ_X1_U = 6; // wave loads (1 KiB each) per batch

// one wave load (u) of a batch against x: bf16 through fp32
template <int DT>
__device__ __forceinline__ void gemv_x1_chunk(const float *xs /* LDS */, int k0, const v4i w, unsigned sbits, unsigned zbits, unsigned,
                                              float &acc, float &acc2) {
    using T = typename Half<DT>::T;
    typedef __attribute__((address_space(3))) v4f lds_v4f;
    const __attribute__((address_space(3))) float *xk = (const __attribute__((address_space(3))) float *)xs + k0;
    const float s = h2f(hfrom<T>((uint16_t)sbits)), z = h2f(hfrom<T>((uint16_t)zbits));
    // 8 int16 = 32 channels: int16 j, nibble e <-> channel 8*e + j of this half (tinychat_utils.py:97-105); dword i holds
    // int16 2i (low half) and 2i+1: byte b of (word & 0x0f0f0f0f) is nibble e = 2*(b&1) of int16 2i + (b>>1), byte b of
    // ((word >> 4) & 0x0f0f0f0f) nibble e = 2*(b&1) + 1
    float xv[32];
#pragma unroll
    for (int v = 0; v < 8; v++) {
        const v4f t = *(const lds_v4f *)(xk + 4 * v);
        xv[4 * v] = t[0]; xv[4 * v + 1] = t[1]; xv[4 * v + 2] = t[2]; xv[4 * v + 3] = t[3];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const unsigned word = (unsigned)w[i];
        const unsigned ev = word & 0x0f0f0f0fu, od = (word >> 4) & 0x0f0f0f0fu;
        float p[8];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int j = 2 * i + (b >> 1);
            const float qe = (float)((ev >> (8 * b)) & 0xffu), qo = (float)((od >> (8 * b)) & 0xffu);
            const float we = round16<T>(__builtin_fmaf(qe, s, z)), wo = round16<T>(__builtin_fmaf(qo, s, z));
            p[2 * b] = we * xv[8 * (2 * (b & 1)) + j];
            p[2 * b + 1] = wo * xv[8 * (2 * (b & 1) + 1) + j];
        }
        acc = add2_rounded<DT>(p[0], p[1], acc);
        acc2 = add2_rounded<DT>(p[2], p[3], acc2);
        acc = add2_rounded<DT>(p[4], p[5], acc);
        acc2 = add2_rounded<DT>(p[6], p[7], acc2);
    }
}

// ... fp16 on the packed pipe (x as stored: 16-bit in LDS)
template <>
__device__ __forceinline__ void gemv_x1_chunk<SVDQ_FP16>(const float *xs /* LDS */, int k0, const v4i w, unsigned sbits, unsigned zbits,
                                                          unsigned magic /* 0x64006400 in a VGPR */, float &acc, float &acc2) {
    typedef __attribute__((address_space(3))) v4i lds_v4i;
    const __attribute__((address_space(3))) uint16_t *xk = (const __attribute__((address_space(3))) uint16_t *)xs + k0;
    const gf16x2 ones = __builtin_bit_cast(gf16x2, 0x3c003c00u);
    const gf16x2 m1024 = __builtin_bit_cast(gf16x2, 0xe400e400u), sixteenth = __builtin_bit_cast(gf16x2, 0x2c002c00u),
                 m64 = __builtin_bit_cast(gf16x2, 0xd400d400u);
    const gf16x2 s2 = __builtin_bit_cast(gf16x2, sbits * 0x10001u), z2 = __builtin_bit_cast(gf16x2, zbits * 0x10001u);
    unsigned xv[16]; // dword r = channels (2r, 2r + 1) of this half
#pragma unroll
    for (int v = 0; v < 4; v++) {
        const v4i t = *(const lds_v4i *)(xk + 8 * v);
        xv[4 * v] = (unsigned)t[0]; xv[4 * v + 1] = (unsigned)t[1]; xv[4 * v + 2] = (unsigned)t[2]; xv[4 * v + 3] = (unsigned)t[3];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const unsigned word = (unsigned)w[i], up = word >> 8;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const unsigned src = e < 2 ? word : up;
            const gf16x2 t = __builtin_bit_cast(gf16x2, (src & ((e & 1) ? 0x00f000f0u : 0x000f000fu)) | magic);
            const gf16x2 q = (e & 1) ? __builtin_elementwise_fma(t, sixteenth, m64) : t + m1024;
            const gf16x2 wq = __builtin_elementwise_fma(q, s2, z2);
            const gf16x2 pr = wq * __builtin_bit_cast(gf16x2, xv[4 * e + i]);
            if (e & 1) acc2 = __builtin_amdgcn_fdot2(pr, ones, acc2, false);
            else acc = __builtin_amdgcn_fdot2(pr, ones, acc, false);
        }
    }
}

// x [K] 16-bit -> LDS (bf16: as fp32; fp16: as stored), cooperatively by the 256 threads of the workgroup; ends with a barrier
template <int DT> __device__ __forceinline__ void gemv_stage_x(float *xs /* LDS */, const uint16_t *__restrict__ x, int K) {
    using T = typename Half<DT>::T;
    for (int k = threadIdx.x * 4; k < K; k += 256 * 4) {
        const u16x4 v = *reinterpret_cast<const u16x4 *>(x + k);
        if constexpr (DT == SVDQ_FP16) *reinterpret_cast<u16x4 *>(reinterpret_cast<uint16_t *>(xs) + k) = v;
        else *reinterpret_cast<v4f *>(xs + k) = v4f{h2f(hfrom<T>(v[0])), h2f(hfrom<T>(v[1])), h2f(hfrom<T>(v[2])), h2f(hfrom<T>(v[3]))};
    }
    __syncthreads();
}

// the whole workgroup calls this (it contains the staging barrier); a wave whose row group lies beyond N only helps staging
template <int DT>
__device__ __forceinline__ void gemv_awq_x1(float *xs /* LDS */, const uint16_t *__restrict__ x, const uint8_t *__restrict__ qw,
                                            const uint16_t *__restrict__ scales, const uint16_t *__restrict__ zeros,
                                            const uint16_t *__restrict__ bias, uint16_t *__restrict__ out, int K, int N,
                                            int ochunks, int rg) {
    using T = typename Half<DT>::T;
    constexpr int U = AWQ_X1_U;
    const int lane = threadIdx.x & 63;
    const bool active = rg * 4 < N; // wave-uniform
    const int row = (lane >> 1) & 3, half = lane & 1, cl = lane >> 3;
    const int n = rg * 4 + row;
    const int chunks = K / AWQ_GROUP;
    const uint8_t *wbase = qw + (size_t)rg * K * 2;
    v4i w[U];
    unsigned s[U], z[U]; // raw 16-bit scale / scaled zero of the lane's (channel, chunk)
    auto request = [&](int c0) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int c = c0 + 8 * u + cl;
            const bool live = active && c < chunks;
            w[u] = live ? __builtin_nontemporal_load(reinterpret_cast<const v4i *>(wbase + (size_t)(c0 + 8 * u) * 128 + lane * 16)) : v4i{0, 0, 0, 0};
            s[u] = live ? (unsigned)scales[(size_t)c * N + n] : 0u;
            z[u] = live ? (unsigned)zeros[(size_t)c * N + n] : 0u;
        }
    };
    request(0);
    gemv_stage_x<DT>(xs, x, K);
    if (!active) return;
    // gfx9 VOP3 takes no literal and one scalar operand: the fp16 path's v_and_or_b32 needs one of its two constants in a VGPR
    unsigned magic = 0x64006400u;
    asm volatile("" : "+v"(magic));
    float acc = 0.f, acc2 = 0.f; // two chains
    for (int c0 = 0;;) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int c = c0 + 8 * u + cl;
            if (c0 + 8 * u >= chunks) break; // wave-uniform
            const int k0 = (c < chunks ? c : 0) * AWQ_GROUP + half * 32; // dead lanes read chunk 0 and add w = 0
            gemv_x1_chunk<DT>(xs, k0, w[u], s[u], z[u], magic, acc, acc2);
        }
        c0 += 8 * U;
        if (c0 >= chunks) break;
        request(c0);
    }
    float a = acc + acc2;
    a += __shfl_xor(a, 1);
    a += __shfl_xor(a, 8);
    a += __shfl_xor(a, 16);
    a += __shfl_xor(a, 32);
    if (half == 0 && cl == 0) {
        float y = round16<T>(a);
        if (bias) y = round16<T>(y + h2f(hfrom<T>(bias[n])));
        const int no = ochunks > 1 ? (n % ochunks) * (N / ochunks) + n / ochunks : n;
        out[no] = hbits(f2h<T>(y));
    }
}

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
    if constexpr (M == 1) {
        __shared__ __attribute__((aligned(16))) float xs[AWQ_XLDS_MAX_K];
        if (K <= AWQ_XLDS_MAX_K) { // block-uniform
            gemv_awq_x1<DT>(xs, x, qw, scales, zeros, bias, out, K, N, ochunks, blockIdx.x * 4 + (threadIdx.x >> 6));
            return;
        }
    }
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
    __shared__ __attribute__((aligned(16))) float xs[AWQ_XLDS_MAX_K];
    if (K <= AWQ_XLDS_MAX_K) { // block-uniform
        gemv_awq_x1<DT>(xs, x, e.qw, e.scales, e.zeros, e.bias, e.out, K, e.N, e.ochunks, ((int)blockIdx.x - first) * 4 + (threadIdx.x >> 6));
        return;
    }
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
