// svdq_gemm_w4a4: fused W4A4 GEMM + per-group dequant + bias + rank-R low-rank correction +
// activation / requantisation / RMSNorm+RoPE epilogues for gfx950 (MI355X).
//
// Replaces the reference's gemm_w4a4_kernel and its epilogue chain
// (gemm_w4a4.cuh:831-928,1046-1095; gemm_base.cuh:367-409,667-792; lora.cuh:110-353;
//  epilogues.cuh:22-44,269-425; dispatch gemm_w4a4_launch_impl.cuh:7-424).
//
// Kernel structure (DESIGN.md "GEMM kernel"):
//   * workgroup = 256 threads = 4 waves (2 x 2), output tile 128 x 128, wave tile 64 x 64 =
//     4 x 4 MFMA tiles of v_mfma_i32_16x16x64_i8; one k-iteration = one 64-channel quantisation
//     group (the int32 partial sums cannot cross groups: scales are per (row, group) x (col, group)).
//   * operands stay INT4 in HBM and in LDS.  Both are stored in the T16 tile order
//     (svdq_common.h), so a (128 rows x 64 k) operand block is one contiguous 4 KiB chunk: staged
//     with one coalesced 16-byte load per thread, read back with conflict-free ds_read_b64 (lane l
//     gets the 16 codes of row l&15, k-slot l>>4).
//   * int4 -> int8 in registers without sign extension: (w<<4)&0xF0F0F0F0 and w&0xF0F0F0F0 give
//     16 x nibble as signed bytes; the 1/256 (1/16 for unsigned activations) is folded into the
//     weight scale when it is staged.  Both operands use the same nibble->byte map, which is all a
//     dot product needs.
//   * the MFMA is issued "transposed" (weights as the A operand, activations as B) so that in the
//     C layout each lane owns ONE output row m and 4 consecutive columns n: 8-byte output stores,
//     RoPE pairs and requantisation bytes are lane-local.
//   * fp32 accumulators: acc += float(psum) * (as[m] * ws[n]).  (The reference accumulates in
//     16-bit, gemm_w4a4.cuh:1080; fp32 is strictly more accurate.)  Bias and the low-rank up
//     projection (a bf16/f16 MFMA issued straight onto the fp32 accumulators) are added before the
//     single rounding to 16-bit that precedes the activation epilogues.
#include "svdq_common.h"

namespace svdq {

constexpr int BM = 128, BN = 128;
constexpr int KG = 2;                                    // quantisation groups per pipeline stage
constexpr int STAGE_BYTES = KG * (4096 + 4096 + 512 + 512); // A, W, as, ws per group
constexpr int MAX_LORA_TILES = 16;                       // R <= 256

struct GemmParams {
    const uint8_t *act;
    const uint8_t *wgt;
    const void *ascales;
    const void *wscales;
    const void *bias;
    const float *lora_act_in;
    const void *lora_up;
    void *out;
    uint8_t *qout;
    void *oscales;
    const void *next_smooth;
    const void *next_lora_down;
    float *lora_act_out;
    const void *norm_q;
    const void *norm_k;
    const float *rotary_emb;
    int M, M_pad, N, K, R, R2, ldo;
    float lora_scales[MAX_LORA_TILES];
};

__device__ __forceinline__ v4i unpack_s4x16(v2i w) {
    // 16 signed nibbles -> 16 signed bytes holding 16*value
    v4i r;
    r[0] = (w[0] << 4) & 0xF0F0F0F0;
    r[1] = w[0] & 0xF0F0F0F0;
    r[2] = (w[1] << 4) & 0xF0F0F0F0;
    r[3] = w[1] & 0xF0F0F0F0;
    return r;
}
__device__ __forceinline__ v4i unpack_u4x16(v2i w) {
    // 16 unsigned nibbles -> 16 bytes holding the value (0..15)
    v4i r;
    r[0] = w[0] & 0x0F0F0F0F;
    r[1] = (w[0] >> 4) & 0x0F0F0F0F;
    r[2] = w[1] & 0x0F0F0F0F;
    r[3] = (w[1] >> 4) & 0x0F0F0F0F;
    return r;
}

__device__ __forceinline__ float gelu_tanh_f(float x) {
    // reference: gemm_utils.cuh:305-312 (tanh.approx there; exact tanhf here)
    float x3 = x * x * x;
    float t = 0.5f + 0.5f * tanhf(0.79788456f * (x + 0.044715f * x3));
    return x * t;
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

template <int DT, bool ACT_UNSIGNED, int FUSE>
__global__ __launch_bounds__(256, 2) void gemm_w4a4_kernel(const GemmParams p) {
    using T = typename Half<DT>::T;
    using V8 = typename Half<DT>::V8;
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * STAGE_BYTES];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lq = lane >> 4;
    const int G = p.K / GROUP;
    const int nbn = p.N / BN;
    const int bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
    const int m0 = bm * BM, n0 = bn * BN;

    // ---- operand streams ---------------------------------------------------------------------
    // one 16-byte load per thread covers a (128 rows x 64 k) block of A and of W; threads 0..127
    // additionally carry the activation scale of row tid, threads 128..255 the weight scale of
    // column tid-128 (no branch in the loop: pointer/stride/multiplier are selected once).
    const uint8_t *a_src = p.act + ((size_t)bm * G) * 4096 + tid * 16;
    const uint8_t *w_src = p.wgt + ((size_t)bn * G) * 4096 + tid * 16;
    const bool is_as = tid < 128;
    const T *s_src = is_as ? (const T *)p.ascales + m0 + tid : (const T *)p.wscales + n0 + (tid - 128);
    const size_t s_stride = is_as ? (size_t)p.M_pad : (size_t)p.N;
    const float s_mul = is_as ? 1.0f : (ACT_UNSIGNED ? (1.0f / 16.0f) : (1.0f / 256.0f));

    v4f acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

    T rs[KG];
    v4i zero = {0, 0, 0, 0};
    // A and W tiles go HBM -> LDS by DMA (global_load_lds, 16 B per lane, no VGPR round trip: a
    // register-staged prefetch is spilled by hipcc and drains vmcnt right behind the loads).  The
    // LDS image is lane-linear (wave base + lane*16), which is exactly the T16 order.
    typedef __attribute__((address_space(3))) void lds_void;
    auto fetch = [&](int g0, uint8_t *st) {
#pragma unroll
        for (int u = 0; u < KG; u++) {
            uint8_t *sg = st + u * (STAGE_BYTES / KG) + wave * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a_src + (size_t)(g0 + u) * 4096),
                                             (lds_void *)sg, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(w_src + (size_t)(g0 + u) * 4096),
                                             (lds_void *)(sg + 4096), 16, 0, 0);
            rs[u] = s_src[(size_t)(g0 + u) * s_stride];
        }
    };
    auto commit = [&](uint8_t *st) {
#pragma unroll
        for (int u = 0; u < KG; u++)
            reinterpret_cast<float *>(st + u * (STAGE_BYTES / KG) + 8192)[tid] = h2f(rs[u]) * s_mul;
    };

    const int NS = G / KG; // K % 128 == 0  ->  G even
    fetch(0, lds);
    commit(lds);
    __syncthreads();

    for (int s = 0; s < NS; s++) {
        const uint8_t *st = lds + (s & 1) * STAGE_BYTES;
        if (s + 1 < NS) fetch((s + 1) * KG, lds + ((s + 1) & 1) * STAGE_BYTES);

#pragma unroll
        for (int u = 0; u < KG; u++) {
            const uint8_t *sg = st + u * (STAGE_BYTES / KG);
            v4i wf[4], af[4];
            v4f wsv[4];
            float asv[4];
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                v2i raw = *reinterpret_cast<const v2i *>(sg + 4096 + ((wn * 4 + nt) * 64 + lane) * 8);
                wf[nt] = unpack_s4x16(raw);
                wsv[nt] = *reinterpret_cast<const v4f *>(sg + 8192 + 512 + (wn * 64 + nt * 16 + lq * 4) * 4);
            }
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                v2i raw = *reinterpret_cast<const v2i *>(sg + ((wm * 4 + mt) * 64 + lane) * 8);
                af[mt] = ACT_UNSIGNED ? unpack_u4x16(raw) : unpack_s4x16(raw);
                asv[mt] = *reinterpret_cast<const float *>(sg + 8192 + (wm * 64 + mt * 16 + lr) * 4);
            }
            // software pipeline over the 16 MFMA tiles of this group: the matrix pipe works on
            // tile t+DEPTH while the VALU dequantises tile t (psum ring of DEPTH+1 tiles in VGPRs;
            // without it hipcc parks all 16 psum tiles in AGPRs and serialises the two phases).
            constexpr int DEPTH = 3;
            v4i ps[DEPTH + 1];
            // The zero C operand is laundered through an empty asm before every MFMA and the freshly
            // updated accumulators after every dequant step: volatile asms keep their program order,
            // which pins "MFMA(t+DEPTH); dequant(t)" -- otherwise LLVM hoists all 16 MFMAs to the top
            // of the group and sinks the 192 dequant ops to the bottom (no overlap, 128 extra VGPRs).
#pragma unroll
            for (int t = 0; t < DEPTH; t++) {
                asm volatile("" : "+v"(zero));
                ps[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[t & 3], af[t >> 2], zero, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 16; t++) {
                if (t + DEPTH < 16) {
                    asm volatile("" : "+v"(zero));
                    ps[(t + DEPTH) % (DEPTH + 1)] = __builtin_amdgcn_mfma_i32_16x16x64_i8(
                        wf[(t + DEPTH) & 3], af[(t + DEPTH) >> 2], zero, 0, 0, 0);
                }
                const int mt = t >> 2, nt = t & 3;
                const v4i q = ps[t % (DEPTH + 1)];
#pragma unroll
                for (int r = 0; r < 4; r++)
                    acc[mt][nt][r] = __builtin_fmaf((float)q[r], asv[mt] * wsv[nt][r], acc[mt][nt][r]);
                asm volatile("" : "+v"(acc[mt][nt]));
            }
        }

        if (s + 1 < NS) commit(lds + ((s + 1) & 1) * STAGE_BYTES);
        __syncthreads();
    }

    // ------------------------------------------------------------------ epilogue
    const int nw0 = n0 + wn * 64; // first column of this wave
    const int mw0 = m0 + wm * 64; // first row of this wave

    // bias (reference EpilogueBias, gemm_base.cuh:710-781)
    if (p.bias) {
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
            u16x4 b = *reinterpret_cast<const u16x4 *>((const T *)p.bias + nw0 + nt * 16 + lq * 4);
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[mt][nt][r] += h2f(hfrom<T>(b[r]));
        }
    }

    // low-rank up projection (reference EpilogueLoraUp, lora.cuh:110-241): fp32 activations are
    // scaled per 16 ranks and rounded to 16-bit (:145-158), multiplied on the matrix cores and
    // accumulated in fp32 -- here straight onto the GEMM accumulators.
    if (p.R > 0) {
        for (int rc = 0; rc < p.R; rc += 32) {
            const int r0 = rc + lq * 8;
            const bool live = r0 < p.R;
            V8 la[4], lu[4];
            const float sc = live ? p.lora_scales[r0 >> 4] : 0.f;
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                if (live) {
                    const float *src = p.lora_act_in + (size_t)(mw0 + mt * 16 + lr) * p.R + r0;
                    v4f x0 = *reinterpret_cast<const v4f *>(src);
                    v4f x1 = *reinterpret_cast<const v4f *>(src + 4);
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        la[mt][j] = f2h<T>(x0[j] * sc);
                        la[mt][4 + j] = f2h<T>(x1[j] * sc);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; j++) la[mt][j] = (T)0.f;
                }
            }
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                if (live) {
                    lu[nt] = *reinterpret_cast<const V8 *>((const T *)p.lora_up + (size_t)(nw0 + nt * 16 + lr) * p.R + r0);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; j++) lu[nt][j] = (T)0.f;
                }
            }
#pragma unroll
            for (int mt = 0; mt < 4; mt++)
#pragma unroll
                for (int nt = 0; nt < 4; nt++) acc[mt][nt] = Half<DT>::mfma(lu[nt], la[mt], acc[mt][nt]);
        }
    }

    // the single rounding to the 16-bit model dtype (the reference's tile is 16-bit from here on)
#pragma unroll
    for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[mt][nt][r] = round16<T>(acc[mt][nt][r]);

    if constexpr (FUSE == SVDQ_FUSE_SILU) {
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int nt = 0; nt < 4; nt++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[mt][nt][r] = round16<T>(silu_f(acc[mt][nt][r]));
    }

    if constexpr (FUSE == SVDQ_FUSE_RMSNORM_ROPE) {
        // reference EpilogueRMSNormRope (epilogues.cuh:269-425).  BN == 128 == one head.
        const int third = p.N / 3;
        const bool is_q = n0 < third;
        const bool is_k = !is_q && n0 < 2 * third;
        if (is_q || is_k) { // block-uniform
            float *sq = reinterpret_cast<float *>(lds); // [2][128]
            float part[4];
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                float s = 0.f;
#pragma unroll
                for (int nt = 0; nt < 4; nt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) s += acc[mt][nt][r] * acc[mt][nt][r];
                s += __shfl_xor(s, 16);
                s += __shfl_xor(s, 32);
                part[mt] = s;
                if (lq == 0) sq[wn * 128 + wm * 64 + mt * 16 + lr] = s;
            }
            __syncthreads();
            const T *nw = (const T *)(is_q ? p.norm_q : p.norm_k);
#pragma unroll
            for (int mt = 0; mt < 4; mt++) {
                const int row = wm * 64 + mt * 16 + lr;
                const float tot = sq[row] + sq[128 + row];
                const float coef = 1.0f / sqrtf(tot / 128.0f + 1e-6f);
                const int m_abs = m0 + row;
                // packed rotary order (models/embeddings.py:100-138): [m/16][d/8][r8*4+p][rh][sin,cos]
                const size_t rbase = ((size_t)(m_abs >> 4) * 16) * 128 + (size_t)((m_abs & 7) * 4) * 4 + ((m_abs >> 3) & 1) * 2;
#pragma unroll
                for (int nt = 0; nt < 4; nt++) {
                    const int c = wn * 64 + nt * 16 + lq * 4; // column inside the head
                    u16x4 wv = *reinterpret_cast<const u16x4 *>(nw + c);
                    const int pi = c >> 1; // first pair index
                    const float *rp = p.rotary_emb + rbase + (size_t)(pi >> 2) * 128 + (size_t)(pi & 3) * 4;
                    float2 sc0 = *reinterpret_cast<const float2 *>(rp);
                    float2 sc1 = *reinterpret_cast<const float2 *>(rp + 4);
                    float v0 = acc[mt][nt][0] * (coef * h2f(hfrom<T>(wv[0])));
                    float v1 = acc[mt][nt][1] * (coef * h2f(hfrom<T>(wv[1])));
                    float v2 = acc[mt][nt][2] * (coef * h2f(hfrom<T>(wv[2])));
                    float v3 = acc[mt][nt][3] * (coef * h2f(hfrom<T>(wv[3])));
                    acc[mt][nt][0] = round16<T>(v0 * sc0.y - v1 * sc0.x);
                    acc[mt][nt][1] = round16<T>(v0 * sc0.x + v1 * sc0.y);
                    acc[mt][nt][2] = round16<T>(v2 * sc1.y - v3 * sc1.x);
                    acc[mt][nt][3] = round16<T>(v2 * sc1.x + v3 * sc1.y);
                }
            }
        }
    }

    if constexpr (FUSE == SVDQ_FUSE_GELU_QUANT) {
        // EpilogueGelu (epilogues.cuh:22-44) -> 16-bit
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int nt = 0; nt < 4; nt++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[mt][nt][r] = round16<T>(gelu_tanh_f(acc[mt][nt][r]));

        // EpilogueLoraDown for the NEXT layer on the GELU output, before shift/smooth
        // (lora.cuh:243-353, launch_impl.cuh:226-262).  In the C layout a lane already holds, for its
        // row m, columns {nt*16 + lq*4 + r}: two n-tiles form the 8-element k-slot of a 16x16x32
        // MFMA operand with no data movement; the weight operand is loaded in the matching order.
        if (p.R2 > 0) {
            const T *ld = (const T *)p.next_lora_down; // rank-major [R2][N]
            for (int rt = 0; rt < p.R2 / 16; rt++) {
                v4f d[4];
#pragma unroll
                for (int mt = 0; mt < 4; mt++) d[mt] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int np = 0; np < 2; np++) {
                    const T *src = ld + (size_t)(rt * 16 + lr) * p.N + nw0 + np * 32 + lq * 4;
                    u16x4 w0 = *reinterpret_cast<const u16x4 *>(src);
                    u16x4 w1 = *reinterpret_cast<const u16x4 *>(src + 16);
                    V8 wv;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        wv[j] = hfrom<T>(w0[j]);
                        wv[4 + j] = hfrom<T>(w1[j]);
                    }
#pragma unroll
                    for (int mt = 0; mt < 4; mt++) {
                        V8 gv;
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            gv[j] = f2h<T>(acc[mt][np * 2][j]);
                            gv[4 + j] = f2h<T>(acc[mt][np * 2 + 1][j]);
                        }
                        d[mt] = Half<DT>::mfma(wv, gv, d[mt]);
                    }
                }
#pragma unroll
                for (int mt = 0; mt < 4; mt++) {
                    float *dst = p.lora_act_out + (size_t)(mw0 + mt * 16 + lr) * p.R2 + rt * 16 + lq * 4;
#pragma unroll
                    for (int i = 0; i < 4; i++) unsafeAtomicAdd(dst + i, d[mt][i]);
                }
            }
        }

        // EpilogueQuantize<false, unsigned> (gemm_w4a4.cuh:930-1043): 16-bit add of the shift,
        // fp32 divide by the next layer's smooth factor -> 16-bit, per (row, 64 columns) absmax,
        // scale = amax/15, unsigned 4-bit codes in T16 order for the next GEMM.
        const int G2 = p.N / GROUP;
        const int g2 = nw0 / GROUP;
        u16x4 sm[4];
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
            sm[nt] = *reinterpret_cast<const u16x4 *>((const T *)p.next_smooth + nw0 + nt * 16 + lq * 4);
#pragma unroll
        for (int mt = 0; mt < 4; mt++) {
            float xh[4][4];
            float amax = 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; nt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float sh = round16<T>(acc[mt][nt][r] + 0.171875f);
                    float v = round16<T>(sh / h2f(hfrom<T>(sm[nt][r])));
                    xh[nt][r] = v;
                    amax = fmaxf(amax, fabsf(v));
                }
            amax = fmaxf(amax, __shfl_xor(amax, 16));
            amax = fmaxf(amax, __shfl_xor(amax, 32));
            const float scale = amax * (1.0f / 15.0f);
            const float rscale = scale == 0.f ? 0.f : 1.0f / scale;
            const int m_abs = mw0 + mt * 16 + lr;
            uint8_t *qrow = p.qout + ((((size_t)(m_abs >> 7) * G2 + g2) * 8 + ((m_abs & 127) >> 4)) * 64 + lr) * 8 + lq * 2;
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                int q0 = (int)fminf(fmaxf(rintf(xh[nt][0] * rscale), 0.f), 15.f);
                int q1 = (int)fminf(fmaxf(rintf(xh[nt][1] * rscale), 0.f), 15.f);
                int q2 = (int)fminf(fmaxf(rintf(xh[nt][2] * rscale), 0.f), 15.f);
                int q3 = (int)fminf(fmaxf(rintf(xh[nt][3] * rscale), 0.f), 15.f);
                unsigned short v = (unsigned short)(q0 | (q1 << 4) | (q2 << 8) | (q3 << 12));
                *reinterpret_cast<unsigned short *>(qrow + (size_t)nt * 16 * 8) = v;
            }
            if (lq == 0) ((T *)p.oscales)[(size_t)g2 * p.M_pad + m_abs] = f2h<T>(scale);
        }
        return;
    }

    // EpilogueDefault (gemm_base.cuh:667-698): store rows < M; fp16 clamps to +-65504
#pragma unroll
    for (int mt = 0; mt < 4; mt++) {
        const int m_abs = mw0 + mt * 16 + lr;
        if (m_abs < p.M) {
            T *orow = (T *)p.out + (size_t)m_abs * p.ldo + nw0 + lq * 4;
#pragma unroll
            for (int nt = 0; nt < 4; nt++) {
                u16x4 o;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float v = acc[mt][nt][r];
                    if constexpr (DT == SVDQ_FP16) v = fminf(fmaxf(v, -65504.f), 65504.f);
                    o[r] = hbits(f2h<T>(v));
                }
                *reinterpret_cast<u16x4 *>(orow + nt * 16) = o;
            }
        }
    }
}

template <int DT, bool UNS, int FUSE>
static void launch_one(const GemmParams &p, hipStream_t st) {
    dim3 grid((p.M_pad / BM) * (p.N / BN)), block(256);
    hipLaunchKernelGGL((gemm_w4a4_kernel<DT, UNS, FUSE>), grid, block, 0, st, p);
}

template <int DT, bool UNS>
static void launch_fuse(const GemmParams &p, int fuse, hipStream_t st) {
    switch (fuse) {
    case SVDQ_FUSE_NONE: launch_one<DT, UNS, SVDQ_FUSE_NONE>(p, st); break;
    case SVDQ_FUSE_SILU: launch_one<DT, UNS, SVDQ_FUSE_SILU>(p, st); break;
    case SVDQ_FUSE_GELU_QUANT: launch_one<DT, UNS, SVDQ_FUSE_GELU_QUANT>(p, st); break;
    case SVDQ_FUSE_RMSNORM_ROPE: launch_one<DT, UNS, SVDQ_FUSE_RMSNORM_ROPE>(p, st); break;
    }
}

} // namespace svdq

using namespace svdq;

extern "C" int svdq_gemm_w4a4(const svdq_gemm_args *a, void *stream) {
    if (!a) { set_error("svdq_gemm_w4a4: args is NULL"); return SVDQ_E_INVALID; }
    if (!a->act || !a->wgt || !a->ascales || !a->wscales) {
        set_error("svdq_gemm_w4a4: act, wgt, ascales and wscales are required");
        return SVDQ_E_INVALID;
    }
    if (a->M <= 0 || a->M_pad < a->M || a->M_pad % 256) {
        set_error("svdq_gemm_w4a4: need 0 < M=%d <= M_pad=%d and M_pad %% 256 == 0", a->M, a->M_pad);
        return SVDQ_E_INVALID;
    }
    if (a->N <= 0 || a->N % 128 || a->K <= 0 || a->K % 128) {
        set_error("svdq_gemm_w4a4: N=%d and K=%d must be positive multiples of 128", a->N, a->K);
        return SVDQ_E_INVALID;
    }
    if (a->R < 0 || a->R % 16 || a->R > 16 * MAX_LORA_TILES || a->R2 < 0 || a->R2 % 16 || a->R2 > 16 * MAX_LORA_TILES) {
        set_error("svdq_gemm_w4a4: R=%d / R2=%d must be multiples of 16 in [0, %d]", a->R, a->R2, 16 * MAX_LORA_TILES);
        return SVDQ_E_INVALID;
    }
    if (a->R > 0 && (!a->lora_act_in || !a->lora_up)) { set_error("svdq_gemm_w4a4: R > 0 needs lora_act_in and lora_up"); return SVDQ_E_INVALID; }
    if (a->dtype != SVDQ_BF16 && a->dtype != SVDQ_FP16) { set_error("svdq_gemm_w4a4: unknown dtype %d", a->dtype); return SVDQ_E_INVALID; }
    switch (a->fuse) {
    case SVDQ_FUSE_NONE:
    case SVDQ_FUSE_SILU:
        if (!a->out) { set_error("svdq_gemm_w4a4: out is required"); return SVDQ_E_INVALID; }
        break;
    case SVDQ_FUSE_GELU_QUANT:
        if (!a->qout || !a->oscales || !a->next_smooth) { set_error("svdq_gemm_w4a4: GELU_QUANT needs qout, oscales and next_smooth"); return SVDQ_E_INVALID; }
        if (a->R2 > 0 && (!a->next_lora_down || !a->lora_act_out)) { set_error("svdq_gemm_w4a4: R2 > 0 needs next_lora_down and lora_act_out"); return SVDQ_E_INVALID; }
        break;
    case SVDQ_FUSE_RMSNORM_ROPE:
        if (!a->out || !a->norm_q || !a->norm_k || !a->rotary_emb) { set_error("svdq_gemm_w4a4: RMSNORM_ROPE needs out, norm_q, norm_k and rotary_emb"); return SVDQ_E_INVALID; }
        if (a->N % 384) { set_error("svdq_gemm_w4a4: RMSNORM_ROPE needs N=%d to be a multiple of 3*128", a->N); return SVDQ_E_INVALID; }
        break;
    default:
        set_error("svdq_gemm_w4a4: unknown fuse mode %d", a->fuse);
        return SVDQ_E_INVALID;
    }
    if (a->out && (a->ldo < a->N || a->ldo % 4)) { set_error("svdq_gemm_w4a4: ldo=%d must be >= N and a multiple of 4", a->ldo); return SVDQ_E_INVALID; }
    if (((uintptr_t)a->act | (uintptr_t)a->wgt | (uintptr_t)a->lora_up | (uintptr_t)a->lora_act_in |
         (uintptr_t)a->next_lora_down | (uintptr_t)a->rotary_emb) & 15) {
        set_error("svdq_gemm_w4a4: act, wgt, lora_up, lora_act_in, next_lora_down, rotary_emb must be 16-byte aligned");
        return SVDQ_E_INVALID;
    }
    if (((uintptr_t)a->out | (uintptr_t)a->bias | (uintptr_t)a->next_smooth | (uintptr_t)a->norm_q | (uintptr_t)a->norm_k) & 7) {
        set_error("svdq_gemm_w4a4: out, bias, next_smooth, norm_q, norm_k must be 8-byte aligned");
        return SVDQ_E_INVALID;
    }

    GemmParams p;
    p.act = (const uint8_t *)a->act;
    p.wgt = (const uint8_t *)a->wgt;
    p.ascales = a->ascales;
    p.wscales = a->wscales;
    p.bias = a->bias;
    p.lora_act_in = a->lora_act_in;
    p.lora_up = a->lora_up;
    p.out = a->out;
    p.qout = (uint8_t *)a->qout;
    p.oscales = a->oscales;
    p.next_smooth = a->next_smooth;
    p.next_lora_down = a->next_lora_down;
    p.lora_act_out = a->lora_act_out;
    p.norm_q = a->norm_q;
    p.norm_k = a->norm_k;
    p.rotary_emb = a->rotary_emb;
    p.M = a->M; p.M_pad = a->M_pad; p.N = a->N; p.K = a->K; p.R = a->R; p.R2 = a->R2; p.ldo = a->ldo;
    for (int i = 0; i < MAX_LORA_TILES; i++) p.lora_scales[i] = (a->lora_scales && i < a->R / 16) ? a->lora_scales[i] : 1.0f;

    hipStream_t st = (hipStream_t)stream;
    const int prof = prof_begin(0, 2.0 * a->M_pad * (double)a->N * a->K + 2.0 * a->M_pad * (double)a->N * a->R, st);
    if (a->dtype == SVDQ_BF16) {
        if (a->act_unsigned) launch_fuse<SVDQ_BF16, true>(p, a->fuse, st);
        else launch_fuse<SVDQ_BF16, false>(p, a->fuse, st);
    } else {
        if (a->act_unsigned) launch_fuse<SVDQ_FP16, true>(p, a->fuse, st);
        else launch_fuse<SVDQ_FP16, false>(p, a->fuse, st);
    }
    prof_end(prof, st);
    return hip_check(hipGetLastError(), "svdq_gemm_w4a4 launch");
}
