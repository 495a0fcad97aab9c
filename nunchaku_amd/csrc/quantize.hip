// svdq_quantize_w4a4_act_fuse_lora: 16-bit activations -> FP6 operand image of the 4-bit codes +
// per-(row, group) scales, fused with the rank-R low-rank down projection.
//
// Replaces the reference's quantize_w4a4_fuse_lora_kernel (gemm_w4a4.cuh:1097-1184; launch
// gemm_w4a4_launch_impl.cuh:451-521).  Arithmetic (DESIGN.md "Quantiser"):
//   lora_act[m, r] = sum_k x[m,k] * lora_down[k,r]        16-bit MFMA, fp32 accumulate, on raw x
//   x_hat = round16(x * rcp(smooth))                      the reference's __fdividef form (svdq_common.h smooth_div16; oracle: quantize_envelope)
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

typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// grouped launch: rows >= split_rows read a second input and use a second parameter set (svdq_quantize_args.x2 ...)
struct QuantSecond {
    const void *x, *smooth, *lora_down, *mod_scale, *mod_shift;
    const float *ln_stats;
    int M, ldx, split_rows; // M: valid rows of the second stream; x == nullptr: off
};

// GLU = svdq_quantize_args.fuse_glu (reference load_act_to_fpsum<fuse_glu>, gemm_base.cuh:606-633): the input row holds 2K values,
// (value, gate) pairs, and the kernel quantises  x[k] = round16(value[k] * round16(silu(gate[k])))  -- silu in fp32 (gemm_utils.cuh:
// 323-327: x * sigmoid(x) with ex2.approx / rcp.approx; here the hardware exp2 and rcp), the product as a 16-bit multiply.
template <int DT, int RT32 /* 32-rank tiles held in registers */, int OCC /* workgroups per CU the register budget allows */, bool GLU = false>
__global__ __launch_bounds__(256, OCC) void quantize_kernel(const typename Half<DT>::T *__restrict__ x,
                                                       const typename Half<DT>::T *__restrict__ smooth,
                                                       const typename Half<DT>::T *__restrict__ lora_down, // [R][K]
                                                       uint8_t *__restrict__ act,
                                                       typename Half<DT>::T *__restrict__ ascales,
                                                       void *__restrict__ lora_act, int M, int K, int R, int ldx,
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

    const T *xrow = x + (size_t)row * ldx; // (GLU: ldx counts the 2K raw columns)
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
        u16x8 graw[GLU ? 2 : 1][GLU ? 8 : 1];
#pragma unroll
        for (int grp = 0; grp < 2; grp++) {
            const int kbase = kp * 128 + grp * 64 + 4 * h; // + 32t + 8c + e
#pragma unroll
            for (int tc = 0; tc < 8; tc++) {
                if constexpr (GLU) { // 4 (value, gate) pairs = 16 bytes; combined below, once every load is in flight
                    if (valid) graw[grp][tc] = *reinterpret_cast<const u16x8 *>(xrow + 2 * (kbase + 8 * tc));
                    else graw[grp][tc] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
                } else {
                    if (valid) xv[grp][tc] = *reinterpret_cast<const u16x4 *>(xrow + kbase + 8 * tc);
                    else xv[grp][tc] = u16x4{0, 0, 0, 0};
                }
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
        if constexpr (GLU) { // x <- round16(value * round16(gate * sigmoid(gate))); padded rows: 0 * silu(0) = 0
#pragma unroll
            for (int grp = 0; grp < 2; grp++)
#pragma unroll
                for (int tc = 0; tc < 8; tc++)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float v = h2f(hfrom<T>(graw[grp][tc][2 * e])), g = h2f(hfrom<T>(graw[grp][tc][2 * e + 1]));
                        const float sg = round16<T>(g * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * g)));
                        xv[grp][tc][e] = hbits(f2h<T>(v * sg));
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
                        xh[4 * tc + e] = smooth_div16<T>(h2f(hfrom<T>(xv[grp][tc][e])), __builtin_amdgcn_rcpf(sm));
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
                            lora_act_add(lora_act, (size_t)m * R + rank, s[i], use_atomics);
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
}


// --------------------------------------------------------------------------------------------------------------------
// Fast path (any rank since round 5; template MULTI for rank > 32): same arithmetic, same outputs, built for
// latency instead of registers.  Counters of the general kernel above at the FLUX.1 shape (profiles/r2_quantize_pmc.txt): a
// wave lives ~21 us for ONE chunk, its VALU is busy 14 % of that; 43 % is s_waitcnt on 80 narrow loads (64 of them re-read
// per-channel parameters every row tile needs alike), 38 % issue stalls; 236 VGPRs hold two workgroups per CU.  Here:
//   * whatever does not depend on the row goes through LDS, converted ONCE per wave: the smoothing factors with their
//     reciprocals, the modulation vectors (fp32, in the element order the packed arithmetic wants) and the chunk's slice of
//     lora_down (the MFMA B operand, XOR-swizzled 16-byte pieces, global -> LDS by LDS-DMA) -- 11 wide coalesced loads
//     instead of 64 broadcast ones, ~100 VGPRs less, 4 workgroups per CU;
//   * the element-wise chain runs on float pairs (v_pk_add / v_pk_mul / v_pk_fma_f32: two values per instruction) and is
//     laid out so that the pairs ARE the operand tuples of the FP6 pack instruction (elements 0,2 | 1,3 of a 4-channel
//     piece): no moves.  The 16-bit rounding points are unchanged (bit-identical codes and scales);
//   * with the LayerNorm front end the recomputed 16-bit activations are put back into channel order (two v_perm_b32 per piece)
//     for the low-rank MFMA, so the chunk's partial sums are bit-identical to the general kernel's.
// --------------------------------------------------------------------------------------------------------------------
struct QuantParams {
    const void *x, *smooth, *lora_down, *mod_scale, *mod_shift;
    const float *ln_stats;
    uint8_t *act;
    void *ascales;
    void *lora_act;
    int M, K, R, ldx, use_atomics; // use_atomics: bit 0 = K is sliced over workgroups, bit 1 = Q31.32 format (lora_act_add)
    QuantSecond s2;
};

constexpr int QV2_PARAM_BYTES = 4 * 128 * 4;     // per wave: smooth, 1/smooth, mod_scale, mod_shift as fp32
constexpr int QV2_SLAB_BYTES = 32 * 256;         // lora_down slice [32 ranks][128 channels] 16-bit
constexpr int QV2_WAVE_BYTES = QV2_PARAM_BYTES + QV2_SLAB_BYTES;
constexpr int QV2_WAVE_BYTES_MULTI = QV2_PARAM_BYTES + 2 * QV2_SLAB_BYTES; // MULTI: two slice buffers (the next 32-rank slab lands under the current one's MFMAs)

template <int DT> struct Pair16;
template <> struct Pair16<SVDQ_BF16> {
    static __device__ __forceinline__ v2f lo(unsigned d0, unsigned d1) { return v2f{__builtin_bit_cast(float, d0 << 16), __builtin_bit_cast(float, d1 << 16)}; }
    static __device__ __forceinline__ v2f hi(unsigned d0, unsigned d1) { return v2f{__builtin_bit_cast(float, d0 & 0xffff0000u), __builtin_bit_cast(float, d1 & 0xffff0000u)}; }
    static __device__ __forceinline__ unsigned pack(v2f v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
    static __device__ __forceinline__ v2f unpack(unsigned u) { return v2f{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)}; }
};
template <> struct Pair16<SVDQ_FP16> {
    static __device__ __forceinline__ v2f lo(unsigned d0, unsigned d1) { return v2f{(float)__builtin_bit_cast(f16x2, d0)[0], (float)__builtin_bit_cast(f16x2, d1)[0]}; }
    static __device__ __forceinline__ v2f hi(unsigned d0, unsigned d1) { return v2f{(float)__builtin_bit_cast(f16x2, d0)[1], (float)__builtin_bit_cast(f16x2, d1)[1]}; }
    static __device__ __forceinline__ unsigned pack(v2f v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2)); }
    static __device__ __forceinline__ v2f unpack(unsigned u) { return __builtin_convertvector(__builtin_bit_cast(f16x2, u), v2f); }
};

// One 4-channel piece (e0 e1 | e2 e3 in the dwords d0 | d1) of a lane's record through the LayerNorm front end and the
// smoothing division.  Parameters arrive in pair order (e0 e2 | e1 e3).  Out: the 16-bit activations the low-rank MFMA
// sees (a0 = e0 e1, a1 = e2 e3: natural order, as in the input), and x_hat as float pairs (e0 e2), (e1 e3).
//   bf16: float pairs, two values per instruction; the 16-bit rounding is its own instruction either way.
//   fp16: element by element, in the expression shapes of the general kernel -- the backend folds an fp32 operation and the
//         conversion that follows into ONE v_fma_mix*_f16 (a single rounding; oracle: _round16_fma), which a packed
//         multiply followed by a packed conversion would round twice (1 value in ~10^6 differs in the last bit).
template <int DT, bool LN, bool SMOOTH>
__device__ __forceinline__ void quant_piece(unsigned d0, unsigned d1, v2f mean2, v2f rstd2, bool valid, v4f ms, v4f mh, v4f rs,
                                            unsigned &a0, unsigned &a1, v2f &xe, v2f &xo) {
    using T = typename Half<DT>::T;
    using P16 = Pair16<DT>;
    a0 = d0; a1 = d1; // (without the front end the MFMA takes the input dwords as they are)
    if constexpr (DT == SVDQ_BF16) {
        xe = P16::lo(d0, d1); xo = P16::hi(d0, d1);
        if constexpr (LN) { // x <- round16(round16(round16((x - mean) * rstd) * scale) + shift); padded rows stay 0
            xe = P16::unpack(P16::pack((xe - mean2) * rstd2));
            xo = P16::unpack(P16::pack((xo - mean2) * rstd2));
            xe = P16::unpack(P16::pack(xe * v2f{ms[0], ms[1]})) + v2f{mh[0], mh[1]};
            xo = P16::unpack(P16::pack(xo * v2f{ms[2], ms[3]})) + v2f{mh[2], mh[3]};
            const unsigned he = valid ? P16::pack(xe) : 0u, ho = valid ? P16::pack(xo) : 0u; // (e0 e2), (e1 e3)
            xe = P16::unpack(he); xo = P16::unpack(ho);
            a0 = __builtin_amdgcn_perm(ho, he, 0x05040100u); // (e0 e1)
            a1 = __builtin_amdgcn_perm(ho, he, 0x07060302u); // (e2 e3)
        }
        if constexpr (SMOOTH) {
            xe = P16::unpack(P16::pack(xe * v2f{rs[0], rs[1]})); // smooth_div16 on pairs
            xo = P16::unpack(P16::pack(xo * v2f{rs[2], rs[3]}));
        }
    } else {
        const f16x2 h0 = __builtin_bit_cast(f16x2, d0), h1 = __builtin_bit_cast(f16x2, d1);
        T x16[4] = {h0[0], h1[0], h0[1], h1[1]}; // pair order: e0 e2 e1 e3
        float xf[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if constexpr (LN) {
                const float lnv = round16<T>((h2f(x16[i]) - mean2[0]) * rstd2[0]);
                const float y = round16<T>(lnv * ms[i]) + mh[i];
                x16[i] = valid ? f2h<T>(y) : (T)0.f;
            }
            xf[i] = h2f(x16[i]);
            if constexpr (SMOOTH) xf[i] = smooth_div16<T>(xf[i], rs[i]);
        }
        if constexpr (LN) {
            a0 = __builtin_bit_cast(unsigned, f16x2{x16[0], x16[2]});
            a1 = __builtin_bit_cast(unsigned, f16x2{x16[1], x16[3]});
        }
        xe = v2f{xf[0], xf[1]}; xo = v2f{xf[2], xf[3]};
    }
}

// MULTI (round 5): rank > 32 -- the r128 checkpoints, a runtime LoRA on top of the rank-32 branch.  The chunk's 16-bit activations (the MFMA A operands:
// 32 registers) are kept, and behind the quantisation proper the remaining 32-rank slabs of lora_down take the same path one after the other: slice ->
// LDS by LDS-DMA, 8 MFMAs, the four waves' partial tiles combined in the fixed order, one set of atomics.  A kernel of its own: the rank <= 32 kernels keep
// their registers (4 workgroups per CU) and their instruction stream.
template <int DT, bool LORA, bool LN, bool SMOOTH, bool MULTI = false>
__global__ __launch_bounds__(256, MULTI ? 2 : 4) void quantize_kernel_v2(QuantParams p) {
    static_assert(!MULTI || LORA, "MULTI: more than one 32-rank slab of lora_down");
    using T = typename Half<DT>::T;
    using V8 = typename Half<DT>::V8;
    using P16 = Pair16<DT>;
    constexpr int WAVE_BYTES = MULTI ? QV2_WAVE_BYTES_MULTI : QV2_WAVE_BYTES;
    __shared__ __attribute__((aligned(16))) uint8_t lds[4 * WAVE_BYTES];
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    typedef __attribute__((address_space(3))) v4f lds_v4f;
    typedef __attribute__((address_space(3))) v2f lds_v2f;
    typedef __attribute__((address_space(3))) v4i lds_v4i;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) u32x2 lds_u2;
    typedef __attribute__((address_space(3))) v16f lds_v16f;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int K = p.K, KP = K / 128;
    const int slices = (KP + 3) / 4;
    const int rt = blockIdx.x / slices, kp = (blockIdx.x % slices) * 4 + wave;
    const bool active = kp < KP; // wave-uniform
    lds_u8 *const W = (lds_u8 *)lds + wave * WAVE_BYTES; // this wave's private region: no workgroup barrier before the final sum

    // grouped launch: this row tile's stream (block-uniform); outputs stay addressed by the joint row tile index rt
    const bool second = p.s2.x != nullptr && rt * 32 >= p.s2.split_rows;
    const T *x = (const T *)(second ? p.s2.x : p.x);
    const T *smooth = (const T *)(second ? p.s2.smooth : p.smooth);
    const T *lora_down = (const T *)(second ? p.s2.lora_down : p.lora_down);
    const float *ln_stats = second ? p.s2.ln_stats : p.ln_stats;
    const T *mod_scale = (const T *)(second ? p.s2.mod_scale : p.mod_scale);
    const T *mod_shift = (const T *)(second ? p.s2.mod_shift : p.mod_shift);
    const int ldx = second ? p.s2.ldx : p.ldx, M = second ? p.s2.M : p.M;
    const int row = rt * 32 + r - (second ? p.s2.split_rows : 0); // row inside this stream's input
    const bool valid = row < M;
    constexpr bool ln = LN; // (the launcher checks that both streams of a grouped launch agree)

    v16f accL;
#pragma unroll
    for (int j = 0; j < 16; j++) accL[j] = 0.f;
    v4i aop[MULTI ? 2 : 1][MULTI ? 4 : 1] = {}; // MULTI: the chunk's MFMA A operands, kept for the slabs beyond rank 32
    unsigned ldo_keep[4] = {0, 0, 0, 0};

    if (active) {
        // ---- 1. every global access of the chunk goes out back to back: 16 activation loads (this lane's record: row r,
        //         channels 64 grp + 8 tc + 4 h + e, 8 bytes per piece), the lora_down slice as LDS-DMA (no registers:
        //         instruction i = ranks 4i .. 4i+3, whole 256-byte rows, a lane fetches the piece whose swizzled slot it
        //         fills), the per-channel parameters (lane l: channels 2l, 2l+1).  Buffer loads: a row >= M / a rank >= R
        //         lies beyond the descriptor's range and reads as zero -- no exec-mask branch around any of them (ranks >= R:
        //         whatever their LDS rows hold only reaches accumulator columns that are never stored).
        // (the launcher sends inputs whose padded extent exceeds the descriptor's 2 GiB reach to the general kernel)
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)(((long long)(M - 1) * ldx + K) * 2), 0x00020000);
        unsigned xoff = ((unsigned)row * (unsigned)ldx + kp * 128 + 4 * h) * 2u; // row >= M: out of range by construction
        unsigned ldo[4]; // slice offsets of instructions i = j, j + 4 (rank 4j + lane/16 [+ 16]; its swizzle only depends on j)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned rank = 4 * j + (lane >> 4);
            ldo[j] = (rank * (unsigned)K + kp * 128 + (((lane & 15) ^ (rank & 15)) * 8)) * 2u;
        }
        // all address arithmetic happens BEFORE the first load is issued (the backend otherwise interleaves it with the loads
        // and, reusing a load's destination as the undefined half of a 64-bit multiply-add operand, waits for that load)
        asm volatile("" : "+v"(xoff), "+v"(ldo[0]), "+v"(ldo[1]), "+v"(ldo[2]), "+v"(ldo[3]));
        if constexpr (MULTI) {
#pragma unroll
            for (int j = 0; j < 4; j++) ldo_keep[j] = ldo[j];
        }
        uint2 xv[2][8];
#pragma unroll
        for (int grp = 0; grp < 2; grp++)
#pragma unroll
            for (int tc = 0; tc < 8; tc++) {
                const auto w = __builtin_amdgcn_raw_buffer_load_b64(rx, xoff + (grp * 64 + 8 * tc) * 2, 0, 0);
                xv[grp][tc] = make_uint2((unsigned)w[0], (unsigned)w[1]);
            }
        if constexpr (LORA) {
            const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void *)lora_down, 0, (int)((size_t)p.R * K * 2), 0x00020000);
            typedef __attribute__((address_space(3))) void lds_void;
#pragma unroll
            for (int i = 0; i < 8; i++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rl, (lds_void *)(W + QV2_PARAM_BYTES + i * 1024), 16, ldo[i & 3] + (i >> 2) * 16 * K * 2, 0, 0, 0); // (rank offset in the range-checked VGPR offset: rank 16 reads its upper half as zero)
        }
        const int ch = kp * 128 + 2 * lane;
        unsigned sm2 = 0, ms2 = 0, mh2 = 0;
        if constexpr (SMOOTH) sm2 = *reinterpret_cast<const unsigned *>(smooth + ch);
        if constexpr (LN) {
            ms2 = *reinterpret_cast<const unsigned *>(mod_scale + ch);
            mh2 = *reinterpret_cast<const unsigned *>(mod_shift + ch);
        }
        v2f mean2 = {0.f, 0.f}, rstd2 = {0.f, 0.f};
        if constexpr (LN) { // (a padded row borrows the last row's statistics: its result is forced to zero below)
            const float2 st = *reinterpret_cast<const float2 *>(ln_stats + 2 * (size_t)(valid ? row : M - 1));
            mean2 = v2f{st.x, st.x};
            rstd2 = v2f{st.y, st.y};
        }
        // ---- 3. parameters -> LDS as fp32 in pair order: a 4-channel piece (e0 e1 e2 e3) is stored (e0 e2 | e1 e3).  Lane l
        //         owns channels 2l, 2l+1 = elements (0,1) or (2,3) of piece l/2: slots {0,2} or {1,3} of that piece
        {
            const int slot = (lane >> 1) * 4 + (lane & 1);
            __attribute__((address_space(3))) float *P = (__attribute__((address_space(3))) float *)W;
            if constexpr (SMOOTH) {
                const v2f sm = P16::unpack(sm2);  // (only the reciprocals are kept: smooth_div16)
                P[128 + slot] = __builtin_amdgcn_rcpf(sm[0]); P[128 + slot + 2] = __builtin_amdgcn_rcpf(sm[1]);
            }
            if constexpr (LN) {
                const v2f ms = P16::unpack(ms2), mh = P16::unpack(mh2);
                P[256 + slot] = ms[0]; P[256 + slot + 2] = ms[1];
                P[384 + slot] = mh[0]; P[384 + slot + 2] = mh[1];
            }
        }
        // (one wave, in-order LDS: its own reads below see its own writes; no barrier)

        // ---- 4. the two groups of the chunk
        uint8_t *dst = p.act + ((size_t)rt * KP + kp) * F6_CHUNK + (size_t)lane * 16;
#pragma unroll
        for (int grp = 0; grp < 2; grp++) {
            v16f ev, od; // elements 0,2 / 1,3 of the lane's eight 4-channel pieces = the two sources of the FP6 pack
            float amax = 0.f;
#pragma unroll
            for (int q = 0; q < 4; q++) { // pieces 2q, 2q+1, then the low-rank MFMA that consumes exactly those two
                unsigned ae[2], ao[2]; // 16-bit activations as the MFMA sees them: (e0 e1), (e2 e3) of the two pieces
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int tc = 2 * q + t;
                    const int pofs = (grp * 64 + 8 * tc + 4 * h) * 4; // byte offset of this piece in a parameter array
                    v4f ms = {}, mh = {}, rs = {};
                    if constexpr (LN) { ms = *(const lds_v4f *)(W + 2 * 512 + pofs); mh = *(const lds_v4f *)(W + 3 * 512 + pofs); }
                    if constexpr (SMOOTH) rs = *(const lds_v4f *)(W + 512 + pofs);
                    v2f xe, xo;
                    quant_piece<DT, LN, SMOOTH>(xv[grp][tc].x, xv[grp][tc].y, mean2, rstd2, valid, ms, mh, rs, ae[t], ao[t], xe, xo);
                    ev[2 * tc] = xe[0]; ev[2 * tc + 1] = xe[1];
                    od[2 * tc] = xo[0]; od[2 * tc + 1] = xo[1];
                    amax = fmaxf(fmaxf(amax, fabsf(xe[0])), fabsf(xe[1]));
                    amax = fmaxf(fmaxf(amax, fabsf(xo[0])), fabsf(xo[1]));
                }
                if constexpr (LORA) {
                    // D[m][rank] += x[m][k] * lora_down[k][rank]: pieces 2q, 2q+1 of every lane are k-slots 8h .. 8h+7
                    const V8 a = __builtin_bit_cast(V8, v4i{(int)ae[0], (int)ao[0], (int)ae[1], (int)ao[1]});
                    if constexpr (MULTI) aop[grp][q] = v4i{(int)ae[0], (int)ao[0], (int)ae[1], (int)ao[1]};
                    const int piece = grp * 8 + 2 * q; // 16-byte pieces of rank r's row: this lane's half (8 h) of pieces `piece`, `piece + 1`
                    if (grp == 0 && q == 0) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the LDS-DMA of the slice has landed
                        if constexpr (MULTI) { // ranks 32 .. 63 -> the second slice buffer, under the whole quantisation pass
                            const __amdgpu_buffer_rsrc_t rl1 = __builtin_amdgcn_make_buffer_rsrc((void *)lora_down, 0, (int)((size_t)p.R * K * 2), 0x00020000);
                            typedef __attribute__((address_space(3))) void lds_void;
#pragma unroll
                            for (int i = 0; i < 8; i++)
                                // (the rank offset rides in the VGPR offset, which the descriptor's range check covers: ranks >= R read as zero.  ADVICE r5:
                                //  an SGPR offset is outside that check)
                                __builtin_amdgcn_raw_ptr_buffer_load_lds(rl1, (lds_void *)(W + QV2_PARAM_BYTES + QV2_SLAB_BYTES + i * 1024), 16,
                                                                         ldo_keep[i & 3] + ((i >> 2) * 16 + 32) * K * 2, 0, 0, 0);
                        }
                    }
                    const u32x2 b0 = *(const lds_u2 *)(W + QV2_PARAM_BYTES + r * 256 + ((piece ^ (r & 15)) << 4) + 8 * h);
                    const u32x2 b1 = *(const lds_u2 *)(W + QV2_PARAM_BYTES + r * 256 + (((piece + 1) ^ (r & 15)) << 4) + 8 * h);
                    const V8 b = __builtin_bit_cast(V8, v4i{(int)b0[0], (int)b0[1], (int)b1[0], (int)b1[1]});
                    accL = Half<DT>::mfma32(a, b, accL);
                }
                // scheduling fence: keeps the backend from hoisting all 64 parameter reads of the chunk to the top (it then needs
                // > 128 VGPRs and spills); the other three waves of the SIMD cover the LDS latency
                asm volatile("" ::: "memory");
            }
            amax = fmaxf(amax, __shfl_xor(amax, 32));
            const float scale = amax * (1.0f / 7.0f);
            const float rscale = scale == 0.f ? 0.f : 1.0f / scale;
            // q = rne(x_hat / scale) as FP6 e2m3 = q/8, 32 x 6 bits in ONE v_cvt_scalef32_2xpk16_fp6_f32 (see the general kernel)
            ev = ev * rscale;
            od = od * rscale;
            const v6i pk = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(ev, od, 8.0f);
            if (grp == 0) {
                *reinterpret_cast<uint4 *>(dst) = make_uint4((unsigned)pk[0], (unsigned)pk[1], (unsigned)pk[2], (unsigned)pk[3]);
                *reinterpret_cast<uint2 *>(dst + F6_PLANE) = make_uint2((unsigned)pk[4], (unsigned)pk[5]);
            } else {
                *reinterpret_cast<uint2 *>(dst + F6_PLANE + 8) = make_uint2((unsigned)pk[0], (unsigned)pk[1]);
                *reinterpret_cast<uint4 *>(dst + 2 * F6_PLANE) = make_uint4((unsigned)pk[2], (unsigned)pk[3], (unsigned)pk[4], (unsigned)pk[5]);
            }
            // S image: [rt][kp][grp][32]; lane (r, h) writes group h
            if (h == grp) ((T *)p.ascales)[(((size_t)rt * KP + kp) * 2 + grp) * 32 + r] = f2h<T>(scale);
        }
    }

    if constexpr (LORA) {
        // combine the four waves' partial sums in a fixed order (each wave parks its tile in its own region)
        auto combine = [&](int rank0, int buf) { // (the tile is parked in the slice buffer whose slab it came from: this wave has finished reading it)
            *(lds_v16f *)(W + QV2_PARAM_BYTES + buf * QV2_SLAB_BYTES + lane * 64) = accL;
            __syncthreads();
            if (wave == 0) {
                const lds_u8 *B = (const lds_u8 *)lds + QV2_PARAM_BYTES + buf * QV2_SLAB_BYTES + lane * 64;
                const v16f s = ((*(const lds_v16f *)(B) + *(const lds_v16f *)(B + WAVE_BYTES)) + *(const lds_v16f *)(B + 2 * WAVE_BYTES)) +
                               *(const lds_v16f *)(B + 3 * WAVE_BYTES);
                if (rank0 + r < p.R) { // C layout: col (rank) = lane & 31, row = (i&3) + 8*(i>>2) + 4*(lane>>5)
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int m = rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                        lora_act_add(p.lora_act, (size_t)m * p.R + rank0 + r, s[i], p.use_atomics);
                    }
                }
            }
        };
        if constexpr (!MULTI) {
            combine(0, 0);
        } else {
            // the slabs beyond rank 32 (up to rank 160: four more): same slice layout, same MFMA operand order.  Slab s sits in the wave's slice buffer s & 1
            // (slab 1 was requested under the quantisation pass); at the top of its turn the slab after it is requested into the other buffer (this wave
            // has finished reading it: the buffers are private to the wave, no workgroup barrier in this loop).  Every slab keeps its own accumulator
            // tile; the four waves' tiles are combined once at the end, one slab per wave.
            const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void *)lora_down, 0, (int)((size_t)p.R * K * 2), 0x00020000);
            typedef __attribute__((address_space(3))) void lds_void;
            v16f accS[4];
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int j = 0; j < 16; j++) accS[s][j] = 0.f;
#pragma unroll
            for (int s = 1; s <= 4; s++) {
                const int rank0 = 32 * s, buf = s & 1;
                if (rank0 < p.R && active) { // wave-uniform
                    if (rank0 + 32 < p.R) {
#pragma unroll
                        for (int i = 0; i < 8; i++) // (ranks >= R lie beyond the descriptor's range and read as zero)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rl, (lds_void *)(W + QV2_PARAM_BYTES + (buf ^ 1) * QV2_SLAB_BYTES + i * 1024), 16,
                                                                     ldo_keep[i & 3] + ((i >> 2) * 16 + rank0 + 32) * K * 2, 0, 0, 0);
                        asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); // this slab has landed (in-order retirement); the next one may still fly
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
#pragma unroll
                    for (int grp = 0; grp < 2; grp++)
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int piece = grp * 8 + 2 * q;
                            const lds_u8 *S = W + QV2_PARAM_BYTES + buf * QV2_SLAB_BYTES;
                            const u32x2 b0 = *(const lds_u2 *)(S + r * 256 + ((piece ^ (r & 15)) << 4) + 8 * h);
                            const u32x2 b1 = *(const lds_u2 *)(S + r * 256 + (((piece + 1) ^ (r & 15)) << 4) + 8 * h);
                            const V8 b = __builtin_bit_cast(V8, v4i{(int)b0[0], (int)b0[1], (int)b1[0], (int)b1[1]});
                            accS[s - 1] = Half<DT>::mfma32(__builtin_bit_cast(V8, aop[grp][q]), b, accS[s - 1]);
                        }
                }
            }
            // combine: every wave parks its tiles of slabs 0 .. 3 in its own region (4 x 4 KiB: the parameter block and both slice buffers are dead), then wave w
            // sums slab w over the four regions in the fixed order and adds it to lora_act -- the atomics of the slabs go out in parallel; slab 4 (rank > 128)
            // takes a second, one-tile round
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            *(lds_v16f *)(W + 0 * 4096 + lane * 64) = accL;
#pragma unroll
            for (int s = 1; s < 4; s++) *(lds_v16f *)(W + s * 4096 + lane * 64) = accS[s - 1];
            __syncthreads();
            auto sum_slab = [&](int slot, int rank0) {
                const lds_u8 *B = (const lds_u8 *)lds + slot * 4096 + lane * 64;
                const v16f sm = ((*(const lds_v16f *)(B) + *(const lds_v16f *)(B + WAVE_BYTES)) + *(const lds_v16f *)(B + 2 * WAVE_BYTES)) + *(const lds_v16f *)(B + 3 * WAVE_BYTES);
                if (rank0 + r < p.R) {
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int m = rt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                        lora_act_add(p.lora_act, (size_t)m * p.R + rank0 + r, sm[i], p.use_atomics);
                    }
                }
            };
            if (32 * wave < p.R) sum_slab(wave, 32 * wave);
            if (p.R > 128) { // block-uniform
                __syncthreads();
                *(lds_v16f *)(W + lane * 64) = accS[3];
                __syncthreads();
                if (wave == 0) sum_slab(0, 128);
            }
        }
    }
}

template <int DT>
static int launch_quantize(const svdq_quantize_args *a, hipStream_t st) {
    using T = typename Half<DT>::T;
    const int KP = a->K / 128, tiles = a->M_pad / 32;
    // enough workgroups to fill 256 CUs several times over, but at least one chunk per wave
    int cpw = 4; // chunks per workgroup = 4 waves x 1 chunk: many short waves hide the HBM round trip
    while ((long)tiles * ((KP + cpw - 1) / cpw) > 8192) cpw *= 2;
    if (cpw > KP) cpw = ((KP + 3) / 4) * 4;
    const int slices = (KP + cpw - 1) / cpw;
    const int q32 = a->lora_act_format == SVDQ_LORA_ACT_Q32 || a->lora_act_format == SVDQ_LORA_ACT_Q32_RUNS;
    const int atomics = (slices > 1 ? 1 : 0) | (q32 ? 2 : 0); // lora_act_add mode
    if (a->R > 0 && (atomics & 1) && !a->lora_act_zeroed) {
        // the reference zeroes the buffer inside the op as well (launch_impl.cuh:487)
        int rc = hip_check(hipMemsetAsync(a->lora_act, 0, (size_t)a->M_pad * a->R * (q32 ? 8 : 4), st), "svdq_quantize memset");
        if (rc) return rc;
    }
    dim3 grid(tiles * slices), block(256);
    const int rt32 = (a->R + 31) / 32;
    QuantSecond s2{a->x2, a->smooth2, a->lora_down2, a->mod_scale2, a->mod_shift2, a->ln_stats2, a->M2, a->ldx2, a->split_rows};
    if (!a->fuse_glu && a->R <= 160 && cpw == 4 && (long long)a->M_pad * (a->ldx > a->ldx2 ? a->ldx : a->ldx2) * 2 < 0x7fffffffLL && (long long)a->R * a->K * 2 < 0x7fffffffLL) {
        // fast path: one chunk per wave, 4 waves = 4 neighbouring chunks of one row tile (same grid as the general kernel at cpw = 4)
        QuantParams qp{a->x, a->smooth, a->lora_down, a->mod_scale, a->mod_shift, a->ln_stats, (uint8_t *)a->act, a->ascales, a->lora_act,
                       a->M, a->K, a->R, a->ldx, atomics, s2};
        const int sel = (a->R > 32 ? 8 : 0) | (a->R > 0 ? 4 : 0) | (a->ln_stats ? 2 : 0) | (a->smooth ? 1 : 0); // (rank > 160: the general kernel, above)
        switch (sel) {
#define SVDQ_QV2(n, LORA, LN, SM, MULTI) case n: hipLaunchKernelGGL((quantize_kernel_v2<DT, LORA, LN, SM, MULTI>), grid, block, 0, st, qp); break;
            SVDQ_QV2(0, false, false, false, false) SVDQ_QV2(1, false, false, true, false) SVDQ_QV2(2, false, true, false, false) SVDQ_QV2(3, false, true, true, false)
            SVDQ_QV2(4, true, false, false, false) SVDQ_QV2(5, true, false, true, false) SVDQ_QV2(6, true, true, false, false) SVDQ_QV2(7, true, true, true, false)
            SVDQ_QV2(12, true, false, false, true) SVDQ_QV2(13, true, false, true, true) SVDQ_QV2(14, true, true, false, true) SVDQ_QV2(15, true, true, true, true)
#undef SVDQ_QV2
        }
        return hip_check(hipGetLastError(), "svdq_quantize_w4a4_act_fuse_lora launch");
    }
#define SVDQ_LAUNCH_QG(RT, GLU)                                                                                      \
    hipLaunchKernelGGL((quantize_kernel<DT, RT, 1, GLU>), grid, block, 0, st, (const T *)a->x, (const T *)a->smooth,     \
                       (const T *)a->lora_down, (uint8_t *)a->act, (T *)a->ascales, a->lora_act, a->M, a->K, a->R,    \
                       a->ldx, cpw, atomics, a->ln_stats, (const T *)a->mod_scale, (const T *)a->mod_shift, s2)
#define SVDQ_LAUNCH_Q(RT) do { if (a->fuse_glu) SVDQ_LAUNCH_QG(RT, true); else SVDQ_LAUNCH_QG(RT, false); } while (0)
    if (rt32 == 0) SVDQ_LAUNCH_Q(0);
    else if (rt32 <= 1) SVDQ_LAUNCH_Q(1);
    else if (rt32 <= 2) SVDQ_LAUNCH_Q(2);
    else if (rt32 <= 4) SVDQ_LAUNCH_Q(4);
    else SVDQ_LAUNCH_Q(8);
#undef SVDQ_LAUNCH_Q
#undef SVDQ_LAUNCH_QG
    return hip_check(hipGetLastError(), "svdq_quantize_w4a4_act_fuse_lora launch");
}

} // namespace svdq

using namespace svdq;

extern "C" int svdq_quantize_w4a4_act_fuse_lora(const svdq_quantize_args *a, void *stream) {
    if (!a) { set_error("svdq_quantize: args is NULL"); return SVDQ_E_INVALID; }
    if (a->fp4) { set_error("svdq_quantize: fp4 (NVFP4) is not supported on gfx950"); return SVDQ_E_UNSUPPORTED; }
    if (a->fuse_glu && (a->ln_stats || a->x2)) { set_error("svdq_quantize: fuse_glu does not combine with the LayerNorm front end or a grouped launch"); return SVDQ_E_INVALID; }
    if (a->lora_act_format != SVDQ_LORA_ACT_F32 && a->lora_act_format != SVDQ_LORA_ACT_Q32 && a->lora_act_format != SVDQ_LORA_ACT_Q32_RUNS) { set_error("svdq_quantize: unknown lora_act_format %d", a->lora_act_format); return SVDQ_E_INVALID; }
    if (a->lora_act_format != SVDQ_LORA_ACT_F32 && ((uintptr_t)a->lora_act & 7)) { set_error("svdq_quantize: a Q31.32 lora_act must be 8-byte aligned"); return SVDQ_E_INVALID; }
    if (!a->x || !a->act || !a->ascales) { set_error("svdq_quantize: x, act and ascales are required"); return SVDQ_E_INVALID; }
    if (a->M <= 0 || a->M_pad < a->M || a->M_pad % 256) {
        set_error("svdq_quantize: need 0 < M=%d <= M_pad=%d and M_pad %% 256 == 0", a->M, a->M_pad);
        return SVDQ_E_INVALID;
    }
    if (a->K <= 0 || a->K % 128) { set_error("svdq_quantize: K=%d must be a positive multiple of 128", a->K); return SVDQ_E_INVALID; }
    if (a->ldx < a->K || a->ldx % 4) { set_error("svdq_quantize: ldx=%d must be >= K and a multiple of 4", a->ldx); return SVDQ_E_INVALID; }
    if (a->fuse_glu && (a->ldx < 2 * a->K || a->ldx % 8 || ((uintptr_t)a->x & 15))) {
        set_error("svdq_quantize: fuse_glu reads rows of 2K (value, gate) pairs: ldx=%d must be >= 2K and a multiple of 8, x 16-byte aligned", a->ldx);
        return SVDQ_E_INVALID;
    }
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
    const double bytes = (double)(a->M + (a->x2 ? a->M2 : 0)) * a->K * (a->fuse_glu ? 4 : 2) + (double)a->M_pad * a->K * 3 / 4 + (double)a->M_pad * (a->K / 64) * 2 +
                         (double)a->M_pad * a->R * 4 + (double)a->K * a->R * 2;
    const int prof = prof_begin(1, bytes, st);
    int rc = a->dtype == SVDQ_BF16 ? launch_quantize<SVDQ_BF16>(a, st) : launch_quantize<SVDQ_FP16>(a, st);
    prof_end(prof, st);
    return rc;
}
