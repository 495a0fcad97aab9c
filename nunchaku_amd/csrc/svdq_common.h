// Shared device/host helpers of the gfx950 SVDQuant kernels.
//
// Operand images (DESIGN.md "Data layout in HBM").  Both GEMM operands -- the activations produced
// by our quantiser / by a GELU_QUANT epilogue, and the weights after svdq_repack_qweight -- are
// stored as the register image of v_mfma_scale_f32_32x32x64_f8f6f4 with FP6 (e2m3) operands: a 4-bit
// code q is exactly the FP6 value q/8 (sign<<5 | |q| for signed codes, the code itself for unsigned
// ones), so the matrix core multiplies the integer codes exactly and no unpack work is left in
// the main loop.
//
//   "F6" image of a matrix [ROWS, K] of 4-bit codes, ROWS % 32 == 0, K % 128 == 0, KP = K / 128:
//     chunk (rt = row/32, kp = k/128) : 3072 bytes at ((rt*KP + kp) * 3072)
//     a chunk is 3 planes of 1024 bytes; plane p holds bytes [16p, 16p+16) of every lane record,
//     lane l at plane offset 16*l  (so one 16 B/lane wave load or LDS-DMA moves one plane);
//     lane l = (row & 31) | (h << 5); its 48-byte record = group 2kp in bytes [0,24), group 2kp+1
//     in bytes [24,48); inside a group record element j (0..31) sits at bits [6j, 6j+6);
//     the group-local channel of (h, j) is  k = 32*(j>>4) + 8*((j>>2)&3) + 4*h + (j&3)
//     (the order in which a 32x32 MFMA accumulator tile holds 64 output columns, so a GELU_QUANT
//     epilogue packs the next layer's activations without any cross-lane movement).
//
//   "S" image of the per-(row, 64-channel group) scales: [ROWS/32][KP][2][32] 16-bit, i.e. the 64
//     scales one K-step (two groups) needs for 32 rows are 128 contiguous bytes.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/svdq_amd.h"

namespace svdq {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v6i __attribute__((ext_vector_type(6)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

constexpr int GROUP = 64;   // int4 quantisation group (reference: gemm_base.cuh:89-95)

// ---- 16-bit model dtype traits -------------------------------------------------------------
template <int DT> struct Half;
template <> struct Half<SVDQ_BF16> {
    using T = __bf16;
    using V8 = bf16x8;
    static __device__ __forceinline__ v4f mfma(V8 a, V8 b, v4f c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ v16f mfma32(V8 a, V8 b, v16f c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Half<SVDQ_FP16> {
    using T = _Float16;
    using V8 = f16x8;
    static __device__ __forceinline__ v4f mfma(V8 a, V8 b, v4f c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ v16f mfma32(V8 a, V8 b, v16f c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

template <typename T> __device__ __forceinline__ float h2f(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T f2h(float v) { return (T)v; } // RNE

template <typename T> __device__ __forceinline__ unsigned short hbits(T v) {
    return __builtin_bit_cast(unsigned short, v);
}
template <typename T> __device__ __forceinline__ T hfrom(unsigned short b) { return __builtin_bit_cast(T, b); }

// round a float to the 16-bit type and come back (the reference keeps its tile in 16-bit between
// epilogue stages; this reproduces those rounding points)
template <typename T> __device__ __forceinline__ float round16(float v) { return (float)(T)v; }
// the same for two values at once.  bf16: ONE v_cvt_pk_bf16_f32 (RNE, both values) + a shift and a mask to come back -- 3 instructions where two
// round16 cost 4.  fp16 keeps the scalar form on purpose: the backend folds an fp32 operation and the conversion that follows it into one
// v_fma_mix*_f16 (one rounding, as the oracle's _round16_fma models it); an explicit packed convert would round the fp32 result a second time.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ f32x2_t round16_pair(float a, float b) {
    if constexpr (sizeof(T) == 2 && !__is_same(T, _Float16)) {
        const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
        return (f32x2_t){__builtin_bit_cast(float, pk << 16), __builtin_bit_cast(float, pk & 0xffff0000u)};
    } else {
        return (f32x2_t){round16<T>(a), round16<T>(b)};
    }
}

// x / smooth the way the reference divides (h2div -> __fdividef, gemm_utils.cuh:329-344; gemm_w4a4.cuh:990-993): the numerator times the hardware
// reciprocal of the denominator -- ONE multiply where the exactly rounded quotient (rounds 1-4: two Newton steps on the reciprocal, 5 operations) cost
// 4.5 of the quantiser's 13 VALU operations per element.  v_rcp_f32 is accurate to 1 ulp, the product adds half an ulp: inside __fdividef's documented
// 2 ulp, i.e. exactly as faithful to the reference as the IEEE quotient (oracle: quantize_envelope; the 16-bit result differs from the IEEE one on
// ~2^-15 (bf16) / ~2^-12 (fp16) of the elements).  `rs` = __builtin_amdgcn_rcpf(smooth): every kernel of the library forms it the same way, so the
// stand-alone quantiser, its fast path and the attention / GEMM epilogues that quantise emit identical bits for identical inputs.
template <typename T> __device__ __forceinline__ float smooth_div16(float x, float rs) { return round16<T>(x * rs); }

// ---- order-independent accumulation format of the low-rank activations ("deterministic mode") -------------------
// fp32 atomics make lora_act depend on the arrival order of the K-slice / column-tile partial sums (the reference's own
// behaviour, lora.cuh:82-94,323).  With SVDQ_LORA_ACT_Q32 the partial sums are converted to Q31.32 fixed point
// (int64: value * 2^32, floor) and accumulated with 64-bit INTEGER atomics: integer addition is associative, so the
// result is bit-reproducible whatever the order.  Resolution 2^-32 (finer than an fp32 ulp for |v| >= 2^-8), range +-2^31.
__device__ __forceinline__ long long float_to_q32(float v) {
    const float fl = __builtin_floorf(v);
    const int hi = (int)fl;                                           // saturates
    const unsigned lo = (unsigned)((v - fl) * 4294967296.0f);         // v - floor(v) is exact; saturates at 2^32 - 1
    return (long long)(((unsigned long long)(unsigned)hi << 32) | lo);
}
__device__ __forceinline__ float q32_to_float(int lo, int hi) {
    return __builtin_fmaf((float)(unsigned)lo, 0x1p-32f, (float)hi);
}
// one partial sum into lora_act[idx]; mode bit 0: other workgroups add to the same element (atomics), bit 1: Q31.32 format
__device__ __forceinline__ void lora_act_add(void *base, size_t idx, float v, int mode) {
    if (mode & 2) {
        const long long q = float_to_q32(v);
        if (mode & 1) __hip_atomic_fetch_add((long long *)base + idx, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else ((long long *)base)[idx] = q;
    } else {
        if (mode & 1) unsafeAtomicAdd((float *)base + idx, v);
        else ((float *)base)[idx] = v;
    }
}

// ---- F6 / S image addressing ----------------------------------------------------------------
constexpr int F6_CHUNK = 3072; // bytes of one (32 rows x 128 k) chunk
constexpr int F6_PLANE = 1024;

// element index j (0..31) and lane half h of group-local channel kg (0..63)
__host__ __device__ __forceinline__ int f6_half(int kg) { return (kg >> 2) & 1; }
__host__ __device__ __forceinline__ int f6_elem(int kg) { return ((kg >> 5) << 4) | (((kg >> 3) & 3) << 2) | (kg & 3); }
// group-local channel of (h, j)
__host__ __device__ __forceinline__ int f6_channel(int h, int j) { return 32 * (j >> 4) + 8 * ((j >> 2) & 3) + 4 * h + (j & 3); }

// code (row, k) lives in the 48-byte lane record starting (plane 0) at f6_record_base(), at bit
// f6_record_bit(k) of that record.
__host__ __device__ __forceinline__ size_t f6_record_base(int row, int k, int KP) {
    // offset of byte 0 of plane 0 of the lane record (plane p adds p*1024)
    const int kg = k & 63;
    const int lane = (row & 31) | (f6_half(kg) << 5);
    return ((size_t)(row >> 5) * KP + (k >> 7)) * F6_CHUNK + (size_t)lane * 16;
}
__host__ __device__ __forceinline__ int f6_record_bit(int k) { return 192 * ((k >> 6) & 1) + 6 * f6_elem(k & 63); }
// byte b (0..47) of a lane record lives at record_base + (b >> 4) * 1024 + (b & 15)
__host__ __device__ __forceinline__ size_t f6_byte(size_t record_base, int b) { return record_base + (size_t)(b >> 4) * F6_PLANE + (b & 15); }

__host__ __device__ __forceinline__ size_t simg_index(int row, int g, int KP) {
    return (((size_t)(row >> 5) * KP + (g >> 1)) * 2 + (g & 1)) * 32 + (row & 31);
}

// FP6 e2m3 encodings of 4-bit codes
__host__ __device__ __forceinline__ unsigned f6_enc_s4(int q) { return q < 0 ? (32u | (unsigned)(-q)) : (unsigned)q; }
__host__ __device__ __forceinline__ int f6_dec(unsigned c, int is_unsigned) {
    return is_unsigned ? (int)(c & 31) : ((c & 32) ? -(int)(c & 15) : (int)(c & 15));
}

// ---- persistent GEMM schedule (shared by the kernel and by svdq_gemm_schedule, which the CPU tests use) --
// A grid of G workgroups walks NT output tiles of KP K-steps each.  Every workgroup first takes F = NT / G
// whole tiles (tile i*G + pos), then the R = NT % G remainder tiles are either handed out whole (gs == 0: the
// first R positions take one each) or split along K ("stream-K", gs >= R): their R*KP K-steps are dealt evenly
// to the positions 0..gs-1, a position's run is cut at tile boundaries into segments and handed out LAST
// SEGMENT FIRST (the head of a tile somebody else owns is published before our own owner duty can wait).
// The segment with kp1 == KP owns its tile: it collects the fp32 partials of the `contributors()` other
// segments and runs the epilogue.
//
// Row runs (init_runs; the GELU_QUANT launch of the 256 x 128 geometry): position `pos` walks ONE run of up to RL
// consecutive column tiles of one row block -- tile ids are row-major here (tile = bm * TN + bn) -- so that the next layer's low-rank
// down projection of its tiles accumulates inside the workgroup and goes to memory once per run instead of once per tile.
struct GemmSegment { int tile, kp0, kp1; long long u0; };
struct GemmSchedule {
    int NT, KP, G, gs, pos, F, R;
    long long RU, su, su_end, su_begin;
    int it_full;
    int run_t, run_end; // row runs: the next tile of this position's run and its end (run_end < 0: not in run mode)
    // run length for a launch of TM x TN tiles on `slots` workgroup slots: as many rounds as the plain schedule needs, longer if the
    // runs would not fit the slots; runs per row = ceil(TN / RL), grid = TM * runs per row
    static __host__ __device__ int run_length(int TM, int TN, int slots) {
        int rl = (TM * TN + slots - 1) / slots;
        if (rl < 1) rl = 1;
        while (rl < TN && TM * ((TN + rl - 1) / rl) > slots) rl++;
        return rl;
    }
    __host__ __device__ void init_runs(int TM, int TN, int KP_, int RL, int pos_) {
        NT = TM * TN; KP = KP_; G = 0; gs = 0; pos = pos_; F = 0; R = 0; RU = su = su_end = su_begin = 0; it_full = 0;
        const int rpr = (TN + RL - 1) / RL, run = pos_;
        if (run >= TM * rpr) { run_t = run_end = 0; return; }
        const int bm = run / rpr, c = run - bm * rpr;
        run_t = bm * TN + c * RL;
        run_end = bm * TN + (c * RL + RL < TN ? c * RL + RL : TN);
    }
    __host__ __device__ void init(int NT_, int KP_, int G_, int gs_, int pos_) {
        NT = NT_; KP = KP_; G = G_; pos = pos_; run_t = 0; run_end = -1;
        F = NT / G; R = NT - F * G;
        gs = (R > 0 && gs_ >= R && gs_ <= G) ? gs_ : 0;
        RU = (long long)R * KP;
        su = su_end = su_begin = 0;
        if (gs && pos < gs) { su = su_begin = ubound(pos); su_end = ubound(pos + 1); }
        it_full = 0;
    }
    __host__ __device__ long long ubound(int q) const { return (long long)q * RU / gs; }             // first K-step of position q
    __host__ __device__ int pos_of(long long u) const { return (int)(((u + 1) * gs - 1) / RU); }     // position holding K-step u
    __host__ __device__ bool next(GemmSegment &sg) {
        if (run_end >= 0) {
            if (run_t < run_end) { sg = GemmSegment{run_t, 0, KP, 0}; run_t++; return true; }
            return false;
        }
        if (it_full < F) { sg = GemmSegment{it_full * G + pos, 0, KP, 0}; it_full++; return true; }
        if (!gs) {
            if (it_full == F && pos < R) { sg = GemmSegment{F * G + pos, 0, KP, 0}; it_full++; return true; }
            return false;
        }
        if (su < su_end) {
            const long long t = (su_end - 1) / KP;
            const long long u0 = su > t * KP ? su : t * KP;
            sg = GemmSegment{F * G + (int)t, (int)(u0 - t * KP), (int)(su_end - t * KP), u0};
            su_end = u0;
            return true;
        }
        return false;
    }
    // partial-tile slot of a non-owner segment: 2 slots per position (a run spans at most two tiles)
    __host__ __device__ int slot(const GemmSegment &sg) const { return pos * 2 + (sg.u0 > su_begin ? 1 : 0); }
    // owner side: positions [first_contributor, pos) hold the earlier K-steps of sg.tile; slot of contributor q
    __host__ __device__ int first_contributor(const GemmSegment &sg) const { return pos_of((long long)(sg.tile - F * G) * KP); }
    __host__ __device__ int contributor_slot(const GemmSegment &sg, int q) const {
        return q * 2 + (ubound(q) < (long long)(sg.tile - F * G) * KP ? 1 : 0);
    }
};

// ---- host-side error plumbing --------------------------------------------------------------
void set_error(const char *fmt, ...);
int hip_check(hipError_t e, const char *what);
// launch profiler hooks (repack.hip): returns a record index or -1 when profiling is off
int prof_begin(int cls, double work, hipStream_t st);
void prof_end(int rec, hipStream_t st);

} // namespace svdq
