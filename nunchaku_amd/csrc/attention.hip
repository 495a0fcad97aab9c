// svdq_attention: bf16/fp16 flash attention over the packed QKV the fused QKV GEMM epilogue emits
// (role of the reference's EpiloguePackQKV + attention_fp16: epilogues.cuh:427-550, attention.cu:11-94,
//  nunchaku/ops/fused.py:82-178 with output=(q,k,v)).
//
// Layouts (no transposes anywhere in the model):
//   Q, K : token-major rows of the QKV GEMM output, element (l, h, d) at base + l*ld + h*128 + d
//   V^T  : channel-major, element (h, d, l) at base + (h*128 + d)*ldvt + l   (written by the RMSNORM_ROPE
//          epilogue of svdq_gemm_w4a4 when out_vt is given) -- the PV MFMA needs 8 consecutive KEYS per lane
//   O    : token-major [L, H*128], directly the input of the output projection's quantiser -- or no 16-bit output at
//          all: with svdq_attention_args.qact the epilogue emits that quantiser's result itself (a wave's 32 rows x 128
//          channels of one head are one F6 chunk in the register layout it holds)
//
// Two workgroup geometries (svdq_attention_args.geometry).  Geometry 1, described here: workgroup = 8 waves = 256 query rows of
// one head (4 waves / 128 rows when L % 256 != 0), wave = 32 query rows.  Geometry 2 (attention_kernel64 below: 4 waves x 64 rows, one wave per
// SIMD, the tile loop in generated assembly) shares the layouts, the task definition, the slabs and the epilogue;
// KV tiles of 64 keys, double-buffered in LDS with XOR-swizzled 16-byte pieces (conflict-free ds_read_b128);
// the score MFMA is issued swapped (S^T = K Q^T) so a lane holds 32 of the 64 scores of ONE query row: the
// online softmax is lane-local plus one lane^32 exchange; P is packed to 16-bit with v_cvt_pk + one
// v_permlane32_swap per dword into exactly the B-operand fragments of the PV MFMA (O^T = V^T P^T), whose
// accumulator again has the query along the lanes -- the running rescale is a per-lane scalar.
//
// Schedule.  A task = (head, 256 query rows) x all L/64 KV tiles.  FLUX.1 at 1024^2: 24 heads x 18 = 432 tasks on 256 CUs =
// 1.69 rounds of workgroups, the second round 69 % full.  With a workspace the launch is PERSISTENT instead: one workgroup
// per CU takes its whole tasks first (432 / 256 = 1 each), then the remaining 176 tasks are split along the keys: their
// (task, KV tile) space is dealt evenly (49.5 tiles each, cut at even tile indices), a workgroup's run is cut at task
// boundaries into segments.  The workgroup holding a task's FIRST tiles owns it: the others publish their un-normalised
// (O, m, l) -- a lane-for-lane image of the registers, fp32 -- through the workspace, the owner folds them in (ascending
// workgroup order: deterministic) and runs the epilogue.  Logical workgroup g = (blockIdx % 8) * G/8 + blockIdx / 8: the
// workgroups of one XCD take neighbouring tasks, i.e. the same few heads, and walk their keys together.
#include "svdq_common.h"
#include "lowrank_split.h"
#include <type_traits>

// tools/ablate/build_attn.py builds timing variants of geometry 2's tile loop (the generator under other options) by redefining these
#ifndef SVDQ_ATTN_LOOP_INC_BF16
#define SVDQ_ATTN_LOOP_INC_BF16 "attention_loop64_bf16.inc"
#define SVDQ_ATTN_LOOP_INC_FP16 "attention_loop64_fp16.inc"
#endif
// clock / phase stamps of the tools-built probe library (tools/ablate/attn_probe_hooks.inc); nothing in the product build
#ifdef SVDQ_PROBE
#include "attn_probe_hooks.inc"
#else
#define SVDQ_ATTN_PROBE_PARAMS
#define SVDQ_ATTN_PROBE_FILL(p)
#define SVDQ_ATTN_PROBE_BEGIN()
#define SVDQ_ATTN_PROBE_LOOP_BEGIN()
#define SVDQ_ATTN_PROBE_LOOP_END(n)
#define SVDQ_ATTN_PROBE_END()
#endif

namespace svdq {

constexpr int ATT_D = 128;     // head dimension (FLUX; the reference's attention kernel is also fixed to 128)
constexpr int ATT_KB = 64;     // keys per tile
constexpr int ATT_TILE = ATT_KB * ATT_D * 2; // bytes of one K tile (= one V^T tile)

struct AttnParams {
    const uint16_t *q, *k, *vt;
    uint16_t *out;
    long long q_hs, k_hs, vt_hs, o_hs;
    int L, H, ldq, ldk, ldvt, ldo;
    float scale_log2e; // softmax scale * log2(e)
    v4i *zero_ptr;     // optional scratch cleared by this launch (svdq_attention_args.zero_ptr)
    long long zero_vec; // its size in 16-byte units
    // fused quantiser of the following output projection (svdq_attention_args.qact ...)
    uint8_t *qact;
    uint16_t *qscales;
    void *qlora_act;
    int qlora_q32;     // qlora_act holds Q31.32 fixed point (order-independent integer atomics over the heads)
    int *status;       // optional host-visible status word (svdq_attention_args.status)
    int kv_len0, kv_start1, kv_end1; // key mask: keys [0, kv_len0) and [kv_start1, kv_end1) are real, the rest padding (kv_len0 == 0: no mask)
    int mask_j0, mask_j1;            // geometry 2 with a key mask: the main segment [mask_j0, mask_j1) of fully real tiles (even count) the assembly loop runs
    const uint16_t *qsmooth, *qlora_down, *qsmooth2, *qlora_down2;
    int qR, qsplit_rows;
    void *qa16;        // split low-rank down (rank 48 .. 160, fp32): the normalised 16-bit output as MFMA operand fragments [L / 32][H * 8 units][64 lanes][8]
                       // (lowrank_split.h); the kernel then runs no low-rank pass -- lowrank_down_split_kernel contracts the image behind it
    // persistent schedule (svdq_attention_args.workspace): arrival counters + error word, then one slab per workgroup
    int *ws_flags;
    float *ws_slabs;
    SVDQ_ATTN_PROBE_PARAMS
};

constexpr int ATT_SLAB_O = 8 * 16 * 64 * 4;            // floats: [wave][j][lane][4] image of o
constexpr int ATT_SLAB_FLOATS = ATT_SLAB_O + 8 * 64 * 2; // + [wave][lane]{m, l}
constexpr int ATT_WS_HEADER = 4096;                    // bytes: 1023 arrival counters + error word
constexpr int ATT_ERR_WORD = 1023;
constexpr int ATT_SPIN_LIMIT = 1 << 22;                // x s_sleep(8) ~ 1 s: a broken workspace contract, not a slow peer
// rescale the running output only when some row's maximum grew by more than this (in log2 units, i.e. after the
// scale*log2(e) factor): until then P = exp2(s - m_stale) <= 2^8, exact in fp32 and with the same RELATIVE rounding
// in the 16-bit P fragments; later tiles almost never rescale (64 multiplies + exp per wave-tile saved)
constexpr float ATT_DEFER_LOG2 = 8.0f;

// The persistent schedule's arithmetic, shared by the kernel and its host replay (svdq_attention_schedule).
// Workgroup g first takes F = tasks / G WHOLE tasks (task f*G + g: the workgroups of an XCD, numbered contiguously, walk the keys
// of the same few heads in lockstep -- K / V^T tiles are fetched once per XCD and hit in its L2 for the others, as in a plain
// grid).  The R = tasks - F*G remainder tasks are split along the keys: linear position = local task * ntiles + KV tile,
// workgroup i < Gs runs [bound(i), bound(i + 1)), cut at even tiles.
struct AttnSchedule {
    int ntiles, G, F, Gs;
    long long half_rem; // R * ntiles / 2
    __host__ __device__ void init(int tasks, int ntiles_, int G_) {
        ntiles = ntiles_; G = G_;
        F = tasks / G;
        half_rem = (long long)(tasks - F * G) * ntiles / 2;
        Gs = half_rem < G ? (int)half_rem : G; // workgroups sharing the remainder (every one of them gets at least two tiles)
    }
    __host__ __device__ int bound(int i) const { return Gs ? (int)(half_rem * i / Gs) * 2 : 0; }
    // the workgroup < g whose run holds the FIRST tile of local task `tl` (g's own first run segment starts inside that task)
    __host__ __device__ int owner_of(int g, int tl) const {
        int o = g - 1;
        while (bound(o) > tl * ntiles) o--;
        return o;
    }
    // the workgroup > g whose run holds the LAST tile of local task `tl` (g's last segment starts the task but does not finish it)
    __host__ __device__ int last_contributor(int g, int tl) const {
        int last = g + 1;
        while (bound(last + 1) < (tl + 1) * ntiles) last++;
        return last;
    }
};

typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// two floats -> one dword of two RNE-rounded 16-bit values (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32)
template <int DT> __device__ __forceinline__ unsigned pack2(float a, float b) {
    if constexpr (DT == SVDQ_BF16) return __builtin_bit_cast(unsigned, __builtin_convertvector((v2f){a, b}, bf16x2));
    else return __builtin_bit_cast(unsigned, __builtin_convertvector((v2f){a, b}, f16x2));
}
// (a * s, b * s) -> one dword of two 16-bit values, rounded as finish_rows' fused quantiser rounds the same products.  fp16: the backend folds a SCALAR
// `(_Float16)(x * s)` into v_fma_mixlo/hi_f16 -- ONE rounding of the exact product -- while the packed conversion of two fp32 products rounds twice
// (fp32, then fp16) and lands one fp16 step away on ~1e-4 of the values.  Until round 6 the 16-bit output path took the packed form and the fused
// quantiser the scalar one (seen in the ISA listing: v_mul_f32 + v_cvt_pk_f16_f32 against v_fma_mixlo_f16): the quantiser then did not quantise exactly
// the values a separate launch stores.  Both take the scalar shape now; bf16 has no mixed-precision FMA and keeps the packed conversion.
template <int DT> __device__ __forceinline__ unsigned pack2_scaled(float a, float b, float s) {
    if constexpr (DT == SVDQ_BF16) return pack2<DT>(a * s, b * s);
    else {
        // (written as two scalar conversions the optimiser re-vectorises them into v_mul_f32 x 2 + v_cvt_pk_f16_f32 -- the double rounding again: the two
        //  mixed-precision FMAs are spelled out)
        unsigned r;
        asm("v_fma_mixlo_f16 %0, %1, %3, 0\n\tv_fma_mixhi_f16 %0, %2, %3, 0" : "=&v"(r) : "v"(a), "v"(b), "v"(s));
        return r;
    }
}

// l += the two 16-bit values of `packed` (v_dot2c_f32_bf16 / v_dot2c_f32_f16 against (1, 1)): the row sum is taken over the
// ROUNDED probabilities, the same numbers the PV MFMA multiplies -- their rounding error cancels in O / l (a row dominated
// by one key returns that key's value exactly, whatever reference point the exponentials use)
template <int DT> __device__ __forceinline__ float sum2(unsigned packed, float l) {
    if constexpr (DT == SVDQ_BF16) return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, packed), __builtin_bit_cast(bf16x2, 0x3f803f80u), l, false);
    else return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, packed), __builtin_bit_cast(f16x2, 0x3c003c00u), l, false);
}

// Normalise one 32-row tile of a wave and store it: lane owns query row q0 + lr and channels 32*dt + 8c + 4h + e of o[dt].
// Either the 16-bit output, or (p.qact) the output projection's quantised activation, or both.
template <int DT>
__device__ __forceinline__ void finish_rows(const AttnParams &p, const v16f (&o)[4], float l_run, int q0, int head, int lane) {
    using V8 = typename Half<DT>::V8;
    const int lr = lane & 31, h = lane >> 5;
    const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32));
    if (p.qact) {
        // ---- fused quantiser of the output projection (quantize.hip, same arithmetic): this wave's 32 rows x 128
        //      channels = F6 chunk (row tile q0/32, kp = head); lane (lr, h) holds exactly its lane record's channels
        using T = typename Half<DT>::T;
        const bool s2 = p.qsplit_rows > 0 && q0 >= p.qsplit_rows; // wave-uniform
        const T *smooth = (const T *)(s2 ? p.qsmooth2 : p.qsmooth) + head * ATT_D;
        const int K = p.H * ATT_D, KP = p.H;
        // lora_act[q][rank] += sum_d o16[q][d] * down[d][rank] over this head's 128 channels, 32 ranks per pass (rank 128 checkpoints and runtime LoRAs
        // take 2 .. 8 passes; the body of a pass is the rank <= 32 code of rounds 2-4, unchanged: it sits at the register limit of the 4 x 64 kernels)
        auto lowrank_pass = [&](int rank0) {
            const T *ld = (const T *)(s2 ? p.qlora_down2 : p.qlora_down) + head * ATT_D; // rank-major [R][K]
            v16f dl;
#pragma unroll
            for (int i = 0; i < 16; i++) dl[i] = 0.f;
            const bool live = rank0 + lr < p.qR;
#pragma unroll
            for (int dt = 0; dt < 4; dt++)
#pragma unroll
                for (int qq = 0; qq < 2; qq++) {
                    V8 wv, gv;
                    if (live) {
                        const T *src = ld + (size_t)(rank0 + lr) * K + dt * 32 + qq * 16 + h * 4;
                        const u16x4 w0 = *reinterpret_cast<const u16x4 *>(src);
                        const u16x4 w1 = *reinterpret_cast<const u16x4 *>(src + 8);
#pragma unroll
                        for (int j = 0; j < 4; j++) { wv[j] = hfrom<T>(w0[j]); wv[4 + j] = hfrom<T>(w1[j]); }
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; j++) wv[j] = (T)0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 8; j++) gv[j] = f2h<T>(o[dt][qq * 8 + j] * inv);
                    dl = Half<DT>::mfma32(gv, wv, dl);
                }
            if (live) { // C layout: column (rank) = lane & 31, rows (i & 3) + 8 (i >> 2) + 4 h
                const size_t at = (size_t)(q0 + h * 4) * p.qR + rank0 + lr;
                const int mode = 1 | (p.qlora_q32 ? 2 : 0); // the H heads add to the same element
#pragma unroll
                for (int i = 0; i < 16; i++) lora_act_add(p.qlora_act, at + (size_t)((i & 3) + 8 * (i >> 2)) * p.qR, dl[i], mode);
            }
        };
        if (p.qa16) {
            // rank 48 .. 160: the 16-bit rows as the A-operand fragments of a contraction that runs behind this kernel -- the lane's 8 channels
            // {16 qq + 8 (j >> 2) + 4 h + (j & 3)} of unit (head, dt, qq) are k-slots 8 h + j as they stand: 8 coalesced 16-byte stores per row tile
            // instead of 2 .. 5 passes of 8 MFMAs + 16 atomic instructions that the H heads aim at the same elements
            V8 *a16 = (V8 *)p.qa16 + ((size_t)(q0 >> 5) * (size_t)(p.H * 8) + (size_t)head * 8) * 64 + lane;
#pragma unroll
            for (int dt = 0; dt < 4; dt++)
#pragma unroll
                for (int qq = 0; qq < 2; qq++) {
                    V8 gv;
#pragma unroll
                    for (int j = 0; j < 8; j++) gv[j] = f2h<T>(o[dt][qq * 8 + j] * inv);
                    a16[(dt * 2 + qq) * 64] = gv;
                }
        } else {
        if (p.qR > 0) lowrank_pass(0);                                     // (straight-line, as in rounds 2-4: the rank-32 step runs exactly this)
        for (int rank0 = 32; rank0 < p.qR; rank0 += 32) lowrank_pass(rank0); // the slabs beyond rank 32
        }
        uint32_t rec[12];
        T sc16[2];
#pragma unroll
        for (int g = 0; g < 2; g++) {
            float xh[32];
            float amax = 0.f;
#pragma unroll
            for (int t = 0; t < 2; t++)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int dt = 2 * g + t;
                    const u16x4 sv = *reinterpret_cast<const u16x4 *>(smooth + dt * 32 + c * 8 + h * 4);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float o16 = round16<T>(o[dt][c * 4 + e] * inv);
                        const float sm = h2f(hfrom<T>(sv[e]));
                        const float v = smooth_div16<T>(o16, __builtin_amdgcn_rcpf(sm));
                        xh[16 * t + c * 4 + e] = v;
                        amax = fmaxf(amax, fabsf(v));
                    }
                }
            amax = fmaxf(amax, __shfl_xor(amax, 32));
            const float scale = amax * (1.0f / 7.0f);
            const float rscale = scale == 0.f ? 0.f : 1.0f / scale;
            // the stored scale is round16 of the fp32 PRODUCT (quantize.hip, the GELU_QUANT epilogue, the oracle: two roundings).  Written plainly, the fp16 build
            // folded `(T)(amax * (1/7))` here -- and only here -- into one v_fma_mixlo_f16 of the exact product: a 16-bit step away on ~3e-4 of the scales, codes
            // identical (round 6: this was the unexplained H = 6 observation of round 5; found in the ISA listing).  The empty asm statement keeps the product opaque.
            float scale_rounded = scale;
            asm volatile("" : "+v"(scale_rounded));
            sc16[g] = f2h<T>(scale_rounded);
            v16f ev, od;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                ev[i] = xh[2 * i] * rscale;
                od[i] = xh[2 * i + 1] * rscale;
            }
            const v6i pk = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(ev, od, 8.0f);
#pragma unroll
            for (int i = 0; i < 6; i++) rec[6 * g + i] = (uint32_t)pk[i];
        }
        const int rt = q0 >> 5;
        uint8_t *dst = p.qact + ((size_t)rt * KP + head) * F6_CHUNK + (size_t)lane * 16;
        *reinterpret_cast<uint4 *>(dst) = make_uint4(rec[0], rec[1], rec[2], rec[3]);
        *reinterpret_cast<uint4 *>(dst + F6_PLANE) = make_uint4(rec[4], rec[5], rec[6], rec[7]);
        *reinterpret_cast<uint4 *>(dst + 2 * F6_PLANE) = make_uint4(rec[8], rec[9], rec[10], rec[11]);
        p.qscales[(((size_t)rt * KP + head) * 2 + h) * 32 + lr] = hbits(sc16[h]);
    }
    if (p.out) {
    uint16_t *orow = p.out + (size_t)(q0 + lr) * p.ldo + (size_t)head * p.o_hs + 8 * h;
#pragma unroll
    for (int dt = 0; dt < 4; dt++)
#pragma unroll
        for (int j2 = 0; j2 < 2; j2++) {
            unsigned x[2], y[2];
#pragma unroll
            for (int d2 = 0; d2 < 2; d2++) {
                x[d2] = pack2_scaled<DT>(o[dt][8 * j2 + 2 * d2], o[dt][8 * j2 + 2 * d2 + 1], inv);
                y[d2] = pack2_scaled<DT>(o[dt][8 * j2 + 4 + 2 * d2], o[dt][8 * j2 + 4 + 2 * d2 + 1], inv);
                auto sw = __builtin_amdgcn_permlane32_swap(x[d2], y[d2], false, false);
                x[d2] = sw[0];
                y[d2] = sw[1];
            }
            *reinterpret_cast<v4i *>(orow + 32 * dt + 16 * j2) = v4i{(int)x[0], (int)x[1], (int)y[0], (int)y[1]};
        }
    } // p.out
}

// NW waves = NW * 32 query rows of one head per workgroup.
template <int DT, int NW, bool PERSIST>
__global__ __launch_bounds__(NW * 64, 8 / NW) void attention_kernel(const AttnParams p) {
    static_assert(!PERSIST || NW == 8, "the persistent schedule is built for the 8-wave workgroup");
    using V8 = typename Half<DT>::V8;
    constexpr int NT = NW * 64;          // threads
    constexpr int PIECES = 1024 / NT;    // 16-byte pieces of each of K and V^T a thread stages per tile
    __shared__ __attribute__((aligned(16))) uint8_t lds[4 * ATT_TILE]; // [buf][K | V^T]
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    typedef __attribute__((address_space(3))) v4i lds_v4i;
    lds_u8 *const L8 = (lds_u8 *)lds;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lr = lane & 31, h = lane >> 5;
    for (long long i = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * NT + tid; i < p.zero_vec; i += (long long)gridDim.x * gridDim.y * NT)
        p.zero_ptr[i] = v4i{0, 0, 0, 0}; // side job: clear the next quantiser's low-rank accumulators

    // ---- this workgroup's run of the linear (task, KV tile) space ---------------------------------------------
    const int ntiles = p.L / ATT_KB; // even: L is a multiple of 128
    const int QT = p.L / (NW * 32);  // tasks per head
    const int G = PERSIST ? (int)gridDim.x : 1;
    const int g = !PERSIST ? 0 : (G % 8 == 0 ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x);
    AttnSchedule sched;
    sched.init(p.H * QT, ntiles, G);
    int whole_left = PERSIST ? sched.F : 1;                       // whole tasks still to do (plain grid: exactly one)
    int pos = PERSIST && g < sched.Gs ? sched.bound(g) : 0;        // remainder run [pos, run_hi), runs start at even tiles
    const int run_hi = PERSIST && g < sched.Gs ? sched.bound(g + 1) : 0;

    // ---- tile staging: 1024 16-byte pieces per matrix per tile, PIECES per thread; XOR-swizzled so that the
    //      16 lanes a ds_read_b128 serves per LDS cycle hit 16 different 16-byte columns.  All per-thread
    //      offsets are loop invariants (32-bit, relative to a wave-uniform tile base / the LDS buffer) ----------
    unsigned koff[PIECES], voff[PIECES], kst[PIECES], vst[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; i++) {
        const int idx = tid + NT * i;
        const int kr = idx >> 4, kc = idx & 15; // K: row (key), 16-byte column
        koff[i] = (unsigned)kr * (unsigned)p.ldk * 2u + kc * 16;
        kst[i] = kr * 256 + ((kc ^ (kr & 15)) << 4);
        const int vd = idx >> 3, vc = idx & 7;  // V^T: row (channel), 16-byte column (8 keys)
        voff[i] = (unsigned)vd * (unsigned)p.ldvt * 2u + vc * 16;
        vst[i] = ATT_TILE + vd * 128 + ((vc ^ ((vd >> 1) & 7)) << 4);
    }
    // fragment read offsets: K rows 32*kt + lr (+8192 per kt), V^T rows 32*dt + lr (+4096 per dt)
    unsigned ka[8], va[4];
#pragma unroll
    for (int ds = 0; ds < 8; ds++) ka[ds] = lr * 256 + (((2 * ds + h) ^ (lr & 15)) << 4);
#pragma unroll
    for (int ks = 0; ks < 4; ks++) va[ks] = ATT_TILE + lr * 128 + (((2 * ks + h) ^ ((lr >> 1) & 7)) << 4);

    const uint8_t *kbase = nullptr, *vtbase = nullptr; // K / V^T of the current segment's head
    v4i kreg[PIECES], vreg[PIECES];
    auto load_tile = [&](int kv0) {
        const uint8_t *kt = kbase + (size_t)kv0 * p.ldk * 2; // wave-uniform
        const uint8_t *vt = vtbase + (size_t)kv0 * 2;
#pragma unroll
        for (int i = 0; i < PIECES; i++) {
            kreg[i] = *reinterpret_cast<const v4i *>(kt + koff[i]);
            vreg[i] = *reinterpret_cast<const v4i *>(vt + voff[i]);
        }
    };
    auto store_tile = [&](int bufoff) {
#pragma unroll
        for (int i = 0; i < PIECES; i++) {
            *(lds_v4i *)(L8 + (kst[i] + bufoff)) = kreg[i];
            *(lds_v4i *)(L8 + (vst[i] + bufoff)) = vreg[i];
        }
    };

    V8 qf[8];
    v16f o[4];
    float m_run = -INFINITY; // running row max, raw score units (shared by the two lanes of a query row)
    v2f l2 = {0.f, 0.f};     // this lane's share of the row sum, as two partial sums (two independent v_dot2c chains)
    const float c = p.scale_log2e;
    int j_end = 0;           // end of the current segment's KV tiles

    auto step = [&](auto bufc, int j) {
        constexpr int BUF = decltype(bufc)::value;
        constexpr int BO = BUF * 2 * ATT_TILE;

        // ---- S^T[k][q] = sum_d K[k][d] Q[q][d]: two 32-key tiles x 8 d-steps ------------------------------
        v16f s[2];
#pragma unroll
        for (int kt = 0; kt < 2; kt++) {
#pragma unroll
            for (int r = 0; r < 16; r++) s[kt][r] = 0.f;
#pragma unroll
            for (int ds = 0; ds < 8; ds++) {
                const v4i kw = *(const lds_v4i *)(L8 + (ka[ds] + (BO + kt * 8192)));
                s[kt] = Half<DT>::mfma32(__builtin_bit_cast(V8, kw), qf[ds], s[kt]);
            }
        }

        if (j + 1 < j_end) load_tile((j + 1) * ATT_KB); // in flight under the softmax and the PV MFMAs

        // ---- key-padding mask (svdq_attention_args.kv_len0 ...): padded keys score -inf, i.e. probability 0.  Only tiles
        //      that contain padding pay for it (workgroup-uniform branch); role of the reference's padded-row handling in
        //      EpiloguePackQKV / attention.cuh (K rows beyond the token count never contribute)
        if (p.kv_len0 > 0) {
            const int k0 = j * ATT_KB;
            const bool all_real = k0 + ATT_KB <= p.kv_len0 || (k0 >= p.kv_start1 && k0 + ATT_KB <= p.kv_end1);
            if (!all_real) {
#pragma unroll
                for (int kt = 0; kt < 2; kt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int key = k0 + 32 * kt + 8 * (r >> 2) + 4 * h + (r & 3); // C layout of S^T: row (key) = 8 (r / 4) + 4 h + r % 4
                        const bool real = key < p.kv_len0 || (key >= p.kv_start1 && key < p.kv_end1);
                        s[kt][r] = real ? s[kt][r] : -INFINITY;
                    }
            }
        }

        // ---- online softmax: lane holds 32 of the 64 scores of query row lr (the partner lane the others) --
        float mloc = fmaxf(s[0][0], s[1][0]);
#pragma unroll
        for (int r = 1; r < 16; r += 2) { // v_max3_f32 chains
            mloc = fmaxf(fmaxf(mloc, s[0][r]), r + 1 < 16 ? s[0][r + 1] : s[0][r]);
            mloc = fmaxf(fmaxf(mloc, s[1][r]), r + 1 < 16 ? s[1][r + 1] : s[1][r]);
        }
        {   // the partner lane (lane ^ 32) holds the other 32 scores of the row: one v_permlane32_swap, no LDS round trip
            // (scalar copies first: this clang's __builtin_bit_cast applied to a vector ELEMENT reads element 0 -- max(sw[0], sw[1])
            //  became sw[0]: half a row's maximum, still a valid reference point in bf16, an overflow to inf in fp16)
            const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, mloc), __builtin_bit_cast(unsigned, mloc), false, false);
            const unsigned mine = sw[0], other = sw[1];
            mloc = fmaxf(__builtin_bit_cast(float, mine), __builtin_bit_cast(float, other));
        }
        // rescale only when some row of this wave outgrew its running maximum by more than ATT_DEFER_LOG2 (m_run = -inf on
        // the first tile: always).  A stale maximum is still a valid reference point of the online softmax: O and l carry
        // the same factor and it cancels in the final division.
        if (__builtin_amdgcn_ballot_w64((mloc - m_run) * c > ATT_DEFER_LOG2) != 0) {
            const float m_new = fmaxf(m_run, mloc);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c); // 0 on the first tile (m_run = -inf)
            m_run = m_new;
            l2 = l2 * alpha;
#pragma unroll
            for (int dt = 0; dt < 4; dt++) o[dt] = o[dt] * alpha;
        }
        const float mc = m_run == -INFINITY ? 0.f : m_run * c; // (a segment that starts inside the padding: every score so far is -inf)
        // scalar FMAs on purpose: beside running MFMAs a packed fp32 instruction (v_pk_fma_f32) costs ~22 cycles more than the two
        // v_fma_f32 it replaces (MI355X_MICROARCH.md "price of one filler beside MFMAs")
        const float nmc = -mc;
#pragma unroll
        for (int kt = 0; kt < 2; kt++)
#pragma unroll
            for (int r = 0; r < 16; r++) s[kt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], c, nmc));

        // ---- P -> 16-bit B-operand fragments of the PV MFMA: lane (q = lr, keys 16*ks + 8h .. +7) ------------
        // regs 8*(ks&1) + {0..3} hold keys 4h + {0..3}, regs + {4..7} keys 8 + 4h + {0..3} of that 16-key step
        V8 pf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const int kt = ks >> 1, r0 = 8 * (ks & 1);
            unsigned x[2], y[2];
#pragma unroll
            for (int d2 = 0; d2 < 2; d2++) {
                x[d2] = pack2<DT>(s[kt][r0 + 2 * d2], s[kt][r0 + 2 * d2 + 1]);
                y[d2] = pack2<DT>(s[kt][r0 + 4 + 2 * d2], s[kt][r0 + 4 + 2 * d2 + 1]);
                l2[0] = sum2<DT>(x[d2], l2[0]); // this lane's own 4 probabilities, before the exchange
                l2[1] = sum2<DT>(y[d2], l2[1]);
                auto sw = __builtin_amdgcn_permlane32_swap(x[d2], y[d2], false, false);
                x[d2] = sw[0];
                y[d2] = sw[1];
            }
            pf[ks] = __builtin_bit_cast(V8, v4i{(int)x[0], (int)x[1], (int)y[0], (int)y[1]});
        }

        // ---- O^T[d][q] += sum_k V^T[d][k] P^T[k][q]: 4 channel tiles x 4 key steps ----------------------------
#pragma unroll
        for (int dt = 0; dt < 4; dt++) {
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const v4i vw = *(const lds_v4i *)(L8 + (va[ks] + (BO + dt * 4096)));
                o[dt] = Half<DT>::mfma32(__builtin_bit_cast(V8, vw), pf[ks], o[dt]);
            }
        }

        if (j + 1 < j_end) store_tile((BUF ^ 1) * 2 * ATT_TILE); // that buffer was last read in iteration j-1
        __syncthreads();
    };
    while (true) {
        // ---- one segment: KV tiles [j0, j1) of one task.  tl: the task's index among the remainder tasks (-1: a whole task)
        int task, j0, j1, tl = -1;
        if (whole_left > 0) {
            if constexpr (PERSIST) task = (sched.F - whole_left) * G + g;
            else {
                // plain grid: workgroups are dealt round-robin to the 8 XCDs in launch order; give XCD x the tasks
                // [x * N/8, (x+1) * N/8) in that order, so the workgroups resident on it share heads (K / V^T tiles in its L2)
                const int b = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x, N = (int)(gridDim.x * gridDim.y);
                task = N % 8 == 0 ? (b % 8) * (N / 8) + b / 8 : b;
            }
            j0 = 0; j1 = ntiles;
            whole_left--;
        } else if (pos < run_hi) {
            tl = pos / ntiles;
            j0 = pos - tl * ntiles;
            j1 = min(ntiles, j0 + (run_hi - pos));
            pos += j1 - j0;
            task = sched.F * G + tl;
        } else break;
        const int head = task / QT;
        const int q0 = (task - head * QT) * (NW * 32) + wave * 32;
        kbase = (const uint8_t *)(p.k + (size_t)head * p.k_hs);
        vtbase = (const uint8_t *)(p.vt + (size_t)head * p.vt_hs);
        {   // Q fragments: B operand of S^T = K Q^T, lane (q = lr, d = 16*ds + 8h .. +7)
            const uint16_t *qrow = p.q + (size_t)(q0 + lr) * p.ldq + (size_t)head * p.q_hs + 8 * h;
#pragma unroll
            for (int ds = 0; ds < 8; ds++) qf[ds] = *reinterpret_cast<const V8 *>(qrow + 16 * ds);
        }
#pragma unroll
        for (int dt = 0; dt < 4; dt++)
#pragma unroll
            for (int r = 0; r < 16; r++) o[dt][r] = 0.f;
        m_run = -INFINITY;
        l2 = v2f{0.f, 0.f};
        j_end = j1;
        load_tile(j0 * ATT_KB);
        store_tile(0); // every wave passed the barrier that ends the previous segment's last step: both buffers are free
        __syncthreads();
        for (int j = j0; j < j1; j += 2) { // j1 - j0 is even
            step(std::integral_constant<int, 0>{}, j);
            step(std::integral_constant<int, 1>{}, j + 1);
        }
        float l_run = l2[0] + l2[1];

        if constexpr (PERSIST) {
            typedef __attribute__((address_space(1))) int gint; // explicit global address space: no flat aperture checks
            typedef __attribute__((address_space(1))) v4f gv4f;
            typedef __attribute__((address_space(1))) v2f gv2f;
            gint *flags = (gint *)p.ws_flags;
            if (j0 > 0) {
                // ---- not the owner: publish the raw state (16 coalesced 1 KiB stores per wave + {m, l}), make it visible
                //      at agent scope, bump the owner's arrival counter.  At most one such segment per workgroup (its first).
                float *slab = p.ws_slabs + (size_t)g * ATT_SLAB_FLOATS;
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const v4f v = {o[j >> 2][(j & 3) * 4 + 0], o[j >> 2][(j & 3) * 4 + 1], o[j >> 2][(j & 3) * 4 + 2], o[j >> 2][(j & 3) * 4 + 3]};
                    *(gv4f *)(slab + ((size_t)(wave * 16 + j) * 64 + lane) * 4) = v;
                }
                *(gv2f *)(slab + ATT_SLAB_O + (size_t)(wave * 64 + lane) * 2) = v2f{m_run, l_run};
                const int owner = sched.owner_of(g, tl);
                __syncthreads();
                if (tid == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_fetch_add(flags + owner, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                continue;
            }
            if (j1 < ntiles) {
                // ---- owner of a split task: fold in the segments of the workgroups g+1 .. last (they hold the rest of
                //      the task: the last one at the START of its run, any in between as their WHOLE run)
                const int last = sched.last_contributor(g, tl);
                if (tid == 0) {
                    // Bounded wait (~1 s): a missing arrival can only come from a broken contract (the workspace shared by
                    // launches in flight on two streams, or not zero-filled).  Then give up instead of hanging the GPU:
                    // raise the sticky error word svdq_attention_workspace_status() reports; this task's rows are garbage.
                    int spins = 0;
                    while (__hip_atomic_load(flags + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < last - g && ++spins < ATT_SPIN_LIMIT)
                        __builtin_amdgcn_s_sleep(8);
                    if (spins >= ATT_SPIN_LIMIT) {
                        __hip_atomic_store(flags + ATT_ERR_WORD, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (p.status) __hip_atomic_store(p.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    __hip_atomic_store(flags + g, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next launch
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                for (int q = g + 1; q <= last; q++) {
                    const float *slab = p.ws_slabs + (size_t)q * ATT_SLAB_FLOATS;
                    const v2f ml = __builtin_nontemporal_load((const gv2f *)(slab + ATT_SLAB_O + (size_t)(wave * 64 + lane) * 2));
                    const float m_new = fmaxf(m_run, ml[0]);
                    const float fa = __builtin_amdgcn_exp2f((m_run - m_new) * c), fb = __builtin_amdgcn_exp2f((ml[0] - m_new) * c);
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        const v4f v = __builtin_nontemporal_load((const gv4f *)(slab + ((size_t)(wave * 16 + j) * 64 + lane) * 4));
#pragma unroll
                        for (int e = 0; e < 4; e++) o[j >> 2][(j & 3) * 4 + e] = o[j >> 2][(j & 3) * 4 + e] * fa + v[e] * fb;
                    }
                    l_run = l_run * fa + ml[1] * fb;
                    m_run = m_new;
                }
            }
        }

    finish_rows<DT>(p, o, l_run, q0, head, lane);
    } // segments
}

// ---------------------------------------------------------------------------------------------------------------------
// Geometry 2: 4 waves x 64 query rows, ONE wave per SIMD with the whole 512-register file (O: 128 registers, two score
// sets: 128, Q: 64).  Against the 8 x 32 geometry above a K / V^T fragment read from LDS feeds two MFMAs instead of one,
// and the instruction stream itself overlaps what the two co-resident waves of a SIMD used to overlap by chance:
//   iteration j:   P(j) = exp2(S'(j)) -> 16-bit fragments   beside the 32 MFMAs of  S'(j+1) = K(j+1) Q'^T - mc
//                  row maxima of S'(j+1)                     beside the 32 MFMAs of  O += V^T(j) P(j)
// The tile loop is generated assembly with explicit registers (tools/gen_attn_loop.py -> attention_loop64_*.inc; DESIGN.md 6b:
// what an instruction beside an MFMA costs was measured, the iteration is packed to it slot by slot, and the hazards a compiler
// would cover are covered by construction); the segment's first and last tile, the schedule, the slabs and the epilogue are C++.
// Scores are RELATIVE and in log2 units: Q is multiplied by scale * log2(e) when it is loaded (16-bit rounding of the product: the
// one numerical difference to geometry 1), a score accumulator starts at -mc instead of 0 (mc: the row's reference point, a stale
// maximum; -inf while the row has seen no finite score -- the start value is 0 then), so P = exp2(s') with no further arithmetic.
// K / V^T staging: LDS-DMA into two [K | V^T] buffers, K(j+2) where K(j) was read one iteration earlier, V^T(j+1) where V^T(j-1)
// was.  Same task / persistent-schedule / slab definitions as geometry 1 (task = 256 rows).  Key-padding mask: the MASK instantiation below
// (plain grid); masked launches that do not qualify run geometry 1.
typedef float v32f __attribute__((ext_vector_type(32)));
typedef int v32i __attribute__((ext_vector_type(32)));

// MASK (round 4; plain grid only): a key-padding mask.  The assembly loop runs an EVEN number of FULL tiles and cannot mask a score, so a masked
// launch splits every task: the main segment = the longest run of fully real tiles, cut to an even count, through the loop as always; every other
// tile that holds a real key -- the odd one of that run, the partially padded tiles at the end of a key range, the tiles of the other range -- is an
// "extra" tile in C++ that CONTINUES the state (scores against the current reference point, padded keys at -inf, the reference point moved when a row
// outgrows it); tiles without a real key are never touched.
template <int DT, bool PERSIST, bool MASK = false>
__global__ __launch_bounds__(256, 1) void attention_kernel64(const AttnParams p) {
    static_assert(!(PERSIST && MASK), "the key mask runs on the plain grid");
    using V8 = typename Half<DT>::V8;
    using T = typename Half<DT>::T;
    constexpr int NT = 256, RT = 2;
    __shared__ __attribute__((aligned(16))) uint8_t lds[4 * ATT_TILE]; // [buf][K | V^T]: K(t) and V^T(t) live in buffer (t - j0) & 1
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    typedef __attribute__((address_space(3))) v4i lds_v4i;
    lds_u8 *const L8 = (lds_u8 *)lds;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    SVDQ_ATTN_PROBE_BEGIN();
    const int lr = lane & 31, h = lane >> 5;
    for (long long i = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * NT + tid; i < p.zero_vec; i += (long long)gridDim.x * gridDim.y * NT)
        p.zero_ptr[i] = v4i{0, 0, 0, 0}; // side job: clear the next quantiser's low-rank accumulators

    const int ntiles = p.L / ATT_KB;
    const int QT = p.L / 256; // tasks per head
    const int G = PERSIST ? (int)gridDim.x : 1;
    const int g = !PERSIST ? 0 : (G % 8 == 0 ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x);
    AttnSchedule sched;
    sched.init(p.H * QT, ntiles, G);
    int whole_left = PERSIST ? sched.F : 1;
    int pos = PERSIST && g < sched.Gs ? sched.bound(g) : 0;
    const int run_hi = PERSIST && g < sched.Gs ? sched.bound(g + 1) : 0;

    // staging: LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes = one 1 KiB piece per instruction, written lane-linearly at
    // M0), no registers, no ds_write.  A K piece = 4 key rows, a V^T piece = 8 channel rows; wave w issues pieces 4w .. 4w+3 of each
    // matrix.  The XOR swizzle of the fragment reads (conflict-free ds_read_b128) is applied on the SOURCE side: the lane that writes
    // 16-byte position p of a row fetches the chunk p ^ f(row) of that row.
    v4i kdma, vdma;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int kr = 16 * wave + 4 * i + (lane >> 4), vd = 32 * wave + 8 * i + (lane >> 3);
        kdma[i] = (int)((unsigned)kr * (unsigned)p.ldk * 2u + (((lane & 15) ^ (kr & 15)) << 4));
        vdma[i] = (int)((unsigned)vd * (unsigned)p.ldvt * 2u + (((lane & 7) ^ ((vd >> 1) & 7)) << 4));
    }
    v8i ka;
    v4i va;
#pragma unroll
    for (int ds = 0; ds < 8; ds++) ka[ds] = lr * 256 + (((2 * ds + h) ^ (lr & 15)) << 4);
#pragma unroll
    for (int ks = 0; ks < 4; ks++) va[ks] = ATT_TILE + lr * 128 + (((2 * ks + h) ^ ((lr >> 1) & 7)) << 4);
    const unsigned lds0 = (unsigned)(uintptr_t)L8 + __builtin_amdgcn_readfirstlane(wave) * 4096; // this wave's pieces of buffer 0's K; V^T: + ATT_TILE

    const uint8_t *kbase = nullptr, *vtbase = nullptr;
    // (inline asm: the compiler orders every later LDS read behind an LDS-DMA builtin with vmcnt(0); its own vmcnt bookkeeping stays
    //  safe -- in-order retirement, extra operations it does not know about can only make its waits longer)
    auto dma_piece = [&](unsigned lds_at, unsigned voff, const uint8_t *src) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_at), "v"(voff), "s"(src) : "memory", "m0");
    };
    auto dma_k = [&](int kv0, int buf) {
        const uint8_t *src = kbase + (size_t)kv0 * p.ldk * 2;
#pragma unroll
        for (int i = 0; i < 4; i++) dma_piece(lds0 + buf * 2 * ATT_TILE + i * 1024, (unsigned)kdma[i], src);
    };
    auto dma_v = [&](int kv0, int buf) {
        const uint8_t *src = vtbase + (size_t)kv0 * 2;
#pragma unroll
        for (int i = 0; i < 4; i++) dma_piece(lds0 + buf * 2 * ATT_TILE + ATT_TILE + i * 1024, (unsigned)vdma[i], src);
    };

    // state, in the shapes the loop's register plan wants (tools/gen_attn_loop.py): a score set = two 32-register halves (row tile rt:
    // key half kt at [16 kt .. 16 kt + 15]); O as four 32-register pieces ((rt, dt) at O[2 rt + dt / 2][16 (dt & 1) ..]); Q' as two
    // (fragment ds at [4 ds .. 4 ds + 3]); the accumulator start values (row tile rt at [16 rt ..]); mc; the row sums {l2a0, l2b0, l2a1, l2b1}
    v32f SA[RT], SB[RT], O[4], MI;
    v32i Q32[RT];
    v2f mc;
    v4f l2;
    const float c = p.scale_log2e;

    auto qfrag = [&](int rt, int ds) {
        return __builtin_bit_cast(V8, v4i{Q32[rt][4 * ds], Q32[rt][4 * ds + 1], Q32[rt][4 * ds + 2], Q32[rt][4 * ds + 3]});
    };
    // P = exp2(s') of one score set -> the 16-bit B fragments of the PV MFMA, and O += V^T P from buffer `buf`; row sums over the
    // rounded probabilities (the segment's last tile: the loop handles every tile that has a successor)
    auto last_tile = [&](v32f (&sd)[RT], int buf) {
        V8 pf[RT][4];
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                float e[8];
#pragma unroll
                for (int i = 0; i < 8; i++) e[i] = __builtin_amdgcn_exp2f(sd[rt][16 * (ks >> 1) + 8 * (ks & 1) + i]);
                unsigned x[2], y[2];
#pragma unroll
                for (int d2 = 0; d2 < 2; d2++) {
                    x[d2] = pack2<DT>(e[2 * d2], e[2 * d2 + 1]);
                    y[d2] = pack2<DT>(e[4 + 2 * d2], e[4 + 2 * d2 + 1]);
                    auto sw = __builtin_amdgcn_permlane32_swap(x[d2], y[d2], false, false);
                    x[d2] = sw[0];
                    y[d2] = sw[1];
                }
                pf[rt][ks] = __builtin_bit_cast(V8, v4i{(int)x[0], (int)x[1], (int)y[0], (int)y[1]});
                l2[2 * rt] = sum2<DT>(y[0], sum2<DT>(x[0], l2[2 * rt]));
                l2[2 * rt + 1] = sum2<DT>(y[1], sum2<DT>(x[1], l2[2 * rt + 1]));
            }
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int dt = 0; dt < 4; dt++) {
                v16f acc;
#pragma unroll
                for (int r = 0; r < 16; r++) acc[r] = O[2 * rt + (dt >> 1)][16 * (dt & 1) + r];
#pragma unroll
                for (int ks = 0; ks < 4; ks++) {
                    const v4i vw = *(const lds_v4i *)(L8 + ((unsigned)va[ks] + (buf * 2 * ATT_TILE + dt * 4096)));
                    acc = Half<DT>::mfma32(__builtin_bit_cast(V8, vw), pf[rt][ks], acc);
                }
#pragma unroll
                for (int r = 0; r < 16; r++) O[2 * rt + (dt >> 1)][16 * (dt & 1) + r] = acc[r];
            }
    };

    // Persistent schedule, geometry 2's order: the remainder run FIRST, whole tasks afterwards.  Every split segment publishes -- a
    // contributor to its slab (as in geometry 1), the owner of a split task (the segment with the task's first tiles) to its stash --
    // and nobody waits: the owner folds the parts together after its whole tasks, when the contributors' slabs are long since written.
    // (Geometry 1 runs whole tasks first and its owners wait with their registers full; the schedule arithmetic is the same.)
    int own_tl = -1; // the split task this workgroup owns (at most one: a run is shorter than a task)
    while (true) {
        int task, j0, j1, tl = -1;
        if (PERSIST && pos < run_hi) {
            tl = pos / ntiles;
            j0 = pos - tl * ntiles;
            j1 = min(ntiles, j0 + (run_hi - pos));
            pos += j1 - j0;
            task = sched.F * G + tl;
        } else if (whole_left > 0) {
            if constexpr (PERSIST) task = (sched.F - whole_left) * G + g;
            else {
                const int b = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x, N = (int)(gridDim.x * gridDim.y);
                task = N % 8 == 0 ? (b % 8) * (N / 8) + b / 8 : b;
            }
            j0 = 0; j1 = ntiles;
            if constexpr (MASK) { j0 = p.mask_j0; j1 = p.mask_j1; } // the main segment (host: attention_mask_segment)
            whole_left--;
        } else break;
        const int head = task / QT;
        const int q0 = (task - head * QT) * 256 + wave * 64;
        kbase = (const uint8_t *)(p.k + (size_t)head * p.k_hs);
        vtbase = (const uint8_t *)(p.vt + (size_t)head * p.vt_hs);
        // prologue: K(j0), V^T(j0) -> buffer 0, K(j0+1) -> buffer 1 (a segment has at least two tiles); every wave passed the barrier
        // that ends the previous segment: all buffers are free.  Q' meanwhile.
        dma_k(j0 * ATT_KB, 0);
        dma_v(j0 * ATT_KB, 0);
        dma_k((j0 + 1) * ATT_KB, 1);
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const uint16_t *qrow = p.q + (size_t)(q0 + 32 * rt + lr) * p.ldq + (size_t)head * p.q_hs + 8 * h;
#pragma unroll
            for (int ds = 0; ds < 8; ds++) {
                const V8 qraw = *reinterpret_cast<const V8 *>(qrow + 16 * ds);
                V8 qv;
#pragma unroll
                for (int i = 0; i < 8; i++) qv[i] = c == 1.0f ? qraw[i] : f2h<T>(h2f(qraw[i]) * c); // (c = 1: Q came prescaled, no second rounding)
                const v4i w = __builtin_bit_cast(v4i, qv);
#pragma unroll
                for (int k = 0; k < 4; k++) Q32[rt][4 * ds + k] = w[k];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int r = 0; r < 32; r++) O[i][r] = 0.f;
        l2 = v4f{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // first tile: S'(j0) against the start value 0, then every row takes its own maximum as the reference point
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
#pragma unroll
            for (int kt = 0; kt < 2; kt++) {
                v16f acc;
#pragma unroll
                for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
                for (int ds = 0; ds < 8; ds++) {
                    const v4i kw = *(const lds_v4i *)(L8 + ((unsigned)ka[ds] + kt * 8192));
                    acc = Half<DT>::mfma32(__builtin_bit_cast(V8, kw), qfrag(rt, ds), acc);
                }
#pragma unroll
                for (int r = 0; r < 16; r++) SA[rt][16 * kt + r] = acc[r];
            }
            float mloc = SA[rt][0];
#pragma unroll
            for (int r = 1; r < 32; r++) mloc = fmaxf(mloc, SA[rt][r]);
            const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, mloc), __builtin_bit_cast(unsigned, mloc), false, false);
            const unsigned mine = sw[0], other = sw[1]; // (scalar copies: bit_cast on a vector element reads element 0 in this clang)
            mloc = fmaxf(__builtin_bit_cast(float, mine), __builtin_bit_cast(float, other));
            const bool finite = mloc > -INFINITY;
            mc[rt] = finite ? mloc : -INFINITY;
            const float shift = finite ? mloc : 0.f;
#pragma unroll
            for (int r = 0; r < 32; r++) SA[rt][r] -= shift;
#pragma unroll
            for (int r = 0; r < 16; r++) MI[16 * rt + r] = -shift;
        }
        __syncthreads(); // K(j0) has been read by every wave before the loop's first iteration requests K(j0+2) into its buffer
        SVDQ_ATTN_PROBE_LOOP_BEGIN();
        {
            // ---- the tile loop: iterations j0 .. j1 - 2 (generated assembly, every operand pinned to a physical register) ----------
            unsigned jj = (unsigned)j0;
            const unsigned jend = (unsigned)j1, kstride = (unsigned)p.ldk * (ATT_KB * 2);
#define SVDQ_ATTN_LOOP_OPERANDS                                                                                                         \
            : "+{v[0:31]}"(SA[0]), "+{v[32:63]}"(SA[1]), "+{v[64:95]}"(SB[0]), "+{v[96:127]}"(SB[1]), "+{v[192:223]}"(MI),           \
              "+{a[0:31]}"(O[0]), "+{a[32:63]}"(O[1]), "+{a[64:95]}"(O[2]), "+{a[96:127]}"(O[3]), "+{v[244:245]}"(mc),               \
              "+{v[248:251]}"(l2), "+{s46}"(jj)                                                                                        \
            : "{a[128:159]}"(Q32[0]), "{a[160:191]}"(Q32[1]), "{v[224:231]}"(ka), "{v[232:235]}"(va), "{v[236:239]}"(kdma),          \
              "{v[240:243]}"(vdma), "{s40}"(lds0), "{s[42:43]}"(kbase), "{s[44:45]}"(vtbase), "{s47}"(jend), "{s48}"(kstride)         \
            : "memory", "scc", "vcc", "m0", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60",    \
              "s61", "s62", "s63", "s64", "s65", "s66", "s67", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135",       \
              "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149",        \
              "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163",        \
              "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177",        \
              "v178", "v179", "v180", "v181", "v182", "v183",                                                                        \
              "v246", "v247", "v252", "v253", "v254", "v255"
            if constexpr (DT == SVDQ_BF16) {
                asm volatile(
#include SVDQ_ATTN_LOOP_INC_BF16
                    SVDQ_ATTN_LOOP_OPERANDS);
            } else {
                asm volatile(
#include SVDQ_ATTN_LOOP_INC_FP16
                    SVDQ_ATTN_LOOP_OPERANDS);
            }
#undef SVDQ_ATTN_LOOP_OPERANDS
        }
        last_tile(SB, 1); // a segment has an even number of tiles: its last one sits in score set B / buffer 1
        __syncthreads();  // the buffers are free for the next segment's prologue
        if constexpr (MASK) {
            // branch-free on purpose: written with short-circuit || / && this clang if-converts the select into AGPR writes under a partial exec mask
            // and loses the scores of range-B keys (round 4: a padded tile at the end of the second range came out wrong on a few rows)
            const unsigned b_len = (unsigned)(p.kv_end1 - p.kv_start1);
            auto real_key = [&](int key) { return (int)(key < p.kv_len0) | (int)((unsigned)(key - p.kv_start1) < b_len); };
            for (int t = 0; t < ntiles; t++) {
                const int k0 = t * ATT_KB;
                if ((t >= j0 && t < j1) || !(k0 < p.kv_len0 || (k0 + ATT_KB > p.kv_start1 && k0 < p.kv_end1))) continue; // done by the loop / no real key
                // ---- an extra tile: K(t), V^T(t) -> buffer 0; S' = K Q'^T against the current start values; padded keys -> -inf; rows that outgrew
                //      their reference point (or see their first finite score) move it: O, l and the start values follow; then exp / PV as a last tile
                dma_k(k0, 0);
                dma_v(k0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                const bool all_real = k0 + ATT_KB <= p.kv_len0 || (k0 >= p.kv_start1 && k0 + ATT_KB <= p.kv_end1);
                // which of this lane's 2 x 16 score rows (keys) are real: ONE bit mask per key tile, computed once per extra tile (it does not depend on
                // the row tile) and made opaque to the optimiser -- the selects below can only become bit test + v_cndmask on it, never control flow
                // around the accumulator writes (the if-conversion hazard described above; tests/test_gpu_attention.py keeps the two-range case)
                unsigned km[2];
#pragma unroll
                for (int kt = 0; kt < 2; kt++) {
                    unsigned m = 0;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int key = k0 + 32 * kt + 8 * (r >> 2) + 4 * h + (r & 3); // C layout of S^T: row (key) = 8 (r / 4) + 4 h + r % 4
                        m |= (unsigned)((int)all_real | real_key(key)) << r;
                    }
                    asm volatile("" : "+v"(m));
                    km[kt] = m;
                }
#pragma unroll
                for (int rt = 0; rt < RT; rt++) {
#pragma unroll
                    for (int kt = 0; kt < 2; kt++) {
                        v16f acc;
#pragma unroll
                        for (int r = 0; r < 16; r++) acc[r] = MI[16 * rt + r];
#pragma unroll
                        for (int ds = 0; ds < 8; ds++) {
                            const v4i kw = *(const lds_v4i *)(L8 + ((unsigned)ka[ds] + kt * 8192));
                            acc = Half<DT>::mfma32(__builtin_bit_cast(V8, kw), qfrag(rt, ds), acc);
                        }
#pragma unroll
                        for (int r = 0; r < 16; r++) SA[rt][16 * kt + r] = ((km[kt] >> r) & 1u) ? acc[r] : -INFINITY;
                    }
                    float mloc = SA[rt][0];
#pragma unroll
                    for (int r = 1; r < 32; r++) mloc = fmaxf(mloc, SA[rt][r]);
                    const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, mloc), __builtin_bit_cast(unsigned, mloc), false, false);
                    const unsigned mine = sw[0], other = sw[1];
                    mloc = fmaxf(__builtin_bit_cast(float, mine), __builtin_bit_cast(float, other)); // the row's largest score RELATIVE to its reference point
                    const bool fresh = !(mc[rt] > -INFINITY);           // the row has seen no finite score yet (its start value is 0)
                    const bool move = mloc > -INFINITY && (fresh || mloc > ATT_DEFER_LOG2);
                    if (__builtin_amdgcn_ballot_w64(move) != 0) {
                        const float d = move ? mloc : 0.f;               // shift of this row's reference point
                        const float f = !move ? 1.f : fresh ? 0.f : __builtin_amdgcn_exp2f(-d); // (fresh: O and l are still zero)
                        mc[rt] = move ? (fresh ? mloc : mc[rt] + d) : mc[rt];
#pragma unroll
                        for (int r = 0; r < 32; r++) SA[rt][r] -= d;
#pragma unroll
                        for (int r = 0; r < 16; r++) MI[16 * rt + r] = move ? -mc[rt] : MI[16 * rt + r];
#pragma unroll
                        for (int r = 0; r < 32; r++) { O[2 * rt][r] *= f; O[2 * rt + 1][r] *= f; }
                        l2[2 * rt] *= f;
                        l2[2 * rt + 1] *= f;
                    }
                }
                last_tile(SA, 0);
                __syncthreads();
            }
        }
        SVDQ_ATTN_PROBE_LOOP_END(j1 - j0);
        float l_run[RT] = {l2[0] + l2[1], l2[2] + l2[3]};

        if constexpr (PERSIST) {
            if (tl >= 0 && (j0 > 0 || j1 < ntiles)) { // a part of a split task: publish the un-normalised state, no epilogue here
                typedef __attribute__((address_space(1))) int gint;
                typedef __attribute__((address_space(1))) v4f gv4f;
                typedef __attribute__((address_space(1))) v2f gv2f;
                // contributor: slab g (the slab image of geometry 1, [wave][rt][j][lane]; the reference point in log2 units); owner: stash G + g
                float *slab = p.ws_slabs + (size_t)(j0 > 0 ? g : G + g) * ATT_SLAB_FLOATS;
#define SVDQ_OREG(rt, j, e) O[2 * (rt) + ((j) >> 3)][16 * (((j) >> 2) & 1) + ((j) & 3) * 4 + (e)] /* register (j = 4 dt + c, e) of row tile rt */
#pragma unroll
                for (int rt = 0; rt < RT; rt++) {
#pragma unroll
                    for (int j = 0; j < 16; j++)
                        *(gv4f *)(slab + ((size_t)((wave * RT + rt) * 16 + j) * 64 + lane) * 4) = v4f{SVDQ_OREG(rt, j, 0), SVDQ_OREG(rt, j, 1), SVDQ_OREG(rt, j, 2), SVDQ_OREG(rt, j, 3)};
                    *(gv2f *)(slab + ATT_SLAB_O + (size_t)((wave * RT + rt) * 64 + lane) * 2) = v2f{mc[rt], l_run[rt]};
                }
#undef SVDQ_OREG
                if (j0 > 0) {
                    const int owner = sched.owner_of(g, tl);
                    __syncthreads();
                    if (tid == 0) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __hip_atomic_fetch_add((gint *)p.ws_flags + owner, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                } else own_tl = tl; // (its own later loads of the stash are ordered behind these stores: same wave, same addresses)
                continue;
            }
        }
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            v16f o4[4];
#pragma unroll
            for (int dt = 0; dt < 4; dt++)
#pragma unroll
                for (int r = 0; r < 16; r++) o4[dt][r] = O[2 * rt + (dt >> 1)][16 * (dt & 1) + r];
            finish_rows<DT>(p, o4, l_run[rt], q0 + 32 * rt, head, lane);
        }
    } // segments
    if constexpr (PERSIST) {
        if (own_tl >= 0) { // ---- the split task this workgroup owns: its own stash + the contributors' slabs, ascending workgroup order, then the epilogue
            typedef __attribute__((address_space(1))) int gint;
            typedef __attribute__((address_space(1))) v4f gv4f;
            typedef __attribute__((address_space(1))) v2f gv2f;
            gint *flags = (gint *)p.ws_flags;
            const int task = sched.F * G + own_tl, head = task / QT, q0 = (task - head * QT) * 256 + wave * 64;
            const int last = sched.last_contributor(g, own_tl);
            if (tid == 0) {
                int spins = 0; // bounded wait, as in geometry 1 (here the arrivals are normally long since complete)
                while (__hip_atomic_load(flags + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < last - g && ++spins < ATT_SPIN_LIMIT)
                    __builtin_amdgcn_s_sleep(8);
                if (spins >= ATT_SPIN_LIMIT) {
                    __hip_atomic_store(flags + ATT_ERR_WORD, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (p.status) __hip_atomic_store(p.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                __hip_atomic_store(flags + g, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                v16f o4[4];
                float mrun = -INFINITY, lrun = 0.f;
#pragma unroll
                for (int dt = 0; dt < 4; dt++)
#pragma unroll
                    for (int r = 0; r < 16; r++) o4[dt][r] = 0.f;
                for (int q = g; q <= last; q++) { // q = g: the owner's own stash
                    const float *slab = p.ws_slabs + (size_t)(q == g ? G + g : q) * ATT_SLAB_FLOATS;
                    const v2f ml = __builtin_nontemporal_load((const gv2f *)(slab + ATT_SLAB_O + (size_t)((wave * RT + rt) * 64 + lane) * 2));
                    const float m_new = fmaxf(mrun, ml[0]); // reference points in log2 units (-inf: that part saw no finite score: weight 0)
                    const float fa = mrun == m_new ? 1.f : __builtin_amdgcn_exp2f(mrun - m_new), fb = ml[0] == m_new ? 1.f : __builtin_amdgcn_exp2f(ml[0] - m_new);
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        const v4f v = __builtin_nontemporal_load((const gv4f *)(slab + ((size_t)((wave * RT + rt) * 16 + j) * 64 + lane) * 4));
#pragma unroll
                        for (int e = 0; e < 4; e++) o4[j >> 2][(j & 3) * 4 + e] = o4[j >> 2][(j & 3) * 4 + e] * fa + v[e] * fb;
                    }
                    lrun = lrun * fa + ml[1] * fb;
                    mrun = m_new;
                }
                finish_rows<DT>(p, o4, lrun, q0 + 32 * rt, head, lane);
            }
        }
    }
    SVDQ_ATTN_PROBE_END();
}

template <int DT, int NW> static void launch_attention(const AttnParams &p, hipStream_t st) {
    dim3 grid(p.L / (NW * 32), p.H), block(NW * 64);
    hipLaunchKernelGGL((attention_kernel<DT, NW, false>), grid, block, 0, st, p);
}
template <int DT> static void launch_attention_persistent(const AttnParams &p, int groups, hipStream_t st) {
    hipLaunchKernelGGL((attention_kernel<DT, 8, true>), dim3(groups), dim3(512), 0, st, p);
}
template <int DT> static void launch_attention64(const AttnParams &p, int groups, hipStream_t st) {
    if (p.kv_len0 > 0) hipLaunchKernelGGL((attention_kernel64<DT, false, true>), dim3(p.L / 256, p.H), dim3(256), 0, st, p);
    else if (groups > 0) hipLaunchKernelGGL((attention_kernel64<DT, true>), dim3(groups), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((attention_kernel64<DT, false>), dim3(p.L / 256, p.H), dim3(256), 0, st, p);
}
// Geometry 2 under a key mask: the longest run of fully real 64-key tiles, cut to an even count (what the assembly loop runs); false when no run has two
static bool attention_mask_segment(const svdq_attention_args *a, int &j0, int &j1) {
    int a0 = 0, a1 = a->kv_len0 / ATT_KB;
    int b0 = (a->kv_start1 + ATT_KB - 1) / ATT_KB, b1 = a->kv_end1 / ATT_KB;
    if (a->kv_end1 <= a->kv_start1 || b1 < b0) b0 = b1 = 0;
    if (b1 - b0 > a1 - a0) { a0 = b0; a1 = b1; }
    const int n = (a1 - a0) & ~1;
    if (n < 2) return false;
    j0 = a0;
    j1 = a0 + n;
    return true;
}

static bool attention_masked_geometry2(const svdq_attention_args *a, int &j0, int &j1) {
    j0 = j1 = 0;
    return a->kv_len0 > 0 && a->geometry != 1 && a->L % 256 == 0 && (a->q_prescaled || a->geometry == 2) && attention_mask_segment(a, j0, j1);
}

// workgroups of the persistent schedule: one per CU (64 KiB of LDS and 8 waves of ~230 VGPRs: exactly one is resident per CU)
static int attention_cus() {
    static int cus = 0; // benign race: every thread computes the same value
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        if (n > 1016) n = 1016; // the workspace header holds 1023 arrival counters
        cus = n >= 8 ? (n / 8) * 8 : n;
    }
    return cus;
}
static int attention_groups_for(int L, int H, int cus);
// Persistent or plain grid?  Whole rounds of tasks need no split (and no partial traffic); otherwise deal the KV tiles of
// all tasks evenly to min(CUs, tiles/2) workgroups, a multiple of 8 (XCD-aware numbering).  0 = plain grid.
static int attention_groups(const AttnParams &p) { return p.ws_flags ? attention_groups_for(p.L, p.H, attention_cus()) : 0; }
static int attention_groups_for(int L, int H, int cus) {
    if (L % 256) return 0;
    const long long tasks = (long long)H * (L / 256), half = tasks * (L / ATT_KB) / 2;
    if (tasks % cus == 0) return 0;
    long long gs = half < cus ? half : cus;
    if (gs >= 8) gs = gs / 8 * 8;
    return gs >= 2 ? (int)gs : 0;
}

} // namespace svdq

using namespace svdq;

// header + two slabs per workgroup of the persistent schedule: the one a contributor publishes, and (geometry 2) the stash an owner parks its own part in
extern "C" int64_t svdq_attention_workspace_bytes(void) { return ATT_WS_HEADER + 2 * (int64_t)attention_cus() * ATT_SLAB_FLOATS * 4; }
// the fused quantiser's low-rank down projection can run split (lowrank_split.h): fp32 accumulators, rank 48 .. 160, K = H * 128 a multiple of 256
static bool attention_split_shape_ok(const svdq_attention_args *a) {
    return a->qact && a->qlora_act && a->qlora_down && a->qlora_act_format == SVDQ_LORA_ACT_F32 && a->L > 0 && a->L % 256 == 0 && a->H > 0 &&
           lowrank_split_shape_ok(a->H * ATT_D, a->qR);
}
// ABI 20: the workspace size with which THIS launch takes every fast path: svdq_attention_workspace_bytes(), plus -- fused quantiser of rank 48 .. 160 -- the
// packed down projection(s) and the 16-bit output image (L * H * 128 * 2 bytes) of the split low-rank down projection.  A smaller workspace is never an error.
extern "C" int64_t svdq_attention_workspace_bytes_for(const svdq_attention_args *a) {
    const int64_t base = svdq_attention_workspace_bytes();
    if (!a || !attention_split_shape_ok(a)) return base;
    return base + lowrank_split_pack_bytes(a->H * ATT_D, a->qR, a->qsmooth2 != nullptr) + (int64_t)a->L * a->H * ATT_D * 2;
}

extern "C" int svdq_attention_workspace_status(void *workspace, void *stream) {
    if (!workspace) { set_error("svdq_attention_workspace_status: workspace is NULL"); return SVDQ_E_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    int word = 0;
    int *dev = reinterpret_cast<int *>(workspace) + ATT_ERR_WORD;
    if (hip_check(hipMemcpyAsync(&word, dev, sizeof(int), hipMemcpyDeviceToHost, st), "svdq_attention_workspace_status copy")) return SVDQ_E_HIP;
    if (hip_check(hipStreamSynchronize(st), "svdq_attention_workspace_status sync")) return SVDQ_E_HIP;
    if (word != 0) {
        (void)hipMemsetAsync(dev, 0, sizeof(int), st);
        set_error("svdq_attention: a task owner timed out waiting for partial results -- the workspace was used by launches in "
                  "flight on more than one stream (or was not zero-filled); results of those launches are invalid");
        return SVDQ_E_HIP;
    }
    return SVDQ_OK;
}

// Host-side replay of the persistent schedule (the same AttnSchedule code the kernel runs): for (L, H) on `cus` compute units
// write up to `cap` records {workgroup, task, j0, j1, owner-or-minus-one, last-contributor-or-minus-one} and return the
// number of segments; 0 = this problem runs on the plain grid; -1 = bad arguments.
extern "C" int svdq_attention_schedule(int32_t L, int32_t H, int32_t cus, int32_t *out, int32_t cap) {
    if (L <= 0 || L % 128 || H <= 0 || cus <= 0 || cus > 1016) return -1;
    const int G = attention_groups_for(L, H, cus >= 8 ? cus / 8 * 8 : cus);
    if (G == 0) return 0;
    const int ntiles = L / ATT_KB, QT = L / 256;
    AttnSchedule sched;
    sched.init(H * QT, ntiles, G);
    int n = 0;
    auto emit = [&](int g, int task, int j0, int j1, int owner, int last) {
        if (out && n < cap) {
            int32_t *r = out + 6 * n;
            r[0] = g; r[1] = task; r[2] = j0; r[3] = j1; r[4] = owner; r[5] = last;
        }
        n++;
    };
    for (int g = 0; g < G; g++) {
        for (int f = 0; f < sched.F; f++) emit(g, f * G + g, 0, ntiles, -1, -1); // whole tasks first
        if (g >= sched.Gs) continue;
        for (int pos = sched.bound(g), hi = sched.bound(g + 1); pos < hi;) {
            const int tl = pos / ntiles, j0 = pos - tl * ntiles;
            const int j1 = ntiles < j0 + (hi - pos) ? ntiles : j0 + (hi - pos);
            pos += j1 - j0;
            emit(g, sched.F * G + tl, j0, j1, j0 > 0 ? sched.owner_of(g, tl) : -1, j0 == 0 && j1 < ntiles ? sched.last_contributor(g, tl) : -1);
        }
    }
    return n;
}

// Which kernel a launch with these arguments would take (host only, nothing is launched): out[0] = workgroup geometry (1 / 2), out[1] = 1 when the
// key mask runs on geometry 2, out[2], out[3] = the main segment [j0, j1) of fully real 64-key tiles its assembly loop walks (the other tiles that
// hold a real key are C++ "extra" tiles).  Pointers are not looked at.
extern "C" int svdq_attention_plan(const svdq_attention_args *a, int32_t *out) {
    if (!a || !out) { set_error("svdq_attention_plan: args and out are required"); return SVDQ_E_INVALID; }
    if (a->L <= 0 || a->L % 128) { set_error("svdq_attention_plan: L=%d must be a positive multiple of 128", a->L); return SVDQ_E_INVALID; }
    int j0 = 0, j1 = 0;
    const bool mask2 = attention_masked_geometry2(a, j0, j1);
    out[0] = a->kv_len0 > 0 ? (mask2 ? 2 : 1) : a->geometry ? a->geometry : (a->L % 256 == 0 && a->q_prescaled ? 2 : 1);
    out[1] = mask2 ? 1 : 0;
    out[2] = j0;
    out[3] = j1;
    return SVDQ_OK;
}

// What the calling thread's last svdq_attention launched (svdq_attention_last_plan): {workgroup geometry, workgroups of the persistent schedule (0 = plain grid),
// key mask on geometry 2, the fused quantiser's low-rank down projection ran split}
static thread_local int32_t g_attn_last_plan[4] = {0, 0, 0, 0};
extern "C" int svdq_attention_last_plan(int32_t *out4) {
    if (!out4) { set_error("svdq_attention_last_plan: out is NULL"); return SVDQ_E_INVALID; }
    for (int i = 0; i < 4; i++) out4[i] = g_attn_last_plan[i];
    return SVDQ_OK;
}

extern "C" int svdq_attention(const svdq_attention_args *a, void *stream) {
    if (!a) { set_error("svdq_attention: args is NULL"); return SVDQ_E_INVALID; }
    if (!a->q || !a->k || !a->vt || (!a->out && !a->qact)) { set_error("svdq_attention: q, k, vt and out (or qact) are required"); return SVDQ_E_INVALID; }
    if (a->qact) {
        if (!a->qscales || !a->qsmooth || a->L % 256 || a->qR < 0 || a->qR > 256 || a->qR % 16 || (a->qR > 0 && (!a->qlora_down || !a->qlora_act))) {
            set_error("svdq_attention: fused quantiser needs qscales, qsmooth, L %% 256 == 0 and R=%d a multiple of 16 in [0, 256] with qlora_down / qlora_act", a->qR);
            return SVDQ_E_INVALID;
        }
        if (a->qsmooth2 && (a->qsplit_rows <= 0 || a->qsplit_rows % 256 || a->qsplit_rows >= a->L || (a->qR > 0 && !a->qlora_down2))) {
            set_error("svdq_attention: fused quantiser: 0 < qsplit_rows < L must be a multiple of 256 (and qlora_down2 given)");
            return SVDQ_E_INVALID;
        }
        if (((uintptr_t)a->qact & 15) || (((uintptr_t)a->qsmooth | (uintptr_t)a->qlora_down | (uintptr_t)a->qsmooth2 | (uintptr_t)a->qlora_down2) & 7)) {
            set_error("svdq_attention: fused quantiser: qact must be 16-byte aligned, the vectors 8-byte");
            return SVDQ_E_INVALID;
        }
    }
    if (a->qlora_act_format != SVDQ_LORA_ACT_F32 && a->qlora_act_format != SVDQ_LORA_ACT_Q32 && a->qlora_act_format != SVDQ_LORA_ACT_Q32_RUNS) { set_error("svdq_attention: unknown qlora_act_format %d", a->qlora_act_format); return SVDQ_E_INVALID; }
    if (a->kv_len0 < 0 || a->kv_len0 > a->L || (a->kv_len0 == 0 && (a->kv_start1 || a->kv_end1)) ||
        (a->kv_len0 > 0 && (a->kv_start1 < a->kv_len0 || a->kv_end1 < a->kv_start1 || a->kv_end1 > a->L) && (a->kv_start1 || a->kv_end1))) {
        set_error("svdq_attention: key mask needs 0 < kv_len0 <= kv_start1 <= kv_end1 <= L=%d (kv_len0 = 0: no mask; kv_start1 = kv_end1 = 0: one range)", a->L);
        return SVDQ_E_INVALID;
    }
    if (a->head_dim != ATT_D) { set_error("svdq_attention: head_dim=%d (only 128 is implemented, as in the reference kernel)", a->head_dim); return SVDQ_E_UNSUPPORTED; }
    if (a->L <= 0 || a->L % 128 || a->H <= 0) { set_error("svdq_attention: L=%d must be a positive multiple of 128 and H=%d positive", a->L, a->H); return SVDQ_E_INVALID; }
    if (a->ldq % 8 || a->ldk % 8 || a->ldvt % 8 || a->ldo % 8 || a->q_hs % 8 || a->k_hs % 8 || a->vt_hs % 8 || a->o_hs % 8 ||
        a->ldvt < a->L || a->ldq < ATT_D || a->ldk < ATT_D || (a->out && a->ldo < ATT_D)) {
        set_error("svdq_attention: strides must be multiples of 8 elements, ldq/ldk/ldo >= 128 and ldvt >= L");
        return SVDQ_E_INVALID;
    }
    if (((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->vt | (uintptr_t)a->out) & 15) { set_error("svdq_attention: q, k, vt, out must be 16-byte aligned"); return SVDQ_E_INVALID; }
    if (a->dtype != SVDQ_BF16 && a->dtype != SVDQ_FP16) { set_error("svdq_attention: unknown dtype %d", a->dtype); return SVDQ_E_INVALID; }
    if (a->zero_ptr && (a->zero_bytes < 0 || a->zero_bytes % 16 || ((uintptr_t)a->zero_ptr & 15))) {
        set_error("svdq_attention: zero_ptr must be 16-byte aligned and zero_bytes a non-negative multiple of 16");
        return SVDQ_E_INVALID;
    }
    AttnParams p;
    p.q = (const uint16_t *)a->q; p.k = (const uint16_t *)a->k; p.vt = (const uint16_t *)a->vt; p.out = (uint16_t *)a->out;
    p.q_hs = a->q_hs; p.k_hs = a->k_hs; p.vt_hs = a->vt_hs; p.o_hs = a->o_hs;
    p.L = a->L; p.H = a->H; p.ldq = a->ldq; p.ldk = a->ldk; p.ldvt = a->ldvt; p.ldo = a->ldo;
    p.scale_log2e = a->q_prescaled ? 1.0f : a->scale * 1.4426950408889634f; // (prescaled: Q carries scale * log2(e), svdq_gemm_args.q_scale)
    p.qact = (uint8_t *)a->qact; p.qscales = (uint16_t *)a->qscales; p.qlora_act = a->qlora_act;
    p.qlora_q32 = a->qlora_act_format != SVDQ_LORA_ACT_F32;
    p.status = a->status;
    p.kv_len0 = a->kv_len0; p.kv_start1 = a->kv_start1; p.kv_end1 = a->kv_end1;
    p.qsmooth = (const uint16_t *)a->qsmooth; p.qlora_down = (const uint16_t *)a->qlora_down;
    p.qsmooth2 = (const uint16_t *)a->qsmooth2; p.qlora_down2 = (const uint16_t *)a->qlora_down2;
    p.qR = a->qR; p.qsplit_rows = a->qsmooth2 ? a->qsplit_rows : 0;
    p.zero_ptr = (v4i *)a->zero_ptr;
    p.zero_vec = a->zero_ptr ? a->zero_bytes / 16 : 0;
    p.ws_flags = nullptr;
    p.ws_slabs = nullptr;
    p.qa16 = nullptr;
    void *split_pack = nullptr;
    SVDQ_ATTN_PROBE_FILL(p);
    if (a->workspace) {
        if (((uintptr_t)a->workspace & 15) || a->workspace_bytes < 0) { set_error("svdq_attention: workspace must be 16-byte aligned"); return SVDQ_E_INVALID; }
        if (a->workspace_bytes >= svdq_attention_workspace_bytes()) { // a smaller one is ignored (plain grid), as in svdq_gemm_w4a4
            p.ws_flags = (int *)a->workspace;
            p.ws_slabs = (float *)((uint8_t *)a->workspace + ATT_WS_HEADER);
        }
        // ... and behind the persistent schedule's part (ABI 20, svdq_attention_workspace_bytes_for): the split low-rank down projection of the fused quantiser
        if (attention_split_shape_ok(a) && a->workspace_bytes >= svdq_attention_workspace_bytes_for(a)) {
            split_pack = (uint8_t *)a->workspace + svdq_attention_workspace_bytes();
            p.qa16 = (uint8_t *)split_pack + lowrank_split_pack_bytes(a->H * ATT_D, a->qR, a->qsmooth2 != nullptr);
        }
    }
    hipStream_t st = (hipStream_t)stream;
    if (a->reserved != 0) { set_error("svdq_attention: reserved must be 0 (timing ablations live in tools/ablate, not in this library)"); return SVDQ_E_INVALID; }
    if (a->geometry < 0 || a->geometry > 2 || (a->geometry == 2 && a->L % 256)) {
        set_error("svdq_attention: geometry=%d (0 = automatic, 1 = 8 waves x 32 rows, 2 = 4 waves x 64 rows: needs L %% 256 == 0)", a->geometry);
        return SVDQ_E_INVALID;
    }
    const int nw = a->L % 256 == 0 ? 8 : 4;
    // automatic: geometry 2 on the plain grid when Q comes prescaled, the length allows it and there is no key mask (the mask lives in
    // geometry 1).  Same box, 24 heads x 4608 tokens: 216 us against 274 for geometry 1 on its better schedule; geometry 2's own
    // persistent schedule 236.  With a raw Q geometry 2 has to scale the 16-bit values itself (a second rounding, ~2.5x the error against
    // fp32): only on explicit request.  An explicit geometry takes the workspace if given.
    // (round 4: a masked launch takes geometry 2 as well -- plain grid, the padded / odd tiles as C++ "extra" tiles behind the assembly loop,
    //  attention_kernel64<.., MASK> -- when the same conditions hold and some run of fully real tiles has at least two)
    const bool mask2 = attention_masked_geometry2(a, p.mask_j0, p.mask_j1);
    const int geometry = a->kv_len0 > 0 ? (mask2 ? 2 : 1) : a->geometry ? a->geometry : (a->L % 256 == 0 && a->q_prescaled ? 2 : 1);
    const int groups = geometry == 2 && (a->geometry == 0 || mask2) ? 0 : attention_groups(p);
    g_attn_last_plan[0] = geometry; g_attn_last_plan[1] = groups; g_attn_last_plan[2] = mask2 ? 1 : 0; g_attn_last_plan[3] = p.qa16 ? 1 : 0;
    const int prof = prof_begin(2, 4.0 * a->L * (double)a->L * a->H * ATT_D, st);
    auto launch = [&]() {
    if (geometry == 2 && a->L % 256 == 0) { if (a->dtype == SVDQ_FP16) launch_attention64<SVDQ_FP16>(p, groups, st); else launch_attention64<SVDQ_BF16>(p, groups, st); }
    else if (groups > 0) { if (a->dtype == SVDQ_FP16) launch_attention_persistent<SVDQ_FP16>(p, groups, st); else launch_attention_persistent<SVDQ_BF16>(p, groups, st); }
    else if (a->dtype == SVDQ_FP16) { if (nw == 8) launch_attention<SVDQ_FP16, 8>(p, st); else launch_attention<SVDQ_FP16, 4>(p, st); }
    else if (nw == 8) launch_attention<SVDQ_BF16, 8>(p, st);
    else launch_attention<SVDQ_BF16, 4>(p, st);
    };
    if (p.qa16) { // pack the down projection(s), the attention kernel (its epilogue stores the 16-bit fragments), the contraction into qlora_act
        const int split_row = a->qsmooth2 ? a->qsplit_rows : 0x7fffffff;
        if (a->dtype == SVDQ_FP16) launch_lowrank_down_split<SVDQ_FP16>(p.qa16, a->qlora_down, a->qlora_down2, split_row, a->L, a->H * ATT_D, a->qR, (float *)a->qlora_act, split_pack, attention_cus(), st, launch, a->qlora_down_packed, a->qlora_down_packed2);
        else launch_lowrank_down_split<SVDQ_BF16>(p.qa16, a->qlora_down, a->qlora_down2, split_row, a->L, a->H * ATT_D, a->qR, (float *)a->qlora_act, split_pack, attention_cus(), st, launch, a->qlora_down_packed, a->qlora_down_packed2);
    } else launch();
    prof_end(prof, st);
    return hip_check(hipGetLastError(), "svdq_attention launch");
}
