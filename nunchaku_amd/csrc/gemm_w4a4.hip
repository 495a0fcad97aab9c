// svdq_gemm_w4a4: fused W4A4 GEMM + per-group dequant + bias + rank-R low-rank correction +
// activation / requantisation / RMSNorm+RoPE epilogues for gfx950 (MI355X).
//
// Replaces the reference's gemm_w4a4_kernel and its epilogue chain
// (gemm_w4a4.cuh:831-928,1046-1095; gemm_base.cuh:367-409,667-792; lora.cuh:110-353;
//  epilogues.cuh:22-44,269-425; dispatch gemm_w4a4_launch_impl.cuh:7-424).
//
// Kernel structure (DESIGN.md "GEMM kernel"):
//   * the 4-bit x 4-bit inner product of one 64-channel quantisation group is ONE
//     v_mfma_scale_f32_32x32x64_f8f6f4 with FP6 (e2m3) operands: a 4-bit code q is exactly the FP6
//     value q/8, products and the 64-term sum are exact in the fp32 accumulator (|sum| <= 7680), and
//     the FP6/FP4 matrix rate is twice the INT8 rate.  Both operands live in HBM and LDS as the
//     register image of that instruction (svdq_common.h "F6"), so the main loop has no unpack work.
//   * the per-group scale tile S[n,m] = ws[n,g]*as[m,g] is a rank-1 product: it is produced by a
//     second MFMA (16-bit operands: the scales themselves in k-slots 0 and 8, zeros elsewhere, so
//     S = 2*ws*as exactly; the 1/2 is folded into the MX block exponent of the FP6 MFMA).  The VALU
//     only does acc = fma(P, S, acc): one op per output element and group, the minimum any exact
//     formulation needs (the int8 formulation needs cvt + mul + fma and is VALU-issue bound at
//     ~1/3 of the matrix peak, profiles/r1_ubench_*).
//   * workgroup = 512 threads = 8 waves (4 along M x 2 along N), output tile 256 x 128, wave tile
//     64 x 64 = 2 x 2 MFMA tiles; two waves per SIMD so that one wave's fma stream overlaps the
//     other's MFMAs (a single wave issues one VALU op per ~8 cycles on gfx950).
//   * K-step = 128 channels (two groups).  Operands go HBM -> LDS by DMA (global_load_lds, 16 B per
//     lane = one 1 KiB plane of an F6 chunk per wave instruction), 3-stage ring (NSTAGE).
//   * the MFMA is issued "transposed" (weights = A operand, activations = B) so that in the C layout
//     a lane owns ONE output row m and 16 columns n = 32*tile + 8c + 4*(lane>>5) + e: 8-byte output
//     stores, RoPE pairs and the next layer's FP6 lane record are lane-local.
//   * fp32 accumulation over groups (the reference accumulates in 16-bit, gemm_w4a4.cuh:1080).  Bias
//     and the low-rank up projection (a 16-bit MFMA issued straight onto the fp32 accumulators) are
//     added before the single rounding to 16-bit that precedes the activation epilogues.
#include "svdq_common.h"
#include "lowrank_split.h"
#include <type_traits>
#include <stdlib.h>

// Clock / phase stamps of the tools-built probe library (tools/ablate/build.py compiles this file with -DSVDQ_PROBE and
// tools/ablate/gemm_probe_hooks.inc on the include path).  The product library has no experiment switches: the hooks
// expand to nothing.
#ifdef SVDQ_PROBE
#include "gemm_probe_hooks.inc"
#else
#define SVDQ_PROBE_PARAMS
#define SVDQ_PROBE_BEGIN()
#define SVDQ_PROBE_STAMP(i)
#define SVDQ_PROBE_NEXT_SEGMENT()
#define SVDQ_PROBE_END()
#define SVDQ_PROBE_FILL(p)
#define SVDQ_PROBE_GRID(g, tiles, slots) (g)
#define SVDQ_PROBE_OFF(bit) false
#define SVDQ_PROBE_WT_MID_ASM
#define SVDQ_PROBE_WT_MID_OUT
#define SVDQ_PROBE_WT_MID_DECL
#define SVDQ_PROBE_WT_MID_STAMP()
#endif
// compile-time timing variants of the RMSNORM_ROPE epilogue (probe builds only: tools/ablate/build.py with SVDQ_PROBE_DEFS; profiles/r6_qkv_epilogue_levers.txt):
// SVDQ_PROBE_ROT 1 = the rotary table read as lane-contiguous 16-byte loads, 2 = no rotary table loads; SVDQ_PROBE_VROW 1 = V tiles row-major into `out`
#ifndef SVDQ_PROBE_ROT
#define SVDQ_PROBE_ROT 0
#endif
#ifndef SVDQ_PROBE_VROW
#define SVDQ_PROBE_VROW 0
#endif

// generated main loops (tools/gen_gemm_loop2.py); the probe build substitutes option variants
#ifndef SVDQ_LOOP_INC_8_BF16
#define SVDQ_LOOP_INC_8_BF16 "gemm_loop2_bf16.inc"
#endif
#ifndef SVDQ_LOOP_INC_8_FP16
#define SVDQ_LOOP_INC_8_FP16 "gemm_loop2_fp16.inc"
#endif
#ifndef SVDQ_LOOP_INC_4_BF16
#define SVDQ_LOOP_INC_4_BF16 "gemm_loop2_w4_bf16.inc"
#endif
#ifndef SVDQ_LOOP_INC_4_FP16
#define SVDQ_LOOP_INC_4_FP16 "gemm_loop2_w4_fp16.inc"
#endif
#ifndef SVDQ_LOOP3_INC_BF16   // the 128 x 64 wave tile kernel: main loop and plain epilogue (tools/gen_gemm_loop3.py)
#define SVDQ_LOOP3_INC_BF16 "gemm_loop3_bf16.inc"
#define SVDQ_LOOP3_INC_FP16 "gemm_loop3_fp16.inc"
#define SVDQ_EPI3_INC_BF16 "gemm_epi3_bf16.inc"
#define SVDQ_EPI3_INC_FP16 "gemm_epi3_fp16.inc"
#endif

namespace svdq {

constexpr int BN = 128;
constexpr int W_BYTES = (BN / 32) * F6_CHUNK;  // 12288
constexpr int AS_BYTES = 1024;                 // (BM / 32) x 128 B used; the DMA writes whole 1 KiB planes
constexpr int WS_BYTES = 1024;                 // 512 used
constexpr int MAX_LORA_TILES = 16;             // R <= 256
struct float16_scales { float v[MAX_LORA_TILES]; };

// Workgroup geometry (tools/gen_gemm_loop2.py: class Geometry -- keep in step).
//   NW = 8: 512 threads, tile 256 x 128, ONE workgroup per CU (156 KiB of LDS), 3-stage ring + the tile's staged epilogue operands.
//   NW = 4: 256 threads, tile 128 x 128, TWO workgroups per CU (80 KiB of LDS each), 3-stage ring: the two workgroups
//           run half a tile out of phase, so one's epilogue (matrix pipe idle: 15-30 % of a K = 3072 tile) and its
//           K-step barrier stalls sit under the other's main loop.
// The wave tile (64 x 64 = 2 x 2 MFMA tiles), the register image of the loop and every epilogue's lane map are the same.
template <int NW> struct Geo {
    static constexpr int BM = 32 * NW;
    static constexpr int THREADS = 64 * NW;
    static constexpr int NSTAGE = 3; // (NW = 8 ran a ring of four until the end of round 3: measured equal, bit-identical; the stage now holds STG_BYTES)
    static constexpr int A_BYTES = (BM / 32) * F6_CHUNK;
    static constexpr int STAGE_BYTES = A_BYTES + W_BYTES + AS_BYTES + WS_BYTES; // 38912 | 26624
    static constexpr int EPI_BYTES = 2 * BM * 4;   // epilogue scratch behind the ring (row sums of the RMSNorm epilogue: [2][BM] fp32)
    static constexpr int MAIL_BYTES = 16;          // one word: the tile id a workgroup's thread 0 drew from the dynamic queue
    // NW = 8: the tile's epilogue operands, staged by the main loop's prologue (tools/gen_gemm_loop2.py stage_epilogue_operands()) and
    // read by the epilogue from LDS instead of from memory: lora_act_in [BM][32] fp32 (16-byte chunk c of row r at position c ^ (r & 7)),
    // lora_up [128][32] 16-bit (linear), bias [128] 16-bit
    static constexpr int STG_OFF = NSTAGE * STAGE_BYTES + EPI_BYTES + MAIL_BYTES;
    // rank > 32 (round 5): the region holds the tile's lora_up for EVERY rank instead -- [128 rows][R / 8 + 1 chunks of 16 bytes] (one pad chunk per row:
    // an odd pitch in 16-byte units, so the epilogue's 16-byte reads of one chunk over 32 rows spread over the banks), up to rank 160 = 43008 bytes,
    // staged by the C++ prologue of the loop call (stage_lu_all below); lora_act_in then comes through registers in batches of 64 ranks
    static constexpr int STG_LU = 32768, STG_BIAS = 43008;
    static constexpr int STG_LU_ALL_MAX_R = 160;
    static constexpr int STG_BYTES = NW == 8 ? STG_BIAS + 256 : 0;
    static constexpr int LDS_BYTES = STG_OFF + STG_BYTES; // 162064 (of 163840) | 80912 (two of them: 158 of 160 KiB)
    static constexpr int WG_PER_CU = NW == 8 ? 1 : 2;
};
// workspace header: 2048 int32 words.  Words [0, 1023): per-remainder-tile arrival counters of the stream-K split (at most
// 2 * CUs - 1 remainder tiles: the persistent grid never exceeds 512 workgroups); word 1023: the sticky error flag;
// words 1024 + 32 x (x = 0..7): ticket counter of XCD x's tile queue (dynamic schedule, one 128-byte line each);
// word 1024 + 32 * 8: workgroups that have left the queue (the last one clears the counters for the next launch)
constexpr int SK_HEADER_BYTES = 8192;
constexpr int SK_ERR_WORD = 1023;
constexpr int DQ_BASE = 1024, DQ_STRIDE = 32, DQ_DONE = DQ_BASE + 8 * DQ_STRIDE;
constexpr int SK_SPIN_LIMIT = 1 << 22;         // x (s_sleep 8 + one L2 round trip) ~ 1 s

struct GemmParams {
    const uint8_t *act;
    const uint8_t *wgt;
    const void *ascales;
    const void *wscales;
    const void *bias;
    const void *lora_act_in;
    const void *lora_up;
    void *out;
    uint8_t *qout;
    void *oscales;
    const void *next_smooth;
    const void *next_lora_down;
    void *lora_act_out;
    const void *norm_q;
    const void *norm_k;
    const float *rotary_emb;
    void *out_vt;            // RMSNORM_ROPE: transposed V output or NULL
    int ldvt;
    // grouped launch: row blocks >= split_row use the second weight set (same N, K, R, epilogue); 0x7fffffff = off
    const uint8_t *wgt2;
    const void *wscales2, *bias2, *lora_up2, *next_smooth2, *next_lora_down2, *norm_q2, *norm_k2;
    int split_row;
    int M, M_pad, N, K, R, R2, ldo;
    uint8_t *workspace;      // stream-K: [1024 int32 flags][2*G slabs of BM*BN fp32] or NULL
    long long workspace_bytes;
    int sk_gs;               // stream-K: workgroups sharing the remainder tiles (0 = whole tiles only); host heuristic
    int stagger;             // NW = 4: the second workgroup of a CU starts half a tile late
    int dynamic;             // NW = 4: tiles are drawn from per-XCD queues in the workspace instead of a fixed list per workgroup
    int *status;             // optional host-visible status word (svdq_gemm_args.status)
    float q_scale;           // RMSNORM_ROPE: factor of the Q third, applied before its rounding to 16-bit (svdq_gemm_args.q_scale; 1 = off)
    int stage_lora;          // NW = 8: rank 32, fp32 lora_act_in, 16-byte aligned operands: the loop stages lora_act_in / lora_up of a tile in LDS
    const void *lu_packed;   // solo-carry kernel (128 x 128 tiles, no LDS to stage lora_up in): lora_up as MFMA operand fragments as well (pack_lora_up_kernel)
    const void *ld_pre, *ld2_pre; // ABI 21: the caller's own fragment images of next_lora_down(2) (svdq_pack_lora_down) or NULL
    bool lu_pre;             // lu_packed is the caller's image (svdq_pack_lora_up): no pack launch
    const void *la_packed;   // all-rank kernels: lora_act_in as 16-bit MFMA operand fragments (pack_lora_act_kernel, in the workspace tail), scales applied
    int stage_lu_all;        // NW = 8, no carry: 32 < rank <= 160, fp32 lora_act_in, 16-byte aligned: the tile's lora_up (all ranks) is staged in LDS, lora_act_in comes
                             // through registers in batches of 64 ranks (every load of a batch in flight at once); value = ceil(65536 / (R / 8 + 1)), the divider of the gather
    int rowrun;              // NW = 8, GELU_QUANT: run length of the row-run schedule (GemmSchedule::init_runs; 0 = the plain schedule): the next layer's
                             // low-rank down projection accumulates in LDS over a workgroup's run of column tiles, one flush of atomics per run
    int solo_carry;          // host dispatch: GELU_QUANT with a next-layer low-rank branch of rank 48 .. 128 (fp32): 128 x 128 tiles, one workgroup per CU, the carry behind its ring
    void *act16_packed;      // split low-rank down (template SPLIT): the GELU_QUANT epilogue writes its 16-bit output as MFMA operand fragments here (workspace, behind the
                             // packed low-rank images) and lowrank_down_split_kernel contracts them with the next layer's down projection; the kernel itself sees R2 = 0
    int split_R2;            // ... the next layer's rank the host kept for that kernel
    int lora_fixed;          // host dispatch (template LAQ): lora_act_in and lora_act_out hold Q31.32 fixed point (svdq_amd.h "lora_act formats")
    float lora_scales[MAX_LORA_TILES];
    SVDQ_PROBE_PARAMS
};

__device__ __forceinline__ float gelu_tanh_f(float x) {
    // reference: gemm_utils.cuh:305-312: x * (0.5 + 0.5 * tanh.approx(u)), u = 0.79788456 * (x + 0.044715 x^3).
    // 0.5 + 0.5 tanh(u) = 1 / (1 + exp(-2u)): one hardware exp2 and one rcp (|error| ~1e-6, the class of
    // tanh.approx), 7 VALU operations per element; -2 u log2(e) = x * (A + B x^2).  exp2 overflow gives
    // x * rcp(inf) = 0 for very negative x, underflow gives x * 1 for very positive x.
    constexpr float A = -2.0f * 0.79788456f * 1.4426950408889634f, B = A * 0.044715f;
    const float t = x * __builtin_fmaf(x * x, B, A);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}
// reference: gemm_utils.cuh:290-303 (silu = x * rcp.approx(1 + ex2.approx(-x * log2 e))), applied by EpilogueSilu (gemm_base.cuh:783-792).  The same two hardware
// approximations here (v_exp_f32, v_rcp_f32: 1 ulp each), held to oracle.silu_envelope; rounds 1-5 computed an exact expf and an IEEE divide:
// 2.5-2.7 k static VALU per wave-tile against ~0.9 k for the default epilogue.
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }

typedef __attribute__((address_space(3))) void lds_void;

// lora_act_in for the all-rank kernels (rank 48 .. 160), packed once per launch: fp32 [M_pad][R] -> the 16-bit MFMA operand fragments the up projection
// consumes, [M_pad / 32 row tiles][R / 16 units][64 lanes][8 values] -- lane (row & 31, h) holds ranks 16 u + 8 h .. + 7 of its row, scaled per 16 ranks and
// rounded exactly as the epilogue's own conversion does (lora.cuh:145-158).  A row-per-lane read of the fp32 rows costs the epilogue the cache lines it
// touches, not its bytes: 32 rows x 128-byte lines per instruction for 16 bytes each, the same 256 KB per tile fetched by both column waves of a row block
// -- 17 k cycles per 256 x 128 tile at rank 128 against 2.5 k for the staged rank-32 operands (profiles/r5_gemm_phase_trace.txt).  Packed, a fragment is ONE
// coalesced 16-byte load per lane (1 KiB per wave instruction, every byte used) and the conversion leaves the epilogue.  The image lives in the tail of the
// caller's workspace (valid for launches ordered on one stream, like the stream-K slabs in front of it).
constexpr long long LA_PACK_BYTES = 16LL << 20; // M_pad * R * 2 bytes: 65536 rows at rank 128
template <int DT>
__global__ __launch_bounds__(256) void pack_lora_act_kernel(const float *__restrict__ la, typename Half<DT>::V8 *__restrict__ out, int R, int units, float16_scales sc) {
    using T = typename Half<DT>::T;
    using V8 = typename Half<DT>::V8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int unit = blockIdx.y * 4 + wave, rt = blockIdx.x;
    if (unit >= units) return;
    const int lr = lane & 31, h = lane >> 5;
    const float *src = la + (size_t)(rt * 32 + lr) * R + unit * 16 + h * 8;
    const v4f a = *reinterpret_cast<const v4f *>(src), b = *reinterpret_cast<const v4f *>(src + 4);
    const float s1 = sc.v[unit];
    V8 o;
    if (s1 == 1.0f) {
#pragma unroll
        for (int j = 0; j < 4; j++) { o[j] = f2h<T>(a[j]); o[4 + j] = f2h<T>(b[j]); }
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) { o[j] = f2h<T>(a[j] * s1); o[4 + j] = f2h<T>(b[j] * s1); }
    }
    out[((size_t)rt * units + unit) * 64 + lane] = o;
}

// the same for lora_up [N][R] 16-bit (a pure permutation): [N / 32 column tiles][R / 16 units][64 lanes][8 values], lane (n & 31, h) <- ranks 16 u + 8 h .. + 7
constexpr long long LU_PACK_BYTES = 8LL << 20;
template <int DT>
__global__ __launch_bounds__(256) void pack_lora_up_kernel(const typename Half<DT>::T *__restrict__ lu, typename Half<DT>::V8 *__restrict__ out, int R, int units) {
    using V8 = typename Half<DT>::V8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int unit = blockIdx.y * 4 + wave, ct = blockIdx.x;
    if (unit >= units) return;
    out[((size_t)ct * units + unit) * 64 + lane] = *reinterpret_cast<const V8 *>(lu + (size_t)(ct * 32 + (lane & 31)) * R + unit * 16 + (lane >> 5) * 8);
}

template <int DT, int FUSE, int NW, bool LAQ /* lora_act_in / lora_act_out are Q31.32 (deterministic mode) */,
          bool CARRY = false /* GELU_QUANT, NW = 8, fp32 lora_act_out of rank <= 32: the next layer's low-rank down projection accumulates in LDS */,
          bool RALL = false /* NW = 8, fp32 lora_act_in of rank 48 .. 160 (the r128 checkpoints, a runtime LoRA on top of rank 32): the tile's lora_up for EVERY rank is
                               staged in LDS, lora_act_in comes through registers in batches of 64 ranks (GemmParams::stage_lu_all).  A kernel of its own, so that the
                               rank-32 kernels of the step keep their instruction stream and register allocation to the bit */,
          bool HYB = false /* CARRY on 256 x 128 tiles with a next-layer rank beyond 32 (a runtime LoRA on fc2; rank 48 .. 80 checkpoints): the first 32 ranks go through
                              the carry, the passes behind them keep their per-tile atomics -- half (rank 48 / 64) or a quarter less of what the next loop's first
                              wait retires behind.  (The carry of more ranks does not fit: 256 rows x 48 ranks x 4 bytes > the 42 KiB staging region.) */,
          bool SPLIT = false /* all-rank GELU_QUANT kernel on 256 x 128 tiles whose next-layer low-rank down projection (rank 96 .. 160) runs as a kernel of its own
                                behind it (lowrank_down_split_kernel): the epilogue stores its 16-bit GELU output as MFMA fragments instead of contracting it per tile */>
__global__ __launch_bounds__(64 * NW, 2) void gemm_w4a4_kernel(const GemmParams p) {
    static_assert(!SPLIT || (RALL && NW == 8 && FUSE == SVDQ_FUSE_GELU_QUANT), "SPLIT: the all-rank GELU_QUANT kernel on 256 x 128 tiles");
    static_assert(!CARRY || FUSE == SVDQ_FUSE_GELU_QUANT, "the low-rank-down carry: GELU_QUANT");
    static_assert(!(CARRY && LAQ) || (NW == 8 && !HYB), "the fp32 carry under fixed-point accumulators (SVDQ_LORA_ACT_Q32_RUNS): 256 x 128 tiles, next-layer rank <= 32");
    static_assert(!RALL || (!LAQ && !CARRY), "the all-rank kernels: fp32 low-rank accumulators, no carry");
    // (RALL on 128 x 128 tiles: no LDS to stage lora_up in -- both low-rank operands come as packed MFMA fragments from the workspace tail, like the solo-carry kernel's)
    static_assert(!HYB || (CARRY && NW == 8), "HYB: a carry kernel on 256 x 128 tiles");
    using T = typename Half<DT>::T;
    using V8 = typename Half<DT>::V8;
    using G_ = Geo<NW>;
    constexpr int BM = G_::BM, NSTAGE = G_::NSTAGE, A_BYTES = G_::A_BYTES, STAGE_BYTES = G_::STAGE_BYTES;
    // CARRY on 128 x 128 tiles (round 5): next-layer ranks 48 .. 128.  ONE 4-wave workgroup per CU (one wave per SIMD: 90 % of the loop throughput of two
    // co-resident workgroups, profiles/r4_gemm_one_wave_per_simd.txt) has the CU's whole LDS: the ring of this geometry (78 KiB) + a carry of
    // [128 rows][128 ranks] fp32 = 64 KiB behind it -- four 32-rank slabs in the layout of the 256 x 128 kernel's one.  The 256-row tile's carry for rank 128
    // (128 KiB) fits nowhere, and per-tile fp32 atomics of 128 ranks cost as much as the whole rank-32 launch (profiles/r5_rank_ab.txt: 615 vs 318 us).
    constexpr int CARRY_SLABS = !CARRY ? 0 : NW == 8 ? 1 : 4;
    constexpr int CARRY_OFF = NW == 8 ? G_::STG_OFF : G_::LDS_BYTES;
    constexpr int LDS_TOTAL = G_::LDS_BYTES + (CARRY && NW == 4 ? CARRY_SLABS * BM * 32 * 4 : 0);
    __shared__ __attribute__((aligned(16))) uint8_t lds[LDS_TOTAL];

    const int tid = threadIdx.x;
    SVDQ_PROBE_BEGIN();
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, h = lane >> 5;
    const int KP = p.K / 128;
    const int TM = p.M_pad / BM, TN = p.N / BN, NT = TM * TN;

    // ---- persistent schedule ---------------------------------------------------------------------
    // One workgroup per CU slot walks a list of segments.  Tiles are enumerated in strips of 8 column
    // tiles (tile_coords) and workgroup b sits at position (b % 8) * (G / 8) + b / 8 -- workgroups are
    // dealt round-robin to the 8 XCDs, so the tiles an XCD works on at a time (32 of 256 x 128 or 64 of 128 x 128) are a
    // 1024 x 1024 patch of the output that shares its activation and weight panels in that XCD's L2.
    const int G = gridDim.x;
    const int pos = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    // row runs (GELU_QUANT, NW = 8): tile ids are row-major and a workgroup walks one run of consecutive column tiles
    const int rowrun = CARRY ? p.rowrun : 0;
    auto tile_coords = [&](int t, int &bm, int &bn) {
        if (CARRY && rowrun > 0) { bm = t / TN; bn = t - bm * TN; return; }
        const int strip = t / (8 * TM);
        const int w = min(8, TN - 8 * strip);
        const int r = t - strip * 8 * TM;
        bm = r / w;
        bn = 8 * strip + r % w;
    };

    v16f acc[2][2]; // [n tile][m tile]
    const v16f zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    // ---- per-wave DMA roles (constant over the kernel; tools/gen_gemm_loop2.py issues them) -------------
    //   NW = 8: every wave the three planes of A chunk `wave`;  waves 0..3: W planes 2w, 2w+1;  waves 4..7: W plane
    //           8 + (w-4);  wave 4 additionally the activation scale image (8 x 128 B), wave 5 the weight scale image
    //           (4 x 128 B, twice), waves 6, 7 their W plane once more (5 DMA instructions per wave and K-step).
    //   NW = 4: every wave the three planes of A chunk `wave` and the three planes of W chunk `wave`;  wave 0 the
    //           activation scale image (4 x 128 B, twice), wave 1 the weight scale image, waves 2, 3 their first W
    //           plane once more (7 per wave and K-step).
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds_base = (unsigned)(size_t)(lds_void *)lds;
    const int px1 = NW == 4 ? 3 * wv : (wv < 4 ? 2 * wv : 8 + (wv - 4));
    const int px2 = 2 * wv + 1; // NW = 8, waves 0..3 only
    const unsigned dX1 = lds_base + A_BYTES + (px1 / 3) * F6_CHUNK + (px1 % 3) * F6_PLANE;
    const int as_wave = NW == 4 ? 0 : 4, ws_wave = as_wave + 1;
    unsigned dX2 = dX1, iX2v = F6_CHUNK;   // default: the first X1 plane once more (same bytes to the same place)
    const unsigned offA = lane * 16, offX1 = lane * 16;
    unsigned offX2 = lane * 16;
    if (NW == 8 && wv < 4) {
        dX2 = lds_base + A_BYTES + (px2 / 3) * F6_CHUNK + (px2 % 3) * F6_PLANE;
    } else if (wv == as_wave) {
        offX2 = ((lane >> 3) & (BM / 32 - 1)) * KP * 128 + (lane & 7) * 16;
        dX2 = lds_base + A_BYTES + W_BYTES;
        iX2v = 128;
    } else if (wv == ws_wave) {
        offX2 = ((lane >> 3) & 3) * KP * 128 + (lane & 7) * 16;
        dX2 = lds_base + A_BYTES + W_BYTES + AS_BYTES;
        iX2v = 128;
    }
    const unsigned dA = lds_base + wv * F6_CHUNK;
    const unsigned in_la = lds_base + (wm * 2) * F6_CHUNK + lane * 16;
    const unsigned in_lw = lds_base + A_BYTES + (wn * 2) * F6_CHUNK + lane * 16;
    const unsigned in_lsa = lds_base + A_BYTES + W_BYTES + (wm * 2) * 128 + lr * 2;
    const unsigned in_lsw = lds_base + A_BYTES + W_BYTES + AS_BYTES + (wn * 2) * 128 + lr * 2;
    const int split_bm = p.split_row == 0x7fffffff ? 0x7fffffff : p.split_row / BM;
    // stream bases of (tile, first K-step)
    auto stream_ptrs = [&](int bm, int bn, int kp0, unsigned long long &a, unsigned long long &x1, unsigned long long &x2) {
        a = (unsigned long long)(p.act + ((size_t)(bm * (BM / 32) + wv) * KP + kp0) * F6_CHUNK);
        const uint8_t *wgt = bm >= split_bm ? p.wgt2 : p.wgt; // grouped launch: weight set of this row block
        const void *wscales = bm >= split_bm ? p.wscales2 : p.wscales;
        x1 = (unsigned long long)(wgt + ((size_t)(bn * 4 + px1 / 3) * KP + kp0) * F6_CHUNK + (px1 % 3) * F6_PLANE);
        if (NW == 8 && wv < 4) x2 = (unsigned long long)(wgt + ((size_t)(bn * 4 + px2 / 3) * KP + kp0) * F6_CHUNK + (px2 % 3) * F6_PLANE);
        else if (wv == as_wave) x2 = (unsigned long long)((const uint8_t *)p.ascales + ((size_t)(bm * (BM / 32)) * KP + kp0) * 128);
        else if (wv == ws_wave) x2 = (unsigned long long)((const uint8_t *)wscales + ((size_t)(bn * 4) * KP + kp0) * 128);
        else x2 = x1;
    };

    // ---- segments of this workgroup ----------------------------------------------------------------
    // F = NT / G whole tiles each (tile i*G + pos), then the R = NT % G remainder tiles.  With a workspace
    // the remainder is split along K ("stream-K"): its R*KP K-steps are dealt evenly to Gs workgroups, a
    // workgroup's run of K-steps is cut at tile boundaries into segments, the workgroup holding a tile's LAST
    // K-steps owns it: it adds the other segments' fp32 partial tiles (published through the workspace) and
    // runs the epilogue.  Without a workspace the first R workgroups take one remainder tile each.
    constexpr long long SLAB_BYTES = (long long)BM * BN * 4;
    const bool ws_ok = p.workspace != nullptr && p.workspace_bytes >= SK_HEADER_BYTES + 2LL * G * SLAB_BYTES;
    GemmSchedule sched;
    if (CARRY && rowrun > 0) sched.init_runs(TM, TN, KP, rowrun, pos);
    else sched.init(NT, KP, G, ws_ok ? p.sk_gs : 0, pos);
    const bool sk = sched.gs > 0;
    const int F = sched.F;
    typedef GemmSegment Seg;
    auto next_seg = [&](Seg &sg) -> bool { return sched.next(sg); };

    // ---- dynamic schedule (NW = 4, p.dynamic) -----------------------------------------------------------------------
    // Two workgroups share a CU's issue slots unevenly (the SIMD arbiter favours the older wave: one tenant finishes
    // 30-40 % earlier than the other, profiles/r3_gemm_placement.txt) and epilogues differ per tile, so a fixed tile list per
    // workgroup ends in a long tail of half-empty CUs.  Instead every workgroup DRAWS its tiles: the tile space is cut into
    // chunks of 64 consecutive tiles (an 8 x 8 patch = 1024 x 1024 outputs sharing 8 activation and 8 weight panels), every XCD
    // owns a contiguous run of chunks, a workgroup takes tickets from the queue of the XCD it runs on (XCC_ID) -- so the
    // tiles an XCD works on at a time still share its L2 -- and from the other XCDs' queues when its own is empty.  Thread 0
    // draws two tiles ahead (the main loop prefetches the next tile's first K-steps, so the next tile must be known at loop
    // entry); the draw is issued at the start of an epilogue and read at its end.  Results do not depend on who computes a tile.
    typedef __attribute__((address_space(1))) int gqint;
    typedef __attribute__((address_space(3))) int lds_int;
    lds_int *mail = (lds_int *)(lds + NSTAGE * STAGE_BYTES + G_::EPI_BYTES);
    const bool dyn = NW == 4 && p.dynamic != 0;
    const int NC = (NT + 63) / 64;
    const int my_xcd = dyn ? (int)(__builtin_amdgcn_s_getreg(3 << 11 | 20) & 7) : 0;
    auto draw = [&]() -> int { // thread 0 only: next tile of this workgroup or -1
        gqint *q = (gqint *)reinterpret_cast<int *>(p.workspace) + DQ_BASE;
        for (int a = 0; a < 8; a++) {
            const int x = (my_xcd + a) & 7;
            // XCD x's queue: the contiguous run of chunks [x * NC / 8, (x + 1) * NC / 8) -- consecutive chunks of a run lie in
            // the same strip of 8 column tiles, so its weight panels stay in the XCD's L2 from one chunk to the next
            const int c0 = x * NC / 8, c1 = (x + 1) * NC / 8;
            int len = (c1 - c0) * 64;
            if (c1 == NC) len -= NC * 64 - NT;            // the very last chunk may be partial
            if (len <= 0) continue;
            const int k = __hip_atomic_fetch_add(q + x * DQ_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (k < len) return c0 * 64 + k;
        }
        return -1;
    };
    auto leave_queue = [&]() { // thread 0, once, after its last draw: the last workgroup out clears the counters
        gqint *q = (gqint *)reinterpret_cast<int *>(p.workspace) + DQ_BASE;
        if (__hip_atomic_fetch_add(q + (DQ_DONE - DQ_BASE), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1) {
            for (int x = 0; x < 8; x++) __hip_atomic_store(q + x * DQ_STRIDE, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(q + (DQ_DONE - DQ_BASE), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto share = [&](int v) -> int { // thread 0's value to the whole workgroup
        __syncthreads();
        if (tid == 0) *mail = v;
        __syncthreads();
        return __builtin_amdgcn_readfirstlane(*mail);
    };
    int dq_pending = -1;   // the tile drawn for the iteration after next
    bool dq_left = false;

    if constexpr (NW == 4) {
        // De-phase the two workgroups of a CU.  They start together and do the same work per tile, so left alone they reach
        // their epilogues together and the matrix pipe idles exactly as with one big workgroup.  The workgroup whose waves
        // sit in the ODD wave slot of their SIMD (HW_ID.wave_id: the second tenant) starts half a tile late, once per launch.
        // (Purely a scheduling hint: results do not depend on which workgroup, if any, waits.)
        if (p.stagger && (F >= 2 || dyn)) {
            const unsigned slot = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4) & 1u; // HW_REG_HW_ID[3:0] = wave slot
            __attribute__((address_space(3))) unsigned *flag = (__attribute__((address_space(3))) unsigned *)(lds + NSTAGE * STAGE_BYTES);
            if (tid == 0) *flag = slot;
            __syncthreads();
            const unsigned late = __builtin_amdgcn_readfirstlane(*flag);
            __syncthreads();
            if (late) {
                for (int i = 0; i < KP; i += 8) __builtin_amdgcn_s_sleep(127); // ~ KP x 1000 cycles = half a tile's loop
            }
        }
    }

    // the workgroup's low-rank-down carry (CARRY kernels): [BM rows][32 ranks] fp32 in the lora_act_in slot of the staging region -- the loop
    // stages lora_up and bias only (stg_flags bit 2) and the epilogue loads its lora_act_in rows from memory.  The column waves of a tile add
    // their partial sums here; the carry goes to lora_act_out when the workgroup leaves the row block: with the row-run schedule once per run.
    typedef __attribute__((address_space(3))) float lds_float;
    lds_float *carry = (lds_float *)((lds_void *)lds) + CARRY_OFF / 4; // [slab][row / 4][32 ranks][row % 4]
    bool carry_dirty = false; // block-uniform
    // (round 6, tried: the wave's 64 rows of lora_act_in kept as 16-bit MFMA fragments -- 16 VGPRs -- across the ~20 column tiles a row-run spends in one row
    //  block, instead of 8 row-per-lane loads + 32 conversions per tile: the GELU_QUANT epilogue has no 16 registers to give, 436 B of scratch per lane.  Dropped.)
    if constexpr (CARRY) {
        typedef __attribute__((address_space(3))) v4f lds_v4f;
        const v4f z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < CARRY_SLABS * BM * 32 / 4 / (64 * NW); i++) ((lds_v4f *)carry)[i * 64 * NW + tid] = z4;
        __syncthreads();
    }

    unsigned ring = 0, npre = 0, landed = 0;
    unsigned long long pA = 0, pX1 = 0, pX2 = 0;
    int bm = 0, bn = 0;
    Seg cur{0, 0, 0, 0}, nxt{0, 0, 0, 0};
    bool have;
    if (dyn) {
        int t0 = -1, t1 = -1;
        if (tid == 0) {
            t0 = draw();
            t1 = t0 >= 0 ? draw() : -1;
            if (t1 < 0) { leave_queue(); dq_left = true; }
        }
        t0 = share(t0);
        dq_pending = share(t1);
        have = t0 >= 0;
        cur = Seg{t0, 0, KP, 0};
    } else {
        have = next_seg(cur);
    }
    if (have) {
        tile_coords(cur.tile, bm, bn);
        stream_ptrs(bm, bn, cur.kp0, pA, pX1, pX2);
    }
    while (have) {
        const int kp0 = cur.kp0, kp1 = cur.kp1;
        const int m0 = bm * BM, n0 = bn * BN;
        // the next segment (its first operands are prefetched by the tail of this one)
        bool have_next;
        if (dyn) {
            have_next = dq_pending >= 0;
            nxt = Seg{dq_pending, 0, KP, 0};
        } else {
            have_next = next_seg(nxt);
        }
        int nbm = 0, nbn = 0;
        unsigned ncnt = 0;
        unsigned long long nA = 0, nX1 = 0, nX2 = 0;
        if (have_next) {
            tile_coords(nxt.tile, nbm, nbn);
            stream_ptrs(nbm, nbn, nxt.kp0, nA, nX1, nX2);
            ncnt = nxt.kp1 - nxt.kp0;
        }

        SVDQ_PROBE_STAMP(0);
        bool stg_counted = false;
        {
            // ---- hand-scheduled main loop (generated inline asm, every operand pinned to a physical register;
            //      DESIGN.md "Main loop"; tools/gen_gemm_loop2.py) ---------------------------------------------
            const unsigned kp_s = kp1 - kp0;
#define SVDQ_LOOP_CLOBBER_V                                                                                            \
                  "v64", "v65", "v66", "v67", "v68", "v69", "v70",     \
                  "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", \
                  "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101",       \
                  "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",  \
                  "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129",  \
                  "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143",  \
                  "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157",  \
                  "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171",  \
                  "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185",  \
                  "v186", "v187", "v188", "v189", "v190", "v191", "v217", "v218", "v219", "v220"
            // (round 6: v192 .. v209 left the loop's register plan -- two-register scale tuples, no MX exponents: tools/gen_gemm_loop2.py -- and are the compiler's
            //  across the loop)
            // buffer resources of the three operand streams (raw buffers, no range check: the loop never issues a
            // DMA beyond the last K-step of the workgroup's last segment)
            auto srd = [](unsigned long long ptr) {
                v4i r;
                r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)ptr);
                r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(ptr >> 32));
                r[2] = -1;
                r[3] = 0x00020000;
                return r;
            };
            v4i rA = srd(pA), rX1 = srd(pX1), rX2 = srd(pX2);
            const unsigned phaseB = wv >= NW / 2 ? 1u : 0u; // generator option "dph" only
            // NW = 8: the loop's prologue stages this tile's epilogue operands in LDS (Geo::STG_OFF) when this segment ends the tile,
            // i.e. runs its epilogue: bit 0 = lora_act_in (this wave's rows 32 w .. 32 w + 31: 4 KiB) + lora_up (its rows 16 w ..: 1 KiB)
            // -- rank 32, fp32 accumulators, 16-byte aligned (p.stage_lora) --, bit 1 = bias (wave 0: 256 B); bits 8..10 = wave
            unsigned stg_flags = 0;
            const char *stg_la = (const char *)p.wgt, *stg_lu = stg_la, *stg_b = stg_la;
            if (NW == 8 && kp1 == KP) {
                if (!LAQ && !RALL && p.stage_lora) {
                    stg_flags |= CARRY ? 5u : 1u;
                    stg_la = (const char *)p.lora_act_in + ((size_t)m0 + 32 * wv) * 128;
                    stg_lu = (const char *)(bm >= split_bm ? p.lora_up2 : p.lora_up) + ((size_t)n0 + 16 * wv) * 64;
                }
                if (p.bias) {
                    stg_flags |= 2u;
                    stg_b = (const char *)(bm >= split_bm ? p.bias2 : p.bias) + (size_t)n0 * 2;
                }
                stg_flags |= (unsigned)wv << 8;
            }
            // DMAs this loop call issues behind the staging ones (5 per K-step while step < tot - NSTAGE): with 15 or more of them the
            // epilogue's "s_waitcnt vmcnt(15)" proves the staged operands landed (vmcnt retires in order) without waiting for the next
            // tile's prefetch; a shorter call simply drains
            const int stg_issues = min((int)kp_s, (int)(kp_s + ncnt) - NSTAGE);
            stg_counted = stg_issues >= 3;
            const unsigned stg_lds = lds_base + G_::STG_OFF;
            if constexpr (RALL && NW == 8) {
                if (kp1 == KP) {
                    // rank > 32: the tile's lora_up rows n0 .. n0 + 127, ALL ranks (256 R contiguous bytes), go to the staging region by LDS-DMA from
                    // here -- older than every DMA of the loop, landing under it like the generated prologue's pieces.  LDS image: row r, 16-byte
                    // chunk c at ((r (C + 1) + c) * 16), C = R / 8: the odd pitch spreads the epilogue's reads (one chunk over 32 rows) over the banks.
                    // An LDS-DMA writes 64 consecutive 16-byte slots: lane i of piece q fills slot 64 q + i, i.e. fetches chunk (slot % (C + 1)) of row
                    // slot / (C + 1) (the pad slot fetches chunk 0 again).  Waves take pieces w, w + 8, ...; the barrier keeps the previous tile's readers
                    // of the region out of the way.
                    __syncthreads();
                    const unsigned C1 = (unsigned)p.R / 8u + 1u, pieces = 2u * C1, magic = (unsigned)p.stage_lu_all;
                    unsigned lane_s = lane; // (through an empty asm statement: the per-lane gather offsets are loop-invariant, and hoisted out of the tile loop they
                    asm volatile("" : "+v"(lane_s)); //  would live -- spilled -- across the main loop's asm block, which leaves ~30 registers free)
                    const char *lu_tile = (const char *)(bm >= split_bm ? p.lora_up2 : p.lora_up) + (size_t)n0 * (unsigned)p.R * 2u;
                    const unsigned long long lu_tile_u = (unsigned long long)lu_tile;
#pragma unroll
                    for (unsigned i = 0; i < (2u * (G_::STG_LU_ALL_MAX_R / 8 + 1) + 7u) / 8u; i++) {
                        const unsigned q = (unsigned)wv + 8u * i;
                        if (q < pieces) { // wave-uniform
                            const unsigned slot = 64u * q + lane_s, row = (slot * magic) >> 16, c = slot - row * C1;
                            const unsigned voff = row * (unsigned)p.R * 2u + (c == C1 - 1u ? 0u : c * 16u);
                            const unsigned dst = __builtin_amdgcn_readfirstlane(stg_lds + 1024u * q);
                            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(lu_tile_u) : "memory", "m0");
                        }
                    }
                }
            }
            const unsigned long long stg_la_u = (unsigned long long)stg_la, stg_lu_u = (unsigned long long)stg_lu, stg_b_u = (unsigned long long)stg_b;
#define SVDQ_LOOP2_OPERANDS                                                                                             \
                : "={v[0:15]}"(acc[0][0]), "={v[16:31]}"(acc[0][1]), "={v[32:47]}"(acc[1][0]), "={v[48:63]}"(acc[1][1]),         \
                  "+{s[72:75]}"(rA), "+{s[76:79]}"(rX1), "+{s[80:83]}"(rX2)                                                      \
                : "{v210}"(in_la), "{v211}"(in_lw), "{v212}"(in_lsa), "{v213}"(in_lsw), "{v214}"(offA), "{v215}"(offX1),         \
                  "{v216}"(offX2), "{s46}"(kp_s), "{s47}"(dA), "{s48}"(dX1), "{s49}"(dX2), "{s50}"(iX2v), "{s52}"(phaseB),         \
                  "{s51}"(stg_flags), "{s54}"(stg_lds), "{s[88:89]}"(stg_la_u), "{s[90:91]}"(stg_lu_u), "{s[92:93]}"(stg_b_u),                  \
                  "{s58}"(ring),                                                                                                   \
                  "{s59}"(npre), "{s60}"(ncnt), "{s[62:63]}"(nA), "{s[64:65]}"(nX1), "{s[66:67]}"(nX2), "{s68}"(landed)            \
                : "memory", "scc", "m0", "s53", "s55", "s56", "s57", "s61", "s69", "s84", "s85", "s86", "s87", SVDQ_LOOP_CLOBBER_V
            if constexpr (NW == 8 && DT == SVDQ_BF16) {
                asm volatile(
#include SVDQ_LOOP_INC_8_BF16
                    SVDQ_LOOP2_OPERANDS);
            } else if constexpr (NW == 8) {
                asm volatile(
#include SVDQ_LOOP_INC_8_FP16
                    SVDQ_LOOP2_OPERANDS);
            } else if constexpr (DT == SVDQ_BF16) {
                asm volatile(
#include SVDQ_LOOP_INC_4_BF16
                    SVDQ_LOOP2_OPERANDS);
            } else {
                asm volatile(
#include SVDQ_LOOP_INC_4_FP16
                    SVDQ_LOOP2_OPERANDS);
            }
#undef SVDQ_LOOP2_OPERANDS
#undef SVDQ_LOOP_CLOBBER_V
            ring = (ring + (kp_s % NSTAGE) * STAGE_BYTES) % (NSTAGE * STAGE_BYTES); // stage of the next segment's K-step 0
            npre = min((unsigned)NSTAGE, ncnt);
            // The loop returns with the next segment's first K-steps still in flight (no vmcnt drain: their latency
            // overlaps the epilogue's own loads).  An epilogue that waits on a global load of its own -- younger than
            // those DMAs, vmcnt retires in order -- proves them landed; otherwise the next prologue waits itself.
            // (NW = 8: the epilogue starts with s_waitcnt vmcnt(0) + a barrier before it reads the staged operands)
            landed = ((NW == 8 || p.bias || p.R > 0) && kp1 == KP) ? npre : 0;
            pA = nA; pX1 = nX1; pX2 = nX2; // the next segment's bases (set by stream_ptrs above)
        }
        SVDQ_PROBE_STAMP(1);
        int dq_drawn = -1;
        if (dyn && have_next && tid == 0 && !dq_left) { // the tile after next: requested now, consumed behind the epilogue
            dq_drawn = draw();
            if (dq_drawn < 0) { leave_queue(); dq_left = true; }
        }

        // ---- stream-K: publish or collect partial tiles -----------------------------------------------
        bool run_epilogue = true;
        if (sk && (kp0 > 0 || kp1 < KP)) {
            typedef __attribute__((address_space(1))) int gint;
            gint *flags = (gint *)reinterpret_cast<int *>(p.workspace); // explicit global address space: no flat aperture checks
            typedef __attribute__((address_space(1))) float gfloat;
            typedef __attribute__((address_space(1))) v4f gv4f;
            gfloat *slabs = (gfloat *)reinterpret_cast<float *>(p.workspace + SK_HEADER_BYTES);
            const int trel = cur.tile - F * G;                 // remainder tile index = flag index
            if (kp1 < KP) {
                // not the owner: store the raw fp32 accumulators (16 coalesced 1 KiB wave stores per wave),
                // make them visible at agent scope, then bump the tile's arrival counter
                gfloat *slab = slabs + (size_t)sched.slot(cur) * (BM * BN);
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    v4f v = {acc[j >> 3][(j >> 2) & 1][(j & 3) * 4 + 0], acc[j >> 3][(j >> 2) & 1][(j & 3) * 4 + 1],
                             acc[j >> 3][(j >> 2) & 1][(j & 3) * 4 + 2], acc[j >> 3][(j >> 2) & 1][(j & 3) * 4 + 3]};
                    *(gv4f *)(slab + ((size_t)(j * NW + wave) * 64 + lane) * 4) = v;
                }
                __syncthreads();
                if (tid == 0) {
                    // (round 4: write-through `sc1` / `sc0 sc1` stores + vmcnt(0) instead of the release fence -- MI355X_MICROARCH.md "publish-large" --
                    //  were tried here: 1-2 % faster, and the owner read stale slab lines in every run; profiles/r4_gemm_streamk_publish.txt)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_fetch_add(flags + trel, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                run_epilogue = false;
            } else {
                // owner of the tile: wait for the segments [ut0, ut0 + kp0) of the workgroups before us
                const int first = sched.first_contributor(cur);
                const int needed = pos - first;
                if (tid == 0) {
                    // Bounded wait (~1 s): a missing arrival can only come from a broken contract (the workspace shared by
                    // launches in flight on two streams, not zero-filled, or a grid that is not co-resident).  Then give up
                    // instead of hanging the GPU: raise the sticky error word (checked by the host at the next launch on this
                    // workspace and by svdq_gemm_workspace_status()); this tile's result is garbage.
                    int spins = 0;
                    while (__hip_atomic_load(flags + trel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < needed && ++spins < SK_SPIN_LIMIT)
                        __builtin_amdgcn_s_sleep(8);
                    if (spins >= SK_SPIN_LIMIT) {
                        __hip_atomic_store(flags + SK_ERR_WORD, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (p.status) __hip_atomic_store(p.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    __hip_atomic_store(flags + trel, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next launch
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                for (int q = first; q < pos; q++) {
                    const gfloat *slab = slabs + (size_t)sched.contributor_slot(cur, q) * (BM * BN);
                    // eight 16-byte loads in flight, then their adds (round 6: written as load / add pairs the compiler kept ONE 4-register temporary and put an
                    // s_waitcnt vmcnt(0) behind every load -- sixteen dependent fabric round trips per contributor, ~10-15 k cycles of an owner's tile; the
                    // loop's P / S / fragment registers are dead here, and the scheduling barriers keep the batch together).  Same adds in the same order.
#pragma unroll
                    for (int jb = 0; jb < 16; jb += 8) {
                        v4f t[8];
#pragma unroll
                        for (int j = 0; j < 8; j++) t[j] = __builtin_nontemporal_load((const gv4f *)(slab + ((size_t)((jb + j) * NW + wave) * 64 + lane) * 4));
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < 8; j++)
#pragma unroll
                            for (int e = 0; e < 4; e++) acc[(jb + j) >> 3][((jb + j) >> 2) & 1][((jb + j) & 3) * 4 + e] += t[j][e];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }

        if (run_epilogue) {
        // ------------------------------------------------------------------ epilogue
        // lane owns rows m = mw0 + 32*mi + lr and columns n = nw0 + 32*ni + 8*c + 4*h + e  (r = 4c + e)
        const int nw0 = n0 + wn * 64;
        const int mw0 = m0 + wm * 64;

        // bias (reference EpilogueBias, gemm_base.cuh:710-781) and low-rank up projection (reference EpilogueLoraUp,
        // lora.cuh:110-241): fp32 activations are scaled per 16 ranks and rounded to 16-bit (:145-158), multiplied on the
        // matrix cores and accumulated in fp32 -- here straight onto the GEMM accumulators.
        // Every operand of the tile is requested BEFORE the first one is consumed (one memory round trip per tile
        // instead of one per dependent stage: ~3.8 us -> ~2 us of exposed epilogue at K = 3072).  The arithmetic and its
        // order are unchanged: acc + bias, then one MFMA per 16 ranks in ascending rank order.
        const bool use_bias = p.bias != nullptr;
        const int Rr = p.R;
        // uniform base pointer + ONE 32-bit per-lane byte offset per tensor: the loads below differ by immediates only
        // (global_load saddr + voffset form), which keeps the address registers of 20 loads down to three.  The offsets
        // are derived from a lane id that passes through an empty asm statement AFTER the main loop: otherwise the
        // compiler hoists all the address arithmetic above the loop's asm block, where only ~30 VGPRs are free, and
        // spills it to scratch.
        unsigned lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const unsigned lr_e = lane_e & 31, h_e = lane_e >> 5;
        // The bias rides on the matrix pipe as well: D[n][m] += bias[n] * 1 is an MFMA whose weight-side operand holds bias[n] in ONE
        // k-slot and whose activation-side operand holds 1.0 there (exact: one non-zero product per output).  Column tile ni uses
        // k-slot 0 of the lanes of half ni, so that the wave needs ONE coalesced 16-bit load (lane (lr, h) <- bias[nw0 + 32 h + lr])
        // where the C-layout add needed eight 8-byte loads, 32 conversions and 64 v_add per wave and tile.
        // NW = 8: bias, lora_act_in and lora_up of the tile were staged in LDS by the main loop's prologue (Geo::STG_OFF) and landed under
        // the loop: the first phase of the epilogue reads LDS instead of waiting on ~13 global loads per wave with the matrix pipe idle.
        typedef __attribute__((address_space(3))) const uint8_t lds_cbytes;
        lds_cbytes *stg = (lds_cbytes *)((lds_void *)lds) + G_::STG_OFF;
        bool staged_l = false; // block-uniform
        if constexpr (NW == 8) {
            staged_l = !LAQ && !RALL && p.stage_lora;
            if (stg_counted) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads(); // every wave's pieces are in LDS
        }
        unsigned bias_bits = 0; // (zero-initialised: conditionally loaded values must not look loop-carried to the register allocator)
        const char *b_base = (const char *)(bm >= split_bm ? p.bias2 : p.bias);
        const unsigned b_off = (unsigned)(nw0 + h_e * 4) * 2u; // (the fused epilogues' per-column vectors, C layout)
        if (use_bias) {
            if constexpr (NW == 8) bias_bits = *reinterpret_cast<__attribute__((address_space(3))) const uint16_t *>(stg + G_::STG_BIAS + (unsigned)(wn * 64 + h_e * 32 + lr_e) * 2u);
            else bias_bits = *reinterpret_cast<const uint16_t *>(b_base + (unsigned)(nw0 + h_e * 32 + lr_e) * 2u);
        }
        const char *la_base = (const char *)p.lora_act_in;
        const char *lu_base = (const char *)(bm >= split_bm ? p.lora_up2 : p.lora_up);
        // lora_act_in: fp32 [M_pad][R], or (LAQ) Q31.32 fixed point in int64 [M_pad][R] -- the order-independent accumulation
        // format of the deterministic mode (svdq_amd.h "lora_act formats"), converted to fp32 on use
        constexpr unsigned LA_B = LAQ ? 8u : 4u;
        const unsigned la_off = ((unsigned)(mw0 + lr_e) * (unsigned)Rr + h_e * 8) * LA_B;
        const unsigned lu_off = ((unsigned)(nw0 + lr_e) * (unsigned)Rr + h_e * 8) * 2u;
        const unsigned la_mi = 32u * Rr * LA_B, lu_ni = 32u * Rr * 2u; // byte strides of the second row tile / column tile
        struct LaRegs { v4i q[2][LAQ ? 4 : 2]; }; // the lane's 8 ranks of both row tiles: 8 fp32 or 8 int64 each
        auto load_la = [&](int rc, LaRegs &t) {
#pragma unroll
            for (int mi = 0; mi < 2; mi++)
#pragma unroll
                for (int j = 0; j < (LAQ ? 4 : 2); j++) t.q[mi][j] = *reinterpret_cast<const v4i *>(la_base + (la_off + mi * la_mi + rc * LA_B + j * 16));
        };
        auto load_lu = [&](int rc, V8 (&u)[2]) {
#pragma unroll
            for (int ni = 0; ni < 2; ni++) u[ni] = *reinterpret_cast<const V8 *>(lu_base + (lu_off + ni * lu_ni + rc * 2));
        };
        // the same operands from the staging region (rank 32, fp32): row r of the tile holds 16-byte chunk c (ranks 4c .. 4c + 3) at
        // position c ^ (r & 7); the lane's 8 ranks rc + 8h .. are chunks rc / 4 + 2h and the next one
        auto staged_la = [&](int rc, LaRegs &t) {
            if constexpr (!LAQ) {
#pragma unroll
                for (int mi = 0; mi < 2; mi++)
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        const unsigned row = (unsigned)(wm * 64 + mi * 32) + lr_e, c = (unsigned)(rc / 4 + j) + 2u * h_e;
                        t.q[mi][j] = *reinterpret_cast<__attribute__((address_space(3))) const v4i *>(stg + row * 128u + ((c ^ (lr_e & 7u)) * 16u));
                    }
            }
        };
        auto staged_lu = [&](int rc, V8 (&u)[2]) {
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
                u[ni] = *reinterpret_cast<__attribute__((address_space(3))) const V8 *>(stg + G_::STG_LU + ((unsigned)(wn * 64 + ni * 32) + lr_e) * 64u + (unsigned)rc * 2u + h_e * 16u);
        };
        auto lora_mfma = [&](int rc, const LaRegs &t, const V8 (&u)[2], auto &&after_convert) {
            const float sc = p.lora_scales[rc >> 4];
            V8 la[2];
            auto convert = [&](auto unit) { // unit: the scale of these 16 ranks is 1 (the default; block-uniform): v * 1 = v, no multiply
#pragma unroll
                for (int mi = 0; mi < 2; mi++)
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        // element j of the lane's 8 ranks: fp32 as stored, or Q31.32 (low word, high word) -> fp32
                        float v;
                        if constexpr (LAQ) v = q32_to_float(t.q[mi][j >> 1][2 * (j & 1)], t.q[mi][j >> 1][2 * (j & 1) + 1]);
                        else { const int w = t.q[mi][j >> 2][j & 3]; v = __builtin_bit_cast(float, w); }
                        la[mi][j] = f2h<T>(decltype(unit)::value ? v : v * sc);
                    }
            };
            if (sc == 1.0f) convert(std::true_type{});
            else convert(std::false_type{});
            after_convert(); // (the all-rank kernel re-uses t's registers for the unit 64 ranks further on)
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int mi = 0; mi < 2; mi++) acc[ni][mi] = Half<DT>::mfma32(u[ni], la[mi], acc[ni][mi]);
        };
        auto no_hook = []() {};
        // rank > 32 with the tile's lora_up staged for every rank (stage_lu_all): lora_act_in comes through registers in batches of 64 ranks -- the 8 (LAQ: -)
        // loads of a batch are all in flight at once, two batches deep where the registers allow it (one round trip for rank <= 128 instead of one per
        // 16 ranks) -- and the up-projection fragments are 16-byte LDS reads.  Same arithmetic, same order: one MFMA per 16 ranks, ascending.
        struct LaBatch { LaRegs x[4]; };
        // (every load of a batch is issued unconditionally -- a unit beyond the rank re-reads the last one -- so that the batch registers are plainly
        //  defined values: conditionally loaded ones look loop-carried to the register allocator, which then keeps them alive, i.e. spilled, across
        //  the main loop's asm block)
        auto issue_batch = [&](int rc0, LaBatch &b) {
#pragma unroll
            for (int i = 0; i < 4; i++) load_la(min(rc0 + 16 * i, Rr - 16), b.x[i]);
        };
        auto staged_lu_all = [&](int rc, V8 (&u)[2]) {
            const unsigned C1 = (unsigned)Rr / 8u + 1u;
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
                u[ni] = *reinterpret_cast<__attribute__((address_space(3))) const V8 *>(stg + ((((unsigned)(wn * 64 + ni * 32) + lr_e) * C1 + (unsigned)rc / 8u + h_e) * 16u));
        };
        LaRegs x0 = {}, x1 = {};
        V8 u0[2] = {}, u1[2] = {};
        // the solo-carry kernel (128 x 128 tiles) with both low-rank operands packed as MFMA fragments (block-uniform): no natural-order loads at all
        constexpr bool SOLO = NW == 4 && (CARRY || RALL); // (RALL on 128 x 128 tiles: always packed)
        bool solo_pk = false;
        if constexpr (SOLO) solo_pk = RALL || (p.lu_packed != nullptr && p.la_packed != nullptr);
        if constexpr (RALL && NW == 8) {
        } else if (SOLO && solo_pk) {
        } else if (staged_l) {
            if constexpr (CARRY) { load_la(0, x0); load_la(16, x1); }  // (the region's lora_act_in slot holds the carry)
            else { staged_la(0, x0); staged_la(16, x1); }
            staged_lu(0, u0); staged_lu(16, u1);
        }
        else {
            if (Rr > 0) { load_la(0, x0); load_lu(0, u0); }
            if (Rr > 16) { if constexpr (!LAQ) load_la(16, x1); load_lu(16, u1); } // (LAQ: twice the registers per value -- ranks 16..31 follow the first MFMA)
        }
        // GELU_QUANT: the next layer's smoothing factors and the first 32 ranks of its low-rank down projection ride on the
        // same round trip (they are consumed ~2000 instructions later, behind the GELU and the requantisation)
        u16x4 nsv[2][4] = {}, ldw[2][2][2] = {};
        auto load_next_params = [&]() {
        if constexpr (FUSE == SVDQ_FUSE_GELU_QUANT) {
            const char *ns_base = (const char *)(bm >= split_bm ? p.next_smooth2 : p.next_smooth);
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int c = 0; c < 4; c++) nsv[ni][c] = *reinterpret_cast<const u16x4 *>(ns_base + b_off + (ni * 32 + c * 8) * 2);
            if (!SPLIT && p.R2 > 0 && (int)lr_e < p.R2) {
                const char *ld_base = (const char *)(bm >= split_bm ? p.next_lora_down2 : p.next_lora_down); // rank-major [R2][N]
                const unsigned ld_off = (lr_e * (unsigned)p.N + nw0 + h_e * 4) * 2u;
#pragma unroll
                for (int ni = 0; ni < 2; ni++)
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        ldw[ni][q][0] = *reinterpret_cast<const u16x4 *>(ld_base + ld_off + (ni * 32 + q * 16) * 2);
                        ldw[ni][q][1] = *reinterpret_cast<const u16x4 *>(ld_base + ld_off + (ni * 32 + q * 16 + 8) * 2);
                    }
            }
        }
        };
        if constexpr (!(RALL && NW == 8)) load_next_params(); // (the all-rank kernel's ring of low-rank activations needs the registers first: it asks behind the up projection)
        auto apply_bias = [&]() {
            if (use_bias) {
#pragma unroll
                for (int ni = 0; ni < 2; ni++) {
                    V8 bw, one;
#pragma unroll
                    for (int j = 0; j < 8; j++) { bw[j] = (T)0.f; one[j] = (T)0.f; }
                    bw[0] = hfrom<T>((uint16_t)(h_e == (unsigned)ni ? bias_bits : 0u));
                    one[0] = h_e == (unsigned)ni ? (T)1.0f : (T)0.f;
#pragma unroll
                    for (int mi = 0; mi < 2; mi++) acc[ni][mi] = Half<DT>::mfma32(bw, one, acc[ni][mi]);
                }
            }
        };
        if constexpr (RALL && NW == 8) {
            // the low-rank activations arrive as packed 16-bit MFMA fragments (pack_lora_act_kernel): one coalesced 16-byte load per lane, row tile and
            // 16-rank unit.  A ring of eight units (64 VGPRs): rank <= 128 is requested whole, up front (above the wait for the staged operands); beyond,
            // a unit's registers are re-used for the unit 128 ranks further on as soon as its MFMAs are issued.
            const V8 *lap = (const V8 *)p.la_packed;
            const unsigned units = (unsigned)Rr / 16u;
            const unsigned lap0 = (((unsigned)(mw0 >> 5)) * units) * 64u + lane_e; // + (mi * units + unit) * 64
            V8 ring[8][2];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const unsigned un = min((unsigned)i, units - 1u); // (unconditional loads: a unit beyond the rank re-reads the last one)
#pragma unroll
                for (int mi = 0; mi < 2; mi++) ring[i][mi] = lap[lap0 + ((unsigned)mi * units + un) * 64u];
            }
            apply_bias();
            for (int rc0 = 0; rc0 < Rr; rc0 += 128) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int rc = rc0 + 16 * i;
                    if (rc < Rr) { // block-uniform
                        V8 u[2];
                        staged_lu_all(rc, u);
#pragma unroll
                        for (int ni = 0; ni < 2; ni++)
#pragma unroll
                            for (int mi = 0; mi < 2; mi++) acc[ni][mi] = Half<DT>::mfma32(u[ni], ring[i][mi], acc[ni][mi]);
                        if (rc + 128 < Rr) {
#pragma unroll
                            for (int mi = 0; mi < 2; mi++) ring[i][mi] = lap[lap0 + ((unsigned)mi * units + (unsigned)(rc + 128) / 16u) * 64u];
                        }
                    }
                }
            }
            load_next_params();
        } else if (SOLO && solo_pk) {
            // eight units in flight (128 VGPRs: two row-tile fragments of lora_act_in + two column-tile fragments of lora_up each -- one wave per SIMD has the
            // registers): rank <= 128 is requested whole, above the bias MFMAs; every load is one coalesced 16 bytes per lane.  (Four in flight: 9.6 k cycles
            // for this phase at rank 128 -- one wave per SIMD has nobody to hide a unit's load latency behind, profiles/r5_gemm_phase_trace_final.txt.)
            const V8 *lap = (const V8 *)p.la_packed, *lup = (const V8 *)p.lu_packed;
            const unsigned units = (unsigned)Rr / 16u;
            const unsigned lap0 = (((unsigned)(mw0 >> 5)) * units) * 64u + lane_e, lup0 = (((unsigned)(nw0 >> 5)) * units) * 64u + lane_e;
            constexpr int RING = CARRY ? 8 : 4; // (two workgroups per CU -- the all-rank kernel of this geometry -- share the SIMD's registers: four units, 64 VGPRs)
            V8 ra[RING][2], ru[RING][2];
#pragma unroll
            for (int i = 0; i < RING; i++) {
                const unsigned un = min((unsigned)i, units - 1u);
#pragma unroll
                for (int t = 0; t < 2; t++) { ra[i][t] = lap[lap0 + ((unsigned)t * units + un) * 64u]; ru[i][t] = lup[lup0 + ((unsigned)t * units + un) * 64u]; }
            }
            apply_bias();
            for (int rc0 = 0; rc0 < Rr; rc0 += 16 * RING) {
#pragma unroll
                for (int i = 0; i < RING; i++) {
                    const int rc = rc0 + 16 * i;
                    if (rc < Rr) { // block-uniform
#pragma unroll
                        for (int ni = 0; ni < 2; ni++)
#pragma unroll
                            for (int mi = 0; mi < 2; mi++) acc[ni][mi] = Half<DT>::mfma32(ru[i][ni], ra[i][mi], acc[ni][mi]);
                        if (rc + 16 * RING < Rr) {
                            const unsigned un = (unsigned)(rc + 16 * RING) / 16u;
#pragma unroll
                            for (int t = 0; t < 2; t++) { ra[i][t] = lap[lap0 + ((unsigned)t * units + un) * 64u]; ru[i][t] = lup[lup0 + ((unsigned)t * units + un) * 64u]; }
                        }
                    }
                }
            }
        } else {
        apply_bias();
        if (Rr > 0) lora_mfma(0, x0, u0, no_hook);
        if (Rr > 16) {
            if constexpr (LAQ) { load_la(16, x0); lora_mfma(16, x0, u1, no_hook); }
            else lora_mfma(16, x1, u1, no_hook);
        }
        // beyond rank 32 without the all-rank kernel (the 128 x 128 geometry, the fixed-point accumulator format, the carry kernel, unaligned operands,
        // rank > 160).  128 x 128 geometry: 32 ranks per round trip, every load of the pair in flight before the first MFMA; elsewhere 16
        if constexpr (NW == 4 && !LAQ) {
            for (int rc = 32; rc < Rr; rc += 32) {
                load_la(rc, x0);
                load_lu(rc, u0);
                if (rc + 16 < Rr) { load_la(rc + 16, x1); load_lu(rc + 16, u1); }
                lora_mfma(rc, x0, u0, no_hook);
                if (rc + 16 < Rr) lora_mfma(rc + 16, x1, u1, no_hook);
            }
        } else {
            for (int rc = 32; rc < Rr; rc += 16) {
                load_la(rc, x0);
                load_lu(rc, u0);
                lora_mfma(rc, x0, u0, no_hook);
            }
        }
        }
        // NW = 8: nothing of this epilogue has gone to memory yet -- whatever is outstanding is the next tile's prefetch (and a fused
        // epilogue's own parameter loads): waiting here proves the prefetch landed (`landed` above) so that the next loop call need not
        // drain this epilogue's stores
        // (GELU_QUANT has loads of its own in flight since the top of the epilogue -- the next layer's smoothing factors and low-rank
        //  down rows -- and waits for them further down: that wait is the proof, as it always was for this epilogue)
        if constexpr (NW == 8 && FUSE != SVDQ_FUSE_GELU_QUANT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SVDQ_PROBE_STAMP(2);

        // the single rounding to the 16-bit model dtype (the reference's tile is 16-bit from here on).  With the default
        // epilogue nothing but the store follows, and the store's own conversion IS this rounding (same RNE; the fp16
        // clamp commutes with it): skipping the round trip through fp32 saves ~128 VALU instructions per tile.
        if constexpr (FUSE != SVDQ_FUSE_NONE) {
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int mi = 0; mi < 2; mi++)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2_t t = round16_pair<T>(acc[ni][mi][r], acc[ni][mi][r + 1]);
                        acc[ni][mi][r] = t[0];
                        acc[ni][mi][r + 1] = t[1];
                    }
        }

        if constexpr (FUSE == SVDQ_FUSE_SILU) {
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int mi = 0; mi < 2; mi++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[ni][mi][r] = round16<T>(silu_f(acc[ni][mi][r]));
        }

        if constexpr (FUSE == SVDQ_FUSE_RMSNORM_ROPE) {
            // reference EpilogueRMSNormRope (epilogues.cuh:269-425).  BN == 128 == one head.
            const int third = p.N / 3;
            const bool is_q = n0 < third;
            const bool is_k = !is_q && n0 < 2 * third;
            if (is_q || is_k) { // block-uniform
                // the norm weights and the rotary table entries of this lane's 2 rows x 32 column pairs are requested
                // BEFORE the row-sum exchange (two barriers): their memory round trip overlaps it instead of following it
                const T *nw = (const T *)(bm >= split_bm ? (is_q ? p.norm_q2 : p.norm_k2) : (is_q ? p.norm_q : p.norm_k));
                u16x4 wvv[2][4] = {};
                float2 rot[2][2][4][2] = {};
#pragma unroll
                for (int ni = 0; ni < 2; ni++)
#pragma unroll
                    for (int c = 0; c < 4; c++) wvv[ni][c] = *reinterpret_cast<const u16x4 *>(nw + wn * 64 + ni * 32 + c * 8 + h * 4);
#pragma unroll
                for (int mi = 0; mi < 2; mi++) {
                    const int m_abs = m0 + wm * 64 + mi * 32 + lr;
                    // packed rotary order (models/embeddings.py:100-138): [m/16][d/8][r8*4+p][rh][sin,cos]; the first pair
                    // of the lane's 4 columns (wn*64 + ni*32 + c*8 + h*4) is pair wn*32 + ni*16 + c*4 + 2h
                    const float *rrow = p.rotary_emb + ((size_t)(m_abs >> 4) * 16) * 128 + (size_t)((m_abs & 7) * 4) * 4 + ((m_abs >> 3) & 1) * 2 + h * 8;
#pragma unroll
                    for (int ni = 0; ni < 2; ni++)
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            if constexpr (SVDQ_PROBE_ROT == 1) { // (timing variant: the same bytes as eight lane-contiguous 16-byte loads per row tile -- a table in lane order)
                                const v4f t4 = *reinterpret_cast<const v4f *>(p.rotary_emb + (size_t)(m0 + wm * 64 + mi * 32) * 128 + wn * 2048 + lane * 32 + (ni * 4 + c) * 4);
                                rot[mi][ni][c][0] = make_float2(t4[0], t4[1]);
                                rot[mi][ni][c][1] = make_float2(t4[2], t4[3]);
                            } else if constexpr (SVDQ_PROBE_ROT == 0) {
                            const float *rp = rrow + (wn * 8 + ni * 4 + c) * 128;
                            rot[mi][ni][c][0] = *reinterpret_cast<const float2 *>(rp);
                            rot[mi][ni][c][1] = *reinterpret_cast<const float2 *>(rp + 4);
                            }
                        }
                }
                __syncthreads(); // the previous tile's readers of the epilogue scratch are done
                // LDS-address-space pointer (no flat cast: the aperture compare it needs trips an LLVM verifier bug here)
                __attribute__((address_space(3))) float *sq = (__attribute__((address_space(3))) float *)(lds + NSTAGE * STAGE_BYTES); // [2 (wn)][BM rows]
#pragma unroll
                for (int mi = 0; mi < 2; mi++) {
                    float s = 0.f;
#pragma unroll
                    for (int ni = 0; ni < 2; ni++)
#pragma unroll
                        for (int r = 0; r < 16; r++) s = __builtin_fmaf(acc[ni][mi][r], acc[ni][mi][r], s); // (fused, as nvcc contracts the reference's)
                    s += __shfl_xor(s, 32);
                    if (h == 0) sq[wn * BM + wm * 64 + mi * 32 + lr] = s;
                }
                __syncthreads();
#pragma unroll
                for (int mi = 0; mi < 2; mi++) {
                    const int row = wm * 64 + mi * 32 + lr;
                    const float tot = sq[row] + sq[BM + row];
                    const float coef = (is_q ? p.q_scale : 1.0f) / sqrtf(tot / 128.0f + 1e-6f); // (q_scale = 1: the same division as without it)
#pragma unroll
                    for (int ni = 0; ni < 2; ni++)
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const u16x4 wv = wvv[ni][c];
                            const float2 sc0 = rot[mi][ni][c][0], sc1 = rot[mi][ni][c][1];
                            float v0 = acc[ni][mi][c * 4 + 0] * (coef * h2f(hfrom<T>(wv[0])));
                            float v1 = acc[ni][mi][c * 4 + 1] * (coef * h2f(hfrom<T>(wv[1])));
                            float v2 = acc[ni][mi][c * 4 + 2] * (coef * h2f(hfrom<T>(wv[2])));
                            float v3 = acc[ni][mi][c * 4 + 3] * (coef * h2f(hfrom<T>(wv[3])));
                            // (one multiply + one fused multiply-add per output; the rounding to 16 bits is the store's own
                            //  conversion below -- nothing else reads these values: 96 VALU instructions per tile less)
                            acc[ni][mi][c * 4 + 0] = __builtin_fmaf(v0, sc0.y, -(v1 * sc0.x));
                            acc[ni][mi][c * 4 + 1] = __builtin_fmaf(v0, sc0.x, v1 * sc0.y);
                            acc[ni][mi][c * 4 + 2] = __builtin_fmaf(v2, sc1.y, -(v3 * sc1.x));
                            acc[ni][mi][c * 4 + 3] = __builtin_fmaf(v2, sc1.x, v3 * sc1.y);
                        }
                }
            }
        }

        if constexpr (FUSE == SVDQ_FUSE_GELU_QUANT) {
            // EpilogueGelu (epilogues.cuh:22-44) -> 16-bit
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int mi = 0; mi < 2; mi++)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2_t t = round16_pair<T>(gelu_tanh_f(acc[ni][mi][r]), gelu_tanh_f(acc[ni][mi][r + 1]));
                        acc[ni][mi][r] = t[0];
                        acc[ni][mi][r + 1] = t[1];
                    }

            if constexpr (SPLIT) {
                // the 16-bit GELU output as the A-operand fragments of the next layer's low-rank down projection (lowrank_down_split_kernel): the lane's 8 columns
                // {16 q + 8 (j >> 2) + 4 h + (j & 3)} of a 16-column unit are MFMA k-slots 8 h + j as they stand -- 8 coalesced 16-byte stores per wave-tile, ahead of
                // the requantisation so that they retire under its arithmetic (vmcnt retires in order: the next loop's first wait would otherwise sit behind them)
                V8 *a16 = (V8 *)p.act16_packed + ((size_t)(mw0 >> 5) * ((unsigned)p.N >> 4) + ((unsigned)nw0 >> 4)) * 64u + lane_e;
#pragma unroll
                for (int mi = 0; mi < 2; mi++)
#pragma unroll
                    for (int ni = 0; ni < 2; ni++)
#pragma unroll
                        for (int q = 0; q < 2; q++) {
                            V8 gv;
#pragma unroll
                            for (int j = 0; j < 8; j++) gv[j] = f2h<T>(acc[ni][mi][q * 8 + j]);
                            a16[((size_t)mi * ((unsigned)p.N >> 4) + (unsigned)(ni * 2 + q)) * 64u] = gv;
                        }
            }

            // EpilogueQuantize<false, unsigned> (gemm_w4a4.cuh:930-1043): 16-bit add of the shift,
            // fp32 divide by the next layer's smooth factor -> 16-bit, per (row, 64 columns) absmax,
            // scale = amax/15, unsigned 4-bit codes.  The wave's 64 columns are exactly one group of the
            // next GEMM and the lane's 32 values (j = 16*ni + r) are exactly its F6 lane record.
            const int KP2 = p.N / 128;
            const int g2 = n0 / GROUP + (wv & 1); // (= nw0 / GROUP, from the SCALAR wave index: its parity selects the record halves below with a scalar
                                                  //  branch -- derived from the lane id the compiler predicates both arms: 6 stores per row tile for 2)
            // 1 / smooth of this lane's 32 columns.  The reference divides with __fdividef (gemm_w4a4.cuh:990-993,
            // gemm_utils.cuh:329-344: an approximate reciprocal times the numerator); so does this epilogue
            // (the stand-alone quantiser keeps the exactly rounded division).
            float smr[2][16];
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const u16x4 sv = nsv[ni][c];
#pragma unroll
                    for (int e = 0; e < 4; e++) smr[ni][c * 4 + e] = __builtin_amdgcn_rcpf(h2f(hfrom<T>(sv[e])));
                }
#pragma unroll
            for (int mi = 0; mi < 2; mi++) {
                float xh[32];
                float amax = 0.f;
#pragma unroll
                for (int ni = 0; ni < 2; ni++)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2_t sh = round16_pair<T>(acc[ni][mi][r] + 0.171875f, acc[ni][mi][r + 1] + 0.171875f);
                        const f32x2_t v = round16_pair<T>(sh[0] * smr[ni][r], sh[1] * smr[ni][r + 1]);
                        xh[ni * 16 + r] = v[0];
                        xh[ni * 16 + r + 1] = v[1];
                        amax = fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1])));
                    }
                amax = fmaxf(amax, __shfl_xor(amax, 32));
                const float scale = amax * (1.0f / 15.0f);
                // 1/16 of the reciprocal: the quotients land in [0, 15/16] and the "clamp to 0" below is the multiply's own clamp
                // modifier ([0, 1]: the upper side never fires) instead of 64 v_max; the pack instruction's scale is 1/2 instead of 8.
                // Power-of-two factors: the codes are bit-identical.
                const float rscale = scale == 0.f ? 0.f : 0.0625f / scale;
                // codes 0..15 as FP6 e2m3 (= code/8 <= 1.875, exact), 32 x 6 bits in one v_cvt_scalef32_2xpk16_fp6_f32
                // (quantize.hip; element 2i from the first source, 2i+1 from the second; RNE).  The shifted GELU
                // output is >= 0 up to rounding; a negative x_hat (possible when smooth < 0 is not: smooth > 0) is
                // clamped to 0 first as the reference's unsigned saturation does.
                v16f ev, od;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    ev[i] = __builtin_amdgcn_fmed3f(xh[2 * i] * rscale, 0.f, 1.0f);
                    od[i] = __builtin_amdgcn_fmed3f(xh[2 * i + 1] * rscale, 0.f, 1.0f);
                }
                const v6i pk = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(ev, od, 0.5f);
                uint32_t rec[6];
#pragma unroll
                for (int i = 0; i < 6; i++) rec[i] = (uint32_t)pk[i];
                // Uniform base (from the SCALAR wave index) + one 32-bit lane offset: the stores take the saddr form, no 64-bit address arithmetic
                // per lane.  The group's parity (= the wave column) decides which 24 bytes of the 48-byte lane record this wave owns: even ->
                // plane 0 [0, 16) + plane 1 [0, 8); odd -> plane 1 [8, 16) + plane 2 [0, 16).  ONE 16-byte and ONE 8-byte store either way, operands
                // selected (written as two branch arms the compiler tail-merges them into five narrow stores).
                const unsigned rt_s = (unsigned)(m0 + (wv >> 1) * 64 + mi * 32) >> 5;
                uint8_t *dst = p.qout + ((size_t)rt_s * KP2 + (g2 >> 1)) * F6_CHUNK;
                const bool odd = (g2 & 1) != 0;  // scalar
                if (!SVDQ_PROBE_OFF(2)) {
                    const uint4 w16 = odd ? make_uint4(rec[2], rec[3], rec[4], rec[5]) : make_uint4(rec[0], rec[1], rec[2], rec[3]);
                    const uint2 w8 = odd ? make_uint2(rec[0], rec[1]) : make_uint2(rec[4], rec[5]);
                    *reinterpret_cast<uint4 *>(dst + (odd ? 2 * F6_PLANE : 0) + lane_e * 16u) = w16;
                    *reinterpret_cast<uint2 *>(dst + F6_PLANE + (odd ? 8 : 0) + lane_e * 16u) = w8;
                }
                // simg_index(m_abs, g2, KP2): the row tile's 32 scales of this group are 64 contiguous bytes
                if (h_e == 0) ((T *)p.oscales + (((size_t)rt_s * KP2 + (g2 >> 1)) * 2 + (g2 & 1)) * 32)[lr_e] = f2h<T>(scale);
            }
            SVDQ_PROBE_STAMP(3);
            // EpilogueLoraDown for the NEXT layer on the GELU output, before shift/smooth
            // (issued AFTER the requantisation below in program order: vmcnt retires in order on CDNA, so any load
            //  that followed these fp32 atomics would wait for their memory-side round trip)
            // (lora.cuh:243-353, launch_impl.cuh:226-262):  D'[m][r2] = sum_n g[m][n] * ld[n][r2].  In the C
            // layout a lane already holds, for its row m, the 8 columns {16q + 8(j>>2) + 4h + (j&3)} of MFMA
            // k-slots 8h + j: no data movement; the weight operand is gathered in the matching order.  With
            // the activations as the A operand the result tile has the RANK along the lanes, so every fp32
            // atomic instruction hits two 128-byte lines instead of 64 different ones.
            // (round 5, tried and removed: summing the two column waves of a row block BEFORE the atomics -- one wave hands its GELU fragments to the other through
            //  the dead staging region, which contracts all 128 columns and issues half the atomics.  Same-box: rank-128 fc1 434.8 vs 455.3 us, rank 32 + 16
            //  312.5 vs 285.4 us, deterministic mode 319 vs 289 us.  What these atomics cost is not their number per CU but their number per WAVE: the next
            //  loop's first vmcnt wait retires in order behind them (profiles/r4_gemm_rowrun.txt), and the reducing wave still issued 32 per pass.
            //  Also tried: the atomics AHEAD of the requantisation, to retire under its ~8 k cycles of arithmetic: rank-128 fc1 486 vs 455 us, rank 32 + 16 348 vs
            //  ~297 us on that box (d, the weights and the requantiser's reciprocals live together: ~120 spill instructions per tile).)
            if (!SPLIT && p.R2 > 0 && !SVDQ_PROBE_OFF(4)) {
                const T *ld = (const T *)(bm >= split_bm ? p.next_lora_down2 : p.next_lora_down); // rank-major [R2][N]
                for (int t2 = 0; t2 < (CARRY && NW == 8 && !HYB ? 32 : p.R2); t2 += 32) { // (CARRY on 256 x 128 tiles: rank <= 32, one pass; HYB: the carry takes pass 0)
                    const bool to_carry = CARRY && (!HYB || t2 == 0); // block-uniform
                    v16f d[2];
#pragma unroll
                    for (int mi = 0; mi < 2; mi++) d[mi] = zero16;
                    const bool live = t2 + lr < p.R2;
                    // this pass's 32 ranks of the down-projection weights are in `ldw` (ranks 0..31: requested at the top of the epilogue; later
                    // passes: by the pass before -- one memory round trip per 32 ranks, hidden under the previous pass's MFMAs and atomics)
                    V8 wv[2][2];
#pragma unroll
                    for (int ni = 0; ni < 2; ni++)
#pragma unroll
                        for (int q = 0; q < 2; q++) {
                            if (live) {
                                const u16x4 w0 = ldw[ni][q][0], w1 = ldw[ni][q][1];
#pragma unroll
                                for (int j = 0; j < 4; j++) { wv[ni][q][j] = hfrom<T>(w0[j]); wv[ni][q][4 + j] = hfrom<T>(w1[j]); }
                            } else {
#pragma unroll
                                for (int j = 0; j < 8; j++) wv[ni][q][j] = (T)0.f;
                            }
                        }
                    if constexpr (!CARRY || NW == 4 || HYB) {
                        if (t2 + 32 < p.R2 && t2 + 32 + lr < p.R2) {
#pragma unroll
                            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                                for (int q = 0; q < 2; q++) {
                                    const T *src = ld + (size_t)(t2 + 32 + lr) * p.N + nw0 + ni * 32 + q * 16 + h * 4;
                                    ldw[ni][q][0] = *reinterpret_cast<const u16x4 *>(src);
                                    ldw[ni][q][1] = *reinterpret_cast<const u16x4 *>(src + 8);
                                }
                        }
                    }
#pragma unroll
                    for (int ni = 0; ni < 2; ni++)
#pragma unroll
                        for (int q = 0; q < 2; q++)
#pragma unroll
                            for (int mi = 0; mi < 2; mi++) {
                                V8 gv;
#pragma unroll
                                for (int j = 0; j < 8; j++) gv[j] = f2h<T>(acc[ni][mi][q * 8 + j]);
                                d[mi] = Half<DT>::mfma32(gv, wv[ni][q], d[mi]);
                            }
#pragma unroll
                    for (int mi = 0; mi < 2; mi++) {
                        const size_t at = (size_t)(mw0 + mi * 32 + h * 4) * p.R2 + t2 + lr;
                        if (SVDQ_PROBE_OFF(1)) {
                            asm volatile("" :: "v"(d[mi]));
                        } else if (to_carry) {
                            // into the workgroup's carry: below, both row tiles at once (the two column waves of a row block take turns).  (LAQ, i.e.
                            // SVDQ_LORA_ACT_Q32_RUNS: the same fp32 carry -- tiles in the run's order, the column waves in turn order: a fixed order --
                            // converted to Q31.32 when the run is flushed)
                        } else if (live && LAQ) {
                            // deterministic mode: Q31.32 fixed point, 64-bit INTEGER atomics -- the sum does not depend on the order
                            long long *dst = (long long *)p.lora_act_out + at;
#pragma unroll
                            for (int i = 0; i < 16; i++)
                                __hip_atomic_fetch_add(dst + (size_t)((i & 3) + 8 * (i >> 2)) * p.R2, float_to_q32(d[mi][i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else if (live) {
                            float *dst = (float *)p.lora_act_out + at;
#pragma unroll
                            for (int i = 0; i < 16; i++) unsafeAtomicAdd(dst + (size_t)((i & 3) + 8 * (i >> 2)) * p.R2, d[mi][i]);
                        }
                    }
                    if constexpr (CARRY) {
                        if (to_carry && !SVDQ_PROBE_OFF(1)) {
                            // carry layout [row / 4][rank][row % 4] fp32: a lane's registers 4 g .. 4 g + 3 (rows 8 g + 4 h + 0..3 of its row tile, rank lr)
                            // are 16 contiguous bytes -- plain 16-byte reads and writes, no LDS atomics (ds_add_f32 runs at ~1 lane per clock:
                            // measured 21 k cycles per tile for these 32 instructions per wave, profiles/r4_gemm_rowrun.txt).  The two column waves
                            // of a row block add to the same words, in two turns separated by a barrier.
                            typedef __attribute__((address_space(3))) v4f lds_v4f;
                            lds_v4f *c4 = (lds_v4f *)carry + (unsigned)(t2 >> 5) * (BM * 8u) + ((unsigned)(wm * 16) + h_e) * 32u + lr_e; // + (mi * 8 + 2 g) * 32
                            carry_dirty = true;
                            // (round 6: both column waves work in both turns -- wave column 0 adds its row tile 0 while wave column 1 adds its row tile 1, a
                            //  barrier, then the other way round: different words per turn, half the serial LDS time of "column 0 first, then column 1")
                            auto add_row_tile = [&](int mi, const v16f &dd) {
                                v4f old[4];
#pragma unroll
                                for (int g = 0; g < 4; g++) old[g] = c4[(mi * 8 + 2 * g) * 32];
#pragma unroll
                                for (int g = 0; g < 4; g++) {
                                    v4f v = old[g];
#pragma unroll
                                    for (int e = 0; e < 4; e++) v[e] += dd[4 * g + e];
                                    c4[(mi * 8 + 2 * g) * 32] = v;
                                }
                            };
#pragma unroll
                            for (int turn = 0; turn < 2; turn++) {
                                if (live) {
                                    if ((wn ^ turn) == 0) add_row_tile(0, d[0]); // (wn: wave-uniform -- a scalar branch, no dynamic register indexing)
                                    else add_row_tile(1, d[1]);
                                }
                                if (turn == 0) __syncthreads();
                            }
                        }
                    }
                }
            }

        }

        SVDQ_PROBE_STAMP(4);
        // EpilogueDefault (gemm_base.cuh:667-698): store rows < M; fp16 clamps to +-65504.
        // A lane holds 4 consecutive columns per (tile, c); the partner lane (lane ^ 32) holds the next 4.
        // One v_permlane32_swap per dword turns two 8-byte pieces per lane into one 16-byte piece, so a
        // wave store writes 32 contiguous bytes per row with dwordx4 stores (8 instead of 32 per wave).
        bool vt_tile = false;
        if constexpr (FUSE == SVDQ_FUSE_RMSNORM_ROPE) vt_tile = p.out_vt != nullptr && n0 >= 2 * (p.N / 3) && !SVDQ_PROBE_VROW; // block-uniform
        if (vt_tile) {
            // V^T for svdq_attention: element (m, n) -> out_vt[(n - 2N/3) * ldvt + m].  A lane owns one row m, so
            // a register is 32 consecutive m of one channel across the half-wave: neighbouring lanes trade halves
            // (one DPP quad_perm) and every lane stores one dword = (m even, m odd) of one channel.
            const int nv0 = nw0 - 2 * (p.N / 3);
            const int odd = lr & 1;
            auto store_vt = [&](auto full) { // full: every row of the tile is a real row (block-uniform): no per-lane predicates
                // address = uniform part (tile origin + the store's channel: scalar arithmetic) + ONE 32-bit lane offset per row tile (the lane's token
                // pair and its channel inside the 8-channel piece): the 32 stores of a wave take the saddr form, no 64-bit address arithmetic per store
                const char *vt_tile = (const char *)p.out_vt + ((size_t)(n0 - 2 * (p.N / 3) + (wv & 1) * 64) * p.ldvt + (size_t)(m0 + (wv >> 1) * 64)) * 2;
#pragma unroll
                for (int mi = 0; mi < 2; mi++) {
                    const int m_base = (mw0 + mi * 32 + lr) & ~1;
                    const unsigned lane_off = ((h_e * 4u + (lr_e & 1u)) * (unsigned)p.ldvt + (unsigned)(mi * 32) + (lr_e & ~1u)) * 2u;
#pragma unroll
                    for (int ni = 0; ni < 2; ni++)
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {
                            float v0 = acc[ni][mi][r], v1 = acc[ni][mi][r + 1];
                            if constexpr (DT == SVDQ_FP16) { v0 = fminf(fmaxf(v0, -65504.f), 65504.f); v1 = fminf(fmaxf(v1, -65504.f), 65504.f); }
                            const unsigned own = (unsigned)hbits(f2h<T>(v0)) | ((unsigned)hbits(f2h<T>(v1)) << 16);
                            const unsigned oth = (unsigned)__builtin_amdgcn_update_dpp(0, (int)own, 0xB1, 0xf, 0xf, true); // lane ^ 1
                            const unsigned val = odd ? ((oth >> 16) | (own & 0xffff0000u)) : ((own & 0xffffu) | (oth << 16));
                            // channel n = nv0 + 32 ni + 8 (r >> 2) + 4 h + (r & 3) + odd: the h / odd part sits in lane_off
                            char *dst = const_cast<char *>(vt_tile) + (size_t)(ni * 32 + (r >> 2) * 8 + (r & 3)) * p.ldvt * 2 + lane_off;
                            if (decltype(full)::value || m_base + 1 < p.M) *reinterpret_cast<unsigned *>(dst) = val;
                            else if (m_base < p.M) *reinterpret_cast<uint16_t *>(dst) = (uint16_t)val;
                        }
                }
            };
            if (m0 + BM <= p.M) store_vt(std::true_type{});
            else store_vt(std::false_type{});
        } else
        if constexpr (FUSE != SVDQ_FUSE_GELU_QUANT) {
#pragma unroll
        for (int mi = 0; mi < 2; mi++) {
            const int m_abs = mw0 + mi * 32 + lr;
            T *orow = (T *)p.out + (size_t)m_abs * p.ldo + nw0 + h * 8;
            const bool ok = m_abs < p.M;
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    unsigned x[2], y[2]; // x: piece c = 2j, y: piece c = 2j + 1 (4 x 16-bit each)
#pragma unroll
                    for (int d = 0; d < 2; d++) {
                        float v0 = acc[ni][mi][(2 * j) * 4 + 2 * d], v1 = acc[ni][mi][(2 * j) * 4 + 2 * d + 1];
                        float w0 = acc[ni][mi][(2 * j + 1) * 4 + 2 * d], w1 = acc[ni][mi][(2 * j + 1) * 4 + 2 * d + 1];
                        if constexpr (DT == SVDQ_FP16) {
                            v0 = fminf(fmaxf(v0, -65504.f), 65504.f); v1 = fminf(fmaxf(v1, -65504.f), 65504.f);
                            w0 = fminf(fmaxf(w0, -65504.f), 65504.f); w1 = fminf(fmaxf(w1, -65504.f), 65504.f);
                        }
                        x[d] = (unsigned)hbits(f2h<T>(v0)) | ((unsigned)hbits(f2h<T>(v1)) << 16);
                        y[d] = (unsigned)hbits(f2h<T>(w0)) | ((unsigned)hbits(f2h<T>(w1)) << 16);
                        // x.hi-lanes <-> y.lo-lanes: lo lanes end up with (own x, partner x), hi lanes (partner y, own y)
                        auto sw = __builtin_amdgcn_permlane32_swap(x[d], y[d], false, false);
                        x[d] = sw[0];
                        y[d] = sw[1];
                    }
                    if (ok) {
                        v4i o = {(int)x[0], (int)x[1], (int)y[0], (int)y[1]};
                        *reinterpret_cast<v4i *>(orow + ni * 32 + j * 16) = o;
                    }
                }
        }
        } // FUSE != GELU_QUANT
        } // run_epilogue
        if (CARRY && carry_dirty && (!have_next || nbm != bm)) {
            // the workgroup leaves the row block: the carry goes to lora_act_out -- 16 atomic instructions per wave where every TILE used to
            // issue 32 per wave; with the row-run schedule once per run of column tiles -- and is cleared for the next row block
            carry_dirty = false;
            __syncthreads();
            // (the thread id passes through an empty asm statement: otherwise the 16 addresses are hoisted above the main loop and spilled)
            unsigned tid_e = tid;
            asm volatile("" : "+v"(tid_e));
            // float4 t + 512 j of the carry = rank t & 31 of the four rows 4 ((t >> 5) + 16 j) + 0..3
            typedef __attribute__((address_space(3))) v4f lds_v4f;
            const unsigned rank = tid_e & 31u, rg0 = tid_e >> 5;
            const v4f z4 = {0.f, 0.f, 0.f, 0.f};
            for (int sl = 0; sl < CARRY_SLABS && 32 * sl < p.R2; sl++) { // (256 x 128 tiles: one slab -- with HYB the first 32 of the next layer's ranks)
                float *dst = (float *)p.lora_act_out + ((size_t)m0 + 4 * rg0) * p.R2 + 32 * sl + rank;
                lds_v4f *src = (lds_v4f *)carry + sl * (BM * 8) + tid_e;
                const bool live = 32 * sl + (int)rank < p.R2;
#pragma unroll
                for (int j = 0; j < BM * 32 / 4 / (64 * NW); j++) {
                    const v4f v = src[j * (64 * NW)];
                    src[j * (64 * NW)] = z4;
                    if (live) {
                        if constexpr (LAQ) {
                            long long *dq = (long long *)p.lora_act_out + (dst - (float *)p.lora_act_out);
#pragma unroll
                            for (int e = 0; e < 4; e++)
                                __hip_atomic_fetch_add(dq + (size_t)(j * (64 * NW / 32) * 4 + e) * p.R2, float_to_q32(v[e]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else {
#pragma unroll
                        for (int e = 0; e < 4; e++) unsafeAtomicAdd(dst + (size_t)(j * (64 * NW / 32) * 4 + e) * p.R2, v[e]);
                        }
                    }
                }
            }
        }
        SVDQ_PROBE_STAMP(5);
        if (dyn) dq_pending = have_next ? share(dq_drawn) : -1;
        SVDQ_PROBE_NEXT_SEGMENT();
        have = have_next;
        cur = nxt;
        bm = nbm;
        bn = nbn;
    }
    SVDQ_PROBE_END();
}


// ---- the 128 x 64 wave tile, ONE wave per SIMD (round 6; geometry 8; the plain epilogue: bias + low-rank up + store) ---------------------------------------------------
// Same workgroup tile (256 x 128), operand images, stage image, persistent schedule, stream-K split and arithmetic as the 8-wave kernel above -- bit-identical
// outputs -- but FOUR waves (2 along M x 2 along N) of 4 x 2 MFMA tiles each: 128 accumulators + the P / S buffers in the 256 architectural VGPRs, fragments and
// scale tuples in 120 AGPRs (MFMA operands and LDS-read destinations only), 332 registers per wave.  What it buys (tools/ablate/gemm128_probe, profiles/
// r6_gemm_wave_tile_probe.txt, same box, bit-exact against this file's 8-wave kernel): 3/4 of the fragment bytes read from LDS per MAC, half the waves at the
// K-step rendezvous, and a chip that is no longer at its power limit at the same cycles per tile-group -- 114 cycles per 32 x 32 x 64 tile-group and SIMD at
// 2.2-2.3 GHz against ~126 at 2.0-2.2: the K = 3072 default launch 49.3 us against 56.6, K = 12288 168.5 against 185.7 (stream-K) on one box.  The generated loop
// (tools/gen_gemm_loop3.py, option set "b2"): a ring of FOUR stages, ONE K-step of LDS-DMA per body spread one or two instructions per tile-group slot (a single
// in-order wave pays every piece's acceptance time: issued as a burst the same loop runs 135 cycles per tile-group), the workgroup barrier in every other body.
// One wave per SIMD runs a VALU-bound epilogue at half the issue rate of two, so only the launches whose epilogue is small take this kernel: FUSE_NONE, rank <= 32,
// fp32 low-rank accumulators (the out-projection and fc2 of a block: 114 of the 228 GEMM launches of a FLUX step).
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int WT_NSTAGE = 4, WT_LDS_BYTES = WT_NSTAGE * Geo<8>::STAGE_BYTES; // 155648 (of 163840)
#define SVDQ_WT_CLOB_V                                                                                                                                                  \
    "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146",  \
    "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165",  \
    "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184",  \
    "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v196", "v197", "v198", "v199", "v204", "v205"
#define SVDQ_WT_CLOB_A                                                                                                                                                  \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23",     \
    "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46",  \
    "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69",  \
    "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92",  \
    "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113",  \
    "a114", "a115", "a116", "a117", "a118", "a119"
#define SVDQ_WT_CLOB_T                                                                                                                                                  \
    "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146",  \
    "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165",  \
    "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184",  \
    "v185", "v186", "v187", "v188", "v189", "v190", "v191"
#define SVDQ_WT_CLOB_ACC                                                                                                                                                \
    "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23",     \
    "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46",  \
    "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69",  \
    "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92",  \
    "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113",  \
    "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"
#define SVDQ_WT_CLOB_EP                                                                                                                                                 \
    "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139",   \
    "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159",   \
    "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179",   \
    "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199",   \
    "a200"
template <int DT>
__global__ __launch_bounds__(256, 1) void gemm_w4a4_wt128_kernel(const GemmParams p) {
    using T = typename Half<DT>::T;
    using V8 = typename Half<DT>::V8;
    constexpr int BM = 256, NW = 4, NSTAGE = WT_NSTAGE, A_BYTES = Geo<8>::A_BYTES, STAGE_BYTES = Geo<8>::STAGE_BYTES;
    __shared__ __attribute__((aligned(16))) uint8_t lds[WT_LDS_BYTES];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int KP = p.K / 128, TM = p.M_pad / BM, TN = p.N / BN, NT = TM * TN;
    const int G = gridDim.x;
    const int pos = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    auto tile_coords = [&](int t, int &bm, int &bn) {
        const int strip = t / (8 * TM);
        const int w = min(8, TN - 8 * strip);
        const int r = t - strip * 8 * TM;
        bm = r / w;
        bn = 8 * strip + r % w;
    };
    // DMA roles (tools/gen_gemm_loop3.py): every wave the three planes of A chunks w and w + 4 and of W chunk w; wave 0 the activation scale image (8 x 128 B),
    // wave 1 the weight scale image (4 x 128 B, twice), waves 2, 3 their first W plane once more: 10 LDS-DMA instructions per wave and K-step
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds_base = (unsigned)(size_t)(lds_void *)lds;
    const unsigned dA = lds_base + wv * F6_CHUNK, dX1 = lds_base + A_BYTES + wv * F6_CHUNK;
    unsigned dX2 = dX1, iX2v = F6_CHUNK, offX2 = lane * 16;
    const unsigned offA = lane * 16, offA2 = lane * 16 + 4u * KP * F6_CHUNK, offX1 = lane * 16;
    if (wv == 0) { offX2 = ((lane >> 3) & 7) * KP * 128 + (lane & 7) * 16; dX2 = lds_base + A_BYTES + W_BYTES; iX2v = 128; }
    else if (wv == 1) { offX2 = ((lane >> 3) & 3) * KP * 128 + (lane & 7) * 16; dX2 = lds_base + A_BYTES + W_BYTES + AS_BYTES; iX2v = 128; }
    const unsigned in_la = lds_base + (wm * 4) * F6_CHUNK + lane * 16;
    const unsigned in_lw = lds_base + A_BYTES + (wn * 2) * F6_CHUNK + lane * 16;
    const unsigned in_lsa = lds_base + A_BYTES + W_BYTES + (wm * 4) * 128 + (lane & 31) * 2;
    const unsigned in_lsw = lds_base + A_BYTES + W_BYTES + AS_BYTES + (wn * 2) * 128 + (lane & 31) * 2;
    const int split_bm = p.split_row == 0x7fffffff ? 0x7fffffff : p.split_row / BM;
    auto stream_ptrs = [&](int bm, int bn, int kp0, unsigned long long &a, unsigned long long &x1, unsigned long long &x2) {
        a = (unsigned long long)(p.act + ((size_t)(bm * (BM / 32) + wv) * KP + kp0) * F6_CHUNK);
        const uint8_t *wgt = bm >= split_bm ? p.wgt2 : p.wgt;
        const void *wscales = bm >= split_bm ? p.wscales2 : p.wscales;
        x1 = (unsigned long long)(wgt + ((size_t)(bn * 4 + wv) * KP + kp0) * F6_CHUNK);
        if (wv == 0) x2 = (unsigned long long)((const uint8_t *)p.ascales + ((size_t)(bm * (BM / 32)) * KP + kp0) * 128);
        else if (wv == 1) x2 = (unsigned long long)((const uint8_t *)wscales + ((size_t)(bn * 4) * KP + kp0) * 128);
        else x2 = x1;
    };
    constexpr long long SLAB_BYTES = (long long)BM * BN * 4;
    const bool ws_ok = p.workspace != nullptr && p.workspace_bytes >= SK_HEADER_BYTES + 2LL * G * SLAB_BYTES;
    GemmSchedule sched;
    sched.init(NT, KP, G, ws_ok ? p.sk_gs : 0, pos);
    const bool sk = sched.gs > 0;
    const int F = sched.F;
    typedef GemmSegment Seg;
    SVDQ_PROBE_BEGIN();

    // lane offsets of the loop call's epilogue operand loads (a[120 + 16 mi + 8 u + 4 j + e] = lora_act_in[mw0 + 32 mi + lr][16 u + 8 h + 4 j + e],
    // a[184 + 4 (2 ni + u) ..] = lora_up[nw0 + 32 ni + lr][16 u + 8 h ..], a200 = bias[nw0 + 32 h + lr]) and of the epilogue's stores
    const unsigned e_off = ((lane & 31) * (unsigned)p.ldo + (lane >> 5) * 8) * 2u, e_lr = lane & 31, e_h = lane >> 5;
    const unsigned ep_vla = ((lane & 31) * 32 + (lane >> 5) * 8) * 4u, ep_vlu = ((lane & 31) * 32 + (lane >> 5) * 8) * 2u, ep_vbi = ((lane >> 5) * 32 + (lane & 31)) * 2u;
    unsigned ring = 0, npre = 0, landed = 0;
    unsigned long long pA = 0, pX1 = 0, pX2 = 0;
    int bm = 0, bn = 0;
    Seg cur{0, 0, 0, 0}, nxt{0, 0, 0, 0};
    bool have = sched.next(cur);
    if (have) { tile_coords(cur.tile, bm, bn); stream_ptrs(bm, bn, cur.kp0, pA, pX1, pX2); }
    while (have) {
        const int kp0 = cur.kp0, kp1 = cur.kp1;
        const int m0 = bm * BM, n0 = bn * BN;
        const bool have_next = sched.next(nxt);
        int nbm = 0, nbn = 0;
        unsigned ncnt = 0;
        unsigned long long nA = 0, nX1 = 0, nX2 = 0;
        if (have_next) { tile_coords(nxt.tile, nbm, nbn); stream_ptrs(nbm, nbn, nxt.kp0, nA, nX1, nX2); ncnt = nxt.kp1 - nxt.kp0; }
        {
            const unsigned kp_s = kp1 - kp0;
            auto srd = [](unsigned long long ptr) {
                v4i r;
                r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)ptr);
                r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(ptr >> 32));
                r[2] = -1;
                r[3] = 0x00020000;
                return r;
            };
            v4i rA = srd(pA), rX1 = srd(pX1), rX2 = srd(pX2);
            // the tile's epilogue operands (bias, this wave's 128 rows of lora_act_in, its 64 rows of lora_up; rank 32) are requested at the top of the loop call into
            // AGPRs a[120:200] and land under the loop (tools/gen_gemm_loop3.py, option "ep"): flags bit 0 = low-rank operands, bit 1 = bias; only the call that
            // ends a tile (kp1 == KP) asks for them
            const bool ep_tile = kp1 == KP;
            const unsigned ep_n = kp_s; // K-steps of DMA this call issues behind the operand loads
            const unsigned ep_flags = ep_tile ? ((p.R == 32 ? 1u : 0u) | (p.bias != nullptr ? 2u : 0u)) : 0u;
            const unsigned mw0 = (unsigned)m0 + (unsigned)(wv >> 1) * 128u, nw0 = (unsigned)n0 + (unsigned)(wv & 1) * 64u;
            const unsigned long long ep_la = (unsigned long long)((const char *)p.lora_act_in + (size_t)mw0 * 128u);
            const unsigned long long ep_lu = (unsigned long long)((const char *)(bm >= split_bm ? p.lora_up2 : p.lora_up) + (size_t)nw0 * 64u);
            const unsigned long long ep_bi = (unsigned long long)((const char *)(bm >= split_bm ? p.bias2 : p.bias) + (size_t)nw0 * 2u);
            // the epilogue (generated assembly too: GenEpi3): acc + bias (an MFMA: bias[n] in one k-slot against 1.0), one MFMA per 16 ranks in ascending rank
            // order, the single rounding to 16 bits (fp16 clamps to +-65504: EpilogueDefault, gemm_base.cuh:667-698), 16-byte stores -- the 8-wave kernel's
            // operations in its order.  lane owns rows m = mw0 + 32 mi + lr and columns n = nw0 + 32 ni + 8 c + 4 h + e  (r = 4 c + e)
            const unsigned sc0 = __builtin_bit_cast(unsigned, p.lora_scales[0]), sc1 = __builtin_bit_cast(unsigned, p.lora_scales[1]);
            const int rows_left = p.M - (int)mw0;
            const unsigned row_tile_bytes = 64u * (unsigned)p.ldo;
            const unsigned long long obase = (unsigned long long)((const char *)p.out + ((size_t)mw0 * p.ldo + nw0) * 2);
#define SVDQ_LOOP3_IO                                                                                                                                    \
      "+{s[72:75]}"(rA), "+{s[76:79]}"(rX1), "+{s[80:83]}"(rX2)
#define SVDQ_LOOP3_IN                                                                                                                                    \
      "{v192}"(in_la), "{v193}"(in_lw), "{v194}"(in_lsa), "{v195}"(in_lsw), "{v200}"(offA), "{v201}"(offA2), "{v202}"(offX1), "{v203}"(offX2),              \
      "{s46}"(kp_s), "{s47}"(dA), "{s48}"(dX1), "{s49}"(dX2), "{s50}"(iX2v), "{s58}"(ring), "{s59}"(npre), "{s60}"(ncnt),                                   \
      "{s[62:63]}"(nA), "{s[64:65]}"(nX1), "{s[66:67]}"(nX2), "{s68}"(landed),                                                                               \
      "{s51}"(ep_flags), "{s54}"(ep_n), "{s[88:89]}"(ep_la), "{s[90:91]}"(ep_lu), "{s[92:93]}"(ep_bi), "{v206}"(ep_vla), "{v207}"(ep_vlu), "{v208}"(ep_vbi)
#define SVDQ_EPI3_IN                                                                                                                                     \
      "{v209}"(e_off), "{v210}"(e_lr), "{v211}"(e_h), "{s94}"(ep_flags), "{s95}"(sc0), "{s96}"(sc1), "{s97}"(rows_left), "{s98}"(row_tile_bytes),          \
      "{s[100:101]}"(obase)
#define SVDQ_LOOP3_CLOB "memory", "scc", "m0", "s52", "v212", "v213", "v214", "s53", "s55", "s56", "s57", "s61", "s84", "s85", "s86", "s87", SVDQ_WT_CLOB_V, SVDQ_WT_CLOB_A
#define SVDQ_EPI3_CLOB "vcc", "s99", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223"
            SVDQ_PROBE_WT_MID_DECL
            SVDQ_PROBE_STAMP(0);
            if (!(sk && (kp0 > 0 || kp1 < KP))) {
                // a whole tile: loop + epilogue as ONE asm statement -- the accumulators (v[0:127]) and the epilogue operands (a[120:200]) never become compiler
                // values (as outputs of one statement and inputs of the next, this clang moved all 128 accumulators through AGPRs around the schedule code)
                if constexpr (DT == SVDQ_BF16) {
                    asm volatile(
#include SVDQ_LOOP3_INC_BF16
                        SVDQ_PROBE_WT_MID_ASM
#include SVDQ_EPI3_INC_BF16
                        : SVDQ_PROBE_WT_MID_OUT SVDQ_LOOP3_IO : SVDQ_LOOP3_IN, SVDQ_EPI3_IN : SVDQ_LOOP3_CLOB, SVDQ_EPI3_CLOB, SVDQ_WT_CLOB_ACC, SVDQ_WT_CLOB_EP);
                } else {
                    asm volatile(
#include SVDQ_LOOP3_INC_FP16
                        SVDQ_PROBE_WT_MID_ASM
#include SVDQ_EPI3_INC_FP16
                        : SVDQ_PROBE_WT_MID_OUT SVDQ_LOOP3_IO : SVDQ_LOOP3_IN, SVDQ_EPI3_IN : SVDQ_LOOP3_CLOB, SVDQ_EPI3_CLOB, SVDQ_WT_CLOB_ACC, SVDQ_WT_CLOB_EP);
                }
                landed = min(3u, ncnt); // the epilogue waited for vmcnt(0) before its first store: the next tile's prefetch has landed, the next loop call need not drain the stores
                SVDQ_PROBE_WT_MID_STAMP();
            } else {
                // ---- a stream-K segment: the loop alone, then publish or collect partial tiles (the 8-wave kernel's protocol; a lane's 32 quads of accumulators as
                // 32 coalesced 1 KiB wave stores), the owner runs the epilogue ----
                v16f acc[2][4]; // [n tile][m tile]
                v16i ep_a[4], ep_u; // the loop call's epilogue operands (AGPRs)
                unsigned ep_b;
#define SVDQ_LOOP3_ACC_OUT                                                                                                                               \
      "={v[0:15]}"(acc[0][0]), "={v[16:31]}"(acc[0][1]), "={v[32:47]}"(acc[0][2]), "={v[48:63]}"(acc[0][3]),                                              \
      "={v[64:79]}"(acc[1][0]), "={v[80:95]}"(acc[1][1]), "={v[96:111]}"(acc[1][2]), "={v[112:127]}"(acc[1][3]),                                          \
      "={a[120:135]}"(ep_a[0]), "={a[136:151]}"(ep_a[1]), "={a[152:167]}"(ep_a[2]), "={a[168:183]}"(ep_a[3]), "={a[184:199]}"(ep_u), "={a200}"(ep_b)
                if constexpr (DT == SVDQ_BF16) {
                    asm volatile(
#include SVDQ_LOOP3_INC_BF16
                        : SVDQ_LOOP3_ACC_OUT, SVDQ_LOOP3_IO : SVDQ_LOOP3_IN : SVDQ_LOOP3_CLOB);
                } else {
                    asm volatile(
#include SVDQ_LOOP3_INC_FP16
                        : SVDQ_LOOP3_ACC_OUT, SVDQ_LOOP3_IO : SVDQ_LOOP3_IN : SVDQ_LOOP3_CLOB);
                }
                landed = 0;
                typedef __attribute__((address_space(1))) int gint;
                typedef __attribute__((address_space(1))) float gfloat;
                typedef __attribute__((address_space(1))) v4f gv4f;
                gint *flags = (gint *)reinterpret_cast<int *>(p.workspace);
                gfloat *slabs = (gfloat *)reinterpret_cast<float *>(p.workspace + SK_HEADER_BYTES);
                const int trel = cur.tile - F * G;
                if (kp1 < KP) {
                    gfloat *slab = slabs + (size_t)sched.slot(cur) * (BM * BN);
#pragma unroll
                    for (int j = 0; j < 32; j++) {
                        v4f v = {acc[j >> 4][(j >> 2) & 3][(j & 3) * 4 + 0], acc[j >> 4][(j >> 2) & 3][(j & 3) * 4 + 1],
                                 acc[j >> 4][(j >> 2) & 3][(j & 3) * 4 + 2], acc[j >> 4][(j >> 2) & 3][(j & 3) * 4 + 3]};
                        *(gv4f *)(slab + ((size_t)(j * NW + wave) * 64 + lane) * 4) = v;
                    }
                    __syncthreads();
                    if (tid == 0) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __hip_atomic_fetch_add(flags + trel, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                } else {
                    const int first = sched.first_contributor(cur);
                    const int needed = pos - first;
                    if (tid == 0) {
                        int spins = 0;
                        while (__hip_atomic_load(flags + trel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < needed && ++spins < SK_SPIN_LIMIT)
                            __builtin_amdgcn_s_sleep(8);
                        if (spins >= SK_SPIN_LIMIT) {
                            __hip_atomic_store(flags + SK_ERR_WORD, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (p.status) __hip_atomic_store(p.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                        __hip_atomic_store(flags + trel, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    }
                    __syncthreads();
                    for (int q = first; q < pos; q++) {
                        const gfloat *slab = slabs + (size_t)sched.contributor_slot(cur, q) * (BM * BN);
#pragma unroll
                        for (int jb = 0; jb < 32; jb += 8) {
                            v4f t[8];
#pragma unroll
                            for (int j = 0; j < 8; j++) t[j] = __builtin_nontemporal_load((const gv4f *)(slab + ((size_t)((jb + j) * NW + wave) * 64 + lane) * 4));
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int j = 0; j < 8; j++)
#pragma unroll
                                for (int e = 0; e < 4; e++) acc[(jb + j) >> 4][((jb + j) >> 2) & 3][((jb + j) & 3) * 4 + e] += t[j][e];
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
#define SVDQ_EPI3_ACC_IO                                                                                                                                 \
      "+{v[0:15]}"(acc[0][0]), "+{v[16:31]}"(acc[0][1]), "+{v[32:47]}"(acc[0][2]), "+{v[48:63]}"(acc[0][3]),                                              \
      "+{v[64:79]}"(acc[1][0]), "+{v[80:95]}"(acc[1][1]), "+{v[96:111]}"(acc[1][2]), "+{v[112:127]}"(acc[1][3])
#define SVDQ_EPI3_EP_IN                                                                                                                                  \
      "{a[120:135]}"(ep_a[0]), "{a[136:151]}"(ep_a[1]), "{a[152:167]}"(ep_a[2]), "{a[168:183]}"(ep_a[3]), "{a[184:199]}"(ep_u), "{a200}"(ep_b)
                    if constexpr (DT == SVDQ_BF16) {
                        asm volatile(
#include SVDQ_EPI3_INC_BF16
                            : SVDQ_EPI3_ACC_IO : SVDQ_EPI3_EP_IN, SVDQ_EPI3_IN : "memory", "scc", "s84", "s85", "s70", "s71", "v215", SVDQ_EPI3_CLOB, SVDQ_WT_CLOB_T);
                    } else {
                        asm volatile(
#include SVDQ_EPI3_INC_FP16
                            : SVDQ_EPI3_ACC_IO : SVDQ_EPI3_EP_IN, SVDQ_EPI3_IN : "memory", "scc", "s84", "s85", "s70", "s71", "v215", SVDQ_EPI3_CLOB, SVDQ_WT_CLOB_T);
                    }
                    landed = min(3u, ncnt);
                }
            }
            ring = (ring + (kp_s % NSTAGE) * STAGE_BYTES) % (NSTAGE * STAGE_BYTES);
            npre = min(3u, ncnt); // every body of this loop leaves three K-steps of the operand stream ahead of the one it computed
            pA = nA; pX1 = nX1; pX2 = nX2;
            SVDQ_PROBE_STAMP(5);
            SVDQ_PROBE_NEXT_SEGMENT();
        }
        have = have_next;
        cur = nxt;
        bm = nbm;
        bn = nbn;
    }
    SVDQ_PROBE_END();
}
#undef SVDQ_WT_CLOB_V
#undef SVDQ_WT_CLOB_A
#undef SVDQ_WT_CLOB_T
#undef SVDQ_WT_CLOB_ACC
#undef SVDQ_WT_CLOB_EP
#undef SVDQ_LOOP3_IO
#undef SVDQ_LOOP3_IN
#undef SVDQ_EPI3_IN
#undef SVDQ_LOOP3_CLOB
#undef SVDQ_EPI3_CLOB
#undef SVDQ_LOOP3_ACC_OUT
#undef SVDQ_EPI3_ACC_IO
#undef SVDQ_EPI3_EP_IN

// What the last svdq_gemm_w4a4 call of this thread launched (svdq_gemm_last_plan): tile rows (256 | 128), kernel variant, grid, stream-K groups, row-run length,
// whether the low-rank operands were packed.  Filled by the dispatch code itself -- tests read it to assert that no rank, shape or format fell back to a slower
// kernel than the one documented for it (include/svdq_amd.h).
enum { PLAN_PLAIN = 0, PLAN_CARRY = 1, PLAN_ALL_RANK = 2, PLAN_HYBRID_CARRY = 3, PLAN_SOLO_CARRY = 4, PLAN_SPLIT_DOWN = 5, PLAN_WAVE_TILE_128 = 6 };
static thread_local int32_t g_last_plan[8] = {0, 0, 0, 0, 0, 0, 0, 0};
static void record_plan(int tile_rows, int variant, int grid, const GemmParams &p) {
    g_last_plan[0] = tile_rows; g_last_plan[1] = variant; g_last_plan[2] = grid; g_last_plan[3] = p.sk_gs; g_last_plan[4] = p.rowrun;
    g_last_plan[5] = (p.la_packed != nullptr && (variant == PLAN_ALL_RANK || variant == PLAN_SOLO_CARRY || variant == PLAN_SPLIT_DOWN)) ? 1 : 0;
    g_last_plan[6] = (p.lu_packed != nullptr && (variant == PLAN_SOLO_CARRY || (variant == PLAN_ALL_RANK && tile_rows == 128))) ? 1 : 0;
    g_last_plan[7] = p.dynamic;
}

// compute units the persistent grids are sized for (a multiple of 8: the XCD-aware numbering deals workgroups to the 8
// XCDs; at most 256: the stream-K header holds 1023 arrival counters for up to 2 x 256 workgroups)
static int device_cus() {
    static int cus = 0; // benign race: every thread computes the same value
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        if (n > 256) n = 256;
        cus = n >= 8 ? (n / 8) * 8 : n;
    }
    return cus;
}
// one size for both geometries: 2 slabs per workgroup slot, 256 x 128 tiles on `cus` slots == 128 x 128 tiles on 2 * cus
static long long workspace_slab_bytes() { return SK_HEADER_BYTES + 2LL * device_cus() * 256 * BN * 4; }
// ... + the packed lora_act_in image of the all-rank kernels (LA_PACK_BYTES) behind the slabs
static long long workspace_bytes_needed() { return workspace_slab_bytes() + LA_PACK_BYTES + LU_PACK_BYTES; }
// ... + (ABI 20) the 16-bit image of a GELU_QUANT launch's output for the split low-rank down projection, behind everything else: M_pad * N * 2 bytes.
// What the arguments alone decide (the workspace size is checked by the caller): GELU_QUANT, fp32 accumulators, own rank on the all-rank path (48 .. 160,
// 16-byte aligned operands, image within its tail), next rank 48 .. 160 whose packed down projection(s) fit the lora_up tail.
static bool split_down_shape_ok(const svdq_gemm_args *a) {
    return a->fuse == SVDQ_FUSE_GELU_QUANT && a->lora_act_format == SVDQ_LORA_ACT_F32 && lowrank_split_shape_ok(a->N, a->R2) && a->R > 32 &&
           a->R <= Geo<8>::STG_LU_ALL_MAX_R && a->lora_act_in && a->lora_up && (((uintptr_t)a->lora_act_in | (uintptr_t)a->lora_up | (uintptr_t)a->lora_up2) & 15) == 0 &&
           (long long)a->M_pad * a->R * 2 <= LA_PACK_BYTES && lowrank_split_pack_bytes(a->N, a->R2, a->wgt2 != nullptr) <= LU_PACK_BYTES;
}
// the library's own choice (geometry 0): every next-layer rank beyond the carry's 32, from a full round of 256 x 128 tiles (same box, profiles/r5_split_down_ab.txt:
// next rank 48: 271 -> 258 us, 64: 296 -> 256 us against the hybrid carry; 128: 354 -> 277 us against the solo carry, 443 with per-tile atomics)
static bool split_down_auto(const svdq_gemm_args *a) { return a->R2 > 32 && (long long)(a->M_pad / 128) * (a->N / BN) >= 2LL * device_cus(); }

// Stream-K heuristic.  The remainder R = tiles % slots of the last round leaves CUs idle for a whole tile time;
// splitting those tiles along K costs every split ~2 x 128 KiB of fp32 partial traffic plus a prologue
// (measured ~10 K-steps worth), so it only pays for long K or nearly empty last rounds.  At most two pieces
// per tile, at least 8 K-steps per piece.  Returns the number of workgroups sharing the remainder (0 = off).
static int streamk_groups_for(int tiles, int KP, int slots) {
    const int R = tiles % slots;
    if (R == 0) return 0;
    constexpr int max_pieces = 2, min_steps = 8;
    long long gs = (long long)R * max_pieces;
    if (gs > slots) gs = slots;
    while (gs > R && (long long)R * KP / gs < min_steps) gs--;
    if (gs <= R) return 0;
    const double with_sk = (double)R * KP / gs + 10.0, without = (double)KP;
    return with_sk < 0.85 * without ? (int)gs : 0;
}
// Grid of a launch WITHOUT stream-K.  Whole rounds on fewer workgroups: 1296 tiles are 6 rounds on 256 slots (the last one 6 % full)
// and 6 rounds on 216 -- the same number of tile times, but no CU sits through a nearly empty round and the chip is
// power-limited: the idle CUs lend their budget to the busy ones (profiles/r2_gemm_loop_variants_ab.txt).  A multiple of 8
// (XCD-aware numbering).
static int whole_rounds_grid(int tiles, int slots) {
    if (tiles <= slots) return tiles;
    const int rounds = (tiles + slots - 1) / slots;
    const int g = ((tiles + rounds - 1) / rounds + 7) / 8 * 8;
    return g <= slots ? g : slots;
}
static int persistent_grid(int tiles, int sk_gs, int slots) {
    if (sk_gs > 0) return tiles < slots ? max(tiles, sk_gs) : slots;
    return whole_rounds_grid(tiles, slots);
}

// Geometry of a launch (svdq_gemm_args.geometry; 0 = this heuristic).
// Measured rule (profiles/r3_gemm_geometry.txt): a tile costs the same CU-cycles in both geometries (the SIMDs are
// throughput-bound in the loop AND in the epilogues: co-residency hides no work), so the 128 x 128 queue wins exactly where
// the 256 x 128 schedule cannot use the chip evenly: whole rounds that leave >= 10 % of the CUs out (1296 tiles = 6 rounds
// on 216 of 256 CUs), provided the queue has >= 2 tiles per workgroup to balance with.  Long K stays on geometry 1 (stream-K).
static int pick_geometry(const svdq_gemm_args *a, bool with_ws) {
    if (a->geometry != 0 && a->geometry != 8) return a->geometry; // (8: the 128 x 64 wave tile where it serves the launch, this rule otherwise)
    if (!with_ws) return 1;
    // rank 48 .. 160: both geometries have all-rank kernels (256 x 128: lora_up of a tile staged in LDS + packed lora_act_in; 128 x 128: both operands packed), so
    // the choice below is the rank-32 one -- except for a grouped launch from rank 96, whose second lora_up only the 256 x 128 kernel serves without the plain
    // kernels' row-per-lane loads (16-18 k cycles per 128 x 128 tile at rank 128, profiles/r5_gemm_phase_trace.txt)
    if (a->wgt2 && a->R >= 96 && a->R <= Geo<8>::STG_LU_ALL_MAX_R && a->lora_act_format == SVDQ_LORA_ACT_F32 && (long long)a->M_pad * a->R * 2 <= LA_PACK_BYTES) return 1;
    const int cus = device_cus();
    const int tiles1 = (a->M_pad / 256) * (a->N / BN), tiles2 = 2 * tiles1;
    if (tiles1 <= cus || streamk_groups_for(tiles1, a->K / 128, cus) > 0) return 1;
    const int rounds1 = (tiles1 + cus - 1) / cus;
    const double use1 = (double)tiles1 / ((double)rounds1 * cus);
    return (use1 < 0.9 && tiles2 >= 4 * cus) ? 2 : 1;
}

// Row runs (GELU_QUANT on 256 x 128 tiles with a next-layer low-rank branch of rank <= 32 and no K split): worth it from two tiles per
// workgroup (profiles/r4_gemm_rowrun.txt: the per-tile atomics of the low-rank down projection cost 12 % of the fc1 launch)
static bool rowrun_applies(int M_pad, int N, int R2, int sk_gs, int slots) {
    return R2 > 0 && sk_gs == 0 && (M_pad / 256) * (N / BN) >= 2 * slots; // (callers: R2 <= 32, or the hybrid carry kernel beyond)
}

template <int DT, int FUSE, int NW, bool LAQ>
static void launch_one_laq(GemmParams &p, bool with_ws, hipStream_t st) {
    using G_ = Geo<NW>;
    const int tiles = (p.M_pad / G_::BM) * (p.N / BN), slots = device_cus() * G_::WG_PER_CU;
    p.sk_gs = with_ws ? streamk_groups_for(tiles, p.K / 128, slots) : 0;
    // dynamic tile queues (128 x 128 geometry): when every slot has more than one tile to do and the remainder is not
    // better served by a K split (long K: few, long tiles -- a queue cannot balance 1.7 tiles per workgroup)
    p.dynamic = NW == 4 && p.dynamic && with_ws && tiles > slots && p.sk_gs == 0;
    int g = persistent_grid(tiles, p.sk_gs, slots);
    if (p.dynamic) g = slots; // every CU hosts two workgroups; the queue balances them
    p.rowrun = 0;
    // the low-rank-down carry (and, from two tiles per workgroup, the row-run schedule that makes it pay): GELU_QUANT on 256 x 128 tiles with an fp32
    // next-layer low-rank branch of rank <= 32
    // ... next-layer ranks 48 .. 128: the carry lives behind the ring of ONE 128 x 128 workgroup per CU (svdq_gemm_w4a4 picks that geometry: solo_carry)
    if constexpr (NW == 4 && FUSE == SVDQ_FUSE_GELU_QUANT && !LAQ) {
        if (p.solo_carry) {
            const int TM = p.M_pad / G_::BM, TN = p.N / BN, cus = device_cus();
            p.sk_gs = 0; p.dynamic = 0; p.stagger = 0;
            p.rowrun = GemmSchedule::run_length(TM, TN, cus);
            const int g4 = (TM * ((TN + p.rowrun - 1) / p.rowrun) + 7) / 8 * 8;
            dim3 grid(SVDQ_PROBE_GRID(g4, tiles, cus)), block(G_::THREADS);
            if (p.lu_packed && p.la_packed) { // both low-rank operands as MFMA fragments (the grouped launch's second lora_up is not packed: see svdq_gemm_w4a4)
                float16_scales sc;
                for (int i = 0; i < MAX_LORA_TILES; i++) sc.v[i] = p.lora_scales[i];
                const int units = p.R / 16;
                hipLaunchKernelGGL((pack_lora_act_kernel<DT>), dim3(p.M_pad / 32, (units + 3) / 4), dim3(256), 0, st, (const float *)p.lora_act_in,
                                   (typename Half<DT>::V8 *)p.la_packed, p.R, units, sc);
                if (!p.lu_pre)
                    hipLaunchKernelGGL((pack_lora_up_kernel<DT>), dim3(p.N / 32, (units + 3) / 4), dim3(256), 0, st, (const typename Half<DT>::T *)p.lora_up,
                                       (typename Half<DT>::V8 *)p.lu_packed, p.R, units);
            }
            record_plan(G_::BM, PLAN_SOLO_CARRY, (int)grid.x, p);
            hipLaunchKernelGGL((gemm_w4a4_kernel<DT, FUSE, NW, LAQ, true>), grid, block, 0, st, p);
            return;
        }
    }
    if constexpr (NW == 4 && !LAQ) {
        if (p.lu_packed && p.la_packed) { // rank 48 .. 160 on 128 x 128 tiles: both low-rank operands packed, then the all-rank kernel of this geometry
            float16_scales sc;
            for (int i = 0; i < MAX_LORA_TILES; i++) sc.v[i] = p.lora_scales[i];
            const int units = p.R / 16;
            hipLaunchKernelGGL((pack_lora_act_kernel<DT>), dim3(p.M_pad / 32, (units + 3) / 4), dim3(256), 0, st, (const float *)p.lora_act_in,
                               (typename Half<DT>::V8 *)p.la_packed, p.R, units, sc);
            if (!p.lu_pre)
                hipLaunchKernelGGL((pack_lora_up_kernel<DT>), dim3(p.N / 32, (units + 3) / 4), dim3(256), 0, st, (const typename Half<DT>::T *)p.lora_up,
                                   (typename Half<DT>::V8 *)p.lu_packed, p.R, units);
            dim3 grid(SVDQ_PROBE_GRID(g, tiles, slots)), block(G_::THREADS);
            record_plan(G_::BM, PLAN_ALL_RANK, (int)grid.x, p);
            hipLaunchKernelGGL((gemm_w4a4_kernel<DT, FUSE, NW, LAQ, false, true>), grid, block, 0, st, p);
            return;
        }
    }
    if constexpr (NW == 8 && FUSE == SVDQ_FUSE_GELU_QUANT && LAQ) {
        // SVDQ_LORA_ACT_Q32_RUNS (ABI 22): the row-run schedule with its fp32 carry, the run's sum converted to fixed point at the flush -- no atomics in a tile
        if (p.lora_fixed == SVDQ_LORA_ACT_Q32_RUNS && p.R2 > 0 && p.R2 <= 32 && rowrun_applies(p.M_pad, p.N, p.R2, p.sk_gs, slots)) {
            const int TM = p.M_pad / G_::BM, TN = p.N / BN;
            p.rowrun = GemmSchedule::run_length(TM, TN, slots);
            g = (TM * ((TN + p.rowrun - 1) / p.rowrun) + 7) / 8 * 8;
            dim3 grid(SVDQ_PROBE_GRID(g, tiles, slots)), block(G_::THREADS);
            record_plan(G_::BM, PLAN_CARRY, (int)grid.x, p);
            hipLaunchKernelGGL((gemm_w4a4_kernel<DT, FUSE, NW, LAQ, true>), grid, block, 0, st, p);
            return;
        }
    }
    constexpr bool CAN_CARRY = NW == 8 && FUSE == SVDQ_FUSE_GELU_QUANT && !LAQ;
    if constexpr (CAN_CARRY) {
        if (p.R2 > 32 && rowrun_applies(p.M_pad, p.N, p.R2, p.sk_gs, slots)) { // beyond rank 32 (and not the solo-carry kernel's case): the hybrid carry
            const int TM = p.M_pad / G_::BM, TN = p.N / BN;
            p.rowrun = GemmSchedule::run_length(TM, TN, slots);
            g = (TM * ((TN + p.rowrun - 1) / p.rowrun) + 7) / 8 * 8;
            dim3 grid(SVDQ_PROBE_GRID(g, tiles, slots)), block(G_::THREADS);
            record_plan(G_::BM, PLAN_HYBRID_CARRY, (int)grid.x, p);
            hipLaunchKernelGGL((gemm_w4a4_kernel<DT, FUSE, NW, LAQ, true, false, true>), grid, block, 0, st, p);
            return;
        }
        if (p.R2 > 0 && p.R2 <= 32) {
            if (rowrun_applies(p.M_pad, p.N, p.R2, p.sk_gs, slots)) {
                const int TM = p.M_pad / G_::BM, TN = p.N / BN;
                p.rowrun = GemmSchedule::run_length(TM, TN, slots);
                g = (TM * ((TN + p.rowrun - 1) / p.rowrun) + 7) / 8 * 8; // a multiple of 8: the XCD-aware numbering keeps consecutive runs on one XCD
            }
            dim3 grid(SVDQ_PROBE_GRID(g, tiles, slots)), block(G_::THREADS);
            record_plan(G_::BM, PLAN_CARRY, (int)grid.x, p);
            hipLaunchKernelGGL((gemm_w4a4_kernel<DT, FUSE, NW, LAQ, true>), grid, block, 0, st, p);
            return;
        }
    }
    dim3 grid(SVDQ_PROBE_GRID(g, tiles, slots)), block(G_::THREADS);
    if constexpr (NW == 8 && !LAQ) {
        if (p.stage_lu_all && p.la_packed) { // 32 < rank <= 160: the kernel with the all-rank lora_up image, behind the pack of its low-rank activations
            float16_scales sc;
            for (int i = 0; i < MAX_LORA_TILES; i++) sc.v[i] = p.lora_scales[i];
            const int units = p.R / 16;
            hipLaunchKernelGGL((pack_lora_act_kernel<DT>), dim3(p.M_pad / 32, (units + 3) / 4), dim3(256), 0, st, (const float *)p.lora_act_in,
                               (typename Half<DT>::V8 *)p.la_packed, p.R, units, sc);
            if constexpr (FUSE == SVDQ_FUSE_GELU_QUANT) {
                if (p.act16_packed) { // split low-rank down: pack the next layer's down projection(s), the GEMM with the fragment-storing epilogue, the contraction
                    record_plan(G_::BM, PLAN_SPLIT_DOWN, (int)grid.x, p);
                    launch_lowrank_down_split<DT>(p.act16_packed, p.next_lora_down, p.next_lora_down2, p.split_row, p.M_pad, p.N, p.split_R2, (float *)p.lora_act_out,
                                                  p.workspace + workspace_slab_bytes() + LA_PACK_BYTES, device_cus(), st,
                                                  [&]() { hipLaunchKernelGGL((gemm_w4a4_kernel<DT, FUSE, NW, LAQ, false, true, false, true>), grid, block, 0, st, p); },
                                                  p.ld_pre, p.ld2_pre);
                    return;
                }
            }
            record_plan(G_::BM, PLAN_ALL_RANK, (int)grid.x, p);
            hipLaunchKernelGGL((gemm_w4a4_kernel<DT, FUSE, NW, LAQ, false, true>), grid, block, 0, st, p);
            return;
        }
    }
    record_plan(G_::BM, PLAN_PLAIN, (int)grid.x, p);
    hipLaunchKernelGGL((gemm_w4a4_kernel<DT, FUSE, NW, LAQ>), grid, block, 0, st, p);
}
template <int DT, int FUSE, int NW>
static void launch_one(GemmParams &p, bool with_ws, hipStream_t st) {
    if (p.lora_fixed) launch_one_laq<DT, FUSE, NW, true>(p, with_ws, st);
    else launch_one_laq<DT, FUSE, NW, false>(p, with_ws, st);
}
// geometry 8: the 128 x 64 wave tile kernel (FUSE_NONE, rank <= 32, fp32 low-rank accumulators); schedule and stream-K split as the 256 x 128 geometry's
template <int DT>
static void launch_wt128(GemmParams &p, bool with_ws, hipStream_t st) {
    const int tiles = (p.M_pad / 256) * (p.N / BN), slots = device_cus();
    p.sk_gs = with_ws ? streamk_groups_for(tiles, p.K / 128, slots) : 0;
    p.dynamic = 0; p.stagger = 0; p.rowrun = 0;
    const int g = persistent_grid(tiles, p.sk_gs, slots);
    dim3 grid(SVDQ_PROBE_GRID(g, tiles, slots)), block(256);
    record_plan(256, PLAN_WAVE_TILE_128, (int)grid.x, p);
    hipLaunchKernelGGL((gemm_w4a4_wt128_kernel<DT>), grid, block, 0, st, p);
}
static bool wt128_serves(const svdq_gemm_args *a) { return a->fuse == SVDQ_FUSE_NONE && a->lora_act_format == SVDQ_LORA_ACT_F32 && (a->R == 32 || a->R == 0); }

template <int DT, int NW>
static void launch_fuse(GemmParams &p, int fuse, bool with_ws, hipStream_t st) {
    switch (fuse) {
    case SVDQ_FUSE_NONE: launch_one<DT, SVDQ_FUSE_NONE, NW>(p, with_ws, st); break;
    case SVDQ_FUSE_SILU: launch_one<DT, SVDQ_FUSE_SILU, NW>(p, with_ws, st); break;
    case SVDQ_FUSE_GELU_QUANT: launch_one<DT, SVDQ_FUSE_GELU_QUANT, NW>(p, with_ws, st); break;
    case SVDQ_FUSE_RMSNORM_ROPE: launch_one<DT, SVDQ_FUSE_RMSNORM_ROPE, NW>(p, with_ws, st); break;
    }
}

} // namespace svdq

using namespace svdq;

extern "C" int64_t svdq_gemm_workspace_bytes(void) { return workspace_bytes_needed(); }
// ABI 20: the size with which THIS launch takes every fast path it has -- svdq_gemm_workspace_bytes(), plus the 16-bit output image of a GELU_QUANT launch whose
// next-layer low-rank down projection can run split (M_pad * N * 2 bytes).  A smaller workspace is never an error: the launch takes the path that fits.
extern "C" int64_t svdq_gemm_workspace_bytes_for(const svdq_gemm_args *a) {
    if (!a) return workspace_bytes_needed();
    if (a->M_pad <= 0 || a->N <= 0 || a->M_pad % 256 || a->N % 128 || a->R < 0 || a->R2 < 0) return workspace_bytes_needed();
    const bool want = split_down_shape_ok(a) && (a->geometry == 7 || (a->geometry == 0 && split_down_auto(a)));
    return workspace_bytes_needed() + (want ? (long long)a->M_pad * a->N * 2 : 0);
}

extern "C" int svdq_gemm_last_plan(int32_t *out8) {
    if (!out8) { set_error("svdq_gemm_last_plan: out is NULL"); return SVDQ_E_INVALID; }
    for (int i = 0; i < 8; i++) out8[i] = g_last_plan[i];
    return SVDQ_OK;
}

// Reads the sticky error word of a stream-K workspace after the work queued on `stream` has drained (this call
// SYNCHRONISES the stream: a test / debugging aid, not part of the hot path -- the hot path's cheap check is the
// host-visible svdq_gemm_args.status word).  Non-zero means an owner workgroup gave up waiting for partial tiles
// (gemm_w4a4_kernel, SK_SPIN_LIMIT): the workspace was shared by launches in flight on two streams, was not zero-filled,
// or the grid was not co-resident.  The word is cleared by the call.
extern "C" int svdq_gemm_workspace_status(void *workspace, void *stream) {
    if (!workspace) { set_error("svdq_gemm_workspace_status: workspace is NULL"); return SVDQ_E_INVALID; }
    hipStream_t st = (hipStream_t)stream;
    int word = 0;
    int *dev = reinterpret_cast<int *>(workspace) + SK_ERR_WORD;
    if (hip_check(hipMemcpyAsync(&word, dev, sizeof(int), hipMemcpyDeviceToHost, st), "svdq_gemm_workspace_status copy")) return SVDQ_E_HIP;
    if (hip_check(hipStreamSynchronize(st), "svdq_gemm_workspace_status sync")) return SVDQ_E_HIP;
    if (word != 0) {
        (void)hipMemsetAsync(dev, 0, sizeof(int), st);
        set_error("svdq_gemm_w4a4: a stream-K owner timed out waiting for partial tiles -- the workspace was used by launches in "
                  "flight on more than one stream, was not zero-filled, or the grid was not co-resident; results of those launches are invalid");
        return SVDQ_E_HIP;
    }
    return SVDQ_OK;
}

// Host-side replay of the persistent schedule (the same GemmSchedule code the kernel runs): for the problem
// (M_pad, N, K) on `cus` compute units with the given geometry (1: 256 x 128 tiles, one workgroup per CU; 2: 128 x 128
// tiles, two per CU), with or without a stream-K workspace, write up to `cap` records
// {position, tile, kp0, kp1, slot-or-minus-one, contributors} and return the number of segments (or -1).
extern "C" int svdq_gemm_schedule_ex(int32_t M_pad, int32_t N, int32_t K, int32_t cus, int32_t with_workspace, int32_t geometry,
                                     int32_t *out, int32_t cap) {
    if (geometry != 1 && geometry != 2 && geometry != 3) return -1;
    const int bm = geometry == 2 ? 128 : 256, slots = geometry == 2 ? 2 * cus : cus;
    if (M_pad <= 0 || N <= 0 || K <= 0 || M_pad % 256 || N % BN || K % 128 || cus <= 0) return -1;
    if (geometry == 3) {
        // the row-run schedule of a GELU_QUANT launch (256 x 128 tiles): tile ids are ROW-MAJOR (tile = bm * TN + bn); -1 when the launch
        // would take the plain schedule (fewer than two tiles per workgroup)
        const int TM = M_pad / 256, TN = N / BN;
        if (!rowrun_applies(M_pad, N, 32, 0, cus)) return -1;
        const int rl = GemmSchedule::run_length(TM, TN, cus);
        const int G = (TM * ((TN + rl - 1) / rl) + 7) / 8 * 8;
        int n = 0;
        for (int pos = 0; pos < G; pos++) {
            GemmSchedule sc;
            sc.init_runs(TM, TN, K / 128, rl, pos);
            GemmSegment sg;
            while (sc.next(sg)) {
                if (out && n < cap) { int32_t *r = out + 6 * n; r[0] = pos; r[1] = sg.tile; r[2] = sg.kp0; r[3] = sg.kp1; r[4] = -1; r[5] = 0; }
                n++;
            }
        }
        return n;
    }
    const int tiles = (M_pad / bm) * (N / BN), KP = K / 128;
    const int gs = with_workspace ? streamk_groups_for(tiles, KP, slots) : 0;
    const int G = persistent_grid(tiles, gs, slots);
    int n = 0;
    for (int pos = 0; pos < G; pos++) {
        GemmSchedule sc;
        sc.init(tiles, KP, G, gs, pos);
        GemmSegment sg;
        while (sc.next(sg)) {
            if (out && n < cap) {
                int32_t *r = out + 6 * n;
                const bool partial = sc.gs > 0 && (sg.kp0 > 0 || sg.kp1 < KP);
                r[0] = pos; r[1] = sg.tile; r[2] = sg.kp0; r[3] = sg.kp1;
                r[4] = partial && sg.kp1 < KP ? sc.slot(sg) : -1;
                r[5] = partial && sg.kp1 == KP ? pos - sc.first_contributor(sg) : 0;
            }
            n++;
        }
    }
    return n;
}
extern "C" int svdq_gemm_schedule(int32_t M_pad, int32_t N, int32_t K, int32_t cus, int32_t with_workspace, int32_t *out, int32_t cap) {
    return svdq_gemm_schedule_ex(M_pad, N, K, cus, with_workspace, 1, out, cap);
}

extern "C" int64_t svdq_pack_lora_down_bytes(int32_t N, int32_t R2) { return lowrank_split_pack_bytes(N, R2, false); }
extern "C" int svdq_pack_lora_down(const void *ld, void *out, int32_t N, int32_t R2, int32_t dtype, void *stream) {
    if (!ld || !out || !lowrank_split_shape_ok(N, R2) || R2 % 16 || (dtype != SVDQ_BF16 && dtype != SVDQ_FP16)) {
        set_error("svdq_pack_lora_down: needs ld, out, N %% 256 == 0, R2 a multiple of 16 in 48 .. 160 (N=%d R2=%d)", N, R2);
        return SVDQ_E_INVALID;
    }
    if (dtype == SVDQ_BF16) launch_pack_lora_down<SVDQ_BF16>(ld, out, N, R2, (hipStream_t)stream);
    else launch_pack_lora_down<SVDQ_FP16>(ld, out, N, R2, (hipStream_t)stream);
    return hip_check(hipGetLastError(), "svdq_pack_lora_down launch");
}
extern "C" int64_t svdq_pack_lora_up_bytes(int32_t N, int32_t R) { return (int64_t)N * R * 2; }
extern "C" int svdq_pack_lora_up(const void *lu, void *out, int32_t N, int32_t R, int32_t dtype, void *stream) {
    if (!lu || !out || N % 32 || R % 16 || R <= 0 || (dtype != SVDQ_BF16 && dtype != SVDQ_FP16)) {
        set_error("svdq_pack_lora_up: needs lu, out, N %% 32 == 0, R a positive multiple of 16 (N=%d R=%d)", N, R);
        return SVDQ_E_INVALID;
    }
    const int units = R / 16;
    if (dtype == SVDQ_BF16)
        hipLaunchKernelGGL((pack_lora_up_kernel<SVDQ_BF16>), dim3(N / 32, (units + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)lu, (bf16x8 *)out, R, units);
    else
        hipLaunchKernelGGL((pack_lora_up_kernel<SVDQ_FP16>), dim3(N / 32, (units + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)lu, (f16x8 *)out, R, units);
    return hip_check(hipGetLastError(), "svdq_pack_lora_up launch");
}

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
    if (a->reserved2 != 0 || !(a->q_scale >= 0.f) || !(a->q_scale < 1e30f) || (a->q_scale != 0.f && a->fuse != SVDQ_FUSE_RMSNORM_ROPE)) {
        set_error("svdq_gemm_w4a4: q_scale=%g must be finite and >= 0 and needs the RMSNORM_ROPE epilogue; reserved2 must be 0", (double)a->q_scale);
        return SVDQ_E_INVALID;
    }
    if (a->variant != 0 || a->reserved != 0) {
        set_error("svdq_gemm_w4a4: variant and reserved must be 0 (timing experiments live in tools/ablate, not in this library)");
        return SVDQ_E_INVALID;
    }
    if (a->geometry < 0 || a->geometry > 8) { set_error("svdq_gemm_w4a4: geometry must be 0 (auto) .. 8"); return SVDQ_E_INVALID; }
    if (a->lora_act_format != SVDQ_LORA_ACT_F32 && a->lora_act_format != SVDQ_LORA_ACT_Q32 && a->lora_act_format != SVDQ_LORA_ACT_Q32_RUNS) { set_error("svdq_gemm_w4a4: unknown lora_act_format %d", a->lora_act_format); return SVDQ_E_INVALID; }
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
        if (a->out_vt && (a->ldvt < a->M || a->ldvt % 2 || ((uintptr_t)a->out_vt & 3))) {
            set_error("svdq_gemm_w4a4: out_vt needs an even ldvt=%d >= M=%d and a 4-byte aligned pointer", a->ldvt, a->M);
            return SVDQ_E_INVALID;
        }
        break;
    default:
        set_error("svdq_gemm_w4a4: unknown fuse mode %d", a->fuse);
        return SVDQ_E_INVALID;
    }
    if (a->wgt2) { // grouped launch: a second weight set for rows >= split_rows
        if (!a->wscales2 || a->split_rows <= 0 || a->split_rows >= a->M_pad || a->split_rows % 256) {
            set_error("svdq_gemm_w4a4: grouped launch needs wscales2 and 0 < split_rows=%d < M_pad=%d, a multiple of 256", a->split_rows, a->M_pad);
            return SVDQ_E_INVALID;
        }
        if ((a->bias != nullptr) != (a->bias2 != nullptr) || (a->R > 0 && !a->lora_up2)) {
            set_error("svdq_gemm_w4a4: grouped launch: bias2 / lora_up2 must mirror bias / lora_up");
            return SVDQ_E_INVALID;
        }
        if (a->fuse == SVDQ_FUSE_GELU_QUANT && (!a->next_smooth2 || (a->R2 > 0 && !a->next_lora_down2))) {
            set_error("svdq_gemm_w4a4: grouped GELU_QUANT launch needs next_smooth2 (and next_lora_down2 when R2 > 0)");
            return SVDQ_E_INVALID;
        }
        if (a->fuse == SVDQ_FUSE_RMSNORM_ROPE && (!a->norm_q2 || !a->norm_k2)) {
            set_error("svdq_gemm_w4a4: grouped RMSNORM_ROPE launch needs norm_q2 and norm_k2");
            return SVDQ_E_INVALID;
        }
        if (((uintptr_t)a->wgt2 | (uintptr_t)a->wscales2 | (uintptr_t)a->lora_up2) & 15 ||
            ((uintptr_t)a->bias2 | (uintptr_t)a->next_smooth2 | (uintptr_t)a->next_lora_down2 | (uintptr_t)a->norm_q2 | (uintptr_t)a->norm_k2) & 7) {
            set_error("svdq_gemm_w4a4: second weight set: wgt2, wscales2, lora_up2 must be 16-byte aligned, the vectors 8-byte");
            return SVDQ_E_INVALID;
        }
    }
    if (a->out && (a->ldo < a->N || a->ldo % 4)) { set_error("svdq_gemm_w4a4: ldo=%d must be >= N and a multiple of 4", a->ldo); return SVDQ_E_INVALID; }
    if (((uintptr_t)a->act | (uintptr_t)a->wgt | (uintptr_t)a->ascales | (uintptr_t)a->wscales | (uintptr_t)a->lora_up |
         (uintptr_t)a->lora_act_in | (uintptr_t)a->qout | (uintptr_t)a->rotary_emb) & 15) {
        set_error("svdq_gemm_w4a4: act, wgt, ascales, wscales, lora_up, lora_act_in, qout, rotary_emb must be 16-byte aligned");
        return SVDQ_E_INVALID;
    }
    if (a->workspace && (((uintptr_t)a->workspace & 15) || a->workspace_bytes < 0)) {
        set_error("svdq_gemm_w4a4: workspace must be 16-byte aligned with a non-negative size");
        return SVDQ_E_INVALID;
    }
    if (((uintptr_t)a->out | (uintptr_t)a->bias | (uintptr_t)a->next_smooth | (uintptr_t)a->next_lora_down |
         (uintptr_t)a->norm_q | (uintptr_t)a->norm_k | (uintptr_t)a->lora_act_out) & 7) {
        set_error("svdq_gemm_w4a4: out, bias, next_smooth, next_lora_down, norm_q, norm_k, lora_act_out must be 8-byte aligned");
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
    p.wgt2 = (const uint8_t *)a->wgt2; p.wscales2 = a->wscales2; p.bias2 = a->bias2; p.lora_up2 = a->lora_up2;
    p.next_smooth2 = a->next_smooth2; p.next_lora_down2 = a->next_lora_down2; p.norm_q2 = a->norm_q2; p.norm_k2 = a->norm_k2;
    p.split_row = a->wgt2 ? a->split_rows : 0x7fffffff;
    p.out_vt = a->fuse == SVDQ_FUSE_RMSNORM_ROPE ? a->out_vt : nullptr;
    p.ldvt = a->ldvt;
    p.workspace = (uint8_t *)a->workspace;
    p.workspace_bytes = a->workspace_bytes;
    p.M = a->M; p.M_pad = a->M_pad; p.N = a->N; p.K = a->K; p.R = a->R; p.R2 = a->R2; p.ldo = a->ldo;
    p.lora_fixed = a->lora_act_format;
    // the 256 x 128 geometry stages a tile's low-rank operands in LDS when they are whole 1 KiB pieces: rank 32, fp32, 16-byte aligned
    p.stage_lora = a->R == 32 && a->lora_act_format == SVDQ_LORA_ACT_F32 && a->lora_act_in && a->lora_up &&
                   (((uintptr_t)a->lora_act_in | (uintptr_t)a->lora_up | (uintptr_t)a->lora_up2) & 15) == 0;
    // ... or, beyond rank 32, the tile's lora_up for every rank (Geo::STG_LU_ALL_MAX_R): value = the gather's divider ceil(65536 / (R / 8 + 1))
    p.stage_lu_all = 0;
    if (a->R > 32 && a->R <= Geo<8>::STG_LU_ALL_MAX_R && a->lora_act_format == SVDQ_LORA_ACT_F32 && a->lora_act_in && a->lora_up &&
        (((uintptr_t)a->lora_act_in | (uintptr_t)a->lora_up | (uintptr_t)a->lora_up2) & 15) == 0)
        p.stage_lu_all = 65536 / (a->R / 8 + 1) + 1;
    p.la_packed = nullptr; // (set below once the workspace is known to hold the image)
    p.status = a->status;
    p.q_scale = a->q_scale == 0.f ? 1.0f : a->q_scale;
    for (int i = 0; i < MAX_LORA_TILES; i++) p.lora_scales[i] = (a->lora_scales && i < a->R / 16) ? a->lora_scales[i] : 1.0f;
    SVDQ_PROBE_FILL(p);

    const bool with_ws = p.workspace && p.workspace_bytes >= workspace_bytes_needed();
    if (p.stage_lu_all && with_ws && (long long)a->M_pad * a->R * 2 <= LA_PACK_BYTES) p.la_packed = p.workspace + workspace_slab_bytes();
    else p.stage_lu_all = 0; // (no workspace, or an image beyond its tail: the rank > 32 fallback loads of the plain kernels)
    p.lu_packed = nullptr;
    int geo = pick_geometry(a, with_ws);
    // GELU_QUANT with a next-layer low-rank branch beyond rank 32 (fp32 accumulators) and at least two 128 x 128 tiles per CU: the solo-carry kernel
    // (geometry 6 asks for it at any size and any next-layer rank <= 128: tests; launches it cannot serve run as with geometry 0)
    const bool solo_ok = a->fuse == SVDQ_FUSE_GELU_QUANT && a->R2 > 0 && a->R2 <= 128 && a->lora_act_format == SVDQ_LORA_ACT_F32;
    // (measured, profiles/r5_rank_ab.txt: next rank 128: 559 us against 615 us with per-tile atomics on 256 x 128 tiles; next rank 48: 310 against 285 -- one wave
    //  per SIMD runs the VALU-bound GELU epilogue at half the issue rate, which only pays once the atomics of >= 96 ranks are what it replaces)
    p.solo_carry = solo_ok && (a->geometry == 6 || (a->geometry == 0 && a->R2 >= 96 && (long long)(a->M_pad / 128) * (a->N / BN) >= 2LL * device_cus()));
    // ... or, with a workspace that holds the launch's 16-bit output image behind the packed operands (svdq_gemm_workspace_bytes_for), the split low-rank down
    // projection: the all-rank kernel on 256 x 128 tiles stores fragments, lowrank_down_split_kernel contracts them (geometry 7 asks for it at any size and from
    // next-layer rank 48: tests, A/B).  It takes the solo-carry kernel's launches: rank-128 fc1 of FLUX 380 -> ~250 us (profiles/r5_split_down_ab.txt).
    p.act16_packed = nullptr;
    p.split_R2 = 0;
    if (p.la_packed && split_down_shape_ok(a) && p.workspace_bytes >= workspace_bytes_needed() + (long long)a->M_pad * a->N * 2 &&
        (a->geometry == 7 || (a->geometry == 0 && split_down_auto(a)))) {
        p.act16_packed = p.workspace + workspace_bytes_needed();
        p.split_R2 = a->R2;
        p.R2 = 0; // (the GEMM kernel neither loads nor contracts the down projection)
        p.solo_carry = 0;
        geo = 1;
    } else if (geo == 7) { svdq_gemm_args b = *a; b.geometry = 0; geo = pick_geometry(&b, with_ws); }
    if (p.solo_carry) {
        geo = 3;
        // its low-rank up projection reads packed fragments of both operands when they fit the workspace tail (rank 48 .. 160, fp32, one weight set)
        if (p.la_packed && !a->wgt2 && (long long)a->N * a->R * 2 <= LU_PACK_BYTES) p.lu_packed = p.workspace + workspace_slab_bytes() + LA_PACK_BYTES;
        else p.la_packed = nullptr;
    }
    else if (geo == 6) { svdq_gemm_args b = *a; b.geometry = 0; geo = pick_geometry(&b, with_ws); }
    // rank 48 .. 160 on 128 x 128 tiles (no LDS to stage in): both low-rank operands as packed fragments when they fit the workspace tail (one weight set)
    if (!p.solo_carry && geo != 1 && p.la_packed && !a->wgt2 && (long long)a->N * a->R * 2 <= LU_PACK_BYTES) p.lu_packed = p.workspace + workspace_slab_bytes() + LA_PACK_BYTES;
    p.dynamic = geo == 2 || geo == 4;
    p.stagger = geo == 4 || geo == 5;
    // ABI 21: the caller's own fragment images replace the workspace copies (and their per-launch pack kernels)
    p.ld_pre = a->next_lora_down_packed; p.ld2_pre = a->next_lora_down_packed2;
    p.lu_pre = p.lu_packed != nullptr && a->lora_up_packed != nullptr;
    if (p.lu_pre) p.lu_packed = a->lora_up_packed;
    hipStream_t st = (hipStream_t)stream;
    const int prof = prof_begin(SVDQ_PROF_GEMM_VARIANT(a->fuse), 2.0 * a->M_pad * (double)a->N * a->K + 2.0 * a->M_pad * (double)a->N * a->R, st);
    // the plain epilogue at rank 0 / 32 with a long K (fc2): the 128 x 64 wave tile, one wave per SIMD -- where the rule above picked 256 x 128 tiles
    // (an explicit geometry 1 keeps the 8-wave kernel: tests, same-box A/B)
    // (end of round 6: since the scale tile runs on the K = 8 MFMA form and the product MFMA lost its MX scales, the 8-wave loop gained more than the wave-tile
    //  loop -- at K = 3072 the 8-wave kernel is the faster one again (51.8-53.1 us against 53.4-53.9), at K = 12288 the wave-tile kernel stays ahead (164.7-167.5
    //  against 169.1-172.4): the library's own choice takes it from K = 8192; profiles/r6_gemm_wave_tile_probe.txt section 8)
    if (wt128_serves(a) && (a->geometry == 8 || (a->geometry == 0 && geo == 1 && a->K >= 8192))) {
        if (a->dtype == SVDQ_BF16) launch_wt128<SVDQ_BF16>(p, with_ws, st);
        else launch_wt128<SVDQ_FP16>(p, with_ws, st);
    } else if (geo == 1) {
        if (a->dtype == SVDQ_BF16) launch_fuse<SVDQ_BF16, 8>(p, a->fuse, with_ws, st);
        else launch_fuse<SVDQ_FP16, 8>(p, a->fuse, with_ws, st);
    } else {
        if (a->dtype == SVDQ_BF16) launch_fuse<SVDQ_BF16, 4>(p, a->fuse, with_ws, st);
        else launch_fuse<SVDQ_FP16, 4>(p, a->fuse, with_ws, st);
    }
    prof_end(prof, st);
    return hip_check(hipGetLastError(), "svdq_gemm_w4a4 launch");
}
