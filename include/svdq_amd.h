/*
 * svdq_amd.h -- C ABI of the MI355X (gfx950) SVDQuant W4A4 + low-rank hot path.
 *
 * This is the drop-in boundary for the two operators the reference exports from its pybind11
 * module `nunchaku._C.ops` (reference: nunchaku/csrc/pybind.cpp:108-116, nunchaku/csrc/ops.h):
 *
 *   ops.quantize_w4a4_act_fuse_lora  (csrc/ops.h:83-112 -> src/kernels/zgemm/zgemm.h:39-46)
 *   ops.gemm_w4a4                    (csrc/ops.h:10-81  -> src/kernels/zgemm/zgemm.h:8-36)
 *
 * plus the load-time re-layout of the reference's checkpoint tensors (which are stored in NVIDIA
 * mma fragment order, nunchaku/lora/flux/packer.py:187-301,362-437) into the CDNA4 operand images
 * the kernels consume.
 *
 * Operand images.  4-bit codes (weights and activations) are held as the register image of
 * v_mfma_scale_f32_32x32x64_f8f6f4 with FP6 (e2m3) operands -- a 4-bit code is exactly an FP6 value --
 * i.e. 6 bits per code: a [ROWS, K] code matrix occupies ROWS*K*3/4 bytes (ROWS % 32 == 0,
 * K % 128 == 0).  This is 1.5x the reference's ROWS*K/2 nibble buffers; the host shim
 * (nunchaku_amd/ops) allocates the opaque activation buffers accordingly.  The bit-level layout is
 * documented in nunchaku_amd/csrc/svdq_common.h and DESIGN.md.
 *
 * Rules of the boundary (mirrors the reference's behaviour):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless stated otherwise;
 *     NULL means "tensor absent" (the reference's `std::optional<Tensor>` / `Tensor::valid()`).
 *   - the caller owns and allocates every buffer; the library allocates nothing and keeps no
 *     pointer after a call returns (reference: ops/gemm.py:107-110, interop/torch.h:8-23).
 *   - every call is asynchronous on the hipStream_t passed as `stream` (reference launches on the
 *     current torch stream, interop/torch.cpp:84-91; no device sync, csrc/ops.h:80).
 *   - errors are returned, never abort()ed (the reference asserts: launch_impl.cuh:44-59).
 *     0 = success; non-zero = SVDQ_E_*; svdq_last_error() gives a thread-local message.
 *   - re-entrant; no global mutable state besides the thread-local error string.
 *
 * "16-bit" below means the model dtype: SVDQ_BF16 or SVDQ_FP16 (the reference picks the kernel
 * dtype from ascales.dtype(), gemm_w4a4.cu:63-65).
 */
#ifndef SVDQ_AMD_H
#define SVDQ_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVDQ_ABI_VERSION 22

/* model dtype of the 16-bit tensors */
enum { SVDQ_BF16 = 0, SVDQ_FP16 = 1 };

/* error codes */
enum {
    SVDQ_OK = 0,
    SVDQ_E_INVALID = 1,     /* bad shape / alignment / missing tensor            */
    SVDQ_E_UNSUPPORTED = 2, /* valid in the reference, not implemented here (fp4, LiteLA, ...) */
    SVDQ_E_HIP = 3          /* a HIP runtime call failed                                    */
};

/* epilogue selector of svdq_gemm_w4a4 (the reference infers it from which tensors are valid,
 * gemm_w4a4_launch_impl.cuh:282-423; the host shim does the same inference and fills `fuse`). */
enum {
    SVDQ_FUSE_NONE = 0,        /* EpilogueDefault: store out[M,N]                (launch_impl.cuh:407-420) */
    SVDQ_FUSE_SILU = 1,        /* EpilogueSilu then store                        (gemm_base.cuh:783-792)   */
    SVDQ_FUSE_GELU_QUANT = 2,  /* GELU -> [LoraDown] -> shift+smooth+u4 requant  (launch_impl.cuh:282-309) */
    SVDQ_FUSE_RMSNORM_ROPE = 3 /* QKV: RMSNorm(q,k)+RoPE then store              (launch_impl.cuh:347-405) */
};

/* lora_act formats.  The low-rank activations lora_act[M_pad, R] are an OPAQUE hand-off between the kernel that produces
 * them (the quantiser, a GELU_QUANT epilogue, the attention kernel's fused quantiser) and the GEMM that consumes them
 * (the reference's order is NVIDIA-fragment specific, lora.cuh:61-80; only the size M_pad*R*4 bytes is visible to callers).
 *   SVDQ_LORA_ACT_F32: natural [m][r] fp32; partial sums (K slices, column tiles, heads) are combined with fp32 atomics,
 *                      so the last bits depend on their arrival order -- the reference's own behaviour (lora.cuh:82-94,323).
 *   SVDQ_LORA_ACT_Q32: natural [m][r] int64 holding value * 2^32 (Q31.32 fixed point), M_pad*R*8 bytes; partial sums are
 *                      combined with 64-bit INTEGER atomics, which are associative: the result is bit-reproducible from
 *                      run to run ("deterministic mode").  Producer and consumer must agree; the buffer must be zeroed
 *                      (all-zero bytes are 0.0 in both formats).  The sums do not depend on the launch configuration
 *                      either (tile geometry, grid, stream-K): "strict".
 *   SVDQ_LORA_ACT_Q32_RUNS (ABI 22): the same buffers and the same integer atomics BETWEEN workgroups, but a GELU_QUANT
 *                      launch on 256 x 128 tiles may first sum the column tiles one workgroup walks in a row (a "row run",
 *                      svdq_gemm_last_plan variant 1) in fp32 in LDS, in a fixed order, and convert the run's sum once: no
 *                      atomics inside a tile.  Bit-reproducible from run to run and between replicas for a given (shape,
 *                      geometry, device) -- what the deterministic mode is used for -- at about half the cost of the strict
 *                      form; NOT equal to the sums of another geometry or grid.  svdq_quantize_w4a4_act_fuse_lora and
 *                      svdq_attention treat it as SVDQ_LORA_ACT_Q32 (their partial sums have no such run). */
enum { SVDQ_LORA_ACT_F32 = 0, SVDQ_LORA_ACT_Q32 = 1, SVDQ_LORA_ACT_Q32_RUNS = 2 };

/* ------------------------------------------------------------------------------------------
 * svdq_quantize_w4a4_act_fuse_lora
 * replaces kernels::quantize_w4a4_act_fuse_lora (zgemm.h:39-46, gemm_w4a4.cuh:1097-1184).
 *
 *   lora_act[M_pad, R] = x @ lora_down            (on the un-smoothed input, fp32)
 *   x_hat = round16(x / smooth);  per (row, 64-channel group): scale = amax/7,
 *   q = sat_s4(rne(x_hat / scale));  rows >= M are treated as zero.
 *
 * act and ascales are OPAQUE (operand images private to this library):
 * act M_pad*K*3/4 bytes (FP6 image), ascales (K/64)*M_pad 16-bit (scale image), lora_act M_pad*R
 * fp32 (natural [m][r], true values).
 * ------------------------------------------------------------------------------------------ */
typedef struct svdq_quantize_args {
    const void *x;         /* [M, K] 16-bit, row stride ldx elements                          */
    const void *smooth;    /* [K] 16-bit, natural order (see svdq_repack_vec); NULL = ones    */
    const void *lora_down; /* [K*R] 16-bit in svdq_repack_lowrank(down=1) order; NULL if R==0 */
    void *act;             /* out: FP6 image of the int4 codes, M_pad*K*3/4 bytes            */
    void *ascales;         /* out: (K/64)*M_pad 16-bit                                        */
    void *lora_act;        /* out: M_pad*R fp32 -- or int64, see lora_act_format -- (zeroed + accumulated inside the call) */
    int32_t M;             /* actual rows                                                     */
    int32_t M_pad;         /* multiple of 256, >= M                                           */
    int32_t K;             /* multiple of 128                                                 */
    int32_t R;             /* multiple of 16 (0 = no low-rank branch)                         */
    int32_t ldx;           /* row stride of x in elements (>= K, multiple of 4)               */
    int32_t dtype;         /* SVDQ_BF16 | SVDQ_FP16                                           */
    int32_t fuse_glu;      /* non-zero: x holds rows of 2K interleaved (value, gate) pairs (ldx >= 2K, a multiple of 8; x 16-byte
                              aligned) and the op quantises round16(value * round16(silu(gate))) -- the reference's
                              load_act_to_fpsum<fuse_glu> (gemm_base.cuh:606-633).  Not with ln_stats or x2.               */
    int32_t fp4;           /* must be 0 (NVFP4 is Blackwell-only)                             */
    /* Optional fused AdaLayerNormZero front end (extension; all three or none).  The quantiser then reads
     *   x' = round16(round16(round16((x - mean) * rstd) * mod_scale) + mod_shift)
     * in place of x -- the rounding points of the reference's 16-bit torch ops `norm(x) * scale[:, None] +
     * shift[:, None]` (nunchaku/models/normalization.py:85-98,155-165, transformer_flux_v2.py:233-234; nunchaku
     * checkpoints carry the +1 of the scale inside the modulation bias, scale_shift = 0) -- for the low-rank
     * projection and the codes alike. */
    const float *ln_stats; /* [M, 2] fp32 (mean, rstd) per row, e.g. from svdq_residual_gate_stats     */
    const void *mod_scale; /* [K] 16-bit                                                              */
    const void *mod_shift; /* [K] 16-bit                                                              */
    /* Grouped launch (optional, x2 != NULL): rows [split_rows, split_rows + M2) of the OUTPUT buffers come from a
     * second input x2 (row stride ldx2) quantised with a second parameter set -- the text and image stream of a joint
     * block in one launch, feeding svdq_gemm_args.wgt2.  Then M must equal split_rows (a multiple of 256). */
    const void *x2, *smooth2, *lora_down2, *mod_scale2, *mod_shift2;
    const float *ln_stats2;
    int32_t M2, ldx2, split_rows;
    int32_t lora_act_zeroed; /* non-zero: the caller guarantees lora_act is already zero on this stream (e.g. cleared by
                                svdq_residual_gate_stats' zero_ptr), so the hipMemsetAsync of the K-sliced reduction is skipped */
    int32_t lora_act_format; /* SVDQ_LORA_ACT_F32 (lora_act: M_pad*R fp32) | SVDQ_LORA_ACT_Q32 (M_pad*R int64, deterministic) */
    int32_t reserved;
} svdq_quantize_args;

int svdq_quantize_w4a4_act_fuse_lora(const svdq_quantize_args *args, void *stream);

/* ------------------------------------------------------------------------------------------
 * svdq_gemm_w4a4
 * replaces kernels::gemm_w4a4 (zgemm.h:8-36, gemm_w4a4_launch_impl.cuh:7-424).
 *
 *   y[m,n] = sum_g ascales[g,m]*wscales[g,n] * (sum_{k in g} act[m,k]*wgt[n,k])      (int4 x int4)
 *            + bias[n] + sum_r round16(lora_act_in[m,r]*lora_scales[r/16]) * lora_up[n,r]
 *   then the epilogue selected by `fuse`.
 * ------------------------------------------------------------------------------------------ */
typedef struct svdq_gemm_args {
    const void *act;          /* FP6 image of the activations (quantize or a GELU_QUANT gemm)  */
    const void *wgt;          /* FP6 image of the weights (svdq_repack_qweight), N*K*3/4 bytes */
    const void *ascales;      /* (K/64)*M_pad 16-bit, scale image                              */
    const void *wscales;      /* (K/64)*N 16-bit, scale image (svdq_repack_wscales)            */
    const void *bias;         /* [N] 16-bit natural order or NULL                              */
    const void *lora_act_in;  /* M_pad*R fp32 (or int64, see lora_act_format) or NULL          */
    const void *lora_up;      /* [N, R] 16-bit natural row-major (svdq_repack_lowrank(down=0)) */
    const float *lora_scales; /* HOST pointer, R/16 floats, or NULL (= all 1.0)                */
    void *out;                /* [M, N] 16-bit, row stride ldo; required unless GELU_QUANT     */
    /* SVDQ_FUSE_GELU_QUANT: this layer emits the NEXT layer's quantized activation */
    void *qout;               /* FP6 image of the uint4 codes, M_pad*N*3/4 bytes               */
    void *oscales;            /* (N/64)*M_pad 16-bit, scale image                              */
    const void *next_smooth;  /* [N] 16-bit natural order                                      */
    const void *next_lora_down; /* [N*R2] 16-bit, svdq_repack_lowrank(down=1) order, or NULL   */
    void *lora_act_out;       /* M_pad*R2 fp32 (or int64, see lora_act_format); MUST be zeroed by the caller on the
                                 same stream (reference: lora_act_out.zero_(), launch_impl.cuh:252) */
    /* SVDQ_FUSE_RMSNORM_ROPE */
    const void *norm_q;       /* [128] 16-bit                                                   */
    const void *norm_k;       /* [128] 16-bit                                                   */
    const float *rotary_emb;  /* [M_pad, 128] fp32 in the reference's pack_rotemb order
                                 (models/embeddings.py:100-138)                                 */
    int32_t M;                /* actual rows (rows >= M are not stored to `out`)                */
    int32_t M_pad;            /* multiple of 256                                                */
    int32_t N;                /* multiple of 128                                                */
    int32_t K;                /* multiple of 128                                                */
    int32_t R;                /* rank of lora_up (multiple of 16 in [0, 256]; 0 = none).  Every rank runs the fused path; 32 (the SVDQuant default),
                               * 48 .. 160 (the r128 checkpoints, a runtime LoRA on top of rank 32) have kernels of their own */
    int32_t R2;               /* rank of next_lora_down (multiple of 16 in [0, 256]; 0 = none)  */
    int32_t ldo;              /* row stride of out in elements (>= N)                           */
    int32_t dtype;            /* SVDQ_BF16 | SVDQ_FP16                                          */
    int32_t act_unsigned;     /* informational: the FP6 image already encodes signedness        */
    int32_t fuse;             /* SVDQ_FUSE_*                                                    */
    int32_t variant;          /* must be 0 */
    int32_t reserved;         /* must be 0 */
    /* optional scratch for the stream-K tail (svdq_gemm_workspace_bytes() bytes, zero-filled ONCE by the
     * caller, then reusable by every later call ON THE SAME STREAM; NULL = whole-tile schedule only).
     * One workspace must never be used by launches that can be in flight concurrently (two streams, two
     * threads on different streams): keep one per (device, stream), as nunchaku_amd/_C.py does.  A violated
     * contract cannot hang the GPU -- an owner's wait is bounded -- but it invalidates the results; it is
     * reported by svdq_gemm_workspace_status(). */
    void *workspace;
    int64_t workspace_bytes;
    /* SVDQ_FUSE_RMSNORM_ROPE only, optional: the V third of the output (columns [2N/3, N)) is written
     * TRANSPOSED, element (m, n) -> out_vt[(n - 2N/3) * ldvt + m], and NOT to `out` -- the key-contiguous
     * operand svdq_attention reads (role of the reference's packed out_v, epilogues.cuh:427-550).  A joint
     * (text + image) attention passes the same buffer to both GEMMs with out_vt offset by the token start. */
    void *out_vt;
    int32_t ldvt;             /* row stride of out_vt in elements (>= total tokens)              */
    /* Workgroup geometry: 0 = chosen by the library; 1 = 256 x 128 tiles, one 512-thread workgroup per CU, fixed tile list
     * per workgroup + stream-K tail; 2 = 128 x 128 tiles, two 256-thread workgroups per CU (one's epilogue and barrier
     * stalls under the other's main loop) that DRAW their tiles from per-XCD queues in the workspace (falls back to 3 without
     * a workspace, for launches of at most one tile per workgroup, and where a stream-K split pays); 3 = 128 x 128 tiles,
     * fixed tile lists + stream-K tail; 4 / 5 = 2 / 3 with the second workgroup of a CU started half a tile late (A/B
     * measurements); 6 = GELU_QUANT launches with a next-layer low-rank branch of rank <= 128 (fp32 accumulators) run 128 x 128 tiles with ONE
     * workgroup per CU, whose LDS then holds the low-rank-down carry of a whole run of column tiles (what geometry 0 picks for next-layer ranks
     * 48 .. 128 at two or more tiles per CU; 6 asks for it at any size -- tests), every other launch as with 0; 7 (ABI 20) = GELU_QUANT launches with a
     * next-layer low-rank branch of rank 48 .. 160 run it SPLIT (256 x 128 tiles whose epilogue stores the 16-bit GELU output as MFMA fragments into the
     * workspace, a second kernel contracts them with next_lora_down; needs a workspace of svdq_gemm_workspace_bytes_for() bytes and R in 48 .. 160; what
     * geometry 0 picks for those ranks from a full round of tiles; 7 asks for it at any size), every other launch as with 0;
     * 8 (ABI 21) = the plain epilogue at rank 0 / 32 (fp32 low-rank accumulators) runs the 128 x 64-per-wave kernel (plan variant 6), every other launch as with 0.
     * Results are bit-identical across geometries for launches without a stream-K split (the split points, hence the fp32 summation order of a split tile, differ); lora_act_out
     * differs by fp32 summation order between all of them (atomics). */
    int32_t geometry;
    /* Grouped launch (optional): rows [split_rows, M_pad) use a SECOND weight set of the same shape, rank and
     * epilogue -- one launch for the text and the image stream of a joint FLUX block (same layer type, different
     * weights, transformer_flux_v2.py:200-260), whose row-side tensors (act, ascales, lora_act_in, out, qout,
     * oscales, lora_act_out, rotary_emb, out_vt) are simply the two streams' buffers back to back.  The 48..192
     * tiles of the 512-token stream then fill the tail of the big GEMM's last round instead of running as a
     * launch-latency-bound GEMM of their own.  wgt2 == NULL: off.  bias / bias2 must both be given or both NULL. */
    const void *wgt2, *wscales2, *bias2, *lora_up2;
    const void *next_smooth2, *next_lora_down2; /* GELU_QUANT */
    const void *norm_q2, *norm_k2;              /* RMSNORM_ROPE */
    int32_t split_rows;       /* multiple of 256, 0 < split_rows < M_pad                         */
    int32_t lora_act_format;  /* SVDQ_LORA_ACT_F32 | SVDQ_LORA_ACT_Q32: format of lora_act_in AND lora_act_out */
    /* Optional host-visible status word (pinned, device-accessible host memory: hipHostMalloc / a torch pinned tensor), or
     * NULL.  A stream-K owner that gives up waiting for partial tiles (see `workspace`) stores 1 here with system scope:
     * the host can poll it WITHOUT synchronising -- nunchaku_amd/_C.py checks it before every launch on the same stream
     * and raises.  The co-residency the persistent schedule relies on: the whole grid (<= 1 or 2 workgroups per CU,
     * depending on the geometry) must become resident while owners wait -- true for a stream that owns the device, not for
     * a CU-masked stream or beside a long-running co-tenant kernel; pass workspace = NULL there. */
    int32_t *status;
    /* SVDQ_FUSE_RMSNORM_ROPE only (extension): the Q third (columns [0, N/3)) is multiplied by q_scale inside the epilogue, BEFORE
     * its single rounding to 16-bit (0 = off = 1.0; q_scale = 1.0 is bit-identical to off).  For svdq_attention_args.q_prescaled: the
     * attention kernel then needs no per-score scaling, and Q carries one rounding as in the reference (scaling an already rounded Q
     * inside the attention kernel would add a second one). */
    float q_scale;
    int32_t reserved2;        /* must be 0 */
    /* ABI 21, optional (NULL = the launch packs them itself, as ABI 19 / 20 did on every launch): the weight-side operands of the rank 48 .. 160 kernels as MFMA
     * fragments, packed ONCE per parameter with svdq_pack_lora_down / svdq_pack_lora_up (nunchaku_amd/_C.py caches them per storage and version, so a set_lora
     * re-packs).  next_lora_down_packed(2): used by a launch whose next-layer low-rank down projection runs split (plan variant 5); lora_up_packed: by the
     * kernels that read lora_up as fragments (128 x 128 tiles at rank 48 .. 160, the solo-carry kernel).  Must hold the image of the very tensor passed in
     * next_lora_down(2) / lora_up. */
    const void *next_lora_down_packed, *next_lora_down_packed2;
    const void *lora_up_packed;
} svdq_gemm_args;

int svdq_gemm_w4a4(const svdq_gemm_args *args, void *stream);

/* ABI 21: the fragment images above.  ld = [R2][N] rank-major (svdq_repack_lowrank(down=1) order), R2 in 48 .. 160, N % 256 == 0 -> out, svdq_pack_lora_down_bytes(N, R2)
 * bytes ([N / 16 units][ceil(R2 / 32) rank blocks][64 lanes][8] 16-bit, ranks >= R2 zero); lu = [N][R] natural -> out, svdq_pack_lora_up_bytes(N, R) bytes
 * ([N / 32][R / 16][64 lanes][8]).  Pure permutations (a weight changes only under a set_lora): pack once, pass with every launch. */
int64_t svdq_pack_lora_down_bytes(int32_t N, int32_t R2);
int svdq_pack_lora_down(const void *ld, void *out, int32_t N, int32_t R2, int32_t dtype, void *stream);
int64_t svdq_pack_lora_up_bytes(int32_t N, int32_t R);
int svdq_pack_lora_up(const void *lu, void *out, int32_t N, int32_t R, int32_t dtype, void *stream);
/* size of the GEMM workspace for the current device: 1023 arrival counters + 1 error word + 2 fp32 tiles of 256 x 128 per CU (the stream-K tail), and
 * -- ABI 19 -- a 24 MB tail in which a launch of rank 48 .. 160 keeps its low-rank operands as packed 16-bit MFMA fragments (written by a small pack
 * kernel the call enqueues in front of the GEMM; valid, like the rest, for launches ordered on ONE stream).  Without a workspace (or with one of the
 * ABI 18 size) such launches take the plain kernels' rank > 32 path: same results, a memory round trip per 16 ranks in every tile's epilogue. */
int64_t svdq_gemm_workspace_bytes(void);
/* ABI 20: the workspace size with which THIS launch takes every fast path it has: svdq_gemm_workspace_bytes(), plus M_pad * N * 2 bytes for a GELU_QUANT
 * launch whose next-layer low-rank down projection can run split (geometry 7 above; geometry 0 for next-layer ranks 48 .. 160 from a full round of tiles).  Only shapes, ranks, fuse, geometry,
 * lora_act_format and the operand pointers' alignment are read.  A workspace of svdq_gemm_workspace_bytes() bytes is never an error: such a launch then runs
 * the solo-carry / hybrid-carry kernels. */
int64_t svdq_gemm_workspace_bytes_for(const svdq_gemm_args *args);
/* What the calling thread's last svdq_gemm_w4a4 launched (ABI 19; a test / diagnostics aid, no effect on results): out8 = {tile rows (256 | 128), kernel variant,
 * grid, stream-K groups (0 = whole tiles), row-run length (0 = plain schedule), lora_act_in packed (0 | 1), lora_up packed (0 | 1), dynamic tile queue (0 | 1)}.
 * Variants: 0 plain (rank <= 32 staged / any rank through the fallback loads; GELU_QUANT: per-tile atomics), 1 low-rank-down carry (GELU_QUANT, next rank <= 32),
 * 2 all-rank (rank 48 .. 160: packed lora_act_in; 256-row tiles: lora_up staged in LDS, 128-row tiles: lora_up packed too), 3 hybrid carry (GELU_QUANT, next rank
 * 48 .. 80 -- or any next rank > 32 the solo kernel does not take; behind variant 5), 4 solo carry (GELU_QUANT, next rank >= 96: 128 x 128 tiles, one workgroup per CU; behind variant 5),
 * 5 split low-rank down (ABI 20: GELU_QUANT, next rank 48 .. 160 with a workspace of svdq_gemm_workspace_bytes_for() bytes: all-rank kernel on 256-row tiles
 * + lowrank_down_split_kernel), 6 wave tile 128 (ABI 21: the plain epilogue at rank 0 / 32 on 256 x 128 tiles with FOUR waves of 128 x 64 each, one per SIMD --
 * generated loop AND epilogue, bit-identical to variant 0; geometry 8 asks for it, geometry 0 picks it wherever it would have launched 256-row tiles). */
#define SVDQ_PLAN_PLAIN 0
#define SVDQ_PLAN_CARRY 1
#define SVDQ_PLAN_ALL_RANK 2
#define SVDQ_PLAN_HYBRID_CARRY 3
#define SVDQ_PLAN_SOLO_CARRY 4
#define SVDQ_PLAN_SPLIT_DOWN 5
#define SVDQ_PLAN_WAVE_TILE_128 6
int svdq_gemm_last_plan(int32_t *out8);
/* Synchronises `stream`, then returns SVDQ_E_HIP (and clears the flag) if a launch that used `workspace` timed out
 * waiting for partial tiles -- see svdq_gemm_args.workspace; SVDQ_OK otherwise.  Test / debugging aid. */
int svdq_gemm_workspace_status(void *workspace, void *stream);
/* Host-side replay of the kernel's persistent / stream-K schedule (test helper; no GPU needed): writes up to
 * `cap` records of 6 int32 {position, tile, kp0, kp1, partial slot or -1, contributors the owner waits for}
 * and returns the number of segments, or -1 for invalid shapes. */
int svdq_gemm_schedule(int32_t M_pad, int32_t N, int32_t K, int32_t cus, int32_t with_workspace, int32_t *out, int32_t cap);
/* the same for an explicit geometry (1 or 2, see svdq_gemm_args.geometry); svdq_gemm_schedule is geometry 1.
 * geometry 3 replays the ROW-RUN schedule a GELU_QUANT launch of geometry 1 takes when it has a next-layer low-rank branch of rank <= 32, no K
 * split and at least two tiles per workgroup (every workgroup walks one run of consecutive column tiles of one row block, so that the next
 * layer's low-rank down projection is accumulated inside the workgroup and written once per run): tile ids are row-major there
 * (tile = row_block * (N / 128) + column_tile); -1 when such a launch would take the plain schedule. */
int svdq_gemm_schedule_ex(int32_t M_pad, int32_t N, int32_t K, int32_t cus, int32_t with_workspace, int32_t geometry,
                          int32_t *out, int32_t cap);

/* ------------------------------------------------------------------------------------------
 * Attention over the packed QKV (reference: ops.attention_fp16, nunchaku/csrc/ops.h:114-121,
 * src/kernels/zgemm/attention.cu:11-94).  Non-causal, optional key-padding mask,
 * head_dim 128, one batch element per call:
 *   O[l, h, :] = softmax_j(scale * Q[l, h, :] . K[j, h, :]) V[j, h, :]
 * element addresses (in 16-bit elements):
 *   Q (l,h,d): q  + l*ldq  + h*q_hs  + d        K (j,h,d): k + j*ldk + h*k_hs + d
 *   V (j,h,d): vt + d*ldvt + h*vt_hs + j        (V TRANSPOSED: keys contiguous)
 *   O (l,h,d): out + l*ldo + h*o_hs  + d
 * With the fused QKV GEMM output [L, 3*H*128] (+ out_vt [H*128, L]): q = qkv, k = qkv + H*128,
 * ldq = ldk = 3*H*128, q_hs = k_hs = 128, vt_hs = 128*ldvt, out [L, H*128]: o_hs = 128, ldo = H*128.
 * L (queries = keys) must be a multiple of 128 (the models pad tokens to 256: fused.py:140-152).
 * ------------------------------------------------------------------------------------------ */
typedef struct svdq_attention_args {
    const void *q, *k, *vt;
    void *out;
    int64_t q_hs, k_hs, vt_hs, o_hs; /* head strides in elements */
    int32_t ldq, ldk, ldvt, ldo;     /* token (q, k, out) / channel (vt) strides in elements */
    int32_t L, H, head_dim, dtype;
    float scale;                     /* softmax scale, e.g. 1/sqrt(head_dim) */
    int32_t reserved;
    /* optional: zero-fill an unrelated scratch buffer in the same launch (the fp32 low-rank accumulators of the output
     * projection's quantiser, see svdq_residual_args.zero_ptr).  zero_bytes must be a multiple of 16. */
    void *zero_ptr;
    int64_t zero_bytes;
    /* Optional fused quantiser (extension): the output projection that follows reads the attention output only through
     * svdq_quantize_w4a4_act_fuse_lora, and a wave's 32 query rows x 128 channels of one head ARE one F6 chunk in the
     * register layout it already holds -- so the kernel can emit that projection's quantised activation directly
     * (same arithmetic as the quantiser on the 16-bit rounded output: bit-identical codes and scales; lora_act summed
     * over the H heads with fp32 atomics).  qact != NULL enables it; out may then be NULL.  Requires L % 256 == 0,
     * K = H*128, R a multiple of 16 <= 256 (32 ranks per pass; rank 48 .. 160 with a workspace of svdq_attention_workspace_bytes_for()
     * bytes: a contraction kernel behind the attention kernel, ABI 20), and qlora_act zeroed by the caller on this stream.
     * Rows >= qsplit_rows (a multiple of 256; 0 = off) use qsmooth2 / qlora_down2 (joint attention: text rows first). */
    void *qact;               /* FP6 image, L * (H*128) * 3/4 bytes                                 */
    void *qscales;            /* scale image, (H*128/64) * L 16-bit                                 */
    void *qlora_act;          /* [L, R] fp32 (or int64: qlora_act_format), pre-zeroed               */
    const void *qsmooth, *qlora_down;   /* [H*128] natural; [R][H*128] rank-major (svdq_repack_lowrank(down=1)) */
    const void *qsmooth2, *qlora_down2;
    int32_t qR, qsplit_rows;
    /* optional scratch of the persistent schedule (svdq_attention_workspace_bytes() bytes, zero-filled ONCE by the caller;
     * the kernel leaves its counters at zero).  A task = (head, 256 query rows) x all KV tiles; when the tasks do not fill
     * whole rounds of compute units (FLUX.1 at 1024^2: 432 tasks on 256 CUs) the launch becomes one workgroup per CU, the
     * KV tiles of all tasks dealt evenly, partial (O, m, l) exchanged through this buffer.  Same contract as
     * svdq_gemm_args.workspace: never shared by launches that can be in flight concurrently; a waiting workgroup gives up
     * after ~1 s and raises the error word svdq_attention_workspace_status() reports.  NULL (or too small): plain grid.
     * Results of the two schedules differ by fp32 summation order only. */
    void *workspace;
    int64_t workspace_bytes;
    int32_t qlora_act_format; /* SVDQ_LORA_ACT_F32 | SVDQ_LORA_ACT_Q32 (deterministic head sum) */
    int32_t q_prescaled;      /* non-zero: Q already carries the factor scale * log2(e) (svdq_gemm_args.q_scale of the QKV GEMM that wrote
                                 it); `scale` is then not applied again.  Either geometry. */
    int32_t *status;          /* optional host-visible status word, as svdq_gemm_args.status */
    /* Key-padding mask (optional; kv_len0 == 0: every one of the L keys is real).  Keys [0, kv_len0) and [kv_start1, kv_end1) are
     * real tokens, all others are padding and get probability 0 -- the token buffers of a pipeline are padded to a multiple of
     * 128 / 256 rows (L itself stays a multiple of 128), a joint [text | image] sequence padded per stream has its padding in the
     * middle: hence two ranges (kv_start1 = kv_end1 = 0: one range).  Padded K / Q rows may hold anything, including NaN; padded
     * V^T columns must be FINITE (0 * NaN is NaN in the matrix unit): zero them.  Output rows of padded queries are unspecified.
     * Role of the reference's padded-row masking (epilogues.cuh:427-550, attention.cuh). */
    int32_t kv_len0, kv_start1, kv_end1;
    /* workgroup geometry: 0 = automatic; 1 = 8 waves x 32 query rows (two waves per SIMD); 2 = 4 waves x 64 query rows (one wave per
     * SIMD with the whole register file, the tile loop in generated assembly; needs L % 256 == 0 and no key mask -- masked launches run
     * geometry 1).  Geometry 2 keeps its scores relative and in log2 units, which needs Q multiplied by scale * log2(e): with q_prescaled
     * the producer has done that (one rounding, as accurate as geometry 1); otherwise the kernel scales the 16-bit Q itself -- a second
     * rounding, ~2.5x the error of geometry 1 against an fp32 reference.  Hence automatic = geometry 2 (on the plain grid) iff
     * q_prescaled, L % 256 == 0 and no mask; an explicit geometry uses the persistent schedule when a workspace is given. */
    int32_t geometry;
    /* ABI 21, optional: qlora_down(2) as the fragment image of svdq_pack_lora_down (N = H * 128) for a launch whose fused quantiser runs its low-rank down
     * projection split (rank 48 .. 160 with the workspace of svdq_attention_workspace_bytes_for()); NULL = packed by the launch */
    const void *qlora_down_packed, *qlora_down_packed2;
} svdq_attention_args;

int svdq_attention(const svdq_attention_args *args, void *stream);
/* size of the persistent-schedule workspace for the current device (1023 arrival counters + 1 error word + two fp32 slabs per CU: the part a
 * contributor publishes and, for geometry 2, the part the owner of a split task parks while it runs its whole tasks) */
int64_t svdq_attention_workspace_bytes(void);
/* ABI 20: the workspace size with which THIS launch takes every fast path it has: svdq_attention_workspace_bytes(), plus -- fused quantiser (qact) with fp32
 * accumulators, rank 48 .. 160 and H * 128 a multiple of 256 -- the room its low-rank down projection needs to run SPLIT: the kernel's epilogue stores the
 * normalised 16-bit rows as MFMA operand fragments (L * H * 128 * 2 bytes) instead of contracting them in 2 .. 5 passes of atomics that all H heads aim at the same
 * elements, and a streaming kernel behind it contracts that image with qlora_down (the kernel svdq_gemm_w4a4 uses for a GELU_QUANT launch's next layer; packed
 * copies of qlora_down / qlora_down2 sit in front of the image).  Only shapes, ranks, formats and which pointers are given are read.  A workspace of
 * svdq_attention_workspace_bytes() bytes is never an error: the kernel then runs its in-epilogue passes.  qlora_act differs by fp32 summation order. */
int64_t svdq_attention_workspace_bytes_for(const svdq_attention_args *args);
/* Synchronises `stream`, then returns SVDQ_E_HIP (and clears the flag) if a launch that used `workspace` timed out waiting
 * for partial results; SVDQ_OK otherwise.  Test / debugging aid. */
int svdq_attention_workspace_status(void *workspace, void *stream);
/* Host-side replay of the persistent schedule for (L, H) on `cus` compute units: up to `cap` records of 6 int32
 * {workgroup, task = head * (L/256) + query tile, first KV tile, end KV tile, owner workgroup or -1 (this segment owns its
 * task), last contributing workgroup or -1 (no other contributor)}; returns the number of segments, 0 when the problem runs
 * on the plain grid (L % 256 != 0 or whole rounds of tasks), -1 on bad arguments.  No GPU needed. */
int svdq_attention_schedule(int32_t L, int32_t H, int32_t cus, int32_t *out, int32_t cap);
/* Which kernel a launch with these arguments would take (host only; pointers are not looked at): out[0] = workgroup geometry (1 / 2), out[1] = 1 when
 * the key mask runs on geometry 2 (round 4: q_prescaled or an explicit geometry 2, L % 256 == 0, a run of at least two fully real 64-key tiles),
 * out[2], out[3] = the main segment [j0, j1) of fully real tiles its assembly loop walks -- the remaining tiles that hold a real key run as C++ "extra"
 * tiles that continue the softmax state, tiles without a real key are skipped. */
int svdq_attention_plan(const svdq_attention_args *args, int32_t out[4]);
/* What the calling thread's last svdq_attention launched (ABI 20; a test / diagnostics aid): out4 = {workgroup geometry (1 | 2), workgroups of the persistent
 * schedule (0 = plain grid), key mask on geometry 2 (0 | 1), the fused quantiser's low-rank down projection ran split (0 | 1)}. */
int svdq_attention_last_plan(int32_t *out4);

/* ------------------------------------------------------------------------------------------
 * Gated residual + LayerNorm statistics (extension; the element-wise glue between the operators of a block,
 * transformer_flux_v2.py:118-342):
 *   t = b ? round16(a + b) : a;      if (gate) t = round16(gate[c] * t);      y = a ? round16(res + t) : res;   [fp16: optional clip]
 *   out[m, :] = y (if out);          stats[m] = (mean(y), 1/sqrt(var(y) + eps)) over the 16-bit values (if stats)
 * i.e. the reference's 16-bit torch ops `residual + gate.unsqueeze(1) * (attn [+ mlp])`
 * (transformer_flux_v2.py:230-251,332-335) with the statistics the next LayerNorm needs produced in the same pass.  out may alias res.  C must be a multiple of 8 with ceil(C/512) in {1..8, 12, 16, 24, 32}.
 * ------------------------------------------------------------------------------------------ */
typedef struct svdq_residual_args {
    const void *res;   /* [M, C] 16-bit, row stride ld */
    const void *a;     /* [M, C] or NULL (statistics of res only) */
    const void *b;     /* [M, C] or NULL */
    const void *gate;  /* [C] 16-bit or NULL */
    void *out;         /* [M, C] or NULL */
    float *stats;      /* [M, 2] fp32 or NULL */
    int32_t M, C, ld;  /* ld: common row stride of res / a / b / out in elements */
    int32_t dtype;
    float eps;
    int32_t clamp_fp16; /* SVDQ_FP16 only: bit 0 (first problem) / bit 1 (second problem) set: y is clipped to +-65504 before
                           the store and the statistics -- the reference's fp16 blocks clip the text stream after a joint
                           block and the hidden state after a single block (transformer_flux_v2.py:254-255, 339-340) */
    /* optional: zero-fill an unrelated scratch buffer in the same pass (the fp32 low-rank accumulators of the
     * quantiser / GELU_QUANT calls that follow: saves their memset launches).  zero_bytes must be a multiple of 16. */
    void *zero_ptr;
    int64_t zero_bytes;
    /* Grouped launch (optional, res2 != NULL): a second, independent problem of the same width C and row stride
     * (M2 rows; the text stream of a joint block beside the image stream) in the same launch.  a2 / b2 / gate2 /
     * out2 / stats2 must mirror a / b / gate / out / stats in being given or NULL. */
    const void *res2, *a2, *b2, *gate2;
    void *out2;
    float *stats2;
    int32_t M2;
    int32_t reserved2;
} svdq_residual_args;

int svdq_residual_gate_stats(const svdq_residual_args *args, void *stream);

/* ------------------------------------------------------------------------------------------
 * AWQ W4A16 GEMV (reference: ops.gemv_awq, nunchaku/csrc/ops.h:123-145 -> src/kernels/awq/gemv_awq.cu:100-286;
 * module nunchaku/models/linear.py:277-414).  The AdaLayerNormZero modulation projections of every block.
 *   out[m, n] = round16( sum_k round16( round16(q[n,k]*scales[k/64,n] + zeros[k/64,n]) * x[m,k] ) ) (+ bias[n])
 * qweight is the CHECKPOINT tensor as stored ([N/4, K/2] int32, tinychat pack_w4 order,
 * text_encoders/tinychat_utils.py:76-107) -- no load-time repack.  scales / zeros are [K/64, N] 16-bit,
 * zeros already scaled and negated (w = q*scale + zeros).  1 <= M <= 8 as in the reference.
 * bias (optional, [N] 16-bit) fuses AWQW4A16Linear.forward's output.add_(bias) (one more 16-bit rounding).
 * ------------------------------------------------------------------------------------------ */
typedef struct svdq_gemv_awq_args {
    const void *x;        /* [M, K] 16-bit, row stride ldx */
    const void *qweight;  /* [N/4, K/2] int32 */
    const void *scales;   /* [K/64, N] 16-bit */
    const void *zeros;    /* [K/64, N] 16-bit */
    const void *bias;     /* [N] 16-bit or NULL */
    void *out;            /* [M, N] 16-bit, contiguous */
    int32_t M, N, K, ldx;
    int32_t group_size;   /* must be 64 */
    int32_t dtype;        /* SVDQ_BF16 | SVDQ_FP16 */
    int32_t out_chunks;   /* 0 / 1: natural out[m, n].  c > 1 (N % c == 0): de-interleaved, element n is written to
                             out[m, (n % c) * (N / c) + n / c] -- the modulation vectors of AdaLayerNormZero are stored
                             interleaved per channel (emb.view(B, -1, 6).permute(2, 0, 1), normalization.py:89) and come
                             out as c contiguous [N/c] vectors */
    int32_t reserved;
} svdq_gemv_awq_args;

int svdq_gemv_awq(const svdq_gemv_awq_args *args, void *stream);
/* `count` (<= 80) independent GEMVs that share x, M, K, dtype and group_size in ONE launch (extension): all the
 * modulation projections of a denoising step depend only on the timestep embedding, so they can be issued together
 * before the first block instead of one launch per block.  Each entry brings its own qweight / scales / zeros / bias /
 * out / N / out_chunks; x, M, K, ldx, group_size and dtype are taken from args[0]. */
#define SVDQ_GEMV_BATCH_MAX 80
int svdq_gemv_awq_batched(const svdq_gemv_awq_args *args, int32_t count, void *stream);

/* ------------------------------------------------------------------------------------------
 * Load-time re-layout of reference checkpoint tensors (NVIDIA fragment order -> CDNA4 order).
 * src and dst must not alias.  All but svdq_repack_qweight (4 -> 6 bits per code) and svdq_repack_wscales (x 32) are pure permutations of equal size.
 * ------------------------------------------------------------------------------------------ */
/* qweight [N, K/2] int8 (packer.py:187-239) -> FP6 image, N*K*3/4 bytes, consumed by svdq_gemm_w4a4 */
int svdq_repack_qweight(const void *src, void *dst, int32_t N, int32_t K, void *stream);
/* wscales [K/64, N] 16-bit (packer.py:241-301) -> scale image [N/32][K/128][2][32]; G must be even.  ABI 21: takes the dtype, and the image holds 32 x the
 * scale: the GEMM's product MFMA runs without MX block scales (P = dot / 64), the scale tile is S = 2 ws' as, and ws' = 32 ws makes P S the fp32 product of
 * the ABI 20 kernels bit for bit.  Exact for bf16; an fp16 scale above 2047 overflows to inf (weights beyond 14 000) -- svdq_unrepack_wscales divides again. */
int svdq_repack_wscales(const void *src, void *dst, int32_t G, int32_t N, int32_t dtype, void *stream);
/* bias / smooth_factor [N] 16-bit (same intra-128 permutation, gemm_base.cuh:713) -> natural */
int svdq_repack_vec(const void *src, void *dst, int32_t N, void *stream);
/* proj_up [N, R] (down=0) -> natural [n][r];  proj_down [K, R] (down=1) -> [r][k] (rank-major)
 * (packer.py:362-398, lora.cuh:43-59).  C = N or K. */
int svdq_repack_lowrank(const void *src, void *dst, int32_t C, int32_t R, int32_t down, void *stream);

/* Inverse re-layouts (kernel order -> the reference's checkpoint order): what state_dict() of a repacked layer returns and
 * what the host-offload manager keeps in pinned memory (4-bit nibbles: 2/3 of the FP6 image's bytes on the PCIe link).
 * Exact inverses of svdq_repack_*: repack(unrepack(x)) == x and unrepack(repack(c)) == c bit for bit. */
int svdq_unrepack_qweight(const void *src, void *dst, int32_t N, int32_t K, void *stream); /* FP6 image -> [N, K/2] int8 */
int svdq_unrepack_wscales(const void *src, void *dst, int32_t G, int32_t N, int32_t dtype, void *stream);
int svdq_unrepack_vec(const void *src, void *dst, int32_t N, void *stream);
int svdq_unrepack_lowrank(const void *src, void *dst, int32_t C, int32_t R, int32_t down, void *stream);

/* Inverse helpers used by the tests to read the opaque formats back:
 * FP6 image -> one int8 code per element, natural [ROWS, K];  scale image -> natural [G][ROWS]. */
int svdq_unpack_act(const void *act, int8_t *codes, int32_t M_pad, int32_t K, int32_t is_unsigned, void *stream);
int svdq_unpack_scales(const void *simg, void *natural, int32_t ROWS, int32_t G, void *stream);

/* ------------------------------------------------------------------------------------------
 * Launch profiler (used by bench.py for the roofline line).  While enabled, every
 * svdq_gemm_w4a4 / svdq_quantize call brackets its kernel launch with two hipEvents recorded on
 * the launch stream.  svdq_prof_read synchronises the recorded events and returns, per kernel
 * class (0 = gemm_w4a4, 1 = quantize, 2 = attention: 4*L*L*H*128 flops, 3 = gemv_awq: bytes read), the number of launches, the summed kernel time in
 * milliseconds and the summed ALGORITHMIC work (gemm: 2*M_pad*N*K + 2*M_pad*N*R operations;
 * quantize: bytes read + written).  Process-global state, guarded by a mutex; off by default.
 * ------------------------------------------------------------------------------------------ */
int svdq_prof_enable(int32_t max_launches); /* 0 disables and frees the event pool */
int svdq_prof_reset(void);
/* Restrict the bracketing to the kernel classes whose bit is set (default: all).  An event pair costs ~3 us of
 * queue serialisation per launch, so bench.py brackets only the kernel its roofline line is about. */
int svdq_prof_select(uint32_t class_mask);
int svdq_prof_read(int32_t kernel_class, int64_t *launches, double *total_ms, double *total_work);
/* Sub-classes (ABI 18): a gemm_w4a4 launch is recorded under its epilogue variant.  svdq_prof_read(0, ...) still sums every GEMM launch;
 * svdq_prof_read(SVDQ_PROF_GEMM_VARIANT(fuse), ...) returns the launches of one epilogue (SVDQ_FUSE_NONE / _SILU / _GELU_QUANT / _RMSNORM_ROPE),
 * so that a bench line can show which variant moved. */
#define SVDQ_PROF_GEMM_VARIANT(fuse) (0 | (((fuse) + 1) << 8))

/* thread-local message of the last failing call on this thread ("" if none) */
const char *svdq_last_error(void);
/* SVDQ_ABI_VERSION the library was built with */
int svdq_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SVDQ_AMD_H */
