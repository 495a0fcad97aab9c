"""CPU oracle for the SVDQuant W4A4 + low-rank hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``nunchaku_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker.

It restates, in numpy, the arithmetic of the reference's CUDA kernels
(/root/reference, nunchaku v1.2.0dev).  Each function cites the reference
file:line it follows.

Parity pin
----------
The reference has NO kernel-level golden vectors or known-answer tests for
this path and its extension cannot be built here
(CUDA/PTX only).  What IS pinned against the reference's own code:

* every tensor *layout* codec below is checked against the reference's
  ``nunchaku/lora/flux/packer.py`` (imported by path in this container, see
  ``tools/make_golden.py``); the resulting fixtures live in ``tests/golden``.

The *arithmetic* (quantiser, per-group dequant, epilogues) is restated from
the CUDA source by reading it: **arithmetic parity is unpinned** by any
reference-run output (no NVIDIA GPU, no checkpoints).  The PTX approximations
``rcp.approx``, ``rsqrt.approx``, ``tanh.approx``, ``ex2.approx`` and
``__fdividef`` have no CPU twin; the oracle uses exact IEEE math in their
place (documented per function).

Approximation envelope (round 5)
--------------------------------
How far may the REFERENCE itself sit from this oracle?  Every PTX approximation on the
path has a documented error bound (``PTX_APPROX`` below).  The ``*_envelope`` functions
carry an interval through the same operation chain -- each approximate instruction's
result is widened by its documented bound, every exactly specified operation (16-bit
roundings, fp32 multiplies, rint, saturation) is applied to both ends (they are monotone)
-- and return, per output, the closed range [lo, hi] that ANY implementation whose
approximate instructions respect those bounds must land in: the reference's CUDA kernels,
the IEEE oracle above, and the MI355X kernels (``x * v_rcp_f32(smooth)`` for
``__fdividef``, exp2 + rcp for ``tanh.approx``).  Tests assert HIP outputs inside the
envelope AND a bounded flip rate against the IEEE values (< 1e-3 for
the quantiser).  The envelope is a statement about arithmetic only; it does not pin the
restatement itself (that still needs reference-run vectors, which do not exist).

Conventions
-----------
16-bit tensors ("half_t" in the reference: bf16 or fp16) are carried as
float32 numpy arrays whose values are exactly representable in that 16-bit
type; ``round16(x, dtype)`` performs the round-to-nearest-even conversion.
Packed on-device tensors are numpy ``int8``/``uint8``/``uint16`` arrays.
"""

from __future__ import annotations

import numpy as np

F32 = np.float32
GROUP = 64  # INT4 group size (gemm_base.cuh:89-95: WARP_K = 64 = one k-iteration)
PAD_M = 256  # BLOCK_M (gemm_base.cuh:34-41); Linear.cpp:445-446
PAD_N = 128  # BLOCK_N; Linear.cpp:92-93
GELU_SHIFT = 0.171875  # gemm_w4a4_launch_impl.cuh:286
RMS_EPS = 1e-6  # gemm_w4a4_launch_impl.cuh:374


# --------------------------------------------------------------------------
# 16-bit helpers
# --------------------------------------------------------------------------
def bf16_round(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest-even bfloat16, returned as float32."""
    x = np.ascontiguousarray(x, dtype=F32)
    u = x.view(np.uint32).astype(np.uint64)
    bias = ((u >> 16) & 1) + 0x7FFF
    r = ((u + bias) & 0xFFFF0000).astype(np.uint32)
    out = r.view(F32).copy()
    nan = np.isnan(x)
    if nan.any():
        out[nan] = np.nan
    return out.reshape(x.shape)


def fp16_round(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        return np.asarray(x, dtype=F32).astype(np.float16).astype(F32)


def round16(x: np.ndarray, dtype: str) -> np.ndarray:
    if dtype == "bf16":
        return bf16_round(x)
    if dtype == "fp16":
        return fp16_round(x)
    raise ValueError(dtype)


def to_bits16(x: np.ndarray, dtype: str) -> np.ndarray:
    """Exactly-representable float32 -> raw 16-bit storage (uint16)."""
    x = np.ascontiguousarray(x, dtype=F32)
    if dtype == "bf16":
        return (x.view(np.uint32) >> 16).astype(np.uint16)
    return x.astype(np.float16).view(np.uint16)


def from_bits16(b: np.ndarray, dtype: str) -> np.ndarray:
    b = np.ascontiguousarray(b, dtype=np.uint16)
    if dtype == "bf16":
        return (b.astype(np.uint32) << 16).view(F32)
    return b.view(np.float16).astype(F32)


def ceil_div(a: int, b: int) -> int:
    return (a + b - 1) // b


# --------------------------------------------------------------------------
# Reference ("on-disk") tensor layouts.  packer.py:187-301,362-437 and the
# device-side consumers gemm_base.cuh:265-355, lora.cuh:43-59.
# --------------------------------------------------------------------------
def _qweight_index(N: int, K: int):
    """For every (n, k) return (word index, nibble index) in the reference qweight.

    packer.py:187-239 (pack_weight): view (n_tiles, 8, 2, 8, 1, k_tiles, 1, 2, 4, 8),
    permute(0, 5, 6, 1, 3, 8, 2, 7, 4, 9); consumed by load_wgt, gemm_base.cuh:278-294.
    """
    n = np.arange(N)[:, None]
    k = np.arange(K)[None, :]
    nt, n_in = n // 128, n % 128
    npk, n16 = n_in // 16, n_in % 16
    n_pack, n_lane = n16 // 8, n16 % 8  # n = np*16 + n_pack*8 + n_lane
    kt, k_in = k // 64, k % 64
    k_pack, k32 = k_in // 32, k_in % 32
    k_lane, r = k32 // 8, k32 % 8  # k = kt*64 + k_pack*32 + k_lane*8 + r
    KT = K // 64
    lane = n_lane * 4 + k_lane
    j = n_pack * 2 + k_pack
    word = (((nt * KT + kt) * 8 + npk) * 32 + lane) * 4 + j
    word = np.broadcast_to(word, (N, K))
    nib = np.broadcast_to(r, (N, K))
    return word, nib


def pack_qweight_ref(q: np.ndarray) -> np.ndarray:
    """int values in [-8, 7], shape [N, K] -> reference packed int8 [N, K/2].

    Same map as :func:`_qweight_index` (which :func:`unpack_qweight_ref` and the tests use to cross-check it), written
    as the packer's own view / permute so that FLUX-sized weights pack in a fraction of a second:
    n = nt*128 + np*16 + n_pack*8 + n_lane, k = kt*64 + k_pack*32 + k_lane*8 + r  ->  word
    ((((nt*KT + kt)*8 + np)*32 + n_lane*4 + k_lane)*4 + n_pack*2 + k_pack), nibble r."""
    N, K = q.shape
    assert N % 128 == 0 and K % 128 == 0, "packer.py:207-212"
    v = (q.astype(np.uint32) & 0xF).reshape(N // 128, 8, 2, 8, K // 64, 2, 4, 8)  # nt np n_pack n_lane kt k_pack k_lane r
    v = v.transpose(0, 4, 1, 3, 6, 2, 5, 7)                                        # nt kt np n_lane k_lane n_pack k_pack r
    words = (v << (4 * np.arange(8, dtype=np.uint32))).sum(axis=-1, dtype=np.uint32)
    return np.ascontiguousarray(words).reshape(-1).view(np.int8).reshape(N, K // 2)


def unpack_qweight_ref(packed: np.ndarray) -> np.ndarray:
    """reference packed int8 [N, K/2] -> int8 values [N, K] (two's-complement s4)."""
    N, Kh = packed.shape
    K = Kh * 2
    words = np.ascontiguousarray(packed).view(np.uint32).ravel()
    word, nib = _qweight_index(N, K)
    v = (words[word] >> (4 * nib).astype(np.uint32)) & 0xF
    v = v.astype(np.int8)
    v[v >= 8] -= 16
    return v


def _scale_perm128():
    """Position inside one 128-channel block for the packed 16-bit scale vectors.

    packer.py:241-301 (pack_scale): reshape(n/128, 1, 8, 2, 4, 2, G), permute(0,6,1,2,4,3,5);
    consumed by load_wscale/broadcast_wscale, gemm_base.cuh:97-132,315-340.
    Returns src[n_pos] = logical channel stored at packed position n_pos.
    """
    pos = np.arange(128)
    lane, e = pos // 4, pos % 4
    return (lane // 4) * 16 + (e // 2) * 8 + (lane % 4) * 2 + e % 2


def pack_wscales_ref(ws: np.ndarray) -> np.ndarray:
    """logical [G, N] -> packed storage [G, N] (flat order ((nt*G+g)*128 + pos))."""
    G, N = ws.shape
    src = _scale_perm128()
    blk = ws.reshape(G, N // 128, 128)[:, :, src]  # [G, nt, pos]
    return np.ascontiguousarray(blk.transpose(1, 0, 2)).reshape(G, N)


def unpack_wscales_ref(packed: np.ndarray) -> np.ndarray:
    G, N = packed.shape
    src = _scale_perm128()
    blk = packed.reshape(N // 128, G, 128)  # [nt, g, pos]
    out = np.empty((G, N // 128, 128), dtype=packed.dtype)
    out[:, :, src] = blk.transpose(1, 0, 2)
    return out.reshape(G, N)


def pack_vec_ref(v: np.ndarray) -> np.ndarray:
    """bias / smooth_factor [N]: same intra-128 permutation with G = 1 (gemm_base.cuh:713)."""
    return pack_wscales_ref(v.reshape(1, -1)).reshape(-1)


def unpack_vec_ref(v: np.ndarray) -> np.ndarray:
    return unpack_wscales_ref(v.reshape(1, -1)).reshape(-1)


def _lowrank_index(C: int, R: int):
    """(c, r) -> flat position of the packed low-rank weights.

    packer.py:362-398 (pack_lowrank_weight), lora.cuh:43-59 (load_lora_wgt): 16x16 tiles in
    mma m16n8k16 fragment order:  flat = ((cp*RP + rp)*32 + lane)*8 + h,
    lane = nl*4 + kl, h = (nps*2 + kps)*2 + rk.
    ``c`` is the 16-tiled "n" axis, ``r`` the 16-tiled "k" axis of the fragment.
    """
    c = np.arange(C)[:, None]
    r = np.arange(R)[None, :]
    cp, c16 = c // 16, c % 16
    nps, nl = c16 // 8, c16 % 8
    rp, r16 = r // 16, r % 16
    kps, r8 = r16 // 8, r16 % 8
    kl, rk = r8 // 2, r8 % 2
    RP = R // 16
    lane = nl * 4 + kl
    h = (nps * 2 + kps) * 2 + rk
    return np.broadcast_to(((cp * RP + rp) * 32 + lane) * 8 + h, (C, R))


def pack_lowrank_ref(w: np.ndarray, down: bool) -> np.ndarray:
    """Logical low-rank weight -> packed storage.

    up   (down=False): logical [N, R] (out-channel, rank) -> stored shape [N, R].
    down (down=True):  logical [R, K] (rank, in-channel)  -> stored shape [K, R]
    (packer.py:378-386: for ``down`` the fragment "n" axis is the rank and the tile
    order is k-major).
    """
    if not down:
        N, R = w.shape
        idx = _lowrank_index(N, R)
        out = np.empty(N * R, dtype=w.dtype)
        out[idx.ravel()] = w.ravel()
        return out.reshape(N, R)
    R, K = w.shape
    # fragment n-axis = r, k-axis = k; tiles ordered (k_pack, r_pack)
    r = np.arange(R)[:, None]
    k = np.arange(K)[None, :]
    rp, r16 = r // 16, r % 16
    nps, nl = r16 // 8, r16 % 8
    kp, k16 = k // 16, k % 16
    kps, k8 = k16 // 8, k16 % 8
    kl, rk = k8 // 2, k8 % 2
    RP = R // 16
    lane = nl * 4 + kl
    h = (nps * 2 + kps) * 2 + rk
    idx = np.broadcast_to(((kp * RP + rp) * 32 + lane) * 8 + h, (R, K))
    out = np.empty(R * K, dtype=w.dtype)
    out[idx.ravel()] = w.ravel()
    return out.reshape(K, R)


def unpack_lowrank_ref(p: np.ndarray, down: bool) -> np.ndarray:
    """Inverse of :func:`pack_lowrank_ref` (packer.py:400-437)."""
    if not down:
        N, R = p.shape
        idx = _lowrank_index(N, R)
        return p.ravel()[idx]
    K, R = p.shape
    r = np.arange(R)[:, None]
    k = np.arange(K)[None, :]
    rp, r16 = r // 16, r % 16
    nps, nl = r16 // 8, r16 % 8
    kp, k16 = k // 16, k % 16
    kps, k8 = k16 // 8, k16 % 8
    kl, rk = k8 // 2, k8 % 2
    RP = R // 16
    lane = nl * 4 + kl
    h = (nps * 2 + kps) * 2 + rk
    idx = np.broadcast_to(((kp * RP + rp) * 32 + lane) * 8 + h, (R, K))
    return p.ravel()[idx]


def pack_rotemb_ref(rot: np.ndarray) -> np.ndarray:
    """[M, 64, 2] (sin, cos) float32 -> packed [M, 128] (models/embeddings.py:100-138;
    consumed by load_rotemb, epilogues.cuh:281-305)."""
    M = rot.shape[0]
    D = rot.shape[1] * 2
    x = rot.reshape(M // 16, 16, D // 8, 8).transpose(0, 2, 1, 3)
    x = x.reshape(M // 16, D // 8, 2, 8, 4, 2).transpose(0, 1, 3, 4, 2, 5)
    return np.ascontiguousarray(x).reshape(M, D)


def unpack_rotemb_ref(packed: np.ndarray) -> np.ndarray:
    M, D = packed.shape
    x = packed.reshape(M // 16, D // 8, 8, 4, 2, 2).transpose(0, 1, 4, 2, 3, 5)
    x = x.reshape(M // 16, D // 8, 16, 8).transpose(0, 2, 1, 3)
    return np.ascontiguousarray(x).reshape(M, D // 2, 2)


# --------------------------------------------------------------------------
# Activation quantiser (+ fused low-rank down projection)
# --------------------------------------------------------------------------
def quantize_rows(xh: np.ndarray, dtype: str, unsigned: bool):
    """Per (row, 64-channel group) 4-bit quantisation of an already smoothed 16-bit matrix.

    gemm_w4a4.cuh:429-523 (quantize_w4a4_from_fpsum_warp):
      amax  = max |x| over the group, in the 16-bit type            (:453-467)
      scale = float(amax) * (1/7)   [or 1/15 unsigned], fp32        (:469-473)
      stored ascale = half_t(scale)                                 (:474-477)
      q = sat_s4|u4( rni( float(x) * rcp(scale) ) ), the UNROUNDED fp32 scale (:479-495;
          cvt.rni + cvt.pack.sat, gemm_utils.cuh:209-229)
    rcp.approx.ftz is replaced by the IEEE reciprocal 1.0f/scale.  A zero group gives
    scale 0 -> rcp = inf -> 0*inf = NaN -> cvt.rni(NaN) = 0, i.e. code 0.

    Returns (codes int8 [M, K], ascales float32-valued-16-bit [K/64, M]).
    """
    M, K = xh.shape
    G = K // GROUP
    xg = xh.reshape(M, G, GROUP).astype(F32)
    amax = np.abs(xg).max(axis=2)  # exact in the 16-bit type
    recip_q = F32(1.0) / F32(15.0 if unsigned else 7.0)
    scale = (amax.astype(F32) * recip_q).astype(F32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        rscale = (F32(1.0) / scale).astype(F32)
        v = (xg * rscale[:, :, None]).astype(F32)
        q = np.rint(v)
    q = np.where(np.isnan(q), 0.0, q)
    lo, hi = (0, 15) if unsigned else (-8, 7)
    q = np.clip(q, lo, hi).astype(np.int8).reshape(M, K)
    ascales = round16(scale, dtype).T.copy()  # [G, M]
    return q, ascales


# --------------------------------------------------------------------------
# Approximation envelope: where an implementation built on approximate instructions may land
# --------------------------------------------------------------------------
# Documented bounds of the PTX instructions the reference uses on this path (PTX ISA, "Floating Point
# Instructions"; CUDA C Programming Guide, "Intrinsic Functions": __fdividef(x, y) has a maximum error of
# 2 ulp for 2^-126 <= y <= 2^126).  Relative to the infinitely precise result.
PTX_APPROX = {
    "rcp.approx.ftz.f32": {"max_ulp": 1, "used": "cuda_frcp: 1 / scale in the quantiser (gemm_utils.cuh:260-264, gemm_w4a4.cuh:479-495)"},
    "div.approx (__fdividef)": {"max_ulp": 2, "used": "h2div: x / smooth (gemm_utils.cuh:329-344, gemm_w4a4.cuh:990-993)"},
    "ex2.approx.ftz.f32": {"max_ulp": 2, "used": "cuda_sigmoidf -> silu (gemm_utils.cuh:290-303)"},
    "rsqrt.approx.ftz.f32": {"max_ulp": 2, "used": "RMSNorm coefficient (epilogues.cuh:343-360)"},
    "tanh.approx.f32": {"max_rel": 2.0 ** -11, "used": "gelu_half2 (gemm_utils.cuh:305-312)"},
}
_ORACLE_SLACK_ULP = 1  # the IEEE value the interval is centred on is itself a rounding of the exact result


def _widen32(x: np.ndarray, ulps: int):
    """[x - ulps, x + ulps] in units of float32 steps (nextafter), elementwise; x float32"""
    lo = np.asarray(x, dtype=F32).copy()
    hi = lo.copy()
    for _ in range(ulps + _ORACLE_SLACK_ULP):
        lo = np.nextafter(lo, F32(-np.inf))
        hi = np.nextafter(hi, F32(np.inf))
    return lo, hi


def quantize_rows_envelope(xlo: np.ndarray, xhi: np.ndarray, dtype: str, unsigned: bool):
    """:func:`quantize_rows` on an interval input [xlo, xhi] (16-bit representable, xlo <= xhi elementwise) with the
    reciprocal of the scale an APPROXIMATE instruction (rcp.approx: 1 ulp).  Every other step is exactly specified and
    monotone, so it maps interval ends to interval ends:
      amax in [max |x|_lo, max |x|_hi];  scale = fp32(amax * (1/7 | 1/15));  stored scale = round16(scale);
      rcp in widen(1 / scale, 1 ulp);  v = fp32(x * rcp);  q = sat(rint(v)).
    Returns dict(q_lo, q_hi int8 [M, K], s_lo, s_hi [K/64, M])."""
    M, K = xlo.shape
    G = K // GROUP
    lo = xlo.reshape(M, G, GROUP).astype(F32)
    hi = xhi.reshape(M, G, GROUP).astype(F32)
    straddle = (lo <= 0) & (hi >= 0)
    abs_lo = np.where(straddle, F32(0), np.minimum(np.abs(lo), np.abs(hi)))
    abs_hi = np.maximum(np.abs(lo), np.abs(hi))
    recip_q = F32(1.0) / F32(15.0 if unsigned else 7.0)
    sc_lo = (abs_lo.max(axis=2).astype(F32) * recip_q).astype(F32)
    sc_hi = (abs_hi.max(axis=2).astype(F32) * recip_q).astype(F32)
    qmin, qmax = (0, 15) if unsigned else (-8, 7)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        r_hi = _widen32((F32(1.0) / sc_lo).astype(F32), PTX_APPROX["rcp.approx.ftz.f32"]["max_ulp"])[1]  # largest reciprocal: of the smallest scale
        r_lo = _widen32((F32(1.0) / sc_hi).astype(F32), PTX_APPROX["rcp.approx.ftz.f32"]["max_ulp"])[0]
        cands = [(a * r[:, :, None]).astype(F32) for a in (lo, hi) for r in (r_lo, r_hi)]
        v_lo = np.minimum.reduce(cands)
        v_hi = np.maximum.reduce(cands)
        q_lo = np.clip(np.where(np.isnan(v_lo), 0.0, np.rint(v_lo)), qmin, qmax)
        q_hi = np.clip(np.where(np.isnan(v_hi), 0.0, np.rint(v_hi)), qmin, qmax)
    zero_hi = sc_hi == 0          # the whole group is exactly zero: scale 0 -> code 0 (quantize_rows)
    zero_lo = (sc_lo == 0) & ~zero_hi  # the group MAY be all zero: nothing can be said about its codes beyond the saturation range
    q_lo = np.where(zero_hi[:, :, None], 0, np.where(zero_lo[:, :, None], qmin, q_lo))
    q_hi = np.where(zero_hi[:, :, None], 0, np.where(zero_lo[:, :, None], qmax, q_hi))
    return {"q_lo": q_lo.astype(np.int8).reshape(M, K), "q_hi": q_hi.astype(np.int8).reshape(M, K),
            "s_lo": round16(sc_lo, dtype).T.copy(), "s_hi": round16(sc_hi, dtype).T.copy()}


def quantize_envelope(x: np.ndarray, smooth: np.ndarray | None, dtype: str = "bf16", pad_size: int = PAD_M):
    """Envelope of :func:`quantize_w4a4_act_fuse_lora`'s codes and scales: the smoothing division is ``__fdividef``
    (2 ulp of the exact quotient, then the exact rounding to 16 bits), the scale reciprocal ``rcp.approx`` (1 ulp)."""
    M, K = x.shape
    M_pad = ceil_div(M, pad_size) * pad_size
    xp = np.zeros((M_pad, K), dtype=F32)
    xp[:M] = x
    if smooth is not None:
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (xp / smooth.astype(F32)[None, :]).astype(F32)
        lo32, hi32 = _widen32(t, PTX_APPROX["div.approx (__fdividef)"]["max_ulp"])
        xlo, xhi = round16(lo32, dtype), round16(hi32, dtype)
    else:
        xlo = xhi = xp
    return quantize_rows_envelope(xlo, xhi, dtype, unsigned=False)


def gelu_tanh_envelope(x: np.ndarray):
    """[lo, hi] of gelu_half2 (gemm_utils.cuh:305-312) in fp32 when tanh is ``tanh.approx.f32`` (relative error <= 2^-11) and the
    surrounding fp32 operations may or may not be contracted into FMAs (nvcc contracts by default: +-2 float32 steps on the argument and
    +-4 on the product cover either choice).  The MI355X form x * rcp(1 + exp2(-2 u log2 e)) lies inside: hardware exp2 and rcp are 1 ulp
    each, the whole expression stays within ~3 float32 steps of the exact value -- three orders of magnitude inside tanh.approx's 2^-11."""
    x64 = x.astype(np.float64)
    u = 0.79788456 * (x64 + 0.044715 * x64 ** 3)
    ulo, uhi = _widen32(u.astype(F32), 2)
    th_lo, th_hi = np.tanh(ulo.astype(np.float64)), np.tanh(uhi.astype(np.float64))
    rel = PTX_APPROX["tanh.approx.f32"]["max_rel"]
    th_lo = th_lo - rel * np.abs(th_lo)
    th_hi = th_hi + rel * np.abs(th_hi)
    a, b = x64 * (0.5 + 0.5 * th_lo), x64 * (0.5 + 0.5 * th_hi)
    lo, _ = _widen32(np.minimum(a, b).astype(F32), 4)
    _, hi = _widen32(np.maximum(a, b).astype(F32), 4)
    return lo, hi


def silu_envelope(x: np.ndarray):
    """[lo, hi] of cuda_sigmoidf-based silu (gemm_utils.cuh:290-303,323-327) in fp32: t = -log2(e) x (exact fp32 multiply), e = ex2.approx(t) (2 ulp),
    d = e + 1 (exact fp32 add), r = rcp.approx(d) (1 ulp), x * r.  The intervals are propagated through the monotone steps."""
    x = x.astype(F32)
    t = (F32(-1.442695041) * x).astype(F32)
    with np.errstate(over="ignore"):
        e = np.exp2(t.astype(np.float64)).astype(F32)
    e_lo, e_hi = _widen32(e, PTX_APPROX["ex2.approx.ftz.f32"]["max_ulp"])
    d_lo, d_hi = (e_lo + F32(1)).astype(F32), (e_hi + F32(1)).astype(F32)
    with np.errstate(divide="ignore"):
        r_hi = _widen32((F32(1) / d_lo).astype(F32), PTX_APPROX["rcp.approx.ftz.f32"]["max_ulp"])[1]
        r_lo = _widen32((F32(1) / d_hi).astype(F32), PTX_APPROX["rcp.approx.ftz.f32"]["max_ulp"])[0]
    a, b = (x * r_lo).astype(F32), (x * r_hi).astype(F32)
    return np.minimum(a, b), np.maximum(a, b)


def rmsnorm_coef_envelope(sq: np.ndarray):
    """[lo, hi] of the RMSNorm coefficient rsqrt.approx(mean(y^2) + 1e-6) (epilogues.cuh:343-360): 2 ulp on the reciprocal square root; the mean's own fp32
    summation order adds a few steps (the reference sums 128 squares across a warp in an unspecified tree: +-4 float32 steps here)."""
    m = (sq.astype(F32) / F32(128.0) + F32(RMS_EPS)).astype(F32)
    m_lo, m_hi = _widen32(m, 4)
    c_hi = _widen32((F32(1.0) / np.sqrt(m_lo.astype(np.float64))).astype(F32), PTX_APPROX["rsqrt.approx.ftz.f32"]["max_ulp"])[1]
    c_lo = _widen32((F32(1.0) / np.sqrt(m_hi.astype(np.float64))).astype(F32), PTX_APPROX["rsqrt.approx.ftz.f32"]["max_ulp"])[0]
    return c_lo, c_hi


def rmsnorm_rope_envelope(y16: np.ndarray, norm_q, norm_k, rot: np.ndarray, dtype: str):
    """[lo, hi] (16-bit) of :func:`rmsnorm_rope`'s Q / K outputs when the coefficient comes from rsqrt.approx (:func:`rmsnorm_coef_envelope`) and the
    rotation's fp32 products may or may not be contracted into FMAs (+- 2^-22 of the products' magnitude on each rotated value: it may be a cancelling difference).  V columns: lo == hi == the input."""
    M, N = y16.shape
    H3 = N // 128
    y = y16.astype(F32).reshape(M, H3, 128)
    lo, hi = y.copy(), y.copy()
    sin = rot[:, :, 0].astype(F32)[:, None, :]
    cos = rot[:, :, 1].astype(F32)[:, None, :]
    for part, w in ((0, norm_q), (1, norm_k)):
        sl = slice(part * H3 // 3, (part + 1) * H3 // 3)
        blk = y[:, sl, :]
        sq = (blk.astype(np.float64) ** 2).sum(axis=2).astype(F32)
        c_lo, c_hi = rmsnorm_coef_envelope(sq)
        outs, mags = [], []
        for c in (c_lo, c_hi):
            b = (blk * (c[:, :, None] * w.astype(F32)[None, None, :]).astype(F32)).astype(F32)
            xe, xo = b[:, :, 0::2], b[:, :, 1::2]
            o, g = np.empty_like(b), np.empty_like(b)
            o[:, :, 0::2] = (xe * cos - xo * sin).astype(F32)
            o[:, :, 1::2] = (xe * sin + xo * cos).astype(F32)
            g[:, :, 0::2] = np.abs(xe * cos) + np.abs(xo * sin)   # magnitude of the two products: what the rounding errors of a (possibly
            g[:, :, 1::2] = np.abs(xe * sin) + np.abs(xo * cos)   # cancelling) difference scale with, whether or not it is contracted into an FMA
            outs.append(o)
            mags.append(g)
        # the rotated values are linear in the coefficient: between its two ends every output moves monotonically; +- 2^-22 of the product magnitude
        # covers the fp32 roundings of either evaluation order
        slack = (np.maximum(mags[0], mags[1]) * F32(2.0 ** -22)).astype(F32)
        lo[:, sl, :], hi[:, sl, :] = np.minimum(outs[0], outs[1]) - slack, np.maximum(outs[0], outs[1]) + slack
    return round16(lo.reshape(M, N), dtype), round16(hi.reshape(M, N), dtype)


def envelope_report(codes: np.ndarray, env: dict, ieee: np.ndarray) -> dict:
    """fraction of codes outside [q_lo, q_hi], flip rate against the IEEE codes, and how wide the envelope itself is"""
    c = codes.astype(np.int32)
    return {"outside": float(((c < env["q_lo"]) | (c > env["q_hi"])).mean()), "flips_vs_ieee": float((c != ieee).mean()),
            "max_abs_diff_vs_ieee": int(np.abs(c - ieee.astype(np.int32)).max()), "envelope_open": float((env["q_lo"] != env["q_hi"]).mean())}


def lora_down_project(x16: np.ndarray, lora_down: np.ndarray) -> np.ndarray:
    """lora_act[M, R] = x @ lora_down, 16-bit operands, fp32 accumulate
    (lora.cuh:253-339 EpilogueLoraDown; fp32 atomics across CTAs :82-94).  The oracle sums in
    float64 and rounds once to fp32 (the reference's summation order is not defined)."""
    return (x16.astype(np.float64) @ lora_down.astype(np.float64)).astype(F32)


def glu_pairs(x2: np.ndarray, dtype: str) -> np.ndarray:
    """load_act_to_fpsum<fuse_glu = true> (gemm_base.cuh:606-633): the row holds (value, gate) pairs;
    out[k] = value[k] * silu(gate[k]) with silu(T) = (T)(float(g) * sigmoid(float(g))) rounded to 16 bits
    (gemm_utils.cuh:323-327) and the product a 16-bit multiply (one more rounding).  [M, 2K] -> [M, K]."""
    v, g = x2[:, 0::2].astype(F32), x2[:, 1::2]
    sg = round16(silu(g), dtype)
    return round16((v.astype(np.float64) * sg.astype(np.float64)).astype(F32), dtype)


def quantize_w4a4_act_fuse_lora(
    x: np.ndarray,
    smooth: np.ndarray | None,
    lora_down: np.ndarray | None,
    dtype: str = "bf16",
    pad_size: int = PAD_M,
    fuse_glu: bool = False,
):
    """kernels::quantize_w4a4_act_fuse_lora  (gemm_w4a4.cuh:1097-1184; launch
    gemm_w4a4_launch_impl.cuh:451-521; caller Linear.cpp:444-502, ops/quantize.py:11-81).

    x         : [M, K] 16-bit input (float32 carrier); with ``fuse_glu`` [M, 2K] interleaved (value, gate) pairs
    smooth    : [K] 16-bit smoothing factor (logical order) or None
    lora_down : [K, R] 16-bit LOGICAL down projection (x @ lora_down) or None

    Steps: rows padded with zeros to M_pad (load_act_to_fpsum zero-fill, gemm_base.cuh:591-646);
    lora_act = x @ lora_down on the UN-smoothed input (gemm_w4a4.cuh:1159-1171);
    x_hat = round16(x / smooth) with an fp32 divide (h2div, gemm_utils.cuh:329-344; the
    reference uses __fdividef, the oracle IEEE division), shift = 0 (:1181);
    then :func:`quantize_rows` (signed).
    Returns (codes int8 [M_pad, K], ascales [K/64, M_pad], lora_act f32 [M_pad, R]).
    """
    if fuse_glu:
        x = glu_pairs(x, dtype)
    M, K = x.shape
    M_pad = ceil_div(M, pad_size) * pad_size
    xp = np.zeros((M_pad, K), dtype=F32)
    xp[:M] = x
    lora_act = None
    if lora_down is not None:
        lora_act = lora_down_project(xp, lora_down)
    if smooth is not None:
        with np.errstate(divide="ignore", invalid="ignore"):
            xh = round16((xp / smooth.astype(F32)[None, :]).astype(F32), dtype)
    else:
        xh = xp
    q, ascales = quantize_rows(xh, dtype, unsigned=False)
    return q, ascales, lora_act


# --------------------------------------------------------------------------
# GEMM + epilogues
# --------------------------------------------------------------------------
def int_group_dot(qa: np.ndarray, qw: np.ndarray) -> np.ndarray:
    """psum[g, m, n] = sum_{k in group g} qa[m, k] * qw[n, k]  (int32)
    gemm_w4a4.cuh:408-426 (mma m16n8k64 s4|u4 x s4 -> s32), :721-735."""
    M, K = qa.shape
    N = qw.shape[0]
    G = K // GROUP
    a = qa.reshape(M, G, GROUP).astype(np.float32)
    w = qw.reshape(N, G, GROUP).astype(np.float32)
    # products and every partial sum are integers of magnitude <= 64 * 8 * 15 = 7680 < 2^24: a float32 matmul is exact
    # in any summation order and uses BLAS (half the memory traffic of float64 at the FLUX shapes)
    return np.einsum("mgk,ngk->gmn", a, w, optimize=True).astype(np.int32)


def gelu_tanh(x: np.ndarray) -> np.ndarray:
    """gemm_utils.cuh:305-312 (gelu_half2): x*(0.5 + 0.5*tanh(0.79788456*(x + 0.044715*x^3)))
    in fp32; tanh.approx replaced by exact tanh."""
    x = x.astype(F32)
    x3 = (x * x * x).astype(F32)
    inner = (F32(0.79788456) * (x + F32(0.044715) * x3)).astype(F32)
    t = (F32(0.5) + F32(0.5) * np.tanh(inner).astype(F32)).astype(F32)
    return (x * t).astype(F32)


def silu(x: np.ndarray) -> np.ndarray:
    """gemm_base.cuh:783-792, gemm_utils.cuh:280-293,314-319: x*sigmoid(x) in fp32
    (ex2.approx/rcp.approx replaced by exact math)."""
    x = x.astype(np.float64)
    return (x / (1.0 + np.exp(-x))).astype(F32)


def rmsnorm_rope(y16: np.ndarray, norm_q, norm_k, rot: np.ndarray, dtype: str) -> np.ndarray:
    """Epilogues::EpilogueRMSNormRope (epilogues.cuh:269-425).

    y16    : [M, N] 16-bit QKV projection, N = 3 * heads * 128; the first third is Q, the
             second K, the last V (untouched) (:414-423).
    norm_* : [128] 16-bit RMSNorm weights.   rot : [M, 64, 2] fp32 (sin, cos).
    Per row and head: coef = rsqrt(mean(y^2) + 1e-6) (rsqrt.approx -> exact), y *= coef*w[c]
    (fp32), adjacent pairs rotated (x, y) -> (x cos - y sin, x sin + y cos) (:343-405), then
    back to 16-bit.
    """
    M, N = y16.shape
    H3 = N // 128
    assert H3 % 3 == 0
    y = y16.astype(F32).reshape(M, H3, 128).copy()
    sin = rot[:, :, 0].astype(F32)[:, None, :]
    cos = rot[:, :, 1].astype(F32)[:, None, :]
    for part, w in ((0, norm_q), (1, norm_k)):
        sl = slice(part * H3 // 3, (part + 1) * H3 // 3)
        blk = y[:, sl, :]
        sq = (blk.astype(np.float64) ** 2).sum(axis=2).astype(F32)
        coef = (F32(1.0) / np.sqrt((sq / F32(128.0) + F32(RMS_EPS)).astype(F32))).astype(F32)
        blk = (blk * (coef[:, :, None] * w.astype(F32)[None, None, :]).astype(F32)).astype(F32)
        xe, xo = blk[:, :, 0::2], blk[:, :, 1::2]
        re = (xe * cos - xo * sin).astype(F32)
        ro = (xe * sin + xo * cos).astype(F32)
        out = np.empty_like(blk)
        out[:, :, 0::2], out[:, :, 1::2] = re, ro
        y[:, sl, :] = out
    return round16(y.reshape(M, N), dtype)


def gemm_w4a4(
    qa: np.ndarray,
    ascales: np.ndarray,
    qw: np.ndarray,
    wscales: np.ndarray,
    *,
    dtype: str = "bf16",
    bias: np.ndarray | None = None,
    lora_act_in: np.ndarray | None = None,
    lora_up: np.ndarray | None = None,
    lora_scales=None,
    fuse: str = "none",  # none | silu | gelu_quant | rmsnorm_rope
    next_smooth: np.ndarray | None = None,
    next_lora_down: np.ndarray | None = None,
    norm_q=None,
    norm_k=None,
    rot=None,
    accum: str = "fp32",  # "fp32": exact accumulation ; "ref16": the reference's 16-bit chain
    envelope: bool = False,  # gelu_quant: also return the approximation envelope of codes and scales ("envelope": quantize_rows_envelope's dict)
):
    """kernels::gemm_w4a4 (gemm_w4a4_launch_impl.cuh:7-424) with the epilogue chain
    Bias -> LoraUp -> Mid(Gelu|Silu|Nop) -> [LoraDown] -> Next(Default|Quantize|RMSNormRope)
    (:172-280, 282-423).

    qa [M, K] int8 codes (signed, or unsigned when the producer was the GELU epilogue),
    ascales [K/64, M], qw [N, K] int8, wscales [K/64, N], bias [N], lora_act_in f32 [M, R],
    lora_up [N, R] logical.  Everything 16-bit is a float32 carrier.

    accum="fp32": lin = sum_g as*ws*psum + bias + lora, exact (float64) then ONE rounding to
                  16-bit -- what the MI355X kernel implements (fp32 accumulators).
    accum="ref16": emulates gemm_base.cuh:367-409 -- fsum = fma16(cvt16(psum), mul16(as, ws),
                  fsum) sequentially over groups in the 16-bit type (USE_FP32_ACCUM=false,
                  gemm_w4a4.cuh:1080), bias add in 16-bit (gemm_base.cuh:717-767), low-rank
                  in fp32 then back to 16-bit (lora.cuh:145-158,221).

    Returns a dict with "out" (16-bit, [M, N]) or, for fuse="gelu_quant",
    "qout" (uint4 codes int8 [M, N]), "oscales" [N/64, M], "lora_act_out" f32 [M, R'].
    """
    M, K = qa.shape
    N = qw.shape[0]
    G = K // GROUP
    psum = int_group_dot(qa, qw)  # [G, M, N]
    as_ = ascales.astype(F32)  # [G, M]
    ws_ = wscales.astype(F32)  # [G, N]

    if accum == "fp32":
        lin = np.einsum("gmn,gm,gn->mn", psum.astype(np.float64), as_.astype(np.float64), ws_.astype(np.float64))
        if bias is not None:
            lin = lin + bias.astype(np.float64)[None, :]
        if lora_up is not None:
            R = lora_up.shape[1]
            sc = np.ones(ceil_div(R, 16), dtype=F32) if lora_scales is None else np.asarray(lora_scales, dtype=F32)
            la = (lora_act_in.astype(F32) * np.repeat(sc, 16)[None, :R]).astype(F32)
            la16 = round16(la, dtype)  # lora.cuh:145-158: fp32 act * scale -> 16-bit
            lin = lin + la16.astype(np.float64) @ lora_up.astype(np.float64).T
        y16 = round16(lin.astype(F32), dtype)
    elif accum == "ref16":
        fsum = np.zeros((M, N), dtype=F32)
        for g in range(G):
            p16 = round16(psum[g].astype(F32), dtype)  # int2half2
            s16 = round16((as_[g][:, None] * ws_[g][None, :]).astype(F32), dtype)  # __hmul2
            fsum = round16((p16.astype(np.float64) * s16.astype(np.float64) + fsum.astype(np.float64)).astype(F32), dtype)
        if bias is not None:
            fsum = round16((fsum + bias.astype(F32)[None, :]).astype(F32), dtype)
        if lora_up is not None:
            R = lora_up.shape[1]
            sc = np.ones(ceil_div(R, 16), dtype=F32) if lora_scales is None else np.asarray(lora_scales, dtype=F32)
            la16 = round16((lora_act_in.astype(F32) * np.repeat(sc, 16)[None, :R]).astype(F32), dtype)
            f32 = fsum.astype(np.float64) + la16.astype(np.float64) @ lora_up.astype(np.float64).T
            fsum = round16(f32.astype(F32), dtype)
        y16 = fsum
    else:
        raise ValueError(accum)

    if fuse == "none":
        out = y16
        if dtype == "fp16":  # EpilogueDefault clamp, gemm_base.cuh:689-695
            out = np.clip(out, -65504.0, 65504.0)
        return {"out": out}
    if fuse == "silu":
        return {"out": round16(silu(y16), dtype)}
    if fuse == "rmsnorm_rope":
        return {"out": rmsnorm_rope(y16, norm_q, norm_k, rot, dtype)}
    if fuse == "gelu_quant":
        # EpilogueGelu (epilogues.cuh:22-44) -> 16-bit
        g16 = round16(gelu_tanh(y16), dtype)
        res = {}
        if next_lora_down is not None:
            # EpilogueLoraDown on the GELU output, before shift/smooth (launch_impl.cuh:226-262)
            res["lora_act_out"] = lora_down_project(g16, next_lora_down)
        # EpilogueQuantize<false, unsigned=true> (gemm_w4a4.cuh:945-1021): 16-bit add of the
        # shift (:986), fp32 divide by smooth -> 16-bit (:990-993), then the group quantiser.
        sh = round16((g16 + F32(GELU_SHIFT)).astype(F32), dtype)
        with np.errstate(divide="ignore", invalid="ignore"):
            xh = round16((sh / next_smooth.astype(F32)[None, :]).astype(F32), dtype)
        q, osc = quantize_rows(xh, dtype, unsigned=True)
        res["qout"], res["oscales"] = q, osc
        if envelope:
            # the same chain on intervals: tanh.approx in the GELU, __fdividef in the smoothing division, rcp.approx in the quantiser
            glo, ghi = gelu_tanh_envelope(y16)
            glo, ghi = round16(glo, dtype), round16(ghi, dtype)
            slo, shi = round16((glo + F32(GELU_SHIFT)).astype(F32), dtype), round16((ghi + F32(GELU_SHIFT)).astype(F32), dtype)
            with np.errstate(divide="ignore", invalid="ignore"):
                ns = next_smooth.astype(F32)[None, :]
                cands = []
                for e in (slo, shi):
                    l32, h32 = _widen32((e / ns).astype(F32), PTX_APPROX["div.approx (__fdividef)"]["max_ulp"])
                    cands += [l32, h32]
            env = quantize_rows_envelope(round16(np.minimum.reduce(cands), dtype), round16(np.maximum.reduce(cands), dtype), dtype, unsigned=True)
            res["envelope"] = env
            res["g16_lo"], res["g16_hi"] = glo, ghi
        return res
    raise ValueError(fuse)


# --------------------------------------------------------------------------
# Whole layers (callers: nunchaku/models/linear.py:161-268, nunchaku/ops/fused.py:14-79)
# --------------------------------------------------------------------------
def svdq_linear(x, layer: dict, dtype="bf16", accum="fp32", fuse="none", **kw):
    """SVDQW4A4Linear.forward (linear.py:161-188): quantize -> gemm.  ``layer`` holds LOGICAL
    tensors: qweight int8 [N,K], wscales [G,N], bias [N]|None, smooth [K], proj_down [K,R],
    proj_up [N,R]."""
    M = x.shape[0]
    q, asc, la = quantize_w4a4_act_fuse_lora(x, layer["smooth"], layer["proj_down"], dtype)
    r = gemm_w4a4(
        q, asc, layer["qweight"], layer["wscales"], dtype=dtype, bias=layer.get("bias"),
        lora_act_in=la, lora_up=layer["proj_up"], accum=accum, fuse=fuse, **kw,
    )
    if "out" in r:
        r["out"] = r["out"][:M]
    return r


def fused_gelu_mlp(x, fc1: dict, fc2: dict, dtype="bf16", accum="fp32"):
    """ops/fused.py:14-79: fc1 GEMM with GELU + unsigned requant + fc2's low-rank down
    projection fused, then fc2 with act_unsigned=True."""
    M = x.shape[0]
    q, asc, la = quantize_w4a4_act_fuse_lora(x, fc1["smooth"], fc1["proj_down"], dtype)
    r = gemm_w4a4(
        q, asc, fc1["qweight"], fc1["wscales"], dtype=dtype, bias=fc1.get("bias"), lora_act_in=la,
        lora_up=fc1["proj_up"], accum=accum, fuse="gelu_quant", next_smooth=fc2["smooth"],
        next_lora_down=fc2["proj_down"],
    )
    o = gemm_w4a4(
        r["qout"], r["oscales"], fc2["qweight"], fc2["wscales"], dtype=dtype, bias=fc2.get("bias"),
        lora_act_in=r["lora_act_out"], lora_up=fc2["proj_up"], accum=accum,
    )
    return o["out"][:M]


# --------------------------------------------------------------------------
# Synthetic SVDQuant layers (SURVEY.md §8d "synthetic inputs")
# --------------------------------------------------------------------------
def make_svdq_layer(K: int, N: int, R: int = 32, seed: int = 0, dtype: str = "bf16", bias: bool = True,
                    cheap: bool = False, svd: str = "full", lowrank_energy: float = 0.0) -> dict:
    """W ~ N(0, 0.02^2) [N, K]; smooth ~ exp(N(0, 0.5^2)); W_hat = W*diag(smooth); rank-R
    truncated SVD -> L1 [K, R], L2 [R, N]; residual -> symmetric s4, group 64, scale = amax/7.
    proj_down = L1 / smooth (it is applied to the raw x), proj_up = L2^T.  ``cheap`` replaces
    the SVD by a random rank-R factor pair (for big shapes in benches)."""
    rng = np.random.default_rng(seed)
    W = (rng.standard_normal((N, K)) * 0.02).astype(F32)
    if lowrank_energy > 0.0:
        # a weight whose energy sits mostly in a rank-R component (fraction `lowrank_energy` of the Frobenius norm^2), the
        # situation SVDQuant is designed for: the 16-bit low-rank branch carries the bulk, the 4-bit branch a small
        # residual -- a forward pass is then well conditioned with respect to +-1 code flips
        A = rng.standard_normal((N, R)).astype(F32)
        B = rng.standard_normal((R, K)).astype(F32)
        low0 = (A @ B) * F32(0.02 / np.sqrt(R))
        W = (np.sqrt(lowrank_energy) * low0 + np.sqrt(1.0 - lowrank_energy) * W).astype(F32)
    smooth = round16(np.exp(rng.standard_normal(K) * 0.5).astype(F32), dtype)
    What = W * smooth[None, :]
    if cheap:
        L2t = (rng.standard_normal((N, R)) * 0.02).astype(F32)
        L1 = (rng.standard_normal((K, R)) * 0.05).astype(F32)
        low = L2t @ L1.T
    elif svd == "randomized":
        # rank-R range finder with two power iterations (Halko et al.): the top-R subspace of a 12288 x 3072 matrix in a
        # few seconds instead of a full SVD; only used to build test layers with checkpoint-like statistics
        A = What.astype(np.float64)
        Q = np.linalg.qr(A @ rng.standard_normal((K, R + 16)))[0]
        for _ in range(2):
            Q = np.linalg.qr(A @ (A.T @ Q))[0]
        Ub, S, Vt = np.linalg.svd(Q.T @ A, full_matrices=False)
        U = Q @ Ub
        L2t = (U[:, :R] * np.sqrt(S[:R])).astype(F32)
        L1 = (Vt[:R].T * np.sqrt(S[:R])).astype(F32)
        low = L2t @ L1.T
    else:
        U, S, Vt = np.linalg.svd(What.astype(np.float64), full_matrices=False)
        L2t = (U[:, :R] * np.sqrt(S[:R])).astype(F32)  # [N, R]
        L1 = (Vt[:R].T * np.sqrt(S[:R])).astype(F32)  # [K, R]
        low = L2t @ L1.T
    res = What - low
    G = K // GROUP
    rg = res.reshape(N, G, GROUP)
    amax = np.abs(rg).max(axis=2)
    ws = round16((amax / 7.0).astype(F32), dtype)  # [N, G]
    ws = np.where(ws == 0, F32(1.0), ws)
    qw = np.clip(np.rint(rg / ws[:, :, None]), -8, 7).astype(np.int8).reshape(N, K)
    layer = {
        "qweight": qw,
        "wscales": ws.T.copy(),  # [G, N]
        "smooth": smooth,
        "proj_down": round16((L1 / smooth[:, None]).astype(F32), dtype),  # [K, R]
        "proj_up": round16(L2t, dtype),  # [N, R]
        "bias": round16((rng.standard_normal(N) * 0.1).astype(F32), dtype) if bias else None,
        "dense": W,
    }
    return layer


def make_random_svdq_layer(K: int, N: int, R: int = 32, seed: int = 0, dtype: str = "bf16", bias: bool = True) -> dict:
    """A layer with the tensor statistics of :func:`make_svdq_layer` drawn directly (no dense weight, no SVD): codes =
    clip(rint(N(0, 2.6^2)), -8, 7), group scales ~ 0.02 * 2.7 / 7 * lognormal, smooth ~ exp(N(0, 0.5^2)), low-rank factors
    ~ N(0, 0.02^2) / N(0, 0.05^2).  For the full-size (3072 x 12288) GPU parity tests, where building a layer must take
    well under a second; the kernels' arithmetic does not depend on where the codes came from."""
    rng = np.random.default_rng(seed)
    qw = np.clip(np.rint(rng.standard_normal((N, K), dtype=F32) * F32(2.6)), -8, 7).astype(np.int8)
    G = K // GROUP
    ws = round16((0.0077 * np.exp(rng.standard_normal((G, N), dtype=F32) * F32(0.3))).astype(F32), dtype)
    smooth = round16(np.exp(rng.standard_normal(K).astype(F32) * F32(0.5)).astype(F32), dtype)
    L1 = (rng.standard_normal((K, R), dtype=F32) * F32(0.05)).astype(F32)
    return {
        "qweight": qw,
        "wscales": ws,
        "smooth": smooth,
        "proj_down": round16((L1 / smooth[:, None]).astype(F32), dtype),
        "proj_up": round16((rng.standard_normal((N, R), dtype=F32) * F32(0.02)).astype(F32), dtype),
        "bias": round16((rng.standard_normal(N).astype(F32) * F32(0.1)).astype(F32), dtype) if bias else None,
    }


def make_activations(M: int, K: int, seed: int = 0, dtype: str = "bf16", positive: bool = False) -> np.ndarray:
    """x ~ N(0,1) with 1 % outlier channels x20."""
    rng = np.random.default_rng(1000 + seed)
    x = rng.standard_normal((M, K)).astype(F32)
    out = rng.choice(K, size=max(1, K // 100), replace=False)
    x[:, out] *= 20.0
    if positive:
        x = np.abs(x)
    return round16(x, dtype)


# --------------------------------------------------------------------------
# AWQ W4A16 GEMV: the AdaLayerNormZero modulation projections.
# Layout: nunchaku/models/text_encoders/tinychat_utils.py:76-107 (pack_w4), consumed by
# src/kernels/awq/gemv_awq.cu:100-286; Python surface nunchaku/models/linear.py:277-414.
# Layout pinned against the reference's pack_w4 / convert_to_tinychat_w4x16y16_linear_weight
# (tests/golden/awq_*.npz, tools/make_golden.py); arithmetic restated from gemv_awq.cu (unpinned).
# --------------------------------------------------------------------------
AWQ_GROUP = 64  # gemv_awq.cu:278 (GROUP_SIZE; asserted against the group_size argument)


def _awq_index(N: int, K: int):
    """(int16 index, nibble) of logical weight (n, k) in the reference's packed AWQ tensor.

    pack_w4 (tinychat_utils.py:97-107): 32 consecutive input channels -> 8 int16, int16 j holds channels
    (j, 8+j, 16+j, 24+j) in nibbles 0..3; then rows are interleaved 4 at a time per 64-channel chunk:
    [N/4][K/64][4 rows][16 int16].  The kernel reads exactly this: gemv_awq.cu:161-176 (row / chunk / half
    of a thread), dequantize.cuh:17-77 + the shuffle at gemv_awq.cu:192-205 (nibble -> channel)."""
    n = np.arange(N)[:, None]
    k = np.arange(K)[None, :]
    rg, i = n // 4, n % 4
    c, half, kk = k // 64, (k % 64) // 32, k % 32
    e, j = kk // 8, kk % 8
    idx = (rg * (K // 64) + c) * 64 + i * 16 + half * 8 + j  # int16 units; a row group holds K int16
    return np.broadcast_to(idx, (N, K)), np.broadcast_to(e, (N, K))


def pack_awq_w4_ref(q: np.ndarray) -> np.ndarray:
    """unsigned 4-bit codes [N, K] -> the reference's qweight parameter, int32 [N/4, K/2] (linear.py:328-330)."""
    N, K = q.shape
    assert N % 4 == 0 and K % 64 == 0
    assert q.min() >= 0 and q.max() <= 15
    idx, nib = _awq_index(N, K)
    words = np.zeros(N * K // 4, dtype=np.uint16)
    np.bitwise_or.at(words, idx.ravel(), (q.astype(np.uint16) << (4 * nib).astype(np.uint16)).ravel())
    return words.view(np.int32).reshape(N // 4, K // 2)


def unpack_awq_w4_ref(packed: np.ndarray) -> np.ndarray:
    """int32 [N/4, K/2] -> codes [N, K] (uint8)."""
    N, K = packed.shape[0] * 4, packed.shape[1] * 2
    w = np.ascontiguousarray(packed).view(np.uint16).ravel()
    idx, nib = _awq_index(N, K)
    return ((w[idx] >> (4 * nib).astype(np.uint16)) & 0xF).astype(np.uint8)


def awq_quantize_ref(w: np.ndarray, dtype: str):
    """Asymmetric 4-bit group-64 quantisation as the reference converter does it
    (tinychat_utils.py:165-188): q = round((w + zero)/scale) in [0, 15]; returns
    (codes [N, K], scales [K/64, N], scaled zeros [K/64, N] = -zero, both 16-bit carriers)."""
    N, K = w.shape
    g = w.reshape(N, K // AWQ_GROUP, AWQ_GROUP).astype(F32)  # the converter works in float32
    lo, hi = g.min(-1), g.max(-1)
    scale = round16(np.maximum((hi - lo) / F32(15.0), F32(1e-8)), dtype)
    zero = round16(-lo, dtype)
    q = np.clip(np.rint((g + zero[..., None]) / scale[..., None]), 0, 15).astype(np.uint8).reshape(N, K)
    return q, scale.T.copy(), (-zero).T.copy()


def awq_gemv_w4a16(x: np.ndarray, q: np.ndarray, scales: np.ndarray, zeros: np.ndarray, dtype: str,
                   bias: np.ndarray | None = None) -> np.ndarray:
    """y[b, n] = sum_k x[b, k] * w16[n, k], gemv_awq.cu:181-236:
      w16 = fma16(q, scale[g, n], scaled_zero[g, n])   one 16-bit rounding (__hfma2, :201)
      p   = round16(w16 * x16)                         16-bit product (__hmul2, :222-224)
      y   = round16(sum_k float(p))                    fp32 accumulate, order unspecified (:225-226, warp_reduce)
    then the 16-bit bias add of AWQW4A16Linear.forward (linear.py:375-377: output.add_(bias)).
    x [B, K], q [N, K] codes, scales / zeros [K/64, N]; returns [B, N] float32 carrier."""
    B, K = x.shape
    N = q.shape[0]
    s = np.repeat(scales.T.astype(np.float64), AWQ_GROUP, axis=1)  # [N, K]
    z = np.repeat(zeros.T.astype(np.float64), AWQ_GROUP, axis=1)
    w16 = round16((q.astype(np.float64) * s + z).astype(np.float64), dtype).astype(np.float64)  # exact fma, then one rounding
    y = np.empty((B, N), dtype=F32)
    for b in range(B):
        p = round16((w16 * x[b].astype(np.float64)[None, :]), dtype).astype(np.float64)
        y[b] = round16(p.sum(axis=1).astype(F32), dtype)
    if bias is not None:
        y = round16(y + bias[None, :].astype(F32), dtype)
    return y


# --------------------------------------------------------------------------
# Element-wise glue of a block (extension kernels svdq_residual_gate_stats / svdq_quantize_args.ln_stats).
# The reference's V2 blocks do this with torch ops (transformer_flux_v2.py:118-342, normalization.py:85-98):
# hidden = hidden + gate * out; n = layer_norm(hidden) * scale + shift  -- restated with the same 16-bit rounding points.
# --------------------------------------------------------------------------
def _round16_fma(exact64: np.ndarray, dtype: str) -> np.ndarray:
    """16-bit result of ONE fp32 operation (add / mul / fma of 16-bit operands) whose exact value is ``exact64``:
    bf16 goes through the fp32 result (two roundings, as torch and the bf16 kernels do); fp16 is rounded ONCE --
    the gfx950 backend folds the fp32 operation and the conversion into v_fma_mix*_f16, in torch's half kernels
    and in ours alike (the GPU tests compare both bit for bit)."""
    if dtype == "fp16":
        with np.errstate(over="ignore"):
            return exact64.astype(np.float16).astype(F32)
    return round16(exact64.astype(F32), dtype)


def residual_gate_ref(res: np.ndarray, a: np.ndarray, gate: np.ndarray | None, b: np.ndarray | None, dtype: str) -> np.ndarray:
    """round16(res + round16(gate * t)), t = round16(a + b) if b is given else a: the reference's 16-bit torch ops
    ``residual + gate.unsqueeze(1) * (attn [+ mlp])`` (transformer_flux_v2.py:230-251, 332-335), one rounding each."""
    t = a.astype(np.float64) if b is None else _round16_fma(a.astype(np.float64) + b.astype(np.float64), dtype).astype(np.float64)
    if gate is not None:
        t = round16((gate.astype(np.float64)[None, :] * t).astype(F32), dtype).astype(np.float64)  # product exact in fp32
    return _round16_fma(res.astype(np.float64) + t, dtype)


def ln_stats_ref(y: np.ndarray, eps: float = RMS_EPS) -> np.ndarray:
    """[M, 2] float32 (mean, 1/sqrt(var + eps)), population variance, float64 accumulation."""
    y64 = y.astype(np.float64)
    mean = y64.mean(axis=1)
    var = ((y64 - mean[:, None]) ** 2).mean(axis=1)
    return np.stack([mean, 1.0 / np.sqrt(var + eps)], axis=1).astype(F32)


def ln_mod_ref(x: np.ndarray, stats: np.ndarray, scale: np.ndarray, shift: np.ndarray, dtype: str) -> np.ndarray:
    """round16(round16(round16((x - mean) * rstd) * scale) + shift): F.layer_norm without affine (fp32 math, 16-bit
    output) followed by the reference's 16-bit ``norm_x * scale[:, None] + shift[:, None]`` (normalization.py:96,164;
    the checkpoint's scale already contains the +1, scale_shift = 0); (x - mean) * rstd in float32."""
    # (x - mean) rounds to fp32, the product with rstd is then exact in float64 (24 x 24 bits); see _round16_fma for fp16
    ln = _round16_fma((x.astype(F32) - stats[:, 0:1]).astype(np.float64) * stats[:, 1:2].astype(np.float64), dtype)
    m = round16((ln.astype(np.float64) * scale.astype(np.float64)[None, :]).astype(F32), dtype)  # exact in fp32
    return _round16_fma(m.astype(np.float64) + shift.astype(np.float64)[None, :], dtype)


# --------------------------------------------------------------------------
# Attention over the packed QKV.  The reference computes it with its own fp16 flash
# kernel (src/kernels/zgemm/attention.cu:11-94, attention.cuh) or torch SDPA (attention_processors/flux.py:62-110): any
# correct softmax(QK^T/sqrt(d))V in 16-bit I/O.  This restatement follows the ARITHMETIC of nunchaku_amd/csrc/attention.hip
# tile by tile, so the GPU test can hold the kernel (plain grid and persistent schedule alike) to ~1 ulp of the 16-bit
# output instead of the loose "as good as SDPA" bar: it pins the kernel against its own specification, not against the
# reference's different-but-equivalent algorithm.
# --------------------------------------------------------------------------
def attention_tiled(q: np.ndarray, k: np.ndarray, v: np.ndarray, scale: float, dtype: str, kb: int = 64, wave_rows: int = 32,
                    defer_log2: float = 8.0) -> np.ndarray:
    """One head.  q [Lq, d], k / v [L, d]: 16-bit values carried as float32.  Per tile of ``kb`` keys:
    S = Q K^T in fp32; row maximum of the tile; the running (O, l) are rescaled only when some row of the 32-row wave
    block outgrew its running maximum by more than ``defer_log2`` (in log2 units, after scale * log2(e)) -- then EVERY
    row of the block takes max(m, m_tile); P = exp2(S c - m c) in fp32, ROUNDED to the 16-bit type; O += P16 V (fp32
    accumulate) and l += sum(P16) -- the row sum is over the rounded probabilities; out = round16(O / l)."""
    Lq, d = q.shape
    L = k.shape[0]
    assert L % kb == 0 and Lq % wave_rows == 0
    c = F32(scale * 1.4426950408889634)
    q, k, v = q.astype(F32), k.astype(F32), v.astype(F32)
    out = np.empty((Lq, d), dtype=F32)
    for w0 in range(0, Lq, wave_rows):
        qb = q[w0:w0 + wave_rows]
        o = np.zeros((wave_rows, d), dtype=F32)
        m = np.full(wave_rows, -np.inf, dtype=F32)
        l = np.zeros(wave_rows, dtype=F32)
        for t0 in range(0, L, kb):
            s = (qb @ k[t0:t0 + kb].T).astype(F32)
            mloc = s.max(axis=1)
            with np.errstate(invalid="ignore"):
                grow = (mloc - m) * c
            if (grow > F32(defer_log2)).any():
                m_new = np.maximum(m, mloc)
                with np.errstate(invalid="ignore"):
                    alpha = np.exp2(((m - m_new) * c).astype(F32)).astype(F32)
                alpha = np.where(np.isneginf(m), F32(0), alpha)
                o *= alpha[:, None]
                l *= alpha
                m = m_new
            p = np.exp2((s * c - (m * c)[:, None]).astype(F32)).astype(F32)
            p16 = round16(p, dtype)
            o += (p16 @ v[t0:t0 + kb]).astype(F32)
            l += p16.sum(axis=1, dtype=F32)
        out[w0:w0 + wave_rows] = o / l[:, None]
    return round16(out, dtype)
