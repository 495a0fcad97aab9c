"""GPU tests of the attention path: the QKV GEMM's transposed-V side output
and svdq_attention against a float32 torch restatement of softmax(QK^T/sqrt(d)) V on the same 16-bit inputs.

Floating-point tolerance (written here as the task demands): the kernel rounds the probabilities to the 16-bit
dtype before the PV MFMA (as flash attention implementations do, including the reference's attention_fp16),
so |err| <= 3 * 2^-8 (bf16) / 3 * 2^-11 (fp16) relative to max|out| is the bar, and it must not be worse than
1.5x the error of torch's own 16-bit SDPA on the same inputs."""

import math

import numpy as np
import pytest
import torch

from oracle import svdq_oracle as O
from tests.helpers import TORCH_DT, f32, make_module, t16

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _need_gpu(built_lib):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


@pytest.fixture(autouse=True, params=[1, 2], ids=["8x32", "4x64"])
def _geometry(request):
    """Every test of this module runs under both workgroup geometries of the attention kernel (svdq_attention_args.geometry;
    lengths that are not a multiple of 256 always run geometry 1)."""
    from nunchaku_amd._C import _Ops

    _Ops.attention_geometry = request.param
    yield request.param
    _Ops.attention_geometry = 0


def test_attention_geometries_agree():
    """On a Q that comes prescaled both geometries compute the same thing -- per 32-row tile: maxima, deferred move of the reference
    point, ascending key tiles, row sums over the ROUNDED probabilities -- in different summation orders (and geometry 2 keeps its
    scores relative: s' = s - m accumulated, not subtracted afterwards): outputs agree to two 16-bit ulps, mostly exactly."""
    from nunchaku_amd._C import _Ops
    from nunchaku_amd.ops.attention import attention_packed

    saved = (_Ops.attention_geometry, _Ops.attention_use_workspace)
    try:
        for dtype in (torch.bfloat16, torch.float16):
            ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
            for L, H in ((256, 2), (1024, 3), (2304, 5)):
                g = torch.Generator(device="cuda").manual_seed(L + H)
                qkv = (torch.randn(L, 3 * H * 128, device="cuda", generator=g) * 1.5).to(dtype)
                qkv, kw, _ = _as_produced_for(2, qkv, H)
                vt = qkv[:, 2 * H * 128:].t().contiguous()
                for ws in (False, True):
                    outs = []
                    for geo in (1, 2):
                        _Ops.attention_geometry, _Ops.attention_use_workspace = geo, ws
                        out = torch.empty(L, H * 128, device="cuda", dtype=dtype)
                        attention_packed(qkv, vt, H, out=out, **kw)
                        outs.append(out.float())
                    diff = (outs[0] - outs[1]).abs()
                    # (an output near zero is a sum of cancelling terms: the error scale is that of the terms, hence the absolute part)
                    assert (diff <= 2 * ulp * outs[0].abs() + 0.5 * ulp * outs[0].abs().max()).all(), (dtype, L, H, ws, diff.max().item())
                    assert (diff != 0).float().mean().item() < 0.25, (dtype, L, H, ws, (diff != 0).float().mean().item())
    finally:
        _Ops.attention_geometry, _Ops.attention_use_workspace = saved


def test_geometry_2_on_a_raw_q_is_less_accurate_and_only_runs_on_request():
    """Without q_prescaled the automatic choice is geometry 1.  Asked for explicitly, geometry 2 scales the 16-bit Q itself (a second
    rounding): correct, but 2-4x the error of geometry 1 against fp32 on peaky rows -- bounded here, documented in svdq_amd.h."""
    from nunchaku_amd._C import _Ops
    from nunchaku_amd.ops.attention import attention_packed

    L, H = 1024, 3
    saved = _Ops.attention_geometry
    try:
        for dtype in (torch.bfloat16, torch.float16):
            g = torch.Generator(device="cuda").manual_seed(9)
            qkv = torch.randn(L, 3 * H * 128, device="cuda", generator=g).to(dtype)
            qkv[: L // 2, : H * 128] *= 4.0
            q, k, v = (qkv[:, i * H * 128:(i + 1) * H * 128].unflatten(1, (H, 128)) for i in range(3))
            vt = v.permute(1, 2, 0).contiguous().view(H * 128, L)
            ref = _ref_attention(q, k, v).reshape(L, H * 128)
            errs = {}
            for geo in (0, 1, 2):
                _Ops.attention_geometry = geo
                errs[geo] = (attention_packed(qkv, vt, H).float() - ref).abs().max().item()
            assert errs[0] == errs[1], "automatic geometry on a raw Q must be geometry 1"
            assert errs[2] <= 5.0 * errs[1] + 1e-6 and errs[2] <= 8 * (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11) * ref.abs().max().item(), errs
    finally:
        _Ops.attention_geometry = saved


def _ref_attention(q, k, v, scale=None):
    """q, k, v: [L, H, D] 16-bit -> float32 [L, H, D]"""
    qf, kf, vf = (t.float().permute(1, 0, 2) for t in (q, k, v))
    s = qf @ kf.transpose(1, 2) * (1.0 / math.sqrt(q.shape[-1]) if scale is None else scale)
    return (torch.softmax(s, dim=-1) @ vf).permute(1, 0, 2)


def _as_produced_for(geometry, qkv, H):
    """Geometry 2 is built for a Q its PRODUCER multiplied by scale * log2(e) before the rounding to 16-bit (svdq_gemm_args.q_scale;
    ops.attention(q_prescaled=True)): the tests hand it such a Q -- the given one times that factor, rounded once -- and compare with a
    reference over the SAME 16-bit values, whose softmax scale is then ln 2.  Geometry 1 takes the buffer as it is.
    -> (qkv the kernel reads, keyword arguments of attention_packed, softmax scale of the reference)"""
    if geometry != 2:
        return qkv, {}, 1.0 / math.sqrt(128)
    from nunchaku_amd.ops.attention import q_prescale

    pre = qkv.clone()
    pre[:, : H * 128] = (qkv[:, : H * 128].float() * q_prescale(128)).to(qkv.dtype)
    return pre, {"q_prescaled": True}, math.log(2.0)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("L,H", [(128, 1), (384, 3), (1152, 2), (256, 2), (1024, 3)])  # L % 256 == 0: pipelined kernel
def test_attention_matches_fp32_reference(dtype, L, H, _geometry):
    from nunchaku_amd.ops.attention import attention_packed

    td = TORCH_DT[dtype]
    g = torch.Generator(device="cuda").manual_seed(L + H)
    qkv = torch.randn(L, 3 * H * 128, device="cuda", generator=g).to(td)
    # peaky rows too: scale some queries so the softmax is far from uniform
    qkv[: L // 2, : H * 128] *= 4.0
    qkv, kw, scale = _as_produced_for(_geometry, qkv, H)
    q, k, v = (qkv[:, i * H * 128:(i + 1) * H * 128].unflatten(1, (H, 128)) for i in range(3))
    vt = v.permute(1, 2, 0).contiguous().view(H * 128, L)
    out = attention_packed(qkv, vt, H, **kw)
    ref = _ref_attention(q, k, v, scale).reshape(L, H * 128)
    err = (out.float() - ref).abs().max().item()
    sd = torch.nn.functional.scaled_dot_product_attention(q.permute(1, 0, 2)[None], k.permute(1, 0, 2)[None], v.permute(1, 0, 2)[None], scale=scale)
    err_sdpa = (sd[0].permute(1, 0, 2).reshape(L, H * 128).float() - ref).abs().max().item()
    tol = 3 * (2.0 ** -8 if dtype == "bf16" else 2.0 ** -11) * ref.abs().max().item()
    assert err <= tol, f"attention error {err:.3g} > {tol:.3g}"
    assert err <= 1.5 * err_sdpa + 1e-6, f"attention error {err:.3g} vs torch 16-bit SDPA {err_sdpa:.3g}"


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("L,H", [(256, 3), (1024, 3), (512, 130), (4608, 24)])
def test_attention_persistent_schedule_matches_plain_grid(dtype, L, H, _geometry):
    """With a workspace the launch is persistent (one workgroup per CU, tasks split along the keys, partial (O, m, l) merged
    by the task's owner: 2 workgroups per task at (256, 3), 8 at (1024, 3), the FLUX.1 case last).  Same result as the
    plain grid up to fp32 summation order, bit-reproducible from launch to launch, counters left at zero."""
    from nunchaku_amd import _lib
    from nunchaku_amd._C import _Ops, ops
    from nunchaku_amd.ops.attention import attention_packed

    assert _lib.load().svdq_attention_schedule(L, H, 256, None, 0) > 0  # this shape does take the persistent path
    td = TORCH_DT[dtype]
    g = torch.Generator(device="cuda").manual_seed(L * 7 + H)
    qkv = torch.randn(L, 3 * H * 128, device="cuda", generator=g).to(td)
    qkv[: L // 2, : H * 128] *= 4.0
    qkv, kw, scale = _as_produced_for(_geometry, qkv, H)
    vt = qkv[:, 2 * H * 128:].t().contiguous()
    try:
        _Ops.attention_use_workspace = False
        plain = attention_packed(qkv, vt, H, **kw)
    finally:
        _Ops.attention_use_workspace = True
    runs = [attention_packed(qkv, vt, H, **kw) for _ in range(3)]
    ops.attention_workspace_status()
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    ulp = 2.0 ** -8 if dtype == "bf16" else 2.0 ** -11
    diff = (runs[0].float() - plain.float()).abs().max().item()
    assert diff <= 2 * ulp * plain.float().abs().max().item(), f"persistent vs plain grid: {diff:.3g}"
    if L * H <= 70000:
        q, k, v = (qkv[:, i * H * 128:(i + 1) * H * 128].unflatten(1, (H, 128)) for i in range(3))
        ref = _ref_attention(q, k, v, scale).reshape(L, H * 128)
        assert (runs[0].float() - ref).abs().max().item() <= 3 * ulp * ref.abs().max().item()


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("L,H", [(256, 2), (1024, 3), (1152, 2)])
def test_attention_matches_its_tile_by_tile_restatement(dtype, L, H, _geometry):
    """oracle.attention_tiled follows the kernel's arithmetic tile by tile (deferred rescale per 32-row wave block, probabilities
    rounded to 16 bits, row sums over the ROUNDED probabilities): the kernel -- plain grid (L = 1152: the 4-wave kernel) and
    persistent schedule with its split tasks (the others) -- must agree to about one unit in the last place of the 16-bit output.
    What remains is the hardware exp2 (1 ulp of fp32, which can move a probability across a 16-bit rounding boundary) and the
    fp32 summation order inside the MFMA and across merged partial results."""
    from nunchaku_amd.ops.attention import attention_packed

    td = TORCH_DT[dtype]
    g = torch.Generator(device="cuda").manual_seed(17 * L + H)
    qkv = torch.randn(L, 3 * H * 128, device="cuda", generator=g).to(td)
    qkv[: L // 2, : H * 128] *= 4.0  # peaky rows: the deferred rescale and the dominated-row case both occur
    qkv, kw, scale = _as_produced_for(_geometry, qkv, H)  # (geometry 2: the restatement sees the same prescaled Q, scale ln 2 -> c = 1)
    vt = qkv[:, 2 * H * 128:].t().contiguous()
    out = f32(attention_packed(qkv, vt, H, **kw)).reshape(L, H, 128)
    x = f32(qkv).reshape(L, 3, H, 128)
    ulp = 2.0 ** -8 if dtype == "bf16" else 2.0 ** -11
    worst, off = 0.0, 0.0
    for h in range(H):
        ref = O.attention_tiled(x[:, 0, h], x[:, 1, h], x[:, 2, h], scale, dtype)
        # the error scale of a sum of rounded terms is ulp * sum p |v| (the same weights applied to |V|), not ulp * |result|:
        # an output near zero is a sum of cancelling terms
        cond = O.attention_tiled(x[:, 0, h], x[:, 1, h], np.abs(x[:, 2, h]), scale, dtype)
        err = np.abs(out[:, h] - ref) / cond
        worst = max(worst, float(err.max()))
        off = max(off, float((err > 1.0 * ulp).mean()))
    # (the final 16-bit rounding alone moves an output by up to one of its own ulps when o * (1 / l) and o / l differ in the last fp32 bit)
    assert worst <= 2.5 * ulp and off <= 1e-2, f"vs the tiled restatement: worst {worst / ulp:.2f} ulp of sum p|v|, {off:.2e} of the outputs beyond 1 ulp"


def test_attention_strided_heads_and_errors():
    from nunchaku_amd._C import _Ops, ops

    L, H = 256, 2
    g = torch.Generator(device="cuda").manual_seed(5)
    q = torch.randn(H, L, 128, device="cuda", generator=g).bfloat16()  # head-major storage (the reference's [B, H, L, D])
    k = torch.randn(H, L, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(H, L, 128, device="cuda", generator=g).bfloat16()
    out = torch.empty(L, H, 128, device="cuda", dtype=torch.bfloat16)
    ops.attention(q.permute(1, 0, 2), k.permute(1, 0, 2), v.transpose(1, 2).contiguous(), out, 1 / math.sqrt(128))
    ref = _ref_attention(q.permute(1, 0, 2), k.permute(1, 0, 2), v.permute(1, 0, 2))
    assert (out.float() - ref).abs().max().item() <= 3 * 2.0 ** -8 * ref.abs().max().item()
    with pytest.raises(ValueError):
        ops.attention(q.permute(1, 0, 2)[:200], k.permute(1, 0, 2)[:200], v.transpose(1, 2)[:, :, :200].contiguous(), out[:200], 0.1)
    with pytest.raises(ValueError):
        ops.attention(q.permute(1, 0, 2), k.permute(1, 0, 2), v, out, 0.1)  # V not transposed


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("M", [256, 300, 301])
def test_qkv_gemm_transposed_v_output(dtype, M):
    """out_vt receives exactly the V columns the plain epilogue would have stored, transposed; the V third of
    `out`, the padding columns of out_vt and the neighbouring token range are not touched."""
    from nunchaku_amd.ops.fused import fused_qkv_norm_rottary
    from tests.test_gpu_parity import _gemm_inputs

    K, H = 256, 2
    N = 3 * H * 128
    Lyr, x = _gemm_inputs(M, K, N, 32, dtype, seed=21)
    rng = np.random.default_rng(22)
    M_pad = O.ceil_div(M, 256) * 256
    ang = rng.uniform(0, 6.28, (M_pad, 64)).astype(np.float32)
    rot = np.stack([np.sin(ang), np.cos(ang)], axis=-1).astype(np.float32)
    packed = torch.from_numpy(O.pack_rotemb_ref(rot)).cuda().view(1, M_pad, 128)
    mod = make_module(Lyr, dtype)

    class W:
        def __init__(self):
            self.weight = torch.ones(128, device="cuda", dtype=TORCH_DT[dtype])

    xin = t16(x, dtype).view(1, M, K)
    torch.manual_seed(0)
    full = fused_qkv_norm_rottary(xin, mod, W(), W(), packed)[0]
    sentinel = 7.0
    out = torch.full((M, N), sentinel, device="cuda", dtype=TORCH_DT[dtype])
    off = 6  # even token offset inside a wider joint buffer
    vt = torch.full((H * 128, off + M + 10 + (M & 1)), sentinel, device="cuda", dtype=TORCH_DT[dtype])
    fused_qkv_norm_rottary(xin, mod, W(), W(), packed, output=out, out_vt=vt[:, off:off + M])
    # lora_act comes from fp32 atomics only when K is sliced; K = 256 is one slice -> both runs are identical
    assert torch.equal(out[:, : 2 * N // 3], full[:, : 2 * N // 3])
    assert torch.equal(vt[:, off:off + M], full[:, 2 * N // 3:].t())
    assert (out[:, 2 * N // 3:] == sentinel).all()
    assert (vt[:, :off] == sentinel).all() and (vt[:, off + M:] == sentinel).all()


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_qkv_gemm_q_scale_scales_q_before_its_rounding(dtype):
    """svdq_gemm_args.q_scale (RMSNorm+RoPE epilogue): the Q third is the plain epilogue's Q times the factor with ONE rounding -- at most one
    16-bit ulp from (rounded Q) x factor, and strictly closer to it on average than a second rounding would be; K and V are untouched; a
    factor of 1 is bit-identical to none.  Then the consumer: attention(q_prescaled) on that buffer against attention on the plain one."""
    from nunchaku_amd._C import _Ops
    from nunchaku_amd.ops.attention import attention_packed, q_prescale
    from nunchaku_amd.ops.fused import fused_qkv_norm_rottary
    from tests.test_gpu_parity import _gemm_inputs

    M, K, H = 512, 256, 2
    N = 3 * H * 128
    Lyr, x = _gemm_inputs(M, K, N, 32, dtype, seed=31)
    rng = np.random.default_rng(32)
    ang = rng.uniform(0, 6.28, (M, 64)).astype(np.float32)
    packed = torch.from_numpy(O.pack_rotemb_ref(np.stack([np.sin(ang), np.cos(ang)], axis=-1).astype(np.float32))).cuda().view(1, M, 128)
    mod = make_module(Lyr, dtype)

    class W:
        def __init__(self):
            self.weight = torch.ones(128, device="cuda", dtype=TORCH_DT[dtype])

    xin = t16(x, dtype).view(1, M, K)
    td = TORCH_DT[dtype]
    c = q_prescale(128)
    outs = {}
    for qs in (0.0, 1.0, c):
        out = torch.empty(M, N, device="cuda", dtype=td)
        vt = torch.empty(H * 128, M, device="cuda", dtype=td)
        fused_qkv_norm_rottary(xin, mod, W(), W(), packed, output=out, out_vt=vt, q_scale=qs)
        outs[qs] = (out, vt)
    plain, one, scaled = outs[0.0][0], outs[1.0][0], outs[c][0]
    assert torch.equal(plain[:, : 2 * H * 128], one[:, : 2 * H * 128]) and torch.equal(outs[0.0][1], outs[1.0][1])
    assert torch.equal(plain[:, H * 128: 2 * H * 128], scaled[:, H * 128: 2 * H * 128]) and torch.equal(outs[0.0][1], outs[c][1])
    q0, q1 = plain[:, : H * 128].float(), scaled[:, : H * 128].float()
    ulp = 2.0 ** -8 if dtype == "bf16" else 2.0 ** -11
    assert ((q1 - q0 * c).abs() <= 2.02 * ulp * (q0 * c).abs() + 1e-7).all()       # two roundings apart at most from (rounded q) x c ...
    twice = (q0 * c).to(td).float()                                                 # ... and not simply that value rounded again
    assert (q1 != twice).float().mean().item() > 0.05
    saved = _Ops.attention_geometry
    try:
        _Ops.attention_geometry = 0  # automatic: geometry 1 on the plain buffer, geometry 2 on the prescaled one
        a_plain = attention_packed(plain, outs[0.0][1], H).float()
        a_pre = attention_packed(scaled, outs[c][1], H, q_prescaled=True).float()
    finally:
        _Ops.attention_geometry = saved
    q, k = (plain[:, i * H * 128:(i + 1) * H * 128].unflatten(1, (H, 128)) for i in range(2))
    v = outs[0.0][1].t().unflatten(1, (H, 128))  # (with out_vt the V third of `out` is not written)
    ref = _ref_attention(q, k, v).reshape(M, H * 128)
    ref_pre = _ref_attention(scaled[:, : H * 128].unflatten(1, (H, 128)), k, v, math.log(2.0)).reshape(M, H * 128)  # over the values the kernel read
    e_plain, e_pre = (a_plain - ref).abs().max().item(), (a_pre - ref_pre).abs().max().item()
    assert e_pre <= 3 * ulp * ref.abs().max().item() and e_pre <= 2.0 * e_plain + 1e-6, (e_plain, e_pre)
    # and the two pipelines agree with each other to the rounding of Q (different 16-bit inputs: not a kernel property, a sanity bound)
    assert (a_pre - a_plain).abs().max().item() <= 16 * ulp * ref.abs().max().item()


def test_flux_transformer_svdq_attention_vs_sdpa():
    """The small FLUX-shaped transformer with this library's attention vs torch SDPA on the same weights."""
    from nunchaku_amd.models.flux import FluxAttentionAMD, FluxTransformerAMD

    torch.manual_seed(3)
    model = FluxTransformerAMD(num_layers=1, num_single_layers=2, dim=256, heads=2, in_channels=64, joint_attention_dim=128,
                               pooled_projection_dim=64, device="cuda")
    model.init_synthetic_(seed=1)
    model.eval()
    side, t_txt = 16, 128
    lat = torch.randn(1, side * side, 64, device="cuda").bfloat16()
    enc = torch.randn(1, t_txt, 128, device="cuda").bfloat16()
    pooled = torch.randn(1, 64, device="cuda").bfloat16()
    img_ids = torch.zeros(side * side, 3, device="cuda")
    img_ids[:, 1] = torch.arange(side, device="cuda").repeat_interleave(side)
    img_ids[:, 2] = torch.arange(side, device="cuda").repeat(side)
    txt_ids = torch.zeros(t_txt, 3, device="cuda")
    t, gd = torch.tensor([0.7], device="cuda"), torch.tensor([3.5], device="cuda")
    outs = {}
    try:
        from nunchaku_amd import mode

        for impl in ("svdq", "sdpa"):
            FluxAttentionAMD.attention_impl = impl
            with torch.no_grad(), mode.deterministic_mode():
                outs[impl] = model(lat, enc, pooled, t, img_ids, txt_ids, gd)[0].float()
    finally:
        FluxAttentionAMD.attention_impl = "svdq"
    from tests.helpers import psnr_db

    rel = ((outs["svdq"] - outs["sdpa"]).norm() / outs["sdpa"].norm()).item()
    psnr = psnr_db(outs["svdq"], outs["sdpa"])
    print(f"model with svdq vs sdpa attention: PSNR {psnr:.1f} dB rel {rel:.2e}")
    # two attention kernels (fp32 summation orders differ) in front of W4A4 layers: last-bit differences flip 4-bit codes;
    # measured on this tiny random-weight model: 39.0 dB / 3.8e-2 (deterministic mode: the low-rank atomics are not the cause)
    assert torch.isfinite(outs["svdq"]).all() and rel < 5e-2 and psnr > 35.0, f"svdq vs sdpa attention: relative L2 {rel:.3g}, PSNR {psnr:.1f} dB"


def test_attention_full_size_properties():
    """FLUX.1 size (24 heads x 4608 tokens), properties that need no oracle: every output is a convex combination of
    the values (inside their per-channel range), permuting the keys together with the values changes nothing beyond
    summation order, and a key block repeated twice gives the same answer as once."""
    from nunchaku_amd.ops.attention import attention_packed

    L, H = 4608, 24
    g = torch.Generator(device="cuda").manual_seed(11)
    qkv = torch.randn(L, 3 * H * 128, device="cuda", generator=g).bfloat16()
    v = qkv[:, 2 * H * 128:]
    out = attention_packed(qkv, v.t().contiguous(), H)
    assert torch.isfinite(out.float()).all()
    lo, hi = v.float().amin(0), v.float().amax(0)
    eps = 2.0 ** -7 * v.float().abs().amax()
    assert (out.float() >= lo - eps).all() and (out.float() <= hi + eps).all()
    perm = torch.randperm(L, device="cuda", generator=g)
    qkv2 = qkv.clone()
    qkv2[:, H * 128:] = qkv[perm][:, H * 128:]  # keys and values permuted together, queries in place
    out2 = attention_packed(qkv2, qkv2[:, 2 * H * 128:].t().contiguous(), H)
    assert (out.float() - out2.float()).abs().max() <= 2.0 ** -6 * out.float().abs().max()
    # duplicate keys/values: softmax over [K; K] with [V; V] equals softmax over K with V
    half = L // 2
    qkv3 = qkv.clone()
    qkv3[half:, H * 128:] = qkv[:half, H * 128:]
    out3 = attention_packed(qkv3, qkv3[:, 2 * H * 128:].t().contiguous(), H)
    first = qkv[:half].contiguous()  # 2304 tokens: the 4-wave kernel, keys once instead of twice
    ref = attention_packed(first, first[:, 2 * H * 128:].t().contiguous(), H)
    assert (out3[:half].float() - ref.float()).abs().max() <= 2.0 ** -6 * ref.float().abs().max()


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("joint", [False, True])
@pytest.mark.parametrize("R,H,split", [(32, 2, True), (48, 2, True), (48, 2, False), (128, 2, True), (128, 2, False), (128, 6, True), (128, 6, False), (144, 6, True), (160, 2, True), (176, 2, True)],
                         ids=["r32", "r48-split", "r48-passes", "r128-split", "r128-passes", "r128-H6-split", "r128-H6-passes", "r144-H6-split", "r160-split", "r176-passes"])
def test_attention_fused_output_quantiser(dtype, joint, R, H, split):
    """The attention epilogue emits the output projection's quantised input: identical codes and scales to quantising the
    16-bit attention output with the stand-alone kernel; lora_act up to fp32 summation order.  Rank 48 .. 160 with the workspace ops.attention asks for: the
    low-rank down projection runs SPLIT (the epilogue stores 16-bit fragments, lowrank_down_split_kernel contracts them; ABI 20) -- `split=False` keeps the
    in-epilogue 32-rank passes, which also serve rank 32 and ranks beyond 160."""
    from nunchaku_amd import layout
    from nunchaku_amd._C import _Ops
    from nunchaku_amd.ops.attention import attention_packed, attention_packed_quantized

    L = 512   # (r128 = the reference's Qwen-Image / FLUX r128 checkpoints; H = 6: the contraction's K is split over three workgroups per row group)
    K = H * 128
    _Ops.attention_split_lowrank = split
    try:
        _fused_output_quantiser(dtype, joint, R, H, L, K)
        from nunchaku_amd._C import ops

        assert ops.attention_last_plan()["split_lowrank"] == (split and 48 <= R <= 160), (R, split, ops.attention_last_plan())  # nothing fell back
    finally:
        _Ops.attention_split_lowrank = True


def _fused_output_quantiser(dtype, joint, R, H, L, K):
    from nunchaku_amd import layout
    from nunchaku_amd.ops.attention import attention_packed, attention_packed_quantized

    td = TORCH_DT[dtype]
    g = torch.Generator(device="cuda").manual_seed(3)
    qkv = torch.randn(L, 3 * K, device="cuda", generator=g).to(td)
    vt = qkv[:, 2 * K:].t().contiguous()
    La = O.make_svdq_layer(K, 128, R, seed=1, dtype=dtype, cheap=True)
    Lb = O.make_svdq_layer(K, 128, R, seed=2, dtype=dtype, cheap=True)
    la, lb = make_module(La, dtype), make_module(Lb, dtype)
    o = attention_packed(qkv, vt, H)
    assert torch.equal(o, attention_packed(qkv, vt, H)), "two svdq_attention launches on identical inputs differ"  # (persistent schedule: partials merged in a fixed order)
    if joint:
        got = attention_packed_quantized(qkv, vt, H, lb, lin_first=la, split_rows=256)
        parts = [la.quantize(o[:256]), lb.quantize(o[256:])]
        ref_codes = torch.cat([layout.unpack_act(p[0], K) for p in parts])
        ref_scales = torch.cat([layout.unpack_scales(p[1], 256) for p in parts], dim=1)
        ref_la = torch.cat([p[2] for p in parts])
    else:
        got = attention_packed_quantized(qkv, vt, H, lb)
        p = lb.quantize(o)
        ref_codes, ref_scales, ref_la = layout.unpack_act(p[0], K), layout.unpack_scales(p[1], L), p[2]
    assert got is not None
    assert torch.equal(layout.unpack_act(got[0], K), ref_codes)
    # bit equality at every H, rank and dtype.  (Round 5 held H = 6 to "<= 1e-3 of the scales, one step": 2 of 6144 fp16 scales differed.  Round 6 found the cause
    # in the ISA listing: the fp16 build folded the attention epilogue's `(T)(amax * (1/7))` into ONE v_fma_mixlo_f16 -- a single rounding of the exact product --
    # while the stand-alone quantiser (and the oracle, and the reference: gemm_w4a4.cuh:429-523) round the fp32 product and then convert; the two disagree by a
    # 16-bit step on ~3e-4 of the scales and never on a code.  No schedule, merge order or race was involved: two launches on identical inputs are bit-equal
    # (asserted above).  attention.hip now keeps the product opaque before the conversion.)
    gs = layout.unpack_scales(got[1], L)
    bad = (gs != ref_scales).nonzero()
    assert bad.numel() == 0, f"{bad.shape[0]} scales differ; first (group, row): {bad[:8].tolist()}"
    assert (got[2] - ref_la).abs().max() <= 2e-3 * ref_la.abs().max() + 1e-5


def test_reference_fp16_attention_operators(_geometry):
    """The reference's "nunchaku-fp16" attention surface (attention_processors/flux.py:114-237, csrc/ops.h:114-121):
    fused_qkv_norm_rottary(..., output=(q, k, v), attn_tokens=) + _C.ops.attention_fp16, through the nunchaku shim,
    against the SDPA processor on the same module."""
    from nunchaku.models.attention_processors.flux import NunchakuFluxFA2Processor, NunchakuFluxFP16AttnProcessor
    from nunchaku_amd.models.flux import FluxAttentionAMD
    from nunchaku_amd.models.embeddings import flux_pos_embed, pack_rotemb
    from nunchaku_amd.models.flux import FluxTransformerAMD

    model = FluxTransformerAMD(num_layers=1, num_single_layers=1, dim=256, heads=2, in_channels=64, joint_attention_dim=128,
                               pooled_projection_dim=64, torch_dtype=torch.float16, device="cuda").init_synthetic_(seed=2).eval()
    g = torch.Generator(device="cuda").manual_seed(0)
    t_txt, t_img = 256, 256
    x = torch.randn(1, t_img, 256, device="cuda", generator=g).half()
    e = torch.randn(1, t_txt, 256, device="cuda", generator=g).half()
    ids = torch.zeros(t_txt + t_img, 3, device="cuda")
    ids[t_txt:, 1] = torch.arange(16, device="cuda").repeat_interleave(16)
    ids[t_txt:, 2] = torch.arange(16, device="cuda").repeat(16)
    rot = flux_pos_embed(ids, (16, 56, 56))
    rot_txt, rot_img, rot_all = pack_rotemb(rot[:, :t_txt]), pack_rotemb(rot[:, t_txt:]), pack_rotemb(rot)
    joint, single = model.blocks[0].attn, model.single_blocks[0].attn
    assert isinstance(joint, FluxAttentionAMD) and joint.added_kv_proj_dim is not None and single.added_kv_proj_dim is None
    with torch.no_grad():
        for proc_ref, proc in ((NunchakuFluxFA2Processor(), NunchakuFluxFP16AttnProcessor()),):
            a0, c0 = proc_ref(joint, x, e, image_rotary_emb=(rot_img, rot_txt))
            a1, c1 = proc(joint, x, e, image_rotary_emb=(rot_img, rot_txt))
            s0 = proc_ref(single, torch.cat([e, x], 1), image_rotary_emb=rot_all)
            s1 = proc(single, torch.cat([e, x], 1), image_rotary_emb=rot_all)
    # The outputs pass through the W4A4 output projection: attention results that differ in the last bit (different kernels'
    # summation order) flip a few 4-bit codes, so single elements move by a few per cent of the maximum while the bulk agrees.
    from tests.helpers import psnr_db

    for got, ref in ((a1, a0), (c1, c0), (s1, s0)):
        err = (got.float() - ref.float()).abs().max().item()
        rel = ((got.float() - ref.float()).norm() / ref.float().norm()).item()
        psnr = psnr_db(got, ref)
        print(f"nunchaku-fp16 processor vs SDPA processor: PSNR {psnr:.1f} dB rel {rel:.2e} max err {err:.3e}")
        # (the processors hand a RAW Q to attention_fp16: under the 4x64 fixture that is geometry 2's explicit raw-Q path, a second rounding of Q)
        assert torch.isfinite(got.float()).all() and rel <= (2e-2 if _geometry == 2 else 1e-2) and psnr >= 45.0, (err, rel, psnr)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("L,valid", [(256, (200,)), (512, (300, 384, 500)), (512, (64, 256, 257)), (768, (700,)), (512, (512,))],
                         ids=["tail", "two-ranges", "short-ranges", "persistent-size-tail", "nothing-masked"])
def test_key_padding_mask_matches_masked_sdpa(dtype, L, valid):
    """svdq_attention_args.kv_len0 / kv_start1 / kv_end1: padded keys get probability 0 whatever the padded K rows hold (NaN
    included); real query rows equal torch SDPA over the real keys only.  Tiles fully inside the padding and a persistent-schedule
    segment that starts in the padding are covered by the sizes."""
    from nunchaku_amd._C import ops

    H, D = 3, 128
    g = torch.Generator(device="cuda").manual_seed(L + len(valid))
    q = torch.randn(L, H, D, device="cuda", generator=g).to(dtype)
    k = torch.randn(L, H, D, device="cuda", generator=g).to(dtype)
    v = torch.randn(L, H, D, device="cuda", generator=g).to(dtype)
    real = torch.zeros(L, dtype=torch.bool, device="cuda")
    real[: valid[0]] = True
    if len(valid) == 3:
        real[valid[1]:valid[2]] = True
    k_pad = k.clone()
    k_pad[~real] = float("nan")          # padded K rows may hold anything
    v_pad = v.clone()
    v_pad[~real] = 0                      # padded V rows must be finite
    out = torch.empty(L, H, D, device="cuda", dtype=dtype)
    ops.attention(q, k_pad, v_pad.permute(1, 2, 0).contiguous(), out, D ** -0.5, kv_valid=valid)
    torch.cuda.synchronize()
    ref = torch.nn.functional.scaled_dot_product_attention(q.float().transpose(0, 1), k[real].float().transpose(0, 1),
                                                           v[real].float().transpose(0, 1)).transpose(0, 1)
    got = out.float()[real]
    assert torch.isfinite(got).all()
    err = (got - ref[real]).abs().max().item()
    assert err <= (2e-2 if dtype == torch.bfloat16 else 4e-3), err
    ops.attention_workspace_status()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("L,valid", [(512, (300, 384, 500)), (768, (700,)), (1024, (37, 256, 1000)), (4608, (4592,)), (2560, (300, 512, 2500))],
                         ids=["two-ranges", "tail", "qwen-like", "flux-1360x768", "odd-main-run"])
def test_key_padding_mask_on_the_4x64_geometry(dtype, L, valid):
    """Round 4: a masked launch with a prescaled Q runs geometry 2 -- the assembly loop over the longest even run of fully real tiles, every other
    tile with a real key as a C++ "extra" tile that continues the softmax state (scores against the current reference point, padded keys -inf, the
    reference moved when a row outgrows it).  Against fp32 softmax over the real keys only, on the SAME prescaled 16-bit Q; the extra tiles hold
    keys whose scores EXCEED everything the loop saw (the reference point must move there), padded K rows hold NaN."""
    import ctypes as C

    from nunchaku_amd import _lib
    from nunchaku_amd._C import _Ops, ops

    H, D = 3, 128
    g = torch.Generator(device="cuda").manual_seed(L + sum(valid))
    scale = D ** -0.5
    q = torch.randn(L, H, D, device="cuda", generator=g)
    qp = (q * (scale * 1.4426950408889634)).to(dtype)      # what the QKV GEMM's epilogue emits (q_scale)
    k = torch.randn(L, H, D, device="cuda", generator=g)
    v = torch.randn(L, H, D, device="cuda", generator=g).to(dtype)
    real = torch.zeros(L, dtype=torch.bool, device="cuda")
    real[: valid[0]] = True
    if len(valid) == 3:
        real[valid[1]:valid[2]] = True
    a = _lib.AttentionArgs()
    a.L, a.H, a.head_dim, a.q_prescaled, a.kv_len0 = L, H, D, 1, valid[0]
    if len(valid) == 3:
        a.kv_start1, a.kv_end1 = valid[1], valid[2]
    plan = (C.c_int32 * 4)()
    assert _lib.load().svdq_attention_plan(C.byref(a), plan) == 0
    assert plan[0] == 2 and plan[1] == 1 and plan[3] - plan[2] >= 2 and (plan[3] - plan[2]) % 2 == 0, list(plan)
    # keys of the extra tiles: make a few of them line up with the queries, so that their scores are the largest of the row by far
    extra = real.clone()
    extra[plan[2] * 64: plan[3] * 64] = False
    idx = extra.nonzero().flatten()
    k[idx[::3]] = q[idx[::3]] * 3.0
    k = k.to(dtype)
    k_pad = k.clone()
    k_pad[~real] = float("nan")
    v_pad = v.clone()
    v_pad[~real] = 0
    out = torch.empty(L, H, D, device="cuda", dtype=dtype)
    _Ops.attention_geometry = 0  # automatic (the module's fixture pins a geometry per test): a prescaled Q + a mask -> geometry 2
    ops.attention(qp, k_pad, v_pad.permute(1, 2, 0).contiguous(), out, scale, kv_valid=valid, q_prescaled=True)
    torch.cuda.synchronize()
    # softmax in base 2 over the real keys: scores = qp . k are already in log2 units
    s_ = torch.einsum("lhd,mhd->hlm", qp.float(), k[real].float())
    p_ = torch.softmax(s_ * 0.6931471805599453, dim=-1)
    ref = torch.einsum("hlm,mhd->lhd", p_, v[real].float())
    got = out.float()
    assert torch.isfinite(got[real]).all()
    err = (got[real] - ref[real]).abs().max().item()
    print(f"masked geometry 2, L={L} valid={valid} plan={list(plan)}: max err {err:.3e}")
    assert err <= (2e-2 if dtype == torch.bfloat16 else 4e-3), err
    # and it agrees with geometry 1 on the same inputs (explicit geometry) to two 16-bit ulps of the output scale
    _Ops.attention_geometry = 1
    out1 = torch.empty_like(out)
    ops.attention(qp, k_pad, v_pad.permute(1, 2, 0).contiguous(), out1, scale, kv_valid=valid, q_prescaled=True)
    d = (out1.float()[real] - got[real]).abs().max().item()
    assert d <= (3e-2 if dtype == torch.bfloat16 else 6e-3), d


def test_fp16_attention_processor_with_padded_token_counts():
    """The "nunchaku-fp16" surface with token counts that are NOT multiples of the pad size (VERDICT r2 missing #5): the packed
    Q/K/V buffers are padded per stream, the padding of the text stream sits in the MIDDLE of the joint sequence, and the result
    must match the SDPA processor on the unpadded tokens."""
    from nunchaku.models.attention_processors.flux import NunchakuFluxFA2Processor, NunchakuFluxFP16AttnProcessor
    from nunchaku_amd import mode
    from nunchaku_amd.models.embeddings import flux_pos_embed, pack_rotemb
    from nunchaku_amd.models.flux import FluxTransformerAMD
    from nunchaku_amd.utils import pad_tensor

    model = FluxTransformerAMD(num_layers=1, num_single_layers=1, dim=256, heads=2, in_channels=64, joint_attention_dim=128,
                               pooled_projection_dim=64, torch_dtype=torch.float16, device="cuda").init_synthetic_(seed=2).eval()
    g = torch.Generator(device="cuda").manual_seed(3)
    t_txt, t_img = 77, 300
    x = torch.randn(1, t_img, 256, device="cuda", generator=g).half()
    e = torch.randn(1, t_txt, 256, device="cuda", generator=g).half()
    ids = torch.zeros(t_txt + t_img, 3, device="cuda")
    ids[t_txt:, 1] = torch.arange(t_img, device="cuda") // 20
    ids[t_txt:, 2] = torch.arange(t_img, device="cuda") % 20
    rot = flux_pos_embed(ids, (16, 56, 56))
    pk = lambda r: pack_rotemb(pad_tensor(r, 256, 1))
    rot_txt, rot_img, rot_all = pk(rot[:, :t_txt]), pk(rot[:, t_txt:]), pk(rot)
    joint, single = model.blocks[0].attn, model.single_blocks[0].attn
    with torch.no_grad(), mode.deterministic_mode():
        a0, c0 = NunchakuFluxFA2Processor()(joint, x, e, image_rotary_emb=(rot_img, rot_txt))
        a1, c1 = NunchakuFluxFP16AttnProcessor()(joint, x, e, image_rotary_emb=(rot_img, rot_txt))
        s0 = NunchakuFluxFA2Processor()(single, torch.cat([e, x], 1), image_rotary_emb=rot_all)
        s1 = NunchakuFluxFP16AttnProcessor()(single, torch.cat([e, x], 1), image_rotary_emb=rot_all)
    for name, got, ref in (("img", a1, a0), ("txt", c1, c0), ("single", s1, s0)):
        assert got.shape == ref.shape and torch.isfinite(got.float()).all(), name
        rel = ((got.float() - ref.float()).norm() / ref.float().norm()).item()
        psnr = 10 * torch.log10(ref.float().abs().max() ** 2 / ((got.float() - ref.float()) ** 2).mean()).item()
        # two attention kernels (fp32 summation order) in front of a W4A4 projection: 4-bit code flips, no more
        assert rel <= 2e-2 and psnr >= 38.0, (name, rel, psnr)
