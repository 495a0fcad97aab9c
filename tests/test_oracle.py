"""Self-consistency of the CPU oracle (the checker must be trustworthy before it checks anything)."""

import numpy as np
import pytest

from oracle import svdq_oracle as O


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_quantizer_definition(dtype):
    x = O.make_activations(40, 256, seed=3, dtype=dtype)
    smooth = O.round16(np.exp(np.random.default_rng(0).standard_normal(256) * 0.3).astype(np.float32), dtype)
    q, asc, la = O.quantize_w4a4_act_fuse_lora(x, smooth, None, dtype)
    assert q.shape == (256, 256) and asc.shape == (4, 256) and la is None
    assert q.min() >= -7 and q.max() <= 7  # scale = amax/7 never needs -8
    assert np.all(q[40:] == 0) and np.all(asc[:, 40:] == 0)  # padded rows: zero codes, zero scale
    # every non-zero group attains |q| == 7 at its absmax element
    qg = np.abs(q[:40].reshape(40, 4, 64)).max(axis=2)
    assert np.all(qg[asc[:, :40].T > 0] == 7)
    # dequantised activations reproduce x/smooth to within half a step (plus the 16-bit scale rounding)
    xh = O.round16(x / smooth[None, :], dtype)
    deq = q[:40].reshape(40, 4, 64) * asc[:, :40].T[:, :, None]
    step = asc[:, :40].T[:, :, None]
    assert np.all(np.abs(deq - xh.reshape(40, 4, 64)) <= 0.5 * step * 1.02 + 1e-6)


def test_zero_input_is_all_zero():
    x = np.zeros((5, 128), dtype=np.float32)
    q, asc, _ = O.quantize_w4a4_act_fuse_lora(x, np.ones(128, np.float32), None)
    assert not q.any() and not asc.any()


def test_unsigned_quantizer_range():
    rng = np.random.default_rng(1)
    xh = O.bf16_round(np.abs(rng.standard_normal((16, 128))).astype(np.float32))
    q, asc = O.quantize_rows(xh, "bf16", unsigned=True)
    assert q.min() >= 0 and q.max() == 15


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_linear_modes_agree_and_track_dense(dtype):
    L = O.make_svdq_layer(256, 128, 32, seed=0, dtype=dtype)
    x = O.round16(np.random.default_rng(5).standard_normal((64, 256)).astype(np.float32), dtype)
    y32 = O.svdq_linear(x, L, dtype, "fp32")["out"]
    y16 = O.svdq_linear(x, L, dtype, "ref16")["out"]
    dense = x @ L["dense"].T + L["bias"]
    n = np.linalg.norm
    assert n(y32 - y16) / n(y32) < (2e-2 if dtype == "bf16" else 3e-3)  # 16-bit chain vs exact
    assert n(y32 - dense) / n(dense) < 0.25  # 4-bit x 4-bit noise on random weights
    # the low-rank branch matters: dropping it must make things worse
    L0 = dict(L)
    q, asc, la = O.quantize_w4a4_act_fuse_lora(x, L["smooth"], L["proj_down"], dtype)
    y_nolr = O.gemm_w4a4(q, asc, L["qweight"], L["wscales"], dtype=dtype, bias=L["bias"])["out"][:64]
    assert n(y_nolr - dense) > n(y32 - dense)


def test_int_group_dot_exact():
    rng = np.random.default_rng(2)
    qa = rng.integers(-8, 8, (8, 128)).astype(np.int8)
    qw = rng.integers(-8, 8, (16, 128)).astype(np.int8)
    p = O.int_group_dot(qa, qw)
    ref = np.einsum("mgk,ngk->gmn", qa.reshape(8, 2, 64).astype(np.int64), qw.reshape(16, 2, 64).astype(np.int64))
    assert np.array_equal(p, ref)


def test_rmsnorm_rope_identity_rotation():
    rng = np.random.default_rng(3)
    y = O.bf16_round(rng.standard_normal((16, 384)).astype(np.float32))
    rot = np.zeros((16, 64, 2), np.float32)
    rot[:, :, 1] = 1.0  # sin = 0, cos = 1
    w = np.ones(128, np.float32)
    out = O.rmsnorm_rope(y, w, w, rot, "bf16")
    rms = np.sqrt((y[:, :128].astype(np.float64) ** 2).mean(axis=1, keepdims=True) + 1e-6)
    assert np.allclose(out[:, :128], y[:, :128] / rms, rtol=2 ** -7)
    assert np.array_equal(out[:, 256:], y[:, 256:])  # V untouched


def test_lora_scales_and_rank16():
    L = O.make_svdq_layer(128, 128, 16, seed=4)
    x = O.make_activations(8, 128, seed=4)
    q, asc, la = O.quantize_w4a4_act_fuse_lora(x, L["smooth"], L["proj_down"])
    a = O.gemm_w4a4(q, asc, L["qweight"], L["wscales"], lora_act_in=la, lora_up=L["proj_up"], lora_scales=[0.0])["out"]
    b = O.gemm_w4a4(q, asc, L["qweight"], L["wscales"])["out"]
    assert np.array_equal(a, b)


def test_fused_mlp_tracks_dense():
    fc1 = O.make_svdq_layer(128, 256, 32, seed=1)
    fc2 = O.make_svdq_layer(256, 128, 32, seed=2)
    x = O.bf16_round(np.random.default_rng(7).standard_normal((32, 128)).astype(np.float32))
    o = O.fused_gelu_mlp(x, fc1, fc2)
    h = O.gelu_tanh(x @ fc1["dense"].T + fc1["bias"])
    ref = h @ fc2["dense"].T + fc2["bias"]
    assert np.linalg.norm(o - ref) / np.linalg.norm(ref) < 0.5  # two stacked 4-bit layers, random weights


def test_quantize_fuse_glu_is_the_plain_quantiser_on_the_gated_input():
    """fuse_glu only changes what is read (gemm_base.cuh:606-633): value * silu(gate), two 16-bit roundings."""
    rng = np.random.default_rng(3)
    M, K = 40, 128
    L = O.make_svdq_layer(K, 128, 16, seed=2, cheap=True)
    x2 = O.round16(rng.standard_normal((M, 2 * K)).astype(np.float32), "bf16")
    g = O.glu_pairs(x2, "bf16")
    assert g.shape == (M, K)
    v, gate = x2[:, 0::2].astype(np.float64), x2[:, 1::2].astype(np.float64)
    exact = v * gate / (1.0 + np.exp(-gate))
    assert np.all(np.abs(g - exact) <= 2.0 ** -7 * np.abs(exact) + 1e-30)  # two roundings of 2^-9 each
    assert np.array_equal(O.round16(g, "bf16"), g)
    a = O.quantize_w4a4_act_fuse_lora(x2, L["smooth"], L["proj_down"], "bf16", fuse_glu=True)
    b = O.quantize_w4a4_act_fuse_lora(g, L["smooth"], L["proj_down"], "bf16")
    assert all(np.array_equal(p, q) for p, q in zip(a, b))


# ----------------------------------------------------------------------------- approximation envelope (round 5)
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_quantize_envelope_holds_the_ieee_oracle_and_every_bounded_error_divider(dtype):
    """oracle.quantize_envelope: the IEEE quotient, a reciprocal-multiply with a 1-ulp reciprocal (this library's v_rcp_f32 form) and a quotient
    pushed to __fdividef's documented 2 ulp in either direction all produce codes and scales inside [lo, hi]; the envelope is narrow (open on
    ~1e-3 of the codes) and a 16-bit step outside it is detected."""
    K = 384
    L = O.make_svdq_layer(K, 128, 32, seed=3, dtype=dtype, cheap=True)
    x = O.make_activations(300, K, seed=3, dtype=dtype)
    env = O.quantize_envelope(x, L["smooth"], dtype)
    q, a, _ = O.quantize_w4a4_act_fuse_lora(x, L["smooth"], None, dtype)
    rep = O.envelope_report(q, env, q)
    assert rep["outside"] == 0.0 and rep["envelope_open"] < 3e-3
    assert np.all(a >= env["s_lo"]) and np.all(a <= env["s_hi"])
    xp = np.zeros((q.shape[0], K), np.float32)
    xp[:300] = x
    rng = np.random.default_rng(0)
    exact = xp.astype(np.float64) / L["smooth"].astype(np.float64)[None, :]
    variants = {
        "x * rcp (1-ulp reciprocal, rounded product)": (xp * (np.float32(1) / L["smooth"]).astype(np.float32)[None, :] *
                                                          (1 + rng.choice([-1, 0, 1], size=K) * 2.0 ** -24).astype(np.float32)[None, :]).astype(np.float32),
        "+2 ulp": (exact * (1 + 2 * 2.0 ** -24)).astype(np.float32),
        "-2 ulp": (exact * (1 - 2 * 2.0 ** -24)).astype(np.float32),
    }
    for name, quot in variants.items():
        qv, av = O.quantize_rows(O.round16(quot, dtype), dtype, unsigned=False)
        rep = O.envelope_report(qv, env, q)
        assert rep["outside"] == 0.0, (name, rep)
        assert rep["flips_vs_ieee"] < 1e-3 and rep["max_abs_diff_vs_ieee"] <= 1, (name, rep)
        assert np.all(av >= env["s_lo"]) and np.all(av <= env["s_hi"]), name
    # a quotient that is one 16-bit step off on every element is NOT inside
    off = O.round16(exact.astype(np.float32), dtype)
    off = np.nextafter(off.astype(np.float16), np.float16(np.inf)).astype(np.float32) if dtype == "fp16" else \
        (off.view(np.uint32) + np.uint32(0x10000)).view(np.float32)
    qo, _ = O.quantize_rows(off, dtype, unsigned=False)
    assert O.envelope_report(qo, env, q)["outside"] > 1e-3


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_gelu_quant_envelope(dtype):
    """the GELU -> requantise chain on intervals: the IEEE oracle and the MI355X epilogue's form of the GELU (x * rcp(1 + exp2(.)), emulated in
    float32) both land inside; the envelope says how far the reference's tanh.approx may move the codes: open on < 1 % of them"""
    K, N = 256, 384
    L1 = O.make_svdq_layer(K, N, 32, seed=5, dtype=dtype, cheap=True)
    L2 = O.make_svdq_layer(N, 128, 32, seed=6, dtype=dtype, cheap=True)
    x = O.make_activations(200, K, seed=5, dtype=dtype)
    q, a, la = O.quantize_w4a4_act_fuse_lora(x, L1["smooth"], L1["proj_down"], dtype)
    r = O.gemm_w4a4(q, a, L1["qweight"], L1["wscales"], dtype=dtype, bias=L1["bias"], lora_act_in=la, lora_up=L1["proj_up"], fuse="gelu_quant",
                    next_smooth=L2["smooth"], next_lora_down=L2["proj_down"], envelope=True)
    env = r["envelope"]
    rep = O.envelope_report(r["qout"], env, r["qout"])
    assert rep["outside"] == 0.0 and rep["envelope_open"] < 1e-2, rep
    # the kernel's GELU form in float32 on the same 16-bit pre-activation
    y16 = O.gemm_w4a4(q, a, L1["qweight"], L1["wscales"], dtype=dtype, bias=L1["bias"], lora_act_in=la, lora_up=L1["proj_up"])["out"]
    f = np.float32
    A = f(-2.0 * 0.79788456 * 1.4426950408889634)
    B = f(A * f(0.044715))
    t = (y16 * (y16 * y16 * B + A).astype(f)).astype(f)
    g = (y16 * (f(1) / (f(1) + np.exp2(t).astype(f))).astype(f)).astype(f)
    g16 = O.round16(g, dtype)
    assert np.all(g16 >= r["g16_lo"]) and np.all(g16 <= r["g16_hi"])


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_silu_and_rmsnorm_rope_envelopes(dtype):
    """the remaining PTX approximations on the path (ex2.approx + rcp.approx in SiLU, rsqrt.approx in the RMSNorm coefficient): the IEEE oracle lies inside
    their envelopes, and the envelopes say how little of the 16-bit output these approximations can move: open on < 2 % of the elements, never by more
    than one 16-bit step"""
    rng = np.random.default_rng(7)
    x = O.round16(rng.standard_normal((64, 256)).astype(np.float32) * 3, dtype)
    lo, hi = O.silu_envelope(x)
    s16, lo16, hi16 = O.round16(O.silu(x), dtype), O.round16(lo, dtype), O.round16(hi, dtype)
    assert np.all(s16 >= lo16) and np.all(s16 <= hi16)
    assert (lo16 != hi16).mean() < 0.02
    # RMSNorm + RoPE
    M, H = 96, 2
    y16 = O.round16(rng.standard_normal((M, 3 * H * 128)).astype(np.float32) * 2, dtype)
    nq = O.round16((1 + 0.1 * rng.standard_normal(128)).astype(np.float32), dtype)
    nk = O.round16((1 + 0.1 * rng.standard_normal(128)).astype(np.float32), dtype)
    ang = rng.uniform(0, 6.28, (M, 64)).astype(np.float32)
    rot = np.stack([np.sin(ang), np.cos(ang)], axis=-1).astype(np.float32)
    ref = O.rmsnorm_rope(y16, nq, nk, rot, dtype)
    lo, hi = O.rmsnorm_rope_envelope(y16, nq, nk, rot, dtype)
    assert np.all(ref >= lo) and np.all(ref <= hi)
    qk = slice(0, 2 * H * 128)
    step = np.abs(ref[:, qk]) * (2.0 ** -7 if dtype == "bf16" else 2.0 ** -10) + 1e-30
    assert (lo[:, qk] != hi[:, qk]).mean() < 0.02 and np.all(hi[:, qk] - lo[:, qk] <= 1.01 * step + 1e-6)
    assert np.array_equal(lo[:, 2 * H * 128:], hi[:, 2 * H * 128:])  # V: untouched
