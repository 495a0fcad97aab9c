"""Two properties of the round-3 GEMM that the oracle parity tests do not state:

* the workgroup geometries and schedules (256 x 128 tiles / one workgroup per CU; 128 x 128 tiles / two per CU with tiles drawn
  from per-XCD queues or from fixed lists) compute every output element with the same operations in the same order --
  launches whose tiles are not split along K are BIT-IDENTICAL across them, for every epilogue, bf16 and fp16;
* deterministic mode (nunchaku_amd.mode): the low-rank activations are accumulated as Q31.32 fixed point with integer
  atomics, so the K-sliced quantiser, the GELU epilogue's column-tile sum and the attention epilogue's head sum are
  bit-reproducible from run to run, and agree with the fp32 format to fp32 accuracy.  (The reference is non-deterministic
  here: fp32 atomics, lora.cuh:82-94,323.)
"""

import numpy as np
import pytest
import torch

from oracle import svdq_oracle as O
from tests.helpers import TORCH_DT, assert_close_16, f32, make_module, t16

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(built_lib):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from nunchaku_amd import _lib

    _lib.load()


def _with_geometry(g, fn):
    from nunchaku_amd._C import _Ops

    _Ops.gemm_geometry = g
    try:
        out = fn()
        torch.cuda.synchronize()
        return out
    finally:
        _Ops.gemm_geometry = 0


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("M,N", [(300, 768), (1536, 768), (4608, 3072)], ids=["M300", "M1536", "M4608-queued"])
def test_geometries_are_bit_identical_without_k_split(dtype, M, N):
    """K = 384 / N: no stream-K split in any geometry (too few K-steps), so every epilogue must agree bit for bit: 256 x 128
    tiles (1), 128 x 128 tiles drawn from the per-XCD queues (2; at M = 4608 there are 864 tiles for 512 workgroups, so the
    queues are really used -- the smaller cases fall back to fixed lists), fixed lists (3), and both with the phase offset
    (4, 5).  Which workgroup computes a tile must not matter."""
    from nunchaku_amd import layout
    from nunchaku_amd.ops.fused import fused_gelu_mlp, fused_qkv_norm_rottary

    K = 384
    L = O.make_svdq_layer(K, N, 32, seed=1, dtype=dtype, cheap=True)
    L2 = O.make_svdq_layer(N, K, 32, seed=2, dtype=dtype, cheap=True)
    mod, mod2 = make_module(L, dtype), make_module(L2, dtype, act_unsigned=True)
    x = t16(O.make_activations(M, K, seed=3, dtype=dtype), dtype).view(1, M, K)
    nq = torch.nn.RMSNorm(128, eps=1e-6, dtype=TORCH_DT[dtype], device="cuda")
    nk = torch.nn.RMSNorm(128, eps=1e-6, dtype=TORCH_DT[dtype], device="cuda")
    with torch.no_grad():
        nq.weight.copy_(torch.rand(128) + 0.5)
        nk.weight.copy_(torch.rand(128) + 0.5)
    M_pad = (M + 255) // 256 * 256
    ang = np.random.default_rng(0).uniform(0, 6.28, (M_pad, 64)).astype(np.float32)
    rot = torch.from_numpy(O.pack_rotemb_ref(np.stack([np.sin(ang), np.cos(ang)], -1).astype(np.float32))).cuda().view(1, M_pad, 128)
    Lq = O.make_svdq_layer(K, N, 32, seed=4, dtype=dtype, cheap=True)  # N = 3 * heads * 128
    modq = make_module(Lq, dtype)

    def run():
        plain = mod(x)
        silu_out = torch.empty(M, N, dtype=TORCH_DT[dtype], device="cuda")
        from nunchaku_amd.ops.gemm import svdq_gemm_w4a4_cuda

        qx, asc, la = mod.quantize(x.view(M, K))
        svdq_gemm_w4a4_cuda(act=qx, wgt=mod.qweight, out=silu_out, ascales=asc, wscales=mod.wscales, lora_act_in=la, lora_up=mod.proj_up,
                            bias=mod.bias, fuse_silu=True)
        mlp = fused_gelu_mlp(x, mod, mod2)
        vt = torch.zeros(N // 3, M_pad, dtype=TORCH_DT[dtype], device="cuda")
        qkv = fused_qkv_norm_rottary(x, modq, nq, nk, rot, out_vt=vt[:, :M])
        return plain, silu_out, mlp, qkv[..., : 2 * N // 3].clone(), vt

    from nunchaku_amd import mode

    with mode.deterministic_mode():  # (fp32 atomics in the low-rank sums would add run-to-run noise of their own)
        ref = _with_geometry(1, run)
        got23 = {g: _with_geometry(g, run) for g in (2, 3, 4, 5)}
    for g in (2, 3, 4, 5):
        got = got23[g]
        for name, a, b in zip(("default", "silu", "gelu_mlp", "qk rope", "v^T"), ref, got):
            assert torch.equal(a, b), f"geometry {g} vs 1, {name}: {(a != b).float().mean():.2e} of the elements differ"


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_geometry2_with_k_split_matches_oracle(dtype):
    """K = 12288, M = 1024: the 128 x 128 geometry splits its remainder tiles along K at different points than the 256 x 128
    one (fp32 summation order of a split tile differs): both within 1 ulp of the oracle, and of each other."""
    K, N, M = 12288, 3072, 1024
    L = O.make_random_svdq_layer(K, N, 32, seed=5, dtype=dtype)
    mod = make_module(L, dtype)
    x = O.make_activations(M, K, seed=6, dtype=dtype)
    xt = t16(x, dtype).view(1, M, K)
    rows = np.array(sorted(set([0, 1, 127, 128, 129, 255, 256, 511, 512, 767, 1023]) | set(np.random.default_rng(0).integers(0, M, 40).tolist())))
    ref = O.svdq_linear(x[rows], L, dtype, "fp32")["out"]
    outs = {}
    qx, asc, la = mod.quantize(xt.view(M, K))  # one quantiser run for both (its K-sliced low-rank sum uses fp32 atomics)
    for g in (1, 2, 3):
        outs[g] = _with_geometry(g, lambda: mod.forward_quant(qx, asc, la)[None])
        # lora_act comes from the GPU quantiser (fp32 order differs from the oracle's float64): 1 ulp + rare flips
        assert_close_16(f32(outs[g])[0][rows], ref, dtype, f"geometry {g}", max_bad_frac=2e-3, ulps=1.0)
    assert_close_16(f32(outs[2]), f32(outs[1]), dtype, "geometry 2 vs 1", ulps=1.0)
    assert torch.equal(outs[2], outs[3]), "long K: geometry 2 takes the fixed lists + stream-K of geometry 3"
    from nunchaku_amd._C import ops

    ops.gemm_workspace_status()


def test_deterministic_quantiser_and_fused_mlp():
    """K-sliced quantiser + GELU epilogue in deterministic mode: int64 lora_act, identical over repeated launches, equal to
    the fp32-format result to fp32 accuracy, outputs within the usual bounds of the oracle."""
    from nunchaku_amd import mode
    from nunchaku_amd.ops.fused import fused_gelu_mlp

    dtype, M, K, N = "bf16", 1024, 3072, 12288
    L1 = O.make_random_svdq_layer(K, N, 32, seed=7, dtype=dtype)
    L2 = O.make_random_svdq_layer(N, K, 32, seed=8, dtype=dtype)
    fc1, fc2 = make_module(L1, dtype), make_module(L2, dtype, act_unsigned=True)
    x = t16(O.make_activations(M, K, seed=9, dtype=dtype), dtype).view(1, M, K)

    _, _, la32 = fc1.quantize(x.view(M, K))
    y32 = fused_gelu_mlp(x, fc1, fc2)
    with mode.deterministic_mode():
        runs = []
        for _ in range(4):
            _, _, la = fc1.quantize(x.view(M, K))
            assert la.dtype == torch.int64
            runs.append((la.clone(), fused_gelu_mlp(x, fc1, fc2)))
        torch.cuda.synchronize()
    for la, y in runs[1:]:
        assert torch.equal(la, runs[0][0]), "fixed-point lora_act differs between two launches"
        assert torch.equal(y, runs[0][1]), "deterministic fused MLP differs between two launches"
    la_det = mode.lora_act_to_float(runs[0][0])
    # fp32 atomics vs exact integer accumulation of the same partial sums: fp32 rounding of a handful of adds
    assert torch.allclose(la_det, la32, rtol=0, atol=4e-6 * float(la32.abs().max()) + 1e-7)
    # the two formats feed the same GEMMs: 16-bit outputs agree up to rare 1-ulp flips (and 4-bit code flips behind them)
    d = (runs[0][1].float() - y32.float()).abs()
    assert float(d.max()) <= 0.05 * float(y32.float().abs().max())
    assert float((d > 0).float().mean()) < 0.2


def test_deterministic_runs_level_row_run_carry():
    """ABI 22, ``mode.deterministic_mode("runs")`` (SVDQ_LORA_ACT_Q32_RUNS): at the FLUX fc1 shape the GELU_QUANT launch takes the row-run schedule with its
    fp32 carry (plan variant "carry": no atomics inside a tile), the run's sum enters the int64 accumulators once -- launches are bit-equal to each other, the
    sums agree with the strict level's to fp32 accuracy; a launch too small for row runs takes the strict path and is bit-equal to it."""
    from nunchaku_amd import mode
    from nunchaku_amd.ops.gemm import svdq_gemm_w4a4_cuda
    from nunchaku_amd.mode import alloc_lora_act
    from nunchaku_amd._C import ops

    dtype, K, N = "bf16", 3072, 12288
    L1 = O.make_random_svdq_layer(K, N, 32, seed=17, dtype=dtype)
    L2 = O.make_random_svdq_layer(N, K, 32, seed=18, dtype=dtype)
    fc1, fc2 = make_module(L1, dtype), make_module(L2, dtype, act_unsigned=True)

    def fc1_launch(x, M):
        qx, asc, la = fc1.quantize(x.view(M, K))
        M_pad = (M + 255) // 256 * 256
        qh = torch.empty(M_pad, N * 3 // 4, dtype=torch.uint8, device="cuda")
        sh = torch.empty(N // 64, M_pad, dtype=x.dtype, device="cuda")
        lh, zeroed = alloc_lora_act(M_pad, 32, "cuda")
        fc1._ensure_layout(); fc2._ensure_layout()
        svdq_gemm_w4a4_cuda(act=qx, wgt=fc1.qweight, qout=qh, ascales=asc, wscales=fc1.wscales, oscales=sh, lora_act_in=la, lora_up=fc1.proj_up,
                            lora_down=fc2.proj_down, lora_act_out=lh, bias=fc1.bias, smooth_factor=fc2.smooth_factor, lora_act_zeroed=zeroed)
        return qh, sh, lh, ops.gemm_last_plan()

    for M, carried in ((4608, True), (1024, False)):
        x = t16(O.make_activations(M, K, seed=19, dtype=dtype), dtype)
        with mode.deterministic_mode("strict"):
            q0, s0, l0, plan0 = fc1_launch(x, M)
            assert l0.dtype == torch.int64 and plan0["variant"] == "plain", plan0
        with mode.deterministic_mode("runs"):
            got = [fc1_launch(x, M) for _ in range(3)]
        torch.cuda.synchronize()
        assert got[0][3]["variant"] == ("carry" if carried else "plain"), got[0][3]
        assert (got[0][3]["rowrun"] > 0) == carried, got[0][3]
        for q, s_, l, _ in got[1:]:
            assert torch.equal(l, got[0][2]) and torch.equal(q, got[0][0]) and torch.equal(s_, got[0][1]), "runs level: two launches differ"
        assert torch.equal(got[0][0], q0) and torch.equal(got[0][1], s0), "codes and scales do not depend on the accumulator format"
        if carried:  # fp32 sums of a run of <= 24 tiles against the exact integer sum of the same partials
            a, b = mode.lora_act_to_float(got[0][2]), mode.lora_act_to_float(l0)
            assert torch.allclose(a, b, rtol=0, atol=8e-6 * float(b.abs().max()) + 1e-7)
            assert not torch.equal(got[0][2], l0), "expected the carried sums to differ from the strict ones in the last bits (else the carry did not run)"
        else:
            assert torch.equal(got[0][2], l0), "no row runs: the runs level IS the strict path"


def test_deterministic_runs_level_grouped_mlp_of_a_joint_block():
    """The launch a FLUX joint block really issues -- text (512 rows) and image (4096 rows) MLPs grouped into one GELU_QUANT launch and one fc2 launch -- at the
    level "runs": the 16-bit outputs of repeated calls are bit-equal (the row runs of a grouped launch are a function of the shapes alone), and they agree with
    the strict level's up to the 4-bit code flips a last-bit difference of the low-rank sums can cause."""
    from nunchaku_amd import mode
    from nunchaku_amd.ops.fused import fused_gelu_mlp_pair

    dtype, K, N, Ma, Mb = "bf16", 3072, 12288, 512, 4096
    mods = []
    for i in range(2):
        L1 = O.make_random_svdq_layer(K, N, 32, seed=31 + 2 * i, dtype=dtype)
        L2 = O.make_random_svdq_layer(N, K, 32, seed=32 + 2 * i, dtype=dtype)
        mods.append((make_module(L1, dtype), make_module(L2, dtype, act_unsigned=True)))
    xa = t16(O.make_activations(Ma, K, seed=41, dtype=dtype), dtype).view(1, Ma, K)
    xb = t16(O.make_activations(Mb, K, seed=42, dtype=dtype), dtype).view(1, Mb, K)
    call = lambda: fused_gelu_mlp_pair(xa, mods[0][0], mods[0][1], xb, mods[1][0], mods[1][1])
    with mode.deterministic_mode("strict"):
        ya0, yb0 = (t.clone() for t in call())
    with mode.deterministic_mode("runs"):
        outs = [tuple(t.clone() for t in call()) for _ in range(3)]
    torch.cuda.synchronize()
    for ya, yb in outs[1:]:
        assert torch.equal(ya, outs[0][0]) and torch.equal(yb, outs[0][1]), "runs level: two grouped MLP calls differ"
    for got, ref in ((outs[0][0], ya0), (outs[0][1], yb0)):
        d = (got.float() - ref.float()).abs()
        assert float(d.max()) <= 0.05 * float(ref.float().abs().max()) and float((d > 0).float().mean()) < 0.05


def test_deterministic_model_forward_is_bit_reproducible():
    """A FLUX-shaped step (fused norms, grouped launches, attention-side quantiser, stream-K GEMMs, persistent attention):
    bit-equal outputs over repeated forwards in deterministic mode."""
    from nunchaku_amd import mode
    from nunchaku_amd.models.flux import FluxTransformerAMD

    torch.manual_seed(0)
    model = FluxTransformerAMD(num_layers=2, num_single_layers=2, dim=512, heads=4, in_channels=64, joint_attention_dim=256,
                               pooled_projection_dim=64, device="cuda").init_synthetic_(seed=0).eval()
    side, t_txt = 32, 256
    g = torch.Generator(device="cuda").manual_seed(0)
    lat = torch.randn(1, side * side, 64, device="cuda", generator=g).bfloat16()
    enc = torch.randn(1, t_txt, 256, device="cuda", generator=g).bfloat16()
    pooled = torch.randn(1, 64, device="cuda", generator=g).bfloat16()
    img_ids = torch.zeros(side * side, 3, device="cuda")
    img_ids[:, 1] = torch.arange(side, device="cuda").repeat_interleave(side)
    img_ids[:, 2] = torch.arange(side, device="cuda").repeat(side)
    args = (lat, enc, pooled, torch.tensor([0.5], device="cuda"), img_ids, torch.zeros(t_txt, 3, device="cuda"), torch.tensor([3.5], device="cuda"))
    with torch.no_grad(), mode.deterministic_mode():
        outs = [model(*args).clone() for _ in range(6)]
    torch.cuda.synchronize()
    assert bool(torch.isfinite(outs[0].float()).all())
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), "deterministic mode: two forwards of one model differ"
    with torch.no_grad():
        fast = model(*args)
    rel = float((fast.float() - outs[0].float()).norm() / outs[0].float().norm())
    assert rel < 0.05, f"deterministic vs fp32-atomics forward differ by {rel:.3e} (W4A4 code-flip level expected: <= 2e-2)"


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("M,K,N,R,bias,split", [(300, 384, 256, 32, True, 0), (512, 128, 128, 0, False, 0), (1000, 640, 384, 16, True, 0), (4608, 3072, 3072, 32, True, 0),
                                                (1536, 3072, 3072, 32, True, 512), (4608, 1152, 3072, 48, True, 0)],
                         ids=["small", "one-k-step-no-lora", "rank16-odd-M-falls-back", "out-projection", "grouped", "rank48-falls-back"])
def test_wave_tile_128_kernel_is_bit_identical_to_the_8_wave_kernel(dtype, M, K, N, R, bias, split):
    """round 6: the 128 x 64-per-wave / one-wave-per-SIMD kernel (geometry 8; the library's own choice for the plain epilogue at rank <= 32 on 256 x 128 tiles)
    computes every output with the operations of the 8-wave kernel in the same order: launches that are not split along K agree BIT FOR BIT (bias, low-rank
    up projection at rank 0 / 32, M tails, one K-step, a grouped launch with two weight sets), and the plan says which kernel ran."""
    from nunchaku_amd._C import ops
    from nunchaku_amd.ops.gemm import svdq_gemm_w4a4_cuda

    L = O.make_svdq_layer(K, N, max(R, 16), seed=11, dtype=dtype, cheap=True)   # (R = 0: a rank-16 layer launched without its low-rank branch)
    Lb = O.make_svdq_layer(K, N, max(R, 16), seed=12, dtype=dtype, cheap=True)
    mod, modb = make_module(L, dtype), make_module(Lb, dtype)
    for m_ in (mod, modb):
        m_._ensure_layout()
    x = t16(O.make_activations(M, K, seed=13, dtype=dtype), dtype)
    qx, asc, la = mod.quantize(x)

    def run():
        out = torch.empty(M, N, dtype=TORCH_DT[dtype], device="cuda")
        second = dict(wgt=modb.qweight, wscales=modb.wscales, bias=modb.bias if bias else None, lora_up=modb.proj_up if R else None) if split else None
        svdq_gemm_w4a4_cuda(act=qx, wgt=mod.qweight, out=out, ascales=asc, wscales=mod.wscales, lora_act_in=la if R else None, lora_up=mod.proj_up if R else None,
                            bias=mod.bias if bias else None, second=second, split_rows=split)
        return out, ops.gemm_last_plan()

    ref, plan1 = _with_geometry(1, run)
    got, plan8 = _with_geometry(8, run)
    auto, plan0 = _with_geometry(0, run)
    assert plan1["variant"] != "wave_tile_128"
    assert plan8["variant"] == ("wave_tile_128" if R in (0, 32) else plan1["variant"]), plan8   # (its generated epilogue takes rank 32 or none)
    # the library's own choice takes the wave-tile kernel from K = 8192 (fc2); below, the 8-wave kernel is the faster one since the MFMA changes of round 6
    assert plan0["variant"] == ("wave_tile_128" if (R in (0, 32) and K >= 8192 and plan1["tile_rows"] == 256) else plan1["variant"]), plan0
    assert plan8["streamk_groups"] == plan1["streamk_groups"] == 0
    assert torch.equal(got, ref), f"geometry 8 vs 1: {(got != ref).float().mean():.2e} of the elements differ"
    assert torch.equal(auto, ref)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_wave_tile_128_kernel_with_k_split_matches_oracle(dtype):
    """K = 12288 (fc2), M = 1024 and 4608: the 128 x 64 wave tile kernel under the stream-K split (its own publish / collect lane map): within 1 ulp of the oracle
    and of the 8-wave kernel, repeatable, no owner timed out."""
    from nunchaku_amd._C import ops

    K, N = 12288, 3072
    L = O.make_random_svdq_layer(K, N, 32, seed=5, dtype=dtype)
    mod = make_module(L, dtype, act_unsigned=False)
    for M in (1024, 4608):
        x = O.make_activations(M, K, seed=6, dtype=dtype)
        xt = t16(x, dtype)
        rows = np.array(sorted(set([0, 1, 127, 128, 255, 256, 511, 512, M - 1]) | set(np.random.default_rng(0).integers(0, M, 40).tolist())))
        ref = O.svdq_linear(x[rows], L, dtype, "fp32")["out"]
        qx, asc, la = mod.quantize(xt)
        outs, plans = {}, {}
        for g in (1, 8):
            outs[g] = _with_geometry(g, lambda: mod.forward_quant(qx, asc, la))
            plans[g] = ops.gemm_last_plan()
            assert_close_16(f32(outs[g])[rows], ref, dtype, f"geometry {g} M={M}", max_bad_frac=2e-3, ulps=1.0)
        assert plans[8]["variant"] == "wave_tile_128" and plans[8]["streamk_groups"] > 0 and plans[8]["streamk_groups"] == plans[1]["streamk_groups"], plans
        _with_geometry(0, lambda: mod.forward_quant(qx, asc, la))
        assert ops.gemm_last_plan()["variant"] == "wave_tile_128", "K = 12288: the library's own choice is the wave-tile kernel"
        assert_close_16(f32(outs[8]), f32(outs[1]), dtype, "geometry 8 vs 1", ulps=1.0)
        assert torch.equal(outs[8], _with_geometry(8, lambda: mod.forward_quant(qx, asc, la))), "two launches of the split schedule differ"
        ops.gemm_workspace_status()
