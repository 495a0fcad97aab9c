"""Oracle parity at the shapes bench.py actually runs (FLUX.1 and Qwen-Image alike: hidden 3072, MLP 12288, QKV 9216; M = 512 /
4096 / 4608 for the 1024^2 configs, 1024 / 1536 / 512+1024 for FLUX.1-schnell 512^2 -- BASELINE config 2),
row-sampled so the numpy oracle finishes in seconds: every epilogue, bf16 and fp16, single and grouped launches, and --
the hole VERDICT r1 named -- the stream-K split-K tail that every K = 12288 launch of a step takes.

Per case the chain of one transformer block's projections runs on the GPU through the C ABI:
    QKV      3072 ->  9216   RMSNorm(q, k) + RoPE epilogue, V written transposed (out_vt)
    out      3072 ->  3072   default epilogue
    fc1      3072 -> 12288   GELU -> unsigned 4-bit requantisation (+ fc2's low-rank down projection)
    fc2     12288 ->  3072   default epilogue on the GPU's own codes (K = 12288: stream-K)
Each GEMM runs TWICE -- with the per-stream workspace (stream-K allowed) and with workspace = NULL (whole tiles only) --
both are held to the oracle (1 ulp of the 16-bit type; the epilogue-specific bounds of tests/test_gpu_parity.py) on
>= 64 sampled rows that include the tile / stream boundaries, and to each other (fp32 add order only).
Tolerances are written at each assert.  The oracle's arithmetic is parity-unpinned."""

import ctypes as C
import functools

import numpy as np
import pytest
import torch

from oracle import svdq_oracle as O
from tests.helpers import TORCH_DT, assert_close_16, checked_codes, f32, make_module, t16

pytestmark = pytest.mark.gpu

HID, MLP, QKV, R = 3072, 12288, 9216, 32


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(built_lib):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from nunchaku_amd import _lib

    _lib.load()


@pytest.fixture(params=[1, 2, 0], ids=["tile256x128", "tile128x128-queued", "auto"], autouse=True)
def _geometry(request):
    """Every test of this module runs under both workgroup geometries of the GEMM (svdq_gemm_args.geometry) and under the library's own choice
    (0: what the step runs -- beyond next-layer rank 32 the GELU_QUANT launch then takes the solo-carry kernel)."""
    from nunchaku_amd._C import _Ops

    _Ops.gemm_geometry = request.param
    yield request.param
    _Ops.gemm_geometry = 0


LORA_STRENGTH = 0.75


@functools.lru_cache(maxsize=None)
def _layer(K, N, seed, dtype, rank=R, lora=0):
    """oracle layer; ``lora`` > 0: the layer a runtime LoRA of that rank turns it into (set_lora: the low-rank branch widened by ceil16(lora) ranks,
    the new ranks scaled by LORA_STRENGTH -- the reference's update_lora_params / set_lora_strength, transformer_flux.py:783-855)"""
    L = O.make_random_svdq_layer(K, N, rank, seed=seed, dtype=dtype)
    if lora:
        rng = np.random.default_rng(1000 + seed)
        rp = (lora + 15) // 16 * 16
        down = np.zeros((rp, K), np.float32)
        up = np.zeros((N, rp), np.float32)
        down[:lora] = O.round16(rng.standard_normal((lora, K)).astype(np.float32) / np.sqrt(K), dtype)
        up[:, :lora] = O.round16(rng.standard_normal((N, lora)).astype(np.float32) * 0.5, dtype)
        L = dict(L)
        L["lora_down_user"], L["lora_up_user"] = down[:lora], up[:, :lora]
        L["proj_down"] = np.concatenate([L["proj_down"], down.T], axis=1)
        L["proj_up"] = np.concatenate([L["proj_up"], up], axis=1)
        L["lora_scales"] = [1.0] * (rank // 16) + [LORA_STRENGTH] * (rp // 16)
    return L


@functools.lru_cache(maxsize=None)
def _module(K, N, seed, dtype, unsigned=False, rank=R, lora=0):
    m = make_module(_layer(K, N, seed, dtype, rank), dtype, act_unsigned=unsigned)
    m._ensure_layout()
    if lora:
        L = _layer(K, N, seed, dtype, rank, lora)
        m.set_lora(t16(L["lora_down_user"], dtype), t16(L["lora_up_user"], dtype), strength=LORA_STRENGTH)
        assert m.rank == L["proj_up"].shape[1] and m.lora_scales == L["lora_scales"]
    return m


def _sample_rows(M, split=0, n=64, seed=0):
    """>= 64 rows: first/last row of every 256-row tile boundary class, the stream boundary, then seeded random ones."""
    must = [0, 1, 31, 32, 255, 256, 257, 511, 512, 513, 1023, 2047, 2048, 4095, 4096, 4097, 4351, 4352, 4607]
    if split:
        must += [split - 1, split, split + 1]
    rows = sorted({r for r in must if 0 <= r < M})
    rng = np.random.default_rng(seed + M)
    extra = [int(r) for r in rng.permutation(M) if r not in rows][: max(0, n - len(rows))]
    return np.array(sorted(rows + extra))


def _streamk_on(M_pad, N, K):
    from nunchaku_amd import _lib
    from nunchaku_amd._C import _Ops

    lib = _lib.load()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    cap = 16384
    buf = (C.c_int32 * (6 * cap))()
    n = lib.svdq_gemm_schedule_ex(M_pad, N, K, cus, 1, 1 if (_Ops.gemm_geometry or 1) == 1 else 2, buf, cap)
    assert 0 < n <= cap
    return any(buf[6 * i + 4] >= 0 for i in range(n))  # a segment that publishes a partial tile


class _Both:
    """Run a launch closure with the stream-K workspace and with workspace = NULL; yields (tag, outputs)."""

    def __init__(self, fn):
        self.fn = fn

    def __iter__(self):
        from nunchaku_amd._C import _Ops

        for tag, use in (("ws", True), ("nows", False)):
            _Ops.gemm_use_workspace = use
            try:
                res = self.fn()
                torch.cuda.synchronize()
            finally:
                _Ops.gemm_use_workspace = True
            yield tag, res
        from nunchaku_amd._C import ops

        ops.gemm_workspace_status()  # no owner timed out


def _check_plan(which, tag, R, R2, grouped, M_pad, N):
    """the kernel the launch took (svdq_gemm_last_plan) is the one DESIGN.md section 5 documents for its rank / shape -- nothing fell back"""
    from nunchaku_amd._C import _Ops, ops

    plan = ops.gemm_last_plan()
    geo, ws = _Ops.gemm_geometry, tag == "ws"
    if which == "fc1":   # GELU_QUANT: the low-rank-down reduction decides
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        if R2 <= 32:
            want = "carry" if (geo in (0, 1) or not ws) and plan["tile_rows"] == 256 else None
        elif geo == 0 and ws and 32 < R2 <= 160 and 32 < R <= 160 and N % 256 == 0 and (M_pad // 128) * (N // 128) >= 2 * cus:
            want = "split_down"   # (ABI 20: the workspace ops.gemm_w4a4 asks for holds the launch's 16-bit output image)
        elif geo == 0 and ws and R2 >= 96 and R2 <= 128 and (M_pad // 128) * (N // 128) >= 2 * cus:
            want = "solo_carry"
        elif plan["tile_rows"] == 256 and (M_pad // 256) * (N // 128) >= 2 * cus and plan["streamk_groups"] == 0:
            want = "hybrid_carry"
        else:
            want = None
        if want:
            assert plan["variant"] == want, (which, tag, geo, R, R2, plan)
        if plan["variant"] == "solo_carry" and 32 < R <= 160 and not grouped and ws:  # (the packed fragments live in the workspace tail)
            assert plan["lora_act_packed"] and plan["lora_up_packed"], (tag, plan)
        if plan["variant"] == "split_down":
            assert plan["tile_rows"] == 256 and plan["lora_act_packed"] and not plan["lora_up_packed"], (tag, plan)
    elif 32 < R <= 160 and ws:   # the all-rank kernels (a workspace holds the packed fragments)
        if plan["tile_rows"] == 256 or not grouped:
            assert plan["variant"] == "all_rank" and plan["lora_act_packed"], (which, tag, geo, R, plan)
            assert plan["lora_up_packed"] == (plan["tile_rows"] == 128), plan
    elif R <= 32:
        # (round 6: the plain epilogue runs the 128 x 64 wave tile kernel from K = 8192 -- fc2; the out-projection, K = 3072, keeps the 8-wave kernel)
        assert plan["variant"] == "plain" and not plan["lora_act_packed"], (which, tag, geo, R, plan)


def _same_up_to_add_order(a, b, dtype, what, ulps=1.0):
    """two schedules of one GEMM differ by the fp32 summation order of the K slices only: <= 1 ulp, on a tiny fraction (the RMSNorm + RoPE epilogue sits
    behind a 16-bit rounding point: a flipped rounding there shows as up to 2 ulp in Q / K, the bound those outputs have against the oracle as well --
    and under the library's own geometry choice the two runs may take different workgroup geometries)"""
    a, b = f32(a), f32(b)
    diff = a != b
    assert diff.mean() < 2e-3, f"{what}: {diff.mean():.2e} of the elements differ between the two schedules"
    assert_close_16(a, b, dtype, what + " (ws vs nows)", ulps=ulps)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("Ma,Mb", [(512, 0), (1024, 0), (1536, 0), (4096, 0), (4608, 0), (512, 1024), (512, 4096)],
                         ids=["M512", "M1024", "M1536", "M4096", "M4608", "grouped512+1024", "grouped512+4096"])
def test_block_projections_at_flux_shapes(dtype, Ma, Mb):
    _block_projections(dtype, Ma, Mb)


@pytest.mark.parametrize("dtype,Ma,Mb,rank,lora", [("bf16", 4608, 0, 128, 0), ("fp16", 4608, 0, 128, 0), ("bf16", 256, 6144, 128, 0), ("bf16", 4608, 0, 32, 16),
                                                  ("bf16", 1536, 0, 64, 0)],
                         ids=["r128-M4608-bf16", "r128-M4608-fp16", "r128-qwen1664x928-grouped256+6144", "r32+lora16-M4608", "r64-M1536"])
def test_block_projections_at_other_ranks(dtype, Ma, Mb, rank, lora):
    """the reference's own Qwen-Image gate runs rank 32 AND rank 128 (tests/v1/qwenimage/test_qwenimage.py:20-26, 1664 x 928 = 6032 image
    tokens -> 6144 padded rows + a padded text stream), and every runtime LoRA makes the rank 32 + r (transformer_flux.py:783-855).  Same chain, same
    bounds as the rank-32 grid above; the launches run the all-rank kernels (lora_up of a tile staged for every rank, lora_act_in in a ring of 64-rank
    batches), the quantiser's multi-slab fast path and the multi-pass low-rank down projection of the GELU_QUANT epilogue."""
    _block_projections(dtype, Ma, Mb, rank=rank, lora=lora)


def _block_projections(dtype, Ma, Mb, rank=R, lora=0):
    from nunchaku_amd import layout
    from nunchaku_amd.ops import fused
    from nunchaku_amd.ops.gemm import svdq_gemm_w4a4_cuda

    td = TORCH_DT[dtype]
    M = Ma + Mb
    grouped = Mb > 0
    rows = _sample_rows(M, split=Ma if grouped else 0)
    sets = [0, 1] if grouped else [0]
    row_set = (rows >= Ma).astype(int) if grouped else np.zeros(len(rows), int)

    x = O.make_activations(M, HID, seed=7, dtype=dtype)
    xt = t16(x, dtype)

    def quantize(K, N, seed0, src, unsigned=False):
        """-> (act, asc, lact, modules, layers): one or two streams quantised into one set of row-side buffers"""
        mods = [_module(K, N, seed0 + s, dtype, unsigned, rank, lora) for s in sets]
        lays = [_layer(K, N, seed0 + s, dtype, rank, lora) for s in sets]
        if grouped:
            act, asc, lact, _ = fused._quantize_pair(src[:Ma].unsqueeze(0), mods[0], src[Ma:].unsqueeze(0), mods[1])
        else:
            act, asc, lact = mods[0].quantize(src)
        return act, asc, lact, mods, lays

    def second(mods, **extra):
        return dict(second=fused._second(mods[1], **extra), split_rows=Ma) if grouped else {}

    def oracle_rows(lays, fn):
        """fn(layer, row indices into `rows`) per stream -> outputs scattered back in sampled-row order"""
        outs = None
        for s in sets:
            idx = np.nonzero(row_set == s)[0]
            if len(idx) == 0:
                continue
            res = fn(lays[s], idx, s)
            if outs is None:
                outs = {k: np.zeros((len(rows),) + v.shape[1:], v.dtype) for k, v in res.items()}
            for k, v in res.items():
                outs[k][idx] = v
        return outs

    # ---------------------------------------------------------------- QKV: RMSNorm + RoPE, V transposed
    act, asc, lact, mods, lays = quantize(HID, QKV, 10, xt)
    M_pad = act.shape[0]
    rng = np.random.default_rng(12)
    nqk = [[O.round16((1 + 0.1 * rng.standard_normal(128)).astype(np.float32), dtype) for _ in range(2)] for _ in sets]
    ang = rng.uniform(0, 6.28, (M_pad, 64)).astype(np.float32)
    rot = np.stack([np.sin(ang), np.cos(ang)], axis=-1).astype(np.float32)
    packed = torch.from_numpy(O.pack_rotemb_ref(rot)).cuda().view(M_pad, 128)
    nq_t = [[t16(w, dtype) for w in pair] for pair in nqk]
    la_gpu = lact.cpu().numpy()

    def qkv_launch():
        out = torch.zeros(M, QKV, dtype=td, device="cuda")
        vt = torch.zeros(QKV // 3, M_pad, dtype=td, device="cuda")
        svdq_gemm_w4a4_cuda(act=act, wgt=mods[0].qweight, out=out, ascales=asc, wscales=mods[0].wscales, lora_act_in=lact,
                            lora_up=mods[0].proj_up, lora_scales=mods[0].lora_scales, bias=mods[0].bias, norm_q=nq_t[0][0], norm_k=nq_t[0][1], rotary_emb=packed,
                            out_vt=vt, **(second(mods, norm_q=nq_t[1][0], norm_k=nq_t[1][1]) if grouped else {}))
        return out, vt

    def qkv_ref(L, idx, s):
        r = rows[idx]
        q, a = checked_codes(act, asc, x, L["smooth"], dtype, rows=r)  # the operands the launch read: envelope + flip budget vs the IEEE oracle
        return {"out": O.gemm_w4a4(q, a, L["qweight"], L["wscales"], dtype=dtype, bias=L["bias"], lora_act_in=la_gpu[r],
                                   lora_up=L["proj_up"], lora_scales=L.get("lora_scales"), fuse="rmsnorm_rope", norm_q=nqk[s][0], norm_k=nqk[s][1], rot=rot[r])["out"]}

    ref = oracle_rows(lays, qkv_ref)["out"]
    got = {}
    for tag, (out, vt) in _Both(qkv_launch):
        got[tag] = out
        _check_plan("qkv", tag, rank + (lora + 15) // 16 * 16, 0, grouped, M_pad, QKV)
        g = f32(out)[rows]
        # Q/K: fp32 epilogue math behind a 16-bit rounding point -> 2 ulp, rare 1-ulp flips of the pre-norm value
        assert_close_16(g[:, : 2 * QKV // 3], ref[:, : 2 * QKV // 3], dtype, f"QK {tag}", max_bad_frac=2e-3, ulps=2.0)
        # V goes to out_vt (transposed), 1 ulp; the V columns of `out` stay untouched
        assert_close_16(f32(vt)[:, rows].T, ref[:, 2 * QKV // 3:], dtype, f"V^T {tag}")
        assert not out[:, 2 * QKV // 3:].any()
    _same_up_to_add_order(got["ws"], got["nows"], dtype, "QKV", ulps=2.0)

    # ---------------------------------------------------------------- attention output projection: default epilogue
    act, asc, lact, mods, lays = quantize(HID, HID, 20, xt)
    la_gpu = lact.cpu().numpy()

    def out_launch():
        out = torch.zeros(M, HID, dtype=td, device="cuda")
        svdq_gemm_w4a4_cuda(act=act, wgt=mods[0].qweight, out=out, ascales=asc, wscales=mods[0].wscales, lora_act_in=lact,
                            lora_up=mods[0].proj_up, lora_scales=mods[0].lora_scales, bias=mods[0].bias, **second(mods))
        return out

    def out_ref(L, idx, s):
        r = rows[idx]
        q, a = checked_codes(act, asc, x, L["smooth"], dtype, rows=r)  # the operands the launch read: envelope + flip budget vs the IEEE oracle
        return {"out": O.gemm_w4a4(q, a, L["qweight"], L["wscales"], dtype=dtype, bias=L["bias"], lora_act_in=la_gpu[r],
                                   lora_up=L["proj_up"], lora_scales=L.get("lora_scales"))["out"]}

    ref = oracle_rows(lays, out_ref)["out"]
    got = {}
    for tag, out in _Both(out_launch):
        got[tag] = out
        _check_plan("out", tag, rank + (lora + 15) // 16 * 16, 0, grouped, M_pad, HID)
        assert_close_16(f32(out)[rows], ref, dtype, f"out-proj {tag}")
    _same_up_to_add_order(got["ws"], got["nows"], dtype, "out-proj")

    # ---------------------------------------------------------------- fc1: GELU -> u4 requantisation + fc2 low-rank down
    act, asc, lact, mods, lays = quantize(HID, MLP, 30, xt)
    la_gpu = lact.cpu().numpy()
    m2 = [_module(MLP, HID, 40 + s, dtype, True, rank, lora) for s in sets]
    l2 = [_layer(MLP, HID, 40 + s, dtype, rank, lora) for s in sets]
    R2 = l2[0]["proj_up"].shape[1]

    def fc1_launch():
        qh = torch.empty(layout.act_image_shape(M_pad, MLP), dtype=torch.uint8, device="cuda")
        sh = torch.empty(MLP // 64, M_pad, dtype=td, device="cuda")
        lh = torch.full((M_pad, R2), 7.0, dtype=torch.float32, device="cuda")  # must be zeroed by the op
        svdq_gemm_w4a4_cuda(act=act, wgt=mods[0].qweight, qout=qh, ascales=asc, wscales=mods[0].wscales, oscales=sh, lora_act_in=lact,
                            lora_up=mods[0].proj_up, lora_scales=mods[0].lora_scales, lora_down=m2[0].proj_down, lora_act_out=lh, bias=mods[0].bias,
                            smooth_factor=m2[0].smooth_factor,
                            **(second(mods, smooth_factor=m2[1].smooth_factor, lora_down=m2[1].proj_down) if grouped else {}))
        return qh, sh, lh

    def fc1_ref(L, idx, s):
        r = rows[idx]
        q, a = checked_codes(act, asc, x, L["smooth"], dtype, rows=r)  # the operands the launch read: envelope + flip budget vs the IEEE oracle
        res = O.gemm_w4a4(q, a, L["qweight"], L["wscales"], dtype=dtype, bias=L["bias"], lora_act_in=la_gpu[r],
                          lora_up=L["proj_up"], lora_scales=L.get("lora_scales"), fuse="gelu_quant", next_smooth=l2[s]["smooth"], next_lora_down=l2[s]["proj_down"], envelope=True)
        return {"qout": res["qout"], "oscales": res["oscales"].T.copy(), "lora": res["lora_act_out"], "q_lo": res["envelope"]["q_lo"], "q_hi": res["envelope"]["q_hi"]}

    ref = oracle_rows(lays, fc1_ref)
    hidden = {}
    for tag, (qh, sh, lh) in _Both(fc1_launch):
        hidden[tag] = (qh, sh, lh)
        _check_plan("fc1", tag, rank + (lora + 15) // 16 * 16, R2, grouped, M_pad, MLP)
        codes = layout.unpack_act(qh, MLP, unsigned=True).cpu().numpy()[rows]
        d = np.abs(codes.astype(int) - ref["qout"].astype(int))
        # tanh-approximation / rounding-boundary flips of the 16-bit GELU output: codes within +-1 on < 0.5 % of the elements
        assert d.max() <= 1 and (d != 0).mean() < 5e-3, f"fc1 codes {tag}: {(d != 0).mean():.2e} differ, max {d.max()}"
        # inside the oracle's approximation envelope (where the reference's tanh.approx / __fdividef / rcp.approx may land) but for the elements
        # whose 16-bit pre-activation the GPU's fp32 accumulation order rounded the other way
        rep = O.envelope_report(codes, ref, ref["qout"])
        assert rep["outside"] < 1e-3, f"fc1 codes {tag}: {rep}"
        s_got = f32(layout.unpack_scales(sh, M_pad))[:, rows].T
        assert (s_got != ref["oscales"]).mean() < 5e-3 and np.allclose(s_got, ref["oscales"], rtol=2.0 ** -6)
        # low-rank down projection of the next layer: a sum over 12288 16-bit GELU outputs; the GPU's exp2/rcp GELU flips
        # the 16-bit rounding of a few of them by one ulp.  Bound: 2e-3 of the largest sum (the old 2e-2 hid
        # regressions), i.e. far below one 16-bit ulp of the value the next GEMM rounds it to.
        la = lh.cpu().numpy()[rows]
        mag = np.abs(ref["lora"]).max()
        assert np.abs(la - ref["lora"]).max() <= 2e-3 * mag + 1e-4, f"fc1 lora_act_out {tag}: {np.abs(la - ref['lora']).max():.3e} vs {mag:.3e}"
    assert torch.equal(hidden["ws"][0], hidden["nows"][0]) or (hidden["ws"][0] != hidden["nows"][0]).float().mean() < 1e-3

    # ---------------------------------------------------------------- fc2: K = 12288 on the GPU's own codes (stream-K)
    if M >= 4096:
        assert _streamk_on(M_pad, HID, MLP), "the K = 12288 GEMM is expected to split its remainder tiles along K on this device"
    qh, sh, lh = hidden["ws"]
    codes_rows = layout.unpack_act(qh, MLP, unsigned=True).cpu().numpy()[rows]
    sc_rows = f32(layout.unpack_scales(sh, M_pad))[:, rows]
    lh_rows = lh.cpu().numpy()[rows]

    def fc2_launch():
        out = torch.zeros(M, HID, dtype=td, device="cuda")
        svdq_gemm_w4a4_cuda(act=qh, wgt=m2[0].qweight, out=out, ascales=sh, wscales=m2[0].wscales, lora_act_in=lh, lora_up=m2[0].proj_up, lora_scales=m2[0].lora_scales,
                            bias=m2[0].bias, act_unsigned=True, **second(m2))
        return out

    def fc2_ref(L, idx, s):
        return {"out": O.gemm_w4a4(codes_rows[idx], sc_rows[:, idx], L["qweight"], L["wscales"], dtype=dtype, bias=L["bias"],
                                   lora_act_in=lh_rows[idx], lora_up=L["proj_up"], lora_scales=L.get("lora_scales"))["out"]}

    ref = oracle_rows(l2, fc2_ref)["out"]
    got = {}
    for tag, out in _Both(fc2_launch):
        got[tag] = out
        assert_close_16(f32(out)[rows], ref, dtype, f"fc2 {tag}")
    _same_up_to_add_order(got["ws"], got["nows"], dtype, "fc2")


def test_two_streams_run_stream_k_gemms_concurrently():
    """Two torch streams, each with its own workspace (nunchaku_amd._C._workspace keys it by stream), run K = 12288 GEMMs
    at the same time; results equal the serial ones bit for bit and no owner times out."""
    from nunchaku_amd._C import _workspaces, ops

    dtype, M = "bf16", 4096
    m2 = _module(MLP, HID, 40, dtype, True)
    qh = torch.randint(0, 256, (M, MLP * 3 // 4), dtype=torch.uint8, device="cuda")
    # any byte pattern is a valid FP6 image; keep the codes inside 0..15 (bit 4 and 5 of every 6-bit field clear is not
    # needed for timing, but the comparison is between two runs of the same kernel anyway)
    sh = (torch.rand(MLP // 64, M, device="cuda") * 0.1 + 0.01).to(torch.bfloat16)
    lh = torch.randn(M, R, device="cuda", dtype=torch.float32)
    assert _streamk_on(M, HID, MLP)

    def run(n):
        outs = []
        for _ in range(n):
            out = torch.empty(M, HID, dtype=torch.bfloat16, device="cuda")
            outs.append(m2.forward_quant(qh, sh, lh, out))
        return outs

    serial = run(1)[0]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    res = {}
    for _ in range(3):
        for name, st in (("a", s1), ("b", s2)):
            with torch.cuda.stream(st):
                res[name] = run(8)
    torch.cuda.synchronize()
    keys = {k for k in _workspaces if k[1] in (s1.cuda_stream, s2.cuda_stream)}
    assert len(keys) == 2 and len({_workspaces[k].buf.data_ptr() for k in keys}) == 2, "each stream must own its workspace"
    for name in ("a", "b"):
        for o in res[name]:
            assert torch.equal(o, serial), f"stream {name}: concurrent stream-K launch differs from the serial result"
    for st in (s1, s2):
        with torch.cuda.stream(st):
            ops.gemm_workspace_status()


def test_workspace_status_reports_a_broken_contract():
    """A workspace whose arrival counter is pre-loaded with garbage that can never reach `needed`... cannot be built
    safely from here; instead check the plumbing: a clean workspace reports OK, a raised error word reports and clears."""
    from nunchaku_amd import _lib
    from nunchaku_amd._C import _workspace, ops

    ws = _workspace(torch.device("cuda", torch.cuda.current_device()))
    ops.gemm_workspace_status()
    ws.buf.view(torch.int32)[1023] = 1
    with pytest.raises(RuntimeError, match="timed out"):
        ops.gemm_workspace_status()
    ops.gemm_workspace_status()  # cleared by the failing call
    assert int(ws.buf.view(torch.int32)[1023]) == 0
    assert _lib.load().svdq_gemm_workspace_bytes() <= ws.buf.numel()  # (ABI 20: a rank > 32 GELU_QUANT launch of this process may have grown it, svdq_gemm_workspace_bytes_for)
    # the hot path's check: the kernels raise the pinned, host-visible status word; the NEXT launch on the stream raises
    # without any synchronisation and the word (and the counters) are cleared
    ws.status[0] = 1
    m = _module(HID, HID, 20, "bf16")
    x = torch.randn(1, 256, HID, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="gave up waiting"):
        m(x)
    assert int(ws.status[0]) == 0
    m(x)  # and the stream is usable again
