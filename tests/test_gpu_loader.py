"""A checkpoint-layout state dict (legacy key names) loads into a fresh model and reproduces the source model bit for bit."""
import pytest
import torch

from nunchaku_amd.models import loader
from nunchaku_amd.models.flux import FluxTransformerAMD

pytestmark = pytest.mark.gpu


def test_legacy_state_dict_loads_and_reproduces(built_lib):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    kw = dict(num_layers=1, num_single_layers=1, dim=256, heads=2, in_channels=64, joint_attention_dim=128, pooled_projection_dim=64)
    src = FluxTransformerAMD(**kw, device="cuda").init_synthetic_(seed=5, repack=False).eval()
    legacy = {k: v.clone() for k, v in loader.export_legacy_state_dict(src).items()}
    dst = loader.load_flux_state_dict(FluxTransformerAMD(**kw, device="cuda"), legacy).eval()
    g = torch.Generator(device="cuda").manual_seed(1)
    side, t_txt = 16, 128
    lat = torch.randn(1, side * side, 64, device="cuda", generator=g).bfloat16()
    enc = torch.randn(1, t_txt, 128, device="cuda", generator=g).bfloat16()
    pooled = torch.randn(1, 64, device="cuda", generator=g).bfloat16()
    img_ids = torch.zeros(side * side, 3, device="cuda")
    img_ids[:, 1] = torch.arange(side, device="cuda").repeat_interleave(side)
    img_ids[:, 2] = torch.arange(side, device="cuda").repeat(side)
    txt_ids = torch.zeros(t_txt, 3, device="cuda")
    t, gd = torch.tensor([0.3], device="cuda"), torch.tensor([3.5], device="cuda")
    from nunchaku_amd import mode

    with torch.no_grad(), mode.deterministic_mode():  # fixed-point low-rank reductions: identical weights -> identical bits
        a = src(lat, enc, pooled, t, img_ids, txt_ids, gd)
        b = dst(lat, enc, pooled, t, img_ids, txt_ids, gd)
    assert torch.isfinite(a).all()
    assert torch.equal(a, b), f"loaded model differs from its source: max |d| {(a.float() - b.float()).abs().max().item():.3e}"
    # src has been repacked by its forward pass: its state dict is exported through the inverse permutations, bit for bit
    again = loader.export_legacy_state_dict(src)
    assert set(again) == set(legacy)
    for k, v in legacy.items():
        assert torch.equal(again[k], v), f"{k}: export after the first forward differs from the checkpoint"


def test_captured_step_replays_like_eager(built_lib):
    """HIP-graph capture of a whole step (nunchaku_amd/graph.py): replays match the eager forward."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from nunchaku_amd.graph import CapturedStep

    kw = dict(num_layers=1, num_single_layers=1, dim=256, heads=2, in_channels=64, joint_attention_dim=128, pooled_projection_dim=64)
    model = FluxTransformerAMD(**kw, device="cuda").init_synthetic_(seed=2).eval()
    side, t_txt = 16, 128
    g = torch.Generator(device="cuda").manual_seed(4)
    enc = torch.randn(1, t_txt, 128, device="cuda", generator=g).bfloat16()
    pooled = torch.randn(1, 64, device="cuda", generator=g).bfloat16()
    img_ids = torch.zeros(side * side, 3, device="cuda")
    img_ids[:, 1] = torch.arange(side, device="cuda").repeat_interleave(side)
    img_ids[:, 2] = torch.arange(side, device="cuda").repeat(side)
    txt_ids = torch.zeros(t_txt, 3, device="cuda")
    gd = torch.tensor([3.5], device="cuda")
    from nunchaku_amd import mode

    fn = lambda lat, t: model(lat, enc, pooled, t, img_ids, txt_ids, gd)
    lat0 = torch.randn(1, side * side, 64, device="cuda", generator=g).bfloat16()
    with mode.deterministic_mode():  # the captured launches carry the fixed-point accumulator format: replays == eager, bit for bit
        cap = CapturedStep(fn, [lat0, torch.tensor([0.5], device="cuda")])
        for seed, tv in ((7, 0.9), (8, 0.2)):
            lat = torch.randn(1, side * side, 64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(seed)).bfloat16()
            t = torch.tensor([tv], device="cuda")
            got = cap(lat, t).clone()
            with torch.no_grad():
                ref = fn(lat, t)
            assert torch.isfinite(got).all()
            assert torch.equal(got, ref), f"replay differs from eager: max |d| {(got.float() - ref.float()).abs().max().item():.3e}"
    # default mode (fp32 atomics): replays agree with eager to the W4A4 code-flip level
    cap = CapturedStep(fn, [lat0, torch.tensor([0.5], device="cuda")])
    got = cap(lat, t).clone()
    with torch.no_grad():
        ref = fn(lat, t)
    from tests.helpers import psnr_db

    assert psnr_db(got, ref) > 45.0, psnr_db(got, ref)


def _tiny_inputs(side=16, t_txt=128, seed=4):
    g = torch.Generator(device="cuda").manual_seed(seed)
    lat = torch.randn(1, side * side, 64, device="cuda", generator=g).bfloat16()
    enc = torch.randn(1, t_txt, 128, device="cuda", generator=g).bfloat16()
    pooled = torch.randn(1, 64, device="cuda", generator=g).bfloat16()
    img_ids = torch.zeros(side * side, 3, device="cuda")
    img_ids[:, 1] = torch.arange(side, device="cuda").repeat_interleave(side)
    img_ids[:, 2] = torch.arange(side, device="cuda").repeat(side)
    return lat, enc, pooled, img_ids, torch.zeros(t_txt, 3, device="cuda")


def test_forward_under_inference_mode(built_lib):
    """ids created under torch.inference_mode() track no version counter: the rotary cache must step aside, not raise"""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from nunchaku_amd import mode

    kw = dict(num_layers=1, num_single_layers=1, dim=256, heads=2, in_channels=64, joint_attention_dim=128, pooled_projection_dim=64)
    model = FluxTransformerAMD(**kw, device="cuda").init_synthetic_(seed=2).eval()
    lat, enc, pooled, img_ids, txt_ids = _tiny_inputs()
    t, gd = torch.tensor([0.5], device="cuda"), torch.tensor([3.5], device="cuda")
    with mode.deterministic_mode():
        with torch.no_grad():
            ref = model(lat, enc, pooled, t, img_ids, txt_ids, gd)
        with torch.inference_mode():
            lat_i, enc_i, pooled_i, img_i, txt_i = _tiny_inputs()  # inference tensors: ._version raises
            assert img_i.is_inference()
            a = model(lat_i, enc_i, pooled_i, t.clone(), img_i, txt_i, gd.clone())
            b = model(lat_i, enc_i, pooled_i, t.clone(), img_i, txt_i, gd.clone())
    assert torch.equal(a, ref) and torch.equal(b, ref)


def test_captured_step_survives_a_later_eager_call_with_other_ids(built_lib):
    """a captured graph must not point at rotary tables that only the one-entry cache keeps alive -- an eager call with another grid
    replaces the cache entry; the next replay has to give the same answer as before"""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from nunchaku_amd import mode
    from nunchaku_amd.graph import CapturedStep

    kw = dict(num_layers=1, num_single_layers=1, dim=256, heads=2, in_channels=64, joint_attention_dim=128, pooled_projection_dim=64)
    model = FluxTransformerAMD(**kw, device="cuda").init_synthetic_(seed=2).eval()
    lat, enc, pooled, img_ids, txt_ids = _tiny_inputs()
    gd = torch.tensor([3.5], device="cuda")
    fn = lambda x, t: model(x, enc, pooled, t, img_ids, txt_ids, gd)
    with mode.deterministic_mode():
        cap = CapturedStep(fn, [lat, torch.tensor([0.5], device="cuda")])
        t = torch.tensor([0.4], device="cuda")
        first = cap(lat, t).clone()
        # another grid / text length through the eager path: evicts the cache entry, allocates and frees tables of its own
        lat2, enc2, pooled2, img2, txt2 = _tiny_inputs(side=32, t_txt=256, seed=9)
        with torch.no_grad():
            for _ in range(2):
                model(lat2, enc2, pooled2, t, img2, txt2, gd)
            junk = [torch.full((1 << 16,), float("nan"), device="cuda") for _ in range(32)]  # recycle whatever the eager call freed
        again = cap(lat, t).clone()
        del junk
    assert torch.isfinite(again).all() and torch.equal(first, again)


def test_pipeline_facing_class_call_contract(built_lib, tmp_path):
    """NunchakuFluxTransformer2DModelV2: from_pretrained on a safetensors file and the keyword call a FluxPipeline makes."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import json

    from safetensors.torch import save_file

    from nunchaku_amd.models.transformer_flux import NunchakuFluxTransformer2DModelV2, NunchakuFluxTransformer2dModel

    assert NunchakuFluxTransformer2dModel is NunchakuFluxTransformer2DModelV2
    cfg = dict(num_layers=1, num_single_layers=1, num_attention_heads=2, attention_head_dim=128, in_channels=64,
               joint_attention_dim=128, pooled_projection_dim=64, guidance_embeds=True, axes_dims_rope=[16, 56, 56])
    src = NunchakuFluxTransformer2DModelV2(cfg, device="cuda").init_synthetic_(seed=1, repack=False)
    path = str(tmp_path / "svdq-int4_r32-tiny.safetensors")
    save_file({k: v.contiguous().cpu() for k, v in loader.export_legacy_state_dict(src).items()}, path,
              metadata={"config": json.dumps(cfg), "quantization_config": json.dumps({"rank": 32})})
    model = NunchakuFluxTransformer2DModelV2.from_pretrained(path, device="cuda", torch_dtype=torch.bfloat16).eval()
    assert model.config.in_channels == 64 and model.config.guidance_embeds and model.dtype == torch.bfloat16
    side, t_txt = 16, 128
    g = torch.Generator(device="cuda").manual_seed(2)
    lat = torch.randn(1, side * side, 64, device="cuda", generator=g).bfloat16()
    img_ids = torch.zeros(side * side, 3, device="cuda")
    img_ids[:, 1] = torch.arange(side, device="cuda").repeat_interleave(side)
    img_ids[:, 2] = torch.arange(side, device="cuda").repeat(side)
    kwargs = dict(hidden_states=lat, timestep=torch.tensor([0.5], device="cuda"), guidance=torch.tensor([3.5], device="cuda"),
                  pooled_projections=torch.randn(1, 64, device="cuda", generator=g).bfloat16(),
                  encoder_hidden_states=torch.randn(1, t_txt, 128, device="cuda", generator=g).bfloat16(),
                  txt_ids=torch.zeros(t_txt, 3, device="cuda"), img_ids=img_ids, joint_attention_kwargs=None)
    from nunchaku_amd import mode

    with torch.no_grad(), mode.deterministic_mode():
        noise_pred = model(**kwargs, return_dict=False)[0]  # exactly how FluxPipeline.__call__ uses its transformer
        out = model(**kwargs)
        ref = src(lat, kwargs["encoder_hidden_states"], kwargs["pooled_projections"], kwargs["timestep"], img_ids,
                  kwargs["txt_ids"], kwargs["guidance"]).sample
    assert noise_pred.shape == lat.shape and torch.isfinite(noise_pred.float()).all()
    assert torch.equal(out.sample, noise_pred)
    assert torch.equal(noise_pred, ref), f"from_pretrained model differs from its source: max |d| {(noise_pred.float() - ref.float()).abs().max().item():.3e}"
    # ControlNet residuals are part of the call contract (round 4; parity: tests/test_flux_block_parity.py): [1, T_img, hidden] per sample
    hidden = model.x_embedder.out_features
    ctrl = [torch.full((1, lat.shape[1], hidden), 0.25, dtype=lat.dtype, device=lat.device)]
    with torch.no_grad():
        with_ctrl = model(**kwargs, controlnet_block_samples=ctrl, controlnet_single_block_samples=ctrl).sample
    assert with_ctrl.shape == noise_pred.shape and torch.isfinite(with_ctrl.float()).all() and not torch.equal(with_ctrl, noise_pred)
