"""Checkpoint plumbing: key conversion pinned to the reference's own
convert_flux_state_dict (tests/golden/flux_keys.json, made by tools/make_golden.py), and a safetensors round trip."""
import json

import pytest
import torch

from nunchaku_amd.models import loader
from nunchaku_amd.models.flux import FluxTransformerAMD


def _small(device="cpu"):
    return FluxTransformerAMD(num_layers=2, num_single_layers=2, dim=256, heads=2, in_channels=64, joint_attention_dim=128,
                              pooled_projection_dim=64, device=device)


def test_key_conversion_agrees_with_reference_converter(golden_dir):
    """legacy key -> (reference converter) V2 key: this package's converter must produce EXACTLY the reference's V2 key (its
    module tree uses the V2 / diffusers names), and a V2 key must pass through unchanged."""
    table = json.load(open(f"{golden_dir}/flux_keys.json"))
    assert len(table) > 100
    own = set(_small().state_dict().keys())
    seen = set()
    for legacy, v2 in table.items():
        a, b = loader.convert_key(legacy), loader.convert_key(v2)
        assert a == b, (legacy, v2, a, b)
        if a is not None:
            assert a == v2, f"{legacy}: converted to {a}, the reference's convert_flux_state_dict gives {v2}"
            import re
            a = re.sub(r"blocks\.(18|37)\.", "blocks.1.", a)  # the fixture names FLUX.1's last blocks; the test model has two
            assert a in own, f"{legacy} -> {a} is not a parameter of FluxTransformerAMD"
            seen.add(a)
    # every SVDQ / AWQ / norm parameter of block 0 and single block 0 is reachable from a checkpoint key
    need = {k for k in own if k.startswith(("transformer_blocks.0.", "single_transformer_blocks.0."))}
    assert need <= seen, sorted(need - seen)[:5]


def test_unknown_and_nvfp4_keys():
    assert loader.convert_key("transformer_blocks.0.qkv_proj.wcscales") is None
    assert loader.convert_key("single_transformer_blocks.3.mlp_fc2.wtscale") is None
    with pytest.raises(KeyError):
        loader.convert_key("transformer_blocks.0.unknown_proj.qweight")
    with pytest.raises(KeyError):
        loader.convert_key("something_else.weight")


def test_safetensors_round_trip(tmp_path):
    from safetensors.torch import save_file

    torch.manual_seed(0)
    src = _small()
    with torch.no_grad():
        for p in src.parameters():
            if p.dtype in (torch.int8, torch.int32):
                p.copy_(torch.randint(-100, 100, p.shape, dtype=torch.int64))
            else:
                p.copy_(torch.randn(p.shape))
    legacy = loader.export_legacy_state_dict(src)
    assert "transformer_blocks.1.mlp_context_fc1.lora_down" in legacy and "single_transformer_blocks.0.norm.linear.qweight" in legacy
    cfg = dict(num_layers=2, num_single_layers=2, num_attention_heads=2, attention_head_dim=128, in_channels=64,
               joint_attention_dim=128, pooled_projection_dim=64, guidance_embeds=True, axes_dims_rope=[16, 56, 56])
    path = str(tmp_path / "svdq-int4_r32-tiny.safetensors")
    save_file({k: v.contiguous() for k, v in legacy.items()}, path,
              metadata={"config": json.dumps(cfg), "quantization_config": json.dumps({"rank": 32})})
    dst = loader.from_pretrained(path, device="cpu")
    a, b = src.state_dict(), dst.state_dict()
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # dtype mismatches are errors, as in the reference's patch_scale_key
    bad = dict(legacy)
    bad["x_embedder.weight"] = bad["x_embedder.weight"].float()
    with pytest.raises(TypeError):
        loader.load_flux_state_dict(_small(), bad)


def test_zero_pool_hands_out_aligned_pieces_once():
    from nunchaku_amd.ops.elementwise import ZeroPool

    pool = ZeroPool(torch.zeros(100))
    a, b = pool.take(10), pool.take(50)
    assert a.numel() == 10 and b.numel() == 50
    assert (b.data_ptr() - a.data_ptr()) == 12 * 4  # pieces start on 16-byte boundaries
    assert pool.take(40) is None and pool.take(36) is not None


def test_qwen_image_safetensors_round_trip(tmp_path):
    """NunchakuQwenImageTransformer2DModel.from_pretrained (reference transformer_qwenimage.py:358-413): the checkpoint's keys are
    the module names themselves; NVFP4-only tensors are dropped; dtype mismatches are errors."""
    from safetensors.torch import save_file

    from nunchaku_amd.models.qwenimage import NunchakuQwenImageTransformer2DModel as Q

    cfg = dict(num_layers=2, num_attention_heads=2, attention_head_dim=128, in_channels=64, out_channels=16, joint_attention_dim=128,
               patch_size=2, axes_dims_rope=[16, 56, 56])
    torch.manual_seed(1)
    src = Q(**cfg, device="cpu")
    with torch.no_grad():
        for p in src.parameters():
            p.copy_(torch.randint(-100, 100, p.shape, dtype=torch.int64) if p.dtype in (torch.int8, torch.int32) else torch.randn(p.shape))
    sd = {k: v.contiguous() for k, v in src.state_dict().items()}
    for k in ("transformer_blocks.0.attn.to_qkv.qweight", "transformer_blocks.1.attn.to_out.0.proj_up", "transformer_blocks.0.img_mod.1.wzeros",
              "transformer_blocks.1.txt_mlp.net.0.proj.wscales", "transformer_blocks.0.img_mlp.net.2.smooth_factor",
              "time_text_embed.timestep_embedder.linear_1.weight", "txt_norm.weight", "img_in.weight", "norm_out.linear.bias", "proj_out.weight"):
        assert k in sd, k
    path = str(tmp_path / "svdq-int4_r32-qwen-tiny.safetensors")
    extra = dict(sd)
    extra["transformer_blocks.0.attn.to_qkv.wtscale"] = torch.ones(1)  # NVFP4-only tensor: ignored
    save_file(extra, path, metadata={"config": json.dumps(cfg), "quantization_config": json.dumps({"rank": 32})})
    dst = Q.from_pretrained(path, device="cpu")
    b = dst.state_dict()
    assert sd.keys() == b.keys() and all(torch.equal(sd[k], b[k]) for k in sd)
    assert dst.config.num_layers == 2 and not dst.offload
    bad = dict(sd)
    bad["img_in.weight"] = bad["img_in.weight"].float()
    save_file(bad, path, metadata={"config": json.dumps(cfg)})
    with pytest.raises(TypeError):
        Q.from_pretrained(path, device="cpu")
