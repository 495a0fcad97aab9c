"""Per-tensor layout bookkeeping of SVDQW4A4Linear (a partial load_state_dict must not corrupt a repacked layer).
CPU only: the repack itself is a GPU kernel, so the "repacked" state is staged by hand (FP6-image shape + names)."""

import pytest
import torch

from nunchaku_amd.models.linear import SVDQW4A4Linear


def _layer():
    lin = SVDQW4A4Linear(128, 256, rank=16, device="cpu")
    with torch.no_grad():
        lin.qweight.copy_(torch.randint(-128, 128, lin.qweight.shape, dtype=torch.int8))
        for p in (lin.wscales, lin.bias, lin.smooth_factor, lin.smooth_factor_orig, lin.proj_down, lin.proj_up):
            p.copy_(torch.randn(p.shape))
    return lin


def _stage_repacked(lin):
    lin.qweight.data = torch.randint(-128, 128, (256, 96), dtype=torch.int8)  # [out, 3*in/4] FP6 image
    lin._amd_layout = True
    assert lin._amd_layout and lin._amd_names == {"qweight", "wscales", "smooth_factor", "bias", "proj_down", "proj_up"}


def test_partial_load_touches_only_the_tensors_it_brings():
    lin = _layer()
    _stage_repacked(lin)
    image = lin.qweight.data.clone()
    new_bias = torch.randn(256).to(lin.bias.dtype)
    res = lin.load_state_dict({"bias": new_bias}, strict=False)
    assert "qweight" in res.missing_keys
    assert tuple(lin.qweight.shape) == (256, 96) and torch.equal(lin.qweight.data, image), "the FP6 image must survive"
    assert torch.equal(lin.bias.data, new_bias)
    # only the bias is back in checkpoint layout: repack_() would convert exactly that tensor
    assert lin._amd_names == {"qweight", "wscales", "smooth_factor", "proj_down", "proj_up"} and not lin._amd_layout


def test_full_load_restores_the_checkpoint_shape_and_marks_everything():
    src, lin = _layer(), _layer()
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    _stage_repacked(lin)
    lin.load_state_dict(sd)
    assert tuple(lin.qweight.shape) == (256, 64) and torch.equal(lin.qweight.data, sd["qweight"])
    assert lin._amd_names == set() and not lin._amd_layout


def test_state_dict_of_a_repacked_layer_goes_through_the_inverse_repack():
    """state_dict() of a layer in the kernel layout is converted back to the checkpoint layout by the GPU's inverse
    permutations (tests/test_gpu_parity.py checks the round trip bit for bit); without a GPU that is an error, never a
    silently wrong "checkpoint"."""
    lin = _layer()
    assert set(lin.state_dict()) >= {"qweight", "wscales", "bias", "proj_down", "proj_up", "smooth_factor"}
    _stage_repacked(lin)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="GPU|cuda|CUDA|HIP"):
            lin.state_dict()
        with pytest.raises(RuntimeError, match="GPU|cuda|CUDA|HIP"):
            torch.nn.Sequential(lin).state_dict()


def test_deepcopy_of_a_repacked_layer_keeps_its_layout_state():
    """Parameter.__deepcopy__ drops python attributes (the per-tensor `_svdq_amd` marks); `_amd_names` survives
    and the marks are re-stamped from it, so the copy is NOT converted a second time."""
    import copy

    lin = _layer()
    _stage_repacked(lin)
    twin = copy.deepcopy(lin)
    assert twin._amd_names == lin._amd_names and twin._amd_layout
    assert not getattr(twin.bias, "_svdq_amd", False), "torch drops tensor attributes on deepcopy (if this fails the re-stamp is moot)"
    twin._ensure_layout()  # nothing to repack, marks restored
    for n in twin._amd_names:
        assert getattr(getattr(twin, n), "_svdq_amd", False), n
    assert torch.equal(twin.qweight.data, lin.qweight.data)


def test_runtime_lora_is_dropped_only_when_low_rank_tensors_arrive():
    lin = _layer()
    lin._amd_layout = True  # set_lora() would repack otherwise (GPU only)
    lin.set_lora(torch.randn(8, 128), torch.randn(256, 8), 0.5)
    assert lin.rank == 32
    lin.load_state_dict({"bias": torch.zeros(256).to(lin.bias.dtype)}, strict=False)
    assert lin.rank == 32, "a bias-only update must keep the attached LoRA"
    lin.load_state_dict({"proj_up": torch.zeros(256, 16).to(lin.proj_up.dtype)}, strict=False)
    assert lin.rank == 16 and lin.lora_scales is None
