"""The ``nunchaku`` import surface: the reference's package layout resolves to the MI355X
implementation, and -- where the reference tree is present (this container; not the GPU box) -- the reference's OWN,
UNMODIFIED ``ops/*.py`` and ``models/linear.py`` import and run against the shim ``nunchaku._C`` up to the first kernel
launch (which needs a GPU: the call must arrive in this library with the reference's positional arguments)."""
import importlib
import importlib.util
import os
import sys
import types

import pytest
import torch

REF = "/root/reference/nunchaku"


def test_reference_import_paths_resolve():
    import nunchaku
    from nunchaku import NunchakuFluxTransformer2DModelV2, NunchakuFluxTransformer2dModel, NunchakuQwenImageTransformer2DModel
    from nunchaku._C import ops, utils
    from nunchaku.models.linear import AWQW4A16Linear, SVDQW4A4Linear
    from nunchaku.models.transformers.transformer_flux_v2 import NunchakuFluxTransformer2DModelV2 as V2
    from nunchaku.models.utils import CPUOffloadManager
    from nunchaku.ops.fused import fused_gelu_mlp, fused_qkv_norm_rottary
    from nunchaku.ops.gemm import svdq_gemm_w4a4_cuda
    from nunchaku.ops.quantize import svdq_quantize_w4a4_act_fuse_lora_cuda
    from nunchaku.utils import ceil_divide, get_precision

    assert V2 is NunchakuFluxTransformer2DModelV2 and NunchakuFluxTransformer2dModel is V2
    for name in ("gemm_w4a4", "quantize_w4a4_act_fuse_lora", "attention_fp16", "gemv_awq"):
        assert callable(getattr(ops, name)), name
    for name in ("set_log_level", "disable_memory_auto_release", "trim_memory", "set_faster_i2f_mode"):
        assert getattr(utils, name)("x") is None if name in ("set_log_level", "set_faster_i2f_mode") else getattr(utils, name)() is None
    assert get_precision() == "int4" and ceil_divide(5, 2) == 3
    lin = SVDQW4A4Linear(128, 256, rank=16, device="cpu")
    assert tuple(lin.qweight.shape) == (256, 64) and lin.qweight.dtype == torch.int8
    assert all(callable(f) for f in (fused_gelu_mlp, fused_qkv_norm_rottary, svdq_gemm_w4a4_cuda, svdq_quantize_w4a4_act_fuse_lora_cuda))
    assert CPUOffloadManager.__name__ == "CPUOffloadManager" and AWQW4A16Linear and NunchakuQwenImageTransformer2DModel and nunchaku.__all__


def _load_reference_as(pkg_name: str):
    """A synthetic package ``pkg_name`` whose ``_C`` is this repo's shim and whose ``utils``, ``ops.*`` and ``models.linear``
    are the reference's files, loaded by path, unmodified."""
    import nunchaku._C as shim_c

    def module(name, path=None, is_pkg=False):
        if path is None:
            m = types.ModuleType(name)
            m.__path__ = []
        else:
            spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[] if is_pkg else None)
            m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        return m

    module(pkg_name)
    sys.modules[pkg_name + "._C"] = shim_c
    sys.modules[pkg_name]._C = shim_c
    u = module(pkg_name + ".utils", os.path.join(REF, "utils.py"))
    try:
        u.__spec__.loader.exec_module(u)
    except ImportError as e:  # the reference's utils.py imports safetensors / huggingface_hub at module level
        pytest.skip(f"reference utils.py needs a package that is not installed: {e}")
    module(pkg_name + ".ops")
    mods = {}
    for sub in ("quantize", "gemm", "gemv", "fused"):
        pass
    module(pkg_name + ".models")
    for name, rel in ((".ops.quantize", "ops/quantize.py"), (".ops.gemm", "ops/gemm.py"), (".ops.gemv", "ops/gemv.py"),
                      (".models.linear", "models/linear.py"), (".ops.fused", "ops/fused.py")):
        m = module(pkg_name + name, os.path.join(REF, rel))
        m.__spec__.loader.exec_module(m)
        mods[name] = m
    return mods


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree only exists in the build container")
def test_unmodified_reference_callers_bind_to_the_shim():
    mods = _load_reference_as("_ref_nunchaku")
    RefLinear = mods[".models.linear"].SVDQW4A4Linear
    lin = RefLinear(128, 256, rank=16, torch_dtype=torch.bfloat16, device="cpu")
    assert tuple(lin.qweight.shape) == (256, 64)
    x = torch.zeros(1, 4, 128, dtype=torch.bfloat16)
    # the reference's forward -> its quantize() -> its ops/quantize.py wrapper (allocates the [M_pad, K/2] buffer) -> shim
    # _C.ops.quantize_w4a4_act_fuse_lora with the reference's 8 positional arguments: arrives in this library, which has no
    # CPU path and says so (on a GPU the same call runs: tests/test_gpu_parity.py::test_reference_style_calls...)
    with pytest.raises(RuntimeError, match="no CPU path|need GPU tensors"):
        lin(x)
    fused = mods[".ops.fused"]
    fc1 = RefLinear(128, 256, rank=16, torch_dtype=torch.bfloat16, device="cpu")
    fc2 = RefLinear(256, 128, rank=16, torch_dtype=torch.bfloat16, device="cpu")
    with pytest.raises(RuntimeError, match="no CPU path|need GPU tensors"):
        fused.fused_gelu_mlp(x, fc1, fc2)
