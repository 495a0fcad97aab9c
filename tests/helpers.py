"""Test-side glue between the numpy oracle (float32 carriers) and torch GPU tensors."""

import numpy as np
import torch

from oracle import svdq_oracle as O

TORCH_DT = {"bf16": torch.bfloat16, "fp16": torch.float16}


def t16(a: np.ndarray, dtype: str, device="cuda") -> torch.Tensor:
    """exactly-representable float32 carrier -> 16-bit torch tensor"""
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(TORCH_DT[dtype]).to(device)


def f32(t: torch.Tensor) -> np.ndarray:
    return t.detach().float().cpu().numpy()


def reference_state_dict(layer: dict, dtype: str) -> dict:
    """Logical oracle layer -> tensors in the REFERENCE checkpoint layout (what load_state_dict sees)."""
    td = TORCH_DT[dtype]
    sd = {
        "qweight": torch.from_numpy(O.pack_qweight_ref(layer["qweight"])),
        "wscales": torch.from_numpy(O.pack_wscales_ref(layer["wscales"])).to(td),
        "smooth_factor": torch.from_numpy(O.pack_vec_ref(layer["smooth"])).to(td),
        "smooth_factor_orig": torch.from_numpy(O.pack_vec_ref(layer["smooth"])).to(td),
        "proj_down": torch.from_numpy(O.pack_lowrank_ref(np.ascontiguousarray(layer["proj_down"].T), down=True)).to(td),
        "proj_up": torch.from_numpy(O.pack_lowrank_ref(layer["proj_up"], down=False)).to(td),
    }
    if layer.get("bias") is not None:
        sd["bias"] = torch.from_numpy(O.pack_vec_ref(layer["bias"])).to(td)
    return sd


def make_module(layer: dict, dtype: str, act_unsigned=False, device="cuda"):
    from nunchaku_amd.models.linear import SVDQW4A4Linear

    N, K = layer["qweight"].shape
    R = layer["proj_up"].shape[1]
    m = SVDQW4A4Linear(K, N, rank=R, bias=layer.get("bias") is not None, act_unsigned=act_unsigned,
                       torch_dtype=TORCH_DT[dtype], device=device)
    m.load_state_dict(reference_state_dict(layer, dtype))
    return m


def assert_close_16(got: np.ndarray, ref: np.ndarray, dtype: str, what: str, max_bad_frac=0.0, ulps=1.0):
    """|got - ref| <= ulps * (one 16-bit ulp relative bound) * |ref| + tiny absolute slack."""
    rel = (2.0 ** -7 if dtype == "bf16" else 2.0 ** -10) * ulps
    atol = rel * 1e-2 * float(np.abs(ref).max() + 1e-30)
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    bad = err > rel * np.abs(ref) + atol
    frac = bad.mean()
    assert frac <= max_bad_frac, f"{what}: {bad.sum()} / {bad.size} elements off by more than {ulps} ulp (max err {err.max():.4g})"


def psnr_db(got, ref) -> float:
    """10 log10(max|ref|^2 / mean (got - ref)^2) of two torch tensors"""
    import torch

    got, ref = got.float(), ref.float()
    mse = ((got - ref) ** 2).mean().item()
    return float("inf") if mse == 0 else 10.0 * float(torch.log10(ref.abs().max() ** 2 / mse))
