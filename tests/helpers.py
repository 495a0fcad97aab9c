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


def checked_codes(qx, asc, x: np.ndarray, smooth, dtype: str, max_flips: float = 1e-3, rows=None):
    """The operands a GEMM launch consumed -- the GPU quantiser's codes [M_pad, K] and scales [K/64, M_pad] as numpy -- after checking them against
    the oracle: every code and scale inside the approximation envelope (oracle.quantize_envelope: the reference divides with __fdividef and
    inverts the scale with rcp.approx; this library with v_rcp_f32 -- neither is the IEEE quotient, both must land in the envelope), and at most
    ``max_flips`` of the codes / scales different from the IEEE oracle's (SURVEY.md section 8c's tolerance: +-1 LSB on < 1e-3 of the elements).
    GEMM tests then hold the GEMM to 1 ulp on the operands it really read.  ``rows``: check (and return) only these rows of a large launch; ``x`` is
    then the full input."""
    from nunchaku_amd import layout

    K = x.shape[1]
    codes = layout.unpack_act(qx, K).cpu().numpy()
    scales = f32(layout.unpack_scales(asc, codes.shape[0]))
    if rows is not None:
        x, codes, scales = x[rows], codes[rows], scales[:, rows]
    q_ieee, a_ieee, _ = O.quantize_w4a4_act_fuse_lora(x, smooth, None, dtype, pad_size=1 if rows is not None else O.PAD_M)
    env = O.quantize_envelope(x, smooth, dtype, pad_size=1 if rows is not None else O.PAD_M)
    rep = O.envelope_report(codes, env, q_ieee)
    assert rep["outside"] == 0.0, f"quantiser codes outside the approximation envelope: {rep}"
    assert rep["flips_vs_ieee"] <= max_flips and rep["max_abs_diff_vs_ieee"] <= 1, f"quantiser codes vs the IEEE oracle: {rep}"
    assert np.all(scales >= env["s_lo"]) and np.all(scales <= env["s_hi"]), "quantiser scales outside the approximation envelope"
    assert (scales != a_ieee).mean() <= max_flips, f"{(scales != a_ieee).sum()} scales differ from the IEEE oracle"
    return codes, scales
