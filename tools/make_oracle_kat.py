#!/usr/bin/env python3
"""Known-answer vectors of the ORACLE itself: tests/golden/oracle_kat_seed{0,1}.npz.

The reference holds no kernel-level vectors for this path, so nothing pins the oracle's ARITHMETIC against the
reference; what CAN be pinned is the oracle against its own past: this script stores inputs and the oracle's outputs for the
quantiser, every GEMM epilogue (both accumulation modes), the fused MLP chain, the tiled attention restatement, the AWQ GEMV and the
element-wise glue, and `tests/test_oracle_kat.py` checks on every run that the oracle still reproduces them.  An edit of
oracle/svdq_oracle.py that changes any result therefore shows up as a changed fixture in review instead of moving oracle and kernel
together unnoticed.  Regenerate ONLY on purpose:

    python tools/make_oracle_kat.py            # rewrites the fixtures
    python tools/make_oracle_kat.py --check    # recompute and compare with the committed files (what the test does)
"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle import svdq_oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
F32 = np.float32


def _layer_inputs(layer):
    return {k: v for k, v in layer.items() if k != "dense" and v is not None}


def compute(seed: int, stored: dict | None = None) -> dict:
    """All KAT entries for one seed.  With ``stored`` the INPUTS are taken from the fixture (so a numpy RNG change cannot move them) and
    only the oracle's outputs are recomputed."""
    d = {}
    KEEP = 48  # rows kept of every row-padded output (M = 40 real rows + 8 of the zero-padded ones): the fixtures stay small

    inputs = set()

    def inp(name, make):
        inputs.add(name)
        if stored is not None:
            d[name] = stored[name]
        else:
            d[name] = make()
        return d[name]

    for dtype in ("bf16", "fp16"):
        t = dtype
        K, N, R, M = 256, 384, 32, 40
        lay = {k: inp(f"{t}.l1.{k}", lambda k=k: _layer_inputs(O.make_svdq_layer(K, N, R, seed=seed, dtype=dtype))[k])
               for k in ("qweight", "wscales", "smooth", "proj_down", "proj_up", "bias")}
        x = inp(f"{t}.x", lambda: O.make_activations(M, K, seed=seed, dtype=dtype))
        # quantiser (signed, smoothing + low-rank down), and its fuse_glu form
        q, asc, la = O.quantize_w4a4_act_fuse_lora(x, lay["smooth"], lay["proj_down"], dtype)
        d[f"{t}.quant.codes"], d[f"{t}.quant.ascales"], d[f"{t}.quant.lora_act"] = q, asc, la
        # round 5: the approximation envelope of the same call (__fdividef / rcp.approx widened by their documented bounds)
        env = O.quantize_envelope(x, lay["smooth"], dtype)
        d[f"{t}.quant.env.q_lo"], d[f"{t}.quant.env.q_hi"], d[f"{t}.quant.env.s_lo"], d[f"{t}.quant.env.s_hi"] = env["q_lo"], env["q_hi"], env["s_lo"], env["s_hi"]
        x2 = inp(f"{t}.x_glu", lambda: O.round16(np.random.default_rng(77 + seed).standard_normal((40, 2 * K)).astype(F32), dtype))
        qg, ascg, _ = O.quantize_w4a4_act_fuse_lora(x2, lay["smooth"], None, dtype, fuse_glu=True)
        d[f"{t}.quant_glu.codes"], d[f"{t}.quant_glu.ascales"] = qg, ascg
        # GEMM epilogues, both accumulation modes
        for accum in ("fp32", "ref16"):
            for fuse in ("none", "silu"):
                r = O.gemm_w4a4(q, asc, lay["qweight"], lay["wscales"], dtype=dtype, bias=lay["bias"], lora_act_in=la, lora_up=lay["proj_up"],
                                fuse=fuse, accum=accum)
                d[f"{t}.gemm.{fuse}.{accum}.out"] = r["out"]
        ls = inp(f"{t}.lora_scales", lambda: np.array([1.0, 0.5], dtype=F32))
        r = O.gemm_w4a4(q, asc, lay["qweight"], lay["wscales"], dtype=dtype, bias=None, lora_act_in=la, lora_up=lay["proj_up"], lora_scales=list(ls))
        d[f"{t}.gemm.lora_scales_nobias.out"] = r["out"]
        # RMSNorm + RoPE (N = 3 heads of 128)
        rng = np.random.default_rng(300 + seed)
        nq = inp(f"{t}.norm_q", lambda: O.round16((1.0 + 0.1 * rng.standard_normal(128)).astype(F32), dtype))
        nk = inp(f"{t}.norm_k", lambda: O.round16((1.0 + 0.1 * rng.standard_normal(128)).astype(F32), dtype))
        Mp = q.shape[0]
        rot = inp(f"{t}.rot", lambda: np.stack([np.sin(a := rng.uniform(0, 6.28, size=(Mp, 64)).astype(F32)), np.cos(a)], axis=-1).astype(F32))
        r = O.gemm_w4a4(q, asc, lay["qweight"], lay["wscales"], dtype=dtype, bias=lay["bias"], lora_act_in=la, lora_up=lay["proj_up"],
                        fuse="rmsnorm_rope", norm_q=nq, norm_k=nk, rot=rot)
        d[f"{t}.gemm.rmsnorm_rope.out"] = r["out"]
        # GELU -> requantise -> next low-rank down, then the fused MLP chain
        lay2 = {k: inp(f"{t}.l2.{k}", lambda k=k: _layer_inputs(O.make_svdq_layer(N, 128, R, seed=seed + 10, dtype=dtype))[k])
                for k in ("qweight", "wscales", "smooth", "proj_down", "proj_up", "bias")}
        r = O.gemm_w4a4(q, asc, lay["qweight"], lay["wscales"], dtype=dtype, bias=lay["bias"], lora_act_in=la, lora_up=lay["proj_up"],
                        fuse="gelu_quant", next_smooth=lay2["smooth"], next_lora_down=lay2["proj_down"], envelope=True)
        d[f"{t}.gemm.gelu_quant.qout"], d[f"{t}.gemm.gelu_quant.oscales"], d[f"{t}.gemm.gelu_quant.lora_act_out"] = r["qout"], r["oscales"], r["lora_act_out"]
        d[f"{t}.gemm.gelu_quant.env.q_lo"], d[f"{t}.gemm.gelu_quant.env.q_hi"] = r["envelope"]["q_lo"], r["envelope"]["q_hi"]
        d[f"{t}.mlp.out"] = O.fused_gelu_mlp(x, lay, lay2, dtype)
        # attention restatement (one head, 128 queries x 192 keys)
        rng = np.random.default_rng(500 + seed)
        aq = inp(f"{t}.att.q", lambda: O.round16(rng.standard_normal((64, 128)).astype(F32), dtype))
        ak = inp(f"{t}.att.k", lambda: O.round16(rng.standard_normal((192, 128)).astype(F32), dtype))
        av = inp(f"{t}.att.v", lambda: O.round16(rng.standard_normal((192, 128)).astype(F32), dtype))
        d[f"{t}.att.out"] = O.attention_tiled(aq, ak, av, 128 ** -0.5, dtype)
        # AWQ W4A16 GEMV
        rng = np.random.default_rng(700 + seed)
        w = inp(f"{t}.awq.w", lambda: (rng.standard_normal((64, 256)) * 0.05).astype(F32))
        aq4, asz, azz = O.awq_quantize_ref(w, dtype)
        ax = inp(f"{t}.awq.x", lambda: O.round16(rng.standard_normal((2, 256)).astype(F32), dtype))
        ab = inp(f"{t}.awq.bias", lambda: O.round16((rng.standard_normal(64) * 0.1).astype(F32), dtype))
        d[f"{t}.awq.codes"], d[f"{t}.awq.scales"], d[f"{t}.awq.zeros"] = aq4, asz, azz
        d[f"{t}.awq.out"] = O.awq_gemv_w4a16(ax, aq4, asz, azz, dtype, bias=ab)
        # element-wise glue
        rng = np.random.default_rng(900 + seed)
        res = inp(f"{t}.glue.res", lambda: O.round16(rng.standard_normal((24, 256)).astype(F32), dtype))
        a = inp(f"{t}.glue.a", lambda: O.round16(rng.standard_normal((24, 256)).astype(F32), dtype))
        b = inp(f"{t}.glue.b", lambda: O.round16(rng.standard_normal((24, 256)).astype(F32), dtype))
        g = inp(f"{t}.glue.gate", lambda: O.round16(rng.standard_normal(256).astype(F32), dtype))
        sc = inp(f"{t}.glue.scale", lambda: O.round16((1 + 0.2 * rng.standard_normal(256)).astype(F32), dtype))
        sh = inp(f"{t}.glue.shift", lambda: O.round16((0.2 * rng.standard_normal(256)).astype(F32), dtype))
        y = O.residual_gate_ref(res, a, g, b, dtype)
        st = O.ln_stats_ref(y)
        d[f"{t}.glue.y"], d[f"{t}.glue.stats"], d[f"{t}.glue.mod"] = y, st, O.ln_mod_ref(y, st, sc, sh, dtype)
    for k in list(d):
        if k not in inputs:   # outputs only: inputs are stored whole
            v = d[k]
            if v.ndim == 2 and v.shape[0] == 256:
                d[k] = v[:KEEP].copy()
            elif v.ndim == 2 and v.shape[1] == 256 and k.endswith(("ascales", "oscales", "s_lo", "s_hi")):
                d[k] = v[:, :KEEP].copy()
    return d


# outputs whose last fp32 bit may depend on the BLAS summation order of the host (float64 / float32 matrix products feeding an fp32 result):
# held to one fp32 ulp instead of bit equality; everything else -- codes, scales, every 16-bit output -- is compared bit for bit
FP32_ULP_KEYS = ("quant.lora_act", "gelu_quant.lora_act_out", "glue.stats")
# the attention restatement multiplies in float32 (numpy sgemm): its 16-bit output may move by one 16-bit ulp on a few elements between hosts
ULP16_KEYS = ("att.out",)


def compare(name, got, want):
    """'' if equal under the rule for `name`, else a message"""
    got, want = np.asarray(got), np.asarray(want)
    if got.shape != want.shape or got.dtype != want.dtype:
        return f"{name}: shape/dtype {got.shape}/{got.dtype} != {want.shape}/{want.dtype}"
    if any(name.endswith(k) for k in FP32_ULP_KEYS):
        ulp = np.spacing(np.abs(want).astype(F32))
        bad = np.abs(got.astype(np.float64) - want.astype(np.float64)) > ulp
        return f"{name}: {int(bad.sum())} values beyond one fp32 ulp" if bad.any() else ""
    if any(name.endswith(k) for k in ULP16_KEYS):
        dt = name.split(".")[0]
        step = np.maximum(np.abs(want) * (2.0 ** -7 if dt == "bf16" else 2.0 ** -10), 1e-30)
        diff = np.abs(got - want)
        if (diff > 1.001 * step).any() or (diff > 0).mean() > 0.002:
            return f"{name}: {int((diff > 0).sum())} of {diff.size} differ, max {float((diff / step).max()):.2f} ulp16"
        return ""
    same = np.array_equal(got, want, equal_nan=True) if got.dtype.kind == "f" else np.array_equal(got, want)
    return "" if same else f"{name}: {int((got != want).sum())} of {got.size} elements differ"


def main():
    check = "--check" in sys.argv
    rc = 0
    for seed in (0, 1):
        path = os.path.join(OUT, f"oracle_kat_seed{seed}.npz")
        if check:
            stored = dict(np.load(path))
            d = compute(seed, stored)
            msgs = [m for k in stored if (m := compare(k, d[k], stored[k]))]
            print(f"seed {seed}: {len(stored)} entries, {len(msgs)} mismatches")
            for m in msgs:
                print("  ", m)
            rc |= bool(msgs)
        else:
            d = compute(seed)
            np.savez_compressed(path, **d)
            print(path, f"{os.path.getsize(path) / 1e3:.0f} KB, {len(d)} entries")
    return rc


if __name__ == "__main__":
    sys.exit(main())
