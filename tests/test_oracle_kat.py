"""The oracle against its own committed known-answer vectors (tests/golden/oracle_kat_seed*.npz, tools/make_oracle_kat.py).

The reference holds no kernel-level vectors for this path: the oracle's arithmetic is parity-unpinned against
the reference.  These fixtures pin it against ITSELF: an edit of oracle/svdq_oracle.py that moves any result fails here and shows up
in review as a regenerated fixture, so oracle and kernels cannot drift together unnoticed.  Codes, scales and every 16-bit output are
compared bit for bit; fp32 sums whose last bit depends on the host BLAS's summation order are held to one fp32 ulp, the attention
restatement (float32 matrix products) to one 16-bit ulp on <= 0.2 % of its elements (tools/make_oracle_kat.py: FP32_ULP_KEYS / ULP16_KEYS).
"""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
spec = importlib.util.spec_from_file_location("make_oracle_kat", os.path.join(ROOT, "tools", "make_oracle_kat.py"))
kat = importlib.util.module_from_spec(spec)
spec.loader.exec_module(kat)


@pytest.mark.parametrize("seed", [0, 1])
def test_oracle_reproduces_its_known_answers(seed):
    stored = dict(np.load(os.path.join(ROOT, "tests", "golden", f"oracle_kat_seed{seed}.npz")))
    got = kat.compute(seed, stored)
    assert set(got) == set(stored)
    msgs = [m for k in sorted(stored) if (m := kat.compare(k, got[k], stored[k]))]
    assert not msgs, "\n".join(msgs)


def test_known_answers_cover_every_operator():
    stored = np.load(os.path.join(ROOT, "tests", "golden", "oracle_kat_seed0.npz"))
    names = set(stored.files)
    for dt in ("bf16", "fp16"):
        for key in ("quant.codes", "quant.ascales", "quant.lora_act", "quant_glu.codes", "gemm.none.fp32.out", "gemm.none.ref16.out",
                    "gemm.silu.fp32.out", "gemm.rmsnorm_rope.out", "gemm.gelu_quant.qout", "gemm.gelu_quant.oscales",
                    "gemm.gelu_quant.lora_act_out", "gemm.lora_scales_nobias.out", "mlp.out", "att.out", "awq.out", "glue.y", "glue.stats", "glue.mod",
                    "quant.env.q_lo", "quant.env.q_hi", "quant.env.s_lo", "quant.env.s_hi", "gemm.gelu_quant.env.q_lo", "gemm.gelu_quant.env.q_hi"):
            assert f"{dt}.{key}" in names, f"{dt}.{key}"
    # the vectors are not degenerate: codes use the whole range, outputs are finite and non-constant
    assert stored["bf16.quant.codes"].min() == -7 and stored["bf16.quant.codes"].max() == 7
    assert stored["bf16.gemm.gelu_quant.qout"].max() == 15
    for k in names:
        if stored[k].dtype.kind == "f":
            assert np.isfinite(stored[k]).all(), k
