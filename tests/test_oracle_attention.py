"""CPU checks of oracle.attention_tiled (the tile-by-tile restatement the GPU attention tests compare against): it must
itself be a correct softmax(QK^T/sqrt(d))V up to the 16-bit roundings it models, whatever the deferred-rescale threshold."""
import math

import numpy as np
import pytest

from oracle import svdq_oracle as O


def _inputs(L, d, dtype, seed, peaky):
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((L, d)).astype(np.float32)
    if peaky:
        q[: L // 2] *= 4.0
    k = rng.standard_normal((L, d)).astype(np.float32)
    v = rng.standard_normal((L, d)).astype(np.float32)
    return tuple(O.round16(t, dtype) for t in (q, k, v))


def _softmax_ref(q, k, v, scale):
    s = q.astype(np.float64) @ k.astype(np.float64).T * scale
    p = np.exp(s - s.max(axis=1, keepdims=True))
    p /= p.sum(axis=1, keepdims=True)
    return p @ v.astype(np.float64), p @ np.abs(v.astype(np.float64))


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("peaky", [False, True])
def test_tiled_restatement_is_a_softmax_attention(dtype, peaky):
    L, d = 256, 128
    q, k, v = _inputs(L, d, dtype, 3, peaky)
    scale = 1.0 / math.sqrt(d)
    out = O.attention_tiled(q, k, v, scale, dtype)
    ref, cond = _softmax_ref(q, k, v, scale)
    ulp = 2.0 ** -8 if dtype == "bf16" else 2.0 ** -11
    # probabilities rounded to 16 bits before the PV product and the output rounded once more: a few ulp of sum p |v|
    assert np.abs(out - ref).max() <= 3 * ulp * cond.max()
    assert (np.abs(out - ref) / cond).max() <= 3 * ulp


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_deferred_rescale_threshold_does_not_change_the_result(dtype):
    """Any reference point of the online softmax is valid as long as numerator and denominator share it: thresholds 0 (always
    rescale to the running maximum) and 8 (the kernel's) agree within 2.5 ulp (16-bit) of sum p |v|, 99 % within one, on peaky rows."""
    L, d = 384, 128
    q, k, v = _inputs(L, d, dtype, 5, True)
    scale = 1.0 / math.sqrt(d)
    a = O.attention_tiled(q, k, v, scale, dtype, defer_log2=0.0)
    b = O.attention_tiled(q, k, v, scale, dtype, defer_log2=8.0)
    _, cond = _softmax_ref(q, k, v, scale)
    ulp = 2.0 ** -8 if dtype == "bf16" else 2.0 ** -11
    assert (np.abs(a - b) / cond).max() <= 2.5 * ulp and (np.abs(a - b) / cond > ulp).mean() < 1e-2
    # a row dominated by ONE key returns that key's value exactly (its probability's rounding cancels between O and l)
    q2 = q.copy()
    q2[0] = O.round16(k[7] * 40.0, dtype)
    out = O.attention_tiled(q2[:32], k, v, scale, dtype)
    assert np.array_equal(out[0], v[7])
