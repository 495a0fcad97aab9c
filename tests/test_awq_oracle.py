"""AWQ W4A16 oracle pieces against the fixtures made by the reference's own tinychat converter
(tools/make_golden.py -> tests/golden/awq_*.npz; reference: text_encoders/tinychat_utils.py:76-188)."""
import numpy as np
import pytest

from oracle import svdq_oracle as O


@pytest.mark.parametrize("name", ["awq_16x128", "awq_64x256"])
def test_pack_matches_reference_converter(golden_dir, name):
    g = np.load(f"{golden_dir}/{name}.npz")
    codes, packed16 = g["codes"], g["packed_int16"]
    N, K = codes.shape
    ours = O.pack_awq_w4_ref(codes)  # int32 [N/4, K/2], the AWQW4A16Linear.qweight parameter
    assert ours.shape == (N // 4, K // 2) and ours.dtype == np.int32
    assert np.array_equal(ours.view(np.int16).reshape(N // 4, K), packed16)
    assert np.array_equal(O.unpack_awq_w4_ref(ours), codes)


@pytest.mark.parametrize("name", ["awq_16x128", "awq_64x256"])
def test_quantiser_and_dequant_match_reference_converter(golden_dir, name):
    g = np.load(f"{golden_dir}/{name}.npz")
    q, s, z = O.awq_quantize_ref(g["weight"], "bf16")
    G = g["weight"].shape[1] // 64  # the converter pads the group axis (ceil_num_groups, tinychat_utils.py:31-74)
    assert np.array_equal(s, g["scales"][:G]) and np.array_equal(z, g["zeros"][:G])
    assert not g["scales"][G:].any() and not g["zeros"][G:].any()
    assert np.array_equal(q, g["codes"])
    # q*scale + scaled_zero reproduces the weight to within half a step
    w = q.astype(np.float64) * np.repeat(s.T, 64, 1) + np.repeat(z.T, 64, 1)
    assert np.abs(w - g["weight"]).max() <= 0.5 * s.max() + 2.0 ** -7 * np.abs(g["weight"]).max()


def test_gemv_oracle_against_dense():
    rng = np.random.default_rng(0)
    N, K, B = 32, 256, 3
    w = O.round16(rng.standard_normal((N, K)).astype(np.float32) * 0.05, "bf16")
    q, s, z = O.awq_quantize_ref(w, "bf16")
    x = O.round16(rng.standard_normal((B, K)).astype(np.float32), "bf16")
    bias = O.round16(rng.standard_normal(N).astype(np.float32), "bf16")
    y = O.awq_gemv_w4a16(x, q, s, z, "bf16", bias=bias)
    wd = q.astype(np.float64) * np.repeat(s.T, 64, 1) + np.repeat(z.T, 64, 1)
    ref = x.astype(np.float64) @ wd.T + bias
    assert y.shape == (B, N)
    assert np.abs(y - ref).max() <= 2.0 ** -6 * np.abs(ref).max() + 0.02
