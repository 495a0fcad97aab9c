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


def test_packed_fp16_dequant_of_the_gemv_kernel_is_the_oracles():
    """The fp16 M = 1 path of gemv_awq.hip never converts a nibble: it ORs the code PAIR of two adjacent channels into
    0x6400 6400 (= 1024 + q, or 1024 + 16 q for the odd nibbles), steps back to q with one packed fp16 operation and feeds
    v_pk_fma_f16.  Restated here in numpy float16 on the checkpoint words: every step before the fma is exact, the pairs
    are the channels (8e + 2i, 8e + 2i + 1), and the fma result is the oracle's w16 -- for all 16 codes."""
    rng = np.random.default_rng(5)
    N, K = 8, 128
    codes = rng.integers(0, 16, size=(N, K)).astype(np.uint8)
    codes[0, :16] = np.arange(16)  # every code value appears
    s = O.round16((rng.random((K // 64, N)).astype(np.float32) + 0.5) * 0.01, "fp16")
    z = O.round16(-7.5 * s, "fp16")
    words = O.pack_awq_w4_ref(codes).view(np.uint32)  # [N/4, K/2]: a row group of 4 channels is 2K bytes
    f16 = np.float16
    got = np.zeros((N, K), np.float32)
    for rg in range(N // 4):
        w = words[rg].reshape(K // 64, 8, 4)  # [chunk][lane-in-chunk = 2*row + half][dword i]: one 16-byte piece per lane
        for c in range(K // 64):
            for row in range(4):
                for half in range(2):
                    n = rg * 4 + row
                    for i in range(4):
                        word = int(w[c, 2 * row + half, i])
                        up = word >> 8
                        for e in range(4):
                            src = word if e < 2 else up
                            t = np.array([(src & (0x00f000f0 if e & 1 else 0x000f000f)) | 0x64006400], np.uint32).view(f16)
                            with np.errstate(over="raise", invalid="raise"):
                                q = t * f16(1 / 16) - f16(64) if e & 1 else t - f16(1024)  # exact: small integers
                            k = c * 64 + half * 32 + 8 * e + 2 * i
                            assert np.array_equal(q.astype(np.int64), codes[n, k:k + 2]), (rg, c, row, half, i, e)
                            w16 = (q.astype(np.float64) * float(s[c, n]) + float(z[c, n])).astype(f16)  # one rounding
                            got[n, k:k + 2] = w16
    ref = O.round16(codes.astype(np.float64) * np.repeat(s.T.astype(np.float64), 64, 1) + np.repeat(z.T.astype(np.float64), 64, 1), "fp16")
    assert np.array_equal(got, ref)
