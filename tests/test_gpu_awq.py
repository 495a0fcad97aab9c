"""GPU parity of the AWQ W4A16 GEMV against the numpy oracle.

Tolerance (floating point, stated as the task demands): the 16-bit roundings of the dequantised weight and
of every product are reproduced exactly; only the fp32 summation ORDER differs (the reference's own order is
thread-layout dependent, gemv_awq.cu:225-236), so outputs may differ by one 16-bit ulp where the fp32 sum
lands next to a rounding boundary: |err| <= 1 ulp, and <= 2 % of the elements may differ at all."""
import numpy as np
import pytest
import torch

from oracle import svdq_oracle as O
from tests.helpers import TORCH_DT, assert_close_16, f32, t16

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _need_gpu(built_lib):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def _layer(N, K, dtype, seed):
    rng = np.random.default_rng(seed)
    w = O.round16(rng.standard_normal((N, K)).astype(np.float32) * 0.05, dtype)
    q, s, z = O.awq_quantize_ref(w, dtype)
    bias = O.round16(rng.standard_normal(N).astype(np.float32) * 0.1, dtype)
    return q, s, z, bias


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("m,N,K", [(1, 64, 128), (3, 256, 576), (8, 128, 1024), (1, 1536, 3072), (1, 18432, 3072)])  # last: FLUX.1 size
def test_gemv_awq_matches_oracle(dtype, m, N, K):
    from nunchaku_amd.models.linear import AWQW4A16Linear

    q, s, z, bias = _layer(N, K, dtype, seed=N + K + m)
    rng = np.random.default_rng(7)
    x = O.round16(rng.standard_normal((m, K)).astype(np.float32), dtype)
    lin = AWQW4A16Linear(K, N, torch_dtype=TORCH_DT[dtype], device="cuda")
    lin.load_state_dict({"qweight": torch.from_numpy(O.pack_awq_w4_ref(q)), "wscales": t16(s, dtype), "wzeros": t16(z, dtype),
                         "bias": t16(bias, dtype)})
    y = lin(t16(x, dtype))
    assert y.shape == (m, N) and y.dtype == TORCH_DT[dtype]
    ref = O.awq_gemv_w4a16(x, q, s, z, dtype, bias=bias)
    got = f32(y)
    assert (got != ref).mean() <= 0.02
    assert_close_16(got, ref, dtype, "gemv_awq")
    if N % 6 == 0:  # de-interleaved output (the modulation vectors of AdaLayerNormZero): a pure permutation of the columns
        lin.out_chunks = 6
        y6 = lin(t16(x, dtype))
        assert torch.equal(y6.view(m, 6, N // 6), y.view(m, N // 6, 6).transpose(1, 2))


def test_gemv_awq_op_surface_and_errors():
    from nunchaku_amd._C import ops
    from nunchaku_amd.ops.gemv import awq_gemv_w4a16_cuda

    q, s, z, _ = _layer(64, 128, "bf16", 1)
    x = t16(np.ones((2, 128), np.float32), "bf16")
    kern = torch.from_numpy(O.pack_awq_w4_ref(q)).cuda()
    y = awq_gemv_w4a16_cuda(x, kern, t16(s, "bf16"), t16(z, "bf16"), 2, 64, 128)  # no bias: the reference op
    ref = O.awq_gemv_w4a16(np.ones((2, 128), np.float32), q, s, z, "bf16")
    assert_close_16(f32(y), ref, "bf16", "gemv_awq(no bias)")
    with pytest.raises(NotImplementedError):
        ops.gemv_awq(x, kern, t16(s, "bf16"), t16(z, "bf16"), 2, 64, 128, 128)
    with pytest.raises(ValueError):
        ops.gemv_awq(x.repeat(5, 1)[:9], kern, t16(s, "bf16"), t16(z, "bf16"), 9, 64, 128, 64)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_gemv_awq_batched_matches_single_launches(dtype):
    from nunchaku_amd.models.linear import AWQW4A16Linear
    from nunchaku_amd.ops.gemv import awq_gemv_w4a16_batched

    K = 256
    layers = []
    for i, (N, chunks) in enumerate([(768, 6), (384, 3), (64, 1), (1536, 6)] * 21):  # 84 layers: two launches
        q, s, z, bias = _layer(N, K, dtype, seed=100 + i % 4)
        lin = AWQW4A16Linear(K, N, torch_dtype=TORCH_DT[dtype], device="cuda")
        lin.load_state_dict({"qweight": torch.from_numpy(O.pack_awq_w4_ref(q)), "wscales": t16(s, dtype), "wzeros": t16(z, dtype),
                             "bias": t16(bias, dtype)})
        lin.out_chunks = chunks
        layers.append(lin)
    x = t16(np.random.default_rng(1).standard_normal((1, K)).astype(np.float32), dtype)
    outs = awq_gemv_w4a16_batched(x, layers)
    assert len(outs) == len(layers)
    for lin, o in list(zip(layers, outs))[::7]:
        assert torch.equal(o, lin(x))


def test_gemv_awq_fp16_packed_path_keeps_the_rounding_points_at_the_edges():
    """M = 1, fp16 runs on the packed 16-bit pipe (v_pk_fma_f16 / v_pk_mul_f16).  Scales and activations chosen so that
    dequantised weights and products land in fp16's subnormal range and next to its overflow threshold: the per-element
    roundings (gemv_awq.cu:192-236) must still be the oracle's."""
    from nunchaku_amd.models.linear import AWQW4A16Linear

    N, K = 64, 256
    rng = np.random.default_rng(11)
    q = rng.integers(0, 16, size=(N, K)).astype(np.uint8)
    s = np.empty((K // 64, N), np.float32)
    s[0], s[1], s[2], s[3] = 2.0 ** -12, 2.0 ** -20, 0.37, 6.0
    s = O.round16(s * (1 + rng.random((K // 64, N)).astype(np.float32) * 0.3), "fp16")
    z = O.round16(-7.5 * s * (1 + 0.1 * rng.standard_normal((K // 64, N)).astype(np.float32)), "fp16")
    lin = AWQW4A16Linear(K, N, bias=False, torch_dtype=torch.float16, device="cuda")
    lin.load_state_dict({"qweight": torch.from_numpy(O.pack_awq_w4_ref(q)), "wscales": t16(s, "fp16"), "wzeros": t16(z, "fp16")},
                        strict=False)
    x = rng.standard_normal((1, K)).astype(np.float32)
    x[:, 0:64] *= 2.0 ** -6     # products around 2^-18 .. 2^-15: subnormal fp16
    x[:, 64:128] *= 2.0 ** -3   # products around 2^-24: the last subnormal steps, or zero
    x[:, 200] = 500.0           # |w| up to ~60: one product per row of up to ~3e4 (fp16 spacing 16-32 there), sums in fp32
    small = x.copy()
    small[:, 128:] = 0.0        # outputs of ~2^-13: made of subnormal weights and subnormal products only
    for xv in (O.round16(small, "fp16"), O.round16(x, "fp16")):
        got = f32(lin(t16(xv, "fp16")))
        ref = O.awq_gemv_w4a16(xv, q, s, z, "fp16")
        assert np.isfinite(ref).all() and np.abs(ref).max() > 0
        assert (got != ref).mean() <= 0.05
        ulp = np.maximum(np.abs(ref) * 2.0 ** -10, 2.0 ** -24)  # one fp16 step at ref (2^-24 in the subnormal range)
        assert (np.abs(got - ref) <= ulp).all(), "gemv_awq fp16 edges: more than one 16-bit step from the oracle"
