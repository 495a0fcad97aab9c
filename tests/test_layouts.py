"""Layout codecs of the oracle vs fixtures produced by the REFERENCE's packer (tools/make_golden.py)."""

import numpy as np
import pytest

from oracle import svdq_oracle as O


@pytest.mark.parametrize("name", ["qweight_128x128", "qweight_256x384"])
def test_qweight_codec(golden_dir, name):
    d = np.load(f"{golden_dir}/{name}.npz")
    assert np.array_equal(O.unpack_qweight_ref(d["packed"]), d["logical"])
    assert np.array_equal(O.pack_qweight_ref(d["logical"]), d["packed"])


@pytest.mark.parametrize("name", ["wscales_2x128", "wscales_6x256"])
def test_wscales_codec(golden_dir, name):
    d = np.load(f"{golden_dir}/{name}.npz")
    assert np.array_equal(O.unpack_wscales_ref(d["packed"]), d["logical"])
    assert np.array_equal(O.pack_wscales_ref(d["logical"]), d["packed"])


def test_vec_codec(golden_dir):
    d = np.load(f"{golden_dir}/vec_256.npz")
    assert np.array_equal(O.unpack_vec_ref(d["packed"]), d["logical"])
    assert np.array_equal(O.pack_vec_ref(d["logical"]), d["packed"])


def test_lowrank_codec(golden_dir):
    d = np.load(f"{golden_dir}/lowrank_128_192_32.npz")
    assert np.array_equal(O.unpack_lowrank_ref(d["up_packed"], False), d["up_logical"])
    assert np.array_equal(O.pack_lowrank_ref(d["up_logical"], False), d["up_packed"])
    assert np.array_equal(O.unpack_lowrank_ref(d["down_packed"], True), d["down_logical"])
    assert np.array_equal(O.pack_lowrank_ref(d["down_logical"], True), d["down_packed"])


def test_rotemb_codec(golden_dir):
    d = np.load(f"{golden_dir}/rotemb_32.npz")
    assert np.array_equal(O.unpack_rotemb_ref(d["packed"]), d["logical"])
    assert np.array_equal(O.pack_rotemb_ref(d["logical"]), d["packed"])


def test_bf16_round_matches_torch():
    import torch

    rng = np.random.default_rng(0)
    x = (rng.standard_normal(100000) * np.exp(rng.uniform(-20, 20, 100000))).astype(np.float32)
    x[:8] = [0.0, -0.0, 1.0, 1.00390625, 1.01171875, 3.3895314e38, 1e-40, -1e-40]
    ref = torch.from_numpy(x).to(torch.bfloat16).float().numpy()
    assert np.array_equal(O.bf16_round(x), ref)
    bits = O.to_bits16(O.bf16_round(x), "bf16")
    assert np.array_equal(O.from_bits16(bits, "bf16"), ref)
