"""Host-side replay of the GEMM's persistent / stream-K schedule (the same GemmSchedule code the kernel runs,
compiled for the host in libsvdq_amd.so): coverage and ownership invariants on CPU, no GPU needed."""
import ctypes as C

import numpy as np
import pytest

from nunchaku_amd import _lib


def schedule(M_pad, N, K, cus, ws):
    lib = _lib.load()
    n = lib.svdq_gemm_schedule(M_pad, N, K, cus, ws, None, 0)
    assert n >= 0
    buf = (C.c_int32 * (6 * max(n, 1)))()
    assert lib.svdq_gemm_schedule(M_pad, N, K, cus, ws, buf, n) == n
    return np.frombuffer(buf, dtype=np.int32)[: 6 * n].reshape(n, 6).copy()


SHAPES = [(4096, 3072, 3072), (4096, 9216, 3072), (4096, 3072, 12288), (4608, 3072, 12288), (4608, 12288, 3072),
          (512, 3072, 3072), (512, 3072, 12288), (256, 128, 128), (256, 384, 1024), (2560, 1152, 2048)]


@pytest.mark.parametrize("M_pad,N,K", SHAPES)
@pytest.mark.parametrize("cus,ws", [(256, 0), (256, 1), (64, 1), (24, 1), (8, 1)])
def test_every_k_step_of_every_tile_is_computed_exactly_once(M_pad, N, K, cus, ws):
    seg = schedule(M_pad, N, K, cus, ws)
    tiles, KP = (M_pad // 256) * (N // 128), K // 128
    cover = np.zeros((tiles, KP), np.int32)
    for pos, tile, k0, k1, slot, contrib in seg:
        assert 0 <= tile < tiles and 0 <= k0 < k1 <= KP
        cover[tile, k0:k1] += 1
    assert (cover == 1).all()
    # exactly one owner (the segment that reaches KP) per tile; it waits for all the other segments of its tile
    for t in range(tiles):
        parts = seg[seg[:, 1] == t]
        owners = parts[parts[:, 3] == KP]
        assert len(owners) == 1
        assert owners[0, 5] == len(parts) - 1
        # contributors sit at lower positions than the owner (they run concurrently: one workgroup per CU)
        assert (parts[parts[:, 3] < KP][:, 0] < owners[0, 0]).all()
    # partial slots are unique and below 2 * grid
    slots = seg[seg[:, 4] >= 0][:, 4]
    assert len(set(slots.tolist())) == len(slots)
    grid = seg[:, 0].max() + 1
    assert grid <= max(cus, 1) and (slots < 2 * grid).all() if len(slots) else True


def test_stream_k_only_where_it_pays():
    # long K with a half-empty last round: split; short K: whole tiles (split overhead ~10 K-steps)
    long_k = schedule(4096, 3072, 12288, 256, 1)
    assert (long_k[:, 3] - long_k[:, 2] < 96).any()
    short_k = schedule(4096, 3072, 3072, 256, 1)
    assert ((short_k[:, 2] == 0) & (short_k[:, 3] == 24)).all()
    # a workgroup publishes the head of the next tile before it takes up its own owner duty
    for pos in np.unique(long_k[:, 0]):
        mine = long_k[long_k[:, 0] == pos]
        partial = mine[(mine[:, 2] > 0) | (mine[:, 3] < 96)]
        if len(partial) == 2:
            assert partial[0, 3] < 96 and partial[1, 3] == 96


def test_invalid_shapes_are_rejected():
    lib = _lib.load()
    assert lib.svdq_gemm_schedule(100, 128, 128, 256, 0, None, 0) == -1
    assert lib.svdq_gemm_schedule(256, 128, 64, 256, 0, None, 0) == -1
