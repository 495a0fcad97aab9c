"""N>1 path (independent replicas + one weight broadcast) on CPU with gloo, world_size 2."""

import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from nunchaku_amd import replica
    from nunchaku_amd.models.linear import SVDQW4A4Linear

    r, lr, w = replica.init_process_group("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)  # different garbage on every rank before the broadcast
    lin = SVDQW4A4Linear(128, 256, rank=16, device="cpu")
    with torch.no_grad():
        lin.qweight.copy_(torch.randint(-128, 128, lin.qweight.shape, dtype=torch.int8))
        for p in (lin.wscales, lin.bias, lin.smooth_factor, lin.smooth_factor_orig, lin.proj_down, lin.proj_up):
            p.copy_(torch.randn(p.shape))
    lin._amd_layout = rank == 0
    if rank == 0:  # the source has already been repacked: qweight is the [out, 3*in/4] FP6 image, the receivers' is [out, in/2]
        lin.qweight.data = torch.randint(-128, 128, (256, 96), dtype=torch.int8)
    nbytes = replica.broadcast_module_(lin, src=0, bucket_bytes=20000)  # several buckets: 24 KB image, 2 KB vectors
    assert tuple(lin.qweight.shape) == (256, 96), "receivers must take the source's (repacked) shape"
    digest = torch.cat([p.detach().view(-1).view(torch.uint8).to(torch.int64) for p in lin.parameters()]).sum().item()
    slow = replica.max_over_ranks(1.0 + rank, "cpu")
    units = replica.shard_units(7, rank, world)
    replica.barrier()
    out.put((rank, nbytes, digest, bool(lin._amd_layout), slow, units))
    dist.destroy_process_group()


def test_broadcast_shard_and_timing_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, n0, d0, f0, s0, u0), (r1, n1, d1, f1, s1, u1) = res
    assert n0 == n1 > 0
    assert d0 == d1, "parameters differ after the broadcast"
    assert f0 and f1, "layout flag must follow the data"
    assert s0 == s1 == 2.0, "timing must be the max over ranks"
    assert u0 == [0, 2, 4, 6] and u1 == [1, 3, 5]  # every image exactly once, no overlap


def test_single_process_is_a_noop():
    from nunchaku_amd import replica

    lin = torch.nn.Linear(4, 4)
    assert replica.broadcast_module_(lin) == 0
    assert replica.max_over_ranks(3.5, "cpu") == 3.5
    assert replica.shard_units(5, 0, 1) == [0, 1, 2, 3, 4]
