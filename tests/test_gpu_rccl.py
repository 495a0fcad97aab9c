"""RCCL leg of the replica path: two ranks on two GPUs, bucketed weight broadcast over "nccl" (= RCCL on ROCm), then each
rank runs its own forward.  Skipped on a 1-GPU box (the gloo twin, tests/test_gpu_replica.py / test_replica.py, always runs)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    from nunchaku_amd import replica
    from nunchaku_amd.models.flux import FluxTransformerAMD

    torch.cuda.set_device(rank)
    replica.init_process_group("nccl")
    dev = torch.device("cuda", rank)
    model = FluxTransformerAMD(num_layers=1, num_single_layers=1, dim=256, heads=2, in_channels=64, joint_attention_dim=128,
                               pooled_projection_dim=64, device=dev)
    if rank == 0:
        model.init_synthetic_(seed=0)  # repacked on the source: receivers must adopt shapes, data and layout masks
    nbytes = replica.broadcast_module_(model, src=0, bucket_bytes=1 << 20)
    model.eval()
    g = torch.Generator(device=dev).manual_seed(7)  # same inputs on both ranks: outputs must agree bit for bit
    lat = torch.randn(1, 256, 64, device=dev, generator=g).bfloat16()
    enc = torch.randn(1, 128, 128, device=dev, generator=g).bfloat16()
    pooled = torch.randn(1, 64, device=dev, generator=g).bfloat16()
    ids = torch.zeros(256, 3, device=dev)
    from nunchaku_amd import mode

    with torch.no_grad(), mode.deterministic_mode():  # fixed-point low-rank sums: bit-equality across ranks is legitimate
        y = model(lat, enc, pooled, torch.tensor([0.5], device=dev), ids, torch.zeros(128, 3, device=dev), torch.tensor([3.5], device=dev))
    torch.cuda.synchronize()
    import hashlib

    out.put((rank, nbytes, hashlib.sha256(y.float().cpu().numpy().tobytes()).hexdigest(), bool(torch.isfinite(y.float()).all())))
    replica.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI); the driver's 8-GPU node runs it")
def test_rccl_broadcast_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, n0, s0, f0), (_, n1, s1, f1) = res
    assert n0 == n1 > 0 and f0 and f1 and s0 == s1
