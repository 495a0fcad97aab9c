"""Replica path with REAL repacked models: two processes on cuda:0 (gloo carries the CUDA tensors; RCCL refuses two ranks
on one device), rank 0 initialises and repacks, the broadcast must bring rank 1 to the same kernel-layout model."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    from nunchaku_amd import replica
    from nunchaku_amd.models.flux import FluxTransformerAMD

    replica.init_process_group("gloo")
    torch.cuda.set_device(0)
    model = FluxTransformerAMD(num_layers=1, num_single_layers=1, dim=256, heads=2, in_channels=64, joint_attention_dim=128,
                               pooled_projection_dim=64, device="cuda")
    if rank == 0:
        model.init_synthetic_(seed=0)  # repacks: qweight becomes the [out, 3*in/4] FP6 image on this rank only
    nbytes = replica.broadcast_module_(model, src=0)
    model.eval()
    g = torch.Generator(device="cuda").manual_seed(1)
    side, t_txt = 16, 256
    lat = torch.randn(1, side * side, 64, device="cuda", generator=g).bfloat16()
    enc = torch.randn(1, t_txt, 128, device="cuda", generator=g).bfloat16()
    pooled = torch.randn(1, 64, device="cuda", generator=g).bfloat16()
    img_ids = torch.zeros(side * side, 3, device="cuda")
    img_ids[:, 1] = torch.arange(side, device="cuda").repeat_interleave(side)
    img_ids[:, 2] = torch.arange(side, device="cuda").repeat(side)
    from nunchaku_amd import mode

    with torch.no_grad(), mode.deterministic_mode():  # fixed-point low-rank sums: replicas must agree BIT FOR BIT
        y = model(lat, enc, pooled, torch.tensor([0.5], device="cuda"), img_ids, torch.zeros(t_txt, 3, device="cuda"),
                  torch.tensor([3.5], device="cuda")).float()
    torch.cuda.synchronize()
    out.put((rank, nbytes, y.cpu().numpy()))  # by value: a torch tensor travels as a shared-memory handle that dies with this process
    replica.barrier()
    dist.destroy_process_group()


def test_two_replicas_agree_after_the_weight_broadcast(built_lib):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, n0, y0), (_, n1, y1) = res
    y0, y1 = torch.from_numpy(y0), torch.from_numpy(y1)
    assert n0 == n1 > 0 and torch.isfinite(y0).all()
    assert torch.equal(y0, y1), "same weights, same inputs, deterministic mode: the two replicas must agree bit for bit"
