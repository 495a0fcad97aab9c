"""Data-parallel replicas: the only multi-GPU mode of this path.

Each image (prompt, seed, latent) is an independent unit of work, so images are sharded
round-robin over one process per GPU and every denoise loop runs without any collective.  The
reference does the same by hand (app/flux.1/t2i/evaluate.py:30-39,68-69 ``--chunk-start/-step``).
RCCL (torch.distributed "nccl" on ROCm, over xGMI) is used only for the one-time broadcast of the
model parameters from rank 0 and for reducing timings at the end.
"""

from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_process_group(backend: str | None = None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun contract)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_units(n_units: int, rank: int, world: int) -> list[int]:
    """Indices of the images this rank owns: i with i % world == rank."""
    return list(range(rank, n_units, world))


@torch.no_grad()
def broadcast_module_(module: torch.nn.Module, src: int = 0) -> int:
    """Broadcast every parameter and buffer of ``module`` from ``src`` in place; returns the bytes
    sent.  Tensors are sent as raw bytes so packed int4/uint8 parameters travel unchanged."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    total = 0
    tensors = list(module.parameters()) + list(module.buffers())
    dev = tensors[0].device
    # shapes first: load-time repacks change them (SVDQW4A4Linear.repack_(): qweight [out, in/2] -> [out, 3*in/4]), and the
    # source may already be in the kernel layout while the receivers still hold freshly constructed parameters
    shapes = torch.zeros(len(tensors), 9, dtype=torch.int64, device=dev)
    if dist.get_rank() == src:
        for i, t in enumerate(tensors):
            if t.dim() > 8:
                raise ValueError("broadcast_module_: tensors of more than 8 dimensions are not supported")
            shapes[i, 0] = t.dim()
            shapes[i, 1:1 + t.dim()] = torch.tensor(list(t.shape), dtype=torch.int64)
    dist.broadcast(shapes, src=src)
    shapes = shapes.cpu().tolist()
    for t, row in zip(tensors, shapes):
        shape = tuple(int(v) for v in row[1:1 + int(row[0])])
        if tuple(t.shape) != shape:
            t.data = torch.empty(shape, dtype=t.dtype, device=t.device)
        flat = t.data.contiguous().view(-1).view(torch.uint8)
        dist.broadcast(flat, src=src)
        if not t.data.is_contiguous():
            t.data.copy_(flat.view(t.dtype).view(t.shape))
        total += flat.numel()
    # layout flags are python attributes, not tensors: make them agree too
    flags = [int(getattr(m, "_amd_layout", False)) for m in module.modules() if hasattr(m, "_amd_layout")]
    if flags:
        ft = torch.tensor(flags, dtype=torch.int32, device=next(module.parameters()).device)
        dist.broadcast(ft, src=src)
        for m, f in zip([m for m in module.modules() if hasattr(m, "_amd_layout")], ft.tolist()):
            m._amd_layout = bool(f)
    return total


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
