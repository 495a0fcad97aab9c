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


BUCKET_BYTES = 256 << 20  # one collective per ~256 MB: xGMI rings are per-link bound, few large broadcasts beat ~2000 small ones


@torch.no_grad()
def broadcast_module_(module: torch.nn.Module, src: int = 0, bucket_bytes: int = BUCKET_BYTES) -> int:
    """Broadcast every parameter and buffer of ``module`` from ``src`` in place; returns the bytes sent.

    Tensors travel as raw bytes (packed int4 / FP6-image parameters unchanged), coalesced into flat buckets of
    ``bucket_bytes``: a FLUX.1 replica (~9 GB, ~2000 tensors) is ~36 collectives plus one for the shapes and one for the
    per-tensor layout masks of the SVDQuant layers."""
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
    is_src = dist.get_rank() == src

    def flush(group):
        nonlocal total
        if not group:
            return
        sizes = [t.numel() * t.element_size() for t in group]
        # every piece starts 16-byte aligned inside the bucket (views of other dtypes need it)
        offs, n = [], 0
        for sz in sizes:
            offs.append(n)
            n += (sz + 15) // 16 * 16
        buf = torch.empty(n, dtype=torch.uint8, device=dev)
        if is_src:
            for t, o, sz in zip(group, offs, sizes):
                buf[o:o + sz].copy_(t.data.contiguous().view(-1).view(torch.uint8))
        dist.broadcast(buf, src=src)
        if not is_src:
            for t, o, sz in zip(group, offs, sizes):
                t.data.copy_(buf[o:o + sz].view(t.dtype).view(t.shape))
        total += sum(sizes)

    group, acc = [], 0
    for t in tensors:
        sz = t.numel() * t.element_size()
        if group and acc + sz > bucket_bytes:
            flush(group)
            group, acc = [], 0
        group.append(t)
        acc += sz
    flush(group)
    # which tensors of an SVDQuant layer hold the kernel layout is python state (SVDQW4A4Linear._amd_names): one bit mask
    # per layer, one collective
    layers = [m for m in module.modules() if hasattr(m, "_amd_names") and hasattr(m, "_LAYOUT_PARAMS")]
    if layers:
        masks = [sum(1 << i for i, n in enumerate(m._LAYOUT_PARAMS) if n in m._amd_names) for m in layers]
        mt = torch.tensor(masks, dtype=torch.int32, device=dev)
        dist.broadcast(mt, src=src)
        for m, mask in zip(layers, mt.tolist()):
            m._set_amd_names({n for i, n in enumerate(m._LAYOUT_PARAMS) if mask >> i & 1})
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
