"""Layer-wise host offload with a two-slot ping-pong on two HIP streams (SURVEY.md section 8 row f2 / BASELINE config 5).

Reference: ``CPUOffloadManager`` (nunchaku/models/utils.py:52-262) -- same constructor, attributes and methods
(``set_device``, ``load_block``, ``step``, ``get_block``, ``initialize``), same schedule: while block ``i`` computes on the
current stream, block ``i+1`` travels host -> device on ``memory_stream`` into the other buffer slot; two events order
"slot free" (compute of the previous tenant done) and "slot filled" (copy done).

MI355X specifics:
  * a device has 288 GB of HBM3E, so offload is never needed for the models of this package (Qwen-Image int4: ~11 GB); it
    exists for API parity and for co-locating many models on one GPU.  It is OFF unless ``set_offload(True)`` is called.
  * the PCIe Gen5 link (~50 GB/s), not the kernels, bounds an offloaded step: a Qwen-Image block computes in ~1.3 ms but
    weighs 113 MB as 4-bit nibbles and 170 MB as the FP6 operand image the GEMM reads.  The host copies therefore keep
    ``qweight`` in the CHECKPOINT (nibble) layout -- two thirds of the bytes on the link -- and the expansion to the FP6
    image (``svdq_repack_qweight``, HBM-bound, ~0.1 ms per block) runs on the memory stream right behind the copy, into the
    buffer slot.  Every other tensor is converted once at ``set_device`` time and travels as is.
  * host memory is pinned (``hipHostMalloc`` through ``Tensor.pin_memory``): pageable copies would serialise on the
    staging buffer of the runtime and never overlap.
"""

from __future__ import annotations

import copy

import torch
from torch import nn

from .. import _lib
from .linear import SVDQW4A4Linear


def copy_params_into(src: nn.Module, dst: nn.Module, non_blocking: bool = True):
    """reference: nunchaku/utils.py:336-366 -- parameters and buffers of ``src`` into the same-structured ``dst``."""
    with torch.no_grad():
        for ps, pd in zip(src.parameters(), dst.parameters()):
            pd.copy_(ps, non_blocking=non_blocking)
        for bs, bd in zip(src.buffers(), dst.buffers()):
            bd.copy_(bs, non_blocking=non_blocking)


class CPUOffloadManager:
    def __init__(self, blocks: list[nn.Module], device: str | torch.device = torch.device("cuda"), use_pin_memory: bool = True,
                 on_gpu_modules: list[nn.Module] = [], num_blocks_on_gpu: int = 1, empty_cache_freq: int = 0):
        self.blocks = blocks
        self.use_pin_memory = use_pin_memory
        self.on_gpu_modules = on_gpu_modules
        self.num_blocks_on_gpu = num_blocks_on_gpu
        assert self.num_blocks_on_gpu > 0
        self.memory_stream = None  # created in set_device
        self.compute_done = torch.cuda.Event(blocking=False)
        self.memory_done = torch.cuda.Event(blocking=False)
        self.buffer_blocks: list[nn.Module] = []
        self._host_nibbles: list[dict] = []   # per block: SVDQ layer name -> pinned [N, K/2] int8 checkpoint-layout qweight
        self._staging: list[dict] = [{}, {}]  # per buffer slot: layer name -> device staging tensor of the same shape
        self.device = None
        self.set_device(device)
        self.current_block_idx = 0
        self.forward_counter = 0
        self.empty_cache_freq = empty_cache_freq

    # ------------------------------------------------------------------ placement
    def set_device(self, device: torch.device | str, force: bool = False):
        """Buffers and resident blocks to ``device``, the other blocks to (pinned) host memory.  SVDQuant layers are
        converted to the kernel layout HERE, once; the host copy of every offloaded ``qweight`` stays in nibble form."""
        if isinstance(device, str):
            device = torch.device(device)
        assert device.type == "cuda"
        if self.device == device and not force:
            return
        self.device = device
        self.memory_stream = torch.cuda.Stream(device=device)
        for module in self.on_gpu_modules:
            module.to(device)
        self._host_nibbles = [{} for _ in self.blocks]
        for i, block in enumerate(self.blocks):
            block.to(device)
            svdq = {n: m for n, m in block.named_modules() if isinstance(m, SVDQW4A4Linear)}
            if i >= self.num_blocks_on_gpu:
                for n, m in svdq.items():  # keep the nibble image for the link before the repack replaces it
                    if "qweight" not in m._amd_names:
                        t = m.qweight.data.to("cpu")
                        self._host_nibbles[i][n] = t.pin_memory() if self.use_pin_memory else t
            for m in svdq.values():
                m.repack_()
            if i == 0:
                # two buffer slots shaped like a converted block (FP6-image qweights, marked as kernel layout)
                self.buffer_blocks = [copy.deepcopy(block), copy.deepcopy(block)]
                for b in self.buffer_blocks:
                    for m in b.modules():
                        if isinstance(m, SVDQW4A4Linear):
                            m._set_amd_names(m._amd_names)
                self._staging = [{n: torch.empty(m.out_features, m.in_features // 2, dtype=torch.int8, device=device)
                                  for n, m in svdq.items()} for _ in range(2)]
            if i >= self.num_blocks_on_gpu:
                for n, m in svdq.items():
                    if n in self._host_nibbles[i]:
                        m.qweight.data = torch.empty(0, dtype=torch.int8)  # the FP6 image never lives on the host
                block.to("cpu")
                if self.use_pin_memory:
                    for p in block.parameters(recurse=True):
                        p.data = p.data.pin_memory()
                    for b in block.buffers(recurse=True):
                        b.data = b.data.pin_memory()

    # ------------------------------------------------------------------ the ping-pong
    def load_block(self, block_idx: int, non_blocking: bool = True):
        """Host -> buffer slot ``block_idx % 2`` on the CURRENT stream (``step`` calls it under ``memory_stream``)."""
        if block_idx < self.num_blocks_on_gpu or block_idx >= len(self.blocks):
            return
        src, dst = self.blocks[block_idx], self.buffer_blocks[block_idx % 2]
        nib, stage = self._host_nibbles[block_idx], self._staging[block_idx % 2]
        lib = _lib.load()
        st = torch.cuda.current_stream().cuda_stream
        with torch.no_grad():
            dst_mods = dict(dst.named_modules())
            for name, ms in src.named_modules():
                md = dst_mods[name]
                for (pn, ps), (_, pd) in zip(ms.named_parameters(recurse=False), md.named_parameters(recurse=False)):
                    if pn == "qweight" and name in nib:
                        stage[name].copy_(nib[name], non_blocking=non_blocking)        # 4-bit nibbles over the link
                        _lib.check(lib.svdq_repack_qweight(stage[name].data_ptr(), pd.data_ptr(), md.out_features, md.in_features, st),
                                   "svdq_repack_qweight")                              # -> FP6 image, in HBM
                    else:
                        pd.copy_(ps, non_blocking=non_blocking)
                for (_, bs), (_, bd) in zip(ms.named_buffers(recurse=False), md.named_buffers(recurse=False)):
                    bd.copy_(bs, non_blocking=non_blocking)

    def step(self, compute_stream: torch.cuda.Stream | None = None):
        """Advance to the next block: its predecessor's compute is recorded, the successor's copy is queued behind the
        event that frees its slot, and the compute stream waits for the copy of the block it is about to run."""
        if compute_stream is None:
            compute_stream = torch.cuda.current_stream()
        next_compute_done = torch.cuda.Event()
        next_compute_done.record(compute_stream)
        with torch.cuda.stream(self.memory_stream):
            self.memory_stream.wait_event(self.compute_done)
            self.load_block(self.current_block_idx + 1)
            next_memory_done = torch.cuda.Event()
            next_memory_done.record(self.memory_stream)
        self.memory_done = next_memory_done
        self.compute_done = next_compute_done
        self.current_block_idx += 1
        if self.current_block_idx < len(self.blocks):
            compute_stream.wait_event(self.memory_done)
        else:
            compute_stream.wait_event(self.compute_done)
            self.current_block_idx = 0
            self.forward_counter += 1
            if self.empty_cache_freq > 0 and self.forward_counter % self.empty_cache_freq == 0:
                torch.cuda.empty_cache()

    def get_block(self, block_idx: int | None = None) -> nn.Module:
        if block_idx is None:
            block_idx = self.current_block_idx
        if block_idx < self.num_blocks_on_gpu:
            return self.blocks[block_idx]
        return self.buffer_blocks[block_idx % 2]

    def initialize(self, stream: torch.cuda.Stream | None = None):
        if stream is None:
            stream = torch.cuda.current_stream()
        self.compute_done.record(stream)
        self.memory_done.record(stream)

    def host_bytes_per_block(self, block_idx: int | None = None) -> int:
        """Bytes that cross the link for one offloaded block (nibble qweights + the other tensors)."""
        i = self.num_blocks_on_gpu if block_idx is None else block_idx
        if i >= len(self.blocks):
            return 0
        n = sum(t.numel() * t.element_size() for t in self._host_nibbles[i].values())
        n += sum(p.numel() * p.element_size() for p in self.blocks[i].parameters())
        return n + sum(b.numel() * b.element_size() for b in self.blocks[i].buffers())
