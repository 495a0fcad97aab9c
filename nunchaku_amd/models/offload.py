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
    staging buffer of the runtime and never overlap.  A block is ONE flat pinned buffer and ONE copy (the reference issues
    one copy per tensor, ~120 per block: at ~10 us of submission each that is a quarter of the copy time here).
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
        self._host_flat: list = []      # per offloaded block: ONE pinned byte buffer with all its tensors
        self._dev_flat: list = [{}, {}]  # per buffer slot: the matching flat device buffer(s)
        self.device = None
        self.set_device(device)
        self.current_block_idx = 0
        self.forward_counter = 0
        self.empty_cache_freq = empty_cache_freq

    # ------------------------------------------------------------------ placement
    @staticmethod
    def _tensors(block: nn.Module):
        """(module name, tensor name, tensor) of every parameter and buffer, in a fixed order"""
        out = []
        for mn, m in block.named_modules():
            for pn, p in m.named_parameters(recurse=False):
                out.append((mn, pn, p))
            for bn, b in m.named_buffers(recurse=False):
                out.append((mn, bn, b))
        return out

    def set_device(self, device: torch.device | str, force: bool = False):
        """Buffers and resident blocks to ``device``, the other blocks to (pinned) host memory.  SVDQuant layers are
        converted to the kernel layout HERE, once; every offloaded block becomes ONE flat pinned byte buffer (all its tensors
        back to back, 256-byte aligned; ``qweight`` in nibble form where the layer had not been repacked before), each buffer
        slot ONE flat device buffer of the same layout whose pieces the slot's parameters view -- a block load is a single
        H2D copy (one SDMA submission instead of ~120) plus the nibble -> FP6 expansions."""
        if isinstance(device, str):
            device = torch.device(device)
        assert device.type == "cuda"
        if self.device == device and not force:
            return
        self.device = device
        self.memory_stream = torch.cuda.Stream(device=device)
        for module in self.on_gpu_modules:
            module.to(device)
        self._host_flat = [None] * len(self.blocks)
        self._layout = None      # [(module name, tensor name, byte offset, byte size, is_nibble_qweight)], same for every block
        self._nibble_mode = [False] * len(self.blocks)
        for i, block in enumerate(self.blocks):
            block.to(device)
            svdq = {n: m for n, m in block.named_modules() if isinstance(m, SVDQW4A4Linear)}
            nibbles = {}
            if i >= self.num_blocks_on_gpu:
                for n, m in svdq.items():  # keep the nibble image for the link before the repack replaces it
                    if "qweight" not in m._amd_names:
                        nibbles[n] = m.qweight.data.clone()
            for m in svdq.values():
                m.repack_()
            if i == 0:
                # two buffer slots shaped like a converted block (FP6-image qweights, marked as kernel layout)
                self.buffer_blocks = [copy.deepcopy(block), copy.deepcopy(block)]
                for b in self.buffer_blocks:
                    for m in b.modules():
                        if isinstance(m, SVDQW4A4Linear):
                            m._set_amd_names(m._amd_names)
            if i < self.num_blocks_on_gpu:
                continue
            # flat host image of this block
            items, off = [], 0
            for mn, tn, t in self._tensors(block):
                nib = tn == "qweight" and mn in nibbles
                src = nibbles[mn] if nib else t.data
                size = src.numel() * src.element_size()
                items.append((mn, tn, off, size, nib, src))
                off = (off + size + 255) // 256 * 256
            if self._layout is None or len(self._layout) != len(items):
                self._layout = None
            flat = torch.empty(off, dtype=torch.uint8, device="cpu")
            if self.use_pin_memory:
                flat = flat.pin_memory()
            for mn, tn, o, size, nib, src in items:
                flat[o:o + size].copy_(src.contiguous().view(-1).view(torch.uint8))
            self._host_flat[i] = flat
            self._nibble_mode[i] = bool(nibbles)
            lay = [(mn, tn, o, size, nib) for mn, tn, o, size, nib, _ in items]
            self._block_layouts = getattr(self, "_block_layouts", {})
            self._block_layouts[i] = lay
            # the block object keeps its structure but not its data (everything lives in the flat image now)
            for _, _, t in self._tensors(block):
                t.data = torch.empty(0, dtype=t.dtype)
        # device side: one flat buffer per slot and layout kind (nibble / image), parameters of the slot view into it
        self._dev_flat = [{}, {}]

    def _bind_slot(self, slot: int, block_idx: int):
        """Point the slot's parameters at the pieces of its flat device buffer for the layout of block ``block_idx``."""
        lay = self._block_layouts[block_idx]
        key = self._nibble_mode[block_idx]
        total = self._host_flat[block_idx].numel()
        bound = self._dev_flat[slot].get(key)
        if bound is not None and bound[0].numel() == total:
            return bound
        flat = torch.empty(total, dtype=torch.uint8, device=self.device)
        mods = dict(self.buffer_blocks[slot].named_modules())
        expand = []  # (nibble view, FP6 image tensor, N, K)
        for mn, tn, o, size, nib in lay:
            m = mods[mn]
            t = getattr(m, tn)
            if nib:
                img = t.data if t.data.numel() == m.out_features * m.in_features * 3 // 4 and t.data.is_cuda else \
                    torch.empty(m.out_features, m.in_features * 3 // 4, dtype=torch.int8, device=self.device)
                t.data = img
                expand.append((flat[o:o + size], img, m.out_features, m.in_features))
            else:
                shape = tuple(t.shape) if t.data.numel() * t.element_size() == size else None
                if shape is None:
                    raise RuntimeError(f"offload: {mn}.{tn} of block {block_idx} does not match the buffer slot's shape")
                t.data = flat[o:o + size].view(t.dtype).view(shape)
        self._dev_flat[slot][key] = (flat, expand)
        return self._dev_flat[slot][key]

    # ------------------------------------------------------------------ the ping-pong
    def load_block(self, block_idx: int, non_blocking: bool = True):
        """Host -> buffer slot ``block_idx % 2`` on the CURRENT stream (``step`` calls it under ``memory_stream``): one copy
        of the block's flat image, then the nibble -> FP6 expansion of its code tensors."""
        if block_idx < self.num_blocks_on_gpu or block_idx >= len(self.blocks):
            return
        slot = block_idx % 2
        flat, expand = self._bind_slot(slot, block_idx)
        flat.copy_(self._host_flat[block_idx], non_blocking=non_blocking)
        if expand:
            lib = _lib.load()
            st = torch.cuda.current_stream().cuda_stream
            for nib, img, N, K in expand:
                _lib.check(lib.svdq_repack_qweight(nib.data_ptr(), img.data_ptr(), N, K, st), "svdq_repack_qweight")

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
        """Bytes that cross the link for one offloaded block (its flat image: nibble qweights + every other tensor)."""
        i = self.num_blocks_on_gpu if block_idx is None else block_idx
        return 0 if i >= len(self.blocks) or self._host_flat[i] is None else self._host_flat[i].numel()

    def nibble_bytes(self, block_idx: int) -> int:
        """Bytes of block ``block_idx``'s flat image that are 4-bit code tensors in checkpoint (nibble) form."""
        return sum(size for _, _, _, size, nib in self._block_layouts.get(block_idx, []) if nib)
