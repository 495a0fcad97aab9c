"""Drop-in for the reference's pybind11 module ``nunchaku._C`` (hot-path subset).

``ops.quantize_w4a4_act_fuse_lora`` and ``ops.gemm_w4a4`` keep the reference's positional
signatures (nunchaku/csrc/ops.h:10-38, 83-90): every tensor is optional, outputs are allocated by
the caller and written in place, the call is asynchronous on the current torch stream and
returns ``None``.  Underneath they marshal raw device pointers into the C ABI
(include/svdq_amd.h) -- torch is only used for ``data_ptr()`` and the stream handle.

Shape errors raise ``ValueError`` (the reference aborts the process on a failed ``assert``,
SURVEY.md section 5), unsupported reference features raise ``NotImplementedError``.
"""

from __future__ import annotations

import collections
import ctypes as C
import weakref

import torch

from . import _lib
from . import mode as _mode

_DT = {torch.bfloat16: _lib.SVDQ_BF16, torch.float16: _lib.SVDQ_FP16}


def _ptr(t: torch.Tensor | None) -> int | None:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("nunchaku_amd ops need GPU tensors (there is no CPU path)")
    if not t.is_contiguous():
        raise ValueError("nunchaku_amd ops need contiguous tensors")
    return t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class _Workspace:
    """Scratch of one (device, stream, kind): the device buffer + a pinned, device-visible int32 status word the kernels
    raise when a waiting workgroup gives up (svdq_gemm_args.status): polled WITHOUT synchronising before every launch."""

    __slots__ = ("buf", "status", "captured")

    def __init__(self, buf, status):
        self.buf, self.status = buf, status
        self.captured = False  # its pointer is baked into a captured HIP graph: never evicted (a replay would use freed memory)

    def check(self, what: str):
        if self.status is not None and int(self.status[0]) != 0:  # a plain host read of pinned memory
            self.status.zero_()
            self.buf[:8192].zero_()  # counters of the abandoned launch (stream-ordered behind it)
            raise RuntimeError(
                f"{what}: an earlier launch on this stream gave up waiting for partial results of its persistent schedule "
                "(the workspace was shared across streams, or the grid was not co-resident: a CU-masked stream / a long-running "
                "co-tenant kernel).  Its results are invalid.  Set ops.gemm_use_workspace / ops.attention_use_workspace = False "
                "for such streams.")


_status_pool = None  # pinned int32 words handed to the workspaces (allocated once, outside any stream capture)
_status_used = 0
_status_free: list = []  # words of evicted workspaces: handed out again only behind a device synchronisation (below)


def _status_word():
    """One word of pinned, device-visible host memory, or None (pool exhausted, or first use inside a stream capture, where
    pinning is not permitted: such a workspace runs without the host-visible flag; the device-side error word stays).

    A word that belonged to an evicted workspace may still be the target of a late give-up write of a kernel queued on that workspace's
    stream (the buffer itself is only freed once that work has finished).  Fresh words are preferred, and an evicted word is reused only
    when the pool of 256 is exhausted and after a device synchronisation (nothing can be in flight behind it): the rare path of a process
    that has gone through more than 256 (device, stream, kind) workspaces."""
    global _status_pool, _status_used
    if _status_pool is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        _status_pool = torch.zeros(256, dtype=torch.int32).pin_memory()
    if _status_used < _status_pool.numel():
        w = _status_pool[_status_used:_status_used + 1]
        _status_used += 1
        return w
    if _status_free and not torch.cuda.is_current_stream_capturing():
        torch.cuda.synchronize()  # every kernel that could still raise an evicted word has retired
        w = _status_free.pop()
        w.zero_()
        return w
    return None


_WORKSPACE_LIMIT = 8  # per kind: LRU bound (88 MB GEMM / 66 MB attention each; + the 16-bit image of a split low-rank down projection once a rank > 32 launch
#                       has asked for it: 113-157 MB GEMM / 28-39 MB attention at the FLUX / Qwen-Image shapes); pools of temporary streams recycle entries
_WORKSPACE_BYTES_LIMIT = 512 << 20  # per kind and device: the cache is bounded by BYTES as well (round 6: eight entries of a split-launch size were ~1.3 GB per kind) --
#                                     the entry in use always stays, older ones go until the rest fits
_workspaces: "collections.OrderedDict[tuple[int, int, str], _Workspace]" = collections.OrderedDict()


def _evict(kind: str, keep) -> None:
    """Drop least recently used workspaces of `kind` on `keep`'s device beyond the entry limit or the byte limit -- never `keep` itself, never one a captured
    graph points at.  The tensor is freed by the caching allocator once the work queued on it has finished."""
    same = [k for k in _workspaces if k[2] == kind and k[0] == keep[0] and not _workspaces[k].captured]
    total = sum(_workspaces[k].buf.numel() for k in same)
    for k in list(same):
        if len(same) <= _WORKSPACE_LIMIT and total <= _WORKSPACE_BYTES_LIMIT:
            break
        if k == keep:
            continue
        old = _workspaces.pop(k)
        same.remove(k)
        total -= old.buf.numel()
        if old.status is not None:
            _status_free.append(old.status)


def _workspace(device: torch.device, kind: str = "gemm", min_bytes: int = 0) -> _Workspace:
    """Scratch of the GEMM's stream-K tail (arrival counters + fp32 partial tiles) or, ``kind="attention"``, of the
    attention kernel's persistent schedule (counters + fp32 partial (O, m, l)); one per (device, STREAM, kind).

    The C ABI allows a workspace to be reused only by launches that are ordered on one stream (include/svdq_amd.h,
    ``svdq_gemm_args.workspace``): two GEMMs in flight on different streams of a device would share counters and
    partial-tile slabs.  So the buffer is keyed by the current stream's handle: side streams (the offload manager's
    schedule, a capturing stream, worker threads with their own streams) each get their own buffer on first use, allocated
    from the torch caching allocator, zero-filled once; the kernels leave the counters at zero.  The cache is bounded
    (least recently used entry dropped beyond ``_WORKSPACE_LIMIT`` per kind); a launch that gave up (status word) has its
    counters cleared by ``_Workspace.check`` before the error is raised, so a later tenant of the stream handle never
    inherits a stale count.

    ``min_bytes`` (GEMM, ABI 20: ``svdq_gemm_workspace_bytes_for``): a launch that wants more than the base size -- the 16-bit output image of a
    GELU_QUANT launch with a split low-rank down projection -- gets a larger buffer in the stream's slot; the smaller one it replaces is freed by the
    caching allocator once the work queued on it has finished, its status word moves over.  A workspace a captured graph points at is never replaced,
    nor is one grown during a capture (the launch then takes the path the existing size allows)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream, kind)
    ws = _workspaces.get(key)
    grown = False
    if ws is not None and ws.buf.numel() < min_bytes and not ws.captured and not torch.cuda.is_current_stream_capturing():
        with torch.cuda.device(idx):
            buf = torch.zeros(int(min_bytes), dtype=torch.uint8, device=device)
        ws = _Workspace(buf, ws.status)
        _workspaces[key] = ws
        grown = True
    if ws is None:
        lib = _lib.load()
        size = lib.svdq_attention_workspace_bytes() if kind == "attention" else lib.svdq_gemm_workspace_bytes()
        with torch.cuda.device(idx):
            buf = torch.zeros(max(int(size), int(min_bytes)), dtype=torch.uint8, device=device)
        ws = _Workspace(buf, _status_word())
        _workspaces[key] = ws
        _evict(kind, key)
    else:
        _workspaces.move_to_end(key)
        if grown:
            _evict(kind, key)
    if torch.cuda.is_current_stream_capturing():
        ws.captured = True
    return ws


def release_workspaces() -> None:
    """Drop every cached workspace (e.g. after a pool of temporary streams has been destroyed).  Workspaces whose pointers are baked into a
    captured HIP graph (``captured``) are KEPT: a replay would otherwise read freed memory -- destroy the graphs first and pass nothing
    else; their entries go when the process does.  The same rule holds for the models' rotary-table caches: tables computed while a stream
    is capturing live in the graph's pool, never in the cache (models/flux.py, models/qwenimage.py)."""
    for k in [k for k, ws in _workspaces.items() if not ws.captured]:
        ws = _workspaces.pop(k)
        if ws.status is not None:
            _status_free.append(ws.status)


# ----------------------------------------------------------------------------------------------------------------------
# Reference-shaped buffers and reference-layout parameters (drop-in for the reference's UNCHANGED Python callers)
#
# The reference's ops/quantize.py, ops/fused.py and models/linear.py allocate the opaque activation-code buffers as
# [M_pad, K/2] bytes and hand over parameters in the checkpoint (NVIDIA fragment) layout.  This library's images are
# [M_pad, 3K/4] (FP6) and its parameters are re-laid out once.  So that those callers run UNCHANGED against this module,
# side data is kept PER STORAGE (the C++ allocation behind a tensor), not per tensor object:
#   * a code buffer of the reference size owns an FP6 image covering its whole storage; any view of the buffer -- .view(),
#     a row slice, the tensor a HIP-graph replay sees again -- resolves to the matching rows of that image.  Callers only
#     ever pass the buffer on to gemm_w4a4; the bytes of the small buffer stay unused;
#   * a weight-side tensor that does not carry the `_svdq_amd` mark (set by SVDQW4A4Linear on its Parameters) is taken to be
#     in the checkpoint layout and converted on first use; the converted copy is cached per storage and reused until the
#     tensor is modified through torch (``_version``) or ``invalidate()`` is called (writes through ``.data`` do not bump
#     the version: LoRA / offload code that updates weights that way must call it);
#   * packed Q / K / V buffers of the "nunchaku-fp16" attention surface remember which token rows are real.
# An entry dies with its storage (weakref.finalize on the storage object: the last view going away frees the side data too,
# stream-ordered through the caching allocator like the buffer itself).
# The fast path of this package (nunchaku_amd.models / nunchaku_amd.ops) allocates FP6-sized buffers and repacks its
# parameters in place: none of this is touched there.
# ----------------------------------------------------------------------------------------------------------------------
class _SideTable:
    """side data per tensor STORAGE (weakly keyed)"""

    def __init__(self):
        self._d: dict[int, object] = {}

    def get(self, t: torch.Tensor):
        return self._d.get(t.untyped_storage()._cdata)

    def put(self, t: torch.Tensor, value):
        st = t.untyped_storage()
        key = st._cdata
        if key not in self._d:
            weakref.finalize(st, self._d.pop, key, None)
        self._d[key] = value

    def pop(self, t: torch.Tensor):
        return self._d.pop(t.untyped_storage()._cdata, None)

    def __len__(self):
        return len(self._d)


_fp6_images = _SideTable()    # storage of a reference-sized code buffer -> {K: image over the whole storage}
_converted = _SideTable()     # storage of a checkpoint-layout parameter -> {(kind, offset, shape): (version, converted)}
_packed_rows = _SideTable()   # storage of a packed K buffer -> {row offset: (valid rows, padded rows)}


def _fp6_image(buf: torch.Tensor, rows: int, K: int, create: bool, what: str) -> torch.Tensor:
    """The FP6 operand image behind the opaque code buffer ``buf`` ([rows, 3K/4]: itself; a view of a [*, K/2] buffer: the
    matching rows of the image that storage owns)."""
    if buf.shape[-1] * 4 == K * 3:
        return buf
    if buf.shape[-1] * 2 != K:
        raise ValueError(f"{what}: the code buffer must be [M_pad, 3K/4] bytes (this library's FP6 image) or the reference's [M_pad, K/2]")
    if not buf.is_contiguous():
        raise ValueError(f"{what}: the code buffer must be contiguous")
    row_bytes = K // 2
    off_bytes = buf.storage_offset() * buf.element_size()
    if off_bytes % row_bytes:
        raise ValueError(f"{what}: a view of a code buffer must start at a row boundary")
    total_rows = buf.untyped_storage().nbytes() // row_bytes
    row0 = off_bytes // row_bytes
    per_k = _fp6_images.get(buf)
    img = per_k.get(K) if per_k else None
    if img is None or img.shape[0] != total_rows or img.device != buf.device:
        if not create:
            raise ValueError(f"{what}: this reference-sized code buffer was not produced by quantize_w4a4_act_fuse_lora / gemm_w4a4 of "
                             "this library (its FP6 image belongs to the buffer's STORAGE: pass the buffer or a view of it on, not a copy)")
        img = torch.empty(total_rows, K * 3 // 4, dtype=torch.uint8, device=buf.device)
        per_k = dict(per_k or {})
        per_k[K] = img
        _fp6_images.put(buf, per_k)
    return img[row0:row0 + rows]


def mark_amd(t: torch.Tensor | None) -> torch.Tensor | None:
    """Declare ``t`` to hold this library's kernel layout (no conversion on use)."""
    if t is not None:
        t._svdq_amd = True
    return t


def invalidate(t: torch.Tensor | None) -> None:
    """Forget the cached kernel-layout conversion of the checkpoint-layout tensor ``t`` (call after writing to it through
    ``.data`` -- such writes do not bump ``t._version``, so the cache cannot see them)."""
    if t is not None:
        _converted.pop(t)


def _cached_conversion(t: torch.Tensor, kind: str, convert):
    key = (kind, t.storage_offset(), tuple(t.shape))
    per = _converted.get(t)
    hit = per.get(key) if per else None
    if hit is not None and hit[0] == t._version:
        return hit[1]
    conv = convert(t.detach())
    per = dict(per or {})
    per[key] = (t._version, conv)
    _converted.put(t, per)
    return conv


def _param(t: torch.Tensor | None, kind: str):
    """A weight-side tensor in the kernel layout: ``t`` itself when marked, else its cached conversion from the checkpoint
    layout (kind: "vec" bias / smooth, "wscales", "up" / "down" low-rank factors)."""
    if t is None or getattr(t, "_svdq_amd", False):
        return t
    from . import layout

    if kind == "vec":
        return _cached_conversion(t, kind, layout.repack_vec)
    if kind == "wscales":
        return _cached_conversion(t, kind, layout.repack_wscales)
    return _cached_conversion(t, kind, lambda src: layout.repack_lowrank(src, down=(kind == "down")))


def _packed_fragments(t: torch.Tensor | None, down: bool, N: int, R: int):
    """ABI 21: the MFMA-fragment image of a low-rank factor in the kernel layout (``down``: rank-major ``[R][N]`` -> svdq_pack_lora_down; else ``[N][R]`` ->
    svdq_pack_lora_up), packed ONCE per storage and version (a ``set_lora`` writes a new tensor or bumps the version: re-packed) -- the rank 48 .. 160 kernels
    otherwise re-pack the weight on every launch (190 pack launches per rank-128 FLUX step).  None where the library has no such image."""
    if t is None or not _Ops.cache_packed_lowrank or R % 16 or t.dtype not in _DT:
        return None
    if down and not (48 <= R <= 160 and N % 256 == 0):
        return None
    if not down and not (48 <= R <= 160 and N % 32 == 0):
        return None
    lib = _lib.load()

    def pack(src):
        nbytes = int(lib.svdq_pack_lora_down_bytes(N, R) if down else lib.svdq_pack_lora_up_bytes(N, R))
        out = torch.empty(nbytes, dtype=torch.uint8, device=src.device)
        fn = lib.svdq_pack_lora_down if down else lib.svdq_pack_lora_up
        _lib.check(fn(src.data_ptr(), out.data_ptr(), N, R, _DT[src.dtype], _stream()), "pack_lora_down" if down else "pack_lora_up")
        return out

    return _cached_conversion(t, f"frag_{'down' if down else 'up'}", pack)


def _weight(wgt: torch.Tensor, K: int) -> torch.Tensor:
    """qweight: [N, 3K/4] FP6 image (kernel layout) or the checkpoint's [N, K/2] int8 (converted once, cached per storage)."""
    if wgt.shape[-1] * 4 == K * 3:
        return wgt
    from . import layout

    return _cached_conversion(wgt, "qweight", lambda src: layout.repack_qweight(src.view(torch.int8)))


class _Ops:
    # Workgroup geometry of gemm_w4a4 (svdq_gemm_args.geometry): 0 = the library's choice, 1 = 256 x 128 tiles / one
    # workgroup per CU, 2 = 128 x 128 tiles / two workgroups per CU drawing tiles from per-XCD queues, 3 = 128 x 128 with
    # fixed tile lists, 4 / 5 = 2 / 3 with a phase offset between a CU's two workgroups.  A plain attribute, not an
    # environment variable: nothing is read from os.environ on the launch path.
    gemm_geometry = 0
    # False: launch without the stream-K workspace (whole-tile schedule only); tests compare the two schedules
    gemm_use_workspace = True

    @staticmethod
    def gemm_workspace_status() -> None:
        """Synchronise the current stream and raise ``RuntimeError`` if a stream-K GEMM on it timed out waiting for
        partial tiles (a workspace shared across streams -- see ``_workspace``).  Test / debugging aid."""
        key = (torch.cuda.current_device(), _stream(), "gemm")
        ws = _workspaces.get(key)
        if ws is not None:
            _lib.check(_lib.load().svdq_gemm_workspace_status(ws.buf.data_ptr(), _stream()), "gemm_workspace_status")
            if ws.status is not None:
                ws.status.zero_()

    cache_packed_lowrank = True   # ABI 21: keep the fragment images of rank 48 .. 160 low-rank factors per parameter (False: the launches pack them, as ABI 20 did)

    PLAN_VARIANTS = ("plain", "carry", "all_rank", "hybrid_carry", "solo_carry", "split_down", "wave_tile_128")

    @staticmethod
    def gemm_last_plan() -> dict:
        """What this thread's last ``gemm_w4a4`` launched (``svdq_gemm_last_plan``): tile rows, kernel variant, grid, stream-K groups, row-run length,
        packed low-rank operands, dynamic queue.  Tests assert with it that no rank / shape / format fell back to a slower kernel than documented."""
        out = (C.c_int32 * 8)()
        _lib.check(_lib.load().svdq_gemm_last_plan(out), "gemm_last_plan")
        return {"tile_rows": out[0], "variant": _Ops.PLAN_VARIANTS[out[1]], "grid": out[2], "streamk_groups": out[3], "rowrun": out[4],
                "lora_act_packed": bool(out[5]), "lora_up_packed": bool(out[6]), "dynamic_queue": bool(out[7])}

    # False: plain grid (one workgroup per task) instead of the persistent schedule; tests compare the two
    attention_use_workspace = True
    # Workgroup geometry of the attention kernel (svdq_attention_args.geometry): 0 = the library's choice, 1 = 8 waves x 32 query
    # rows, 2 = 4 waves x 64 query rows (L % 256 == 0 only; other lengths always run geometry 1)
    attention_geometry = 0
    @staticmethod
    def attention_last_plan() -> dict:
        """What this thread's last ``attention`` launched (``svdq_attention_last_plan``)."""
        out = (C.c_int32 * 4)()
        _lib.check(_lib.load().svdq_attention_last_plan(out), "attention_last_plan")
        return {"geometry": out[0], "persistent_groups": out[1], "masked_geometry2": bool(out[2]), "split_lowrank": bool(out[3])}

    # False: the fused quantiser's low-rank down projection stays inside the attention epilogue at every rank (tests / A/B of the split path, rank 48 .. 160)
    attention_split_lowrank = True

    @staticmethod
    def attention_workspace_status() -> None:
        """Raises if an attention launch on the current stream gave up waiting for partial results.  Test / debugging aid."""
        ws = _workspaces.get((torch.cuda.current_device(), _stream(), "attention"))
        if ws is not None:
            _lib.check(_lib.load().svdq_attention_workspace_status(ws.buf.data_ptr(), _stream()), "attention_workspace_status")
            if ws.status is not None:
                ws.status.zero_()

    @staticmethod
    def quantize_w4a4_act_fuse_lora(input, output, oscales, lora_down, lora_act_out, smooth, fuse_glu=False, fp4=False,
                                    ln_stats=None, mod_scale=None, mod_shift=None, lora_act_zeroed=False, second=None):
        """reference: csrc/ops.h:83-112 -> kernels::quantize_w4a4_act_fuse_lora (zgemm.h:39-46).

        ``ln_stats`` / ``mod_scale`` / ``mod_shift`` (extension, all or none): quantise
        ``layer_norm(input) * scale + shift`` (16-bit torch-op rounding, scale with the +1 included) instead of ``input``;
        ``ln_stats`` is the ``[M, 2]`` float32 (mean, rstd) tensor ``ops.residual_gate_stats`` returns.
        ``second`` (extension, grouped launch): dict ``input, smooth, lora_down[, ln_stats, mod_scale, mod_shift]`` of a
        second stream whose rows follow the first stream's (a multiple of 256) in the same output buffers."""
        lib = _lib.load()
        if input is None or output is None or oscales is None:
            raise ValueError("quantize_w4a4_act_fuse_lora: input, output and oscales are required")
        if input.dtype not in _DT:
            raise ValueError(f"quantize_w4a4_act_fuse_lora: unsupported dtype {input.dtype}")
        if input.dim() != 2:
            input = input.reshape(-1, input.shape[-1])
        if input.stride(-1) != 1:
            input = input.contiguous()
        M, K = input.shape
        if fuse_glu:  # rows of (value, gate) pairs: the quantised width is half the input's (launch_impl.cuh:463)
            if K % 2 or ln_stats is not None or second is not None:
                raise ValueError("quantize_w4a4_act_fuse_lora: fuse_glu needs an even input width and does not combine with ln_stats / second")
            K //= 2
        M_pad = output.numel() // output.shape[-1]
        R = 0 if lora_down is None else lora_down.shape[-1]
        image = _fp6_image(output, M_pad, K, True, "quantize_w4a4_act_fuse_lora")
        smooth, lora_down = _param(smooth, "vec"), _param(lora_down, "down")
        a = _lib.QuantizeArgs()
        a.x = input.data_ptr()
        a.smooth = _ptr(smooth)
        a.lora_down = _ptr(lora_down)
        a.act = _ptr(image)
        a.ascales = _ptr(oscales)
        a.lora_act = _ptr(lora_act_out)
        a.M, a.M_pad, a.K, a.R = M, M_pad, K, R
        a.ldx = input.stride(0)
        a.dtype = _DT[input.dtype]
        a.fuse_glu, a.fp4 = int(bool(fuse_glu)), int(bool(fp4))
        a.lora_act_zeroed = int(bool(lora_act_zeroed))  # extension: lora_act_out already cleared on this stream
        if ln_stats is not None or mod_scale is not None or mod_shift is not None:
            if ln_stats is None or mod_scale is None or mod_shift is None:
                raise ValueError("quantize_w4a4_act_fuse_lora: ln_stats, mod_scale and mod_shift go together")
            if ln_stats.dtype != torch.float32 or ln_stats.numel() != 2 * M:
                raise ValueError("quantize_w4a4_act_fuse_lora: ln_stats must be float32 [M, 2]")
            if mod_scale.numel() != K or mod_shift.numel() != K or mod_scale.dtype != input.dtype or mod_shift.dtype != input.dtype:
                raise ValueError("quantize_w4a4_act_fuse_lora: mod_scale / mod_shift must be [K] in the input dtype")
            a.ln_stats, a.mod_scale, a.mod_shift = _ptr(ln_stats), _ptr(mod_scale), _ptr(mod_shift)
        keep2 = None
        if second is not None:
            x2 = second["input"].reshape(-1, K)
            if x2.stride(-1) != 1:
                x2 = x2.contiguous()
            a.x2, a.M2, a.ldx2, a.split_rows = x2.data_ptr(), x2.shape[0], x2.stride(0), M
            sm2, ld2 = _param(second.get("smooth"), "vec"), _param(second.get("lora_down"), "down")
            a.smooth2, a.lora_down2 = _ptr(sm2), _ptr(ld2)
            a.ln_stats2, a.mod_scale2, a.mod_shift2 = _ptr(second.get("ln_stats")), _ptr(second.get("mod_scale")), _ptr(second.get("mod_shift"))
            keep2 = (x2, second, sm2, ld2)
        if oscales.numel() != (K // 64) * M_pad:
            raise ValueError("quantize_w4a4_act_fuse_lora: oscales must hold (K/64)*M_pad scales")
        if R and (lora_act_out.numel() != M_pad * R or lora_act_out.dtype not in (torch.float32, torch.int64)):
            raise ValueError("quantize_w4a4_act_fuse_lora: lora_act_out must hold M_pad*R float32 (or int64: the deterministic fixed-point format)")
        if R and lora_act_out.dtype == torch.int64:
            a.lora_act_format = _lib.LORA_ACT_Q32
        _lib.check(lib.svdq_quantize_w4a4_act_fuse_lora(C.byref(a), _stream()), "quantize_w4a4_act_fuse_lora")
        del keep2

    @staticmethod
    def gemm_w4a4(
        act, wgt, out, qout, ascales, wscales, oscales, poolout, lora_act_in, lora_up, lora_down, lora_act_out,
        norm_q, norm_k, rotary_emb, bias, smooth_factor, out_vk, out_linearattn, act_unsigned, lora_scales,
        fuse_silu, fp4, alpha, wcscales, out_q, out_k, out_v, attn_tokens, out_vt=None, lora_act_zeroed=False, second=None, split_rows=0,
        q_scale=0.0,
    ):
        """reference: csrc/ops.h:10-81 -> kernels::gemm_w4a4 (zgemm.h:8-36).  The epilogue is inferred
        from which optional tensors are present, exactly as gemm_w4a4_launch_impl.cuh:282-423 does.

        ``out_vt`` (extension, RMSNorm+RoPE epilogue only): a ``[N/3, tokens]`` view with unit column stride
        that receives V transposed instead of the V columns of ``out`` -- the operand ``ops.attention`` reads
        (role of the reference's packed out_q/out_k/out_v, epilogues.cuh:427-550).

        ``second`` / ``split_rows`` (extension, grouped launch): rows ``>= split_rows`` use the weight-side tensors
        of the dict ``second`` (keys ``wgt, wscales, bias, lora_up, smooth_factor, lora_down, norm_q, norm_k``;
        same shapes as the first set) -- the text and image streams of a joint block in ONE launch; all row-side
        tensors are the two streams' buffers back to back."""
        lib = _lib.load()
        if fp4:
            raise NotImplementedError("gemm_w4a4: fp4 (NVFP4) is Blackwell-only; use int4 checkpoints")
        if alpha is not None and float(alpha) != 1.0:
            raise ValueError("gemm_w4a4: alpha must be 1.0 for int4 (launch_impl.cuh:107)")
        if out_linearattn is not None or out_vk is not None:
            raise NotImplementedError("gemm_w4a4: the SANA LiteLA epilogue is out of scope")
        packed_qkv = None
        if out_q is not None or out_k is not None or out_v is not None:
            # the reference's "nunchaku-fp16" attention path (attention_processors/flux.py:114-237): Q / K / V go to three
            # opaque [B, H, T_pad, 128] buffers that only ops.attention_fp16 reads.  Adapter: the GEMM runs into a scratch
            # [M, N] tensor and the thirds are scattered head-major into the caller's (strided) views below.
            if out_q is None or out_k is None or out_v is None or out is not None:
                raise ValueError("gemm_w4a4: out_q, out_k and out_v go together (and without out)")
            for t in (out_q, out_k, out_v):
                if t.dim() != 4 or t.shape[0] != 1 or t.shape[-1] != 128 or t.stride(-1) != 1 or t.shape != out_q.shape:
                    raise NotImplementedError("gemm_w4a4: out_q / out_k / out_v must be [1, H, T_pad, 128] views (batch 1, head_dim 128)")
            packed_qkv = (out_q, out_k, out_v, int(attn_tokens))
            out = torch.empty(int(attn_tokens) if attn_tokens else out_q.shape[2], 3 * out_q.shape[1] * 128, dtype=out_q.dtype, device=out_q.device)
        if act is None or wgt is None or ascales is None or wscales is None:
            raise ValueError("gemm_w4a4: act, wgt, ascales and wscales are required")
        if ascales.dtype not in _DT:
            raise ValueError(f"gemm_w4a4: unsupported dtype {ascales.dtype}")

        a = _lib.GemmArgs()
        M_pad = act.numel() // act.shape[-1]
        K = ascales.numel() // M_pad * 64  # (the code buffer may be reference-sized: K comes from the scale image)
        N = wgt.shape[0]
        if wgt.shape[-1] * 4 != K * 3 and wgt.shape[-1] * 2 != K:
            raise ValueError("gemm_w4a4: wgt must be the [N, 3K/4]-byte FP6 operand image or the checkpoint's [N, K/2] with the K of act")
        act_img = _fp6_image(act, M_pad, K, False, "gemm_w4a4 (act)")
        wgt_img = _weight(wgt, K)
        wscales, bias = _param(wscales, "wscales"), _param(bias, "vec")
        a.act, a.wgt, a.ascales, a.wscales = _ptr(act_img), _ptr(wgt_img), _ptr(ascales), _ptr(wscales)
        a.bias = _ptr(bias)
        R = 0
        if lora_up is not None and lora_up.numel() > 0:
            R = lora_up.shape[-1]
            lora_up = _param(lora_up, "up")
            a.lora_up, a.lora_act_in = _ptr(lora_up), _ptr(lora_act_in)
            keep_frag = [_packed_fragments(lora_up, False, N, R) if second is None else None]
            a.lora_up_packed = _ptr(keep_frag[0])
        keep = None
        if lora_scales is not None and R:
            keep = (C.c_float * (R // 16))(*[float(s) for s in list(lora_scales)[: R // 16]])
            a.lora_scales = C.cast(keep, C.POINTER(C.c_float))
        a.M_pad, a.N, a.K, a.R = M_pad, N, K, R
        a.M = M_pad
        a.dtype = _DT[ascales.dtype]
        a.act_unsigned = int(bool(act_unsigned))
        if _Ops.gemm_use_workspace:
            ws = _workspace(act.device)
            ws.check("gemm_w4a4")  # host-visible status word of the earlier launches on this stream: no synchronisation
            a.workspace, a.workspace_bytes = ws.buf.data_ptr(), ws.buf.numel()
            a.status = None if ws.status is None else ws.status.data_ptr()
        a.geometry = _Ops.gemm_geometry
        fmt_in = lora_act_in.dtype if R else None

        if qout is not None and oscales is not None:
            a.fuse = _lib.FUSE_GELU_QUANT
            qout_img = _fp6_image(qout, M_pad, N, True, "gemm_w4a4 (qout)")
            smooth_factor = _param(smooth_factor, "vec")
            a.qout, a.oscales, a.next_smooth = _ptr(qout_img), _ptr(oscales), _ptr(smooth_factor)
            if lora_down is not None and lora_down.numel() > 0:
                a.R2 = lora_down.shape[-1]
                lora_down = _param(lora_down, "down")
                a.next_lora_down, a.lora_act_out = _ptr(lora_down), _ptr(lora_act_out)
                frag_down = _packed_fragments(lora_down, True, N, a.R2)
                a.next_lora_down_packed = _ptr(frag_down)
                if lora_act_out.dtype not in (torch.float32, torch.int64) or (fmt_in is not None and lora_act_out.dtype != fmt_in):
                    raise ValueError("gemm_w4a4: lora_act_in and lora_act_out must share one format (float32, or int64 fixed point)")
                fmt_in = lora_act_out.dtype
                if not lora_act_zeroed:  # extension: the caller cleared it already (residual_gate_stats(..., zero=))
                    lora_act_out.zero_()  # launch_impl.cuh:252
        elif rotary_emb is not None:
            if out is None or norm_q is None or norm_k is None:
                raise ValueError("gemm_w4a4: the RMSNorm+RoPE epilogue needs out, norm_q and norm_k")
            if rotary_emb.dtype != torch.float32 or rotary_emb.shape[-1] != 128 or rotary_emb.numel() != M_pad * 128:
                raise ValueError("gemm_w4a4: rotary_emb must be float32 [M_pad, 128] (pack_rotemb order)")
            a.fuse = _lib.FUSE_RMSNORM_ROPE
            a.norm_q, a.norm_k, a.rotary_emb = _ptr(norm_q), _ptr(norm_k), _ptr(rotary_emb)
            if out_vt is not None:
                if out_vt.dim() != 2 or out_vt.stride(1) != 1 or out_vt.shape[0] * 3 != N or out_vt.dtype != ascales.dtype:
                    raise ValueError("gemm_w4a4: out_vt must be a [N/3, tokens] view with unit column stride in the model dtype")
                if not out_vt.is_cuda:
                    raise RuntimeError("nunchaku_amd ops need GPU tensors (there is no CPU path)")
                a.out_vt, a.ldvt = out_vt.data_ptr(), out_vt.stride(0)
        elif out is not None:
            a.fuse = _lib.FUSE_SILU if fuse_silu else _lib.FUSE_NONE
        else:
            raise ValueError("gemm_w4a4: no output tensor given")
        if out is not None:
            a.out = _ptr(out)
            a.M = out.numel() // out.shape[-1]
            a.ldo = out.shape[-1]
            if out.shape[-1] != N:
                raise ValueError("gemm_w4a4: out.shape[-1] must equal N")
            if a.M > M_pad or M_pad - a.M >= 256:
                raise ValueError("gemm_w4a4: out rows must satisfy M <= M_pad < M + 256 (launch_impl.cuh:55)")
        keep2 = None
        if second is not None:
            if second.get("wgt") is None or tuple(second["wgt"].shape) != tuple(wgt.shape):
                raise ValueError("gemm_w4a4: second['wgt'] must have the shape of wgt")
            conv = {"wgt": _weight(second["wgt"], K), "wscales": _param(second.get("wscales"), "wscales"),
                    "bias": _param(second.get("bias"), "vec"), "lora_up": _param(second.get("lora_up"), "up"),
                    "smooth_factor": _param(second.get("smooth_factor"), "vec"), "lora_down": _param(second.get("lora_down"), "down"),
                    "norm_q": second.get("norm_q"), "norm_k": second.get("norm_k")}
            g = lambda k: _ptr(conv.get(k))
            a.wgt2, a.wscales2, a.bias2 = g("wgt"), g("wscales"), g("bias")
            a.lora_up2 = g("lora_up") if R else None
            a.next_smooth2, a.norm_q2, a.norm_k2 = g("smooth_factor"), g("norm_q"), g("norm_k")
            a.next_lora_down2 = g("lora_down") if a.R2 else None
            frag_down2 = _packed_fragments(conv.get("lora_down"), True, N, a.R2) if a.R2 else None
            a.next_lora_down_packed2 = _ptr(frag_down2)
            if a.next_lora_down_packed2 is None:
                a.next_lora_down_packed = None  # (both sets or none)
            a.split_rows = int(split_rows)
            keep2 = (second, conv)  # keeps the tensors alive until the launch has been issued
        if fmt_in is not None:
            if fmt_in not in (torch.float32, torch.int64):
                raise ValueError("gemm_w4a4: lora_act_in must be float32 (or int64: the deterministic fixed-point format)")
            a.lora_act_format = _lib.LORA_ACT_F32 if fmt_in != torch.int64 else _lib.LORA_ACT_Q32_RUNS if _mode.gemm_lora_act_format_runs() else _lib.LORA_ACT_Q32
        a.q_scale = float(q_scale)  # extension: the Q third times q_scale before its rounding (for attention(q_prescaled=True))
        if out_vt is not None and a.fuse != _lib.FUSE_RMSNORM_ROPE:
            raise ValueError("gemm_w4a4: out_vt needs the RMSNorm+RoPE epilogue (rotary_emb, norm_q, norm_k)")
        if out_vt is not None and out_vt.shape[1] < a.M:
            raise ValueError("gemm_w4a4: out_vt has fewer columns than out has rows")
        if _Ops.gemm_use_workspace and a.fuse == _lib.FUSE_GELU_QUANT and a.R2 > 32:
            # ABI 20: a next-layer rank beyond 32 can run its low-rank down projection split, given room for the launch's 16-bit output image
            need = int(lib.svdq_gemm_workspace_bytes_for(C.byref(a)))
            if need > a.workspace_bytes:
                ws = _workspace(act.device, min_bytes=need)
                a.workspace, a.workspace_bytes = ws.buf.data_ptr(), ws.buf.numel()
        _lib.check(lib.svdq_gemm_w4a4(C.byref(a), _stream()), "gemm_w4a4")
        del keep, keep2
        if packed_qkv is not None:
            oq, ok, ov, tokens = packed_qkv
            H, M, T_pad = oq.shape[1], out.shape[0], oq.shape[2]
            if tokens and tokens != M:
                raise ValueError("gemm_w4a4: attn_tokens must equal the number of rows of the projection")
            if M > T_pad:
                raise ValueError("gemm_w4a4: packed Q/K/V buffers are shorter than the projection's rows")
            for dst, third in zip((oq, ok, ov), out.view(M, 3, H, 128).unbind(1)):
                dst[0][:, :M].copy_(third.transpose(0, 1))  # [H, T, 128] head-major, the layout ops.attention_fp16 reads
            if M < T_pad:
                # padded token rows (the reference masks them inside its attention kernel, epilogues.cuh:427-550): V must be
                # finite there, K / Q may hold anything -- attention_fp16 masks the keys by the row counts remembered here
                ov[0][:, M:].zero_()
                ok[0][:, M:].zero_()
                oq[0][:, M:].zero_()
            base_tokens = ok.stride(1) // 128 if ok.stride(1) % 128 == 0 else T_pad
            row0 = (ok.storage_offset() // 128) % max(base_tokens, 1)
            # this write covers rows [row0, row0 + T_pad) of the buffer: segments an earlier use of the same storage left inside that range
            # (another text / image split) are gone with it
            rows = {r0: v for r0, v in (_packed_rows.get(ok) or {}).items() if r0 + v[1] <= row0 or r0 >= row0 + T_pad}
            rows[row0] = (M, T_pad)
            _packed_rows.put(ok, rows)

    @staticmethod
    def residual_gate_stats(res, a, b, gate, out, stats, eps=1e-6, zero=None, second=None, clamp_fp16=0):
        """Extension: ``out = res + gate * (a [+ b])`` (16-bit torch-op rounding of a block's gated residual)
        and/or the LayerNorm statistics ``stats[m] = (mean, rstd)`` of the result, in one pass.  2-D row-major
        views with a common row stride; ``a`` None = statistics of ``res`` itself; ``out`` may be ``res``."""
        lib = _lib.load()
        if res.dim() != 2 or res.stride(1) != 1 or res.dtype not in _DT:
            raise ValueError("residual_gate_stats: res must be a 2-D 16-bit view with unit column stride")
        M, Cc = res.shape
        args = _lib.ResidualArgs()
        for name, t in (("a", a), ("b", b), ("out", out)):
            if t is not None and (tuple(t.shape) != (M, Cc) or t.stride() != res.stride() or t.dtype != res.dtype):
                raise ValueError(f"residual_gate_stats: {name} must match res in shape, strides and dtype")
        if gate is not None and (gate.numel() != Cc or gate.dtype != res.dtype or not gate.is_contiguous()):
            raise ValueError("residual_gate_stats: gate must be a contiguous [C] tensor in the dtype of res")
        if stats is not None and (stats.dtype != torch.float32 or stats.numel() != 2 * M or not stats.is_contiguous()):
            raise ValueError("residual_gate_stats: stats must be contiguous float32 [M, 2]")
        for t in (res, a, b, gate, out, stats):
            if t is not None and not t.is_cuda:
                raise RuntimeError("nunchaku_amd ops need GPU tensors (there is no CPU path)")
        dp = lambda t: None if t is None else t.data_ptr()
        args.res, args.a, args.b, args.gate, args.out, args.stats = dp(res), dp(a), dp(b), dp(gate), dp(out), dp(stats)
        args.M, args.C, args.ld, args.dtype, args.eps = M, Cc, res.stride(0), _DT[res.dtype], float(eps)
        args.clamp_fp16 = int(clamp_fp16)  # bit 0 / bit 1: clip the first / second problem's result to +-65504 (fp16 only)
        if zero is not None:  # scratch cleared in the same pass (fp32 low-rank accumulators of the calls that follow)
            if not zero.is_cuda or not zero.is_contiguous() or (zero.numel() * zero.element_size()) % 16:
                raise ValueError("residual_gate_stats: zero must be a contiguous GPU tensor of a multiple of 16 bytes")
            args.zero_ptr, args.zero_bytes = zero.data_ptr(), zero.numel() * zero.element_size()
        if second is not None:  # (res, a, b, gate, out, stats) of an independent second problem of the same width
            r2, a2, b2, g2, o2, s2 = second
            if r2.dim() != 2 or r2.shape[1] != Cc or r2.stride() != res.stride() or r2.dtype != res.dtype:
                raise ValueError("residual_gate_stats: the second problem must have the width, row stride and dtype of the first")
            for t in (a2, b2, o2):
                if t is not None and (tuple(t.shape) != tuple(r2.shape) or t.stride() != r2.stride()):
                    raise ValueError("residual_gate_stats: second problem: a, b, out must match res in shape and strides")
            args.res2, args.a2, args.b2, args.gate2, args.out2, args.stats2 = dp(r2), dp(a2), dp(b2), dp(g2), dp(o2), dp(s2)
            args.M2 = r2.shape[0]
        _lib.check(lib.svdq_residual_gate_stats(C.byref(args), _stream()), "residual_gate_stats")

    @staticmethod
    def gemv_awq(in_feats, kernel, scaling_factors, zeros, m, n, k, group_size, bias=None, out_chunks=1):
        """reference: csrc/ops.h:123-145 -> gemv_awq (src/kernels/awq/gemv_awq.cu:253-286): allocates and returns
        the output, shape ``in_feats.shape[:-1] + (n,)``.  ``kernel`` is the checkpoint's ``qweight`` as stored
        ([n/4, k/2] int32); ``zeros`` are the scaled zeros.  ``bias`` (extension) fuses the module's 16-bit
        ``output.add_(bias)``; ``out_chunks`` = c (extension) de-interleaves the output into c contiguous ``[n/c]``
        vectors (element j goes to ``(j % c) * n/c + j // c``), the order ``emb.view(B, -1, c).permute(2, 0, 1)`` reads."""
        lib = _lib.load()
        if in_feats.dtype not in _DT or scaling_factors.dtype != in_feats.dtype or zeros.dtype != in_feats.dtype:
            raise ValueError("gemv_awq: in_feats, scaling_factors and zeros must share one 16-bit dtype")
        if in_feats.shape[-1] != k or in_feats.numel() != m * k:
            raise ValueError("gemv_awq: in_feats must hold m rows of k features")
        if kernel.numel() * kernel.element_size() * 2 != n * k:
            raise ValueError("gemv_awq: kernel must hold n*k 4-bit codes")
        G = (k + group_size - 1) // group_size
        if scaling_factors.dim() != 2 or scaling_factors.shape[1] != n or scaling_factors.shape[0] < G or zeros.shape != scaling_factors.shape:
            raise ValueError("gemv_awq: scaling_factors / zeros must be [>= k/group_size, n]")
        x2 = in_feats.reshape(m, k)
        if x2.stride(1) != 1:
            x2 = x2.contiguous()
        out = torch.empty(*in_feats.shape[:-1], n, dtype=in_feats.dtype, device=in_feats.device)
        a = _lib.GemvAwqArgs()
        a.x, a.qweight, a.scales, a.zeros = x2.data_ptr(), _ptr(kernel), _ptr(scaling_factors), _ptr(zeros)
        a.bias, a.out = _ptr(bias), out.data_ptr()
        if not x2.is_cuda:
            raise RuntimeError("nunchaku_amd ops need GPU tensors (there is no CPU path)")
        a.M, a.N, a.K, a.ldx = m, n, k, x2.stride(0) if m > 1 else k
        a.group_size, a.dtype, a.out_chunks = group_size, _DT[in_feats.dtype], int(out_chunks)
        _lib.check(lib.svdq_gemv_awq(C.byref(a), _stream()), "gemv_awq")
        return out

    @staticmethod
    def gemv_awq_batched(in_feats, layers):
        """Extension: the AWQ GEMVs of ``layers`` (``AWQW4A16Linear``-like objects: ``qweight, wscales, wzeros, bias,
        out_features, in_features, group_size, out_chunks``) on the same single-row ``in_feats`` in ONE launch per 80
        layers; returns the list of outputs (``[1, out_features]`` each, views of one buffer)."""
        lib = _lib.load()
        if in_feats.dtype not in _DT or in_feats.numel() != in_feats.shape[-1] or not in_feats.is_cuda:
            raise ValueError("gemv_awq_batched: in_feats must be one row of a 16-bit GPU tensor")
        x = in_feats.reshape(1, -1).contiguous()
        k = x.shape[1]
        total = sum(l.out_features for l in layers)
        buf = torch.empty(total, dtype=x.dtype, device=x.device)
        outs, off = [], 0
        arr = (_lib.GemvAwqArgs * len(layers))()
        for a, l in zip(arr, layers):
            if l.in_features != k or l.wscales.dtype != x.dtype:
                raise ValueError("gemv_awq_batched: every layer must take in_feats' width and dtype")
            o = buf[off:off + l.out_features]
            off += l.out_features
            outs.append(o.view(1, -1))
            a.x, a.qweight, a.scales, a.zeros = x.data_ptr(), _ptr(l.qweight), _ptr(l.wscales), _ptr(l.wzeros)
            a.bias, a.out = _ptr(l.bias), o.data_ptr()
            a.M, a.N, a.K, a.ldx = 1, l.out_features, k, k
            a.group_size, a.dtype, a.out_chunks = l.group_size, _DT[x.dtype], int(getattr(l, "out_chunks", 1))
        for s0 in range(0, len(layers), 80):
            n = min(80, len(layers) - s0)
            _lib.check(lib.svdq_gemv_awq_batched(C.cast(C.byref(arr[s0]), C.POINTER(_lib.GemvAwqArgs)), n, _stream()), "gemv_awq_batched")
        return outs

    @staticmethod
    def attention_fp16(q, k, v, o, scale):
        """reference: csrc/ops.h:114-121 -> kernels::attention_fp16 (attention.cu:11-94): ``q`` / ``k`` / ``v`` are the opaque
        [1, H, T_pad, 128] buffers a ``gemm_w4a4(..., out_q, out_k, out_v, attn_tokens)`` call filled, ``o`` is the linear
        [1, T_pad, H*128] output.  Non-causal, no mask.  (The reference kernel is fp16-only; this one takes bf16 too.)
        V is needed key-contiguous by this library's kernel: one transposing copy (the fused path, ``out_vt``, has none)."""
        for name, t in (("q", q), ("k", k), ("v", v)):
            if t.dim() != 4 or t.shape[0] != 1 or t.shape[-1] != 128 or t.stride(-1) != 1:
                raise NotImplementedError(f"attention_fp16: {name} must be [1, H, T_pad, 128] (batch 1, head_dim 128)")
        H, T = q.shape[1], q.shape[2]
        if tuple(o.shape) != (1, T, H * 128) or not o.is_contiguous():
            raise ValueError("attention_fp16: o must be a contiguous [1, T_pad, H*128] tensor")
        vt = v[0].transpose(1, 2).contiguous()  # [H, 128, T]
        # which key rows are real tokens: remembered per storage by the gemm_w4a4 calls that filled k (one per stream)
        kv_valid = None
        segs = sorted((r0, n, pad) for r0, (n, pad) in (_packed_rows.get(k) or {}).items() if r0 < T)
        if any(n < pad for _, n, pad in segs):
            if len(segs) > 2 or segs[0][0] != 0:
                raise NotImplementedError("attention_fp16: at most two padded token streams ([text | image]) are supported")
            kv_valid = (segs[0][1],) if len(segs) == 1 else (segs[0][1], segs[1][0], segs[1][0] + segs[1][1])
        _Ops.attention(q[0].transpose(0, 1), k[0].transpose(0, 1), vt, o[0].view(T, H, 128), scale, kv_valid=kv_valid)

    @staticmethod
    def attention(q, k, vt, out, scale, zero=None, quant=None, kv_valid=None, q_prescaled=False):
        """Non-causal attention, head_dim 128 (role of the reference's ``ops.attention_fp16``, csrc/ops.h:114-121
        -> attention.cu:11-94).  Strided views, no copies: ``q``/``k``/``out`` are ``[L, H, 128]`` (any token and head
        stride, unit channel stride), ``vt`` is ``[H, 128, L]`` with unit token stride (V transposed, as the QKV
        GEMM's ``out_vt`` writes it).  L must be a multiple of 128 (pad the buffers; ``kv_valid = (n,)`` or ``(n0, start1, end1)``
        masks the padded keys: keys ``[0, n0)`` and ``[start1, end1)`` are real; padded V^T columns must be finite).
        ``q_prescaled``: the producer of ``q`` multiplied it by ``scale * log2(e)`` before rounding (``gemm_w4a4(q_scale=...)``):
        the kernel then runs its faster geometry without losing accuracy (``svdq_attention_args.geometry``)."""
        lib = _lib.load()
        for name, t in (("q", q), ("k", k), ("vt", vt), ("out", out)):
            if name == "out" and t is None and quant is not None:
                continue  # fused quantiser: the 16-bit output is not needed
            if t is None or t.dim() != 3 or t.stride(2) != 1:
                raise ValueError(f"attention: {name} must be a 3-D view with unit innermost stride")
            if not t.is_cuda:
                raise RuntimeError("nunchaku_amd ops need GPU tensors (there is no CPU path)")
        if q.dtype not in _DT or k.dtype != q.dtype or vt.dtype != q.dtype or (out is not None and out.dtype != q.dtype):
            raise ValueError("attention: q, k, vt, out must share one 16-bit dtype")
        L, H, D = q.shape
        if tuple(k.shape) != (L, H, D) or (out is not None and tuple(out.shape) != (L, H, D)) or tuple(vt.shape) != (H, D, L):
            raise ValueError("attention: expected q/k/out [L, H, D] and vt [H, D, L]")
        a = _lib.AttentionArgs()
        a.q, a.k, a.vt = q.data_ptr(), k.data_ptr(), vt.data_ptr()
        a.ldq, a.q_hs = q.stride(0), q.stride(1)
        a.ldk, a.k_hs = k.stride(0), k.stride(1)
        if out is not None:
            a.out, a.ldo, a.o_hs = out.data_ptr(), out.stride(0), out.stride(1)
        if quant is not None:
            # extension: emit the following output projection's quantised activation directly (dict: act, ascales,
            # lora_act (pre-zeroed), smooth, lora_down, R [, smooth2, lora_down2, split_rows])
            Kq = H * D
            if quant["act"].numel() != L * Kq * 3 // 4 or quant["ascales"].numel() != (Kq // 64) * L:
                raise ValueError("attention: quant['act'] / quant['ascales'] must be the [L, 3K/4] / [K/64, L] operand images")
            a.qact, a.qscales, a.qlora_act = _ptr(quant["act"]), _ptr(quant["ascales"]), _ptr(quant.get("lora_act"))
            if quant.get("lora_act") is not None and quant["lora_act"].dtype == torch.int64:
                a.qlora_act_format = _lib.LORA_ACT_Q32
            a.qsmooth, a.qlora_down, a.qR = _ptr(quant["smooth"]), _ptr(quant.get("lora_down")), int(quant.get("R", 0))
            a.qsmooth2, a.qlora_down2 = _ptr(quant.get("smooth2")), _ptr(quant.get("lora_down2"))
            fq = _packed_fragments(quant.get("lora_down"), True, Kq, a.qR)
            fq2 = _packed_fragments(quant.get("lora_down2"), True, Kq, a.qR)
            if fq is not None and (quant.get("lora_down2") is None or fq2 is not None):
                a.qlora_down_packed, a.qlora_down_packed2 = _ptr(fq), _ptr(fq2)
            a.qsplit_rows = int(quant.get("split_rows", 0))
        a.vt_hs, a.ldvt = vt.stride(0), vt.stride(1)
        a.L, a.H, a.head_dim, a.dtype = L, H, D, _DT[q.dtype]
        a.scale = float(scale)
        a.q_prescaled = 1 if q_prescaled else 0  # Q already carries scale * log2(e) (gemm_w4a4(q_scale=...)): `scale` is not applied again
        if kv_valid is not None:
            kv = tuple(int(v) for v in kv_valid)
            a.kv_len0 = kv[0]
            if len(kv) == 3:
                a.kv_start1, a.kv_end1 = kv[1], kv[2]
            elif len(kv) != 1:
                raise ValueError("attention: kv_valid must be (n,) or (n0, start1, end1)")
        if zero is not None:  # scratch cleared by the same launch (see residual_gate_stats)
            if not zero.is_cuda or not zero.is_contiguous() or (zero.numel() * zero.element_size()) % 16:
                raise ValueError("attention: zero must be a contiguous GPU tensor of a multiple of 16 bytes")
            a.zero_ptr, a.zero_bytes = zero.data_ptr(), zero.numel() * zero.element_size()
        a.geometry = _Ops.attention_geometry if L % 256 == 0 else 0
        if _Ops.attention_use_workspace and L % 256 == 0:
            # (ABI 20: a fused quantiser of rank 48 .. 160 runs its low-rank down projection split when the workspace also holds its 16-bit output image)
            need = int(lib.svdq_attention_workspace_bytes_for(C.byref(a))) if (quant is not None and _Ops.attention_split_lowrank) else 0
            ws = _workspace(q.device, "attention", min_bytes=need)
            ws.check("attention")
            a.workspace, a.workspace_bytes = ws.buf.data_ptr(), ws.buf.numel()
            if not _Ops.attention_split_lowrank:  # (A/B and tests: the in-epilogue passes even if an earlier launch grew the buffer)
                a.workspace_bytes = min(a.workspace_bytes, int(lib.svdq_attention_workspace_bytes()))
            a.status = None if ws.status is None else ws.status.data_ptr()
        _lib.check(lib.svdq_attention(C.byref(a), _stream()), "attention")


class _Utils:
    """reference: csrc/pybind.cpp:118-123 -- logging / sm_75 toggles; no-ops here."""

    @staticmethod
    def set_log_level(level: str):
        return None

    @staticmethod
    def set_faster_i2f_mode(mode: str):
        return None

    @staticmethod
    def disable_memory_auto_release():
        return None

    @staticmethod
    def trim_memory():
        return None


ops = _Ops()
utils = _Utils()
