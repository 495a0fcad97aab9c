"""Host-side logic that needs no GPU: cache keys, workspace bookkeeping."""
import torch


def test_ids_versions_steps_aside_for_inference_tensors():
    from nunchaku_amd.models.flux import _ids_versions

    a, b = torch.zeros(4, 3), torch.zeros(2, 3)
    v0 = _ids_versions(a, b)
    assert v0 == (a._version, b._version)
    a.add_(1)
    assert _ids_versions(a, b) != v0  # an in-place write invalidates the key
    with torch.inference_mode():
        c = torch.zeros(4, 3)
    assert c.is_inference() and _ids_versions(c, b) is None and _ids_versions(a, c) is None


def test_status_words_are_not_recycled_while_fresh_ones_remain(monkeypatch):
    from nunchaku_amd import _C

    pool = torch.zeros(4, dtype=torch.int32)
    monkeypatch.setattr(_C, "_status_pool", pool)
    monkeypatch.setattr(_C, "_status_used", 0)
    monkeypatch.setattr(_C, "_status_free", [])
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    synced = []
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: synced.append(1))
    w0 = _C._status_word()
    _C._status_free.append(w0)          # its workspace was evicted; kernels may still be in flight
    others = [_C._status_word() for _ in range(3)]
    assert all(o.data_ptr() != w0.data_ptr() for o in others) and not synced
    w0[0] = 7
    again = _C._status_word()            # pool exhausted: the evicted word, behind a device synchronisation, zeroed
    assert again.data_ptr() == w0.data_ptr() and synced and int(again[0]) == 0
    assert _C._status_word() is None


def test_gemm_workspace_bytes_for_the_split_low_rank_down(built_lib):
    """ABI 20: svdq_gemm_workspace_bytes_for() adds the launch's 16-bit output image (M_pad * N * 2 bytes) exactly for the GELU_QUANT launches whose next-layer
    low-rank down projection can run split: fp32 accumulators, own rank 48 .. 160, next rank 48 .. 160 at a full round of tiles (geometry 0) or at any size
    (geometry 7), N a multiple of 256; every other launch gets the base size (no GPU needed: shapes and pointer alignment only)."""
    import ctypes as C

    from nunchaku_amd import _lib

    lib = _lib.load()
    base = lib.svdq_gemm_workspace_bytes()

    def need(**kw):
        a = _lib.GemmArgs()
        a.fuse, a.M_pad, a.N, a.K, a.R, a.R2 = _lib.FUSE_GELU_QUANT, 4608, 12288, 3072, 128, 128
        a.lora_act_in, a.lora_up = 0x1000, 0x2000  # (alignment only: never dereferenced)
        a.lora_act_format = _lib.LORA_ACT_F32
        for k, v in kw.items():
            setattr(a, k, v)
        return lib.svdq_gemm_workspace_bytes_for(C.byref(a))

    img = 4608 * 12288 * 2
    assert need() == base + img
    assert need(R2=160, R=144) == base + img
    assert need(M_pad=6400, wgt2=0x3000, lora_up2=0x4000) == base + 6400 * 12288 * 2      # the grouped Qwen-Image fc1 of the reference's 1664 x 928 gate
    assert need(R2=64) == base + img and need(R2=48, R=48) == base + img                 # every next rank beyond the carry's 32 (rank 32 + a rank-16 LoRA)
    assert need(R2=32) == base and need(R2=176) == base                                  # the carry's ranks; more than five rank blocks
    assert need(R=32) == base and need(R=176) == base                                    # own rank off the all-rank path
    assert need(M_pad=256, N=1024) == base and need(M_pad=256, N=1024, geometry=7) == base + 256 * 1024 * 2
    assert need(N=12288 + 128) == base                                                   # N % 256
    assert need(geometry=1) == base and need(geometry=6) == base                         # explicit geometries keep their kernels
    assert need(lora_act_format=_lib.LORA_ACT_Q32) == base                               # the deterministic mode keeps its integer atomics
    assert need(fuse=_lib.FUSE_NONE) == base
    assert need(lora_up=0x2004) == base                                                  # unaligned operand: no all-rank kernel
    assert lib.svdq_gemm_workspace_bytes_for(None) == base


def test_attention_workspace_bytes_for_the_split_low_rank_down(built_lib):
    """ABI 20: svdq_attention_workspace_bytes_for() adds the packed down projection(s) + the 16-bit output image exactly for a fused quantiser with fp32
    accumulators, rank 48 .. 160 and H * 128 a multiple of 256"""
    import ctypes as C

    from nunchaku_amd import _lib

    lib = _lib.load()
    base = lib.svdq_attention_workspace_bytes()

    def need(**kw):
        a = _lib.AttentionArgs()
        a.L, a.H, a.head_dim, a.qR = 4608, 24, 128, 128
        a.qact, a.qlora_act, a.qlora_down = 0x1000, 0x2000, 0x3000  # (which pointers are given: never dereferenced)
        a.qlora_act_format = _lib.LORA_ACT_F32
        for k, v in kw.items():
            setattr(a, k, v)
        return lib.svdq_attention_workspace_bytes_for(C.byref(a))

    K = 24 * 128
    img, pack = 4608 * K * 2, (K // 16) * 4 * 1024
    assert need() == base + pack + img
    assert need(qsmooth2=0x4000) == base + 2 * pack + img                      # joint attention: two down projections
    assert need(qR=48) == base + (K // 16) * 2 * 1024 + img and need(qR=160) == base + (K // 16) * 5 * 1024 + img
    assert need(qR=32) == base and need(qR=176) == base and need(qR=0) == base
    assert need(H=3) == base                                                   # K = 384: not a multiple of 256
    assert need(qact=None) == base and need(qlora_act_format=_lib.LORA_ACT_Q32) == base
    assert need(L=4608 + 128) == base
    assert lib.svdq_attention_workspace_bytes_for(None) == base


def test_workspace_grows_on_request_but_never_under_a_captured_graph(built_lib, monkeypatch):
    """ABI 20: a stream's workspace is replaced by a larger one the first time a launch asks for more (the 16-bit image of a split low-rank down projection);
    its status word moves over; a workspace a captured graph points at is never replaced, nor is one grown during a capture (the launch then takes the path the
    existing size allows)."""
    import collections
    import contextlib

    from nunchaku_amd import _C

    class _Stream:
        cuda_stream = 7

    capturing = [False]
    monkeypatch.setattr(_C, "_workspaces", collections.OrderedDict())
    monkeypatch.setattr(_C, "_status_pool", torch.zeros(8, dtype=torch.int32))
    monkeypatch.setattr(_C, "_status_used", 0)
    monkeypatch.setattr(_C, "_status_free", [])
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _Stream())
    monkeypatch.setattr(torch.cuda, "device", lambda *a, **k: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: capturing[0])

    class _Fake:  # stands in for a device buffer: only its size is looked at here
        def __init__(self, n):
            self.n = n

        def numel(self):
            return self.n

    monkeypatch.setattr(torch, "zeros", lambda n, dtype=None, device=None: _Fake(int(n)))
    dev = torch.device("cpu")
    base = _C._lib.load().svdq_gemm_workspace_bytes()
    ws0 = _C._workspace(dev)
    assert ws0.buf.numel() == base and _C._workspace(dev) is ws0 and _C._workspace(dev, min_bytes=base - 1) is ws0
    ws1 = _C._workspace(dev, min_bytes=base + 1000)
    assert ws1 is not ws0 and ws1.buf.numel() == base + 1000 and ws1.status is ws0.status     # replaced; the status word moved over
    assert _C._workspace(dev) is ws1 and _C._workspace(dev, min_bytes=base + 10) is ws1       # the larger one serves every later launch of the stream
    capturing[0] = True
    assert _C._workspace(dev, min_bytes=base + 5000) is ws1 and ws1.captured                    # no growth during a capture ...
    capturing[0] = False
    assert _C._workspace(dev, min_bytes=base + 5000) is ws1                                     # ... nor afterwards: a captured graph points at it
    assert len(_C._workspaces) == 1


def test_workspace_cache_is_bounded_by_bytes(built_lib, monkeypatch):
    """round 6: the per-kind LRU is bounded by bytes as well as entries -- split-launch sized workspaces (113-157 MB each) of many streams
    no longer add up to gigabytes; the entry in use always stays."""
    import collections
    import contextlib

    from nunchaku_amd import _C

    stream = [1]

    class _Stream:
        @property
        def cuda_stream(self):
            return stream[0]

    monkeypatch.setattr(_C, "_workspaces", collections.OrderedDict())
    monkeypatch.setattr(_C, "_status_pool", torch.zeros(64, dtype=torch.int32))
    monkeypatch.setattr(_C, "_status_used", 0)
    monkeypatch.setattr(_C, "_status_free", [])
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _Stream())
    monkeypatch.setattr(torch.cuda, "device", lambda *a, **k: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)

    class _Fake:
        def __init__(self, n):
            self.n = n

        def numel(self):
            return self.n

    monkeypatch.setattr(torch, "zeros", lambda n, dtype=None, device=None: _Fake(int(n)))
    dev = torch.device("cpu")
    big = 200 << 20
    for s in range(1, 7):
        stream[0] = s
        _C._workspace(dev, min_bytes=big)
    held = [k for k in _C._workspaces if k[2] == "gemm"]
    assert sum(_C._workspaces[k].buf.numel() for k in held) <= _C._WORKSPACE_BYTES_LIMIT and (0, 6, "gemm") in held and len(held) == 2, held
    # one entry alone above the limit is kept (it is in use)
    stream[0] = 9
    ws = _C._workspace(dev, min_bytes=_C._WORKSPACE_BYTES_LIMIT + 1)
    assert _C._workspaces[(0, 9, "gemm")] is ws and [k for k in _C._workspaces if k[2] == "gemm"] == [(0, 9, "gemm")]
    # ... and goes as soon as another stream needs room (the total must fit again)
    stream[0] = 10
    small = _C._workspace(dev)
    assert list(_C._workspaces) == [(0, 10, "gemm")]
    # growing an existing entry runs the same eviction
    stream[0] = 11
    _C._workspace(dev, min_bytes=300 << 20)
    stream[0] = 10
    grown = _C._workspace(dev, min_bytes=300 << 20)
    assert grown is not small and list(_C._workspaces) == [(0, 10, "gemm")]


def test_deterministic_levels_of_the_mode():
    """nunchaku_amd.mode: False | True / "strict" | "runs" (ABI 22); the level decides the int64 allocation (both) and which fixed-point format the GEMM wrapper names;
    the header's enumerators and the binding's constants agree."""
    import os
    import re

    import pytest

    from nunchaku_amd import _lib, mode

    assert mode.deterministic is False and mode.lora_act_words() == 1 and not mode.gemm_lora_act_format_runs()
    with mode.deterministic_mode():
        assert mode.deterministic is True and mode.lora_act_words() == 2 and not mode.gemm_lora_act_format_runs()
        t, zeroed = mode.alloc_lora_act(4, 32, "cpu")
        assert t.dtype == torch.int64 and t.shape == (4, 32) and not zeroed
        with mode.deterministic_mode("runs"):
            assert mode.deterministic == "runs" and mode.lora_act_words() == 2 and mode.gemm_lora_act_format_runs()
            assert mode.alloc_lora_act(4, 32, "cpu")[0].dtype == torch.int64
            with mode.deterministic_mode(False):
                assert mode.alloc_lora_act(4, 32, "cpu")[0].dtype == torch.float32
        assert mode.deterministic is True
    assert mode.deterministic is False
    mode.set_deterministic("strict")
    assert mode.deterministic is True
    mode.set_deterministic(False)
    with pytest.raises(ValueError):
        mode.set_deterministic("fast")
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "svdq_amd.h")).read()
    m = re.search(r"enum \{ SVDQ_LORA_ACT_F32 = (\d+), SVDQ_LORA_ACT_Q32 = (\d+), SVDQ_LORA_ACT_Q32_RUNS = (\d+) \}", hdr)
    assert m and tuple(int(x) for x in m.groups()) == (_lib.LORA_ACT_F32, _lib.LORA_ACT_Q32, _lib.LORA_ACT_Q32_RUNS)
    assert int(re.search(r"#define SVDQ_ABI_VERSION (\d+)", hdr).group(1)) == _lib.ABI_VERSION
