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
