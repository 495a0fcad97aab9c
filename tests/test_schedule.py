"""Host-side replay of the GEMM's persistent / stream-K schedule (the same GemmSchedule code the kernel runs,
compiled for the host in libsvdq_amd.so): coverage and ownership invariants on CPU, no GPU needed."""
import ctypes as C

import numpy as np
import pytest

from nunchaku_amd import _lib


def schedule(M_pad, N, K, cus, ws, geometry=1):
    """geometry 1: 256 x 128 tiles, one workgroup per CU; 2: 128 x 128 tiles, two per CU (svdq_gemm_args.geometry)"""
    lib = _lib.load()
    n = lib.svdq_gemm_schedule_ex(M_pad, N, K, cus, ws, geometry, None, 0)
    assert n >= 0
    if geometry == 1:
        assert lib.svdq_gemm_schedule(M_pad, N, K, cus, ws, None, 0) == n  # the short form is geometry 1
    buf = (C.c_int32 * (6 * max(n, 1)))()
    assert lib.svdq_gemm_schedule_ex(M_pad, N, K, cus, ws, geometry, buf, n) == n
    return np.frombuffer(buf, dtype=np.int32)[: 6 * n].reshape(n, 6).copy()


SHAPES = [(4096, 3072, 3072), (4096, 9216, 3072), (4096, 3072, 12288), (4608, 3072, 12288), (4608, 12288, 3072),
          (512, 3072, 3072), (512, 3072, 12288), (256, 128, 128), (256, 384, 1024), (2560, 1152, 2048)]


@pytest.mark.parametrize("geometry", [1, 2])
@pytest.mark.parametrize("M_pad,N,K", SHAPES)
@pytest.mark.parametrize("cus,ws", [(256, 0), (256, 1), (64, 1), (24, 1), (8, 1)])
def test_every_k_step_of_every_tile_is_computed_exactly_once(M_pad, N, K, cus, ws, geometry):
    seg = schedule(M_pad, N, K, cus, ws, geometry)
    tiles, KP = (M_pad // (256 if geometry == 1 else 128)) * (N // 128), K // 128
    cover = np.zeros((tiles, KP), np.int32)
    for pos, tile, k0, k1, slot, contrib in seg:
        assert 0 <= tile < tiles and 0 <= k0 < k1 <= KP
        cover[tile, k0:k1] += 1
    assert (cover == 1).all()
    # exactly one owner (the segment that reaches KP) per tile; it waits for all the other segments of its tile
    for t in range(tiles):
        parts = seg[seg[:, 1] == t]
        owners = parts[parts[:, 3] == KP]
        assert len(owners) == 1
        assert owners[0, 5] == len(parts) - 1
        # contributors sit at lower positions than the owner (they run concurrently: one workgroup per CU)
        assert (parts[parts[:, 3] < KP][:, 0] < owners[0, 0]).all()
    # partial slots are unique and below 2 * grid
    slots = seg[seg[:, 4] >= 0][:, 4]
    assert len(set(slots.tolist())) == len(slots)
    grid = seg[:, 0].max() + 1
    assert grid <= max(cus, 1) * geometry  # one or two workgroups per CU
    assert (slots < 2 * grid).all() if len(slots) else True
    # the arrival counters of the remainder tiles fit the workspace header (1023 words) and the slabs its body
    assert grid - 1 < 1023 and 2 * grid * (256 // geometry) * 128 * 4 + 4096 <= _lib.load().svdq_gemm_workspace_bytes() or cus < 256


def test_stream_k_only_where_it_pays():
    # long K with a half-empty last round: split; short K: whole tiles (split overhead ~10 K-steps)
    long_k = schedule(4096, 3072, 12288, 256, 1)
    assert (long_k[:, 3] - long_k[:, 2] < 96).any()
    short_k = schedule(4096, 3072, 3072, 256, 1)
    assert ((short_k[:, 2] == 0) & (short_k[:, 3] == 24)).all()
    # a workgroup publishes the head of the next tile before it takes up its own owner duty
    for pos in np.unique(long_k[:, 0]):
        mine = long_k[long_k[:, 0] == pos]
        partial = mine[(mine[:, 2] > 0) | (mine[:, 3] < 96)]
        if len(partial) == 2:
            assert partial[0, 3] < 96 and partial[1, 3] == 96


def test_launches_without_stream_k_run_whole_rounds_on_fewer_workgroups():
    """K = 3072 launches do not split their remainder (the split costs more than it saves); they run on ceil(tiles / rounds)
    workgroups, rounded up to a multiple of 8, so that no round is nearly empty: the idle CUs' power budget goes to the busy
    ones (measured: -2 % on the denoise step)."""
    for (M_pad, N, K), grid in (((4608, 9216, 3072), 216), ((4608, 3072, 3072), 216), ((4608, 12288, 3072), 248), ((4096, 3072, 3072), 192)):
        seg = schedule(M_pad, N, K, 256, 1)
        assert ((seg[:, 2] == 0) & (seg[:, 3] == K // 128)).all()  # whole tiles only
        assert seg[:, 0].max() + 1 == grid
        per_wg = np.bincount(seg[:, 0])
        assert per_wg.max() - per_wg.min() <= 1 and per_wg.max() == -(-(M_pad // 256) * (N // 128) // grid)
    # fewer tiles than CUs: one workgroup per tile; long K: stream-K on all CUs
    assert schedule(512, 3072, 3072, 256, 1)[:, 0].max() + 1 == 48
    assert schedule(4608, 3072, 12288, 256, 1)[:, 0].max() + 1 == 256


def test_invalid_shapes_are_rejected():
    lib = _lib.load()
    assert lib.svdq_gemm_schedule(100, 128, 128, 256, 0, None, 0) == -1
    assert lib.svdq_gemm_schedule_ex(256, 128, 128, 256, 0, 6, None, 0) == -1  # the replay knows geometries 1, 2 and 3 (row runs)
    assert lib.svdq_gemm_schedule(256, 128, 64, 256, 0, None, 0) == -1


# ---------------------------------------------------------------------------------------------------------------------
# svdq_attention's persistent schedule (attention.hip: AttnSchedule), replayed on the host by the same code
# ---------------------------------------------------------------------------------------------------------------------
def _attn_schedule(L, H, cus):
    lib = _lib.load()
    n = lib.svdq_attention_schedule(L, H, cus, None, 0)
    if n <= 0:
        return n, None
    buf = (C.c_int32 * (6 * n))()
    assert lib.svdq_attention_schedule(L, H, cus, buf, n) == n
    return n, np.ctypeslib.as_array(buf).reshape(n, 6).copy()


@pytest.mark.parametrize("L,H,cus", [(4608, 24, 256), (256, 3, 256), (1024, 3, 256), (512, 130, 256), (4096, 24, 256),
                                     (4608, 24, 304), (2048, 5, 64), (256, 1, 256), (32768, 25, 256), (768, 7, 8)])
def test_attention_schedule_covers_every_tile_once_and_owners_agree(built_lib, L, H, cus):
    n, seg = _attn_schedule(L, H, cus)
    ntiles, tasks = L // 64, H * (L // 256)
    assert n > 0
    G = int(seg[:, 0].max()) + 1
    assert G <= cus and (G % 8 == 0 or G < 8) and G <= tasks * ntiles // 2
    cover = np.zeros((tasks, ntiles), dtype=np.int32)
    for g, task, j0, j1, owner, last in seg:
        assert 0 <= j0 < j1 <= ntiles and j0 % 2 == 0 and j1 % 2 == 0  # even cuts: the tile loop is unrolled by two
        cover[task, j0:j1] += 1
    assert (cover == 1).all()
    per_wg = np.bincount(seg[:, 0], weights=seg[:, 3] - seg[:, 2], minlength=G)
    assert per_wg.max() - per_wg.min() <= 2  # balanced to one tile pair
    whole = tasks // G  # every workgroup first takes `whole` complete tasks (task f * G + g), then its share of the split remainder
    first_seg = {}  # workgroup -> index of the first segment of its remainder run
    count = {}
    for i, (g, task, j0, j1, owner, last) in enumerate(seg):
        k = count.get(int(g), 0)
        count[int(g)] = k + 1
        if k < whole:
            assert task == k * G + g and j0 == 0 and j1 == ntiles and owner == -1 and last == -1
        elif k == whole:
            first_seg[int(g)] = i
    owners = {int(task): int(g) for g, task, j0, j1, owner, last in seg if j0 == 0}
    assert len(owners) == tasks
    contributors = {}
    for i, (g, task, j0, j1, owner, last) in enumerate(seg):
        if j0 > 0:
            assert first_seg[int(g)] == i, "a workgroup publishes at most once: the FIRST segment of its remainder run"
            assert owner == owners[int(task)] and owner < g
            contributors.setdefault(int(task), []).append(int(g))
        else:
            assert owner == -1
            if j1 < ntiles:
                assert i + 1 == len(seg) or seg[i + 1][0] != g, "a split task's owner segment is its workgroup's LAST"
    for g, task, j0, j1, owner, last in seg:
        if j0 == 0:
            cs = contributors.get(int(task), [])
            if j1 < ntiles:
                assert cs == list(range(g + 1, last + 1))  # exactly the workgroups g+1 .. last, each once
            else:
                assert last == -1 and cs == []


def test_attention_schedule_plain_grid_cases_and_errors(built_lib):
    lib = _lib.load()
    assert lib.svdq_attention_schedule(8192, 8, 256, None, 0) == 0   # 256 tasks on 256 CUs: whole rounds
    assert lib.svdq_attention_schedule(384, 4, 256, None, 0) == 0    # L % 256 != 0: the 4-wave kernel
    assert lib.svdq_attention_schedule(200, 4, 256, None, 0) == -1
    assert lib.svdq_attention_schedule(256, 0, 256, None, 0) == -1


# row runs: the schedule of a GELU_QUANT launch with a next-layer low-rank branch (gemm_w4a4.hip "row runs", GemmSchedule::init_runs)
@pytest.mark.parametrize("M_pad,N", [(4608, 12288), (4096, 12288), (1536, 12288), (4608, 3072), (2560, 1152), (9216, 12288)])
@pytest.mark.parametrize("cus", [256, 64, 24, 8])
def test_row_runs_cover_every_tile_once_and_stay_inside_a_row_block(M_pad, N, cus):
    lib = _lib.load()
    TM, TN = M_pad // 256, N // 128
    n = lib.svdq_gemm_schedule_ex(M_pad, N, 3072, cus, 1, 3, None, 0)
    if TM * TN < 2 * cus:
        assert n == -1  # fewer than two tiles per workgroup: the plain schedule
        return
    seg = schedule(M_pad, N, 3072, cus, 1, geometry=3)
    assert len(seg) == TM * TN and sorted(seg[:, 1].tolist()) == list(range(TM * TN))  # every tile once, whole (no K split)
    assert (seg[:, 2] == 0).all() and (seg[:, 3] == 3072 // 128).all()
    grid = seg[:, 0].max() + 1
    assert grid <= max(cus, TM)  # (more row blocks than slots: one whole-row run per row block, the surplus workgroups queue)
    longest = 0
    for pos in range(grid):
        t = seg[seg[:, 0] == pos][:, 1]
        if len(t) == 0:
            continue
        assert (np.diff(t) == 1).all()                 # consecutive column tiles ...
        assert t[0] // TN == t[-1] // TN               # ... of ONE row block (row-major tile ids)
        longest = max(longest, len(t))
    # no more rounds than the plain schedule needs when the runs fit the slots
    rounds = -(-TM * TN // cus)
    assert longest >= min(rounds, TN)
    if rounds <= TN and TM * -(-TN // rounds) <= cus:
        assert longest == rounds


# svdq_attention_plan: which kernel a (masked) attention launch takes -- host only
def _attn_plan(L, valid=None, q_prescaled=1, geometry=0):
    lib = _lib.load()
    a = _lib.AttentionArgs()
    a.L, a.H, a.head_dim, a.q_prescaled, a.geometry = L, 24, 128, q_prescaled, geometry
    if valid:
        a.kv_len0 = valid[0]
        if len(valid) == 3:
            a.kv_start1, a.kv_end1 = valid[1], valid[2]
    out = (C.c_int32 * 4)()
    assert lib.svdq_attention_plan(C.byref(a), out) == 0
    return list(out)


def test_attention_plan_puts_masked_launches_on_the_fast_geometry(built_lib):
    assert _attn_plan(4608) == [2, 0, 0, 0]                      # the headline shape: geometry 2, no mask
    assert _attn_plan(4608, q_prescaled=0) == [1, 0, 0, 0]       # a raw Q: geometry 1 (a second rounding of Q otherwise)
    # FLUX 1360 x 768: 512 + 4080 tokens, padding at the end: 71 full tiles -> the loop runs 70, tiles 70 (full) and 71 (48 real keys) are extras
    assert _attn_plan(4608, (4592,)) == [2, 1, 0, 70]
    # Qwen-Image 1664 x 928 with a 37-token prompt: [37 | pad to 256 | 6032 | pad to 6400]: the image run [4, 98) is the main segment, the
    # text tile 0 and the image's last tile 98 (16 real keys) are extras, tiles 1..3 and 99 are never touched
    assert _attn_plan(6400, (37, 256, 6288)) == [2, 1, 4, 98]
    assert _attn_plan(512, (300, 384, 500)) == [2, 1, 0, 4]
    assert _attn_plan(512, (64, 256, 257)) == [1, 0, 0, 0]       # no run of two full tiles: geometry 1
    assert _attn_plan(4608, (4592,), q_prescaled=0) == [1, 0, 0, 0]
    assert _attn_plan(4608, (4592,), q_prescaled=0, geometry=2) == [2, 1, 0, 70]   # explicit geometry 2 scales Q itself
    assert _attn_plan(4608, (4592,), geometry=1) == [1, 0, 0, 0]
    assert _attn_plan(384, (300,)) == [1, 0, 0, 0]               # L % 256 != 0: the 4-wave variant of geometry 1
