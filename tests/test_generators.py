"""The generated kernel sources in the tree are what their generators emit, and the attention iteration plan is complete.

CPU-only: no GPU, no compiler.  (tools/gen_gemm_loop2.py -> gemm_loop2_*.inc: the GEMM main loops; tools/gen_attn_step.py ->
attention_step64.inc: the slot placement of one KV-tile iteration of the 4 x 64 attention kernel.)"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nunchaku_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_gemm_loops_in_tree_match_their_generator(tmp_path):
    import gen_gemm_loop2 as G2

    for name, mfma, nw in (("gemm_loop2_bf16.inc", "v_mfma_f32_32x32x16_bf16", 8), ("gemm_loop2_fp16.inc", "v_mfma_f32_32x32x16_f16", 8),
                           ("gemm_loop2_w4_bf16.inc", "v_mfma_f32_32x32x16_bf16", 4), ("gemm_loop2_w4_fp16.inc", "v_mfma_f32_32x32x16_f16", 4)):
        out = tmp_path / name
        G2.emit(str(out), mfma, "", nw=nw)
        assert out.read_text() == open(os.path.join(CSRC, name)).read(), f"{name} is stale: run python tools/gen_gemm_loop2.py"


def test_attention_step_in_tree_matches_its_generator():
    import gen_attn_step as GA

    assert GA.build() == open(os.path.join(CSRC, "attention_step64.inc")).read(), "attention_step64.inc is stale: run python tools/gen_attn_step.py"


def _ops(text):
    return re.findall(r"(A64_[A-Z0-9]+)(?:\(([^)]*)\))?", text)


def test_attention_step_plan_is_complete_and_ordered():
    """Every operation of the iteration exactly once; a fragment is read before the MFMA that consumes it; a P fragment is finished
    before its first PV MFMA; every row-maximum operation comes behind the last score MFMA; one MFMA per slot."""
    import gen_attn_step as GA

    for kw in ({}, {"lead": 2}, {"lead": 6, "budget": 6}, {"dma0": 9}):
        text = GA.build(**kw)
        ops = [(name, tuple(int(x) for x in args.split(",")) if args else ()) for name, args in _ops(text) if name not in ("A64_SB", "A64_STAMP")]
        pos = {}
        for i, op in enumerate(ops):
            assert op not in pos, f"{op} appears twice ({kw})"
            pos[op] = i
        want = {("A64_QK", (ds, kt, rt)) for ds in range(8) for kt in range(2) for rt in range(2)}
        want |= {("A64_PV", (ks, dt, rt)) for ks in range(4) for dt in range(4) for rt in range(2)}
        want |= {("A64_KREAD", (ds, kt)) for ds in range(8) for kt in range(2)} | {("A64_VREAD", (ks, dt)) for ks in range(4) for dt in range(4)}
        want |= {(n, (rt, ks, i)) for n in ("A64_FMA", "A64_EXP") for rt in range(2) for ks in range(4) for i in range(8)}
        want |= {("A64_CVT", (rt, ks, d)) for rt in range(2) for ks in range(4) for d in range(4)}
        want |= {("A64_SWAP", (rt, ks, d)) for rt in range(2) for ks in range(4) for d in range(2)}
        want |= {("A64_FIN", (rt, ks)) for rt in range(2) for ks in range(4)} | {("A64_DOT8", (ks,)) for ks in range(4)}
        want |= {("A64_RMAX", (rt, i)) for rt in range(2) for i in range(17)} | {("A64_SETTLE", ())}
        want |= {("A64_DMAK", (i,)) for i in range(4)} | {("A64_DMAV", (i,)) for i in range(4)}
        assert set(pos) == want, (sorted(set(pos) ^ want)[:8], kw)
        for ds in range(8):
            for kt in range(2):
                assert pos[("A64_KREAD", (ds, kt))] < pos[("A64_QK", (ds, kt, 0))] < pos[("A64_QK", (ds, kt, 1))]
        last_qk = max(p for (n, _), p in pos.items() if n == "A64_QK")
        for ks in range(4):
            first_pv = min(pos[("A64_PV", (ks, dt, rt))] for dt in range(4) for rt in range(2))
            last_pv = max(pos[("A64_PV", (ks, dt, rt))] for dt in range(4) for rt in range(2))
            for dt in range(4):
                assert pos[("A64_VREAD", (ks, dt))] < pos[("A64_PV", (ks, dt, 0))]
            for rt in range(2):
                assert pos[("A64_FIN", (rt, ks))] < first_pv, (rt, ks, kw)
                for i in range(8):  # exp behind its scaling, pack behind the exponentials, exchange behind the pack
                    assert pos[("A64_FMA", (rt, ks, i))] < pos[("A64_EXP", (rt, ks, i))] < pos[("A64_CVT", (rt, ks, i // 2))]
                for d in range(4):
                    assert pos[("A64_CVT", (rt, ks, d))] < pos[("A64_SWAP", (rt, ks, d % 2))] < pos[("A64_FIN", (rt, ks))]
            assert first_pv < pos[("A64_DOT8", (ks,))] and pos[("A64_FIN", (1, ks))] < pos[("A64_DOT8", (ks,))]
            assert pos[("A64_DOT8", (ks,))] >= last_pv - 1  # the packed fragments are still live: the slot of the step's last PV MFMA
        assert last_qk < pos[("A64_SETTLE", ())] < min(p for (n, _), p in pos.items() if n == "A64_RMAX")
        for line in text.splitlines():
            assert len(re.findall(r"A64_(?:QK|PV)\(", line)) <= 1, line
