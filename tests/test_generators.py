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


def test_attention_loops_in_tree_match_their_generator(tmp_path):
    import gen_attn_loop as GA

    for dt in ("bf16", "fp16"):
        out = tmp_path / f"attention_loop64_{dt}.inc"
        GA.emit(str(out), dt)
        assert out.read_text() == open(os.path.join(CSRC, out.name)).read(), f"{out.name} is stale: run python tools/gen_attn_loop.py"


def test_attention_loop_plan_is_complete_and_ordered():
    """Every operation of an iteration exactly once; a fragment is read before the MFMA that consumes it and its ring register is not
    re-read into while it is still needed; a P fragment is finished before its first PV MFMA; the row maxima of S'(j+1) come behind
    its last MFMA with the 11 wait states of the MFMA -> VALU hazard to spare; the counted LDS waits are right; the assembly text
    only touches the registers of the documented plan."""
    import gen_attn_loop as GA

    for kw in ({}, {"lead": 2}, {"lead": 6, "budget": 6}, {"dma0": 9, "rm_from": 40}):
        lines, bodies = GA.build("bf16", **kw)
        for body in bodies:
            plan = body.plan
            seq = []  # program order: (slot, op)
            seq += [(-1, op) for op in plan["pre"]]
            for i, m in enumerate(plan["mfmas"]):
                seq.append((i, ("MFMA",) + m))
                seq += [(i, op) for op in plan["slots"][i]]
            seq += [(len(plan["mfmas"]), op) for op in plan["tail"]]
            pos = {}
            for n, (_, op) in enumerate(seq):
                assert op not in pos, (op, kw)
                pos[op] = n
            want = {("MFMA", "QK", ds, kt, rt) for ds in range(8) for kt in range(2) for rt in range(2)}
            want |= {("MFMA", "PV", ks, dt, rt) for ks in range(4) for dt in range(4) for rt in range(2)}
            want |= {("READ", ("K", ds, kt)) for ds in range(8) for kt in range(2)} | {("READ", ("V", ks, dt)) for ks in range(4) for dt in range(4)}
            want |= {("EXP", rt, ks, i) for rt in range(2) for ks in range(4) for i in range(8)}
            want |= {("CVT", rt, ks, d) for rt in range(2) for ks in range(4) for d in range(4)}
            want |= {("SWAP", rt, ks, d) for rt in range(2) for ks in range(4) for d in range(2)}
            want |= {("DOT8", ks) for ks in range(4)} | {("RMAX", rt, i) for rt in range(2) for i in range(17)} | {("DMA", i) for i in range(8)}
            assert set(pos) == want, (sorted(map(str, set(pos) ^ want))[:6], kw)
            reads = [op[1] for _, op in seq if op[0] == "READ"]
            for r, fid in enumerate(reads):
                users = [pos[("MFMA", "QK" if fid[0] == "K" else "PV", fid[1], fid[2], rt)] for rt in range(2)]
                assert pos[("READ", fid)] < min(users)
                if r + 4 < len(reads):  # the read that reuses this ring register comes behind the last user
                    assert pos[("READ", reads[r + 4])] > max(users), (fid, kw)
            last_qk = max(pos[("MFMA", "QK", ds, kt, rt)] for ds in range(8) for kt in range(2) for rt in range(2))
            mfma_pos = sorted(p for op, p in pos.items() if op[0] == "MFMA")
            first_rmax = min(pos[("RMAX", rt, i)] for rt in range(2) for i in range(17))
            assert sum(1 for p in mfma_pos if last_qk < p < first_rmax) >= 2, "row maxima too close behind the last score MFMA"
            for ks in range(4):
                first_pv = min(pos[("MFMA", "PV", ks, dt, rt)] for dt in range(4) for rt in range(2))
                last_pv = max(pos[("MFMA", "PV", ks, dt, rt)] for dt in range(4) for rt in range(2))
                for rt in range(2):
                    for i in range(8):
                        assert pos[("EXP", rt, ks, i)] < pos[("CVT", rt, ks, (i // 2))], (rt, ks, i)
                    for d in range(4):
                        assert pos[("CVT", rt, ks, d)] < pos[("SWAP", rt, ks, d % 2)] < first_pv
                    assert pos[("SWAP", rt, ks, 1)] < pos[("DOT8", ks)]
                assert pos[("DOT8", ks)] <= last_pv + 1 + len(plan["slots"][0]) * 0 + 8  # while the packed fragments are live
        text = "\n".join(lines)
        # counted waits: every s_waitcnt lgkmcnt(n) is followed by an MFMA whose fragment was the (n + 1)-th newest read at that point
        newest = []
        for idx, ln in enumerate(lines):
            m = re.match(r"ds_read_b128 v\[(\d+):", ln)
            if m:
                newest.append(int(m.group(1)))
            m = re.match(r"s_waitcnt lgkmcnt\((\d+)\)", ln)
            if m:
                want_reg = newest[-1 - int(m.group(1))]
                nxt = lines[idx + 1]
                assert f"v[{want_reg}:{want_reg + 3}]" in nxt, (ln, nxt)
            if ln.startswith("; ---- iteration"):
                newest = []
        vregs = {int(x) for x in re.findall(r"\bv(\d+)\b", text)} | {r for a, b in re.findall(r"v\[(\d+):(\d+)\]", text) for r in range(int(a), int(b) + 1)}
        assert not (vregs & set(range(184, 192))), "v184..v191 belong to the compiler (spilled SGPRs live there)"
        assert max(vregs) <= 255
        aregs = {r for a, b in re.findall(r"a\[(\d+):(\d+)\]", text) for r in range(int(a), int(b) + 1)} | {int(x) for x in re.findall(r"\ba(\d+)\b", text)}
        assert max(aregs) <= 191
