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


def _regs(tok):
    """registers named by one operand token: ('v', n) / ('a', n) / ('s', n)"""
    out = []
    m = re.fullmatch(r"-?([vas])\[(\d+):(\d+)\]", tok)
    if m:
        return [(m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1)]
    m = re.fullmatch(r"-?([vas])(\d+)", tok)
    if m:
        return [(m.group(1), int(m.group(2)))]
    if tok in ("vcc", "m0", "exec"):
        return [(tok, 0)]
    return out


def test_attention_loop_hazards_are_covered_by_construction():
    """Static walk over the generated assembly (straight-line order; the rare rescale blocks are part of it): the wait states a compiler's
    hazard recogniser would insert are there.  Checked: >= 2 wait states between a VALU write and a v_permlane32_swap that reads it; >= 1
    between an M0 write and the LDS-DMA that uses it; >= 1 between a v_exp_f32 and the instruction that reads its result; an MFMA result in
    VGPRs is not read by a VALU instruction before 12 wait states or two further MFMAs have passed; an AGPR tile an MFMA wrote is not
    read by v_accvgpr_read before 18 wait states; every ds_read result is covered by a counted s_waitcnt before its first reader."""
    import gen_attn_loop as GA

    for dt in ("bf16", "fp16"):
        lines, _ = GA.build(dt)
        last_valu, last_exp, last_mfma_v, last_mfma_a, last_m0 = {}, {}, {}, {}, None  # register -> (wait-state clock, mfma count) of its last writer
        pending_ds = {}  # VGPR -> index of the ds_read that will write it (cleared by a wait that covers it)
        ds_order = []
        clock, mfmas = 0, 0
        for ln in lines:
            if not ln or ln.startswith((";", ".")) or ln.endswith(":"):
                continue
            op, _, rest = ln.partition(" ")
            toks = [t.strip() for t in rest.split(",")] if rest else []
            if op == "s_nop":
                clock += int(toks[0]) + 1
                continue
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", rest)
                if m:
                    keep = int(m.group(1))
                    done = ds_order[: len(ds_order) - keep] if keep else ds_order[:]
                    for regs in done:
                        for r in regs:
                            pending_ds.pop(r, None)
                    ds_order = ds_order[len(ds_order) - keep:] if keep else []
                clock += 1
                continue
            if op.startswith("s_") and op not in ("s_add_u32", "s_mov_b32"):
                clock += 1
                if op in ("s_barrier", "s_branch") or op.startswith("s_cbranch"):
                    clock += 4
                continue
            dst = _regs(toks[0]) if toks else []
            srcs = [r for t in toks[1:] for r in _regs(t.split(" ")[0])]
            if op.startswith("v_permlane32_swap"):
                srcs = srcs + dst  # reads and writes both operands
            if op in ("v_fmac_f32",) or op.startswith("v_dot2c") or op.startswith("v_mfma") and False:
                srcs = srcs + dst
            # ---- checks on the readers
            for r in srcs:
                assert r not in pending_ds, f"{ln}: reads {r} before the ds_read that loads it is waited for"
            if op.startswith("v_permlane32_swap"):
                for r in srcs:
                    if r in last_valu:
                        assert clock - last_valu[r] >= 2, f"{ln}: {r} written {clock - last_valu[r]} wait states earlier"
            if op.startswith("global_load_lds"):
                assert last_m0 is not None and clock - last_m0 >= 1, ln
            is_mfma = op.startswith("v_mfma")
            is_valu = op.startswith("v_") and not is_mfma
            if is_valu:
                for r in srcs:
                    if r in last_exp:
                        assert clock - last_exp[r] >= 1, f"{ln}: reads the v_exp result {r} in the next issue slot"
                    if r[0] == "v" and r in last_mfma_v:
                        c0, m0_ = last_mfma_v[r]
                        assert clock - c0 >= 12 or mfmas - m0_ >= 2, f"{ln}: reads the MFMA result {r} too early"
            if op == "v_accvgpr_read_b32":
                for r in srcs:
                    if r in last_mfma_a:
                        assert clock - last_mfma_a[r][0] >= 18 or mfmas - last_mfma_a[r][1] >= 3, f"{ln}: reads {r} too early behind the MFMA that wrote it"
            # ---- bookkeeping on the writers
            clock += 1
            if op == "s_add_u32" and toks and toks[0] == "m0":
                last_m0 = clock
                continue
            if op.startswith("ds_read"):
                ds_order.append(dst)
                for r in dst:
                    pending_ds[r] = True
                continue
            if is_mfma:
                mfmas += 1
                for r in dst:
                    (last_mfma_v if r[0] == "v" else last_mfma_a)[r] = (clock, mfmas)
                    last_valu.pop(r, None)
                continue
            if is_valu:
                for r in dst + (srcs if op.startswith("v_permlane32_swap") else []):
                    last_valu[r] = clock
                    last_exp.pop(r, None)
                    last_mfma_v.pop(r, None)
                if op.startswith("v_exp_f32"):
                    for r in dst:
                        last_exp[r] = clock


def test_the_hazard_walk_notices_missing_wait_states():
    """The checker above is not vacuous: strip the generator's own wait states / counted waits and it objects."""
    import gen_attn_loop as GA

    orig = GA.build
    try:
        for strip in ("s_nop 1", "s_nop 0", "s_waitcnt lgkmcnt(1)"):
            def stripped(dt, _strip=strip, **kw):
                lines, bodies = orig(dt, **kw)
                return [ln for ln in lines if ln != _strip], bodies

            GA.build = stripped
            try:
                test_attention_loop_hazards_are_covered_by_construction()
            except AssertionError:
                continue
            raise AssertionError(f"removing every '{strip}' went unnoticed")
    finally:
        GA.build = orig


def test_gemm_loop_lds_dma_discipline():
    """Static walk over the generated GEMM loops (the staging block of the 256 x 128 geometry went wrong exactly here once):
    every LDS-DMA instruction is preceded -- since the last M0 write -- by at least one other instruction (the M0 -> LDS-DMA wait
    state); LDS-DMA in the global form carries no instruction offset (it would move the memory AND the LDS address); no LDS-DMA's
    address register is written again before the next one issues from a different register; the staging block exists for nw = 8
    only, sits directly behind the entry barrier, and moves 4 pieces of lora_act_in + 1 of lora_up + the bias per wave into
    [0, 32 KiB) / [32 KiB, 40 KiB) / 40 KiB of the staging region."""
    import gen_gemm_loop2 as G2

    for nw in (8, 4):
        g = G2.Gen("v_mfma_f32_32x32x16_bf16", "", nw)
        lines = [ln for ln in g.build() if not ln.startswith(";")]
        since_m0 = None  # instructions since the last M0 write (None: M0 not written yet)
        for ln in lines:
            op = ln.split()[0]
            if op.endswith(":"):
                continue
            is_dma = (op.startswith("buffer_load") and ln.rstrip().endswith(" lds")) or op.startswith("global_load_lds")
            if is_dma:
                assert since_m0 is not None and since_m0 >= 1, f"nw={nw}: LDS-DMA right behind its M0 write: {ln}"
                if op.startswith("global_load_lds"):
                    assert "offset:" not in ln, f"nw={nw}: {ln}"
            if re.search(r"\bm0\b", ln.split(",")[0]) and op.startswith("s_"):
                since_m0 = 0
            elif since_m0 is not None:
                since_m0 += 1
        stg = [i for i, ln in enumerate(lines) if ln.startswith("global_load_lds")]
        if nw == 4:
            assert not stg
            continue
        assert len(stg) == 6
        bar = max(i for i, ln in enumerate(lines[:stg[0]]) if ln == "s_barrier")
        assert not any(ln.startswith(("buffer_load", "ds_read", "v_mfma")) for ln in lines[bar + 1:stg[0]])  # nothing of the loop in between
        regs = [lines[i].split()[1].rstrip(",") for i in stg]
        assert len(set(regs[1:5])) == 4 and regs[0] == f"v{G2.OFF_A}"  # lora_up: 16 * lane; lora_act_in: one address register per piece
        block = lines[bar + 1:stg[-1] + 1]
        for i, r in zip(stg, regs):  # no write to an address register behind an LDS-DMA that read it
            assert not any(re.match(rf"v_\w+ {r},", ln) for ln in lines[i + 1:stg[-1] + 1]), (r, lines[i])
        assert sum(1 for ln in block if ln == "s_add_u32 m0, m0, 1024") == 3
        assert f"s_add_u32 m0, m0, {G2.STG_LU}" in block and f"s_add_u32 m0, s{G2.S_STGB}, {G2.STG_BIAS}" in block
        assert g.geo.nstage == 3 and g.geo.stage * 3 + 2 * 256 * 4 + 16 + G2.STG_BIAS + 256 <= 160 * 1024  # the kernel's LDS map (Geo<8>)
