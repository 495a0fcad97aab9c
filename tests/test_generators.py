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


def test_wave_tile_loop_and_epilogue_in_tree_match_their_generator(tmp_path):
    """round 6: the 128 x 64 wave tile kernel's loop (tools/gen_gemm_loop3.py, product options) and its generated epilogue"""
    import gen_gemm_loop3 as G3

    for dt, mfma in (("bf16", "v_mfma_f32_32x32x16_bf16"), ("fp16", "v_mfma_f32_32x32x16_f16")):
        out = tmp_path / f"gemm_loop3_{dt}.inc"
        G3.emit(str(out), mfma, G3.PRODUCT_OPTS)
        assert out.read_text() == open(os.path.join(CSRC, out.name)).read(), f"{out.name} is stale: run python tools/gen_gemm_loop3.py"
        out = tmp_path / f"gemm_epi3_{dt}.inc"
        G3.emit_epilogue(str(out), dt)
        assert out.read_text() == open(os.path.join(CSRC, out.name)).read(), f"{out.name} is stale: run python tools/gen_gemm_loop3.py"


def _reg_range(tok):
    """'v[12:15]' / 'a7' / 's[62:63]' -> (file, first, last)"""
    m = re.match(r"^([vas])\[(\d+):(\d+)\]$", tok) or re.match(r"^([vas])(\d+)$", tok)
    if not m:
        return None
    g = m.groups()
    return (g[0], int(g[1]), int(g[2] if len(g) == 3 else g[1]))


def test_wave_tile_loop_register_plan_and_dma_discipline():
    """Static walk over the generated wave-tile loop: every body of the ring issues exactly 16 + 16 MFMAs, 256 v_fmac (each accumulator register once per group),
    36 LDS reads and the K-step's 10 LDS-DMA instructions; an LDS-DMA never issues right behind its M0 write; the registers the text names stay inside the plan the
    C++ side pins and clobbers (acc v[0:127], P / S v[128:191], inputs v[192:208] + the operand-load temporaries v[212:214], fragments / tuples a[0:119], the
    epilogue operands a[120:200]); the product MFMA is the unscaled form and the scale tile the K = 8 form (round 6)."""
    import gen_gemm_loop3 as G3

    g = G3.Gen3("v_mfma_f32_32x32x16_bf16", G3.PRODUCT_OPTS)
    lines = [ln for ln in g.build() if not ln.startswith(";")]
    assert not any("v_mfma_scale" in ln or "32x32x16_bf16" in ln for ln in lines)
    since_m0 = None
    for ln in lines:
        op = ln.split()[0]
        if op.endswith(":"):
            continue
        if op.startswith("buffer_load") and ln.rstrip().endswith(" lds"):
            assert since_m0 is not None and since_m0 >= 1, f"LDS-DMA right behind its M0 write: {ln}"
        if op.startswith("s_") and re.search(r"\bm0\b", ln.split(",")[0]):
            since_m0 = 0
        elif since_m0 is not None:
            since_m0 += 1
        for tok in re.findall(r"[vas]\[\d+:\d+\]|\b[va]\d+\b", ln):
            r = _reg_range(tok)
            if r is None:
                continue
            f, lo, hi = r
            if f == "v":
                assert hi <= 208 or 212 <= lo <= 214, f"VGPR outside the plan: {ln}"
            elif f == "a":
                assert hi <= 200, f"AGPR outside the plan: {ln}"
    bodies = [i for i, ln in enumerate(lines) if re.match(r"\.Lsvdq3_bin\d_\d+_%=:", ln)]
    assert len(bodies) == 4
    top = next(i for i, ln in enumerate(lines) if re.match(r"\.Lsvdq3_top_\d+_%=:", ln))
    end = next(i for i, ln in enumerate(lines) if ln.startswith("s_branch .Lsvdq3_top_"))
    assert top < bodies[0] and bodies[-1] < end
    for b, e in zip(bodies, bodies[1:] + [end]):
        body = [ln for ln in lines[b:e] if not ln.endswith(":")]
        ops = [ln.split()[0] for ln in body]
        assert ops.count("v_mfma_f32_32x32x64_f8f6f4") == 16 and ops.count("v_mfma_f32_32x32x8bf16_1k") == 16
        fm = [ln for ln in body if ln.startswith("v_fmac_f32")]
        assert len(fm) == 256
        dst = [int(re.match(r"v_fmac_f32 v(\d+),", ln).group(1)) for ln in fm]
        assert sorted(dst[:128]) == list(range(128)) and sorted(dst[128:]) == list(range(128)), "every accumulator once per group"
        assert sum(1 for o in ops if o.startswith("ds_read")) == 36
        assert sum(1 for ln in body if ln.startswith("buffer_load") and ln.endswith(" lds")) == 10
    # even bodies carry the wait + barrier, odd ones none (ring of four, option b2)
    for k, (b, e) in enumerate(zip(bodies, bodies[1:] + [end])):
        assert ("s_barrier" in lines[b:e]) == (k % 2 == 0)


def test_wave_tile_epilogue_walk():
    """The generated plain epilogue: 8 bias MFMAs + 16 low-rank MFMAs on every accumulator tile, v_permlane32_swap never within two instructions of a write of its
    operands, 16 stores of 16 bytes per wave with the offsets of the 8-wave kernel's store, fp16 clamps before its conversions, and no register outside
    acc v[0:127] / temporaries v[128:191] / inputs v[209:211] / a[120:200]."""
    import gen_gemm_loop3 as G3

    for dt in ("bf16", "fp16"):
        lines = [ln for ln in G3.GenEpi3(dt).build() if not ln.startswith(";")]
        mf = [ln for ln in lines if ln.startswith("v_mfma")]
        assert len(mf) == 8 + 16
        tiles = [int(re.match(r"\S+ v\[(\d+):", ln).group(1)) // 16 for ln in mf]
        assert sorted(tiles[:8]) == list(range(8)) and sorted(tiles[8:16]) == list(range(8)) and sorted(tiles[16:]) == list(range(8))
        stores = [ln for ln in lines if ln.startswith("global_store_dwordx4")]
        assert len(stores) == 16
        offs = sorted(int(m.group(1)) if (m := re.search(r"offset:(\d+)", ln)) else 0 for ln in stores)
        assert offs == sorted([0, 32, 64, 96] * 4)
        written = {}   # register -> index of the last VALU instruction that wrote it
        idx = 0
        for ln in lines:
            if ln.endswith(":"):
                continue
            op = ln.split()[0]
            toks = re.findall(r"[va]\[\d+:\d+\]|\b[va]\d+\b", ln)
            for tok in toks:
                f, lo, hi = _reg_range(tok)
                assert (f == "v" and (hi <= 191 or 209 <= lo <= 211)) or (f == "a" and 120 <= lo and hi <= 200), ln
            if op == "s_nop":
                idx += int(ln.split()[1]) + 1
                continue
            if op == "v_permlane32_swap_b32":
                for tok in toks:
                    f, lo, hi = _reg_range(tok)
                    for r in range(lo, hi + 1):
                        assert idx - written.get(r, -100) > 2, f"{dt}: {ln} reads v{r} too soon after its write"
            if op.startswith("v_") and toks and not op.startswith("v_cmp"):
                f, lo, hi = _reg_range(toks[0])
                if f == "v":
                    for r in range(lo, hi + 1):
                        written[r] = idx
            idx += 1
        if dt == "fp16":
            assert sum(1 for ln in lines if ln.startswith("v_med3_f32")) == 128
