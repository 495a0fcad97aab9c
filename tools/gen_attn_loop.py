#!/usr/bin/env python3
"""The tile loop of the 4-wave x 64-row attention kernel (attention.hip, geometry 2) as assembly with explicit registers.

    python tools/gen_attn_loop.py        # writes nunchaku_amd/csrc/attention_loop64_{bf16,fp16}.inc

Why assembly (profiles/r3_attention_geometry2.txt): one wave per SIMD owns the whole 512-register file and nothing
but its own instruction stream fills the shadow of an MFMA.  The C++ version of this loop (slots separated by sched_barrier) runs
~60 cycles per MFMA where the measured price list (profiles/r3_mfma_filler_prices.txt: <= 5 plain VALU per MFMA nearly free, v_exp /
v_permlane32_swap two slots, v_dot2c +19 once) allows 37-43; the formulation that drops the 64 score-scaling v_fma per tile (Q
prescaled by scale * log2 e, score accumulators started at -max) and would let the row maxima into the slots makes hipcc's register
allocator answer with ~100 AGPR<->AGPR copies per two tiles.  Here the registers are assigned by hand and the hazards the compiler
used to cover are covered by construction (comments at each).

One call = the steady-state iterations j = s46 .. j_end - 2 of a segment (every tile that has a successor; the segment's last tile
and everything around the loop stay C++).  Iteration j:  S'(j+1) = K(j+1) Q'^T - mc  beside  P(j) = exp2(S'(j)) -> 16-bit fragments,
then  O += V^T(j) P(j)  beside the row maxima of S'(j+1);  the 8 LDS-DMA pieces of K(j+2) / V^T(j+1) issue from inside the first
slots; at the end the (rare) move of the reference point, vmcnt(0), the workgroup barrier.  Two bodies (buffer parity), as in C++.

Register plan (the C++ side pins its asm operands to the same registers, attention.hip):
  v[0:63]   score set A   (row tile rt, key half kt) -> v[16 (2 rt + kt) ..+15]        v[64:127]  score set B
  v[128:159] P fragments  (rt, key step ks) -> v[128 + 4 (4 rt + ks) ..+3] = {x0, x1, y0, y1}
  v[160:175] fragment ring (4 x 4)     v[176:183] exp temporaries     v[184:191] left to the compiler (it keeps spilled SGPRs there)
  v[192:223] start values of the score accumulators: -mc (or 0) of row tile rt in v[192 + 16 rt ..+15]
  v[224:231] ka: LDS offsets of the K fragment reads (d-step ds)      v[232:235] va: of the V^T reads (key step ks; + ATT_TILE)
  v[236:239] / v[240:243] per-lane source offsets of this wave's 4 K / 4 V^T DMA pieces
  v244 v245 mc (reference point of row tile 0 / 1, log2 units; -inf: no finite score yet)   v246 v247 row maxima
  v248..v251 row sums l2a[0], l2b[0], l2a[1], l2b[1]        v252..v255 scratch
  a[0:127]  O: (rt, channel tile dt) -> a[16 (4 rt + dt) ..+15]        a[128:191] Q': (rt, ds) -> a[128 + 4 (8 rt + ds) ..+3]
  s40 LDS address of this wave's first piece of buffer 0's K     s[42:43] / s[44:45] K / V^T of the head     s46 j (in / out)
  s47 j_end     s48 bytes between two K tiles     s49..s67 scratch (s[64:65] / s[66:67]: sources of the DMA pieces)
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ATT_TILE = 16384
NEG_INF = "0xff800000"


def S(setb, rt, kt):
    return (64 if setb else 0) + 16 * (2 * rt + kt)


def PF(rt, ks):
    return 128 + 4 * (4 * rt + ks)


def O(rt, dt):
    return 16 * (4 * rt + dt)


def Q(rt, ds):
    return 128 + 4 * (8 * rt + ds)


FRAG, ETMP, MINIT, KA, VA, KDMA, VDMA, MC, MLOC, L2 = 160, 176, 192, 224, 232, 236, 240, 244, 246, 248
T0, T1, T2, T3 = 252, 253, 254, 255
UNITS = {"EXP": 2, "CVT": 1, "SWAP": 2.5, "RMAX": 1, "DMA": 1.5, "READ": 0.25, "DOT8": 0}


def vr(a, n=1):
    return f"v{a}" if n == 1 else f"v[{a}:{a + n - 1}]"


def ar(a, n=1):
    return f"a{a}" if n == 1 else f"a[{a}:{a + n - 1}]"


class Body:
    """One iteration for buffer parity `buf`: score set (buf ? B : A) is exponentiated, the other one accumulated."""

    def __init__(self, dt, buf, tag, lead=4, budget=5.0, rm_from=36, dma0=1, drop=0):
        self.dt, self.buf, self.tag = dt, buf, tag
        self.mfma = "v_mfma_f32_32x32x16_bf16" if dt == "bf16" else "v_mfma_f32_32x32x16_f16"
        self.lines = []
        self.reads = []      # fragment ids in issue order
        self.lead, self.budget, self.rm_from, self.dma0 = lead, budget, rm_from, dma0
        self.drop = drop  # timing experiments (wrong results): 1 = no vmcnt wait at the end of an iteration, 2 = no barrier, 4 = no DMA

    def e(self, s):
        self.lines.append(s)

    # ---- operations ------------------------------------------------------------------------------------------
    def frag_reg(self, fid):
        return FRAG + 4 * (self.reads.index(fid) % 4)

    def op_read(self, fid):
        kind, a, b = fid
        self.reads.append(fid)
        reg = self.frag_reg(fid)
        if kind == "K":  # K(j+1) lives in the OTHER buffer
            self.e(f"ds_read_b128 {vr(reg, 4)}, {vr(KA + a)} offset:{(self.buf ^ 1) * 2 * ATT_TILE + b * 8192}")
        else:
            self.e(f"ds_read_b128 {vr(reg, 4)}, {vr(VA + a)} offset:{self.buf * 2 * ATT_TILE + b * 4096}")

    def wait_frag(self, fid):
        newer = len(self.reads) - 1 - self.reads.index(fid)
        self.e(f"s_waitcnt lgkmcnt({newer})")

    def op_qk(self, ds, kt, rt):
        if rt == 0:
            self.wait_frag(("K", ds, kt))
        acc = S(self.buf ^ 1, rt, kt)
        c = vr(MINIT + 16 * rt, 16) if ds == 0 else vr(acc, 16)
        self.e(f"{self.mfma} {vr(acc, 16)}, {vr(self.frag_reg(('K', ds, kt)), 4)}, {ar(Q(rt, ds), 4)}, {c}")

    def op_pv(self, ks, dt, rt):
        if rt == 0:
            self.wait_frag(("V", ks, dt))
        self.e(f"{self.mfma} {ar(O(rt, dt), 16)}, {vr(self.frag_reg(('V', ks, dt)), 4)}, {vr(PF(rt, ks), 4)}, {ar(O(rt, dt), 16)}")

    def op_exp(self, rt, ks, i):
        bank = ETMP  # (one bank: a piece's conversions are issued before the next piece's exponentials)
        self.e(f"v_exp_f32 {vr(bank + i)}, {vr(S(self.buf, rt, ks >> 1) + 8 * (ks & 1) + i)}")

    def op_cvt(self, rt, ks, d):
        bank = ETMP
        dst = PF(rt, ks) + (d if d < 2 else 2 + (d - 2))
        src = bank + (2 * d if d < 2 else 4 + 2 * (d - 2))
        self.e(f"{'v_cvt_pk_bf16_f32' if self.dt == 'bf16' else 'v_cvt_pk_f16_f32'} {vr(dst)}, {vr(src)}, {vr(src + 1)}")

    def op_swap(self, rt, ks, d2):
        # hazard: a VALU write of either operand needs 2 wait states before v_permlane32_swap reads it
        self.e("s_nop 1")
        self.e(f"v_permlane32_swap_b32 {vr(PF(rt, ks) + d2)}, {vr(PF(rt, ks) + 2 + d2)}")

    def op_dot8(self, ks):
        one = "0x3f803f80" if self.dt == "bf16" else "0x3c003c00"
        dot = "v_dot2c_f32_bf16" if self.dt == "bf16" else "v_dot2c_f32_f16"
        for rt in range(2):
            for w, acc in ((0, L2 + 2 * rt), (1, L2 + 2 * rt + 1), (2, L2 + 2 * rt), (3, L2 + 2 * rt + 1)):
                self.e(f"{dot} {vr(acc)}, {one}, {vr(PF(rt, ks) + w)}")

    def op_rmax(self, rt, i):
        sn = lambda kt: S(self.buf ^ 1, rt, kt)
        if i == 0:
            self.e(f"v_max_f32 {vr(MLOC + rt)}, {vr(sn(0))}, {vr(sn(1))}")
        else:
            r, kt = 2 * ((i - 1) >> 1) + 1, (i - 1) & 1
            self.e(f"v_max3_f32 {vr(MLOC + rt)}, {vr(MLOC + rt)}, {vr(sn(kt) + r)}, {vr(sn(kt) + (r + 1 if r + 1 < 16 else r))}")

    def op_dma(self, i):
        if self.drop & 4:
            return
        # K(j+2) -> K half of buffer `buf` (K(j) was read from it one iteration ago), V^T(j+1) -> V half of the other buffer.
        # hazard: SALU write of M0 -> LDS-DMA needs one wait state
        if i < 4:
            self.e(f"s_add_u32 m0, s40, {self.buf * 2 * ATT_TILE + i * 1024}")
            self.e("s_nop 0")
            self.e(f"global_load_lds_dwordx4 {vr(KDMA + i)}, s[64:65]")
        else:
            self.e(f"s_add_u32 m0, s40, {(self.buf ^ 1) * 2 * ATT_TILE + ATT_TILE + (i - 4) * 1024}")
            self.e("s_nop 0")
            self.e(f"global_load_lds_dwordx4 {vr(VDMA + i - 4)}, s[66:67]")

    # ---- the rare move of the reference point of row tile rt (and of everything that is relative to it) -----
    def finish(self, rt):
        t = self.tag
        ml, mc = MLOC + rt, MC + rt
        e = self.e
        e(f"v_mov_b32 {vr(T0)}, {vr(ml)}")
        e("s_nop 1")
        e(f"v_permlane32_swap_b32 {vr(T0)}, {vr(ml)}")       # both lanes of a row see both half-row maxima
        e(f"v_max_f32 {vr(ml)}, {vr(T0)}, {vr(ml)}")
        # move when: a started row outgrew its reference point by more than 8 (log2 units), or a fresh row saw its first finite score
        e(f"v_cmp_lt_f32 vcc, 0x41000000, {vr(ml)}")                       # 8 < mloc
        e(f"v_cmp_eq_f32 s[50:51], s49, {vr(mc)}")                         # fresh: mc == -inf (s49)
        e(f"v_cmp_lt_f32 s[52:53], s49, {vr(ml)}")                         # finite: -inf < mloc
        e("s_and_b64 s[54:55], s[50:51], s[52:53]")
        e("s_or_b64 vcc, vcc, s[54:55]")
        e("s_and_b64 vcc, exec, vcc")
        e(f"s_cbranch_vccz .Lsvdqa_keep{rt}_{t}_%=")
        # shift = fresh ? (finite ? mloc : 0) : max(mloc, 0);  alpha = fresh ? 0 : exp2(-shift)
        e(f"v_max_f32 {vr(T0)}, 0, {vr(ml)}")
        e(f"v_cndmask_b32 {vr(T1)}, 0, {vr(ml)}, s[52:53]")
        e(f"v_cndmask_b32 {vr(T0)}, {vr(T0)}, {vr(T1)}, s[50:51]")          # T0 = shift
        e(f"v_exp_f32_e64 {vr(T1)}, -{vr(T0)}")
        e("s_nop 0")                                                       # trans result -> VALU use: one wait state
        e(f"v_cndmask_b32 {vr(T1)}, {vr(T1)}, 0, s[50:51]")                 # T1 = alpha
        # mc = (fresh ? 0 : mc) + shift, but a fresh row without a finite score stays fresh
        e(f"v_cndmask_b32 {vr(T2)}, {vr(mc)}, 0, s[50:51]")
        e(f"v_add_f32 {vr(T2)}, {vr(T2)}, {vr(T0)}")
        e("s_andn2_b64 s[56:57], s[50:51], s[52:53]")
        e(f"v_mov_b32 {vr(T3)}, s49")
        e(f"v_cndmask_b32 {vr(mc)}, {vr(T2)}, {vr(T3)}, s[56:57]")
        # start value of the next score accumulators: -mc, or 0 while fresh
        e(f"v_cmp_eq_f32 vcc, s49, {vr(mc)}")
        e(f"v_sub_f32 {vr(T2)}, 0, {vr(mc)}")
        e(f"v_cndmask_b32 {vr(T2)}, {vr(T2)}, 0, vcc")
        for r in range(16):
            e(f"v_mov_b32 {vr(MINIT + 16 * rt + r)}, {vr(T2)}")
        # the scores of tile j+1 were accumulated against the old reference point
        for kt in range(2):
            for r in range(16):
                x = S(self.buf ^ 1, rt, kt) + r
                e(f"v_sub_f32 {vr(x)}, {vr(x)}, {vr(T0)}")
        e(f"v_mul_f32 {vr(L2 + 2 * rt)}, {vr(L2 + 2 * rt)}, {vr(T1)}")
        e(f"v_mul_f32 {vr(L2 + 2 * rt + 1)}, {vr(L2 + 2 * rt + 1)}, {vr(T1)}")
        # O of this row tile.  hazard: MFMA write of an AGPR tile -> v_accvgpr_read needs up to 18 wait states (the last PV MFMA may be
        # a few instructions back)
        e("s_nop 15")
        e("s_nop 7")
        tmps = (T0, T2, T3)
        # (T0 = shift is dead from here on: three rotating temporaries keep the read -> multiply -> write chains apart)
        for i in range(0, 64, 3):
            grp = [O(rt, 0) + k for k in range(i, min(i + 3, 64))]
            for k, a in enumerate(grp):
                e(f"v_accvgpr_read_b32 {vr(tmps[k])}, {ar(a)}")
            for k, a in enumerate(grp):
                e(f"v_mul_f32 {vr(tmps[k])}, {vr(tmps[k])}, {vr(T1)}")
            for k, a in enumerate(grp):
                e(f"v_accvgpr_write_b32 {ar(a)}, {vr(tmps[k])}")
        e(f".Lsvdqa_keep{rt}_{t}_%=:")

    # ---- one iteration -------------------------------------------------------------------------------------------
    def build(self):
        mf = [("QK", ds, kt, rt) for ds in range(8) for kt in range(2) for rt in range(2)]
        first_pv = {}
        for ks in range(4):
            first_pv[ks] = len(mf)
            mf += [("PV", ks, dt, rt) for dt in range(4) for rt in range(2)]
        n = len(mf)
        slots = [[] for _ in range(n)]
        used = [0.0] * n
        pre = []

        def place_read(slot, fid):
            (pre if slot < 0 else slots[slot]).append(("READ", fid))
            if slot >= 0:
                used[slot] += UNITS["READ"]

        for ds in range(8):
            for kt in range(2):
                place_read((ds * 2 + kt) * 2 - self.lead, ("K", ds, kt))
        for ks in range(4):
            for dt in range(4):
                place_read(first_pv[ks] + dt * 2 - self.lead, ("V", ks, dt))
        ex = []
        for ks in range(4):
            for rt in range(2):
                ex += [("EXP", rt, ks, i) for i in range(8)] + [("CVT", rt, ks, d) for d in range(4)] + [("SWAP", rt, ks, d2) for d2 in range(2)]
        rm = [("RMAX", rt, i) for i in range(17) for rt in range(2)]
        dot_slot = {first_pv[ks] + 7: ks for ks in range(4)}
        dma_slot = {self.dma0 + 2 * i: i for i in range(8)}
        done_slot = {}
        for i in range(n):
            if i in dot_slot:
                slots[i].append(("DOT8", dot_slot[i]))
                continue
            if i in dma_slot:
                slots[i].append(("DMA", dma_slot[i]))
                used[i] += UNITS["DMA"]
            while ex and used[i] + UNITS[ex[0][0]] <= self.budget:
                op = ex.pop(0)
                used[i] += UNITS[op[0]]
                slots[i].append(op)
                if op[0] == "SWAP" and op[3] == 1:
                    done_slot[(op[1], op[2])] = i
            while i >= self.rm_from and rm and used[i] + UNITS["RMAX"] <= self.budget:
                used[i] += UNITS["RMAX"]
                slots[i].append(rm.pop(0))
        assert not ex, f"{len(ex)} softmax operations do not fit"
        for ks in range(4):
            for rt in range(2):
                assert done_slot[(rt, ks)] < first_pv[ks], f"P fragment ({rt}, {ks}) is ready in slot {done_slot[(rt, ks)]}, needed in {first_pv[ks]}"
        self.plan = dict(pre=list(pre), slots=[list(s) for s in slots], tail=list(rm), mfmas=list(mf))  # (for tests/test_generators.py)

        def emit(op):
            k = op[0]
            if k == "READ": self.op_read(op[1])
            elif k == "EXP": self.op_exp(*op[1:])
            elif k == "CVT": self.op_cvt(*op[1:])
            elif k == "SWAP": self.op_swap(*op[1:])
            elif k == "RMAX": self.op_rmax(*op[1:])
            elif k == "DMA": self.op_dma(op[1])
            elif k == "DOT8": self.op_dot8(op[1])

        e = self.e
        e(f"; ---- iteration, buffer parity {self.buf}")
        # sources of this iteration's DMA pieces: K tile min(j + 2, j_end - 1) (past the segment's end: a re-fetch into a dead buffer),
        # V^T tile j + 1
        e("s_add_u32 s59, s46, 2")
        e("s_sub_u32 s60, s47, 1")
        e("s_min_u32 s59, s59, s60")
        e("s_mul_i32 s61, s59, s48")
        e("s_mul_hi_u32 s62, s59, s48")
        e("s_add_u32 s64, s42, s61")
        e("s_addc_u32 s65, s43, s62")
        e("s_add_u32 s59, s46, 1")
        e("s_lshl_b32 s59, s59, 7")
        e("s_add_u32 s66, s44, s59")
        e("s_addc_u32 s67, s45, 0")
        for op in pre:
            emit(op)
        for i, (kind, a, b, rt) in enumerate(mf):
            if kind == "QK":
                self.op_qk(a, b, rt)
            else:
                self.op_pv(a, b, rt)
            for op in slots[i]:
                emit(op)
        for op in rm:  # row-maximum operations that found no slot
            emit(op)
        for rt in range(2):
            self.finish(rt)
        if not self.drop & 1:
            e("s_waitcnt vmcnt(0)")      # this wave's DMA pieces have landed ...
        if not self.drop & 2:
            e("s_barrier")               # ... and everybody's; every read of this iteration's buffers is done
        e("s_add_u32 s46, s46, 1")
        return self.lines


def build(dt, **kw):
    """the loop: body 0, exit test, body 1, back.  A segment has an even number of tiles and the loop runs all but the last one: it
    always leaves after a body 0, with S'(j_end - 1) in score set B."""
    out = ["; ---- svdq attention tile loop, geometry 2 (generated by tools/gen_attn_loop.py) ----", f"s_mov_b32 s49, {NEG_INF}", ".Lsvdqa_top_%=:"]
    b0 = Body(dt, 0, "a", **kw)
    out += b0.build()
    out += ["s_add_u32 s58, s46, 1", "s_cmp_ge_u32 s58, s47", "s_cbranch_scc1 .Lsvdqa_done_%="]
    b1 = Body(dt, 1, "b", **kw)
    out += b1.build()
    out += ["s_branch .Lsvdqa_top_%=", ".Lsvdqa_done_%=:"]
    return out, (b0, b1)


def emit(path, dt, **kw):
    lines, _ = build(dt, **kw)
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_attn_loop.py -- do not edit.\n")
        for ln in lines:
            f.write('"' + ln + '\\n"\n')
    return len(lines)


if __name__ == "__main__":
    csrc = os.path.join(ROOT, "nunchaku_amd", "csrc")
    n = emit(os.path.join(csrc, "attention_loop64_bf16.inc"), "bf16")
    emit(os.path.join(csrc, "attention_loop64_fp16.inc"), "fp16")
    print(f"wrote attention_loop64_{{bf16,fp16}}.inc ({n} lines each)")
