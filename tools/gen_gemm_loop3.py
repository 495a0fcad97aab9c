#!/usr/bin/env python3
"""PROBE generator (round 6, VERDICT r5 #2 i): the W4A4 main loop with a 128 x 64 wave tile, ONE wave per SIMD.

    python tools/gen_gemm_loop3.py        # writes tools/ablate/gen/gemm_loop3_{bf16,fp16}.inc (used by tools/ablate/gemm128_probe.hip only)

Same arithmetic as tools/gen_gemm_loop2.py (one FP6 product MFMA + one 16-bit scale-tile MFMA + 16 v_fmac per 32 x 32 x 64 tile-group, groups in
ascending order: bit-identical accumulators), same workgroup tile (256 x 128), same LDS stage image (8 A chunks, 4 W chunks, two scale images = 38 912 B,
ring of three), same segment / prefetch protocol -- but FOUR waves per workgroup (2 along M x 2 along N), each owning 4 x 2 MFMA tiles:

  v[0:127]  acc (tile t = 4 ni + mi at 16 t)      v[128:143] P0  v[144:159] S0  v[160:175] P1  v[176:191] S1   (the VALU cannot read AGPRs)
  a[0:35] / a[36:71]     fragment buffers 0 / 1 (group parity): W fragments 2 x 6, A fragments 4 x 6      (MFMA A / B operands and ds_read destinations may be AGPRs)
  a[72:95] / a[96:119]   scale tuples 0 / 1: W 2 x 4, A 4 x 4 (k-slot 0 holds the scale, the rest stays zero)
  v192..v195 in: per-lane LDS offsets of A frags, W frags, as, ws     v196..v199 the same + 2 * STAGE
  v200 v201 in: per-lane DMA offsets of the wave's two A chunks (chunk w and chunk w + 4)   v202 v203 in: of streams X1, X2     v204 v205 MX exponents
  scalars as in gen_gemm_loop2.py.

Per K-step and wave: 32 MFMAs + 256 v_fmac + 36 LDS reads + 10 LDS-DMA instructions (two A chunks = 6 planes, one W chunk = 3 planes, one scale image / repeat)
against 16 + 128 + 24 + 5 for the 64 x 64 wave tile: per tile-group 20.9 vector-side instructions instead of 21.6, 3/4 of the fragment bytes read from LDS,
four waves at the K-step rendezvous instead of eight.  325 registers per wave: one wave per SIMD.
"""
import os

CHUNK, PLANE = 3072, 1024
NSTAGE = 3
A_BYTES = 8 * CHUNK
STAGE = A_BYTES + 4 * CHUNK + 1024 + 1024
DMA_PER_STEP = 10

ACC = 0
PBUF = [128, 160]
SBUF = [144, 176]
FRAG = [0, 36]        # AGPR
SCL = [72, 96]        # AGPR
IN_L = [192, 193, 194, 195]   # A, W, SA, SW (stages 0, 1)
HI_L = [196, 197, 198, 199]   # + 2 * STAGE
OFF_A, OFF_A2, OFF_X1, OFF_X2 = 200, 201, 202, 203
MXA, MXB = 204, 205
S_KP, S_DA, S_DX1, S_DX2, S_IX2 = 46, 47, 48, 49, 50
S_STEP, S_TOT4, S_KP4, S_TMP, S_TOT = 53, 55, 56, 57, 61
S_RING, S_NPRE, S_NCNT, S_LANDED = 58, 59, 60, 68
S_NA, S_NX1, S_NX2 = 62, 64, 66
SRD_A, SRD_X1, SRD_X2 = 72, 76, 80
S_KOFF, S_KOFF2, S_DSTEP, S_STG = 84, 85, 86, 87
S_NODMA = 51  # option "sp": this K-step issues no DMA
# option "ep" (the epilogue operands of a tile requested by the loop call): in: flags, DMA K-steps this call issues, operand bases, lane offsets; out: a[120:200]
S_EPF, S_EPN, S_ELA, S_ELU, S_EB = 51, 54, 88, 90, 92
V_LA, V_LU, V_BI = 206, 207, 208
V_EPT = 212           # v212 .. v214: + 32 / 64 / 96 rows of lora_act_in
EP_STEP = 4           # option "ep": the body of K-step 3 or 4 -- whichever runs on an EVEN ring stage -- requests the epilogue operands (a call of at most 4 K-steps: at its top)
S_EPSTEP = 52
A_EP = 120


def vr(a, n=1):
    return f"v{a}" if n == 1 else f"v[{a}:{a + n - 1}]"


def ar(a, n=1):
    return f"a{a}" if n == 1 else f"a[{a}:{a + n - 1}]"


def sr(a, n=1):
    return f"s{a}" if n == 1 else f"s[{a}:{a + n - 1}]"


class Gen3:
    def __init__(self, smfma, opts=""):
        self.smfma = smfma
        self.opts = set(o for o in opts.split("+") if o)
        self.lines = []
        self.label = 0
        self.nstage = 4 if "b2" in self.opts else 3   # "b2": ring of four, the K-step barrier in every other body (kstep_b2)

    def e(self, s):
        self.lines.append(s)

    def new_label(self, tag=""):
        self.label += 1
        return f".Lsvdq3_{tag}{self.label}_%="

    def lds(self, kind, stage):
        if stage < 2:
            return IN_L[kind], stage * STAGE
        return HI_L[kind], (stage - 2) * STAGE

    def frag_reads(self, buf, grp, stage):
        """the 18 LDS reads of one group: W0 A0 sw0 sa0 | A1 sa1 | A2 sa2 | A3 sa3 | W1 sw1 (six reads for the first tile-group, three per further one)"""
        def frag(kind, i, roff):
            base, imm = self.lds(kind, stage)
            f = FRAG[buf] + roff + 6 * i
            o = imm + i * CHUNK
            if grp == 0:
                return [f"ds_read_b128 {ar(f, 4)}, {vr(base)} offset:{o}", f"ds_read_b64 {ar(f + 4, 2)}, {vr(base)} offset:{o + PLANE}"]
            return [f"ds_read_b64 {ar(f, 2)}, {vr(base)} offset:{o + PLANE + 8}", f"ds_read_b128 {ar(f + 2, 4)}, {vr(base)} offset:{o + 2 * PLANE}"]

        def scale(kind, i, roff):
            base, imm = self.lds(kind, stage)
            return [f"ds_read_u16 {ar(SCL[buf] + roff + 4 * i)}, {vr(base)} offset:{imm + i * 128 + grp * 64}"]

        out = frag(1, 0, 0) + frag(0, 0, 12) + scale(3, 0, 0) + scale(2, 0, 8)
        for i in range(1, 4):
            out += frag(0, i, 12) + scale(2, i, 8)
        out += frag(1, 1, 0) + scale(3, 1, 0)
        assert len(out) == 18
        if "rd1" in self.opts:  # timing only: one LDS read per group
            return out[:1]
        return out

    def p_mfma(self, dst_buf, buf, t):
        ni, mi = t >> 2, t & 3
        w = FRAG[buf] + 6 * ni
        a = FRAG[buf] + 12 + 6 * mi
        if "mx" not in self.opts:
            # round 6: the product MFMA WITHOUT MX block scales: P = dot / 64 (both code images hold value / 8); the factor 32 the block exponents used to supply lives
            # in the weight-side scale image (svdq_repack_wscales: 32 x ws), so P S is the same fp32 product bit for bit.  The scaled form is a PAIR of instructions
            # (v_mfma_ld_scale_b32 + the MFMA, 16 bytes) reading two more VGPRs: 126.0 -> 121.9 / 110.0 -> 105.9 cycles per tile-group on the probe
            # (profiles/r6_gemm_wave_tile_probe.txt section 6).  "mx": the scaled pair of rounds 1-5 (needs the ABI 20 scale image: probes only)
            return f"v_mfma_f32_32x32x64_f8f6f4 {vr(PBUF[dst_buf], 16)}, {ar(w, 6)}, {ar(a, 6)}, 0 cbsz:2 blgp:2"
        return (f"v_mfma_scale_f32_32x32x64_f8f6f4 {vr(PBUF[dst_buf], 16)}, {ar(w, 6)}, {ar(a, 6)}, 0, "
                f"{vr(MXA)}, {vr(MXB)} op_sel_hi:[0,0,0] cbsz:2 blgp:2")

    def s_mfma(self, dst_buf, buf, t):
        ni, mi = t >> 2, t & 3
        if "k16" not in self.opts:
            # the legacy K = 8 form on the first two registers of each tuple (k-slots 0 and 4 hold the scale: the same S = 2 ws as, bit for bit).  Same 8 passes on
            # gfx950 (tools/ablate/mfma_k8_probe: 32.6 cycles either way), half the operand registers read: 130 -> 126 / 114 -> 110 cycles per tile-group on the
            # probe (profiles/r6_gemm_wave_tile_probe.txt section 5).  "k16": the K = 16 form of rounds 1-5
            op = "v_mfma_f32_32x32x8bf16_1k" if self.smfma.endswith("bf16") else "v_mfma_f32_32x32x8f16"
            return f"{op} {vr(SBUF[dst_buf], 16)}, {ar(SCL[buf] + 4 * ni, 2)}, {ar(SCL[buf] + 8 + 4 * mi, 2)}, 0"
        return f"{self.smfma} {vr(SBUF[dst_buf], 16)}, {ar(SCL[buf] + 4 * ni, 4)}, {ar(SCL[buf] + 8 + 4 * mi, 4)}, 0"

    def fma(self, t, pb, lo, hi):
        return [f"v_fmac_f32 {vr(ACC + 16 * t + r)}, {vr(PBUF[pb] + r)}, {vr(SBUF[pb] + r)}" for r in range(lo, hi)]

    # tile order inside a group: the first tile-group needs W0 A0, then A1 A2 A3 against W0, then W1 against A0..A3
    TILE_ORDER = [0, 1, 2, 3, 4, 5, 6, 7]   # t = 4 ni + mi

    def dma_units(self, stage_imm):
        """the K-step's 10 LDS-DMA instructions as separately placeable units (option "sp": one or two per tile-group slot behind the barrier instead of one burst --
        a single in-order wave pays every piece's acceptance time, MI355X_MICROARCH.md: ~60 cycles among MFMAs, 100-185 in a burst)"""
        ld = "buffer_load_dwordx4"
        def a(off, pl):
            return f"{ld} {vr(off)}, {sr(SRD_A, 4)}, {sr(S_KOFF)} offen" + (f" offset:{pl * PLANE}" if pl else "") + " lds"
        def x1(pl):
            return f"{ld} {vr(OFF_X1)}, {sr(SRD_X1, 4)}, {sr(S_KOFF)} offen" + (f" offset:{pl * PLANE}" if pl else "") + " lds"
        if "nodma" in self.opts:  # timing only
            return [[f"s_add_u32 {sr(S_DSTEP)}, {sr(S_DSTEP)}, 1"]] + [[] for _ in range(8)] + \
                   [[f"s_add_u32 {sr(S_KOFF)}, {sr(S_KOFF)}, {CHUNK}", f"s_add_u32 {sr(S_KOFF2)}, {sr(S_KOFF2)}, {sr(S_IX2)}"]]
        return [
            [f"s_add_u32 m0, {sr(S_DA)}, {stage_imm}", f"s_add_u32 {sr(S_DSTEP)}, {sr(S_DSTEP)}, 1", a(OFF_A, 0)],
            [a(OFF_A, 1)], [a(OFF_A, 2)],
            [f"s_add_u32 m0, m0, {4 * CHUNK}", "s_nop 0", a(OFF_A2, 0)],
            [a(OFF_A2, 1)], [a(OFF_A2, 2)],
            [f"s_add_u32 m0, {sr(S_DX1)}, {stage_imm}", "s_nop 0", x1(0)],
            [x1(1)], [x1(2)],
            [f"s_add_u32 m0, {sr(S_DX2)}, {stage_imm}", f"s_add_u32 {sr(S_KOFF)}, {sr(S_KOFF)}, {CHUNK}",
             f"{ld} {vr(OFF_X2)}, {sr(SRD_X2, 4)}, {sr(S_KOFF2)} offen lds", f"s_add_u32 {sr(S_KOFF2)}, {sr(S_KOFF2)}, {sr(S_IX2)}"],
        ]

    def dma_issue(self, stage_imm=None):
        if "nodma" in self.opts and stage_imm is not None:  # timing only: the in-loop DMAs are not issued (the prologue's are: operands of the first K-steps only)
            return [f"s_add_u32 {sr(S_DSTEP)}, {sr(S_DSTEP)}, 1", f"s_add_u32 {sr(S_KOFF)}, {sr(S_KOFF)}, {CHUNK}", f"s_add_u32 {sr(S_KOFF2)}, {sr(S_KOFF2)}, {sr(S_IX2)}"]
        def m0(dst, extra=0):
            if stage_imm is not None:
                return f"s_add_u32 m0, {sr(dst)}, {stage_imm + extra}"
            return f"s_add_u32 m0, {sr(dst)}, {sr(S_STG)}"
        ld = "buffer_load_dwordx4"
        out = [m0(S_DA), f"s_add_u32 {sr(S_DSTEP)}, {sr(S_DSTEP)}, 1"]
        out += [f"{ld} {vr(OFF_A)}, {sr(SRD_A, 4)}, {sr(S_KOFF)} offen" + (f" offset:{pl * PLANE}" if pl else "") + " lds" for pl in range(3)]
        out += [f"s_add_u32 m0, m0, {4 * CHUNK}", "s_nop 0"]
        out += [f"{ld} {vr(OFF_A2)}, {sr(SRD_A, 4)}, {sr(S_KOFF)} offen" + (f" offset:{pl * PLANE}" if pl else "") + " lds" for pl in range(3)]
        out += [m0(S_DX1), "s_nop 0"]
        out += [f"{ld} {vr(OFF_X1)}, {sr(SRD_X1, 4)}, {sr(S_KOFF)} offen" + (f" offset:{pl * PLANE}" if pl else "") + " lds" for pl in range(3)]
        out += [m0(S_DX2), f"s_add_u32 {sr(S_KOFF)}, {sr(S_KOFF)}, {CHUNK}",
                f"{ld} {vr(OFF_X2)}, {sr(SRD_X2, 4)}, {sr(S_KOFF2)} offen lds",
                f"s_add_u32 {sr(S_KOFF2)}, {sr(S_KOFF2)}, {sr(S_IX2)}"]
        return out

    def epilogue_operand_loads(self):
        """Option "ep" (the product loop): the tile's epilogue operands -- bias, this wave's rows of lora_act_in, its rows of lora_up (rank 32, fp32
        low-rank activations) -- are requested at the TOP of the loop call into AGPRs a[120:200] and land under the main loop: one wave per SIMD has nobody
        to hide an epilogue's memory round trip behind, and 80 more live registers in the epilogue made the compiler park the 128 accumulators in AGPRs
        (256 v_accvgpr moves per tile).  in: s51 flags (bit 0: low-rank operands, bit 1: bias), s[88:89] lora_act_in + (first row of the wave) * 128,
        s[90:91] lora_up + (first column of the wave) * 64, s[92:93] bias + (first column of the wave) * 2, v206 / v207 / v208 lane offsets
        ((lr * 32 + 8 h) * 4, (lr * 32 + 8 h) * 2, (32 h + lr) * 2).  a[120 + 8 (2 mi + u) + 4 j ..]: ranks 16 u + 8 h + 4 j .. + 3 of row 32 mi + lr;
        a[184 + 4 (2 ni + u) ..]: ranks 16 u + 8 h .. + 7 of column 32 ni + lr; a200: bias[32 h + lr]."""
        Lnol, Lnob = self.new_label("nola"), self.new_label("nobias")
        t = [V_EPT, V_EPT + 1, V_EPT + 2]   # row-tile offsets of the lora_act_in loads (set once at the top of the call: EP_SETUP)
        out = [f"s_bitcmp1_b32 {sr(S_EPF)}, 0", f"s_cbranch_scc0 {Lnol}"]
        for mi in range(4):
            off = V_LA if mi == 0 else t[mi - 1]
            for u in range(2):
                for j in range(2):
                    imm = 64 * u + 16 * j
                    out.append(f"global_load_dwordx4 {ar(A_EP + 8 * (2 * mi + u) + 4 * j, 4)}, {vr(off)}, {sr(S_ELA, 2)}" + (f" offset:{imm}" if imm else ""))
        for ni in range(2):
            for u in range(2):
                imm = 2048 * ni + 32 * u
                out.append(f"global_load_dwordx4 {ar(A_EP + 64 + 4 * (2 * ni + u), 4)}, {vr(V_LU)}, {sr(S_ELU, 2)}" + (f" offset:{imm}" if imm else ""))
        out += [f"{Lnol}:", f"s_bitcmp1_b32 {sr(S_EPF)}, 1", f"s_cbranch_scc0 {Lnob}",
                f"global_load_ushort {ar(A_EP + 80)}, {vr(V_BI)}, {sr(S_EB, 2)}", f"{Lnob}:"]
        return out

    def switch_segment(self):
        return [
            f"s_mov_b64 {sr(SRD_A, 2)}, {sr(S_NA, 2)}",
            f"s_mov_b64 {sr(SRD_X1, 2)}, {sr(S_NX1, 2)}",
            f"s_mov_b64 {sr(SRD_X2, 2)}, {sr(S_NX2, 2)}",
            f"s_mov_b32 {sr(S_KOFF)}, 0",
            f"s_mov_b32 {sr(S_KOFF2)}, 0",
        ]

    def kstep(self, j, exit_label, ool):
        e = self.e
        nj = (j + 1) % self.nstage
        reads_g1 = self.frag_reads(1, 1, j)
        reads_n0 = self.frag_reads(0, 0, nj)
        Lin = self.new_label(f"inst{j}_")
        e(f"{Lin}:")
        for q in range(16):
            t = self.TILE_ORDER[q & 7]
            pb = q & 1
            qn = q + 1
            nbuf, nt = (qn >> 3) & 1, self.TILE_ORDER[qn & 7]
            if q in (7, 15):
                e("s_waitcnt lgkmcnt(0)")
            e(self.p_mfma(pb ^ 1, nbuf, nt))
            misc = []
            if q < 6:
                misc = reads_g1[3 * q:3 * q + 3]
            elif q == 8 and "sp" in self.opts:
                # spread: the barrier here, the ten DMAs over slots 8 .. 15 ([2, 1, 1, 2, 1, 1, 1, 1]); a K-step without DMAs (the last self.nstage of the workgroup's
                # whole run) raises S_NODMA and every later slot of the body skips its unit
                Ltail, Lback, Lsw, Lbsw = (self.new_label(x) for x in ("tail", "back", "sw", "bsw"))
                units = self.dma_units(j * STAGE)
                self.spread = {8: units[0] + units[1], 9: units[2], 10: units[3], 11: units[4] + units[5], 12: units[6], 13: units[7], 14: units[8], 15: units[9]}
                misc = [
                    f"s_mov_b32 {sr(S_NODMA)}, 0",
                    f"s_cmp_lt_u32 {sr(S_STEP)}, {sr(S_TOT4)}",
                    f"s_cbranch_scc0 {Ltail}",
                    f"s_waitcnt vmcnt({(self.nstage - 2) * DMA_PER_STEP})",
                    "s_nop 0" if "nobar" in self.opts else "s_barrier",
                    f"s_cmp_eq_u32 {sr(S_STEP)}, {sr(S_KP4)}", f"s_cbranch_scc1 {Lsw}", f"{Lbsw}:"] + self.spread[8]
                ool += [f"{Lsw}:"] + self.switch_segment() + [f"s_branch {Lbsw}"]
                misc += [f"{Lback}:"] + reads_n0[0:3]
                ool += [f"{Ltail}:", f"s_mov_b32 {sr(S_NODMA)}, 1", "s_waitcnt vmcnt(0)", "s_barrier", f"s_branch {Lback}"]
            elif q == 8:
                Ltail, Lback, Lsw, Lbsw = (self.new_label(x) for x in ("tail", "back", "sw", "bsw"))
                misc = [
                    f"s_cmp_lt_u32 {sr(S_STEP)}, {sr(S_TOT4)}",
                    f"s_cbranch_scc0 {Ltail}",
                    f"s_waitcnt vmcnt({(self.nstage - 2) * DMA_PER_STEP})",
                    "s_nop 0" if "nobar" in self.opts else "s_barrier",
                    f"s_cmp_eq_u32 {sr(S_STEP)}, {sr(S_KP4)}", f"s_cbranch_scc1 {Lsw}", f"{Lbsw}:"] + self.dma_issue(j * STAGE)
                ool += [f"{Lsw}:"] + self.switch_segment() + [f"s_branch {Lbsw}"]
                misc += [f"{Lback}:"] + reads_n0[0:3]
                ool += [f"{Ltail}:", "s_waitcnt vmcnt(0)", "s_barrier", f"s_branch {Lback}"]
            elif 9 <= q <= 13:
                misc = reads_n0[3 * (q - 8):3 * (q - 8) + 3]
            elif q == 15:
                misc = [f"s_add_u32 {sr(S_STEP)}, {sr(S_STEP)}, 1"]
            for ln in misc:
                e(ln)
            if "sp" in self.opts and 9 <= q <= 15:
                Lskip = self.new_label("nd")
                e(f"s_cmp_eq_u32 {sr(S_NODMA)}, 0")
                e(f"s_cbranch_scc0 {Lskip}")
                for ln in self.spread[q]:
                    e(ln)
                e(f"{Lskip}:")
            for ln in self.fma(t, pb, 0, 8):
                e(ln)
            e(self.s_mfma(pb ^ 1, nbuf, nt))
            for ln in self.fma(t, pb, 8, 16):
                e(ln)
        e(f"s_cmp_lt_u32 {sr(S_STEP)}, {sr(S_KP)}")
        e(f"s_cbranch_scc0 {exit_label}")
        return Lin

    def kstep_b2(self, j, exit_label, ool):
        """Option "b2": ring of FOUR stages, ONE K-step of DMA per body, the workgroup barrier only in the even bodies.
        Even body (step s, stage j): slot 8 = s_waitcnt vmcnt(0) + barrier -- K-steps s + 1, s + 2 have landed for every wave and every wave is done with
        stages j - 1 and j -- then the DMA group of K-step s + 3 (-> stage j - 1) over slots 8 .. 15.  Odd body (step s + 1, stage j + 1): no wait, no
        barrier (its reads of stage j + 2 were covered by the even body's barrier); the DMA group of K-step s + 4 (-> stage j) over slots 0 .. 9.
        A group issues K-step d = S_DSTEP iff d <= step + 3 (after a loop entry the prologue has already issued it) and d < tot; d == kp switches the
        operand streams to the next segment.  Every body leaves exactly three K-steps of the stream ahead of the one it computed: the C++ side's `npre`."""
        e = self.e
        nj = (j + 1) % self.nstage
        reads_g1 = self.frag_reads(1, 1, j)
        reads_n0 = self.frag_reads(0, 0, nj)
        units = self.dma_units(((j + 3) % self.nstage) * STAGE)
        even = j % 2 == 0
        if even:
            spread = {8: units[0] + units[1], 9: units[2], 10: units[3], 11: units[4] + units[5], 12: units[6], 13: units[7], 14: units[8], 15: units[9]}
            gstart = 8
        else:
            spread = {q: units[q] for q in range(10)}
            gstart = 0
        if "burst" in self.opts:
            spread = {gstart: [ln for u in units for ln in u]}
        Lin = self.new_label(f"inst{j}_")
        e(f"{Lin}:")
        for q in range(16):
            t = self.TILE_ORDER[q & 7]
            pb = q & 1
            qn = q + 1
            nbuf, nt = (qn >> 3) & 1, self.TILE_ORDER[qn & 7]
            if q in (7, 15):
                e("s_waitcnt lgkmcnt(0)")
            e(self.p_mfma(pb ^ 1, nbuf, nt))
            if "pad" in self.opts and q in (2, 6, 10, 14):   # timing probe: what four more scalar issue slots per K-step cost
                e("s_nop 0")
            if "padv" in self.opts and q in (2, 6, 10, 14):  # ... four more VALU instructions
                e(f"v_mov_b32 {vr(215)}, {vr(215)}")
            if q == 8 and even:
                e("s_waitcnt vmcnt(0)")
                e("s_nop 0" if "nobar" in self.opts else "s_barrier")
            if q == gstart:
                Loff, Lsw, Lbsw = (self.new_label(x) for x in ("off", "sw", "bsw"))
                e(f"s_cmp_lt_u32 {sr(S_DSTEP)}, {sr(S_TOT)}")
                e(f"s_cbranch_scc0 {Loff}")
                e(f"s_cmp_eq_u32 {sr(S_DSTEP)}, {sr(S_KP)}")
                e(f"s_cbranch_scc1 {Lsw}")
                e(f"{Lbsw}:")
                ool += [f"{Lsw}:"] + self.switch_segment() + [f"s_branch {Lbsw}"]
                # past the end of the workgroup's operand stream: num_records = 0 -- the group's loads fetch nothing (no branch around each of them)
                ool += [f"{Loff}:"] + [f"s_mov_b32 {sr(b + 2)}, 0" for b in (SRD_A, SRD_X1, SRD_X2)] + [f"s_branch {Lbsw}"]
            if q in spread:
                for ln in spread[q]:
                    e(ln)
            misc = []
            if q < 6:
                misc = reads_g1[3 * q:3 * q + 3]
            elif q == 14 and even and "epmid" in self.opts:
                # behind this body's vmcnt(0) + barrier: the next vmcnt(0) is two bodies away, the loads' round trip fits in between
                Lep, Lepb = self.new_label("epreq"), self.new_label("epback")
                misc = [f"s_cmp_eq_u32 {sr(S_STEP)}, {sr(S_EPSTEP)}", f"s_cbranch_scc1 {Lep}", f"{Lepb}:"]
                ool += [f"{Lep}:"] + self.epilogue_operand_loads() + [f"s_branch {Lepb}"]
            elif 8 <= q <= 13:
                misc = reads_n0[3 * (q - 8):3 * (q - 8) + 3]
            elif q == 15:
                misc = [f"s_add_u32 {sr(S_STEP)}, {sr(S_STEP)}, 1"]
            for ln in misc:
                e(ln)
            for ln in self.fma(t, pb, 0, 8):
                e(ln)
            e(self.s_mfma(pb ^ 1, nbuf, nt))
            for ln in self.fma(t, pb, 8, 16):
                e(ln)
        e(f"s_cmp_lt_u32 {sr(S_STEP)}, {sr(S_KP)}")
        e(f"s_cbranch_scc0 {exit_label}")
        return Lin

    def build(self):
        e = self.e
        e("; ---- svdq gemm main loop, 128 x 64 wave tile, one wave per SIMD (generated by tools/gen_gemm_loop3.py) ----")
        if "ep" in self.opts:
            # 21 loads per wave, 16 of them a 16-byte piece of a 128-byte row per lane (32 lines per instruction): ~2 k cycles of the address unit per K = 3072
            # tile, paid inside the loop wherever they are issued -- option "epmid" asks from an even body of K-step 3 or 4 instead of here and measured 2 k cycles
            # WORSE (profiles/r6_gemm_wave_tile_probe.txt): at the top they overlap the pipeline fill.
            Lskip = self.new_label("eptop")
            for i in range(3):
                e(f"v_add_u32 {vr(V_EPT + i)}, {4096 * (i + 1)}, {vr(V_LA)}")
            if "epmid" in self.opts:
                e(f"s_cmp_gt_u32 {sr(S_KP)}, {EP_STEP}")
                e(f"s_cbranch_scc1 {Lskip}")
            for ln in self.epilogue_operand_loads():
                e(ln)
            e(f"{Lskip}:")
        for r in range(128):
            e(f"v_mov_b32 {vr(ACC + r)}, 0")
        for b in range(2):
            for tpl in range(6):
                for k in range(1, 4):
                    e(f"v_accvgpr_write_b32 {ar(SCL[b] + 4 * tpl + k)}, 0")
        e(f"v_mov_b32 {vr(MXA)}, 0x82828282")
        e(f"v_mov_b32 {vr(MXB)}, 0x81818181")
        for k in range(4):
            e(f"v_add_u32 {vr(HI_L[k])}, {2 * STAGE}, {vr(IN_L[k])}")
        e(f"s_mov_b32 {sr(S_STEP)}, 0")
        e(f"s_add_u32 {sr(S_TOT)}, {sr(S_KP)}, {sr(S_NCNT)}")
        e(f"s_mov_b32 {sr(S_TMP)}, {self.nstage}")
        e(f"s_sub_u32 {sr(S_TOT4)}, {sr(S_TOT)}, {sr(S_TMP)}")
        e(f"s_cselect_b32 {sr(S_TOT4)}, 0, {sr(S_TOT4)}")
        e(f"s_sub_u32 {sr(S_KP4)}, {sr(S_KP)}, {sr(S_TMP)}")
        e(f"s_cselect_b32 {sr(S_KP4)}, -1, {sr(S_KP4)}")
        e(f"s_mov_b32 {sr(S_DSTEP)}, {sr(S_NPRE)}")
        e(f"s_mul_i32 {sr(S_KOFF)}, {sr(S_NPRE)}, {CHUNK}")
        e(f"s_mul_i32 {sr(S_KOFF2)}, {sr(S_NPRE)}, {sr(S_IX2)}")
        e(f"s_mul_i32 {sr(S_TMP)}, {sr(S_NPRE)}, {STAGE}")
        e(f"s_add_u32 {sr(S_STG)}, {sr(S_RING)}, {sr(S_TMP)}")
        e(f"s_cmp_lt_u32 {sr(S_STG)}, {self.nstage * STAGE}")
        e(f"s_cselect_b32 {sr(S_TMP)}, 0, {self.nstage * STAGE}")
        e(f"s_sub_u32 {sr(S_STG)}, {sr(S_STG)}, {sr(S_TMP)}")
        Ltop, Ltopdone, Lnosw = self.new_label("top"), self.new_label("topdone"), self.new_label("nosw")
        e(f"{Ltop}:")
        e(f"s_cmp_lt_u32 {sr(S_DSTEP)}, 3")   # (three K-steps ahead: what every body of either ring maintains)
        e(f"s_cbranch_scc0 {Ltopdone}")
        e(f"s_cmp_lt_u32 {sr(S_DSTEP)}, {sr(S_TOT)}")
        e(f"s_cbranch_scc0 {Ltopdone}")
        e(f"s_cmp_eq_u32 {sr(S_DSTEP)}, {sr(S_KP)}")
        e(f"s_cbranch_scc0 {Lnosw}")
        for ln in self.switch_segment():
            e(ln)
        e(f"{Lnosw}:")
        for ln in self.dma_issue(None):
            e(ln)
        e(f"s_add_u32 {sr(S_STG)}, {sr(S_STG)}, {STAGE}")
        e(f"s_cmp_lt_u32 {sr(S_STG)}, {self.nstage * STAGE}")
        e(f"s_cselect_b32 {sr(S_STG)}, {sr(S_STG)}, 0")
        e(f"s_branch {Ltop}")
        e(f"{Ltopdone}:")
        Lpw = self.new_label("pw")
        e(f"s_cmp_gt_u32 {sr(S_LANDED)}, 0")
        e(f"s_cbranch_scc1 {Lpw}")
        e("s_waitcnt vmcnt(0)")
        e(f"{Lpw}:")
        e("s_barrier")
        entry = [self.new_label(f"in{j}_") for j in range(self.nstage)]
        for j in range(1, self.nstage):
            e(f"s_cmp_eq_u32 {sr(S_RING)}, {j * STAGE}")
            e(f"s_cbranch_scc1 {entry[j]}")
        body_in = [self.new_label(f"bin{j}_") for j in range(self.nstage)]
        Lexit = self.new_label("exit")
        for j in range(self.nstage):
            e(f"{entry[j]}:")
            if "epmid" in self.opts:
                e(f"s_mov_b32 {sr(S_EPSTEP)}, {3 if j % 2 == 1 else 4}")   # K-step 3 runs on stage (j + 3) % 4: even iff j is odd
            for ln in self.frag_reads(0, 0, j):
                e(ln)
            e("s_waitcnt lgkmcnt(0)")
            e(self.p_mfma(0, 0, self.TILE_ORDER[0]))
            e(self.s_mfma(0, 0, self.TILE_ORDER[0]))
            e("s_nop 7")
            e(f"s_branch {body_in[j]}")
        ool = []
        top = self.new_label("top_")
        e(f"{top}:")
        for j in range(self.nstage):
            start = len(self.lines)
            lin = self.kstep_b2(j, Lexit, ool) if "b2" in self.opts else self.kstep(j, Lexit, ool)
            self.lines = [ln.replace(lin, body_in[j]) for ln in self.lines[:start]] + [ln.replace(lin, body_in[j]) for ln in self.lines[start:]]
        e(f"s_branch {top}")
        for ln in ool:
            e(ln)
        e(f"{Lexit}:")
        if "ep" in self.opts:
            # the epilogue operands requested at the top of this call must have landed when the block ends (the compiler cannot see them).  vmcnt retires in
            # order and this call has issued S_EPN K-steps of DMA (10 instructions each) behind them: waiting for all but the youngest min(S_EPN, 3) x 10
            # proves them landed without draining the next tile's prefetch (a whole tile later they have long arrived: no wait in steady state)
            L3, L2, L1, Ld = (self.new_label(x) for x in ("ep3_", "ep2_", "ep1_", "epd_"))
            e(f"s_cmp_ge_u32 {sr(S_EPN)}, 3")
            e(f"s_cbranch_scc1 {L3}")
            e(f"s_cmp_eq_u32 {sr(S_EPN)}, 2")
            e(f"s_cbranch_scc1 {L2}")
            e(f"s_cmp_eq_u32 {sr(S_EPN)}, 1")
            e(f"s_cbranch_scc1 {L1}")
            e("s_waitcnt vmcnt(0)")
            e(f"s_branch {Ld}")
            e(f"{L3}:")
            e("s_waitcnt vmcnt(30)")
            e(f"s_branch {Ld}")
            e(f"{L2}:")
            e("s_waitcnt vmcnt(20)")
            e(f"s_branch {Ld}")
            e(f"{L1}:")
            e("s_waitcnt vmcnt(10)")
            e(f"{Ld}:")
        e("s_waitcnt lgkmcnt(0)")
        e("s_nop 15")
        e("s_nop 7")
        return self.lines


# ---- the plain epilogue of the 128 x 64 wave tile kernel (bias + rank-32 low-rank up + the single rounding to 16 bits + store) ----------------------------------
# One wave per SIMD runs compiler-scheduled epilogue code at half the issue rate of two, and with 128 accumulators + 81 operand registers live this clang parks
# the accumulators in AGPRs (1300 v_accvgpr moves and 644 B of scratch per tile in the first build).  So the epilogue is generated too: ~330 instructions per
# wave-tile, every register named.  Same operations in the same order as gemm_w4a4_kernel's C++ (bit-identical outputs): acc += bias (an MFMA: bias[n] in k-slot 0
# of the weight-side operand against 1.0), acc += lora_up x lora_act_in per 16 ranks in ascending order (activations rounded to 16 bits, times lora_scales[u] when
# that is not 1), convert (fp16: clamp to +-65504 first), one v_permlane32_swap per dword so that a lane holds 8 consecutive columns, 16-byte stores.
#   in:  v[0:127] acc; a[120:183] lora_act_in, a[184:199] lora_up, a200 bias (the loop call's "ep" loads)
#        v209 = (lr * ldo + 8 h) * 2 (store offset of the lane's row), v210 = lr, v211 = h
#        s94 flags (bit 0 low-rank, bit 1 bias), s95 / s96 lora_scales[0] / [1] (fp32 bits), s97 = M - (first row of the wave), s98 = 64 * ldo (bytes per 32 rows),
#        s[100:101] = &out[first row of the wave][first column of the wave]
#   clobbers: v128 .. v191, s99, s[84:85], vcc, scc
# None of these is an operand or a clobber of the main loop: a whole-tile segment runs loop + epilogue as ONE asm statement (the accumulators never become
# compiler values: with them as outputs of one statement and inputs of the next, this clang shuffled all 128 through AGPRs around the schedule code in between).
E_BW = [132, 140]      # bias operand (weight side) per ni: 4 registers
E_ONE = [136, 144]     # 1.0 operand per ni
E_LA = [148, 164]      # lora_act_in as 16-bit fragments per unit u: 4 mi x 4 registers
E_T = 180              # 8 temporaries
E_OUT = 128            # output staging: 16 registers (reused after the bias operands are dead)
E_OFF = [209, 188, 189, 190]
E_S_FLAGS, E_S_SC, E_S_ROWS, E_S_RTB, E_S_TMP, E_S_OUT, E_V_LR, E_V_H = 94, 95, 97, 98, 99, 100, 210, 211   # store offsets of the four row tiles


class GenEpi3:
    def __init__(self, dt, opts=""):
        self.dt = dt           # "bf16" | "fp16"
        self.opts = set(o for o in opts.split("+") if o)
        self.lines = []
        self.label = 0

    def stamp(self, i):
        """probe builds (option "stamp"): shader cycles (low 32 bits of s_memtime) at phase boundary i into lane i of v215 (an output of the statement)"""
        if "stamp" in self.opts:
            self.e(f"s_memtime {sr(70, 2)}")
            self.e("s_waitcnt lgkmcnt(0)")
            self.e(f"v_writelane_b32 {vr(215)}, {sr(70)}, {i}")

    def e(self, s):
        self.lines.append(s)

    def new_label(self, tag):
        self.label += 1
        return f".Lsvdq3e_{tag}{self.label}_%="

    def mfma(self, t, a, b):
        op = "v_mfma_f32_32x32x16_bf16" if self.dt == "bf16" else "v_mfma_f32_32x32x16_f16"
        return f"{op} {vr(ACC + 16 * t, 16)}, {a}, {b}, {vr(ACC + 16 * t, 16)}"

    def cvt(self, dst, a, b):
        op = "v_cvt_pk_bf16_f32" if self.dt == "bf16" else "v_cvt_pk_f16_f32"
        return f"{op} {vr(dst)}, {vr(a)}, {vr(b)}"

    def build(self):
        e = self.e
        e("; ---- svdq gemm plain epilogue, 128 x 64 wave tile (generated by tools/gen_gemm_loop3.py) ----")
        Lnob, Lnol = self.new_label("nobias"), self.new_label("nolora")
        self.stamp(1)
        # bias
        e(f"s_bitcmp1_b32 {sr(E_S_FLAGS)}, 1")
        e(f"s_cbranch_scc0 {Lnob}")
        e(f"v_accvgpr_read_b32 {vr(E_T)}, {ar(A_EP + 80)}")
        e(f"v_mov_b32 {vr(E_T + 1)}, {'0x3f80' if self.dt == 'bf16' else '0x3c00'}")
        e(f"v_mov_b32 {vr(E_T + 2)}, 0")
        for ni in range(2):
            for k in range(1, 4):
                e(f"v_mov_b32 {vr(E_BW[ni] + k)}, 0")
                e(f"v_mov_b32 {vr(E_ONE[ni] + k)}, 0")
        e(f"v_cmp_eq_u32 vcc, 0, {vr(E_V_H)}")
        e(f"v_cndmask_b32 {vr(E_BW[0])}, {vr(E_T + 2)}, {vr(E_T)}, vcc")      # h == 0 ? bias : 0
        e(f"v_cndmask_b32 {vr(E_ONE[0])}, {vr(E_T + 2)}, {vr(E_T + 1)}, vcc")
        e(f"v_cndmask_b32 {vr(E_BW[1])}, {vr(E_T)}, {vr(E_T + 2)}, vcc")      # h == 1 ? bias : 0
        e(f"v_cndmask_b32 {vr(E_ONE[1])}, {vr(E_T + 1)}, {vr(E_T + 2)}, vcc")
        e("s_nop 1")
        for ni in range(2):
            for mi in range(4):
                e(self.mfma(4 * ni + mi, vr(E_BW[ni], 4), vr(E_ONE[ni], 4)))
        e(f"{Lnob}:")
        # low-rank up projection, rank 32 = two units of 16
        e(f"s_bitcmp1_b32 {sr(E_S_FLAGS)}, 0")
        e(f"s_cbranch_scc0 {Lnol}")
        for u in range(2):
            Lsc, Ldone = self.new_label("scaled"), self.new_label("cvtdone")
            e(f"s_cmp_eq_u32 {sr(E_S_SC + u)}, 0x3f800000")
            e(f"s_cbranch_scc0 {Lsc}")
            body = {False: [], True: []}
            for scaled in (False, True):
                o = body[scaled]
                for mi in range(4):
                    for d in range(4):
                        t0, t1 = E_T + 2 * (d & 3), E_T + 2 * (d & 3) + 1
                        o.append(f"v_accvgpr_read_b32 {vr(t0)}, {ar(A_EP + 16 * mi + 8 * u + 2 * d)}")
                        o.append(f"v_accvgpr_read_b32 {vr(t1)}, {ar(A_EP + 16 * mi + 8 * u + 2 * d + 1)}")
                        if scaled:
                            o.append(f"v_mul_f32 {vr(t0)}, {sr(E_S_SC + u)}, {vr(t0)}")
                            o.append(f"v_mul_f32 {vr(t1)}, {sr(E_S_SC + u)}, {vr(t1)}")
                        o.append(self.cvt(E_LA[u] + 4 * mi + d, t0, t1))
            for ln in body[False]:
                e(ln)
            e(f"s_branch {Ldone}")
            e(f"{Lsc}:")
            for ln in body[True]:
                e(ln)
            e(f"{Ldone}:")
            e("s_nop 1")
            for ni in range(2):
                for mi in range(4):
                    e(self.mfma(4 * ni + mi, ar(A_EP + 64 + 4 * (2 * ni + u), 4), vr(E_LA[u] + 4 * mi, 4)))
        e(f"{Lnol}:")
        # the MFMA results must have left the pipe before the VALU reads them
        e("s_nop 15")
        e("s_nop 7")
        self.stamp(2)
        # nothing of this epilogue has gone to memory yet: whatever is outstanding is the next tile's prefetch -- waiting here proves it landed, so the next
        # loop call need not drain this epilogue's stores (the C++ side sets `landed`)
        e("s_waitcnt vmcnt(0)")
        self.stamp(3)
        st64 = "st64" in self.opts
        if st64:
            # option "st64": one more exchange (v_permlane16_swap between the register sets of columns [0, 16) and [16, 32) of a 32-column tile) so that a store
            # instruction covers 16 rows x 64 contiguous bytes instead of 32 rows x 32: half the write requests per tile
            L15, OA, OB = 188, [216, 217, 218, 219], [220, 221, 222, 223]
            e(f"s_lshr_b32 {sr(E_S_TMP)}, {sr(E_S_RTB)}, 5")                   # bytes per output row
            e(f"v_and_b32 {vr(L15)}, 15, {vr(E_V_LR)}")
            e(f"v_mul_u32_u24 {vr(OA[0])}, {sr(E_S_TMP)}, {vr(L15)}")
            e(f"v_lshrrev_b32 {vr(E_T)}, 4, {vr(E_V_LR)}")
            e(f"v_lshlrev_b32 {vr(E_T)}, 5, {vr(E_T)}")                        # lanes 16 .. 31 of a half: the second 32 bytes
            e(f"v_lshlrev_b32 {vr(E_T + 1)}, 4, {vr(E_V_H)}")                  # h: the second 16 bytes
            e(f"v_add3_u32 {vr(OA[0])}, {vr(OA[0])}, {vr(E_T)}, {vr(E_T + 1)}")
            e(f"s_lshl_b32 {sr(E_S_TMP)}, {sr(E_S_TMP)}, 4")                   # 16 rows
            e(f"v_add_u32 {vr(OB[0])}, {sr(E_S_TMP)}, {vr(OA[0])}")
            for mi in range(1, 4):
                e(f"s_mul_i32 {sr(E_S_TMP)}, {sr(E_S_RTB)}, {mi}")
                e(f"v_add_u32 {vr(OA[mi])}, {sr(E_S_TMP)}, {vr(OA[0])}")
                e(f"v_add_u32 {vr(OB[mi])}, {sr(E_S_TMP)}, {vr(OB[0])}")
        else:
            for mi in range(1, 4):
                e(f"s_mul_i32 {sr(E_S_TMP)}, {sr(E_S_RTB)}, {mi}")
                e(f"v_add_u32 {vr(E_OFF[mi])}, {sr(E_S_TMP)}, {vr(E_OFF[0])}")
        if self.dt == "fp16":
            e(f"s_mov_b32 {sr(E_S_TMP)}, 0x477fe000")     # 65504.0
        for mi in range(4):
            k = 0
            for ni in range(2):
                t = 4 * ni + mi
                for j in range(2):
                    for half in range(2):          # x (columns 8 c + 4 h + e of c = 2 j), y (c = 2 j + 1)
                        for d in range(2):
                            a0 = ACC + 16 * t + (2 * j + half) * 4 + 2 * d
                            if self.dt == "fp16":
                                e(f"v_med3_f32 {vr(E_T)}, {vr(a0)}, -{sr(E_S_TMP)}, {sr(E_S_TMP)}")
                                e(f"v_med3_f32 {vr(E_T + 1)}, {vr(a0 + 1)}, -{sr(E_S_TMP)}, {sr(E_S_TMP)}")
                                e(self.cvt(E_OUT + 4 * k + 2 * half + d, E_T, E_T + 1))
                            else:
                                e(self.cvt(E_OUT + 4 * k + 2 * half + d, a0, a0 + 1))
                    k += 1
            e("s_nop 1")
            for k in range(4):
                for d in range(2):
                    e(f"v_permlane32_swap_b32 {vr(E_OUT + 4 * k + d)}, {vr(E_OUT + 4 * k + 2 + d)}")
            e("s_nop 1")
            if st64:
                for ni in range(2):
                    for d in range(4):
                        e(f"v_permlane16_swap_b32 {vr(E_OUT + 8 * ni + d)}, {vr(E_OUT + 8 * ni + 4 + d)}")
                e("s_nop 1")
                for half, off in ((0, OA), (1, OB)):   # rows 32 mi + 16 half + (lr & 15)
                    e(f"s_sub_i32 {sr(84)}, {sr(E_S_ROWS)}, {32 * mi + 16 * half}")
                    e(f"v_cmp_gt_i32 vcc, {sr(84)}, {vr(L15)}")
                    e(f"s_and_saveexec_b64 {sr(84, 2)}, vcc")
                    for ni in range(2):
                        if "nost" not in self.opts:
                            e(f"global_store_dwordx4 {vr(off[mi])}, {vr(E_OUT + 8 * ni + 4 * half, 4)}, {sr(E_S_OUT, 2)}" + (f" offset:{64 * ni}" if ni else ""))
                    e(f"s_mov_b64 exec, {sr(84, 2)}")
                continue
            # rows at or beyond M are not stored
            e(f"s_sub_i32 {sr(84)}, {sr(E_S_ROWS)}, {32 * mi}")
            e(f"v_cmp_gt_i32 vcc, {sr(84)}, {vr(E_V_LR)}")
            e(f"s_and_saveexec_b64 {sr(84, 2)}, vcc")
            k = 0
            for ni in range(2):
                for j in range(2):
                    imm = 64 * ni + 32 * j
                    if "nost" not in self.opts:
                        e(f"global_store_dwordx4 {vr(E_OFF[mi])}, {vr(E_OUT + 4 * k, 4)}, {sr(E_S_OUT, 2)}" + (f" offset:{imm}" if imm else ""))
                    k += 1
            e(f"s_mov_b64 exec, {sr(84, 2)}")
        self.stamp(4)
        return self.lines


def emit_epilogue(path, dt, opts=""):
    lines = GenEpi3(dt, opts).build()
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_gemm_loop3.py (plain epilogue of the 128 x 64 wave tile kernel) -- do not edit.\n")
        for ln in lines:
            f.write('"' + ln + '\\n"\n')
    return len(lines)


def emit(path, smfma, opts=""):
    lines = Gen3(smfma, opts).build()
    with open(path, "w") as f:
        f.write(f"// GENERATED by tools/gen_gemm_loop3.py (128 x 64 wave tile, one wave per SIMD; options {opts!r}) -- do not edit.\n")
        for ln in lines:
            f.write('"' + ln + '\\n"\n')
    return len(lines)


PRODUCT_OPTS = "b2+ep"   # the loop the library ships: ring of four, barrier every other K-step, DMAs spread over the tile-group slots


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    opts = os.environ.get("SVDQ_GEN3_OPTS")
    if opts is None:
        # the product loops: nunchaku_amd/csrc/gemm_loop3_{bf16,fp16}.inc (included by gemm_w4a4.hip: gemm_w4a4_wt128_kernel)
        root = os.path.join(os.path.dirname(here), "nunchaku_amd", "csrc")
        n = emit(os.path.join(root, "gemm_loop3_bf16.inc"), "v_mfma_f32_32x32x16_bf16", PRODUCT_OPTS)
        emit(os.path.join(root, "gemm_loop3_fp16.inc"), "v_mfma_f32_32x32x16_f16", PRODUCT_OPTS)
        print(f"wrote nunchaku_amd/csrc/gemm_loop3_{{bf16,fp16}}.inc ({n} lines each, options {PRODUCT_OPTS!r})")
        n = emit_epilogue(os.path.join(root, "gemm_epi3_bf16.inc"), "bf16")
        emit_epilogue(os.path.join(root, "gemm_epi3_fp16.inc"), "fp16")
        print(f"wrote nunchaku_amd/csrc/gemm_epi3_{{bf16,fp16}}.inc ({n} lines)")
    else:
        # probe variants (tools/ablate/build128.sh): tools/ablate/gen/gemm_loop3_{bf16,fp16}[_<opts>].inc
        root = os.path.join(here, "ablate", "gen")
        os.makedirs(root, exist_ok=True)
        sfx = ("_" + opts.replace("+", "_")) if opts else ""
        n = emit(os.path.join(root, f"gemm_loop3_bf16{sfx}.inc"), "v_mfma_f32_32x32x16_bf16", opts)
        emit(os.path.join(root, f"gemm_loop3_fp16{sfx}.inc"), "v_mfma_f32_32x32x16_f16", opts)
        emit_epilogue(os.path.join(root, f"gemm_epi3_bf16{sfx}.inc"), "bf16", opts)
        emit_epilogue(os.path.join(root, f"gemm_epi3_fp16{sfx}.inc"), "fp16", opts)
        print(f"wrote tools/ablate/gen/gemm_{{loop3,epi3}}_{{bf16,fp16}}{sfx}.inc ({n} lines each)")
