#!/usr/bin/env python3
"""Generate main loop v2 of svdq_gemm_w4a4 (gfx950): same arithmetic and register image as tools/gen_gemm_loop.py
(bit-identical results), ~40 fewer instructions per K-step.

    python tools/gen_gemm_loop2.py        # writes nunchaku_amd/csrc/gemm_loop2_{bf16,fp16}.inc

Why (profiles/r2_gemm_ablation_cycles.txt): the loop is bound by instruction ISSUE, not by any pipe -- a SIMD with its
two waves retires ~1 instruction per 4.5 cycles whatever the mix, MFMA and VALU barely overlap (P+S 64 cycles, P+S+16 fma
86 per 32x32x64 tile-group, one wave or two), and every DMA / LDS / scalar instruction of the memory side adds its ~4.5
cycles on top (compute alone 86, whole loop 129 cycles per tile-group).  v1 spends ~230 instructions per wave and K-step,
144 of them arithmetic.  v2 removes bookkeeping, not work:
  * the LDS ring (4 stages then, 3 now) is unrolled: one copy of the K-step body per stage, every LDS address is base + immediate
    (no per-step v_mov / v_add / s_cselect stage rotation);
  * operand streams are MUBUF LDS-DMA (buffer_load_dwordx4 ... lds) with the K-step offset in ONE shared soffset SGPR per
    stream class: 2 scalar adds per K-step instead of 6 add/addc, segment switch = 3 s_mov_b64 out of line;
  * every wave issues exactly 5 DMA instructions per K-step (waves 6, 7 repeat their W plane), so the wait for "my DMAs of
    K-step s+1 have landed" is the constant s_waitcnt vmcnt(5 * (stages - 2)) -- no counting ladder, no taken branch on the hot path;
    the DMA-less last K-steps of a workgroup's last segment take an out-of-line path with vmcnt(0);
  * loop control: 3 + 4 scalar instructions per K-step, all branches not taken in steady state.

Register plan (the C++ side pins its asm operands to the same registers, gemm_w4a4.hip):
  v[0:63] acc   v[64:79] P0  v[80:95] S0  v[96:111] P1  v[112:127] S1   v[128:151] / v[152:175] fragment buffers 0 / 1
  v[176:183] / v[184:191] scale tuples 0 / 1 (round 6: two registers each -- the K = 8 scale-tile MFMA; v192 .. v209 are free for the caller since the
  product MFMA lost its MX exponents: the GELU_QUANT carry kernel keeps a row block's low-rank activations there across a run of column tiles)
  v210..v213 in: per-lane LDS offsets of A frags, W frags, as, ws (stage-relative)   v214..v216 in: per-lane DMA offsets
  v217..v220: the same four LDS offsets + 2*STAGE (stages 2, 3: ds offsets are 16-bit)
  s46 K-steps of this segment   s47 s48 s49 LDS destinations of the A / X1 / X2 planes (stage-relative)
  s50 per-K-step soffset increment of stream X2 (3072 for a W plane, 128 for a scale plane)
  s[72:75] s[76:79] s[80:83] buffer resources of streams A, X1, X2 (in: this segment's bases; the loop moves them on)
  s[62:63] s[64:65] s[66:67] bases of the NEXT segment's streams   s60 its K-step count (0 = none)
  s58 in: LDS stage offset of this segment's K-step 0 (the caller advances it)   s59 this segment's K-steps already in flight
  s68 leading K-steps known to have landed (0 or s59)   s53 s55 s56 s57 s61 s84 s85 s86 s87 scratch
"""
import os

CHUNK, PLANE = 3072, 1024


class Geometry:
    """Workgroup geometry of the loop.  nw = 8: 512 threads, tile 256 x 128, one workgroup per CU, 3-stage ring (+ the staged epilogue operands), 5 DMA
    instructions per wave and K-step (three A planes, one W plane, one W plane / scale image / repeat).  nw = 4: 256 threads,
    tile 128 x 128, TWO workgroups per CU (80 KiB of LDS each), 3-stage ring, 7 DMA instructions per wave and K-step (the
    three planes of A chunk `wave`, the three planes of W chunk `wave`, one scale image / repeat).  The wave tile (64 x 64),
    the register plan and the arithmetic are the same."""

    def __init__(self, nw, nstage=None):
        assert nw in (4, 8)
        self.nw = nw
        self.a_bytes = nw * CHUNK
        self.stage = self.a_bytes + 4 * CHUNK + 1024 + 1024
        # nw = 8: a ring of three since the end of round 3 (measured equal to four, bit-identical: profiles/r3_gemm_ring3.txt); the
        # freed stage holds the tile's epilogue operands (lora_act_in 32 KB, lora_up 8 KB, bias 256 B), staged by the loop's prologue
        self.nstage = nstage or 3
        self.staging = nw == 8
        self.x1_planes = 1 if nw == 8 else 3
        self.dma_per_step = 3 + self.x1_planes + 1

ACC = 0
PBUF = [64, 96]
SBUF = [80, 112]
FRAG = [128, 152]
SCL = [176, 184]
TS = 2                # registers per scale tuple (k-slots 0 .. 3: the K = 8 form)
MXA, MXB = 208, 209
IN_L = [210, 211, 212, 213]   # A, W, SA, SW (stages 0, 1)
HI_L = [217, 218, 219, 220]   # + 2 * STAGE (stages 2, 3)
OFF_A, OFF_X1, OFF_X2 = 214, 215, 216
S_KP, S_DA, S_DX1, S_DX2, S_IX2 = 46, 47, 48, 49, 50
S_STEP, S_TOT4, S_KP4, S_TMP, S_TOT = 53, 55, 56, 57, 61
S_RING, S_NPRE, S_NCNT, S_LANDED = 58, 59, 60, 68
S_NA, S_NX1, S_NX2 = 62, 64, 66
SRD_A, SRD_X1, SRD_X2 = 72, 76, 80
S_KOFF, S_KOFF2, S_DSTEP, S_STG = 84, 85, 86, 87
S_PHASE = 52  # in ("dph" builds): 1 for waves 4..7
S_NODMA = 69  # option "sp": this K-step issues no DMA
# staging of the epilogue operands (nw = 8): in: s51 flags (bit 0: low-rank operands, bit 1: bias, bit 2: of the low-rank operands only lora_up; bits 8..10: wave), s54 LDS address of the
# staging region, s[88:89] this wave's 4 KiB of lora_act_in, s[90:91] its 1 KiB of lora_up, s[92:93] the tile's 256 bytes of bias
S_STGF, S_STGB, S_SLA, S_SLU, S_SB = 51, 54, 88, 90, 92
STG_LU, STG_BIAS = 32768, 43008   # offsets inside the staging region (lora_act_in at 0; rank > 32: the tile's lora_up for every rank at 0, gemm_w4a4.hip Geo)


def vr(a, n=1):
    return f"v{a}" if n == 1 else f"v[{a}:{a + n - 1}]"


def sr(a, n=1):
    return f"s{a}" if n == 1 else f"s[{a}:{a + n - 1}]"


class Gen:
    def __init__(self, smfma, opts="", nw=8):
        self.smfma = smfma
        self.opts = set(o for o in opts.split("+") if o)
        self.geo = Geometry(nw, 4 if "st4" in self.opts else None)  # "st4": the former ring of four (nw = 8; timing only: the kernel's LDS map assumes three)
        self.lines = []
        self.label = 0

    def e(self, s):
        self.lines.append(s)

    def new_label(self, tag=""):
        self.label += 1
        return f".Lsvdq2_{tag}{self.label}_%="

    # ---- addressing -------------------------------------------------------------------
    def lds(self, kind, stage):
        """(address VGPR, immediate) of operand kind 0..3 in ring stage `stage`"""
        if stage < 2:
            return IN_L[kind], stage * self.geo.stage
        return HI_L[kind], (stage - 2) * self.geo.stage

    def frag_reads(self, buf, grp, stage):
        """the 12 LDS reads of one group's fragments and scales.  Timing-only options (garbage results): "rd20u" drops the
        group-1 halves of plane 1 (what a merged 16-byte read of plane 1 would save: 4 of 24 reads per K-step), "rd18u" also one
        weight-scale read per group (what packed weight-scale pairs would save), "rd0" every read but one."""
        out = self._frag_reads(buf, grp, stage)
        if "rd0" in self.opts:
            return out[:1]
        if "rd20u" in self.opts or "rd18u" in self.opts:
            out = [ln for ln in out if not (grp == 1 and ln.startswith("ds_read_b64"))]
        if "rd18u" in self.opts:
            k = [i for i, ln in enumerate(out) if ln.startswith("ds_read_u16")][1]
            out = out[:k] + out[k + 1:]
        return out

    def _frag_reads(self, buf, grp, stage):
        out = []
        for kind, roff in ((1, 0), (0, 12)):  # W fragments first (v1's order), then A
            base, imm = self.lds(kind, stage)
            for i in range(2):
                f = FRAG[buf] + roff + 6 * i
                o = imm + i * CHUNK
                if grp == 0:
                    out.append(f"ds_read_b128 {vr(f, 4)}, {vr(base)} offset:{o}")
                    out.append(f"ds_read_b64 {vr(f + 4, 2)}, {vr(base)} offset:{o + PLANE}")
                else:
                    out.append(f"ds_read_b64 {vr(f, 2)}, {vr(base)} offset:{o + PLANE + 8}")
                    out.append(f"ds_read_b128 {vr(f + 2, 4)}, {vr(base)} offset:{o + 2 * PLANE}")
        for kind, roff in ((3, 0), (2, 8)):
            base, imm = self.lds(kind, stage)
            for i in range(2):
                out.append(f"ds_read_u16 {vr(SCL[buf] + (roff // 4 + i) * TS)}, {vr(base)} offset:{imm + i * 128 + grp * 64}")
        return out

    def p_mfma(self, dst_buf, buf, t):
        ni, mi = t >> 1, t & 1
        w = FRAG[buf] + 6 * ni
        a = FRAG[buf] + 12 + 6 * mi
        if "mx" not in self.opts:
            # round 6: no MX block scales -- ONE instruction instead of the v_mfma_ld_scale + MFMA pair, two VGPR reads fewer; P = dot / 64 and the weight-side scale
            # image carries the factor 32 (svdq_repack_wscales): the same fp32 product P S bit for bit (tools/gen_gemm_loop3.py: p_mfma)
            return f"v_mfma_f32_32x32x64_f8f6f4 {vr(PBUF[dst_buf], 16)}, {vr(w, 6)}, {vr(a, 6)}, 0 cbsz:2 blgp:2"
        return (f"v_mfma_scale_f32_32x32x64_f8f6f4 {vr(PBUF[dst_buf], 16)}, {vr(w, 6)}, {vr(a, 6)}, 0, "
                f"{vr(MXA)}, {vr(MXB)} op_sel_hi:[0,0,0] cbsz:2 blgp:2")

    def s_mfma(self, dst_buf, buf, t):
        ni, mi = t >> 1, t & 1
        # round 6: the scale tile S = 2 ws as needs two non-zero k-slots; the legacy K = 8 form of the 16-bit MFMA (k-slots 0 and 4: the two registers of a tuple)
        # computes the same bits in the same 8 passes (tools/ablate/mfma_k8_probe: 32.6 cycles either way) and reads half the operand registers:
        # -3.5 % loop cycles, -2.5 ... -3.2 % launch time on the wave-tile probe, bit-exact (profiles/r6_gemm_wave_tile_probe.txt section 5)
        op = "v_mfma_f32_32x32x8bf16_1k" if self.smfma.endswith("bf16") else "v_mfma_f32_32x32x8f16"
        return f"{op} {vr(SBUF[dst_buf], 16)}, {vr(SCL[buf] + ni * TS, 2)}, {vr(SCL[buf] + (2 + mi) * TS, 2)}, 0"

    def fma(self, t, pb, lo, hi):
        """acc += P * S for registers lo .. hi-1 of tile t.  "pkf": packed fp32 FMAs (v_pk_fma_f32: two accumulators per
        instruction at the same issue cost; same fp32 fused multiply-add per element, bit-identical) -- half the VALU
        instructions of the loop, which is bound by instruction issue."""
        if "pkf" in self.opts:
            return [f"v_pk_fma_f32 {vr(ACC + 16 * t + r, 2)}, {vr(PBUF[pb] + r, 2)}, {vr(SBUF[pb] + r, 2)}, {vr(ACC + 16 * t + r, 2)}"
                    for r in range(lo, hi, 2)]
        return [f"v_fmac_f32 {vr(ACC + 16 * t + r)}, {vr(PBUF[pb] + r)}, {vr(SBUF[pb] + r)}" for r in range(lo, hi)]

    # ---- DMA --------------------------------------------------------------------------
    def dma_issue(self, stage_imm=None):
        """the 5 LDS-DMA wave loads of one K-step at the DMA cursor, into ring stage `stage_imm` (byte offset; None:
        the stage offset is in S_STG), then advance the cursor by one K-step.  Scalar bookkeeping sits in the M0 -> DMA
        wait-state slots."""
        def m0(dst):
            return f"s_add_u32 m0, {sr(dst)}, {stage_imm}" if stage_imm is not None else f"s_add_u32 m0, {sr(dst)}, {sr(S_STG)}"
        ld = "buffer_load_dwordx4"
        # cache-policy options (same results, timing / traffic experiments: profiles/r4_gemm_nt_streams.txt): "nta" / "ntw" mark the activation
        # stream / the weight-side streams non-temporal (streamed once through the L2: the other operand's panels are what a later tile re-reads)
        nta = " nt" if "nta" in self.opts else ""
        ntw = " nt" if "ntw" in self.opts else ""
        return [
            m0(S_DA),
            f"s_add_u32 {sr(S_DSTEP)}, {sr(S_DSTEP)}, 1",
            f"{ld} {vr(OFF_A)}, {sr(SRD_A, 4)}, {sr(S_KOFF)} offen{nta} lds",
            f"{ld} {vr(OFF_A)}, {sr(SRD_A, 4)}, {sr(S_KOFF)} offen offset:{PLANE}{nta} lds",
            f"{ld} {vr(OFF_A)}, {sr(SRD_A, 4)}, {sr(S_KOFF)} offen offset:{2 * PLANE}{nta} lds",
            m0(S_DX1),
            "s_nop 0",
        ] + [f"{ld} {vr(OFF_X1)}, {sr(SRD_X1, 4)}, {sr(S_KOFF)} offen" + (f" offset:{pl * PLANE}" if pl else "") + f"{ntw} lds"
             for pl in range(self.geo.x1_planes)] + [
            m0(S_DX2),
            f"s_add_u32 {sr(S_KOFF)}, {sr(S_KOFF)}, {CHUNK}",
            f"{ld} {vr(OFF_X2)}, {sr(SRD_X2, 4)}, {sr(S_KOFF2)} offen{ntw} lds",
            f"s_add_u32 {sr(S_KOFF2)}, {sr(S_KOFF2)}, {sr(S_IX2)}",
        ]

    def stage_epilogue_operands(self):
        """nw = 8, behind the entry barrier (every wave has left the previous tile's epilogue): this wave's share of the tile's
        epilogue operands goes to the staging region by LDS-DMA and lands under the main loop -- 4 pieces of 1 KiB of lora_act_in
        (rows 32 w .. 32 w + 31 of the tile, fp32 [row][32 ranks] = 128 B per row; lane i of a piece fetches 16-byte chunk
        (i & 7) ^ ((i >> 3) & 7) of row i >> 3, i.e. the LDS image holds chunk c of row r at position c ^ (r & 7): the epilogue's
        16-byte reads of one chunk over 32 rows then spread over the banks), 1 piece of lora_up (rows 16 w .. 16 w + 15, linear)
        and, wave 0, the bias (4 bytes per lane).  The DMAs are older than every DMA of the loop: its vmcnt waits only get stricter."""
        if not self.geo.staging:
            return []
        Lno, Lnob = self.new_label("nostg"), self.new_label("nostgb")
        t0, t1, t2 = 64, 65, 66   # P buffer registers: free until the first MFMA
        out = [
            f"s_bitcmp1_b32 {sr(S_STGF)}, 0",
            f"s_cbranch_scc0 {Lno}",
            f"s_bfe_u32 {sr(S_TMP)}, {sr(S_STGF)}, 0x30008",                 # wave
            f"v_lshrrev_b32 {vr(t0)}, 7, {vr(OFF_A)}",                        # row in piece = lane >> 3   (OFF_A = 16 * lane)
            f"v_bfe_u32 {vr(t1)}, {vr(OFF_A)}, 4, 3",                         # chunk position = lane & 7
            f"v_and_b32 {vr(t2)}, 7, {vr(t0)}",
            f"v_xor_b32 {vr(t1)}, {vr(t1)}, {vr(t2)}",                        # chunk fetched
            f"v_lshlrev_b32 {vr(t1)}, 4, {vr(t1)}",
            f"v_lshl_or_b32 {vr(t0)}, {vr(t0)}, 7, {vr(t1)}",                 # byte offset inside the 1 KiB piece
            f"s_lshl_b32 {sr(S_TMP)}, {sr(S_TMP)}, 10",                       # wave * 1024
            f"s_add_u32 m0, {sr(S_STGB)}, {sr(S_TMP)}",
            f"s_add_u32 m0, m0, {STG_LU}",
            "s_nop 0",
            f"global_load_lds_dwordx4 {vr(OFF_A)}, {sr(S_SLU, 2)}",
            f"s_bitcmp1_b32 {sr(S_STGF)}, 2",                                 # bit 2: lora_up only (the lora_act_in slot of the region holds the
            f"s_cbranch_scc1 {Lno}",                                          #        workgroup's low-rank-down carry: gemm_w4a4.hip "row runs")
            f"s_lshl_b32 {sr(S_TMP)}, {sr(S_TMP)}, 2",                        # wave * 4096
            f"s_add_u32 m0, {sr(S_STGB)}, {sr(S_TMP)}",
        ]
        # (no instruction offsets: an LDS-DMA's immediate offset moves the memory AND the LDS address; one address register per piece)
        out += [f"v_add_u32 {vr(t0 + 2 + i)}, {0x400 * i}, {vr(t0)}" for i in range(1, 4)]
        for i in range(4):
            out += ["s_nop 0", f"global_load_lds_dwordx4 {vr(t0 + 2 + i if i else t0)}, {sr(S_SLA, 2)}"]
            if i < 3:
                out.append("s_add_u32 m0, m0, 1024")
        out += [
            f"{Lno}:",
            f"s_bitcmp1_b32 {sr(S_STGF)}, 1",
            f"s_cbranch_scc0 {Lnob}",
            f"s_bfe_u32 {sr(S_TMP)}, {sr(S_STGF)}, 0x30008",
            f"s_cmp_eq_u32 {sr(S_TMP)}, 0",
            f"s_cbranch_scc0 {Lnob}",
            f"v_lshrrev_b32 {vr(t0 + 6)}, 2, {vr(OFF_A)}",                    # 4 * lane (a register no DMA above has used as its address)
            f"s_add_u32 m0, {sr(S_STGB)}, {STG_BIAS}",
            "s_nop 0",
            f"global_load_lds_dword {vr(t0 + 6)}, {sr(S_SB, 2)}",
            f"{Lnob}:",
        ]
        return out

    def switch_segment(self):
        """DMA cursor moves to the first K-step of the next segment"""
        return [
            f"s_mov_b64 {sr(SRD_A, 2)}, {sr(S_NA, 2)}",
            f"s_mov_b64 {sr(SRD_X1, 2)}, {sr(S_NX1, 2)}",
            f"s_mov_b64 {sr(SRD_X2, 2)}, {sr(S_NX2, 2)}",
            f"s_mov_b32 {sr(S_KOFF)}, 0",
            f"s_mov_b32 {sr(S_KOFF2)}, 0",
        ]

    # ---- one K-step body for ring stage j --------------------------------------------------
    def kstep(self, j, exit_label, ool, phase=0):
        """K-step body for ring stage j.  phase 0: the wave issues its DMAs right behind the K-step barrier (tile-group 4).
        phase 1 ("dph" option, waves 4..7): it issues them four tile-groups later, at the top of the NEXT K-step body, so
        that the two waves of a SIMD are never both busy with their DMA bursts; returns the label to enter this body at
        (phase 1: behind the deferred-issue block, which the first K-step of a segment must skip)."""
        e = self.e
        nj = (j + 1) % self.geo.nstage
        pj = (j - 1) % self.geo.nstage
        reads_g1 = self.frag_reads(1, 1, j)
        reads_n0 = self.frag_reads(0, 0, nj)
        if phase == 1:
            # DMA of K-step step+3 (deferred from the previous body) into the stage K-step step-1 just vacated
            Lskip, Lsw, Lbsw = (self.new_label(x) for x in ("dskip", "dsw", "dbsw"))
            e(f"s_cmp_lt_u32 {sr(S_STEP)}, {sr(S_TOT4)}")
            e(f"s_cbranch_scc0 {Lskip}")
            e(f"s_cmp_eq_u32 {sr(S_STEP)}, {sr(S_KP4)}")
            e(f"s_cbranch_scc1 {Lsw}")
            e(f"{Lbsw}:")
            for ln in self.dma_issue(pj * self.geo.stage):
                e(ln)
            e(f"{Lskip}:")
            ool += [f"{Lsw}:"] + self.switch_segment() + [f"s_branch {Lbsw}"]
        Lin = self.new_label(f"in{phase}st{j}_")
        e(f"{Lin}:")
        for q in range(8):
            t = q & 3
            pb = q & 1
            qn = q + 1
            nbuf, nt = (qn >> 2) & 1, qn & 3
            # LDS returns in order.  The 12 fragment / scale reads of a group are W0 W0 W1 W1 | A0 A0 A1 A1 | sw0 sw1 sa0 sa1;
            # each MFMA waits only for the reads it consumes (lgkmcnt = reads allowed to remain outstanding):
            #   P(tile 0) needs reads 1..6, S(tile 0) 1..11, P(tile 1) 1..8, S(tile 1) all 12 (+4 newer ones issued since)
            fine = "lgkf" in self.opts  # opt-in: measured 1.4 % SLOWER than the two plain lgkmcnt(0) waits (same box A/B)
            if q in (3, 7):
                e("s_waitcnt lgkmcnt(6)" if fine else "s_waitcnt lgkmcnt(0)")
            elif q in (0, 4) and fine:
                e("s_waitcnt lgkmcnt(4)")
            e(self.p_mfma(pb ^ 1, nbuf, nt))
            misc = []
            if q < 3:
                misc = reads_g1[4 * q:4 * q + 4]
            elif q == 4:
                Ltail, Lback, Lsw, Lbsw = (self.new_label(x) for x in ("tail", "back", "sw", "bsw"))
                misc = [
                    # in-line path iff the wave's youngest 10 DMAs are exactly those of K-steps step+2, step+3
                    # (phase 0: iff it issues K-step step+4 now; phase 1: iff K-step step+3 has been issued)
                    f"s_cmp_lt_u32 {sr(S_STEP)}, {sr(S_TOT4)}",
                    f"s_cbranch_scc0 {Ltail}",
                    # my 5 DMAs of K-step step+1 have landed: the 10 younger ones (step+2, step+3) may stay in flight
                    f"s_waitcnt vmcnt({(self.geo.nstage - 2) * self.geo.dma_per_step})",
                    # "nobar": timing ablation only (waves race: wrong results); "bar1of3" / "bar2of3" (round 6, timing only as well): the K-step barrier in one /
                    # two of the ring's three bodies -- what a rendezvous every 2-3 K-steps (a deeper ring, a K-step of 256 channels) could save at most
                    "s_nop 0" if ("nobar" in self.opts or ("bar1of3" in self.opts and j != 0) or ("bar2of3" in self.opts and j == 2)) else "s_barrier",
                ]
                spread = {}
                if phase == 0 and "burst" not in self.opts:
                    # round 6: the K-step's DMAs one unit per tile-group slot behind the barrier instead of one burst (what the wave-tile loop needs; here the
                    # second wave of the SIMD fills most of an acceptance stall): -4 % workgroup cycles, -1 ... -2 % launch time at the power limit, identical
                    # outputs (tools/gpu/r6_sp2.sh, profiles/r6_gemm_wave_tile_probe.txt section 9).  "burst": the round 2-5 form
                    d = self.dma_issue(j * self.geo.stage)
                    ia = [i for i, ln in enumerate(d) if ln.startswith("s_add_u32 m0")]
                    units = [d[ia[0]:ia[0] + 3], d[ia[0] + 3:ia[0] + 4], d[ia[0] + 4:ia[1]], d[ia[1]:ia[2]], d[ia[2]:]]
                    misc = [f"s_mov_b32 {sr(S_NODMA)}, 0"] + misc
                    misc += [f"s_cmp_eq_u32 {sr(S_STEP)}, {sr(S_KP4)}", f"s_cbranch_scc1 {Lsw}", f"{Lbsw}:"] + units[0] + units[1]
                    spread = {5: units[2], 6: units[3], 7: units[4]}
                    ool += [f"{Lsw}:"] + self.switch_segment() + [f"s_branch {Lbsw}"]
                    self._spread = spread
                elif phase == 0:
                    misc += [f"s_cmp_eq_u32 {sr(S_STEP)}, {sr(S_KP4)}", f"s_cbranch_scc1 {Lsw}", f"{Lbsw}:"] + self.dma_issue(j * self.geo.stage)
                    ool += [f"{Lsw}:"] + self.switch_segment() + [f"s_branch {Lbsw}"]
                misc += [f"{Lback}:"] + reads_n0[0:4]
                ool += [f"{Ltail}:"] + ([f"s_mov_b32 {sr(S_NODMA)}, 1"] if "burst" not in self.opts else []) + ["s_waitcnt vmcnt(0)", "s_barrier", f"s_branch {Lback}"]
            elif q in (5, 6):
                misc = reads_n0[4 * (q - 4):4 * (q - 4) + 4]
            elif q == 7:
                misc = [f"s_add_u32 {sr(S_STEP)}, {sr(S_STEP)}, 1"]
            if phase == 0 and "burst" not in self.opts and q in (5, 6, 7):
                Lnd = self.new_label("nd")
                misc = list(misc) + [f"s_cmp_eq_u32 {sr(S_NODMA)}, 0", f"s_cbranch_scc0 {Lnd}"] + self._spread[q] + [f"{Lnd}:"]
            for ln in misc:
                e(ln)
            for ln in self.fma(t, pb, 0, 8):
                e(ln)
            if fine and q in (3, 7):
                e("s_waitcnt lgkmcnt(1)")
            elif fine and q in (0, 4):
                e("s_waitcnt lgkmcnt(4)")
            e(self.s_mfma(pb ^ 1, nbuf, nt))
            for ln in self.fma(t, pb, 8, 16):
                e(ln)
        e(f"s_cmp_lt_u32 {sr(S_STEP)}, {sr(S_KP)}")
        if phase == 0:
            e(f"s_cbranch_scc0 {exit_label}")
        else:
            # leaving the segment: the deferred DMA (K-step step+3, the next segment's) is issued here, into this stage
            Lstub = self.new_label("xstub")
            e(f"s_cbranch_scc0 {Lstub}")
            ool += [f"{Lstub}:", f"s_cmp_lt_u32 {sr(S_STEP)}, {sr(S_TOT4)}", f"s_cbranch_scc0 {exit_label}"] + \
                self.dma_issue(j * self.geo.stage) + [f"s_branch {exit_label}"]
        return Lin

    # ---- the whole asm block ---------------------------------------------------------------
    def build(self):
        e = self.e
        e("; ---- svdq gemm main loop v2 (generated by tools/gen_gemm_loop2.py) ----")
        # "prio1": static priority 1 for waves 4..7 (the younger wave of every SIMD, the arbitration loser) for the whole loop;
        # "prio0": the same for waves 0..3 (control experiment)
        if "prio1" in self.opts or "prio0" in self.opts:
            Lp = self.new_label("prio")
            e(f"s_cmp_eq_u32 {sr(S_PHASE)}, {1 if 'prio1' in self.opts else 0}")
            e(f"s_cbranch_scc0 {Lp}")
            e("s_setprio 1")
            e(f"{Lp}:")
        for r in range(64):
            e(f"v_mov_b32 {vr(ACC + r)}, 0")
        for b in range(2):
            for tpl in range(4):
                for k in range(1, TS):
                    e(f"v_mov_b32 {vr(SCL[b] + TS * tpl + k)}, 0")
        for k in range(4):
            e(f"v_add_u32 {vr(HI_L[k])}, {2 * self.geo.stage}, {vr(IN_L[k])}")
        e(f"s_mov_b32 {sr(S_STEP)}, 0")
        e(f"s_add_u32 {sr(S_TOT)}, {sr(S_KP)}, {sr(S_NCNT)}")
        # thresholds on `step`: DMA of K-step step+4 exists iff step < tot-4; it is the next segment's first iff step == kp-4
        if "dph" in self.opts:  # phase-1 waves run their DMA stream one K-step later: thresholds tot-3 / kp-3
            e(f"s_sub_u32 {sr(S_TMP)}, {self.geo.nstage}, {sr(S_PHASE)}")
        else:
            e(f"s_mov_b32 {sr(S_TMP)}, {self.geo.nstage}")
        e(f"s_sub_u32 {sr(S_TOT4)}, {sr(S_TOT)}, {sr(S_TMP)}")
        e(f"s_cselect_b32 {sr(S_TOT4)}, 0, {sr(S_TOT4)}")          # borrow (tot < 4): never
        e(f"s_sub_u32 {sr(S_KP4)}, {sr(S_KP)}, {sr(S_TMP)}")
        e(f"s_cselect_b32 {sr(S_KP4)}, -1, {sr(S_KP4)}")            # kp < 4: the prologue below has switched already
        # DMA cursor: K-step `npre` of this segment
        e(f"s_mov_b32 {sr(S_DSTEP)}, {sr(S_NPRE)}")
        e(f"s_mul_i32 {sr(S_KOFF)}, {sr(S_NPRE)}, {CHUNK}")
        e(f"s_mul_i32 {sr(S_KOFF2)}, {sr(S_NPRE)}, {sr(S_IX2)}")
        # prologue: top the stream up to self.geo.nstage K-steps in flight (all of them for a workgroup's first segment, normally
        # none afterwards: the previous segment has fetched them).  Generic, branchy, rare.
        e(f"s_mul_i32 {sr(S_TMP)}, {sr(S_NPRE)}, {self.geo.stage}")
        e(f"s_add_u32 {sr(S_STG)}, {sr(S_RING)}, {sr(S_TMP)}")
        e(f"s_cmp_lt_u32 {sr(S_STG)}, {self.geo.nstage * self.geo.stage}")
        e(f"s_cselect_b32 {sr(S_TMP)}, 0, {self.geo.nstage * self.geo.stage}")
        e(f"s_sub_u32 {sr(S_STG)}, {sr(S_STG)}, {sr(S_TMP)}")     # ring + npre*self.geo.stage mod ring size (npre <= self.geo.nstage)
        Ltop, Ltopdone, Lnosw = self.new_label("top"), self.new_label("topdone"), self.new_label("nosw")
        e(f"{Ltop}:")
        e(f"s_cmp_lt_u32 {sr(S_DSTEP)}, {self.geo.nstage}")
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
        e(f"s_add_u32 {sr(S_STG)}, {sr(S_STG)}, {self.geo.stage}")
        e(f"s_cmp_lt_u32 {sr(S_STG)}, {self.geo.nstage * self.geo.stage}")
        e(f"s_cselect_b32 {sr(S_STG)}, {sr(S_STG)}, 0")
        e(f"s_branch {Ltop}")
        e(f"{Ltopdone}:")
        Lpw = self.new_label("pw")
        e(f"s_cmp_gt_u32 {sr(S_LANDED)}, 0")
        e(f"s_cbranch_scc1 {Lpw}")
        e("s_waitcnt vmcnt(0)")
        e(f"{Lpw}:")
        e("s_barrier")
        for ln in self.stage_epilogue_operands():
            e(ln)
        # enter the unrolled ring at this segment's stage
        entry = [self.new_label(f"in{j}_") for j in range(self.geo.nstage)]
        for j in range(1, self.geo.nstage):
            e(f"s_cmp_eq_u32 {sr(S_RING)}, {j * self.geo.stage}")
            e(f"s_cbranch_scc1 {entry[j]}")
        nph = 2 if "dph" in self.opts else 1
        body_in = [[self.new_label(f"b{ph}in{j}_") for j in range(self.geo.nstage)] for ph in range(nph)]
        Lexit = self.new_label("exit")
        # per entry: first fragments + the first tile-group's MFMAs from that stage, then into the ring
        for j in range(self.geo.nstage):
            e(f"{entry[j]}:")
            for ln in self.frag_reads(0, 0, j):
                e(ln)
            e("s_waitcnt lgkmcnt(0)")
            e(self.p_mfma(0, 0, 0))
            e(self.s_mfma(0, 0, 0))
            e("s_nop 7")
            if nph == 2:
                e(f"s_cmp_eq_u32 {sr(S_PHASE)}, 1")
                e(f"s_cbranch_scc1 {body_in[1][j]}")
            e(f"s_branch {body_in[0][j]}")
        ool = []
        for ph in range(nph):
            top = self.new_label(f"top{ph}_")
            # "al6" / "al8" / "al12": the ring's first body (the target of the back branch, once per NSTAGE K-steps) starts on a
            # 64 / 256 / 4096-byte boundary -- takes the loop's position in the instruction stream out of the compiler's hands
            for o in self.opts:
                if o.startswith("al") and o[2:].isdigit():
                    e(f".p2align {int(o[2:])}")
            e(f"{top}:")
            for j in range(self.geo.nstage):
                # the body's own entry label sits behind a phase-1 body's deferred-issue block: alias it
                start = len(self.lines)
                lin = self.kstep(j, Lexit, ool, ph)
                self.lines = [ln.replace(lin, body_in[ph][j]) for ln in self.lines[:start]] + \
                             [ln.replace(lin, body_in[ph][j]) for ln in self.lines[start:]]
            e(f"s_branch {top}")
        for ln in ool:
            e(ln)
        e(f"{Lexit}:")
        # drain: the speculative fragment reads and MFMAs of the non-existent next step must not be in flight when the
        # compiler's epilogue reuses their destination registers (XDL write -> VALU write needs <= 19 wait states; the
        # compiler cannot see them).  The DMAs of the next segment stay in flight: no vmcnt wait here.
        e("s_waitcnt lgkmcnt(0)")
        e("s_nop 15")
        e("s_nop 7")
        if "prio1" in self.opts or "prio0" in self.opts:
            e("s_setprio 0")
        return self.lines


def emit(path, smfma, opts="", nw=8):
    g = Gen(smfma, opts, nw)
    lines = g.build()
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_gemm_loop2.py -- do not edit.\n")
        for ln in lines:
            f.write('"' + ln + '\\n"\n')
    return len(lines)


if __name__ == "__main__":
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nunchaku_amd", "csrc")
    opts = os.environ.get("SVDQ_GEN2_OPTS", "")
    n = emit(os.path.join(root, "gemm_loop2_bf16.inc"), "v_mfma_f32_32x32x16_bf16", opts)
    emit(os.path.join(root, "gemm_loop2_fp16.inc"), "v_mfma_f32_32x32x16_f16", opts)
    n4 = emit(os.path.join(root, "gemm_loop2_w4_bf16.inc"), "v_mfma_f32_32x32x16_bf16", opts, nw=4)
    emit(os.path.join(root, "gemm_loop2_w4_fp16.inc"), "v_mfma_f32_32x32x16_f16", opts, nw=4)
    print(f"wrote gemm_loop2_{{bf16,fp16}}.inc ({n} lines each), gemm_loop2_w4_{{bf16,fp16}}.inc ({n4} lines each)")
