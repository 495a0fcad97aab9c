#!/usr/bin/env python3
"""Generate the hand-scheduled main loop of svdq_gemm_w4a4 (gfx950) as inline-asm text.

    python tools/gen_gemm_loop.py        # writes nunchaku_amd/csrc/gemm_loop_{bf16,fp16}.inc

Why a generator: the loop is software-pipelined by hand (hipcc cannot be steered into this
interleave, DESIGN.md "Main loop"): for every 32x32 tile and 64-channel group ("tile-group")
    P = v_mfma_scale_f32_32x32x64_f8f6f4(W codes, A codes)      FP6 operands = exact int4 product
    S = v_mfma_f32_32x32x16_{bf16,f16}(ws, as)                  rank-1 scale tile (= 2*ws*as)
    acc += P * S                                                16 v_fmac_f32
and the MFMAs of tile-group q+1 are issued between the two fma halves of tile-group q, with the
LDS fragment reads of the next group, the LDS-DMA of K-step s+3 and the one barrier per K-step
placed in the remaining issue slots.  Register allocation is fixed (the C++ side pins its asm
operands to the same physical registers):

  v[0:63]    acc, tile t = 2*ni + mi at v[16t:16t+15]      (output operands)
  v[64:79] P0   v[80:95] S0   v[96:111] P1   v[112:127] S1
  v[128:151] fragment buffer 0 (group 0 of a K-step): W0 W1 A0 A1, 6 regs each
  v[152:175] fragment buffer 1 (group 1)
  v[176:191] scale tuples buffer 0: sw0 sw1 sa0 sa1, 4 regs each = {scale, 0, 0, 0}
  v[192:207] scale tuples buffer 1
  v208 v209  MX block exponents of the FP6 MFMA
  v210..v213 inputs: per-lane LDS offsets (stage-relative) of A frags, W frags, as, ws
  v214..v216 inputs: per-lane global byte offsets of the DMA streams A, X1, X2
  v217..v220 LDS addresses of the current stage, v221..v224 of the next stage
  s[40:41] s[42:43] s[44:45]  inputs: global base of streams A, X1, X2 (advanced here)
  s46 KP   s47 s48 s49 LDS destinations (stage-relative) of A, X1, X2   s50 s51 step of X1, X2
  s52 number of X streams (1 or 2)     s53..s57, s61 scratch
  s58 ring position = LDS stage offset of this segment's K-step 0 (in/out: segments chain)
  s59 number of this segment's K-steps whose DMA is already in flight (issued by the previous segment)
  s60 K-steps of the NEXT segment (0 = none); s[62:63] s[64:65] s[66:67] its stream bases A, X1, X2
  s68 number of leading K-steps of this segment whose DMA is KNOWN to have landed (the caller waited on
      a younger global load of its own: vmcnt retires in order), so neither the prologue nor those
      steps wait on vmcnt -- which lets the previous tile's output stores drain under this tile's loop

A "segment" is a run of K-steps [kp0, kp1) of one output tile (a whole tile, or a slice of one when
the tail of the tile list is split along K).  The DMA stream runs ahead across segment boundaries: the
last K-steps of a segment already fetch the first three K-steps of the next one, so the epilogue of a
tile overlaps the HBM/L2 latency of the next tile's first operands.
"""
import os

STAGE = 24576 + 12288 + 1024 + 1024  # A, W, as, ws(+pad)
NSTAGE = int(os.environ.get('SVDQ_GEN_NSTAGE', '4'))
CHUNK, PLANE = 3072, 1024

ACC = 0
PBUF = [64, 96]
SBUF = [80, 112]
FRAG = [128, 152]  # + 0: W0, 6: W1, 12: A0, 18: A1
SCL = [176, 192]   # + 0: sw0, 4: sw1, 8: sa0, 12: sa1
MXA, MXB = 208, 209
IN_LA, IN_LW, IN_LSA, IN_LSW = 210, 211, 212, 213
OFF_A, OFF_X1, OFF_X2 = 214, 215, 216
CUR = [217, 218, 219, 220]   # A, W, SA, SW address of the current stage
NXT = [221, 222, 223, 224]
S_PA, S_PX1, S_PX2 = 40, 42, 44
S_KP, S_DA, S_DX1, S_DX2, S_IX1, S_IX2, S_NX = 46, 47, 48, 49, 50, 51, 52
S_STEP, S_NEXT, S_TMP, S_DMASTEP, S_TOT = 53, 55, 56, 57, 61
S_CUR, S_NPRE, S_NCNT = 58, 59, 60
S_PA_N, S_PX1_N, S_PX2_N = 62, 64, 66
S_LANDED = 68


def vr(a, n=1):
    return f"v{a}" if n == 1 else f"v[{a}:{a + n - 1}]"


def sr(a, n=1):
    return f"s{a}" if n == 1 else f"s[{a}:{a + n - 1}]"


class Gen:
    def __init__(self, smfma, ablate=""):
        self.smfma = smfma
        self.ablate = ablate  # debug builds: "dma" | "lds" | "fma" | "smfma" | "barrier" dropped from the loop
        self.lines = []
        self.label = 0
        self.in_loop = False
        self.keep_all = False

    def e(self, s):
        a = self.ablate
        if self.in_loop:
            if "dma" in a.split("+") and s.startswith("global_load_lds"):
                return
            if "lds" in a.split("+") and s.startswith("ds_read"):
                return
            if "nofma" in a.split("+") or ("fma" in a.split("+")) or "pmfma+smfma+fma" in a:
                if s.startswith("v_pk_fma") or s.startswith("v_fmac"):
                    return
            if ("smfma" in a.split("+") or "pmfma+smfma+fma" in a or "nos" in a.split("+")) and s.startswith(self.smfma):
                return
            if ("pmfma" in a.split("+") or "pmfma+smfma+fma" in a or "nop" in a.split("+")) and s.startswith("v_mfma_scale"):
                return
            if "barrier" in a.split("+") and s.startswith("s_barrier"):
                return
            if "bare" in a:  # keep only the MFMAs, the fmas and the loop control
                keep = s.startswith("v_mfma") or s.startswith("v_fmac") or s.startswith("v_pk_fma") or \
                    f"s_add_u32 {sr(S_STEP)}, {sr(S_STEP)}, 1" in s or self.keep_all
                if "bare2" in a and s.startswith("s_waitcnt lgkmcnt"):
                    keep = True
                if "bare3" in a and (s.startswith("v_mov") or s.startswith("v_add") or s.startswith("s_nop")):
                    keep = True
                if "bare4" in a and s.startswith("s_") and not s.startswith("s_barrier") and not s.startswith("s_waitcnt vmcnt") and "m0" not in s:
                    keep = True
                if "bare4" in a and s.endswith(":"):
                    keep = True
                if not keep:
                    return
            if "fmov" in a and s.startswith("v_fmac"):      # same issue slots, no dependency on the MFMA results
                s = "v_mov_b32 " + s.split()[1] + " 0"
            if "findep" in a and s.startswith("v_fmac"):    # fma whose sources are not MFMA results
                s = s.split(",")[0] + f", {vr(IN_LA)}, {vr(IN_LW)}"
        self.lines.append(s)

    def new_label(self):
        self.label += 1
        return f".Lsvdq{self.label}_%="

    # ---- building blocks -------------------------------------------------------------
    def frag_reads(self, buf, grp, addr):
        """LDS reads of the 4 fragments + 4 scale tuples of one group into buffer `buf`.
        addr = [A, W, SA, SW] address VGPRs of the stage to read from.  Returns 12 instructions."""
        out = []
        for which, base_reg, roff in (("W", addr[1], 0), ("A", addr[0], 12)):
            for i in range(2):
                f = FRAG[buf] + roff + 6 * i
                o = i * CHUNK
                if grp == 0:
                    out.append(f"ds_read_b128 {vr(f, 4)}, {vr(base_reg)} offset:{o}")
                    out.append(f"ds_read_b64 {vr(f + 4, 2)}, {vr(base_reg)} offset:{o + PLANE}")
                else:
                    out.append(f"ds_read_b64 {vr(f, 2)}, {vr(base_reg)} offset:{o + PLANE + 8}")
                    out.append(f"ds_read_b128 {vr(f + 2, 4)}, {vr(base_reg)} offset:{o + 2 * PLANE}")
        for base_reg, roff in ((addr[3], 0), (addr[2], 8)):
            for i in range(2):
                out.append(f"ds_read_u16 {vr(SCL[buf] + roff + 4 * i)}, {vr(base_reg)} offset:{i * 128 + grp * 64}")
        return out

    def p_mfma(self, dst_buf, buf, t):
        ni, mi = t >> 1, t & 1
        w = FRAG[buf] + 6 * ni
        a = FRAG[buf] + 12 + 6 * mi
        return (f"v_mfma_scale_f32_32x32x64_f8f6f4 {vr(PBUF[dst_buf], 16)}, {vr(w, 6)}, {vr(a, 6)}, 0, "
                f"{vr(MXA)}, {vr(MXB)} op_sel_hi:[0,0,0] cbsz:2 blgp:2")

    def s_mfma(self, dst_buf, buf, t):
        ni, mi = t >> 1, t & 1
        sw = SCL[buf] + 4 * ni
        sa = SCL[buf] + 8 + 4 * mi
        return f"{self.smfma} {vr(SBUF[dst_buf], 16)}, {vr(sw, 4)}, {vr(sa, 4)}, 0"

    def fma(self, t, pb, lo, hi):
        # plain v_fmac_f32, NOT v_pk_fma_f32: on gfx950 the packed form does not overlap with a running
        # MFMA at all (P + 8 pk = 35 ns = sum of parts) while 16 scalar fmas do (P,8fma,S,8fma = 41 ns vs
        # 57 ns for the parts; profiles/r1_ubench_fp6_issue.jsonl, 2 waves per SIMD)
        if os.environ.get("SVDQ_GEN_PK"):
            return [f"v_pk_fma_f32 {vr(ACC + 16 * t + r, 2)}, {vr(PBUF[pb] + r, 2)}, {vr(SBUF[pb] + r, 2)}, {vr(ACC + 16 * t + r, 2)}"
                    for r in range(lo, hi, 2)]
        return [f"v_fmac_f32 {vr(ACC + 16 * t + r)}, {vr(PBUF[pb] + r)}, {vr(SBUF[pb] + r)}" for r in range(lo, hi)]

    def dma_issue(self, stage_reg):
        """5 (or 4) LDS-DMA wave loads of one K-step into the stage whose offset is in stage_reg,
        then advance the stream pointers."""
        L = self.new_label()
        o = [
            f"s_add_u32 m0, {sr(stage_reg)}, {sr(S_DA)}",
            "s_nop 0",
            f"global_load_lds_dwordx4 {vr(OFF_A)}, {sr(S_PA, 2)}",
            f"global_load_lds_dwordx4 {vr(OFF_A)}, {sr(S_PA, 2)} offset:{PLANE}",
            f"global_load_lds_dwordx4 {vr(OFF_A)}, {sr(S_PA, 2)} offset:{2 * PLANE}",
            f"s_add_u32 m0, {sr(stage_reg)}, {sr(S_DX1)}",
            f"s_add_u32 {sr(S_PA)}, {sr(S_PA)}, {CHUNK}",
            f"global_load_lds_dwordx4 {vr(OFF_X1)}, {sr(S_PX1, 2)}",
            f"s_addc_u32 {sr(S_PA + 1)}, {sr(S_PA + 1)}, 0",
            f"s_add_u32 {sr(S_PX1)}, {sr(S_PX1)}, {sr(S_IX1)}",
            f"s_addc_u32 {sr(S_PX1 + 1)}, {sr(S_PX1 + 1)}, 0",
            f"s_cmp_eq_u32 {sr(S_NX)}, 2",
            f"s_cbranch_scc0 {L}",
            f"s_add_u32 m0, {sr(stage_reg)}, {sr(S_DX2)}",
            "s_nop 0",
            f"global_load_lds_dwordx4 {vr(OFF_X2)}, {sr(S_PX2, 2)}",
            f"s_add_u32 {sr(S_PX2)}, {sr(S_PX2)}, {sr(S_IX2)}",
            f"s_addc_u32 {sr(S_PX2 + 1)}, {sr(S_PX2 + 1)}, 0",
            f"{L}:",
        ]
        return o

    def issue_step(self, stage_reg):
        """DMA of K-step S_DMASTEP (counted from the start of this segment) into the stage in stage_reg:
        from this segment while S_DMASTEP < count, from the next one while < count + next count."""
        Lskip, Lnosw = self.new_label(), self.new_label()
        return [
            f"s_cmp_lt_u32 {sr(S_DMASTEP)}, {sr(S_TOT)}",
            f"s_cbranch_scc0 {Lskip}",
            f"s_cmp_eq_u32 {sr(S_DMASTEP)}, {sr(S_KP)}",
            f"s_cbranch_scc0 {Lnosw}",
            f"s_mov_b64 {sr(S_PA, 2)}, {sr(S_PA_N, 2)}",
            f"s_mov_b64 {sr(S_PX1, 2)}, {sr(S_PX1_N, 2)}",
            f"s_mov_b64 {sr(S_PX2, 2)}, {sr(S_PX2_N, 2)}",
            f"{Lnosw}:",
        ] + self.dma_issue(stage_reg) + [f"{Lskip}:"]

    def stage_addrs(self, dst, stage_reg):
        return [f"v_add_u32 {vr(dst[i])}, {sr(stage_reg)}, {vr(src)}"
                for i, src in enumerate((IN_LA, IN_LW, IN_LSA, IN_LSW))]

    # ---- the kernel body ---------------------------------------------------------------
    def build(self):
        e = self.e
        e("; ---- svdq gemm main loop (generated by tools/gen_gemm_loop.py) ----")
        for r in range(64):
            e(f"v_mov_b32 {vr(ACC + r)}, 0")
        for b in range(2):
            for tpl in range(4):
                for k in range(1, 4):
                    e(f"v_mov_b32 {vr(SCL[b] + 4 * tpl + k)}, 0")
        e(f"v_mov_b32 {vr(MXA)}, 0x82828282")
        e(f"v_mov_b32 {vr(MXB)}, 0x81818181")
        e(f"s_mov_b32 {sr(S_STEP)}, 0")
        e(f"s_add_u32 {sr(S_TOT)}, {sr(S_KP)}, {sr(S_NCNT)}")
        # prologue: top the DMA stream up to three K-steps in flight (all three for the first segment of a
        # workgroup, normally none afterwards: the previous segment already fetched them)
        e(f"s_mov_b32 {sr(S_NEXT)}, {sr(S_CUR)}")
        for j in range(NSTAGE):
            L = self.new_label()
            e(f"s_mov_b32 {sr(S_DMASTEP)}, {j}")
            e(f"s_cmp_le_u32 {sr(S_NPRE)}, {j}")
            e(f"s_cbranch_scc0 {L}")
            for ln in self.issue_step(S_NEXT):
                e(ln)
            e(f"{L}:")
            e(f"s_add_u32 {sr(S_NEXT)}, {sr(S_NEXT)}, {STAGE}")
            e(f"s_cmp_lt_u32 {sr(S_NEXT)}, {NSTAGE * STAGE}")
            e(f"s_cselect_b32 {sr(S_NEXT)}, {sr(S_NEXT)}, 0")
        e(f"s_mov_b32 {sr(S_DMASTEP)}, {NSTAGE}")
        # S_NEXT has gone once round the ring: back at S_CUR; advance to the stage of K-step 1
        e(f"s_add_u32 {sr(S_NEXT)}, {sr(S_CUR)}, {STAGE}")
        e(f"s_cmp_lt_u32 {sr(S_NEXT)}, {NSTAGE * STAGE}")
        e(f"s_cselect_b32 {sr(S_NEXT)}, {sr(S_NEXT)}, 0")
        for ln in self.stage_addrs(CUR, S_CUR) + self.stage_addrs(NXT, S_NEXT):
            e(ln)
        Lpw = self.new_label()
        e(f"s_cmp_gt_u32 {sr(S_LANDED)}, 0")
        e(f"s_cbranch_scc1 {Lpw}")
        e("s_waitcnt vmcnt(0)")
        e(f"{Lpw}:")
        e("s_barrier")
        for ln in self.frag_reads(0, 0, CUR):
            e(ln)
        e("s_waitcnt lgkmcnt(0)")
        e(self.p_mfma(0, 0, 0))
        e(self.s_mfma(0, 0, 0))
        e("s_nop 7")

        loop = self.new_label()
        e(f"{loop}:")
        self.in_loop = True
        reads_g1 = self.frag_reads(1, 1, CUR)   # (s, group 1) from the current stage
        reads_n0 = self.frag_reads(0, 0, NXT)   # (s+1, group 0) from the next stage
        for q in range(8):
            t = q & 3
            pb = q & 1                      # P/S buffer holding tile-group q
            qn = q + 1
            nbuf, nt = (qn >> 2) & 1, qn & 3  # fragment buffer / tile of tile-group q+1
            if q in (3, 7):
                e("s_waitcnt lgkmcnt(0)")
            e(self.p_mfma(pb ^ 1, nbuf, nt))
            misc = []
            if q < 3:
                misc = reads_g1[4 * q:4 * q + 4]
            elif q == 3:
                misc = ["s_nop 2"]
            elif q == 4:
                # stage s+1 must have landed (own DMAs), then rendezvous: afterwards every wave's planes
                # of stage s+1 are visible and nobody reads stage s any more -> refill it with step s+3
                Lwd = self.new_label()
                misc = [
                    f"s_add_u32 {sr(S_TMP)}, {sr(S_STEP)}, 1",
                    f"s_cmp_lt_u32 {sr(S_TMP)}, {sr(S_LANDED)}",
                    f"s_cbranch_scc1 {Lwd}",
                ]
                # own DMAs of K-step s+1 must have landed; the younger ones (K-steps s+2 .. s+NSTAGE-1, as far
                # as they exist) may stay in flight: vmcnt retires in order, nd DMA instructions per K-step
                misc += [f"s_sub_u32 {sr(S_TMP)}, {sr(S_TOT)}, {sr(S_STEP)}"]
                klabels = {kk: self.new_label() for kk in range(1, NSTAGE - 1)}
                for kk in range(NSTAGE - 2, 0, -1):
                    misc += [f"s_cmp_ge_u32 {sr(S_TMP)}, {kk + 2}", f"s_cbranch_scc1 {klabels[kk]}"]
                misc += ["s_waitcnt vmcnt(0)", f"s_branch {Lwd}"]
                for kk in range(1, NSTAGE - 1):
                    L4 = self.new_label()
                    misc += [f"{klabels[kk]}:", f"s_cmp_eq_u32 {sr(S_NX)}, 2", f"s_cbranch_scc0 {L4}",
                             f"s_waitcnt vmcnt({5 * kk})", f"s_branch {Lwd}", f"{L4}:", f"s_waitcnt vmcnt({4 * kk})",
                             f"s_branch {Lwd}"]
                misc += [
                    f"{Lwd}:",
                    "s_barrier",
                ] + self.issue_step(S_CUR) + reads_n0[0:4]
            elif q in (5, 6):
                misc = reads_n0[4 * (q - 4):4 * (q - 4) + 4]
            else:  # q == 7: rotate the ring
                misc = [
                    f"s_mov_b32 {sr(S_CUR)}, {sr(S_NEXT)}",
                    f"s_add_u32 {sr(S_NEXT)}, {sr(S_NEXT)}, {STAGE}",
                    f"s_cmp_lt_u32 {sr(S_NEXT)}, {NSTAGE * STAGE}",
                    f"s_cselect_b32 {sr(S_NEXT)}, {sr(S_NEXT)}, 0",
                    f"s_add_u32 {sr(S_DMASTEP)}, {sr(S_DMASTEP)}, 1",
                    f"s_add_u32 {sr(S_STEP)}, {sr(S_STEP)}, 1",
                ] + [f"v_mov_b32 {vr(CUR[i])}, {vr(NXT[i])}" for i in range(4)]
            order = int(os.environ.get("SVDQ_GEN_ORDER", "0"))
            for tok in self.ablate.split("+"):
                if tok.startswith("ord"):
                    order = int(tok[3:])
            split = {0: 8, 1: 0, 2: 4, 3: 12, 4: 16}[order]
            if order == 1:
                e(self.s_mfma(pb ^ 1, nbuf, nt))
            for ln in misc:
                e(ln)
            for ln in self.fma(t, pb, 0, split):
                e(ln)
            if order != 1:
                e(self.s_mfma(pb ^ 1, nbuf, nt))
            if q == 7:
                for ln in self.stage_addrs(NXT, S_NEXT):
                    e(ln)
            for ln in self.fma(t, pb, split, 16):
                e(ln)
        self.keep_all = True
        e(f"s_cmp_lt_u32 {sr(S_STEP)}, {sr(S_KP)}")
        e(f"s_cbranch_scc1 {loop}")
        self.keep_all = False
        self.in_loop = False
        # drain: the speculative MFMAs / fragment reads of the non-existent next step
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        e("s_nop 15")
        e("s_nop 15")
        e("s_nop 15")
        return self.lines


def emit(path, smfma, ablate=""):
    g = Gen(smfma, ablate)
    lines = g.build()
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_gemm_loop.py -- do not edit.\n")
        for ln in lines:
            f.write('"' + ln + '\\n"\n')
    return len(lines)


ABLATIONS = ("dma", "lds", "fma", "smfma", "barrier", "dma+lds", "fma+smfma", "dma+barrier", "dma+lds+barrier",
             "bare", "bare+bare2", "bare+bare3", "bare+bare4", "pmfma+smfma+fma",
             # compute side alone (variants 16..): what keeps P + S + 16 fma from the 64-cycle MFMA bound
             "bare+nofma", "bare+nos", "bare+nop", "bare+fmov", "bare+findep", "bare+ord1", "bare+ord2", "bare+ord3", "bare+ord4",
             # memory side alone (variants 25..): DMA / LDS reads / barrier without any arithmetic
             "pmfma+smfma+fma+lds", "pmfma+smfma+fma+dma", "pmfma+smfma+fma+barrier", "pmfma+smfma+fma+lds+barrier",
             "pmfma+smfma+fma+dma+barrier", "pmfma+smfma+fma+dma+lds")


if __name__ == "__main__":
    # product loops only; the timing ablations are generated by tools/ablate/build.py into tools/ablate/gen/ (untracked)
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nunchaku_amd", "csrc")
    n = emit(os.path.join(root, "gemm_loop_bf16.inc"), "v_mfma_f32_32x32x16_bf16")
    emit(os.path.join(root, "gemm_loop_fp16.inc"), "v_mfma_f32_32x32x16_f16")
    print(f"wrote gemm_loop_{{bf16,fp16}}.inc ({n} lines each)")
