#!/usr/bin/env python3
"""Instruction placement of one KV-tile iteration of the 4-wave x 64-row attention kernel (attention.hip, geometry 2).

One wave per SIMD: nothing but this wave's own instruction stream can fill the gap behind an MFMA.  What fits was measured on
the part (tools/ablate/filler_probe.hip -> profiles/r3_mfma_filler_prices.txt, cycles per v_mfma_f32_32x32x16_bf16 with N
independent fillers of one kind between two MFMAs):

    nothing 34.1 | plain VALU (fma, cvt_pk, max3, add, mov): 35.1 35.6 36.6 37.1 at N = 1 2 4 5, then 43 (6) and 58 (8)
    v_exp_f32, v_permlane32_swap: 35.6 36.6 45.6 53.6 at N = 1 2 4 5 -- two slots each
    v_dot2c_f32_bf16, v_pk_mul_f32: 53.6 at N = 1 (+4 per further one) -- they wait for the matrix pipe: never beside an MFMA
    ds_read_b128: 35.1 35.9 at N = 1 2, 64 at N = 4 (16 cycles of LDS pipe each) | s_nop, SALU: 0.5 each

So a slot = one MFMA + at most 5 units of other work (plain VALU 1, exp / swap 2, at most one fragment read).  hipcc's scheduler,
given the iteration as one block, puts most of the softmax in front of the MFMAs and then issues the MFMAs back to back; this
script writes the iteration as SLOTS separated by __builtin_amdgcn_sched_barrier(0), so the order below is the order in the binary:

  slots  0..31   S(j+1) = K(j+1) Q^T            (ds-major: 4 accumulator chains)
  slots 32..63   O += V^T(j) P(j)               (key-step-major: 8 chains); the LAST slot of a key step also takes the step's 8
                                                 v_dot2c (row sums over the rounded probabilities): together they cost ~45 cycles
                                                 once, one by one they would cost ~20 each
  the exp / pack stream of S(j) (pieces in the order the PV MFMAs need them) fills the slots from the front, the row maxima of
  S(j+1) follow once its last MFMA has retired; K / V^T fragment reads LEAD slots ahead of their first MFMA.

Output: C++ statements over the A64_* macros of attention.hip -> nunchaku_amd/csrc/attention_step64.inc.
    python tools/gen_attn_step.py [--lead N] [--pre N] [--budget U]
"""
import argparse
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNITS = {"A64_DOT8": 0, "A64_FMA": 1, "A64_EXP": 2, "A64_CVT": 1, "A64_SWAP": 2, "A64_FIN": 0, "A64_NMC": 3, "A64_RMAX": 1, "A64_SETTLE": 0}


def units(op):
    return UNITS[op.split("(")[0]]


def build(lead=4, pre=28, budget=5.25, rm_from=36, drop=0, stamps=0, dma0=1, dma_units=1.0):
    """drop (timing experiments, wrong results): bit 0 = no exp, bit 1 = no softmax VALU at all, bit 2 = no fragment reads, bit 3 = no row maxima"""
    mf = [("QK", ds, kt, rt) for ds in range(8) for kt in range(2) for rt in range(2)]
    first_pv = {}
    for ks in range(4):
        first_pv[ks] = len(mf)
        mf += [("PV", ks, dt, rt) for dt in range(4) for rt in range(2)]
    n = len(mf)
    slots = [[] for _ in range(n)]
    reads = [0.0] * n
    preamble = []

    def place_read(slot, op):
        if slot < 0:
            preamble.append(op)
        else:
            slots[slot].append(op)
            reads[slot] += 0.25

    for ds in range(8):
        for kt in range(2):
            place_read((ds * 2 + kt) * 2 - lead, f"A64_KREAD({ds}, {kt})")
    for ks in range(4):
        for dt in range(4):
            place_read(first_pv[ks] + dt * 2 - lead, f"A64_VREAD({ks}, {dt})")

    # the softmax stream, in dependency order: pieces (key step ks, row tile rt) in the order the PV MFMAs need them
    ex = []
    for ks in range(4):
        for rt in range(2):
            for i in range(8):
                ex.append(f"A64_FMA({rt}, {ks}, {i})")
                if i >= 1:
                    ex.append(f"A64_EXP({rt}, {ks}, {i - 1})")
            ex.append(f"A64_EXP({rt}, {ks}, 7)")
            ex += [f"A64_CVT({rt}, {ks}, {d})" for d in range(4)]
            ex += [f"A64_SWAP({rt}, {ks}, {d2})" for d2 in range(2)]
            ex.append(f"A64_FIN({rt}, {ks})")
    rm = ["A64_SETTLE"] + [f"A64_RMAX({rt}, {i})" for i in range(17) for rt in range(2)]

    u = 0
    while ex and u + units(ex[0]) <= pre:
        u += units(ex[0])
        preamble.append(ex.pop(0))
    fin_slot = {}
    dot_slot = {first_pv[ks] + 7: ks for ks in range(4)}
    # the 8 LDS-DMA pieces of the next tiles: one per slot, early, so that they have the rest of the iteration to land
    dma_slot = {dma0 + 2 * i: (f"A64_DMAK({i})" if i < 4 else f"A64_DMAV({i - 4})") for i in range(8)}
    for i in range(n):
        u = reads[i]
        if i in dot_slot:  # this slot is the dots': nothing else beside them
            slots[i].append(f"A64_DOT8({dot_slot[i]})")
            continue
        if i in dma_slot:
            slots[i].append(dma_slot[i])
            u += dma_units
        while ex and u + units(ex[0]) <= budget:
            op = ex.pop(0)
            u += units(op)
            slots[i].append(op)
            if op.startswith("A64_FIN"):
                fin_slot[op] = i
        if not ex and i >= rm_from:
            while rm and u + units(rm[0]) <= budget:
                u += units(rm[0])
                slots[i].append(rm.pop(0))
    assert not ex, f"{len(ex)} softmax operations do not fit: raise --budget"
    tail = rm  # row-maximum operations that found no slot follow the last MFMA
    for ks in range(4):
        for rt in range(2):
            s = fin_slot.get(f"A64_FIN({rt}, {ks})", -1)
            assert s < first_pv[ks], f"P fragment ({rt}, {ks}) is ready in slot {s}, its first PV MFMA is slot {first_pv[ks]}"

    def keep(op):
        name = op.split("(")[0]
        if drop & 1 and name == "A64_EXP": return False
        if drop & 2 and name in ("A64_FMA", "A64_EXP", "A64_CVT", "A64_SWAP", "A64_NMC"): return False
        if drop & 4 and name in ("A64_KREAD", "A64_VREAD"): return False
        if drop & 8 and name == "A64_RMAX": return False
        return True

    out = [f"// generated by tools/gen_attn_step.py --lead {lead} --pre {pre} --budget {budget}" + (f" drop={drop} (TIMING ONLY)" if drop else "") + ": do not edit",
           *filter(keep, preamble), "A64_SB"]
    for i, (kind, a, b, rt) in enumerate(mf):
        if stamps and i % 8 == 0:
            out.append(f"A64_STAMP({i // 8})")
        call = f"A64_{kind}({a}, {b}, {rt})"
        out.append(call + " " + " ".join(filter(keep, slots[i])) + " A64_SB")
    if tail:
        out.append(" ".join(filter(keep, tail)) + " A64_SB")
    if stamps:
        out.append("A64_STAMP(8)")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--lead", type=int, default=4)
    ap.add_argument("--pre", type=int, default=28)
    ap.add_argument("--budget", type=float, default=5.25)
    ap.add_argument("-o", default=os.path.join(ROOT, "nunchaku_amd", "csrc", "attention_step64.inc"))
    a = ap.parse_args()
    open(a.o, "w").write(build(a.lead, a.pre, a.budget))
    print(a.o)
