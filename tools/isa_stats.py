#!/usr/bin/env python3
"""Static instruction statistics of the GEMM kernels in libsvdq_amd.so (gfx950 code objects, llvm-objdump).

For every gemm_w4a4_kernel<DT, FUSE, NW, LAQ, CARRY> instantiation: instruction-class counts of the code BEHIND the generated main loop (everything after
the last FP6 product MFMA of the loop: the epilogue paths, the stream-K publish / collect code and the schedule bookkeeping -- a static count, an upper
bound of what one tile executes) and, inside that, of the address range that holds no stream-K code.  tests/test_generators.py holds the counts to
per-epilogue budgets so that an edit that fattens an epilogue is caught on CPU.

    python tools/isa_stats.py [path/to/libsvdq_amd.so]
"""
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
FUSE_NAMES = {0: "default", 1: "silu", 2: "gelu_quant", 3: "rmsnorm_rope"}


def disassemble(lib):
    """{mangled kernel name: [instruction lines]} of every device function in the library"""
    tmp = tempfile.mkdtemp(prefix="svdq_isa_")
    try:
        shutil.copy(lib, os.path.join(tmp, "lib.so"))
        subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        funcs = {}
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([OBJDUMP, "-d", f], cwd=tmp, check=True, capture_output=True, text=True).stdout.split("\n")
            cur = None
            for ln in txt:
                m = re.match(r"^[0-9a-f]+ <(.*)>:", ln)
                if m:
                    cur = funcs.setdefault(m.group(1), [])
                elif cur is not None and re.match(r"\s+[a-z]", ln):
                    cur.append(ln.strip())
        return funcs
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def kernel_resources(lib, pattern):
    """{mangled kernel name: {vgpr_count, agpr_count, sgpr_count, vgpr_spill_count, private_segment_fixed_size, group_segment_fixed_size}} of the
    kernels whose name contains `pattern` (the code objects' metadata notes, llvm-readelf)"""
    readelf = os.path.join(os.path.dirname(OBJDUMP), "llvm-readelf")
    tmp = tempfile.mkdtemp(prefix="svdq_isa_")
    out = {}
    try:
        shutil.copy(lib, os.path.join(tmp, "lib.so"))
        subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([readelf, "--notes", f], cwd=tmp, check=True, capture_output=True, text=True).stdout
            for block in txt.split("  - .agpr_count:")[1:]:
                rec = {"agpr_count": int(block.split("\n")[0])}
                for ln in block.split("\n")[1:]:
                    m = re.match(r"\s+\.(\w+):\s+(\S+)", ln)
                    if m:
                        rec[m.group(1)] = m.group(2)
                name = rec.get("name", "")
                if pattern in name:
                    out[name] = {k: int(rec[k]) for k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "private_segment_fixed_size",
                                                          "group_segment_fixed_size") if k in rec}
                    out[name]["agpr_count"] = rec["agpr_count"]
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def gemm_stats(funcs):
    out = {}
    for name, ins in funcs.items():
        m = re.search(r"gemm_w4a4_kernelILi(\d)ELi(\d)ELi(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)E", name)
        if not m:
            continue
        dt, fuse, nw, laq, carry, rall, hyb, split = (int(x) for x in m.groups())
        if rall:  # the rank 48 .. 160 kernels (all-rank lora_up image): keyed with carry = 2
            carry = 2
        if hyb:   # the hybrid carry kernels (next-layer rank > 32: carry for the first 32 ranks, atomics behind): carry = 3
            carry = 3
        if split:  # the all-rank GELU_QUANT kernel that stores 16-bit fragments for the split low-rank down projection: carry = 4
            carry = 4
        ops = [ln.split()[0] for ln in ins]
        loop = [i for i, o in enumerate(ops) if o.startswith(("v_mfma_scale", "v_mfma_f32_32x32x64_f8f6f4"))]  # the FP6 product MFMA (rounds 1-5: the MX-scaled pair)
        post = ops[loop[-1] + 1:]
        cls = collections.Counter(classify(o) for o in post)
        hist = collections.Counter(post)
        out[(dt, fuse, nw, laq, carry)] = {
            "post_loop": dict(cls), "histogram": hist,
            "scratch": sum(1 for o in ops if o.startswith("scratch_")),
            "lds_atomics": sum(1 for o in ops if o.startswith(("ds_add_f32", "ds_add_rtn_f32"))),
            "lds_cas": sum(1 for o in ops if o.startswith("ds_cmpst")),
            "global_atomics": sum(1 for o in ops if o.startswith("global_atomic")),
            "total": len(ops),
        }
    return out


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "nunchaku_amd", "csrc", "libsvdq_amd.so")
    st = gemm_stats(disassemble(lib))
    for (dt, fuse, nw, laq, carry), r in sorted(st.items()):
        p = r["post_loop"]
        print(f"{'bf16' if dt == 0 else 'fp16'} {FUSE_NAMES[fuse]:13s} NW={nw} LAQ={laq} CARRY={carry}{' (RALL)' if carry == 2 else ' (HYB)' if carry == 3 else ' (SPLIT)' if carry == 4 else ''}: behind the loop VALU {p.get('valu', 0):5d} MFMA {p.get('mfma', 0):3d} "
              f"SALU {p.get('salu', 0):5d} LDS {p.get('lds', 0):4d} VMEM {p.get('vmem', 0):4d} | scratch {r['scratch']} ds_add_f32 {r['lds_atomics']} "
              f"ds_cmpst {r['lds_cas']} global_atomic {r['global_atomics']}")
    if "-v" in sys.argv:
        for k, r in sorted(st.items()):
            print(k, [(o, n) for o, n in r["histogram"].most_common(30) if o.startswith("v_")])


if __name__ == "__main__":
    main()
