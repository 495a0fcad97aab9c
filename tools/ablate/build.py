#!/usr/bin/env python3
"""Build the probe twin of the product library: tools/ablate/libsvdq_amd_probe.so (-DSVDQ_PROBE) + tools/ablate/gemm_probe.

The product library (nunchaku_amd/csrc/libsvdq_amd.so) contains no experiment switches.  The probe build is the SAME sources
plus tools/ablate/gemm_probe_hooks.inc: per-workgroup {shader cycles, 100 MHz ticks} of every GEMM launch (effective clock)
and workgroup 0's phase stamps per tile.  Optionally the main loops are generated with an option string of
tools/gen_gemm_loop2.py (timing experiments; "nobar" computes garbage):

    python tools/ablate/build.py                 # probe library with the product loops
    python tools/ablate/build.py nobar           # -> libsvdq_amd_probe_nobar.so with the loops generated with option "nobar"

(The round-1 loop generator and its instruction-class ablations were removed in round 3; their measurements are
profiles/r1_gemm_ablation_*.txt and profiles/r2_gemm_ablation_cycles.txt, the code is in the git history.)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "nunchaku_amd", "csrc")
GEN = os.path.join(HERE, "gen")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_gemm_loop2 as G2  # noqa: E402

SOURCES = ["repack.hip", "quantize.hip", "gemm_w4a4.hip", "attention.hip", "gemv_awq.hip", "residual.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-DSVDQ_PROBE", f"-I{HERE}"]


def build(opts: str = "", opts3: str = ""):
    """opts: option string of gen_gemm_loop2.py (the 64 x 64 wave tile loops); opts3: extra options of gen_gemm_loop3.py (the 128 x 64 wave tile kernel's loop and
    epilogue are ALWAYS regenerated for the probe library with "stamp": phase stamps inside the one-statement loop + epilogue)"""
    import gen_gemm_loop3 as G3
    flags = list(FLAGS)
    suffix = ""
    os.makedirs(GEN, exist_ok=True)
    o3 = G3.PRODUCT_OPTS + "+stamp" + ("+" + opts3 if opts3 else "")
    s3 = "_" + o3.replace("+", "_")
    for dt, mf in (("bf16", "v_mfma_f32_32x32x16_bf16"), ("fp16", "v_mfma_f32_32x32x16_f16")):
        G3.emit(os.path.join(GEN, f"gemm_loop3_{dt}{s3}.inc"), mf, o3)
        G3.emit_epilogue(os.path.join(GEN, f"gemm_epi3_{dt}{s3}.inc"), dt, o3)
        flags += [f'-DSVDQ_LOOP3_INC_{dt.upper()}="gemm_loop3_{dt}{s3}.inc"', f'-DSVDQ_EPI3_INC_{dt.upper()}="gemm_epi3_{dt}{s3}.inc"']
    flags.append(f"-I{GEN}")
    if opts3:
        suffix = "_wt_" + opts3.replace("+", "_")
    if opts:
        suffix += "_" + opts.replace("+", "_")
        for nw, tag in ((8, "8"), (4, "4")):
            for dt, mf in (("BF16", "v_mfma_f32_32x32x16_bf16"), ("FP16", "v_mfma_f32_32x32x16_f16")):
                name = f"gemm_loop2_w{nw}_{dt.lower()}{suffix}.inc"
                G2.emit(os.path.join(GEN, name), mf, opts, nw=nw)
                flags.append(f'-DSVDQ_LOOP_INC_{tag}_{dt}="{name}"')
    # SVDQ_PROBE_DEFS="-DSVDQ_PROBE_ROT=1 ..." SVDQ_PROBE_TAG=rot1: compile-time timing variants of an epilogue (gemm_w4a4.hip "compile-time timing variants")
    if os.environ.get("SVDQ_PROBE_DEFS"):
        flags += os.environ["SVDQ_PROBE_DEFS"].split()
        suffix += "_" + os.environ.get("SVDQ_PROBE_TAG", "defs")
    lib = os.path.join(HERE, f"libsvdq_amd_probe{suffix}.so")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run([hipcc, *flags, "-o", lib, *SOURCES], cwd=CSRC, check=True)
    # the hash of the kernel sources this probe library was built from (bench.kernel_sources_sha16): tools/epilogue_share.py stamps its output with THIS value,
    # so a trace taken with a probe library older than the product sources shows up as stale in the bench line instead of borrowing the current hash
    sys.path.insert(0, ROOT)
    import bench
    with open(lib + ".sha16", "w") as f:
        f.write(bench.kernel_sources_sha16())
    probe = os.path.join(HERE, "gemm_probe")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", f"-I{os.path.join(ROOT, 'include')}", "-o", probe,
                    os.path.join(HERE, "gemm_probe.hip"), "-ldl"], check=True)
    return lib, probe


if __name__ == "__main__":
    print(*build(sys.argv[1] if len(sys.argv) > 1 else "", sys.argv[2] if len(sys.argv) > 2 else ""), sep="\n")
