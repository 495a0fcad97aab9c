"""Build the HIP shared library in-tree (nunchaku_amd/csrc/libsvdq_amd.so) with hipcc for gfx950."""

import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libsvdq_amd.so")
SOURCES = ["repack.hip", "quantize.hip", "gemm_w4a4.hip", "attention.hip", "gemv_awq.hip", "residual.hip"]
HEADERS = ["svdq_common.h", "lowrank_split.h", "attention_loop64_bf16.inc", "attention_loop64_fp16.inc", "gemm_loop2_bf16.inc", "gemm_loop2_fp16.inc", "gemm_loop2_w4_bf16.inc", "gemm_loop2_w4_fp16.inc", "gemm_loop3_bf16.inc", "gemm_loop3_fp16.inc", "gemm_epi3_bf16.inc", "gemm_epi3_fp16.inc", os.path.join("..", "..", "include", "svdq_amd.h")]
# -ffp-contract=off: the quantiser's arithmetic is specified operation by operation;
# the hot loop uses explicit fma.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, *FLAGS, "-o", LIB, *SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, cwd=CSRC, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
