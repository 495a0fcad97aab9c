#!/bin/bash
# epilogue change: GEMM parity tests, then the denoise step against the previous build (tools/ablate/libsvdq_amd_prev.so), interleaved.
# usage: tools/gpu/r3_epi_ab.sh <outdir-name>
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_fullsize.py tests/test_gpu_geometry_determinism.py tests/test_gpu_fused_norm.py -m gpu -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
bash tools/gpu/r3_step_ab.sh $1/a tools/ablate/libsvdq_amd_prev.so nunchaku_amd/csrc/libsvdq_amd.so
bash tools/gpu/r3_step_ab.sh $1/b tools/ablate/libsvdq_amd_prev.so nunchaku_amd/csrc/libsvdq_amd.so
