#!/bin/bash
# staged epilogue operands (ring of three + LDS staging in the 256 x 128 geometry): GEMM parity first (stop on failure), then the denoise
# step against the previous build, then every GPU test.  usage: tools/gpu/r3_staged.sh <outdir-name>
O=gpurun_out/$1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_fullsize.py tests/test_gpu_geometry_determinism.py tests/test_gpu_fused_norm.py -m gpu -x -q > $O/pytest_gemm.txt 2>&1
rc=$?; tail -8 $O/pytest_gemm.txt
if [ $rc -ne 0 ]; then echo "GEMM PARITY FAILED rc=$rc"; grep -E "^(FAILED|ERROR)|Error|assert" $O/pytest_gemm.txt | head -20; exit 1; fi
bash tools/gpu/r3_step_ab.sh $1/a tools/ablate/libsvdq_amd_prev.so nunchaku_amd/csrc/libsvdq_amd.so
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity.py --deselect tests/test_gpu_parity_fullsize.py --deselect tests/test_gpu_geometry_determinism.py --deselect tests/test_gpu_fused_norm.py > $O/pytest_rest.txt 2>&1; tail -5 $O/pytest_rest.txt
