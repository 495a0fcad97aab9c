#!/bin/bash
# round 6: the K = 8 scale-tile MFMA in every product loop: parity suites, isolated launches old vs new library (gpurun_in/old = the tree before), same-box step A/B
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_fullsize.py tests/test_gpu_geometry_determinism.py tests/test_capi.py tests/test_gpu_fused_norm.py -m gpu -q -x > $O/pytest_parity.txt 2>&1; tail -4 $O/pytest_parity.txt
{
for s in "4608 3072 3072 0" "4608 12288 3072 0" "4608 3072 12288 2" "4608 3072 9216 3"; do set -- $s
  for rep in 1 2; do for l in gpurun_in/old/nunchaku_amd/csrc/libsvdq_amd.so nunchaku_amd/csrc/libsvdq_amd.so; do
    echo -n "fuse=$4 K=$2 N=$3 $(echo $l | cut -c1-10): "; timeout 120 tools/ablate/gemm_probe --lib $l --shape $1 $2 $3 --fuse $4 --R 32 --R2 32 --geoms 0 --iters 50 | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(r[chr(117)+chr(115)], chr(117)+chr(115))"
  done; done
done
} 2>&1 | tee $O/launch_ab.txt
bash tools/gpu_ab_step.sh 2>&1 | tee $O/step_ab.txt
