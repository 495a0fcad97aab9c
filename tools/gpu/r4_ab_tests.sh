#!/bin/bash
# round 4: full GPU test suite + same-box step A/B against the round-3 library.   usage: r4_ab_tests.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
T0=$(date +%s); timeout 1700 python -m pytest tests -m gpu -q > $O/pytest_all.txt 2>&1; tail -6 $O/pytest_all.txt; echo "pytest $(( $(date +%s) - T0 )) s"
bash tools/gpu/r3_step_ab.sh $1 tools/ablate/libsvdq_amd_r3.so nunchaku_amd/csrc/libsvdq_amd.so
