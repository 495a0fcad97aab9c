#!/bin/bash
# full GPU test suite + optional extra command; usage: tools/gpu/r3_tests.sh <outdir-name> [quick test paths...]
O=gpurun_out/$1; mkdir -p $O; shift
if [ $# -gt 0 ]; then timeout 900 python -m pytest "$@" -m gpu -x -q > $O/pytest_quick.txt 2>&1; tail -30 $O/pytest_quick.txt; fi
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_all.txt 2>&1; tail -30 $O/pytest_all.txt
