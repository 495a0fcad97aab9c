#!/bin/bash
# round 4: GPU test suite (optionally a -k expression / file list) with timing.  usage: r4_tests.sh <outdir> [pytest args...]
O=gpurun_out/$1; shift; mkdir -p $O
if [ $# -eq 0 ]; then set -- tests; fi
T0=$(date +%s)
timeout 1700 python -m pytest "$@" -m gpu -q --durations=15 > $O/pytest.txt 2>&1; tail -45 $O/pytest.txt; echo "elapsed $(( $(date +%s) - T0 )) s"
