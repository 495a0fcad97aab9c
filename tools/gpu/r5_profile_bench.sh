#!/bin/bash
# round 5: rocprofv3 evidence for profiles/ -- the round-4 script with the round-5 names (kernel-trace stats of the bench command, fabric-side traffic and
# matrix-pipe utilisation / clock of every gemm_w4a4 dispatch of the same command in separate --pmc passes).   usage: r5_profile_bench.sh <outdir-name> [bench args...]
exec bash tools/gpu/r4_profile_bench.sh "$@"
