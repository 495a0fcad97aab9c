#!/bin/bash
# round 6: scale-tile MFMA as the legacy K = 8 form (same passes, half the multiplier activity): power / clock effect on the wave-tile loop probe, interleaved
O=gpurun_out/$1; mkdir -p $O
tools/ablate/mfma_k8_probe > $O/mfma_k8.jsonl
for rep in 1 2 3; do for v in ${VARIANTS:-b2 b2_k8}; do for s in "4608 3072 3072" "4608 12288 3072"; do set -- $s
  echo -n "$v K=$2: "; timeout 120 tools/ablate/gemm128_probe_$v --shape $1 $2 $3 --iters 100 --warm 200 | python3 -c "import sys,json; r=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(r['us'], 'us', r['eff_GHz'], 'GHz', r['cycles_per_tile_group_incl_epilogue'], 'cyc/tg  mismatches', r['mismatches'])"
done; done; done 2>&1 | tee $O/k8_ab.txt
