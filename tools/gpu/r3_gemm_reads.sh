#!/bin/bash
# round 3: what do the main loop's LDS reads cost?  Timing-only loop variants (tools/gen_gemm_loop2.py options rd20u / rd18u /
# rd0: 20 / 18 / ~0 of the 24 reads per wave and K-step; garbage results) against the product loops, same box, interleaved.
# usage: tools/gpu/r3_gemm_reads.sh <outdir-name>
O=gpurun_out/$1; mkdir -p $O
P=tools/ablate/gemm_probe
{
for rep in 1 2 3; do
for v in "" _rd20u _rd18u _rd0; do
  echo "{\"note\":\"variant=$v rep=$rep\"}"
  for s in "4608 3072 3072 0" "4608 3072 9216 3" "4608 3072 12288 2" "4608 12288 3072 0"; do
    set -- $s
    timeout 120 $P --lib tools/ablate/libsvdq_amd_probe$v.so --shape $1 $2 $3 --fuse $4 || echo "PROBE_FAILED $s rc=$?"
  done
done
done
} > $O/reads.jsonl 2> $O/reads.err
python3 - $O <<'PY'
import json,sys
for l in open(sys.argv[1]+'/reads.jsonl'):
    try: r=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    if 'note' in r: print(r['note']); continue
    if 'us' not in r: continue
    print(f"M={r['M']:5d} K={r['K']:5d} N={r['N']:5d} fuse={r['fuse']} geo={r.get('geometry')} {r['us']:7.2f} us {r['TOPS']:7.1f} TOPS {r.get('wg_cycles',0)/1e3:7.1f} kcyc {r.get('eff_GHz',0):.3f} GHz")
PY
