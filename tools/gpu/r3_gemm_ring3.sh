#!/bin/bash
# round 3, last experiment: the 256 x 128 geometry on a ring of THREE stages (generator option st3; frees 38 KB of LDS, e.g. for staging a
# tile's epilogue operands) against the product ring of four: launch times, three interleaved repetitions, and the output checksums
# (launches that are not split along K must be bit-identical).  Record of a finished experiment: the ring of three IS the product since then
# (profiles/r3_gemm_ring3.txt).  usage: tools/gpu/r3_gemm_ring3.sh <outdir-name>
O=gpurun_out/$1; mkdir -p $O
P=tools/ablate/gemm_probe
{
for rep in 1 2 3; do
for v in "" _st3; do
  echo "{\"note\":\"variant=$v rep=$rep\"}"
  for s in "4608 3072 3072 0" "4608 3072 9216 3" "4608 3072 12288 2" "4608 12288 3072 0" "512 3072 3072 0"; do
    set -- $s
    timeout 120 $P --lib tools/ablate/libsvdq_amd_probe$v.so --shape $1 $2 $3 --fuse $4 --geoms 1 || echo "PROBE_FAILED $s rc=$?"
  done
done
done
} > $O/ring3.jsonl 2> $O/ring3.err
python3 - $O <<'PY'
import json,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list)); var=None
for l in open(sys.argv[1]+'/ring3.jsonl'):
    try: r=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    if 'note' in r: var=r['note'].split()[0]; continue
    if 'us' in r: acc[(r['M'],r['K'],r['N'],r['fuse'])][var].append((r['us'],r.get('wg_cycles',0)/1e3,r.get('sum')))
for k,v in acc.items():
    print(k)
    for var,x in v.items(): print("   %-14s us %s   kcyc %s   sums %s" % (var, " ".join("%.2f"%a for a,_,_ in x), " ".join("%.1f"%b for _,b,_ in x), sorted(set(c for _,_,c in x))))
PY
tail -3 $O/ring3.err
