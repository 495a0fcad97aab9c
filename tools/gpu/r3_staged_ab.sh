#!/bin/bash
# staged epilogue operands: isolated launches, previous product build vs this one, same box, three interleaved repetitions (256 x 128 geometry)
O=gpurun_out/$1; mkdir -p $O
P=tools/ablate/gemm_probe
{
for rep in 1 2 3; do
for lib in tools/ablate/libsvdq_amd_prev.so nunchaku_amd/csrc/libsvdq_amd.so; do
  echo "{\"note\":\"lib=$lib rep=$rep\"}"
  for s in "4608 3072 3072 0" "4608 3072 12288 2" "4608 12288 3072 0" "4608 3072 9216 3" "512 3072 3072 0"; do
    set -- $s
    timeout 120 $P --lib $lib --shape $1 $2 $3 --fuse $4 --geoms 1 || echo "PROBE_FAILED $s rc=$?"
  done
done
done
} > $O/ab.jsonl 2> $O/ab.err
python3 - $O <<'PY'
import json,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list)); var=None
for l in open(sys.argv[1]+'/ab.jsonl'):
    try: r=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    if 'note' in r: var='previous' if 'prev' in r['note'] else 'staged'; continue
    if 'us' in r: acc[(r['M'],r['K'],r['N'],r['fuse'])][var].append(r['us'])
for k,v in acc.items():
    print("M=%d K=%d N=%d fuse=%d  " % k + "   ".join("%s %s" % (var, " ".join("%.2f"%a for a in x)) for var,x in v.items()))
PY
