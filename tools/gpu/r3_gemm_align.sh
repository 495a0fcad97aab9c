#!/bin/bash
# round 3: does the position of the main loop in the instruction stream matter?  Loops whose first ring body starts on a 64 / 256 / 4096-byte
# boundary (generator options al6 / al8 / al12) against the product loops, three interleaved repetitions.  usage: tools/gpu/r3_gemm_align.sh <outdir-name>
O=gpurun_out/$1; mkdir -p $O
P=tools/ablate/gemm_probe
{
for rep in 1 2 3; do
for v in "" _al6 _al8 _al12; do
  echo "{\"note\":\"variant=$v rep=$rep\"}"
  for s in "4608 3072 3072 0 1" "4608 3072 9216 3 1" "4608 3072 9216 3 2" "4608 3072 12288 2 1" "4608 12288 3072 0 1"; do
    set -- $s
    timeout 120 $P --lib tools/ablate/libsvdq_amd_probe$v.so --shape $1 $2 $3 --fuse $4 --geoms $5 || echo "PROBE_FAILED $s rc=$?"
  done
done
done
} > $O/align.jsonl 2> $O/align.err
python3 - $O <<'PY'
import json,sys,collections
acc=collections.defaultdict(list); var=None
for l in open(sys.argv[1]+'/align.jsonl'):
    try: r=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    if 'note' in r: var=r['note'].split()[0]; continue
    if 'us' in r: acc[(r['M'],r['K'],r['N'],r['fuse'],r.get('geometry'))].append((var,r['us'],r.get('wg_cycles',0)/1e3))
for k,v in acc.items():
    print(k)
    by=collections.defaultdict(list)
    for var,us,kc in v: by[var].append((us,kc))
    for var,x in by.items(): print("   %-14s us %s   kcyc %s" % (var, " ".join("%.2f"%a for a,_ in x), " ".join("%.1f"%b for _,b in x)))
PY
