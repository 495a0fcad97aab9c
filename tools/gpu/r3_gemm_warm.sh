#!/bin/bash
# round 3: L2 warm-up of a tile's epilogue operands from inside the main loop (generator option "pf") vs the same sources without it:
# launch times (three interleaved repetitions) and workgroup 0's phase stamps.  (Record of a finished experiment: the option and its
# kernel-side plumbing were removed after it showed no effect -- commit 42bb9bd's parent tree has them; profiles/r3_gemm_epilogue_warmup_ab.txt.)  usage: tools/gpu/r3_gemm_warm.sh <outdir-name>
O=gpurun_out/$1; mkdir -p $O
P=tools/ablate/gemm_probe
{
for rep in 1 2 3; do
for v in "" _pf; do
  echo "{\"note\":\"variant=$v rep=$rep\"}"
  for s in "4608 3072 3072 0 1" "4608 3072 9216 3 1" "4608 3072 9216 3 2" "4608 3072 12288 2 1" "4608 12288 3072 0 1"; do
    set -- $s
    tr=""; [ $rep = 1 ] && tr="--trace"
    timeout 120 $P --lib tools/ablate/libsvdq_amd_probe$v.so --shape $1 $2 $3 --fuse $4 --geoms $5 $tr || echo "PROBE_FAILED $s rc=$?"
  done
done
done
} > $O/warm.jsonl 2> $O/warm.err
python3 - $O <<'PY'
import json,sys
for l in open(sys.argv[1]+'/warm.jsonl'):
    try: r=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    if 'note' in r: print(r['note']); continue
    if 'segments' in r:
        for s in r['segments'][1:4]:
            d=[]; prev=s[1]
            for x in s[2:]:
                if x>0: d.append((x-prev)/1e3); prev=x
                else: d.append(0)
            print("      loop %.1f | bias+lowrank %.1f  fuse-math %.1f  rest %.1f  stores %.1f kcyc" % ((s[1]-s[0])/1e3, d[0], d[1], d[2], d[3]))
    elif 'us' in r: print(f"M={r['M']:5d} K={r['K']:5d} N={r['N']:5d} fuse={r['fuse']} geo={r.get('geometry')} {r['us']:7.2f} us {r['TOPS']:7.1f} TOPS {r.get('wg_cycles',0)/1e3:7.1f} kcyc {r.get('eff_GHz',0):.3f} GHz")
PY
