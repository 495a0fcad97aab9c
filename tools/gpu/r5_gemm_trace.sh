#!/bin/bash
# round 5: phase stamps of workgroup 0 (probe library, tools/ablate) for the production launches at rank 32 and rank 128.  usage: tools/gpu/r5_gemm_trace.sh <outdir-name>
O=gpurun_out/$1; mkdir -p $O
P=tools/ablate/gemm_probe; PL=tools/ablate/libsvdq_amd_probe.so
{
for R in 32 128; do
for s in "4608 3072 9216 3" "4608 3072 12288 2" "4608 3072 3072 0" "4608 12288 3072 0"; do
  set -- $s
  echo "{\"case\": \"R=$R M=$1 K=$2 N=$3 fuse=$4\"}"
  timeout 120 $P --lib $PL --shape $1 $2 $3 --fuse $4 --R $R --R2 $R --geoms 0,1,2 --trace || echo "PROBE_FAILED $s rc=$?"
done
done
} > $O/trace.jsonl 2> $O/trace.err
python3 - $O <<'PY'
import json,sys
for l in open(sys.argv[1]+'/trace.jsonl'):
    try: r=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    if 'case' in r: print(r['case'])
    elif 'segments' in r:
        print("  trace of geometry", r.get('trace_variant'))
        for s in r['segments'][:4]:
            d=[]; prev=s[1]
            for x in s[2:]:
                if x>0: d.append((x-prev)/1e3); prev=x
                else: d.append(0)
            print("     loop %.1f | bias+lowrank %.1f  fuse-math %.1f  lowrank-down %.1f  stores %.1f kcyc" % ((s[1]-s[0])/1e3, d[0], d[1], d[2], d[3]))
    elif 'us' in r: print(f"  geo={r.get('geometry')} {r['us']:.1f} us {r['TOPS']:.0f} TOPS {r.get('eff_GHz',0):.3f} GHz")
PY
