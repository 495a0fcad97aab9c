#!/bin/bash
# round 3, call 4: XCD-contiguous tile walk A/B (time + fabric traffic), auto geometry, bench.  usage: tools/gpu/r3_walk.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
R=$PWD; P=$R/tools/ablate/gemm_probe; PL=$R/tools/ablate/libsvdq_amd_probe.so
export TMPDIR=/tmp
{
for walk in "" 1; do
  echo "{\"note\":\"SVDQ_PROBE_NOWALK=$walk\"}"
  for s in "4608 3072 3072 0" "4608 3072 9216 3" "4608 3072 12288 2" "4608 12288 3072 0"; do
    set -- $s
    env ${walk:+SVDQ_PROBE_NOWALK=1} timeout 120 $P --lib $PL --shape $1 $2 $3 --fuse $4 --geoms 0,1,2 || echo "PROBE_FAILED $s rc=$?"
  done
done
} > $O/probe.jsonl 2> $O/probe.err
python3 - $O <<'PY'
import json,sys
for l in open(sys.argv[1]+'/probe.jsonl'):
    try: r=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    if 'note' in r: print(r['note']); continue
    if 'placement' in r: pl=r['placement']; continue
    if 'segments' in r: continue
    print(f"M={r['M']:5d} K={r['K']:5d} N={r['N']:5d} fuse={r['fuse']} geo={r['geometry']} {r['us']:7.2f} us {r['TOPS']:7.1f} TOPS {r['eff_GHz']:.3f} GHz sum={r['sum'][:8]} | wgs {pl['wgs']} life avg {pl['life_us_paired'] or pl['life_us_alone']:.0f} span {pl['span_us']:.0f}")
PY
# fabric-side traffic per launch (separate PMC passes), walk on / off, geometry 1 and 2, QKV and fc1 shapes
cd /tmp
for walk in on off; do for ctr in FETCH_SIZE WRITE_SIZE; do for shp in "4608 3072 9216 3" "4608 3072 12288 2" "4608 12288 3072 0"; do
  set -- $shp
  for g in 1 2; do
    [ "$g" = 2 ] && [ "$2" = 12288 ] && continue
    d=$R/$O/pmc_${walk}_${ctr}_N$3_g$g
    env $( [ $walk = off ] && echo SVDQ_PROBE_NOWALK=1 ) timeout 120 rocprofv3 --pmc $ctr --output-format csv -d $d -o p -- $P --lib $PL --shape $1 $2 $3 --fuse $4 --geoms $g --warm 20 --iters 10 > /dev/null 2>&1
  done
done; done; done
cd $R
python3 - $O <<'PY'
import csv,glob,sys,collections,os,json
out={}
for d in sorted(glob.glob(sys.argv[1]+'/pmc_*')):
    vals=[]
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'gemm_w4a4' in r['Kernel_Name']: vals.append(float(r['Counter_Value']))
    if vals: out[os.path.basename(d)]={'avg_KB':sum(vals)/len(vals),'n':len(vals)}
json.dump(out,open(sys.argv[1]+'/pmc_summary.json','w'),indent=1)
for k,v in out.items(): print(k, round(v['avg_KB']/1024,1),'MB/launch', v['n'])
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
timeout 600 python -m pytest tests/test_gpu_geometry_determinism.py tests/test_gpu_parity_fullsize.py tests/test_gpu_fused_norm.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for g in 1 0; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --geometry $g > $O/bench_g$g.json 2> $O/bench_g$g.err; python3 -c "
import json,sys
try:
    d=json.loads(open('$O/bench_g$g.json').read().strip().splitlines()[-1]); print('bench geometry $g:', round(d['ms_per_step'],2),'ms/step frac', round(d['roofline']['frac'],3), 'gemm ms', round(d['roofline']['gemm_ms_per_step'],2))
except Exception as e: print('bench $g failed', e)
"; done
