#!/bin/bash
# round 3, call 3: dynamic per-XCD tile queues of the 128 x 128 geometry.  usage: tools/gpu/r3_dyn.sh <outdir-name>
O=gpurun_out/$1; mkdir -p $O
P=tools/ablate/gemm_probe; PL=tools/ablate/libsvdq_amd_probe.so
{
timeout 60 $P --lib $PL --shape 4608 384 3072 --geoms 1,2,3,4,5 --warm 10 --iters 5 || echo "PROBE_SMALL_FAILED rc=$?"
for s in "4608 3072 3072 0" "4608 3072 9216 3" "4608 3072 12288 2" "4608 12288 3072 0" "1536 3072 9216 3" "1536 3072 12288 2" "1536 3072 3072 0"; do
  set -- $s
  timeout 120 $P --lib $PL --shape $1 $2 $3 --fuse $4 --geoms 1,2,3,4 --trace || echo "PROBE_FAILED $s rc=$?"
done
timeout 120 $P --lib $PL --shape 4608 3072 9216 --fuse 3 --split 512 --geoms 1,2 || echo "PROBE_FAILED grouped qkv"
timeout 120 $P --lib $PL --shape 4608 3072 12288 --fuse 2 --split 512 --geoms 1,2 || echo "PROBE_FAILED grouped fc1"
} > $O/probe.jsonl 2> $O/probe.err
python3 - $O <<'PY'
import json,sys
for l in open(sys.argv[1]+'/probe.jsonl'):
    try: r=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    if 'placement' in r: pl=r['placement']; continue
    if 'segments' in r: tr=[[s[1]-s[0], s[5]-s[1]] for s in r['segments'][:3]]; continue
    print(f"M={r['M']:5d} K={r['K']:5d} N={r['N']:5d} fuse={r['fuse']} split={r['split']:4d} geo={r['geometry']} {r['us']:7.2f} us {r['TOPS']:7.1f} TOPS {r['eff_GHz']:.3f} GHz sum={r['sum'][:8]} | wgs {pl['wgs']} life avg {pl['life_us_paired'] or pl['life_us_alone']:.0f} min {pl['life_us_min']:.0f} max {pl['life_us_max']:.0f} span {pl['span_us']:.0f} | {tr if 'tr' in dir() else ''}")
PY
timeout 900 python -m pytest tests/test_gpu_geometry_determinism.py tests/test_gpu_parity_fullsize.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
for g in 1 2; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --geometry $g > $O/bench_g$g.json 2> $O/bench_g$g.err; python3 -c "
import json,sys
try:
    d=json.loads(open('$O/bench_g$g.json').read().strip().splitlines()[-1]); print('bench geometry $g:', round(d['ms_per_step'],2),'ms/step frac', round(d['roofline']['frac'],3), 'gemm ms', round(d['roofline']['gemm_ms_per_step'],2))
except Exception as e: print('bench $g failed', e)
"; done
