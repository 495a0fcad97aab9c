#!/bin/bash
# round 3, call 1: the two-workgroups-per-CU GEMM geometry -- checksums across geometries, timing of the FLUX shapes (probe
# library: clocks + phase stamps), then the parity suites.  usage: tools/gpu/r3_geometry.sh <outdir-name>
O=gpurun_out/$1; mkdir -p $O
P=tools/ablate/gemm_probe; PL=tools/ablate/libsvdq_amd_probe.so
rocminfo | grep -E "Compute Unit|Max Clock|Marketing" | head -6 > $O/rocminfo.txt 2>&1
{
# quick life sign first: small shape, all geometries (a hang here must not take the whole call)
timeout 60 $P --lib $PL --shape 512 384 768 --geoms 1,2,3 --warm 10 --iters 5 || echo "PROBE_SMALL_FAILED rc=$?"
for s in "4608 3072 3072 0" "4608 3072 9216 3" "4608 3072 12288 2" "4608 12288 3072 0"; do
  set -- $s
  timeout 120 $P --lib $PL --shape $1 $2 $3 --fuse $4 --geoms 1,2,3 --trace || echo "PROBE_FAILED $s rc=$?"
done
# grouped launches as the model issues them
timeout 120 $P --lib $PL --shape 4608 3072 9216 --fuse 3 --split 512 --geoms 1,2 || echo "PROBE_FAILED grouped qkv"
timeout 120 $P --lib $PL --shape 4608 3072 12288 --fuse 2 --split 512 --geoms 1,2 || echo "PROBE_FAILED grouped fc1"
timeout 120 $P --lib $PL --shape 4608 12288 3072 --split 512 --geoms 1,2 || echo "PROBE_FAILED grouped fc2"
# small-M launches (schnell / text-only)
timeout 120 $P --lib $PL --shape 1536 3072 9216 --fuse 3 --geoms 1,2 || echo "PROBE_FAILED 1536"
timeout 120 $P --lib $PL --shape 1536 12288 3072 --geoms 1,2 || echo "PROBE_FAILED 1536 fc2"
# deterministic format on the consumer side
timeout 120 $P --lib $PL --shape 4608 3072 9216 --fuse 3 --geoms 1,2 --q32 || echo "PROBE_FAILED q32"
} > $O/probe.jsonl 2> $O/probe.err
python3 - $O <<'PY'
import json,sys
for l in open(sys.argv[1]+'/probe.jsonl'):
    try: r=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    if 'segments' in r:
        segs=r['segments'][:3]; print("   trace geo-run:", [[s[1]-s[0], s[5]-s[1]] for s in segs], "(loop cycles, epilogue cycles) of the first tiles"); continue
    print(f"M={r['M']:5d} K={r['K']:5d} N={r['N']:5d} fuse={r['fuse']} split={r['split']:4d} geo={r['geometry']} {r['us']:7.2f} us {r['TOPS']:7.1f} TOPS {r['wg_cycles']/1e3:7.1f} kcyc {r['eff_GHz']:.3f} GHz sum={r['sum']}")
PY
timeout 900 python -m pytest tests/test_gpu_geometry_determinism.py tests/test_gpu_parity.py tests/test_gpu_parity_fullsize.py -m gpu -x -q > $O/pytest_a.txt 2>&1; tail -15 $O/pytest_a.txt
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_all.txt 2>&1; tail -15 $O/pytest_all.txt
for g in 1 2; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --geometry $g > $O/bench_g$g.json 2> $O/bench_g$g.err; python3 -c "
import json,sys
try:
    d=json.loads(open('$O/bench_g$g.json').read().strip().splitlines()[-1]); print('bench geometry $g:', round(d['ms_per_step'],2),'ms/step frac', round(d['roofline']['frac'],3), 'gemm ms', round(d['roofline']['gemm_ms_per_step'],2))
except Exception as e: print('bench $g failed', e)
"; done
