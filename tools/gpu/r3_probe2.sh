#!/bin/bash
# round 3, call 2: where do the workgroups of the two-per-CU geometry sit and how long do they live (placement probe),
# whole-rounds grid vs every slot; then the tests touched since call 1.  usage: tools/gpu/r3_probe2.sh <outdir-name>
O=gpurun_out/$1; mkdir -p $O
P=tools/ablate/gemm_probe; PL=tools/ablate/libsvdq_amd_probe.so
{
for grid in "" full; do
  echo "{\"note\":\"SVDQ_PROBE_GRID=$grid\"}"
  for s in "4608 3072 3072 0" "4608 3072 9216 3" "4608 3072 12288 2" "4608 12288 3072 0"; do
    set -- $s
    SVDQ_PROBE_GRID=$grid timeout 120 $P --lib $PL --shape $1 $2 $3 --fuse $4 --geoms 1,2,3 --trace || echo "PROBE_FAILED $s rc=$?"
  done
done
} > $O/probe.jsonl 2> $O/probe.err
python3 - $O <<'PY'
import json,sys
for l in open(sys.argv[1]+'/probe.jsonl'):
    try: r=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    if 'note' in r: print(r['note']); continue
    if 'placement' in r: print("   ", r['placement']); continue
    if 'segments' in r:
        segs=r['segments'][:4]; print("    trace:", [[s[1]-s[0], s[5]-s[1]] for s in segs]); continue
    print(f"M={r['M']:5d} K={r['K']:5d} N={r['N']:5d} fuse={r['fuse']} geo={r['geometry']} {r['us']:7.2f} us {r['TOPS']:7.1f} TOPS {r['wg_cycles']/1e3:7.1f} kcyc {r['eff_GHz']:.3f} GHz")
PY
timeout 900 python -m pytest tests/test_gpu_geometry_determinism.py tests/test_gpu_qwenimage.py tests/test_gpu_loader.py tests/test_gpu_parity.py -m gpu -q > $O/pytest.txt 2>&1; tail -25 $O/pytest.txt
