#!/bin/bash
# round 4: stream-K partial tiles published with write-through stores.  parity (stream-K tests) + isolated A/B vs the round-3 library + trace.
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity_fullsize.py tests/test_gpu_geometry_determinism.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
P=tools/ablate/gemm_probe
for rep in 1 2 3; do for lib in tools/ablate/libsvdq_amd_r3.so nunchaku_amd/csrc/libsvdq_amd.so; do
  for s in "4608 12288 3072 0" "4096 12288 3072 0" "512 12288 3072 0" "1536 12288 3072 0"; do
    set -- $s
    echo -n "$lib " >> $O/iso_ab.txt
    timeout 120 $P --lib $lib --shape $1 $2 $3 --fuse $4 --geoms 0 2>>$O/iso_ab.err | grep '"us"' | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['M'],r['K'],r['N'],'fuse',r['fuse'],r['us'],'us',r['TOPS'],'TOPS sum',r['sum'])" >> $O/iso_ab.txt
  done; done; done
cat $O/iso_ab.txt
timeout 120 $P --lib tools/ablate/libsvdq_amd_probe.so --shape 4608 12288 3072 --fuse 0 --geoms 1 --trace > $O/trace.jsonl 2>$O/trace.err
python3 - $O <<'PY'
import json,sys
for l in open(sys.argv[1]+'/trace.jsonl'):
    try: r=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    if 'segments' in r:
        for s in r['segments'][:8]: print("   stamps kcyc", [round(x/1e3,1) for x in s])
    elif 'us' in r: print(f"M={r['M']} K={r['K']} N={r['N']} fuse={r['fuse']} {r['us']:.1f} us {r['TOPS']:.0f} TOPS wg_cycles {r['wg_cycles']:.0f} {r.get('eff_GHz',0):.3f} GHz")
    else: print(l.strip()[:300])
PY
