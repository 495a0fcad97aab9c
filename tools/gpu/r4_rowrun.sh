#!/bin/bash
# round 4: the low-rank-down carry + row-run schedule of the GELU_QUANT launch.  parity tests, isolated A/B against the round-3 library, tile trace, step A/B.
# usage: r4_rowrun.sh <outdir> [pytest -k expression]
O=gpurun_out/$1; mkdir -p $O
K=${2:-"gelu or mlp or fused or determinis"}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_fullsize.py tests/test_gpu_geometry_determinism.py tests/test_gpu_fused_norm.py -m gpu -q -x -k "$K" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
P=tools/ablate/gemm_probe
for rep in 1 2 3; do for lib in tools/ablate/libsvdq_amd_r3.so nunchaku_amd/csrc/libsvdq_amd.so; do
  for s in "4608 3072 12288 2" "4096 3072 12288 2" "1536 3072 12288 2"; do
    set -- $s
    echo -n "$lib " >> $O/iso_ab.txt
    timeout 120 $P --lib $lib --shape $1 $2 $3 --fuse $4 --geoms 0 2>>$O/iso_ab.err | grep '"us"' | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['M'],r['K'],r['N'],'fuse',r['fuse'],r['us'],'us',r['TOPS'],'TOPS sum',r['sum'])" >> $O/iso_ab.txt
  done; done; done
cat $O/iso_ab.txt
timeout 120 $P --lib tools/ablate/libsvdq_amd_probe.so --shape 4608 3072 12288 --fuse 2 --geoms 1 --trace > $O/trace.jsonl 2>$O/trace.err
python3 - $O <<'PY'
import json,sys
for l in open(sys.argv[1]+'/trace.jsonl'):
    try: r=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    if 'segments' in r:
        for s in r['segments'][:8]:
            d=[]; prev=s[1]
            for x in s[2:]:
                if x>0: d.append((x-prev)/1e3); prev=x
                else: d.append(0)
            print("   loop %.1f | bias+lowrank %.1f  fuse-math %.1f  lowrank-down %.1f  stores+flush %.1f kcyc" % ((s[1]-s[0])/1e3, d[0], d[1], d[2], d[3]))
    elif 'us' in r: print(f"M={r['M']} K={r['K']} N={r['N']} fuse={r['fuse']} {r['us']:.1f} us {r['TOPS']:.0f} TOPS wg_cycles {r['wg_cycles']:.0f} {r.get('eff_GHz',0):.3f} GHz")
    else: print(l.strip()[:300])
PY
bash tools/gpu/r3_step_ab.sh $1 tools/ablate/libsvdq_amd_r3.so nunchaku_amd/csrc/libsvdq_amd.so
