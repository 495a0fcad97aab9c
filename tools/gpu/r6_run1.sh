#!/bin/bash
# round 6, call 1: (a) the two parity loose ends (attention fused quantiser bit equality, SiLU envelope); (b) barrier-share timing probes (racy loops: timing only);
# (c) fc1 low-rank-down phase ablation; (d) stream-K collect A/B old vs new product library; (e) same-box step A/B.   usage: tools/gpu/r6_run1.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
P=tools/ablate/gemm_probe
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_gpu_parity.py -m gpu -q -x -k "fused_output_quantiser or silu or geometry" > $O/pytest_loose_ends.txt 2>&1; tail -5 $O/pytest_loose_ends.txt
{
echo "## barrier share (probe builds; bar*/nobar are racy: timing only)"
for s in "4608 3072 3072 0" "4608 12288 3072 0" "4608 3072 12288 2"; do
  set -- $s
  for v in "" _bar2of3 _bar1of3 _nobar; do
    echo "{\"case\": \"lib=probe$v M=$1 K=$2 N=$3 fuse=$4\"}"
    timeout 120 $P --lib tools/ablate/libsvdq_amd_probe$v.so --shape $1 $2 $3 --fuse $4 --R 32 --R2 32 --geoms 1 --iters 50 || echo "PROBE_FAILED rc=$?"
  done
done
echo "## fc1 low-rank-down ablation (SVDQ_PROBE_OFF: 1 = no carry/atomics, 4 = no low-rank-down block, 2 = no code stores)"
for off in 0 1 4 2; do
  echo "{\"case\": \"fc1 PROBE_OFF=$off\"}"
  SVDQ_PROBE_OFF=$off timeout 120 $P --lib tools/ablate/libsvdq_amd_probe.so --shape 4608 3072 12288 --fuse 2 --R 32 --R2 32 --geoms 0 --trace --iters 50 || echo "PROBE_FAILED rc=$?"
done
echo "## stream-K collect: old vs new product library, fc2 shape"
for rep in 1 2 3; do
  for l in gpurun_in/old/nunchaku_amd/csrc/libsvdq_amd.so nunchaku_amd/csrc/libsvdq_amd.so; do
    echo "{\"case\": \"lib=$l fc2\"}"
    timeout 120 $P --lib $l --shape 4608 12288 3072 --fuse 0 --R 32 --geoms 0 --iters 100 || echo "PROBE_FAILED rc=$?"
  done
done
} > $O/probes.jsonl 2> $O/probes.err
python3 - $O <<'PY'
import json,sys
for l in open(sys.argv[1]+'/probes.jsonl'):
    l=l.strip()
    if l.startswith('#'): print(l); continue
    try: r=json.loads(l)
    except Exception: print(l[:200]); continue
    if 'case' in r: print(r['case'])
    elif 'segments' in r:
        for s in r['segments'][:3]:
            d=[]; prev=s[1]
            for x in s[2:]:
                if x>0: d.append((x-prev)/1e3); prev=x
                else: d.append(0)
            print("     loop %.1f | bias+lowrank %.1f  fuse-math %.1f  lowrank-down %.1f  stores %.1f kcyc" % ((s[1]-s[0])/1e3, d[0], d[1], d[2], d[3]))
    elif 'us' in r: print(f"  geo={r.get('geometry')} {r['us']:.1f} us {r['TOPS']:.0f} TOPS {r.get('eff_GHz',0):.3f} GHz")
PY
bash tools/gpu_ab_step.sh > $O/step_ab.txt 2>&1; cat $O/step_ab.txt
