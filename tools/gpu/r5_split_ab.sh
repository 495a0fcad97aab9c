#!/bin/bash
# round 5: the split low-rank down projection (ABI 20) on ONE box: the parity tests that cover it, launch-level A/B of the rank-128 fc1 launches under geometry
# 1 (hybrid carry + atomics) / 6 (solo carry) / 7 (split) with the probe, then bench lines with the library's choice against --geometry 6.
# usage: r5_split_ab.sh <outdir> [no-tests]
O=gpurun_out/$1; mkdir -p $O
if [ -z "$2" ]; then
  T0=$(date +%s)
  timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "next_low_rank_down" > $O/pytest_small.txt 2>&1; tail -4 $O/pytest_small.txt
  timeout 1200 python -m pytest tests/test_gpu_parity_fullsize.py -q -k "other_ranks" > $O/pytest_full.txt 2>&1; tail -6 $O/pytest_full.txt
  timeout 600 python -m pytest tests/test_gpu_qwenimage.py tests/test_gpu_loader.py -q -k "odd_token or rank or capture" > $O/pytest_qwen.txt 2>&1; tail -3 $O/pytest_qwen.txt
  echo "pytest $(( $(date +%s) - T0 )) s"
fi
P=tools/ablate/gemm_probe; PL=tools/ablate/libsvdq_amd_probe.so
{
for c in "4608 128 128 0 1,6,7,0" "6400 128 128 256 1,6,7,0" "4608 48 48 0 0,7" "4608 64 64 0 0,7" "4608 160 160 0 1,7" "1536 128 128 0 0,6,7"; do
  set -- $c
  echo "{\"case\": \"fc1 M=$1 K=3072 N=12288 R=$2 R2=$3 split_rows=$4\"}"
  timeout 120 $P --lib $PL --shape $1 3072 12288 --fuse 2 --R $2 --R2 $3 --split $4 --geoms $5 --trace || echo "PROBE_FAILED $c rc=$?"
done
} > $O/trace.jsonl 2> $O/trace.err
python3 - $O <<'PY'
import json,sys
for l in open(sys.argv[1]+'/trace.jsonl'):
    try: r=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    if 'case' in r: print(r['case'])
    elif 'segments' in r:
        s=r['segments'][1] if len(r['segments'])>1 else r['segments'][0]
        d=[]; prev=s[1]
        for x in s[2:]:
            if x>0: d.append((x-prev)/1e3); prev=x
            else: d.append(0)
        print("     geometry %s: loop %.1f | bias+lowrank %.1f  fuse-math %.1f  lowrank-down %.1f  stores %.1f kcyc" % (r.get('trace_variant'), (s[1]-s[0])/1e3, d[0], d[1], d[2], d[3]))
    elif 'us' in r: print(f"  geo={r.get('geometry')} {r['us']:.1f} us {r['TOPS']:.0f} TOPS {r.get('eff_GHz',0):.3f} GHz")
PY
run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline --steps 10 --warmup 2 --prof-steps 4 "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python3 - $O/bench_$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r=d['roofline']
    print(sys.argv[1].split('/')[-1], 'ms/step', round(d['ms_per_step'],2), 'frac', round(r['frac'],4), 'gemm', round(r['gemm_ms_per_step'],2), 'quant', round(r['quantize']['ms_per_step'],2), 'attn', round(r['attention']['ms_per_step'],2), 'clock', r.get('effective_clock_ghz'), {k:(round(v['avg_launch_us'],1), round(v['frac'],3)) for k,v in r['per_variant'].items()})
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
}
run dev1024_r32
run dev1024_r128 --rank 128
run dev1024_r128_solo --rank 128 --geometry 6
run dev1024_lora16 --lora 16
run dev1024_lora16_split --lora 16 --geometry 7
run qwen1664x928_r128 --config qwen1024 --resolution 1664 928 --txt-tokens 37 --rank 128
run qwen1664x928_r128_solo --config qwen1024 --resolution 1664 928 --txt-tokens 37 --rank 128 --geometry 6
