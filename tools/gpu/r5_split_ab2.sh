#!/bin/bash
# round 5: second A/B of the split low-rank down projection on ONE box: attention epilogue split on / off, the per-kernel times of a rank-128 step (rocprofv3).
# usage: r5_split_ab2.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_attention.py -q -x -k "fused_output_quantiser" > $O/pytest_attn.txt 2>&1; tail -4 $O/pytest_attn.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "next_low_rank_down" > $O/pytest_small.txt 2>&1; tail -3 $O/pytest_small.txt
timeout 1200 python -m pytest tests/test_gpu_parity_fullsize.py -q -k "other_ranks" > $O/pytest_full.txt 2>&1; tail -4 $O/pytest_full.txt
timeout 900 python -m pytest tests/test_gpu_qwenimage.py tests/test_gpu_loader.py tests/test_gpu_geometry_determinism.py -q > $O/pytest_models.txt 2>&1; tail -4 $O/pytest_models.txt
echo "pytest $(( $(date +%s) - T0 )) s"
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
run dev1024_r128_attn_passes --rank 128 --no-attention-split
run dev1024_lora16 --lora 16
run dev1024_lora16_attn_passes --lora 16 --no-attention-split
run qwen1664x928_r128 --config qwen1024 --resolution 1664 928 --txt-tokens 37 --rank 128
run qwen1664x928_r128_attn_passes --config qwen1024 --resolution 1664 928 --txt-tokens 37 --rank 128 --no-attention-split
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_dev -o bench -- python $R/bench.py --rank 128 --steps 3 --warmup 1 --prof-steps 1 --no-cpu-baseline > $R/$O/trace_dev.log 2>&1
cd $R
f=$(find $O/trace_dev -name "*kernel_stats.csv" | head -1); cp "$f" $O/bench_kernel_stats_r128_dev.csv; head -14 $O/bench_kernel_stats_r128_dev.csv | cut -c1-170; grep "lowrank_down_split\|pack_lora" $O/bench_kernel_stats_r128_dev.csv | cut -c1-200
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/trace_dev
