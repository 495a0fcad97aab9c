#!/bin/bash
# round 6: rank 128 -- the cached fragment images of the weight-side low-rank operands (ABI 21) against the per-launch packs, same box, interleaved; its test
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_attention.py tests/test_gpu_parity_fullsize.py -m gpu -q -x -k "cached_fragment or fused_output_quantiser or other_ranks or by_rank" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
P='import sys,json; r=json.loads(sys.stdin.read().strip().split("\n")[-1]); ro=r["roofline"]; print("%.2f ms/step  gemm %.2f  quant %.2f  attn %.2f" % (r["ms_per_step"], ro["gemm_ms_per_step"], ro["quantize"]["ms_per_step"], ro["attention"]["ms_per_step"]))'
for rep in 1 2; do
  echo -n "rank 32            : "; timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$P"
  echo -n "rank 128 per-launch: "; timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --rank 128 --no-fragment-cache 2>/dev/null | python -c "$P"
  echo -n "rank 128 cached    : "; timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --rank 128 2>/dev/null | python -c "$P"
done 2>&1 | tee $O/r128_ab.txt
