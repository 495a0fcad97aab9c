#!/bin/bash
# round 6: the carry kernel's lora_act_in rows requested above the main loop instead of at the top of the epilogue: fc1 launch against the library before (gpurun_in/prev)
for rep in 1 2 3 4; do
  for l in gpurun_in/prev/libsvdq_amd.so nunchaku_amd/csrc/libsvdq_amd.so; do
    echo -n "fc1 $(echo $l | cut -c1-14): "; timeout 120 tools/ablate/gemm_probe --lib $l --shape 4608 3072 12288 --fuse 2 --R 32 --R2 32 --geoms 0 --iters 60 | python3 -c "
import sys,json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: continue
    if 'us' in r: print(r['us'], 'us', r['sum'])"
  done
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_fullsize.py -m gpu -q -x -k "gelu or fc1 or mlp or block" 2>&1 | tail -3
