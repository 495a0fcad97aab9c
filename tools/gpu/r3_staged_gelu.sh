#!/bin/bash
# staged epilogue, GELU_QUANT variant without the early drain: its parity tests, then the fc1 launch against the previous build (same box)
O=gpurun_out/$1; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_fullsize.py tests/test_gpu_fused_norm.py tests/test_gpu_geometry_determinism.py -m gpu -x -q -k "gelu or mlp or fused or fc1 or determin" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
P=tools/ablate/gemm_probe
for rep in 1 2 3; do for lib in tools/ablate/libsvdq_amd_prev.so nunchaku_amd/csrc/libsvdq_amd.so; do
  timeout 60 $P --lib $lib --shape 4608 3072 12288 --fuse 2 --geoms 1 | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$lib', round(r['us'],2), 'us', r.get('sum'))"
done; done
