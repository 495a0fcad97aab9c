#!/bin/bash
# round 3: RMSNorm+RoPE epilogue trimmed (fused multiply-adds, the store's conversion as the only rounding, unpredicated V^T stores on full
# tiles): parity tests, then the QKV launch against the previous build on one box.  usage: tools/gpu/r3_qkv_epi.sh <outdir-name>
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_fullsize.py tests/test_gpu_attention.py tests/test_gpu_geometry_determinism.py -m gpu -q -k "rope or qkv or norm or attention or geometr" > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
P=tools/ablate/gemm_probe
{
for rep in 1 2 3; do
for lib in tools/ablate/libsvdq_amd_prev.so nunchaku_amd/csrc/libsvdq_amd.so; do
  echo "{\"note\":\"lib=$lib rep=$rep\"}"
  for s in "4608 3072 9216 3 1" "4608 3072 9216 3 2" "512 3072 9216 3 1"; do
    set -- $s
    timeout 120 $P --lib $lib --shape $1 $2 $3 --fuse $4 --geoms $5 || echo "PROBE_FAILED $s rc=$?"
  done
done
done
} > $O/qkv.jsonl 2> $O/qkv.err
python3 - $O <<'PY'
import json,sys
for l in open(sys.argv[1]+'/qkv.jsonl'):
    try: r=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    if 'note' in r: print(r['note']); continue
    if 'us' in r: print(f"M={r['M']:5d} K={r['K']:5d} N={r['N']:5d} fuse={r['fuse']} geo={r.get('geometry')} {r['us']:7.2f} us {r['TOPS']:7.1f} TOPS")
PY
tail -3 $O/qkv.err
