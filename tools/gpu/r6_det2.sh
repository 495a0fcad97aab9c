for args in "--q32 --R2 32" "--q32 --R2 0" "--R2 32" "--R2 0"; do
  echo -n "fc1 $args: "; timeout 120 tools/ablate/gemm_probe --lib nunchaku_amd/csrc/libsvdq_amd.so --shape 4608 3072 12288 --fuse 2 --R 32 $args --geoms 0,1 --iters 50 | python3 -c "
import sys,json
for l in sys.stdin:
    try: r=json.loads(l)
    except Exception: continue
    if 'us' in r: print(r['us'], 'us', end=' | ')
print()"
done
