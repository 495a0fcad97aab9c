O=gpurun_out/gelu_carry2; mkdir -p $O; rm -f $O/a.jsonl
P=tools/ablate/gemm_probe; AB=tools/ablate/libsvdq_amd_ablate.so
for rep in 1 2; do
for r in 0 16 2048 2064 64; do
$P --lib $AB --shape 4608 3072 12288 --fuse 2 --variants 0 --reserved $r >> $O/a.jsonl 2>&1
done; done
python3 - <<'PY'
import json
for l in open('gpurun_out/gelu_carry2/a.jsonl'):
    try: r=json.loads(l)
    except Exception: print(l.strip()); continue
    print(f"M={r['M']} reserved={r['reserved']:5d}  {r['us']:7.2f} us {r['TOPS']:7.1f} TOPS {r['wg_cycles']/1e3:7.1f} kcyc {r['eff_GHz']:.3f} GHz")
PY
