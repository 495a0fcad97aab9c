#!/bin/bash
# round 6: the 128 x 64 probe's variants (sp = DMAs spread over the tile-group slots; nodma / rd1 / nobar = timing-only ablations: garbage results).  usage: r6_gemm128_b.sh <outdir> [variants...]
O=gpurun_out/$1; shift; mkdir -p $O
V="${@:-base sp nodma rd1 nobar sp_rd1}"
{
for v in $V; do
  b=tools/ablate/gemm128_probe; [ "$v" != base ] && b=${b}_$v
  for s in "4608 3072 3072" "4608 12288 3072"; do
    set -- $s
    echo "{\"variant\":\"$v\"}"
    timeout 120 $b --shape $1 $2 $3 --iters 50 || echo "{\"rc\":$?}"
  done
done
} > $O/gemm128.jsonl 2> $O/gemm128.err
python3 - $O/gemm128.jsonl <<'PY'
import json,sys
v=None
for l in open(sys.argv[1]):
    try: r=json.loads(l)
    except Exception: print(l.strip()[:100]); continue
    if 'variant' in r: v=r['variant']
    elif 'us' in r: print(f"{v:8s} K={r['K']:5d} {r['us']:7.2f} us ({r['product_us']:.2f} product, ratio {r['ratio']:.3f}) {r['eff_GHz']:.3f} GHz  {r['cycles_per_tile_group_incl_epilogue']:.1f} cyc/tile-group  mismatches {r['mismatches']}")
    else: print(v, r)
PY
tail -3 $O/gemm128.err
