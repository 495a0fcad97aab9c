#!/bin/bash
# round 6: phase stamps of the 128 x 64 wave tile kernel (probe library): per tile {loop call start, loop end, -, -, -, segment end} of workgroup 0, beside the
# 8-wave kernel's, with and without bias + low-rank operands; and the stand-alone loop probe (R = 0, no bias, its own simple schedule) on the same box.
O=gpurun_out/$1; mkdir -p $O
L=${2:-tools/ablate/libsvdq_amd_probe.so}
{
for s in "4608 3072 3072" "4608 12288 3072"; do set -- $s
  timeout 120 tools/ablate/gemm_probe --lib $L --shape $1 $2 $3 --R 32 --geoms 1,8 --iters 50 --trace
  timeout 120 tools/ablate/gemm_probe --lib $L --shape $1 $2 $3 --R 0 --no-bias --geoms 1,8 --iters 50 --trace
  timeout 120 tools/ablate/gemm128_probe_b2 --shape $1 $2 $3 --iters 50
done
} > $O/trace.jsonl 2> $O/trace.err
python3 - $O/trace.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    try: r=json.loads(l)
    except Exception: print(l[:200]); continue
    if 'segments' in r:
        segs=r['segments']; prev=None
        print(' trace geometry', r['trace_variant'])
        for s in segs[:8]:
            gap = (s[0]-prev) if prev is not None else s[0]
            if r['trace_variant'] == 8 and s[2]: print('   sched/gap %6d  loop %7d  bias+lowrank %5d  vmcnt wait %5d  convert+store %5d  tail %5d' % (gap, s[1]-s[0], s[2]-s[1], s[3]-s[2], s[4]-s[3], s[5]-s[4]))
            else: print('   sched/gap %6d  loop %7d  epilogue+tail %6d   (raw %s)' % (gap, s[1]-s[0], s[5]-s[1], s))
            prev=s[5]
    elif 'us' in r:
        print({k:r[k] for k in r if k in ('M','K','N','R','geometry','us','wg_cycles','eff_GHz','probe','product_us','cycles_per_tile_group_incl_epilogue')})
PY
tail -3 $O/trace.err
