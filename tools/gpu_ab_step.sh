#!/bin/bash
# same-box A/B of the whole denoise step: the tree under gpurun_in/old (an earlier commit, built) vs this tree.  Prepare with
#   mkdir -p gpurun_in/old && git archive <rev> | tar -x -C gpurun_in/old && (cd gpurun_in/old && python -m nunchaku_amd.build)
# (gpurun_in/ is git-ignored but travels to the GPU box with the snapshot)
P='import sys,json; r=json.loads(sys.stdin.read()); ro=r["roofline"]; print("%.2f ms/step  gemm %.2f ms (frac %.3f)  quant %.2f  attn %s" % (r["ms_per_step"], ro["gemm_ms_per_step"], ro["frac"], ro["quantize"]["ms_per_step"], ro.get("attention",{}).get("ms_per_step")))'
for rep in 1 2; do
  echo -n "old: "; (cd gpurun_in/old && timeout 600 python bench.py --steps 20 --warmup 3 --no-prof 2>/dev/null | tail -1 | python -c "$P")
  echo -n "new: "; timeout 600 python bench.py --steps 20 --warmup 3 --no-prof 2>/dev/null | tail -1 | python -c "$P"
done
