#!/bin/bash
# in-kernel clock and phase stamps of the 4 x 64 attention kernel (probe builds in tools/ablate).  usage: r3_attn_probe.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
for lib in tools/ablate/libsvdq_amd_attn_probe*.so; do
  echo "== $lib"; SVDQ_LIB=$PWD/$lib PYTHONPATH=$PWD timeout 200 python tools/ablate/attn_probe.py 2>&1 | grep -v amdgpu.ids
done > $O/attn_probe.txt 2>&1
cat $O/attn_probe.txt
