#!/bin/bash
# build tools/ablate/gemm128_probe[_<opts>] (round 6: the 128 x 64 wave tile probe and its timing ablations).  usage: tools/ablate/build128.sh [opts ...]   e.g. "" sp nodma rd1 nobar
cd "$(dirname "$0")/../.."
for o in "${@:-}"; do
  sfx=""; [ -n "$o" ] && sfx="_${o//+/_}"
  SVDQ_GEN3_OPTS="$o" python tools/gen_gemm_loop3.py >/dev/null || exit 1
  ns=3; [[ "$o" == *b2* ]] && ns=4
  /opt/rocm/bin/hipcc -DPROBE_NSTAGE=$ns --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Inunchaku_amd/csrc -Itools/ablate/gen \
    -DSVDQ_LOOP3_BF16="\"gemm_loop3_bf16$sfx.inc\"" -DSVDQ_LOOP3_FP16="\"gemm_loop3_fp16$sfx.inc\"" \
    -o tools/ablate/gemm128_probe$sfx tools/ablate/gemm128_probe.hip -ldl 2>/dev/null || { echo "build failed: $o"; exit 1; }
  echo "built tools/ablate/gemm128_probe$sfx"
done
