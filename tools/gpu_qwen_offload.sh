#!/bin/bash
# config 5 evidence: timing + a kernel / memory-copy trace of two offloaded forwards (overlap of H2D copies with kernels)
mkdir -p gpurun_out/qwen
python tools/bench_qwen_offload.py --layers 12 --steps 5 > gpurun_out/qwen/bench.log 2>&1; tail -1 gpurun_out/qwen/bench.log | cut -c1-1500
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/qwen/trace -o q -- python $R/tools/bench_qwen_offload.py --layers 6 --steps 2 > $R/gpurun_out/qwen/trace.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob
# overlap: fraction of the H2D copy time that lies inside some kernel's [start, end) interval
ks, cs = [], []
for f in glob.glob('gpurun_out/qwen/trace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)): ks.append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
for f in glob.glob('gpurun_out/qwen/trace/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'HOST_TO_DEVICE' in r.get('Direction','').upper() or 'H2D' in r.get('Direction','').upper(): cs.append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
ks.sort(); tot=0; ov=0
import bisect
starts=[k[0] for k in ks]
# merge kernel intervals
merged=[]
for s,e in ks:
    if merged and s<=merged[-1][1]: merged[-1][1]=max(merged[-1][1],e)
    else: merged.append([s,e])
ms=[m[0] for m in merged]
for s,e in cs:
    tot+=e-s
    i=max(0,bisect.bisect_right(ms,s)-1)
    while i<len(merged) and merged[i][0]<e:
        ov+=max(0,min(e,merged[i][1])-max(s,merged[i][0])); i+=1
# and the other way round: kernel time that runs while a copy is in flight
cm=[]
for s,e in sorted(cs):
    if cm and s<=cm[-1][1]: cm[-1][1]=max(cm[-1][1],e)
    else: cm.append([s,e])
cms=[m[0] for m in cm]; kt=0; kov=0
for s,e in ks:
    kt+=e-s
    i=max(0,bisect.bisect_right(cms,s)-1)
    while i<len(cm) and cm[i][0]<e:
        kov+=max(0,min(e,cm[i][1])-max(s,cm[i][0])); i+=1
open('gpurun_out/qwen/overlap.txt','w').write(f"H2D copies: {len(cs)}, total {tot/1e6:.2f} ms, of which {ov/1e6:.2f} ms ({100*ov/max(tot,1):.0f} %) overlap a running kernel\n"
    f"kernels: {len(ks)}, total {kt/1e6:.2f} ms, of which {kov/1e6:.2f} ms ({100*kov/max(kt,1):.0f} %) run while an H2D copy is in flight\n")
print(open('gpurun_out/qwen/overlap.txt').read())
PY
find gpurun_out/qwen/trace -name "*_trace.csv" -delete
ls gpurun_out/qwen/trace/*/ 2>/dev/null | head
