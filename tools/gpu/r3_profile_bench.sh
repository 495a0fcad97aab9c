#!/bin/bash
# rocprofv3 evidence for profiles/ (round 3): kernel-trace stats of the bench command, fabric-side traffic and matrix-pipe
# utilisation of every gemm_w4a4 dispatch of the same command (separate --pmc passes, MI355X_MICROARCH.md "HBM" / "PMC slots").
# usage: tools/gpu/r3_profile_bench.sh <outdir-name> [bench args...]
O=gpurun_out/$1; shift; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $R/$O/trace.log 2>&1
for ctr in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $ctr | cut -d' ' -f1)
  rocprofv3 --pmc $ctr --output-format csv -d $R/$O/pmc_$tag -o g -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $R/$O/pmc_$tag.log 2>&1
done
cd $R
grep -h '"metric"' $O/trace.log | head -1 > $O/bench_line_under_trace.json
python - $O <<'PY'
import csv, glob, collections, json, sys, os, shutil
O=sys.argv[1]
for f in glob.glob(O+'/trace/**/*kernel_stats.csv', recursive=True):
    shutil.copy(f, O+'/bench_kernel_stats.csv')
    rows=list(csv.DictReader(open(f)))
    for r in rows[:10]: print({k:(v[:70] if k=='Name' else v) for k,v in r.items() if k in ('Name','Calls','AverageNs','Percentage')})
hbm={}; sq=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+'/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm_w4a4' not in r['Kernel_Name']: continue
        variant = r['Kernel_Name'].split('(')[0][-40:]
        sq[r['Counter_Name']][variant].append(float(r['Counter_Value']))
for c in ('FETCH_SIZE','WRITE_SIZE'):
    allv=[v for vs in sq.get(c,{}).values() for v in vs]
    if allv: hbm[c]={'avg_per_dispatch_KB':sum(allv)/len(allv),'dispatches':len(allv)}
json.dump(hbm, open(O+'/bench_gemm_hbm_counters.json','w'), indent=1)
util={'per_variant':{}}
names=set(k for c in sq for k in sq[c])
tot_m=tot_b=0
for v in sorted(names):
    g=lambda c: sum(sq[c][v])/max(len(sq[c][v]),1) if sq.get(c,{}).get(v) else None
    m,b=g('SQ_VALU_MFMA_BUSY_CYCLES'),g('SQ_BUSY_CYCLES')
    rec={c:g(c) for c in ('SQ_VALU_MFMA_BUSY_CYCLES','SQ_BUSY_CYCLES','SQ_INSTS_MFMA','SQ_INSTS_VALU','SQ_WAVE_CYCLES','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_ANY','GRBM_GUI_ACTIVE')}
    rec['dispatches']=len(sq.get('SQ_BUSY_CYCLES',{}).get(v,[]))
    util['per_variant'][v]=rec
    if m and b: tot_m+=sum(sq['SQ_VALU_MFMA_BUSY_CYCLES'][v]); tot_b+=sum(sq['SQ_BUSY_CYCLES'][v])
util['note']='raw per-dispatch averages over the gemm_w4a4 dispatches of one bench step (rocprofv3 --pmc, gfx950: counters summed over the SEs/XCDs as the tool reports them); mfma_busy_over_sq_busy is their ratio, not a calibrated utilisation'
util['mfma_busy_over_sq_busy']=tot_m/tot_b if tot_b else None
json.dump(util, open(O+'/bench_gemm_mfma_util.json','w'), indent=1)
print(json.dumps(hbm)); print('mfma_busy/sq_busy', util['mfma_busy_over_sq_busy'])
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/trace $O/pmc_*/
