#!/bin/bash
# rocprofv3 evidence for profiles/ (round 4): kernel-trace stats of the bench command, fabric-side traffic and matrix-pipe utilisation / clock of
# every gemm_w4a4 dispatch of the same command (separate --pmc passes, MI355X_MICROARCH.md "HBM" / "PMC slots").  Every JSON is stamped with the
# hash of the kernel sources it was measured on (bench.kernel_sources_sha16): bench.py prints "stale" next to a number from other sources.
# usage: tools/gpu/r4_profile_bench.sh <outdir-name> [bench args...]
O=gpurun_out/$1; shift; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o bench -- python $R/bench.py --steps 3 --warmup 1 --prof-steps 1 --no-cpu-baseline "$@" > $R/$O/trace.log 2>&1
for ctr in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $ctr | cut -d' ' -f1)
  rocprofv3 --pmc $ctr --output-format csv -d $R/$O/pmc_$tag -o g -- python $R/bench.py --steps 1 --warmup 1 --prof-steps 1 --no-cpu-baseline "$@" > $R/$O/pmc_$tag.log 2>&1
done
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_GRBM -o g -- python $R/bench.py --steps 1 --warmup 1 --prof-steps 1 --no-cpu-baseline "$@" > $R/$O/pmc_GRBM.log 2>&1
cd $R
grep -h '"metric"' $O/trace.log | head -1 > $O/bench_line_under_trace.json
python - $O "$*" <<'PY'
import csv, glob, collections, json, sys, os, shutil
sys.path.insert(0, os.getcwd())
import bench
O, cmd = sys.argv[1], sys.argv[2]
sha = bench.kernel_sources_sha16()
for f in glob.glob(O+'/trace/**/*kernel_stats.csv', recursive=True):
    shutil.copy(f, O+'/bench_kernel_stats.csv')
    rows=list(csv.DictReader(open(f)))
    for r in rows[:10]: print({k:(v[:70] if k=='Name' else v) for k,v in r.items() if k in ('Name','Calls','AverageNs','Percentage')})
sq=collections.defaultdict(lambda: collections.defaultdict(list))
dur={}
for f in glob.glob(O+'/pmc_GRBM/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)): dur[r['Dispatch_Id']]=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
for f in glob.glob(O+'/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gemm_w4a4' not in r['Kernel_Name']: continue
        variant = r['Kernel_Name'].split('(')[0].replace('void ', '')
        sq[r['Counter_Name']][variant].append(float(r['Counter_Value']))
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE' and r['Dispatch_Id'] in dur: sq['us'][variant].append(dur[r['Dispatch_Id']])
hbm={'csrc_sha16': sha, 'command': 'python bench.py --steps 1 --warmup 1 --no-cpu-baseline ' + cmd,
     'source': 'tools/gpu/r5_profile_bench.sh (= r4_profile_bench.sh): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over every gemm_w4a4 dispatch of that command'}
for c in ('FETCH_SIZE','WRITE_SIZE'):
    allv=[v for vs in sq.get(c,{}).values() for v in vs]
    if allv: hbm[c]={'avg_per_dispatch_KB':sum(allv)/len(allv),'dispatches':len(allv)}
json.dump(hbm, open(O+'/bench_gemm_hbm_counters.json','w'), indent=1)
util={'csrc_sha16': sha, 'source': 'tools/gpu/r5_profile_bench.sh (= r4_profile_bench.sh): rocprofv3 --pmc passes over the gemm_w4a4 dispatches of `python bench.py --steps 1 --warmup 1 ' + cmd + '`',
      'definition': 'mfma_util = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs): fraction of the kernel time the matrix pipe of a SIMD is busy (32 cycles per 32x32 MFMA issued; half of them the FP6 product, half the 16-bit scale tile: one product MFMA per 64 cycles is the INT8-rate equivalent, so mfma_util x clock / 2.4 GHz ~ frac_int8).  clock_GHz = GRBM_GUI_ACTIVE / 8 / dispatch duration of the same dispatches (agrees with rocm-smi sclk: profiles/r4_clock_instrument.txt).',
      'per_variant': {}}
tm=tg=0
for v in sorted(set(k for c in sq for k in sq[c])):
    g=lambda c: sum(sq[c][v])/max(len(sq[c][v]),1) if sq.get(c,{}).get(v) else None
    m,gr,us=g('SQ_VALU_MFMA_BUSY_CYCLES'),g('GRBM_GUI_ACTIVE'),g('us')
    rec={'dispatches_in_the_command (1 warm-up + 1 timed + 1 event-bracketed + 3 single-class steps)':len(sq.get('GRBM_GUI_ACTIVE',{}).get(v,[])), 'avg_us_under_pmc':us}
    if m and gr:
        rec.update(mfma_busy_cycles_per_simd=m/1024, gui_active_cycles_per_xcd=gr/8, mfma_util=(m/1024)/(gr/8))
        tm+=sum(sq['SQ_VALU_MFMA_BUSY_CYCLES'][v])/1024; tg+=sum(sq['GRBM_GUI_ACTIVE'][v])/8
    if gr and us: rec['clock_GHz']=gr/8/us/1e3
    if g('SQ_INSTS_VALU') and g('SQ_INSTS_MFMA'): rec['valu_insts_per_mfma']=g('SQ_INSTS_VALU')/g('SQ_INSTS_MFMA')
    if g('SQ_WAVE_CYCLES'):
        rec['wave_cycles_split']={'active_issue':g('SQ_ACTIVE_INST_ANY')/g('SQ_WAVE_CYCLES'),'issue_stall(SQ_WAIT_INST_ANY)':g('SQ_WAIT_INST_ANY')/g('SQ_WAVE_CYCLES'),'parked(SQ_WAIT_ANY)':g('SQ_WAIT_ANY')/g('SQ_WAVE_CYCLES')}
    util['per_variant'][v]=rec
util['mfma_util']=tm/tg if tg else None
json.dump(util, open(O+'/bench_gemm_mfma_util.json','w'), indent=1)
print(json.dumps(hbm)); print(json.dumps(util, indent=1)[:3000])
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/trace $O/pmc_*/
