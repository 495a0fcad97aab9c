#!/bin/bash
# round 4, call 1: (a) register-placement probe, (b) what the fc1 / GELU_QUANT tile's extra loop cycles are (probe library, SVDQ_PROBE_OFF bits),
# (c) ONE clock instrument: dispatch duration + GRBM_GUI_ACTIVE + SQ counters + in-kernel s_memtime / s_memrealtime + rocm-smi under a sustained loop,
# for the pure-MFMA filler loop, the GEMM's register-only arithmetic and the four production GEMM launches.   usage: r4_probe1.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
R=$PWD
P=tools/ablate/gemm_probe; PL=tools/ablate/libsvdq_amd_probe.so
timeout 120 tools/ablate/bank_probe > $O/bank_probe.txt 2>&1; cat $O/bank_probe.txt
echo "== fc1 ablation"
for off in 0 1 2 3 4 6; do
  echo "# SVDQ_PROBE_OFF=$off" >> $O/fc1_ablation.jsonl
  SVDQ_PROBE_OFF=$off timeout 120 $P --lib $PL --shape 4608 3072 12288 --fuse 2 --geoms 1 --trace >> $O/fc1_ablation.jsonl 2>> $O/fc1_ablation.err
done
python3 - $O <<'PY'
import json,sys
for l in open(sys.argv[1]+'/fc1_ablation.jsonl'):
    if l.startswith('#'): print(l.strip()); continue
    try: r=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    if 'segments' in r:
        for s in r['segments'][:7]:
            d=[]; prev=s[1]
            for x in s[2:]:
                if x>0: d.append((x-prev)/1e3); prev=x
                else: d.append(0)
            print("   loop %.1f | bias+lowrank %.1f  fuse-math %.1f  lowrank-down %.1f  stores %.1f kcyc" % ((s[1]-s[0])/1e3, d[0], d[1], d[2], d[3]))
    elif 'us' in r: print(f"M={r['M']} K={r['K']} N={r['N']} fuse={r['fuse']} {r['us']:.1f} us {r['TOPS']:.0f} TOPS wg_cycles {r['wg_cycles']:.0f} {r.get('eff_GHz',0):.3f} GHz")
PY
echo "== clock instrument"
smi() { # sample sclk / power every 0.25 s while the command runs
  ( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > $2 &
  local pid=$!
  eval "$1"
  kill $pid 2>/dev/null; wait $pid 2>/dev/null
}
smi "timeout 60 tools/ablate/filler_probe sustain 4 mfma > $O/sustain_mfma.json" $O/smi_mfma.txt
smi "timeout 60 tools/ablate/filler_probe sustain 4 gemm > $O/sustain_gemmarith.json" $O/smi_gemmarith.txt
for s in "4608 3072 12288 2" "4608 3072 9216 3" "4608 3072 3072 0" "4608 12288 3072 0"; do
  set -- $s
  smi "timeout 60 $P --lib $PL --shape $1 $2 $3 --fuse $4 --geoms 0 --sustain 3 > $O/sustain_gemm_$3_$4.json" $O/smi_gemm_$3_$4.txt
done
cat $O/sustain_*.json
export TMPDIR=/tmp
cd /tmp
CTR="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
timeout 200 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $R/$O/pmc_filler -o g -- $R/tools/ablate/filler_probe sustain 0.3 mfma > $R/$O/pmc_filler.log 2>&1
timeout 200 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $R/$O/pmc_fillerg -o g -- $R/tools/ablate/filler_probe sustain 0.3 gemm > $R/$O/pmc_fillerg.log 2>&1
for s in "4608 3072 12288 2" "4608 3072 9216 3" "4608 3072 3072 0" "4608 12288 3072 0"; do
  set -- $s
  timeout 200 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $R/$O/pmc_gemm_$3_$4 -o g -- $R/$P --lib $R/nunchaku_amd/csrc/libsvdq_amd.so --shape $1 $2 $3 --fuse $4 --geoms 0 --iters 20 --warm 20 > $R/$O/pmc_gemm_$3_$4.log 2>&1
done
cd $R
python3 - $O <<'PY'
import csv, glob, collections, json, sys, os, re
O=sys.argv[1]
res={}
for d in sorted(glob.glob(O+'/pmc_*/')):
    tag=os.path.basename(d.rstrip('/'))[4:]
    dur={}
    for f in glob.glob(d+'/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)): dur[r['Dispatch_Id']]=((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r['Kernel_Name'])
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Dispatch_Id'] not in dur: continue
            k=r['Kernel_Name'].split('(')[0][-60:]
            acc[k][r['Counter_Name']].append((r['Dispatch_Id'], float(r['Counter_Value'])))
    for k,c in acc.items():
        ids=[i for i,_ in c['GRBM_GUI_ACTIVE']]
        ids=ids[len(ids)//2:]  # second half: warm
        m=lambda x: sum(v for i,v in c[x] if i in ids)/max(len(ids),1)
        us=sum(dur[i][0] for i in ids)/max(len(ids),1)
        res[tag+' :: '+k]={'dispatches':len(ids),'us':us,'GRBM_GHz':m('GRBM_GUI_ACTIVE')/8/us/1e3,'SQ_BUSY_per_us':m('SQ_BUSY_CYCLES')/us,
            'mfma_busy_cycles_per_simd':m('SQ_VALU_MFMA_BUSY_CYCLES')/1024,'mfma_insts_per_simd':m('SQ_INSTS_MFMA')/1024,
            'busy_cycles_per_mfma':m('SQ_VALU_MFMA_BUSY_CYCLES')/max(m('SQ_INSTS_MFMA'),1),
            'GHz_if_pipe_never_idle':m('SQ_VALU_MFMA_BUSY_CYCLES')/1024/us/1e3,
            'mfma_util_vs_GRBM':m('SQ_VALU_MFMA_BUSY_CYCLES')/1024/(m('GRBM_GUI_ACTIVE')/8),
            'wave_quadcycles':m('SQ_WAVE_CYCLES')}
json.dump(res, open(O+'/clock_instrument.json','w'), indent=1); print(json.dumps(res, indent=1))
PY
for f in $O/smi_*.txt; do echo "$f: $(sort $f | uniq -c | sort -rn | head -3 | tr '\n' '|')"; done
rm -rf $O/pmc_*/
