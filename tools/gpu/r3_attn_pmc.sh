#!/bin/bash
# per-kernel durations and SQ counters of the four attention kernels (geometry x schedule) at the FLUX.1 shape.  usage: r3_attn_pmc.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$PWD
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -o a -- python $R/tools/prof_attention.py > $R/$O/trace.log 2>&1
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --output-format csv -d $R/$O/pmc_$i -o g -- python $R/tools/prof_attention.py > $R/$O/pmc_$i.log 2>&1
done
cd $R
python - $O <<'PY'
import csv, glob, collections, json, sys
O=sys.argv[1]
res=collections.defaultdict(dict)
for f in glob.glob(O+'/trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'attention_kernel' in r['Name']: res[r['Name'].split('(')[0].replace('void svdq::','')]['avg_us']=float(r['AverageNs'])/1e3
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+'/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'attention_kernel' not in r['Kernel_Name']: continue
        acc[r['Kernel_Name'].split('(')[0].replace('void svdq::','')][r['Counter_Name']].append(float(r['Counter_Value']))
for k,cs in acc.items():
    for c,v in cs.items(): res[k][c]=sum(v[1:])/max(len(v)-1,1)   # (first launch of each kind: cold)
for k,r in res.items():
    if 'GRBM_GUI_ACTIVE' in r and 'avg_us' in r: r['clock_GHz_under_pmc']=r['GRBM_GUI_ACTIVE']/8/r['avg_us']/1e3
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in r and 'GRBM_GUI_ACTIVE' in r: r['mfma_util']=r['SQ_VALU_MFMA_BUSY_CYCLES']/1024/(r['GRBM_GUI_ACTIVE']/8)
    if 'SQ_WAVE_CYCLES' in r:
        for c in ('SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_ANY'): r[c+'_frac']=r.get(c,0)/r['SQ_WAVE_CYCLES']
json.dump(res, open(O+'/attention_pmc.json','w'), indent=1)
print(json.dumps(res, indent=1))
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/trace $O/pmc_*/
