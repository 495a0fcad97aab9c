#!/bin/bash
# effective clock of the attention kernels: GRBM_GUI_ACTIVE and the dispatch duration of the SAME dispatches (rocprofv3 --pmc + --kernel-trace).
# usage: r3_attn_clock.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp PYTHONPATH=$PWD ATT_REPS=8
R=$PWD
cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/$O/clk -o g -- python $R/tools/prof_attention.py > $R/$O/clk.log 2>&1
cd $R
python - $O <<'PY'
import csv, glob, collections, json, sys
O=sys.argv[1]
dur={}
for f in glob.glob(O+'/clk/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'attention_kernel' in r['Kernel_Name']: dur[r['Dispatch_Id']]=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+'/clk/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'attention_kernel' not in r['Kernel_Name'] or r['Dispatch_Id'] not in dur: continue
        k=r['Kernel_Name'].split('(')[0].replace('void svdq::','')
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        if r['Counter_Name']=='GRBM_GUI_ACTIVE': acc[k]['us'].append(dur[r['Dispatch_Id']])
res={}
for k,c in acc.items():
    n=len(c['us']); sl=slice(n//2, n)   # second half: warm
    m=lambda x: sum(c[x][sl])/max(len(c[x][sl]),1)
    res[k]={'us':m('us'),'clock_GHz':m('GRBM_GUI_ACTIVE')/8/m('us')/1e3,'mfma_util':m('SQ_VALU_MFMA_BUSY_CYCLES')/1024/(m('GRBM_GUI_ACTIVE')/8),
            'issuing':m('SQ_ACTIVE_INST_ANY')/m('SQ_WAVE_CYCLES'),'issue_stalled':m('SQ_WAIT_INST_ANY')/m('SQ_WAVE_CYCLES'),'parked':m('SQ_WAIT_ANY')/m('SQ_WAVE_CYCLES'),
            'wave_cycles_x4_per_wave': m('SQ_WAVE_CYCLES')*4/ (432*(4 if '64' in k else 8) if 'false' in k else 256*(4 if '64' in k else 8))}
json.dump(res, open(O+'/attention_clock.json','w'), indent=1); print(json.dumps(res, indent=1))
PY
rm -rf $O/clk
