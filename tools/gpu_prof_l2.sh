#!/bin/bash
# L2 / fabric view of the GEMM: hit rate, requests, HBM-side fetch bytes (separate PMC passes)
mkdir -p gpurun_out/prof_l2
export TMPDIR=/tmp
R=$PWD
SHAPE="${1:-4096 12288 3072}"
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_l2/trace -o gemm --output-format csv -- python $R/tools/prof_gemm.py $SHAPE > $R/gpurun_out/prof_l2/trace.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d $R/gpurun_out/prof_l2/hit -o gemm --output-format csv -- python $R/tools/prof_gemm.py $SHAPE > $R/gpurun_out/prof_l2/hit.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_l2/fetch -o gemm --output-format csv -- python $R/tools/prof_gemm.py $SHAPE > $R/gpurun_out/prof_l2/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_RDREQ_sum -d $R/gpurun_out/prof_l2/write -o gemm --output-format csv -- python $R/tools/prof_gemm.py $SHAPE > $R/gpurun_out/prof_l2/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d $R/gpurun_out/prof_l2/sq -o gemm --output-format csv -- python $R/tools/prof_gemm.py $SHAPE > $R/gpurun_out/prof_l2/sq.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/prof_l2/trace/**/*kernel_stats.csv', recursive=True):
    print(open(f).read()[:900])
for d in ('hit','fetch','write','sq'):
    for f in glob.glob(f'gpurun_out/prof_l2/{d}/**/*counter_collection.csv', recursive=True):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'gemm_w4a4' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in agg.items(): print(d,k,sum(v)/len(v), len(v))
PY
