#!/bin/bash
# GPU batch 1 (round 2): where does the GEMM's time go, in cycles -- effective clock, instruction-class ablations, power.
O=gpurun_out/batch1; mkdir -p $O
P=tools/ablate/gemm_probe
AB=tools/ablate/libsvdq_amd_ablate.so
PR=nunchaku_amd/csrc/libsvdq_amd.so
rocminfo | grep -E "Compute Unit|Max Clock|Marketing" | head -8 > $O/rocminfo.txt
rocm-smi --showpower --showclocks --showmaxpower > $O/smi_idle.txt 2>&1
# product library: the shapes of a FLUX step, each epilogue
{
for s in "4608 3072 3072" "4608 12288 3072" "512 3072 3072" "512 12288 3072"; do $P --lib $PR --shape $s; done
$P --lib $PR --shape 4608 3072 9216 --fuse 3
$P --lib $PR --shape 4608 3072 12288 --fuse 2
$P --lib $PR --shape 4608 3072 9216 --fuse 3 --split 4096
$P --lib $PR --shape 4608 3072 12288 --fuse 2 --split 4096
$P --lib $PR --shape 4608 3072 3072 --split 4096
$P --lib $PR --shape 4608 12288 3072 --split 4096
} > $O/prod.jsonl 2> $O/prod.err
# ablations with the clock probe
$P --lib $AB --shape 4096 12288 3072 --variants 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15 > $O/ablate_fc2.jsonl 2> $O/ablate_fc2.err
$P --lib $AB --shape 4608 3072 9216 --variants 0,2,3,4,5,6,7,8,9,10,11,12,13,14,15 > $O/ablate_qkv.jsonl 2> $O/ablate_qkv.err
$P --lib $AB --shape 4096 12288 3072 --variants 0,10,11,15 --zero > $O/ablate_fc2_zero.jsonl 2>&1
$P --lib $AB --shape 4096 12288 3072 --variants 0 --reserved 3 > $O/ablate_fc2_noepi.jsonl 2>&1
# power / clock while the kernel runs back to back
( $P --lib $AB --shape 4096 12288 3072 --variants 0 --sustain 4 > $O/sustain.jsonl 2>&1 ) &
sleep 1.2
for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" >> $O/smi_busy.txt; sleep 0.6; done
wait
( $P --lib $AB --shape 4096 12288 3072 --variants 11 --sustain 4 > $O/sustain_bare.jsonl 2>&1 ) &
sleep 1.2
for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" >> $O/smi_busy_bare.txt; sleep 0.6; done
wait
# SQ counters of the production loop (few launches: every dispatch is a CSV row)
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA -d $R/$O/pmc1 -o g -- $R/$P --lib $R/$AB --shape 4096 12288 3072 --warm 20 --iters 5 > $R/$O/pmc1.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $R/$O/pmc2 -o g -- $R/$P --lib $R/$AB --shape 4096 12288 3072 --warm 20 --iters 5 > $R/$O/pmc2.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ('pmc1','pmc2'):
    for f in glob.glob(f'gpurun_out/batch1/{d}/**/*counter_collection.csv', recursive=True):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'gemm_w4a4' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
        with open('gpurun_out/batch1/pmc_summary.txt','a') as o:
            for k,v in sorted(agg.items()): o.write(f"{d} {k} {sum(v)/len(v):.1f} n={len(v)}\n")
PY
rm -rf $O/pmc1 $O/pmc2
cat $O/prod.jsonl | head -3
