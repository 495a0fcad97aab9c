#!/bin/bash
# round 4: non-temporal operand streams (generator options "ntw" / "nta"): isolated launch times, interleaved, + fabric-side fetch of the fc1 launch.
# usage: r4_nt.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
P=tools/ablate/gemm_probe
LIBS="tools/ablate/libsvdq_amd_probe.so tools/ablate/libsvdq_amd_probe_ntw.so tools/ablate/libsvdq_amd_probe_nta.so tools/ablate/libsvdq_amd_probe_ntw_nta.so"
for rep in 1 2 3; do for lib in $LIBS; do
  for s in "4608 3072 12288 2" "4608 3072 9216 3" "4608 3072 3072 0" "4608 12288 3072 0" "512 3072 3072 0"; do
    set -- $s
    echo -n "$(basename $lib) " >> $O/nt_ab.txt
    timeout 120 $P --lib $lib --shape $1 $2 $3 --fuse $4 --geoms 0 2>>$O/nt_ab.err | grep '"us"' | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['M'],r['K'],r['N'],'fuse',r['fuse'],r['us'],'us',r['TOPS'],'TOPS',r['eff_GHz'],'GHz sum',r['sum'])" >> $O/nt_ab.txt
  done; done; done
cat $O/nt_ab.txt
export TMPDIR=/tmp; R=$PWD; cd /tmp
for lib in $LIBS; do
  b=$(basename $lib .so)
  for s in "4608 3072 12288 2" "4608 3072 9216 3" "4608 12288 3072 0"; do
    set -- $s
    timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/f_${b}_$3_$4 -o g -- $R/$P --lib $R/$lib --shape $1 $2 $3 --fuse $4 --geoms 0 --iters 5 --warm 5 > /dev/null 2>&1
  done
done
cd $R
python3 - $O <<'PY'
import csv,glob,sys,os
O=sys.argv[1]
for d in sorted(glob.glob(O+'/f_*')):
    v=[]
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'gemm_w4a4' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE': v.append(float(r['Counter_Value']))
    if v: print(os.path.basename(d), 'FETCH_SIZE KB per dispatch (x2 for bytes on gfx950):', round(sum(v[len(v)//2:])/len(v[len(v)//2:]),1), 'n', len(v))
PY
rm -rf $O/f_*
