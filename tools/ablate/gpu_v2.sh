#!/bin/bash
O=gpurun_out/v2; mkdir -p $O
P=tools/ablate/gemm_probe; AB=tools/ablate/libsvdq_amd_ablate.so; PR=nunchaku_amd/csrc/libsvdq_amd.so
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_fullsize.py -m gpu -x -q > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
$P --lib $AB --shape 4096 12288 3072 --variants 0,2 > $O/fc2.jsonl 2>&1
$P --lib $AB --shape 4608 3072 9216 --variants 0,2 >> $O/fc2.jsonl 2>&1
$P --lib $AB --shape 4608 3072 3072 --variants 0,2 >> $O/fc2.jsonl 2>&1
$P --lib $AB --shape 512 3072 3072 --variants 0,2 >> $O/fc2.jsonl 2>&1
$P --lib $AB --shape 4096 12288 3072 --variants 0,2 --reserved 3 >> $O/fc2.jsonl 2>&1
cut -c1-220 $O/fc2.jsonl
