bash tools/gpu/r5_tests_bench.sh t3 tests/test_gpu_parity.py tests/test_gpu_parity_fullsize.py tests/test_gpu_qwenimage.py tests/test_gpu_geometry_determinism.py
bash tools/gpu/r5_rank_ab.sh ab2
