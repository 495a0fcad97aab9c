"""Run one GEMM shape a few times (for rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.bench_kernels import rand_layer
M, K, N = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 3072, 9216))]
lin = rand_layer(K, N)
x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
qx, asc, la = lin.quantize(x)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    lin.forward_quant(qx, asc, la, out)
torch.cuda.synchronize()
