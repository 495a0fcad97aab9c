"""AWQ W4A16 GEMV at the FLUX modulation shapes vs the 16-bit nn.Linear it replaces (hipBLASLt)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_kernels import timeit
from nunchaku_amd.models.linear import AWQW4A16Linear

for (K, N) in ((3072, 18432), (3072, 9216)):
    lin = AWQW4A16Linear(K, N, device="cuda")
    lin.qweight.data.copy_(torch.randint(-2**31, 2**31, lin.qweight.shape, device="cuda"))
    lin.wscales.data.fill_(0.01); lin.wzeros.data.fill_(-0.075); lin.bias.data.zero_()
    ref = torch.nn.Linear(K, N, dtype=torch.bfloat16, device="cuda")
    for m in (1, 4):
        x = torch.randn(m, K, device="cuda", dtype=torch.bfloat16)
        t = timeit(lambda: lin(x), 50)
        t2 = timeit(lambda: ref(x), 50)
        by = N * K / 2 + 4 * (K // 64) * N
        print(f"K={K} N={N} m={m}: awq gemv {t*1e6:.1f} us ({by/t/1e9:.0f} GB/s of int4 bytes)   bf16 nn.Linear {t2*1e6:.1f} us")
