"""A/B the hand-scheduled main loop (variant 0) against the compiler-scheduled one (variant 1): bitwise."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nunchaku_amd.models.linear import SVDQW4A4Linear
from tools.bench_kernels import rand_layer

def run(M, K, N, dtype, seed=0):
    torch.manual_seed(seed)
    lin = rand_layer(K, N)
    if dtype == torch.float16:
        lin = lin.to(torch.float16); lin.torch_dtype = torch.float16
    x = torch.randn(M, K, device="cuda", dtype=dtype)
    qx, asc, la = lin.quantize(x)
    outs = []
    for v in (1, 0, 0, 0):
        os.environ["SVDQ_GEMM_VARIANT"] = str(v)
        out = torch.empty(M, N, device="cuda", dtype=dtype)
        lin.forward_quant(qx, asc, la, out)
        torch.cuda.synchronize()
        outs.append(out.float().cpu().numpy())
    ref = outs[0]
    for i, o in enumerate(outs[1:]):
        bad = np.argwhere(o != ref)
        print(f"M={M} K={K} N={N} {dtype} run{i}: mismatches {len(bad)} / {o.size}", flush=True)
        if len(bad):
            for (m, n) in bad[:12]:
                print(f"   m={m} (blk {m//256} wm {(m%256)//64} mi {(m%64)//32} lr {m%32})  n={n} (blk {n//128} wn {(n%128)//64} ni {(n%64)//32} r-idx c={(n%32)//8} h={(n%8)//4} e={n%4})  got {o[m,n]:.6f} ref {ref[m,n]:.6f}")
    os.environ["SVDQ_GEMM_VARIANT"] = "0"

for (M, K, N) in [(512, 3072, 384), (512, 3072, 384), (4096, 3072, 3072), (256, 128, 128), (256, 256, 128)]:
    for dt in (torch.bfloat16, torch.float16):
        run(M, K, N, dt)
