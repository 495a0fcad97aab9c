"""svdq_attention vs torch SDPA at the FLUX.1 shape (1 x 24 heads x 4608 tokens x 128), plus the QKV GEMM with and
without the transposed-V side output."""
import math, os, sys, torch, torch.nn.functional as F
import nunchaku_amd._lib as _L
_L._LIB_PATH = os.environ.get("SVDQ_LIB", _L._LIB_PATH)  # same-box A/B against another build of the library (tools only)
from nunchaku_amd.ops.attention import attention_packed

L, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4608, 24)
import os
qkv = torch.randn(L, 3 * H * 128, device="cuda").bfloat16()
if os.environ.get("ATT_ZERO"): qkv.zero_()  # power / clock experiment: same instructions, no toggling
vt = qkv[:, 2 * H * 128:].t().contiguous()
out = torch.empty(L, H * 128, device="cuda", dtype=torch.bfloat16)

def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3

fl = 4 * H * L * L * 128
from nunchaku_amd._C import _Ops, ops
for rep in range(2):  # same box, interleaved: persistent schedule (workspace) vs plain grid
    _Ops.attention_use_workspace = False
    us0 = t(lambda: attention_packed(qkv, vt, H, out=out)); print(f"svdq_attention plain grid  {us0:.1f} us  {fl/us0/1e6:.0f} TFLOP/s")
    _Ops.attention_use_workspace = True
    us = t(lambda: attention_packed(qkv, vt, H, out=out)); print(f"svdq_attention persistent  {us:.1f} us  {fl/us/1e6:.0f} TFLOP/s")
ops.attention_workspace_status()
q, k, v = (qkv[:, i * H * 128:(i + 1) * H * 128].unflatten(1, (H, 128)).permute(1, 0, 2)[None] for i in range(3))
us2 = t(lambda: F.scaled_dot_product_attention(q, k, v)); print(f"torch sdpa (strided views) {us2:.1f} us  {fl/us2/1e6:.0f} TFLOP/s")
ref = F.scaled_dot_product_attention(q.float(), k.float(), v.float())[0].permute(1, 0, 2).reshape(L, H * 128)
print("max err vs fp32:", (out.float() - ref).abs().max().item(), " sdpa bf16 err:",
      (F.scaled_dot_product_attention(q, k, v)[0].permute(1, 0, 2).reshape(L, H * 128).float() - ref).abs().max().item())
