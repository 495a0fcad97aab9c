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
outs, best = {}, {}
REPS = int(os.environ.get("ATT_REPS", "5"))
for rep in range(REPS):  # same box, interleaved: both geometries x persistent schedule (workspace) vs plain grid; best of REPS
    for geo in (1, 2):
        if geo == 2 and L % 256: continue
        _Ops.attention_geometry = geo
        for ws in (False, True):
            _Ops.attention_use_workspace = ws
            us = t(lambda: attention_packed(qkv, vt, H, out=out), it=30)
            best[(geo, ws)] = min(best.get((geo, ws), 1e9), us)
        outs[geo] = out.clone()
for (geo, ws), us in sorted(best.items()):
    print(f"svdq_attention geometry {geo} {'persistent' if ws else 'plain grid'}  {us:.1f} us  {fl/us/1e6:.0f} TFLOP/s  (best of {REPS} x 30 launches)", flush=True)
ops.attention_workspace_status()
if 2 in outs: print("geometry 2 vs 1: max |diff|", (outs[2].float() - outs[1].float()).abs().max().item())
_Ops.attention_geometry = int(os.environ.get("ATT_GEOMETRY", "0")); attention_packed(qkv, vt, H, out=out)
q, k, v = (qkv[:, i * H * 128:(i + 1) * H * 128].unflatten(1, (H, 128)).permute(1, 0, 2)[None] for i in range(3))
us2 = t(lambda: F.scaled_dot_product_attention(q, k, v)); print(f"torch sdpa (strided views) {us2:.1f} us  {fl/us2/1e6:.0f} TFLOP/s")
ref = F.scaled_dot_product_attention(q.float(), k.float(), v.float())[0].permute(1, 0, 2).reshape(L, H * 128)
print("max err vs fp32:", (out.float() - ref).abs().max().item(), " sdpa bf16 err:",
      (F.scaled_dot_product_attention(q, k, v)[0].permute(1, 0, 2).reshape(L, H * 128).float() - ref).abs().max().item())
