"""What the attention kernel's fused output quantiser costs at the FLUX.1 shape (1 x 24 heads x 4608 tokens x 128, Q prescaled: the 4 x 64 kernel on the plain grid):
16-bit output only / fused quantiser at rank 32 (codes + scales + low-rank down with atomics over the 24 heads) / fused quantiser at rank 0 (no low-rank part)."""
import math, os, sys, torch
import nunchaku_amd._lib as _L
_L._LIB_PATH = os.environ.get("SVDQ_LIB", _L._LIB_PATH)  # same-box A/B against another build of the library (tools only)
from oracle import svdq_oracle as O
from tests.helpers import make_module
from nunchaku_amd.ops.attention import attention_packed, attention_packed_quantized, q_prescale

L, H = 4608, 24
qkv = torch.randn(L, 3 * H * 128, device="cuda").bfloat16()
qkv[:, : H * 128] *= q_prescale()
vt = qkv[:, 2 * H * 128:].t().contiguous()
out = torch.empty(L, H * 128, device="cuda", dtype=torch.bfloat16)
lin32 = make_module(O.make_random_svdq_layer(3072, 3072, 32, seed=1, dtype="bf16"), "bf16")
lin0 = make_module(O.make_random_svdq_layer(3072, 3072, 0, seed=1, dtype="bf16"), "bf16") if os.environ.get("RANK0", "1") == "1" else None

def t(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3

cases = {"16-bit output": lambda: attention_packed(qkv, vt, H, out=out, q_prescaled=True),
         "fused quantiser rank 32": lambda: attention_packed_quantized(qkv, vt, H, lin32, q_prescaled=True)}
if lin0 is not None:
    cases["fused quantiser rank 0"] = lambda: attention_packed_quantized(qkv, vt, H, lin0, q_prescaled=True)
best = {}
for rep in range(5):
    for k, fn in cases.items():
        best[k] = min(best.get(k, 1e9), t(fn))
fl = 4 * H * L * L * 128
for k, us in best.items():
    print(f"{k}: {us:.1f} us  {fl/us/1e6:.0f} TFLOP/s (best of 5 x 30; the quantised variants include the memset of lora_act)")
