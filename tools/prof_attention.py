"""A few launches of every attention kernel variant at the FLUX.1 shape, for rocprofv3 (tools/gpu/r3_attn_pmc.sh)."""
import os, sys, torch
from nunchaku_amd._C import _Ops
from nunchaku_amd.ops.attention import attention_packed

L, H = 4608, 24
qkv = torch.randn(L, 3 * H * 128, device="cuda").bfloat16()
vt = qkv[:, 2 * H * 128:].t().contiguous()
out = torch.empty(L, H * 128, device="cuda", dtype=torch.bfloat16)
for rep in range(int(os.environ.get("ATT_REPS", "6"))):
    for geo in (1, 2):
        for ws in (False, True):
            _Ops.attention_geometry, _Ops.attention_use_workspace = geo, ws
            attention_packed(qkv, vt, H, out=out)
torch.cuda.synchronize()
