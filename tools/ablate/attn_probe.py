"""Read the clock / phase stamps of a probe build of the attention kernel (tools/ablate/build_attn.py probe[,...]).

    SVDQ_LIB=$PWD/tools/ablate/libsvdq_amd_attn_probe.so PYTHONPATH=$PWD python tools/ablate/attn_probe.py
"""
import ctypes, json, os, sys, torch
import nunchaku_amd._lib as _L
_L._LIB_PATH = os.environ.get("SVDQ_LIB", _L._LIB_PATH)
from nunchaku_amd._C import _Ops
from nunchaku_amd.ops.attention import attention_packed

L, H = 4608, 24
lib = _L.load()
clk = torch.zeros(4 * 1024, dtype=torch.int64, device="cuda")
trace = torch.zeros(16, dtype=torch.int64, device="cuda")
lib.svdq_ablate_set_attn_clk.argtypes = [ctypes.c_void_p]
lib.svdq_ablate_set_attn_trace.argtypes = [ctypes.c_void_p]
lib.svdq_ablate_set_attn_clk(clk.data_ptr())
lib.svdq_ablate_set_attn_trace(trace.data_ptr())
qkv = torch.randn(L, 3 * H * 128, device="cuda").bfloat16()
vt = qkv[:, 2 * H * 128:].t().contiguous()
out = torch.empty(L, H * 128, device="cuda", dtype=torch.bfloat16)
_Ops.attention_geometry = 2
for ws in (False, True):
    _Ops.attention_use_workspace = ws
    for _ in range(5):
        attention_packed(qkv, vt, H, out=out)
    torch.cuda.synchronize()
    clk.zero_(); trace.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); attention_packed(qkv, vt, H, out=out); e1.record(); torch.cuda.synchronize()
    c = clk.view(-1, 4).cpu()
    c = c[c[:, 0] > 0].double()
    rec = {"schedule": "persistent" if ws else "plain grid", "us": round(e0.elapsed_time(e1) * 1e3, 1), "workgroups": len(c),
           "clock_GHz": round(float((c[:, 0] / c[:, 1]).mean()) * 0.1, 3),
           "cycles_per_tile_in_loop": round(float(c[:, 2].sum() / c[:, 3].sum()), 1),
           "loop_share_of_kernel_cycles": round(float(c[:, 2].sum() / c[:, 0].sum()), 3),
           "kernel_cycles_min_mean_max": [int(c[:, 0].min()), int(c[:, 0].mean()), int(c[:, 0].max())]}
    print(json.dumps(rec), flush=True)
