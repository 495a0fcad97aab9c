"""Quantiser kernel time (library profiler: HIP events around the launch) at FLUX shapes, plain and with the fused LN."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nunchaku_amd._lib as _L0
_L0._LIB_PATH = os.environ.get("SVDQ_LIB", _L0._LIB_PATH)  # same-box A/B against another build of the library (tools only)
from tools.bench_kernels import rand_layer
from nunchaku_amd import _lib
from nunchaku_amd.ops.elementwise import residual_gate_stats
lib = _lib.load()
for M in (4096, 4608, 512):
    lin = rand_layer(3072, 3072)
    x = torch.randn(M, 3072, device="cuda", dtype=torch.bfloat16)
    _, st = residual_gate_stats(x)
    sc = torch.randn(3072, device="cuda", dtype=torch.bfloat16) * 0.1; sh = torch.randn(3072, device="cuda", dtype=torch.bfloat16) * 0.1
    for name, ln in (("plain", None), ("fused-LN", (st, sc, sh))):
        for _ in range(3): lin.quantize(x, ln=ln)
        lib.svdq_prof_enable(256); lib.svdq_prof_reset()
        for _ in range(20): lin.quantize(x, ln=ln)
        n, ms, w = C.c_int64(), C.c_double(), C.c_double()
        lib.svdq_prof_read(1, C.byref(n), C.byref(ms), C.byref(w)); lib.svdq_prof_enable(0)
        print(f"M={M} {name}: {ms.value/n.value*1e3:.1f} us per call (incl. memset), {w.value/ms.value/1e6:.0f} GB/s")
