"""AWQ W4A16 GEMV kernel time (library profiler: HIP events around the launch), bf16 and fp16: the FLUX modulation projection
3072 -> 18432 alone and a step's worth of them (19 x 2 x 18432 + 38 x 9216 outputs) as ONE batched launch."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nunchaku_amd._lib as _L0
_L0._LIB_PATH = os.environ.get("SVDQ_LIB", _L0._LIB_PATH)  # same-box A/B against another build of the library (tools only)
from nunchaku_amd import _lib
from nunchaku_amd.models.linear import AWQW4A16Linear
from nunchaku_amd.ops.gemv import awq_gemv_w4a16_batched
lib = _lib.load()


def layer(K, N, dt):
    lin = AWQW4A16Linear(K, N, torch_dtype=dt, device="cuda")
    lin.qweight.data = torch.randint(-2 ** 31, 2 ** 31 - 1, lin.qweight.shape, dtype=torch.int32, device="cuda")
    lin.wscales.data = (torch.rand(lin.wscales.shape, device="cuda") * 0.01 + 0.005).to(dt)
    lin.wzeros.data = (-7.5 * lin.wscales.data.float()).to(dt)
    lin.bias.data = (torch.randn(N, device="cuda") * 0.1).to(dt)
    return lin


def timed(fn, reps=20):
    for _ in range(3): fn()
    lib.svdq_prof_select(1 << 3); lib.svdq_prof_enable(256); lib.svdq_prof_reset()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    n, ms, w = C.c_int64(), C.c_double(), C.c_double()
    lib.svdq_prof_read(3, C.byref(n), C.byref(ms), C.byref(w)); lib.svdq_prof_enable(0); lib.svdq_prof_select(0xFFFFFFFF)
    return ms.value / n.value * 1e3, w.value / ms.value / 1e6


for dt in (torch.bfloat16, torch.float16):
    x = torch.randn(1, 3072, device="cuda").to(dt)
    one = layer(3072, 18432, dt)
    us, gbps = timed(lambda: one(x))
    print(f"{str(dt)[6:]:9s} 3072 -> 18432, one launch: {us:7.1f} us  {gbps:6.0f} GB/s")
    step = [layer(3072, 18432, dt) for _ in range(38)] + [layer(3072, 9216, dt) for _ in range(38)]
    for l in step: l.out_chunks = 6 if l.out_features == 18432 else 3
    us, gbps = timed(lambda: awq_gemv_w4a16_batched(x, step), reps=10)
    print(f"{str(dt)[6:]:9s} a FLUX.1-dev step's 76 projections, one batched launch: {us:7.1f} us  {gbps:6.0f} GB/s")
    del step
