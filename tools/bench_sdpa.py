import torch, time, torch.nn.functional as F
q=torch.randn(1,24,4608,128,device="cuda",dtype=torch.bfloat16); k=torch.randn_like(q); v=torch.randn_like(q)
def t(fn,it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/it*1e3
fl=4*24*4608*4608*128
us=t(lambda: F.scaled_dot_product_attention(q,k,v)); print("default sdpa us",us,"TF",fl/us/1e6)
try:
    print("preferred lib:", torch.backends.cuda.preferred_rocm_fa_library())
    torch.backends.cuda.preferred_rocm_fa_library("ck")
    us=t(lambda: F.scaled_dot_product_attention(q,k,v)); print("ck sdpa us",us,"TF",fl/us/1e6)
except Exception as e: print("ck pref failed:", repr(e)[:200])
from torch.nn.attention import sdpa_kernel, SDPBackend
for b in (SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION, SDPBackend.MATH):
    try:
        with sdpa_kernel(b):
            us=t(lambda: F.scaled_dot_product_attention(q,k,v),5); print(b,"us",us,"TF",fl/us/1e6)
    except Exception as e: print(b,"failed",repr(e)[:120])
# layout variant: [B, L, H, D] transposed views
q2=q.transpose(1,2).contiguous().transpose(1,2)
us=t(lambda: F.scaled_dot_product_attention(q2,k,v)); print("BLHD-strided q us",us)
