"""Where does svdq_attention differ from the fp32 reference?  (debug aid)"""
import math, sys, torch
from nunchaku_amd.ops.attention import attention_packed

def ref_att(q, k, v):
    qf, kf, vf = (t.float().permute(1, 0, 2) for t in (q, k, v))
    s = qf @ kf.transpose(1, 2) / math.sqrt(q.shape[-1])
    return (torch.softmax(s, dim=-1) @ vf).permute(1, 0, 2), s

for L, H, td in ((1152, 2, torch.float16), (1152, 2, torch.bfloat16), (384, 3, torch.float16), (1024, 3, torch.float16)):
    g = torch.Generator(device="cuda").manual_seed(L + H)
    qkv = torch.randn(L, 3 * H * 128, device="cuda", generator=g).to(td)
    qkv[: L // 2, : H * 128] *= 4.0
    q, k, v = (qkv[:, i * H * 128:(i + 1) * H * 128].unflatten(1, (H, 128)) for i in range(3))
    vt = v.permute(1, 2, 0).contiguous().view(H * 128, L)
    out = attention_packed(qkv, vt, H).float().view(L, H, 128)
    ref, s = ref_att(q, k, v)
    bad = ~torch.isfinite(out)
    print(L, H, td, "nan/inf elements:", int(bad.sum()), "max err (finite):", (out - ref)[~bad].abs().max().item())
    if bad.any():
        rows = bad.any(dim=2).nonzero()
        print("  bad (row, head) pairs:", rows.shape[0], rows[:12].tolist())
        r, hh = rows[0].tolist()
        sc = s[hh, r] * math.log2(math.e)
        tiles = sc.view(-1, 64).amax(dim=1)
        print("  row", r, "head", hh, "per-tile max of score*log2e:", [round(x, 1) for x in tiles.tolist()])
        print("  out row sample:", out[r, hh, :8].tolist())
