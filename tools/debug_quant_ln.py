import numpy as np, torch, sys
sys.path.insert(0, ".")
from oracle import svdq_oracle as O
from tests.helpers import TORCH_DT, f32, make_module, t16
from tests.test_gpu_parity import _gemm_inputs
from nunchaku_amd import layout
for dtype in ("fp16", "bf16"):
    M, K = 512, 3072
    L, x = _gemm_inputs(M, K, 128, 32, dtype, seed=M + K)
    rng = np.random.default_rng(3)
    x = O.round16(x * 2 + 0.5, dtype)
    scale = O.round16(1 + rng.standard_normal(K).astype(np.float32) * 0.3, dtype)
    shift = O.round16(rng.standard_normal(K).astype(np.float32) * 0.2, dtype)
    stats = O.ln_stats_ref(x)
    mod = make_module(L, dtype)
    q, asc, la = mod.quantize(t16(x, dtype), ln=(torch.from_numpy(stats).cuda(), t16(scale, dtype), t16(shift, dtype)))
    xn = O.ln_mod_ref(x, stats, scale, shift, dtype)
    rq, ra, rl = O.quantize_w4a4_act_fuse_lora(xn, L["smooth"], L["proj_down"], dtype)
    gq = layout.unpack_act(q, K).cpu().numpy(); ga = f32(layout.unpack_scales(asc, q.shape[0]))
    bad = np.argwhere(ga != ra)
    print(dtype, "code mismatches", int((gq != rq).sum()), "scale mismatches", len(bad), "of", ra.size, "shape", ra.shape)
    for g, m in bad[:10]:
        print("  group", g, "row", m, "got", ga[g, m], "ref", ra[g, m], "codes equal in group:", np.array_equal(gq[m, g*64:(g+1)*64], rq[m, g*64:(g+1)*64]),
              "max |x_hat| ref", np.abs(rq[m, g*64:(g+1)*64]).max())
