#!/usr/bin/env python3
"""Which feature of the product path moves the hidden-3072 forward away from the CPU twin?  GPU variants against each
other (cheap) and the torch-op variant against the oracle-backed twin (one forward)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nunchaku_amd.models.flux import FluxTransformerAMD, FluxAttentionAMD
from tests.flux_ref import Ref, fill_model_, synthetic_inputs, psnr_rel

realistic = "--cheap" not in sys.argv
model = FluxTransformerAMD(num_layers=1, num_single_layers=1, dim=3072, heads=24, in_channels=64, joint_attention_dim=4096,
                           pooled_projection_dim=768, device="cuda")
layers = fill_model_(model, seed=11, realistic=realistic)
model.eval()
lat, enc, pooled, img_ids, txt_ids = synthetic_inputs(16, 256, 4096, 768, seed=12)
t, gd = torch.tensor([1.0]), torch.tensor([3.5])

def run(**kw):
    saved = (FluxTransformerAMD.fused_norm, FluxTransformerAMD.batched_mods, FluxAttentionAMD.grouped, FluxAttentionAMD.fused_out_quant, FluxAttentionAMD.attention_impl)
    FluxTransformerAMD.fused_norm = kw.get("fused_norm", True); FluxTransformerAMD.batched_mods = kw.get("batched_mods", True)
    FluxAttentionAMD.grouped = kw.get("grouped", True); FluxAttentionAMD.fused_out_quant = kw.get("fused_out_quant", True)
    FluxAttentionAMD.attention_impl = kw.get("attention_impl", "svdq")
    with torch.no_grad():
        out = model(lat.cuda().bfloat16()[None], enc.cuda().bfloat16()[None], pooled.cuda().bfloat16(), t.cuda(), img_ids.cuda(), txt_ids.cuda(), gd.cuda())[0].float().cpu()
    (FluxTransformerAMD.fused_norm, FluxTransformerAMD.batched_mods, FluxAttentionAMD.grouped, FluxAttentionAMD.fused_out_quant, FluxAttentionAMD.attention_impl) = saved
    return out

outs = {
    "default": run(),
    "torch_ops": run(fused_norm=False),
    "torch_ops_sdpa": run(fused_norm=False, attention_impl="sdpa"),
    "no_grouped": run(grouped=False),
    "no_attn_quant": run(fused_out_quant=False),
    "no_batched_mods": run(batched_mods=False),
    "sdpa": run(attention_impl="sdpa"),
}
base = outs["torch_ops_sdpa"]
for k, v in outs.items():
    print(k, "vs torch_ops_sdpa: psnr %.1f rel %.3e" % psnr_rel(v, base), " |out| max %.3f" % v.abs().max().item(), flush=True)
with torch.no_grad():
    ref = Ref(model, layers).forward(lat, enc, pooled, t, img_ids, txt_ids, gd)
for k, v in outs.items():
    print(k, "vs CPU twin: psnr %.1f rel %.3e" % psnr_rel(v, ref), flush=True)
