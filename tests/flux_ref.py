"""CPU twin of FluxTransformerAMD for the block / latent parity checks (test infrastructure; used by
tests/test_flux_block_parity.py and tools/latent_parity.py): the SAME module structure driven by hand with plain torch
fp32 for the 16-bit pieces (one explicit 16-bit rounding per torch op of the reference) and the numpy oracle for every
SVDQuant / AWQ operator."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import svdq_oracle as O
from tests.helpers import reference_state_dict

DT = "bf16"


def r16(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).float()


class Ref:
    """CPU twin of FluxTransformerAMD.forward with oracle-backed SVDQ layers."""

    def __init__(self, model, layers):
        self.m, self.L = model, layers  # layers: module name -> logical oracle layer

    def lin(self, mod, x):  # nn.Linear with 16-bit weights, fp32 accumulate, 16-bit output
        return r16(F.linear(x, mod.weight.float().cpu(), mod.bias.float().cpu()))

    def awq(self, name, x):  # AWQW4A16Linear (modulation): oracle GEMV with the fused 16-bit bias add
        L = self.L[name]
        return torch.from_numpy(O.awq_gemv_w4a16(x.numpy(), L["q"], L["s"], L["z"], DT, bias=L["bias"]))

    def svdq(self, name, x):
        return torch.from_numpy(O.svdq_linear(x.numpy(), self.L[name], DT, "fp32")["out"])

    def qkv(self, name, x, nq, nk, rot):
        L = self.L[name]
        q, a, la = O.quantize_w4a4_act_fuse_lora(x.numpy(), L["smooth"], L["proj_down"], DT)
        M_pad = q.shape[0]
        rp = np.zeros((M_pad, 64, 2), np.float32)
        rp[: x.shape[0]] = rot
        out = O.gemm_w4a4(q, a, L["qweight"], L["wscales"], dtype=DT, bias=L["bias"], lora_act_in=la, lora_up=L["proj_up"],
                          fuse="rmsnorm_rope", norm_q=nq.float().cpu().numpy(), norm_k=nk.float().cpu().numpy(), rot=rp)["out"]
        return torch.from_numpy(out[: x.shape[0]])

    def mlp(self, n1, n2, x):
        return torch.from_numpy(O.fused_gelu_mlp(x.numpy(), self.L[n1], self.L[n2], DT))

    @staticmethod
    def ln_mod(x, scale, shift):
        # NunchakuAdaLayerNormZero, scale_shift = 0: norm(x) * scale + shift, one 16-bit rounding per torch op
        return r16(r16(r16(F.layer_norm(x, (x.shape[-1],), eps=1e-6)) * scale[None]) + shift[None])

    @staticmethod
    def attend(qkv, heads):
        T = qkv.shape[0]
        q, k, v = [t.reshape(T, heads, 128).transpose(0, 1) for t in qkv.chunk(3, dim=-1)]
        o = torch.softmax(q @ k.transpose(1, 2) / 128 ** 0.5, dim=-1) @ v
        return r16(o.transpose(0, 1).reshape(T, heads * 128))

    def forward(self, lat, enc, pooled, t, img_ids, txt_ids, g, control=None, control_single=None):
        """``control`` / ``control_single``: one ControlNet residual [T_img, dim] added to the image stream behind the joint / the single
        block (diffusers' FluxTransformer2DModel.forward: a 16-bit add)"""
        from nunchaku_amd.models.flux import timestep_embedding
        from nunchaku_amd.models.embeddings import flux_pos_embed
        m = self.m
        emb = lambda e, x: self.lin(e.linear_2, r16(F.silu(self.lin(e.linear_1, x))))
        hidden = self.lin(m.x_embedder, lat)
        temb = emb(m.time_embed, r16(timestep_embedding(r16(r16(t) * 1000))))  # diffusers: timestep.to(dtype) * 1000, a 16-bit op
        if m.guidance_embed is not None:  # FLUX.1-schnell has no guidance embedding
            temb = r16(temb + emb(m.guidance_embed, r16(timestep_embedding(r16(r16(g) * 1000)))))
        temb = r16(temb + emb(m.text_embed, pooled))
        ta = r16(F.silu(temb))
        e = self.lin(m.context_embedder, enc)
        rot = flux_pos_embed(torch.cat([txt_ids, img_ids], 0), m.axes)[0, :, :, 0].numpy()  # [T, 64, (sin, cos)]
        tt = e.shape[0]
        b = m.blocks[0]
        mm = self.awq("transformer_blocks.0.norm1.linear", ta).view(-1, 6).T
        cc = self.awq("transformer_blocks.0.norm1_context.linear", ta).view(-1, 6).T
        n_h, n_e = self.ln_mod(hidden, mm[1], mm[0]), self.ln_mod(e, cc[1], cc[0])
        qkv = torch.cat([self.qkv("transformer_blocks.0.attn.add_qkv_proj", n_e, b.attn.norm_added_q.weight, b.attn.norm_added_k.weight, rot[:tt]),
                         self.qkv("transformer_blocks.0.attn.to_qkv", n_h, b.attn.norm_q.weight, b.attn.norm_k.weight, rot[tt:])])
        o = self.attend(qkv, b.attn.heads)
        a, ca = self.svdq("transformer_blocks.0.attn.to_out.0", o[tt:]), self.svdq("transformer_blocks.0.attn.to_add_out", o[:tt])
        hidden = r16(hidden + r16(mm[2][None] * a))
        hidden = r16(hidden + r16(mm[5][None] * self.mlp("transformer_blocks.0.ff.net.0.proj", "transformer_blocks.0.ff.net.2", self.ln_mod(hidden, mm[4], mm[3]))))
        e = r16(e + r16(cc[2][None] * ca))
        e = r16(e + r16(cc[5][None] * self.mlp("transformer_blocks.0.ff_context.net.0.proj", "transformer_blocks.0.ff_context.net.2", self.ln_mod(e, cc[4], cc[3]))))
        if control is not None:
            hidden = r16(hidden + control)
        x = torch.cat([e, hidden])
        s = m.single_blocks[0]
        sm = self.awq("single_transformer_blocks.0.norm.linear", ta).view(-1, 3).T
        n = self.ln_mod(x, sm[1], sm[0])
        mlp = self.mlp("single_transformer_blocks.0.mlp_fc1", "single_transformer_blocks.0.mlp_fc2", n)
        att = self.svdq("single_transformer_blocks.0.attn.to_out",
                        self.attend(self.qkv("single_transformer_blocks.0.attn.to_qkv", n, s.attn.norm_q.weight, s.attn.norm_k.weight, rot), s.attn.heads))
        x = r16(x + r16(sm[2][None] * r16(att + mlp)))[tt:]
        if control_single is not None:
            x = r16(x + control_single)
        sc, sh = self.lin(m.norm_out_mod, ta).chunk(2, dim=-1)
        x = r16(r16(F.layer_norm(x, (x.shape[-1],), eps=1e-6)) * r16(1 + sc) + sh)
        return self.lin(m.proj_out, x)




def fill_model_(model, seed: int = 0, realistic: bool = False, rank: int = 32, lowrank_energy: float = 0.0):
    """Deterministic weights for ``model`` (a FluxTransformerAMD on the GPU) and the matching logical oracle layers.
    ``realistic``: SVDQuant layers from a dense Gaussian weight, smoothing, a rank-``rank`` (randomised) SVD and the
    4-bit residual -- the code / scale statistics of a real checkpoint; otherwise the cheap random-factor variant."""
    from nunchaku_amd.models.linear import AWQW4A16Linear, SVDQW4A4Linear

    torch.manual_seed(seed)
    layers = {}
    with torch.no_grad():
        for name, mod in model.named_modules():
            if isinstance(mod, SVDQW4A4Linear):
                L = O.make_svdq_layer(mod.in_features, mod.out_features, rank, seed=seed * 1000 + len(layers), dtype=DT,
                                      cheap=not realistic, svd="randomized", lowrank_energy=lowrank_energy)
                layers[name] = L
                mod.load_state_dict({k: v.cuda() for k, v in reference_state_dict(L, DT).items()})
            elif isinstance(mod, AWQW4A16Linear):
                rng = np.random.default_rng(seed * 1000 + 500 + len(layers))
                w = O.round16(rng.standard_normal((mod.out_features, mod.in_features)).astype(np.float32) / mod.in_features ** 0.5, DT)
                q, s_, z_ = O.awq_quantize_ref(w, DT)
                bias = rng.standard_normal(mod.out_features).astype(np.float32) * 0.02
                bias.reshape(-1, mod.out_features // mod.in_features)[:, 1::3] += 1.0  # scale chunks carry the +1 (scale_shift = 0)
                bias = O.round16(bias, DT)
                layers[name] = {"q": q, "s": s_, "z": z_, "bias": bias}
                mod.load_state_dict({"qweight": torch.from_numpy(O.pack_awq_w4_ref(q)).cuda(), "wscales": torch.from_numpy(s_).cuda().bfloat16(),
                                     "wzeros": torch.from_numpy(z_).cuda().bfloat16(), "bias": torch.from_numpy(bias).cuda().bfloat16()})
            elif isinstance(mod, torch.nn.Linear):
                mod.weight.copy_(torch.randn_like(mod.weight, dtype=torch.float32) / mod.in_features ** 0.5)
                mod.bias.copy_(torch.randn_like(mod.bias, dtype=torch.float32) * 0.02)
            elif isinstance(mod, torch.nn.RMSNorm):
                mod.weight.copy_(1 + 0.1 * torch.randn_like(mod.weight, dtype=torch.float32))
    return layers


def synthetic_inputs(side, t_txt: int, joint_dim: int, pooled_dim: int, seed: int = 1):
    """``side``: the latent grid, an int (square) or (height, width) in 2 x 2 patches -- e.g. (58, 104) for 1664 x 928 pixels"""
    h, w = (side, side) if isinstance(side, int) else side
    g = torch.Generator().manual_seed(seed)
    lat = r16(torch.randn(h * w, 64, generator=g))
    enc = r16(torch.randn(t_txt, joint_dim, generator=g))
    pooled = r16(torch.randn(1, pooled_dim, generator=g))
    img_ids = torch.zeros(h * w, 3)
    img_ids[:, 1] = torch.arange(h).repeat_interleave(w)
    img_ids[:, 2] = torch.arange(w).repeat(h)
    return lat, enc, pooled, img_ids, torch.zeros(t_txt, 3)


def psnr_rel(got: torch.Tensor, ref: torch.Tensor):
    mse = ((got - ref) ** 2).mean().item()
    return 10 * np.log10(ref.abs().max().item() ** 2 / max(mse, 1e-30)), (got - ref).norm().item() / ref.norm().item()


def euler_parity(model, layers, lat, enc, pooled, img_ids, txt_ids, sigmas, guidance=3.5, alt=None):
    """Run the Euler / flow-matching loop ``x += (sigma[i+1] - sigma[i]) * v(x, sigma[i])`` on the GPU model and on the CPU
    twin from identical inputs; per step: PSNR / relative L2 of the predicted update v and of the latent after the step."""
    ref_model = Ref(model, layers)
    gd = None if model.guidance_embed is None else torch.tensor([guidance])
    xg, xc = lat.clone(), lat.clone()
    out = []
    with torch.no_grad():
        for i in range(len(sigmas) - 1):
            t = torch.tensor([sigmas[i]], dtype=torch.float32)
            vg = model(xg.cuda().bfloat16()[None], enc.cuda().bfloat16()[None], pooled.cuda().bfloat16(), t.cuda(), img_ids.cuda(),
                       txt_ids.cuda(), None if gd is None else gd.cuda())[0].float().cpu()
            vc = Ref.forward(ref_model, xc, enc, pooled, t, img_ids, txt_ids, gd)
            va = None
            if alt is not None:  # a second GPU forward of the SAME input through another valid op sequence (sensitivity probe)
                with alt():
                    va = model(xg.cuda().bfloat16()[None], enc.cuda().bfloat16()[None], pooled.cuda().bfloat16(), t.cuda(),
                               img_ids.cuda(), txt_ids.cuda(), None if gd is None else gd.cuda())[0].float().cpu()
            d = sigmas[i + 1] - sigmas[i]
            xg = r16(xg + r16(torch.tensor(d) * vg))   # the latent lives in the model dtype between steps on both sides
            xc = r16(xc + r16(torch.tensor(d) * vc))
            pv, rv = psnr_rel(vg, vc)
            px, rx = psnr_rel(xg, xc)
            rec = {"step": i, "sigma": float(sigmas[i]), "v_psnr_db": pv, "v_rel_l2": rv, "latent_psnr_db": px, "latent_rel_l2": rx,
                   "finite": bool(torch.isfinite(vg).all())}
            if va is not None:
                rec["gpu_alt_v_psnr_db"], rec["gpu_alt_v_rel_l2"] = psnr_rel(va, vg)
            out.append(rec)
    return out
