"""Generate tests/golden/*.npz from the REFERENCE's own packer (run in the build container only).

The reference (`/root/reference`) cannot be imported as a package (its `__init__` needs the CUDA
extension and diffusers), so `nunchaku/lora/flux/packer.py` and `nunchaku/models/embeddings.py`
(`pack_rotemb`) are loaded by path behind stub parent packages.  The fixtures hold
(logical tensor, reference-packed tensor) pairs; `tests/test_layouts.py` checks the oracle's
layout codecs -- and, on the GPU, the HIP repack kernels -- against them.  The reference does not
exist on the GPU box, so only the committed .npz files travel.

    python tools/make_golden.py
"""

import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_packer():
    for pkg in ("nunchaku", "nunchaku.lora", "nunchaku.lora.flux"):
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m
    _load("nunchaku.utils", f"{REF}/nunchaku/utils.py")
    _load("nunchaku.lora.flux.utils", f"{REF}/nunchaku/lora/flux/utils.py")
    return _load("nunchaku.lora.flux.packer", f"{REF}/nunchaku/lora/flux/packer.py")


def load_reference_pack_rotemb():
    src = open(f"{REF}/nunchaku/models/embeddings.py").read()
    # only the pure-torch helper is needed; the module header imports diffusers
    start = src.index("def pack_rotemb")
    ns = {"torch": torch}
    exec(compile(src[start:], "embeddings.pack_rotemb", "exec"), ns)
    return ns["pack_rotemb"]


def main():
    os.makedirs(OUT, exist_ok=True)
    packer_mod = load_reference_packer()
    packer = packer_mod.NunchakuWeightPacker(bits=4)
    rng = np.random.default_rng(20260922)

    # --- qweight: N x K int4 -------------------------------------------------
    for (N, K) in ((128, 128), (256, 384)):
        q = rng.integers(-8, 8, size=(N, K), dtype=np.int32)
        packed = packer.pack_weight(torch.from_numpy(q.copy())).numpy()
        np.savez_compressed(f"{OUT}/qweight_{N}x{K}.npz", logical=q.astype(np.int8), packed=packed)

    # --- wscales [G, N] (packer takes [N, G]) and bias/smooth [N] -------------
    for (G, N) in ((2, 128), (6, 256)):
        s = rng.standard_normal((N, G)).astype(np.float32)
        st = torch.from_numpy(s).to(torch.bfloat16)
        packed = packer.pack_scale(st.clone(), group_size=64)
        assert packed.shape == (G, N)
        np.savez_compressed(
            f"{OUT}/wscales_{G}x{N}.npz",
            logical=st.float().numpy().T.copy(),  # [G, N]
            packed=packed.float().numpy(),
        )
    v = torch.from_numpy(rng.standard_normal(256).astype(np.float32)).to(torch.bfloat16)
    pv = packer.pack_scale(v.clone().view(-1, 1), group_size=-1)
    np.savez_compressed(f"{OUT}/vec_256.npz", logical=v.float().numpy(), packed=pv.float().numpy())

    # --- low-rank weights ------------------------------------------------------
    # up: logical [N, R] ; down: logical [R, K] (packer's input), stored [K, R]
    N, K, R = 128, 192, 32
    up = torch.from_numpy(rng.standard_normal((N, R)).astype(np.float32)).to(torch.bfloat16)
    pup = packer.pack_lowrank_weight(up.clone(), down=False)
    assert pup.shape == (N, R)
    down = torch.from_numpy(rng.standard_normal((R, K)).astype(np.float32)).to(torch.bfloat16)
    pdown = packer.pack_lowrank_weight(down.clone(), down=True)
    assert pdown.shape == (K, R), pdown.shape
    # the reference's own inverse must agree with its pack (sanity of our reading)
    assert torch.equal(packer.unpack_lowrank_weight(pup, down=False), up)
    assert torch.equal(packer.unpack_lowrank_weight(pdown, down=True), down)
    np.savez_compressed(
        f"{OUT}/lowrank_{N}_{K}_{R}.npz",
        up_logical=up.float().numpy(), up_packed=pup.float().numpy(),
        down_logical=down.float().numpy(), down_packed=pdown.float().numpy(),
    )

    # --- rotary embedding packing ----------------------------------------------
    pack_rotemb = load_reference_pack_rotemb()
    M, D = 32, 128
    rot = torch.from_numpy(rng.standard_normal((1, M, D // 2, 1, 2)).astype(np.float32))
    prot = pack_rotemb(rot)
    np.savez_compressed(
        f"{OUT}/rotemb_{M}.npz", logical=rot.numpy().reshape(M, D // 2, 2), packed=prot.numpy().reshape(M, D)
    )
    # --- AWQ W4A16 (tinychat) weights: the reference's own converter -------------------------
    tc = _load("tinychat_utils_ref", f"{REF}/nunchaku/models/text_encoders/tinychat_utils.py")
    for (N, K) in ((16, 128), (64, 256)):
        w = torch.from_numpy(rng.standard_normal((N, K)).astype(np.float32)).to(torch.bfloat16)
        g = w.float().view(N, K // 64, 64)
        lo, hi = g.amin(-1), g.amax(-1)
        scale = ((hi - lo) / 15).clamp_min(1e-8).to(torch.bfloat16)
        zero = (-lo).to(torch.bfloat16)
        pw, ps, pz = tc.convert_to_tinychat_w4x16y16_linear_weight(w.clone(), scale, zero, group_size=64)
        # logical codes as the converter computes them (tinychat_utils.py:180)
        q = (g + zero.float()[..., None]).div(scale.float()[..., None]).round().view(N, K).to(torch.int32)
        assert torch.equal(tc.pack_w4(q), pw)
        np.savez_compressed(
            f"{OUT}/awq_{N}x{K}.npz", weight=w.float().numpy(), codes=q.numpy().astype(np.uint8),
            packed_int16=pw.numpy(), scales=ps.float().numpy(), zeros=pz.float().numpy(),
        )
    # --- FLUX checkpoint key conversion: the reference's convert_flux_state_dict on every legacy key name ----------
    src = open(f"{REF}/nunchaku/models/transformers/transformer_flux_v2.py").read()
    ns = {"torch": torch}
    exec(compile(src[src.index("def convert_flux_state_dict"):], "convert_flux_state_dict", "exec"), ns)
    svdq = ["qweight", "wscales", "bias", "lora_down", "lora_up", "smooth", "smooth_orig", "wtscale", "wcscales"]
    awq = ["qweight", "wscales", "wzeros", "bias"]
    keys = []
    for i in (0, 18):
        b = f"transformer_blocks.{i}."
        for mod in ("qkv_proj", "qkv_proj_context", "out_proj", "out_proj_context", "mlp_fc1", "mlp_fc2", "mlp_context_fc1", "mlp_context_fc2"):
            keys += [f"{b}{mod}.{p}" for p in svdq]
        for mod in ("norm1.linear", "norm1_context.linear"):
            keys += [f"{b}{mod}.{p}" for p in awq]
        keys += [f"{b}{n}.weight" for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k")]
    for i in (0, 37):
        b = f"single_transformer_blocks.{i}."
        for mod in ("qkv_proj", "out_proj", "mlp_fc1", "mlp_fc2"):
            keys += [f"{b}{mod}.{p}" for p in svdq]
        keys += [f"{b}norm.linear.{p}" for p in awq] + [f"{b}norm_q.weight", f"{b}norm_k.weight"]
    for top in ("x_embedder", "context_embedder", "proj_out", "norm_out.linear", "time_text_embed.timestep_embedder.linear_1",
                "time_text_embed.timestep_embedder.linear_2", "time_text_embed.guidance_embedder.linear_1",
                "time_text_embed.guidance_embedder.linear_2", "time_text_embed.text_embedder.linear_1",
                "time_text_embed.text_embedder.linear_2"):
        keys += [f"{top}.weight", f"{top}.bias"]
    conv = ns["convert_flux_state_dict"]({k: i for i, k in enumerate(keys)})
    inv = {v: k for k, v in conv.items()}
    import json
    json.dump({k: inv[i] for i, k in enumerate(keys)}, open(f"{OUT}/flux_keys.json", "w"), indent=0, sort_keys=True)
    print("golden fixtures written to", os.path.abspath(OUT))


if __name__ == "__main__":
    main()
