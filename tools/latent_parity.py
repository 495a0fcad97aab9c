#!/usr/bin/env python3
"""Step-level parity artefact: N-step latent PSNR of the GPU path against the
oracle-backed CPU twin on identical seeds and latents, at the REAL FLUX.1 width.

    python tools/latent_parity.py            # on a MI355X box -> gpurun_out/latent_psnr.json (copy to profiles/r3_latent_psnr.json)

Model: hidden 3072 = 24 heads x 128, MLP 12288, 1 joint + 1 single block (every operator of a FLUX step appears once
per stream), rank-32 SVDQuant layers built from a dense Gaussian weight by smoothing + rank-32 (randomised) SVD + 4-bit
residual quantisation (oracle.make_svdq_layer(cheap=False): checkpoint-like code / scale statistics), AWQ W4A16 modulation
projections.  Two geometries, two Euler steps each (sigma 1.0 -> 0.5 -> 0.0):
  dev      256 image + 256 text tokens, guidance embedding on   (FLUX.1-dev, BASELINE configs 3/4, reduced token count)
  schnell  1024 image + 512 text tokens, no guidance embedding  (FLUX.1-schnell 512^2, BASELINE config 2, full token count)
GPU side: the product path (HIP kernels through the C ABI, fused norm / grouped launches as in bench.py).  CPU side:
tests/flux_ref.py (numpy oracle for every quantised operator, fp32 torch with the reference's 16-bit rounding points for
the rest).  The reference CUDA path itself cannot run here: this is the parity evidence that is
reproducible in this environment; the oracle's arithmetic is parity-unpinned.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


import contextlib  # noqa: E402


@contextlib.contextmanager
def torch_op_path():
    """the reference's torch-op sequence for LayerNorm / modulation / gated residuals and torch SDPA instead of the fused
    passes and this library's attention kernel: another VALID 16-bit op sequence of the same model"""
    from nunchaku_amd.models.flux import FluxAttentionAMD, FluxTransformerAMD

    saved = (FluxTransformerAMD.fused_norm, FluxAttentionAMD.attention_impl)
    FluxTransformerAMD.fused_norm, FluxAttentionAMD.attention_impl = False, "sdpa"
    try:
        yield
    finally:
        FluxTransformerAMD.fused_norm, FluxAttentionAMD.attention_impl = saved


def run(name, side, t_txt, guidance, seed, steps=(1.0, 0.5, 0.0), lowrank_energy=0.0):
    from nunchaku_amd.models.flux import FluxTransformerAMD
    from tests.flux_ref import euler_parity, fill_model_, synthetic_inputs

    t0 = time.time()
    model = FluxTransformerAMD(num_layers=1, num_single_layers=1, dim=3072, heads=24, in_channels=64, joint_attention_dim=4096,
                               pooled_projection_dim=768, guidance_embeds=guidance, device="cuda")
    layers = fill_model_(model, seed=seed, realistic=True, lowrank_energy=lowrank_energy)
    model.eval()
    t_build = time.time() - t0
    lat, enc, pooled, img_ids, txt_ids = synthetic_inputs(side, t_txt, 4096, 768, seed=seed + 1)
    t0 = time.time()
    res = euler_parity(model, layers, lat, enc, pooled, img_ids, txt_ids, list(steps), alt=torch_op_path)
    weights = ("dense N(0, 0.02^2)" if lowrank_energy == 0 else f"{lowrank_energy:.0%} of the energy in a rank-32 component + N(0, 0.02^2)")
    rec = {"config": name, "hidden": 3072, "heads": 24, "blocks": "1 joint + 1 single", "image_tokens": side * side, "text_tokens": t_txt,
           "guidance_embeds": guidance, "weights": f"make_svdq_layer(cheap=False, svd='randomized') on a weight with {weights}; rank 32, seed {seed}",
           "dtype": "bf16", "sigmas": list(steps), "steps": res, "build_s": round(t_build, 1), "run_s": round(time.time() - t0, 1)}
    print(json.dumps(rec), flush=True)
    del model
    torch.cuda.empty_cache()
    return rec


def run_qwen_block(t_img=256, t_txt=256, seed=21, lowrank_energy=0.95):
    """One Qwen-Image dual-stream block at the REAL width (hidden 3072 = 24 heads x 128, MLP 12288; BASELINE config 5): the GPU
    path (fused QKV + RMSNorm + Qwen rotary epilogue, svdq attention, grouped launches) against the CPU twin of the reference's
    op sequence (tests/test_gpu_qwenimage.py:BlockTwin, numpy oracle for every quantised operator).  PSNR of both streams'
    block outputs (a block output is what the next block consumes: the Qwen analogue of the latent of a 1+1-block FLUX step)."""
    from nunchaku_amd.models.qwenimage import NunchakuQwenImageTransformerBlock
    from tests.flux_ref import psnr_rel
    from tests.test_gpu_qwenimage import BlockTwin, _block_inputs, _fill

    t0 = time.time()
    block = NunchakuQwenImageTransformerBlock(3072, 24, 128, device="cuda").eval()
    layers = _fill(block, seed=seed, lowrank_energy=lowrank_energy)
    t_build = time.time() - t0
    hidden, enc, temb, img_f, txt_f = _block_inputs(3072, t_img, t_txt, seed=seed + 1)
    cs = lambda f: torch.stack([f.real.float(), f.imag.float()], dim=-1)
    t0 = time.time()
    with torch.no_grad():
        e_ref, h_ref = BlockTwin(block, layers).forward(hidden, enc, temb, cs(img_f), cs(txt_f))
        e, h = block(hidden.cuda().bfloat16()[None], enc.cuda().bfloat16()[None], None, temb.cuda().bfloat16(), (img_f.cuda(), txt_f.cuda()))
    res = []
    for name, got, ref in (("text stream", e[0].float().cpu(), e_ref), ("image stream", h[0].float().cpu(), h_ref)):
        psnr, rel = psnr_rel(got, ref)
        res.append({"output": name, "latent_psnr_db": round(float(psnr), 2), "rel_l2": float(rel)})
    rec = {"config": f"Qwen-Image block ({t_img} image + {t_txt} text tokens), low-rank-dominated weights", "hidden": 3072, "heads": 24,
           "blocks": "1 dual-stream block", "image_tokens": t_img, "text_tokens": t_txt, "dtype": "bf16", "steps": res,
           "weights": f"make_svdq_layer(cheap=False, svd='randomized'), {lowrank_energy:.0%} of the energy in a rank-32 component; seed {seed}",
           "build_s": round(t_build, 1), "run_s": round(time.time() - t0, 1)}
    print(json.dumps(rec), flush=True)
    del block
    torch.cuda.empty_cache()
    return rec


def main():
    out = os.path.join(ROOT, "gpurun_out", "latent_psnr.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    # (1) the regime SVDQuant is built for: the 16-bit low-rank branch carries most of the energy.  A forward pass is well
    #     conditioned, so GPU vs CPU twin measures the IMPLEMENTATION: this is the gated number.
    recs = [run("dev geometry (256 + 256 tokens), low-rank-dominated weights", 16, 256, True, seed=11, lowrank_energy=0.95)]
    recs.append(run_qwen_block())
    if "--quick" not in sys.argv:
        recs.append(run("schnell 512^2 geometry (1024 + 512 tokens), low-rank-dominated weights", 32, 512, False, seed=12, steps=(1.0, 0.5),
                        lowrank_energy=0.95))
        # (2) unstructured Gaussian weights: rank 32 absorbs ~30 % of the energy, W4A4 noise is ~10 % of the signal and a
        #     forward pass is CHAOTIC in the last bit of its 16-bit intermediates (+-1 code flips of the 4-bit activations):
        #     two valid op sequences on the same GPU differ as much from each other ("gpu_alt_*") as either does from the
        #     CPU twin.  Reported, not gated -- it bounds what a PSNR between two W4A4 implementations can mean.
        recs.append(run("dev geometry (256 + 256 tokens), dense Gaussian weights", 16, 256, True, seed=11))
    gated = [s["latent_psnr_db"] for r in recs if "low-rank" in r["config"] for s in r["steps"]]
    worst = min(gated)
    json.dump({"tool": "tools/latent_parity.py", "gate_db": 50.0, "gated_runs": "low-rank-dominated weights", "worst_latent_psnr_db": worst,
               "runs": recs}, open(out, "w"), indent=1)
    print(f"worst gated latent PSNR {worst:.1f} dB -> {out}")
    return 0 if worst >= 50.0 else 1


if __name__ == "__main__":
    sys.exit(main())
