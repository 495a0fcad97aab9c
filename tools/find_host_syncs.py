#!/usr/bin/env python3
"""Where does a denoise step make the host wait for the GPU?  Runs a few FLUX-shaped steps with torch's sync debug mode (warnings with the Python
stack of every synchronising torch call) and prints the host time a step takes to ENQUEUE against the GPU time it takes to run.

    python tools/find_host_syncs.py [--config dev1024|qwen1024] [--layers J S]
"""
import argparse
import os
import sys
import time
import traceback
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="dev1024")
    ap.add_argument("--layers", type=int, nargs=2, default=(19, 38))
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    from nunchaku_amd.models.flux import FluxTransformerAMD

    model = FluxTransformerAMD(num_layers=args.layers[0], num_single_layers=args.layers[1], guidance_embeds=True, device=dev)
    model.init_synthetic_(seed=0)
    model.eval()
    g = torch.Generator(device=dev).manual_seed(0)
    t_img, t_txt, side = 4096, 512, 64
    lat = torch.randn(1, t_img, 64, device=dev, generator=g).bfloat16()
    enc = torch.randn(1, t_txt, 4096, device=dev, generator=g).bfloat16()
    pooled = torch.randn(1, 768, device=dev, generator=g).bfloat16()
    img_ids = torch.zeros(t_img, 3, device=dev)
    img_ids[:, 1] = torch.arange(side, device=dev).repeat_interleave(side)
    img_ids[:, 2] = torch.arange(side, device=dev).repeat(side)
    txt_ids = torch.zeros(t_txt, 3, device=dev)
    guidance = torch.tensor([3.5], device=dev)
    sigmas = torch.linspace(1.0, 0.0, 9, device=dev)

    def step(i, x):
        with torch.no_grad():
            v = model(x, enc, pooled, sigmas[i].reshape(1), img_ids, txt_ids, guidance)
        return x + (sigmas[i + 1] - sigmas[i]).to(v.dtype) * v

    for i in range(2):
        lat = step(i, lat)
    torch.cuda.synchronize()
    # host enqueue time against GPU time
    for i in range(2, 5):
        t0 = time.perf_counter()
        lat = step(i, lat)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"step {i}: host returned after {1e3 * (t1 - t0):.1f} ms, GPU done after {1e3 * (t2 - t0):.1f} ms")
    # synchronising torch calls of one step
    seen = {}

    def show(message, category, filename, lineno, file=None, line=None):
        stack = [f for f in traceback.extract_stack() if "nunchaku_amd" in f.filename or f.filename.endswith("find_host_syncs.py")]
        key = tuple((f.filename, f.lineno) for f in stack[-3:])
        seen[key] = seen.get(key, 0) + 1
        if seen[key] == 1:
            print("SYNC:", str(message).split("\n")[0], "<-", " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno} {f.line}" for f in reversed(stack[-3:])))

    warnings.showwarning = show
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode("warn")
    lat = step(5, lat)
    torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    print("synchronising call sites in one step:", sum(seen.values()))


if __name__ == "__main__":
    main()
