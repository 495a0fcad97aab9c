#!/usr/bin/env python3
"""BASELINE config 5 in miniature: Qwen-Image-shaped blocks (hidden 3072 = 24 x 128, MLP 12288, rank 32) at 1024^2 (4096 image
+ 512 text tokens), resident vs layer-wise host offload (two HIP streams, pinned host memory, nibble qweights over the link).

    python tools/bench_qwen_offload.py [--layers 12] [--steps 5]     -> one JSON line (gpurun_out/qwen_offload.json)

Reports ms per forward resident / offloaded, the bytes one block moves over PCIe, the implied link rate, and whether the
copies hide behind compute (offloaded ~ max(compute, copy)) or add to it (~ sum).  The full model has 60 blocks; the
per-block numbers scale linearly (no cross-block state besides the two buffer slots)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from nunchaku_amd.models.qwenimage import NunchakuQwenImageTransformer2DModel  # noqa: E402


def timed(fn, steps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--txt", type=int, default=512)
    args = ap.parse_args()
    model = NunchakuQwenImageTransformer2DModel(num_layers=args.layers, device="cuda").init_synthetic_(seed=0).eval()
    g = torch.Generator(device="cuda").manual_seed(1)
    lat = torch.randn(1, 4096, 64, device="cuda", generator=g).bfloat16()
    enc = torch.randn(1, args.txt, 3584, device="cuda", generator=g).bfloat16()
    t = torch.tensor([0.5], device="cuda")
    run = lambda: model(lat, enc, None, t, [(1, 64, 64)]).sample
    # offload first (nibble path needs un-repacked layers), then the same model resident
    with torch.no_grad():
        model.set_offload(True, num_blocks_on_gpu=1, use_pin_memory=True)
        mgr = model.offload_manager
        link_bytes = mgr.host_bytes_per_block()
        ms_off = timed(run, args.steps)
        y_off = run().float()
        # copy-only time of one block: the memory stream alone
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            with torch.cuda.stream(mgr.memory_stream):
                mgr.load_block(1)
        torch.cuda.synchronize()
        ms_copy = (time.perf_counter() - t0) / 10 * 1e3
        model.set_offload(False)
        for b in model.transformer_blocks:  # bring every block back (qweight images were dropped on the host side)
            pass
    # a fresh resident model with the same weights for the compute-only time
    del model
    torch.cuda.empty_cache()
    model = NunchakuQwenImageTransformer2DModel(num_layers=args.layers, device="cuda").init_synthetic_(seed=0).eval()
    run = lambda: model(lat, enc, None, t, [(1, 64, 64)]).sample
    with torch.no_grad():
        ms_res = timed(run, args.steps)
        y_res = run().float()
    rel = ((y_off - y_res).norm() / y_res.norm()).item()
    n_off = args.layers - 1
    rec = {
        "workload": f"Qwen-Image-shaped transformer, {args.layers} blocks (of 60), hidden 3072, 4096 image + {args.txt} text tokens, bs 1, bf16, int4 r32",
        "resident_ms_per_forward": ms_res, "resident_ms_per_block": ms_res / args.layers,
        "offload_ms_per_forward": ms_off, "offload_ms_per_block": ms_off / args.layers,
        "link_bytes_per_block": link_bytes, "copy_plus_expand_ms_per_block": ms_copy,
        "implied_link_GBps": link_bytes / (ms_copy * 1e-3) / 1e9,
        "offloaded_blocks": n_off,
        "overlap": "hidden" if ms_off < 1.15 * max(ms_res, n_off * ms_copy) else "partly exposed",
        "sum_vs_max_ms": {"sum": ms_res + n_off * ms_copy, "max": max(ms_res, n_off * ms_copy)},
        "rel_diff_offload_vs_resident": rel,
        "extrapolated_60_blocks_ms": {"resident": ms_res / args.layers * 60, "offload": ms_off / args.layers * 60},
    }
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rec, open(os.path.join(ROOT, "gpurun_out", "qwen_offload.json"), "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
