#!/usr/bin/env python3
"""GPU-recorded known answers of the quantiser (the quantiser divides with x * v_rcp_f32(smooth) since round 5 -- the reference's __fdividef
form -- so the IEEE oracle no longer pins its codes bit for bit; the envelope test accepts a +-1 flip on < 1e-3 of the elements, which would also let a
low-rate off-by-one regression through).  This script runs the PRODUCT quantiser on seeded inputs on an MI355X and records codes and scales; the fixture
(tests/golden/gpu_quantize_kat.npz) pins the kernel against itself across builds: tests/test_gpu_parity.py::test_quantiser_matches_its_recorded_gpu_answers.

    gpurun -- python tools/make_gpu_golden.py gpurun_out/gpu_golden/gpu_quantize_kat.npz      # then copy to tests/golden/

Recorded with: the stand-alone quantiser (quantize_kernel_v2, rank 32, and the MULTI slabs at rank 128), bf16 and fp16; inputs from the oracle's seeded generators
(numpy: identical on every host).  The record also holds how many codes differ from the IEEE oracle, for the reader."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = [("bf16", 256, 512, 32, 101), ("fp16", 256, 512, 32, 102), ("bf16", 300, 384, 128, 103), ("fp16", 77, 1024, 16, 104)]


def run_case(dtype, M, K, R, seed):
    from helpers import make_module, t16
    from nunchaku_amd import layout
    from oracle import svdq_oracle as O

    L = O.make_svdq_layer(K, 128, R, seed=seed, dtype=dtype, cheap=True)
    x = O.make_activations(M, K, seed=seed, dtype=dtype)
    mod = make_module(L, dtype)
    qx, asc, _ = mod.quantize(t16(x, dtype))
    codes = layout.unpack_act(qx, K).cpu().numpy().astype(np.int8)
    scales = layout.unpack_scales(asc, qx.shape[0]).view(__import__("torch").int16).cpu().numpy()
    q_ref = O.quantize_w4a4_act_fuse_lora(x, L["smooth"], L["proj_down"], dtype)[0]
    return codes, scales, int((codes.astype(np.int32) != q_ref).sum())


if __name__ == "__main__":
    out = sys.argv[1]
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    rec = {}
    for dtype, M, K, R, seed in CASES:
        codes, scales, ndiff = run_case(dtype, M, K, R, seed)
        key = f"{dtype}_{M}_{K}_{R}_{seed}"
        rec[key + "_codes"], rec[key + "_scales"], rec[key + "_ieee_diff"] = codes, scales, np.int64(ndiff)
        print(key, codes.shape, scales.shape, "codes off the IEEE oracle:", ndiff)
    np.savez_compressed(out, **rec)
    print("wrote", out)
