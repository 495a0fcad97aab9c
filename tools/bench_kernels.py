"""Kernel-level timing of the hot path at FLUX.1 shapes (SURVEY.md section 8d metric 1).

    python tools/bench_kernels.py [--iters 20] [--json gpurun_out/kernels.json]
Random int4 codes / scales (no oracle involved); HIP events on the launch stream.
"""

import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nunchaku_amd.models.linear import SVDQW4A4Linear  # noqa: E402
from nunchaku_amd.ops.fused import fused_gelu_mlp  # noqa: E402

INT8_PEAK_TOPS = 256 * 4 * 1024 * 2 * 2.4e9 / 1e12  # 256 CU x 4 SIMD x 1024 MAC/clk x 2 x 2.4 GHz = 5033


def rand_layer(K, N, R=32, act_unsigned=False):
    m = SVDQW4A4Linear(K, N, rank=R, act_unsigned=act_unsigned, device="cuda")
    with torch.no_grad():
        m.qweight.copy_(torch.randint(-128, 128, m.qweight.shape, dtype=torch.int8, device="cuda"))
        m.wscales.copy_((torch.rand_like(m.wscales, dtype=torch.float32) * 0.01 + 0.005).to(torch.bfloat16))
        m.bias.copy_(torch.randn_like(m.bias, dtype=torch.float32) * 0.1)
        m.smooth_factor.copy_((torch.rand_like(m.smooth_factor, dtype=torch.float32) + 0.5).to(torch.bfloat16))
        m.proj_down.copy_(torch.randn_like(m.proj_down, dtype=torch.float32) * 0.02)
        m.proj_up.copy_(torch.randn_like(m.proj_up, dtype=torch.float32) * 0.02)
    m.repack_()  # random nibbles in the checkpoint layout -> FP6 operand image
    return m


def timeit(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--json", default=None)
    ap.add_argument("--shape", type=int, nargs=3, default=None, help="M K N: time a single shape")
    ap.add_argument("--zero", action="store_true", help="all-zero codes (power/clock experiment)")
    args = ap.parse_args()
    rows = []
    shapes = [(3072, 9216), (3072, 3072), (3072, 12288), (12288, 3072)]
    Ms = (512, 4096, 4608)
    if args.shape:
        Ms, shapes = (args.shape[0],), [(args.shape[1], args.shape[2])]
    for M in Ms:
        for (K, N) in shapes:
            lin = rand_layer(K, N, act_unsigned=False)
            x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
            qx, asc, la = lin.quantize(x)
            if args.zero:
                qx.zero_(); lin.qweight.data.zero_()
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            tq = timeit(lambda: lin.quantize(x), args.iters)
            tg = timeit(lambda: lin.forward_quant(qx, asc, la, out), args.iters)
            ops = 2.0 * M * N * K + 2.0 * M * N * 32
            row = {
                "M": M, "K": K, "N": N, "quantize_us": tq * 1e6, "gemm_us": tg * 1e6,
                "gemm_TOPS": ops / tg / 1e12, "gemm_frac_int8_peak": ops / tg / 1e12 / INT8_PEAK_TOPS,
                "quant_GBps": (M * K * 2 + M * K * 0.75) / tq / 1e9,
            }
            rows.append(row)
            print(json.dumps(row), flush=True)
            del lin
    # fused MLP (fc1 with GELU+requant+lora-down epilogue, fc2 unsigned)
    for M in (() if args.shape else (4096, 4608)):
        fc1, fc2 = rand_layer(3072, 12288), rand_layer(12288, 3072, act_unsigned=True)
        x = torch.randn(1, M, 3072, device="cuda", dtype=torch.bfloat16)
        t = timeit(lambda: fused_gelu_mlp(x, fc1, fc2), args.iters)
        ops = 2.0 * M * 3072 * 12288 * 2
        row = {"fused_gelu_mlp_M": M, "us": t * 1e6, "TOPS": ops / t / 1e12, "frac_int8_peak": ops / t / 1e12 / INT8_PEAK_TOPS}
        rows.append(row)
        print(json.dumps(row), flush=True)
    if args.json:
        os.makedirs(os.path.dirname(args.json), exist_ok=True)
        json.dump(rows, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
