"""Time the GELU_QUANT GEMM (fc1 + GELU + lora-down + requantise) alone, under SVDQ_GEMM_DEBUG ablation bits."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_kernels import rand_layer, timeit
from nunchaku_amd.ops.gemm import svdq_gemm_w4a4_cuda

dbgs = [int(a) for a in sys.argv[1:]] or [0, 16, 32, 1]
for M in (4096, 512):
    fc1, fc2 = rand_layer(3072, 12288), rand_layer(12288, 3072, act_unsigned=True)
    x = torch.randn(M, 3072, device="cuda", dtype=torch.bfloat16)
    qx, asc, la = fc1.quantize(x)
    M_pad = qx.shape[0]
    qh = torch.empty(M_pad, 12288 * 3 // 4, dtype=torch.uint8, device="cuda")
    sh = torch.empty(12288 // 64, M_pad, dtype=torch.bfloat16, device="cuda")
    lh = torch.empty(M_pad, 32, dtype=torch.float32, device="cuda")
    out = torch.empty(M, 12288, dtype=torch.bfloat16, device="cuda")
    def gelu():
        svdq_gemm_w4a4_cuda(act=qx, wgt=fc1.qweight, qout=qh, ascales=asc, wscales=fc1.wscales, oscales=sh, lora_act_in=la,
                            lora_up=fc1.proj_up, lora_down=fc2.proj_down, lora_act_out=lh, bias=fc1.bias,
                            smooth_factor=fc2.smooth_factor)
    def plain():
        svdq_gemm_w4a4_cuda(act=qx, wgt=fc1.qweight, out=out, ascales=asc, wscales=fc1.wscales, lora_act_in=la,
                            lora_up=fc1.proj_up, bias=fc1.bias)
    os.environ["SVDQ_GEMM_DEBUG"] = "0"
    print(f"M={M} plain 3072->12288: {timeit(plain, 20)*1e6:.1f} us")
    for d in dbgs:
        os.environ["SVDQ_GEMM_DEBUG"] = str(d)
        print(f"M={M} gelu_quant debug={d}: {timeit(gelu, 20)*1e6:.1f} us")
    os.environ["SVDQ_GEMM_DEBUG"] = "0"
