"""Time the epilogue variants of the GEMM at FLUX shapes (M=4096): plain, GELU->requant, RMSNorm+RoPE."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.bench_kernels import rand_layer, timeit
from nunchaku_amd.ops.gemm import svdq_gemm_w4a4_cuda
from nunchaku_amd import layout

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
it = 30
fc1, fc2 = rand_layer(3072, 12288), rand_layer(12288, 3072, act_unsigned=True)
x = torch.randn(M, 3072, device="cuda", dtype=torch.bfloat16)
qx, asc, la = fc1.quantize(x)
M_pad = qx.shape[0]
out = torch.empty(M, 12288, device="cuda", dtype=torch.bfloat16)
t_plain = timeit(lambda: fc1.forward_quant(qx, asc, la, out), it)
qh = torch.empty(layout.act_image_shape(M_pad, 12288), dtype=torch.uint8, device="cuda")
sh = torch.empty(12288 // 64, M_pad, dtype=torch.bfloat16, device="cuda")
lh = torch.zeros(M_pad, 32, dtype=torch.float32, device="cuda")
def gelu():
    svdq_gemm_w4a4_cuda(act=qx, wgt=fc1.qweight, qout=qh, ascales=asc, wscales=fc1.wscales, oscales=sh, lora_act_in=la,
                        lora_up=fc1.proj_up, lora_down=fc2.proj_down, lora_act_out=lh, bias=fc1.bias, smooth_factor=fc2.smooth_factor)
t_gelu = timeit(gelu, it)
def gelu_nolora():
    svdq_gemm_w4a4_cuda(act=qx, wgt=fc1.qweight, qout=qh, ascales=asc, wscales=fc1.wscales, oscales=sh, lora_act_in=la,
                        lora_up=fc1.proj_up, bias=fc1.bias, smooth_factor=fc2.smooth_factor)
t_gelu2 = timeit(gelu_nolora, it)
print(json.dumps({"M": M, "fc1_plain_us": t_plain * 1e6, "fc1_gelu_quant_us": t_gelu * 1e6, "fc1_gelu_quant_no_loradown_us": t_gelu2 * 1e6}))
qkv = rand_layer(3072, 9216)
qx, asc, la = qkv.quantize(x)
out = torch.empty(M, 9216, device="cuda", dtype=torch.bfloat16)
t_plain = timeit(lambda: qkv.forward_quant(qx, asc, la, out), it)
rot = torch.randn(M_pad, 128, device="cuda", dtype=torch.float32)
nq = torch.ones(128, device="cuda", dtype=torch.bfloat16)
def rope():
    svdq_gemm_w4a4_cuda(act=qx, wgt=qkv.qweight, out=out, ascales=asc, wscales=qkv.wscales, lora_act_in=la, lora_up=qkv.proj_up,
                        bias=qkv.bias, norm_q=nq, norm_k=nq, rotary_emb=rot)
t_rope = timeit(rope, it)
print(json.dumps({"M": M, "qkv_plain_us": t_plain * 1e6, "qkv_rmsnorm_rope_us": t_rope * 1e6}))
