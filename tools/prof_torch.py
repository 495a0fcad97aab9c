"""torch.profiler view of one bench step: which ops launch the __amd_rocclr_copyBuffer / fill kernels."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
from nunchaku_amd.models.flux import FluxTransformerAMD
dev = torch.device("cuda")
model = FluxTransformerAMD(num_layers=2, num_single_layers=2, device=dev).init_synthetic_(seed=0).eval()
side, t_txt = 64, 512
lat = torch.randn(1, side * side, 64, device=dev).bfloat16(); enc = torch.randn(1, t_txt, 4096, device=dev).bfloat16()
pooled = torch.randn(1, 768, device=dev).bfloat16()
img_ids = torch.zeros(side * side, 3, device=dev); txt_ids = torch.zeros(t_txt, 3, device=dev)
t = torch.tensor([0.5], device=dev); g = torch.tensor([3.5], device=dev)
with torch.no_grad():
    for _ in range(2): model(lat, enc, pooled, t, img_ids, txt_ids, g)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        model(lat, enc, pooled, t, img_ids, txt_ids, g); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60))
