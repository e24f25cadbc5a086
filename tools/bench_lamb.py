"""Developer micro-benchmark: the fused LAMB + EMA step on the ViT-H parameter set (355.8 M parameters, 982 tensors), HIP-event
time per step and GB/s at 54 B per parameter.  OCTIC_LIB selects a library build (A/B of kernel variants)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octic_vits_amd.deit_models import create_model
from octic_vits_amd.train import FusedLamb, param_groups_weight_decay, library_gemm_layers

torch.manual_seed(0)
model = create_model("hybrid_deit_huge_patch14", num_classes=1000, drop_path_rate=0.5, img_size=224).cuda()
opt = FusedLamb(param_groups_weight_decay(model, 0.02, model.no_weight_decay()), ema_decay=0.99996, shadow_layers=library_gemm_layers(model))
n = sum(p.numel() for p in opt.params)
for p in opt.params:
    p.grad = torch.randn_like(p) * 1e-3
best = 1e9
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        opt.step()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10)
print(f"lamb step ({os.environ.get('OCTIC_LIB', 'default lib')}): {best:.3f} ms for {n / 1e6:.1f} M parameters = {n * 54 / best / 1e9:.0f} GB/s at 54 B/param")
