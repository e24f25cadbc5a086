"""Which gradients differ between immediate and postponed finish launches (two distinct standard blocks)?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from octic_vits_amd import ops as o
from octic_vits_amd.vit import Layer_scale_init_Block
torch.manual_seed(0)
blk = torch.nn.Sequential(*[Layer_scale_init_Block(dim=256, num_heads=4, qkv_bias=True, init_values=0.5, drop_path=0.0) for _ in range(2)]).cuda().train()
x = torch.randn(6, 65, 256, device="cuda")
cot = torch.randn(6, 65, 256, device="cuda")
res = {}
for mode in (False, True):
    blk.zero_grad(set_to_none=True)
    xi = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = blk(xi)
    o.DEFERRED_FINISHES.enabled = mode
    y.backward(cot)
    o.DEFERRED_FINISHES.enabled = False
    torch.cuda.synchronize()
    print("mode", mode, "jobs left", len(o.DEFERRED_FINISHES.jobs))
    res[mode] = [("x", xi.grad.clone())] + [(n, p.grad.clone()) for n, p in blk.named_parameters()]
for (n, a), (_, b) in zip(res[False], res[True]):
    if not torch.equal(a, b):
        print("DIFF", n, tuple(a.shape), float((a - b).abs().max()), "nan" if torch.isnan(b).any() else "")
