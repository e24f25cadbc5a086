"""Developer probe: torch.compile over the model - where does Dynamo break the graph, and does a compiled block equal eager?
usage: compile_probe.py [small|huge]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch._dynamo as dynamo
from octic_vits_amd.model import OcticVisionTransformer
from octic_vits_amd.d8_layers import Layer_scale_init_BlockD8
from octic_vits_amd.vit import Layer_scale_init_Block
from octic_vits_amd.functional import Octic

torch.manual_seed(0)
kw = dict(img_size=56, patch_size=14, num_classes=10, embed_dim=640, depth=4, num_heads=8, qkv_bias=True, init_scale=0.1,
          octic_block_layers=Layer_scale_init_BlockD8, standard_block_layers=Layer_scale_init_Block, drop_path_rate=0.0)
net = OcticVisionTransformer(**kw).cuda().train()
x = torch.randn(4, 3, 56, 56, device="cuda")

def run(m):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return m(x)

ex = dynamo.explain(run)(net)
print("graphs", ex.graph_count, "breaks", ex.graph_break_count)
for r in ex.break_reasons:
    print("  BREAK:", str(r.reason)[:300].replace("\n", " | "))
    for fs in r.user_stack[-2:]:
        print("      at", fs.filename.split("/")[-1], fs.lineno, fs.name)

# compiled model vs eager: bf16 train step (forward + backward), every parameter gradient
def grads(m, seed):
    torch.manual_seed(seed); torch.cuda.manual_seed(seed)
    for p in net.parameters():
        p.grad = None
    out = run(m)
    out.float().square().mean().backward()
    return out.detach().float(), {n: p.grad.detach().float().clone() for n, p in net.named_parameters() if p.grad is not None}

o1, g1 = grads(net, 1)
cm = torch.compile(net, backend="aot_eager")
t0 = time.time(); o2, g2 = grads(cm, 1); print(f"first compiled step {time.time() - t0:.1f} s")
print("forward max diff", float((o1 - o2).abs().max()), "scale", float(o1.abs().max()))
worst = max(((g1[n] - g2[n]).norm() / g1[n].norm().clamp_min(1e-12)).item() for n in g1)
print("worst relative gradient difference", worst, "tensors", len(g1), len(g2))
