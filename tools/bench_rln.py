import os, sys, torch
sys.path.insert(0, "/root/repo")
from octic_vits_amd import ops
M, d, T = 16448, 1280, 257
x = torch.randn(M, d, device="cuda"); yb = torch.randn(M, d, device="cuda").bfloat16()
g = torch.rand(d, device="cuda"); rs = torch.ones(64, device="cuda"); w = torch.rand(d, device="cuda"); b = torch.rand(d, device="cuda")
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n): fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("rpb", os.environ.get("OCTIC_RLN_RPB"), "fused %.1f us" % t(lambda: ops.dense_resid_layernorm_fwd(x, yb, g, rs, T, w, b, 1e-6, torch.bfloat16)),
      "separate %.1f + %.1f us" % (t(lambda: ops.scale_residual_fwd(x, yb, g, rs, T)), t(lambda: ops.dense_layernorm_fwd(x, w, b, 1e-6, torch.bfloat16))))
