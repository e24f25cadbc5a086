"""Developer micro-benchmark: attention backward (dq + dkv) on the standard half's strided rows and on packed octic rows,
HIP-event time over 20 back-to-back calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octic_vits_amd import functional as OF

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best

B, H, T, hd = 64, 16, 257, 80
q, k, v = (torch.randn(B, H, T, hd, device="cuda").bfloat16().requires_grad_(True) for _ in range(3))
do = torch.randn(B, H, T, hd, device="cuda").bfloat16()
o = OF.AttnFn.apply(q, k, v, hd ** -0.5)
t_plain = timeit(lambda: torch.autograd.grad(o, (q, k, v), do, retain_graph=True))
c = 10 * H
qkv = (torch.randn(B, T, 3 * 8 * c, device="cuda") * 0.7).bfloat16().requires_grad_(True)
dop = torch.randn(B, T, 8 * c, device="cuda").bfloat16()
op = OF.AttnPackedFn.apply(qkv, H, c, hd ** -0.5)
t_packed = timeit(lambda: torch.autograd.grad(op, qkv, dop, retain_graph=True))
print(f"attn bwd single pass: plain {t_plain:6.1f} us   packed {t_packed:6.1f} us")
from octic_vits_amd import ops
ops.ATTN_BWD_FUSED = False
t_plain2 = timeit(lambda: torch.autograd.grad(o, (q, k, v), do, retain_graph=True))
t_packed2 = timeit(lambda: torch.autograd.grad(op, qkv, dop, retain_graph=True))
print(f"attn bwd (dq + dkv): plain {t_plain2:6.1f} us   packed {t_packed2:6.1f} us")
