"""Developer micro-benchmark: fused fc1 + D8-GELU (and dfc2 + GELU') vs the separate kernels, ViT-H, B = 64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octic_vits_amd import ops as o

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B, T, cin, cout = 64, 257, 160, 640
dt = torch.bfloat16
M = B * T
x = torch.randn(B, T, 8 * cin, device="cuda").to(dt)
W = [torch.randn(cout, cin, device="cuda") / 12 for _ in range(4)] + [torch.randn(2 * cout, 2 * cin, device="cuda") / 17]
bias = torch.randn(cout, device="cuda")
wb, wt = o.linear_prep(W, None, cin, cout, dt)
h = torch.empty(B, T, 8 * cout, dtype=dt, device="cuda"); y = torch.empty_like(h); g = torch.empty_like(h); dh = torch.empty_like(h)
t1 = timeit(lambda: o.linear_fwd(o.pview(x, cin), wb, bias, o.pview(h, cout), M, cin, cout, dt, dt, x))
t2 = timeit(lambda: o.gelu_fwd(o.pview(h, cout), o.pview(y, cout), M, cout, dt, h))
t3 = timeit(lambda: o.mlp_d8_gelu(x, wb[0], bias, cin, cout, 0))
print(f"forward : separate fc1 {t1:.1f} + gelu {t2:.1f} = {t1 + t2:.1f} us | fused {t3:.1f} us  ({2 * M * 8 * (cin + 2 * cout) / t3 / 1e3:.0f} GB/s algorithmic)")
W2 = [torch.randn(cin, cout, device="cuda") / 25 for _ in range(4)] + [torch.randn(2 * cin, 2 * cout, device="cuda") / 35]
_, w2t = o.linear_prep(W2, None, cout, cin, dt)
dy = torch.randn(B, T, 8 * cin, device="cuda").to(dt)
t4 = timeit(lambda: o.linear_fwd(o.pview(dy, cin), w2t, None, o.pview(g, cout), M, cin, cout, dt, dt, dy))
t5 = timeit(lambda: o.gelu_bwd(o.pview(g, cout), o.pview(h, cout), o.pview(dh, cout), M, cout, dt, h))
t6 = timeit(lambda: o.mlp_d8_gelu(dy, w2t[0], None, cin, cout, 1, h=h))
print(f"backward: separate dfc2 {t4:.1f} + gelu' {t5:.1f} = {t4 + t5:.1f} us | fused {t6:.1f} us")
