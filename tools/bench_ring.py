"""Developer micro-benchmark: the two-operand ring LinearD8 GEMM at the ViT-H long-K shapes (fc2 forward, input gradients
of qkv and fc1), B 64, T 257, c 160, bf16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octic_vits_amd import ops

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B, T, c = 64, 257, 160
M = B * T
bf = torch.bfloat16
out = []
for name, cin, cout in (("fc2", 4 * c, c), ("dqkv", 3 * c, c)):
    x = torch.randn(B, T, 8 * cin, device="cuda").to(bf)
    w = [(torch.randn(cout, cin, device="cuda") * 0.05).to(bf) for _ in range(4)] + [(torch.randn(2 * cout, 2 * cin, device="cuda") * 0.05).to(bf)]
    y = torch.empty(B, T, 8 * cout, device="cuda", dtype=bf)
    t = timeit(lambda: ops.linear_fwd(ops.pview(x, cin), w, None, ops.pview(y, cout), M, cin, cout, bf, bf, x))
    nb = M * 8 * (cin + cout) * 2 + 24 * cin * cout
    out.append(f"{name} {t:6.1f} us ({nb / t / 1e3:5.0f} GB/s)")
print("   ".join(out))
