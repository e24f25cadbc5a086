"""Developer micro-benchmark: hand-written dense GEMM vs torch (hipBLASLt) on the ViT-H standard-block shapes."""
import sys, os
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

M = 16448
for (N, K) in [(3840, 1280), (1280, 1280), (5120, 1280), (1280, 5120), (1280, 3840)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    t_mine = timeit(lambda: ops.dense_gemm_nt(a, b, 0, bias=bias))
    t_lib = timeit(lambda: torch.nn.functional.linear(a, b, bias.to(torch.bfloat16)))
    fl = 2.0 * M * N * K
    print(f"N={N:5d} K={K:5d}: mine {t_mine:7.1f} us {fl / t_mine / 1e6:7.1f} TF | lib {t_lib:7.1f} us {fl / t_lib / 1e6:7.1f} TF", flush=True)
x = torch.randn(M, 1280, device="cuda")
a = torch.randn(M, 5120, device="cuda").to(torch.bfloat16); b = (torch.randn(1280, 5120, device="cuda") / 70).to(torch.bfloat16)
g = torch.rand(1280, device="cuda"); rs = torch.ones(64, device="cuda")
print("fc2 resid mode", timeit(lambda: ops.dense_gemm_nt(a, b, 2, bias=g, gamma=g, rs=rs, rps=257, x=x)))
a = torch.randn(M, 1280, device="cuda").to(torch.bfloat16); b = (torch.randn(5120, 1280, device="cuda") / 36).to(torch.bfloat16)
g5 = torch.rand(5120, device="cuda")
print("fc1 gelu mode", timeit(lambda: ops.dense_gemm_nt(a, b, 1, bias=g5)))
