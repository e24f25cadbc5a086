import time, torch, torch.nn.functional as F
from torch.nn.attention import sdpa_kernel, SDPBackend
dev = "cuda"
def bench(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for hd in (80, 96, 128):
    q, k, v = (torch.randn(64, 16, 257, hd, device=dev, dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
    for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION)):
        try:
            with sdpa_kernel(be):
                scale = 80 ** -0.5
                tf = bench(lambda: F.scaled_dot_product_attention(q, k, v, scale=scale))
                o = F.scaled_dot_product_attention(q, k, v, scale=scale)
                go = torch.randn_like(o)
                tb = bench(lambda: torch.autograd.grad(o, (q, k, v), go, retain_graph=True))
            print(f"hd={hd:3d} {name:9s} fwd {tf:7.1f} us  bwd {tb:7.1f} us")
        except Exception as e:
            print(f"hd={hd} {name} FAILED {str(e)[:80]}")
