"""One-off GPU probe: what the stock PyTorch-ROCm ops of the standard half deliver on this box."""
import time
import torch
import torch.nn.functional as F

dev = "cuda"
print(torch.__version__, torch.cuda.get_device_name(0))

def bench(fn, n=20, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n

M = 64 * 257
for (K, N) in [(1280, 3840), (1280, 1280), (1280, 5120), (5120, 1280)]:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    t = bench(lambda: F.linear(x, w))
    print(f"linear bf16 M={M} K={K} N={N}: {t*1e6:8.1f} us  {2*M*K*N/t/1e12:7.1f} TFLOP/s")
for hd, H in [(80, 16), (64, 16), (128, 10)]:
    q, k, v = (torch.randn(64, H, 257, hd, device=dev, dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
    for name in ("flash", "efficient", "math"):
        try:
            from torch.nn.attention import sdpa_kernel, SDPBackend
            be = {"flash": SDPBackend.FLASH_ATTENTION, "efficient": SDPBackend.EFFICIENT_ATTENTION, "math": SDPBackend.MATH}[name]
            with sdpa_kernel(be):
                t = bench(lambda: F.scaled_dot_product_attention(q, k, v))
                o = F.scaled_dot_product_attention(q, k, v)
                go = torch.randn_like(o)
                tb = bench(lambda: torch.autograd.grad(o, (q, k, v), go, retain_graph=True))
            print(f"sdpa {name:9s} hd={hd} H={H}: fwd {t*1e6:8.1f} us  bwd {tb*1e6:8.1f} us")
        except Exception as e:
            print(f"sdpa {name} hd={hd}: FAILED {type(e).__name__}: {str(e)[:100]}")
x = torch.randn(M, 1280, device=dev)
t = bench(lambda: F.layer_norm(x, (1280,)))
print(f"layer_norm f32 [{M},1280]: {t*1e6:.1f} us -> {2*x.numel()*4/t/1e12:.2f} TB/s")
xb = torch.randn(M, 5120, device=dev, dtype=torch.bfloat16)
t = bench(lambda: F.gelu(xb))
print(f"gelu bf16 [{M},5120]: {t*1e6:.1f} us -> {2*xb.numel()*2/t/1e12:.2f} TB/s")
a = torch.empty(256 * 1024 * 1024, device=dev, dtype=torch.float32)
b = torch.empty_like(a)
t = bench(lambda: b.copy_(a))
print(f"copy 1 GiB: {t*1e6:.1f} us -> {2*a.numel()*4/t/1e12:.2f} TB/s")
