"""Developer micro-benchmark: attention core forward / backward at the ViT-H shape (B 64, 16 heads, 257 tokens, hd 80)."""
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

B, H, T, hd = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (64, 16, 257, 80)))
from octic_vits_amd import _lib
for layout in ("contiguous", "fused-qkv"):
    if layout == "contiguous":
        q, k, v = (torch.randn(B, H, T, hd, device="cuda").bfloat16() for _ in range(3))
        dq, dk, dv = (torch.empty_like(q) for _ in range(3))
    else:
        qkv = torch.randn(B, T, 3, H, hd, device="cuda").bfloat16()
        q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        g = torch.empty_like(qkv)
        dq, dk, dv = (g[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    do = torch.randn(B, H, T, hd, device="cuda").bfloat16()
    o, lse = ops.attn_fwd(q, k, v, hd ** -0.5)
    tf = timeit(lambda: ops.attn_fwd(q, k, v, hd ** -0.5))
    tb = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, hd ** -0.5, dq, dk, dv))
    fl = 4.0 * B * H * T * T * hd
    _lib.route_override(_lib.ROUTE_ATTN_ONLINE, 1)
    tfo = timeit(lambda: ops.attn_fwd(q, k, v, hd ** -0.5))
    _lib.route_override(_lib.ROUTE_ATTN_ONLINE, 0)
    print(f"(B {B} H {H} T {T} hd {hd}) forward with the online-softmax kernel forced: {tfo:6.1f} us")
    print(f"{layout:11s} fwd {tf:6.1f} us ({fl / tf / 1e6:5.0f} TF)   bwd dq+dkv {tb:6.1f} us ({3.5 * fl / tb / 1e6:5.0f} TF)")
