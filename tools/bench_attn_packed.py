"""Developer micro-benchmark: AttentionD8 core on packed rows vs pack -> attention -> unpack (ViT-H: B 64, T 257, 16 heads)."""
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

B, T, H, c = 64, 257, 16, 160
qkv = (torch.randn(B, T, 3 * 8 * c, device="cuda") * 0.7).bfloat16()
do = torch.randn(B, T, 8 * c, device="cuda").bfloat16()
o, lse = ops.attn_fwd_packed(qkv, H, c, 80 ** -0.5)
tf = timeit(lambda: ops.attn_fwd_packed(qkv, H, c, 80 ** -0.5))
tb = timeit(lambda: ops.attn_bwd_packed(qkv, o, do, lse, H, c, 80 ** -0.5))
q, k, v = ops.pack_heads(qkv, B, T, H, c, 3)
tp = timeit(lambda: ops.pack_heads(qkv, B, T, H, c, 3))
o2, l2 = ops.attn_fwd(q, k, v, 80 ** -0.5)
tf2 = timeit(lambda: ops.attn_fwd(q, k, v, 80 ** -0.5))
tu = timeit(lambda: ops.unpack_heads([o2], B, T, H, c))
dq, dk, dv = (torch.empty_like(q) for _ in range(3))
dob = ops.pack_heads(do, B, T, H, c, 1)[0]
tb2 = timeit(lambda: ops.attn_bwd(q, k, v, o2, dob, l2, 80 ** -0.5, dq, dk, dv))
tp1 = timeit(lambda: ops.pack_heads(do, B, T, H, c, 1))
tu3 = timeit(lambda: ops.unpack_heads([dq, dk, dv], B, T, H, c))
print(f"packed  : fwd {tf:.1f} us   bwd {tb:.1f} us   total {tf + tb:.1f}")
print(f"separate: fwd pack {tp:.1f} + {tf2:.1f} + unpack {tu:.1f} = {tp + tf2 + tu:.1f} | bwd pack {tp1:.1f} + {tb2:.1f} + unpack {tu3:.1f} = "
      f"{tp1 + tb2 + tu3:.1f} | total {tp + tf2 + tu + tp1 + tb2 + tu3:.1f}")
