"""Developer micro-benchmark: AttentionD8's core on packed rows (octic_attn_{fwd,bwd}_packed), forward / backward, with the
single-pass backward and with the dq + dkv pair.   usage: bench_attn_packed.py [B=256] [H=16] [T=37] [w=10]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octic_vits_amd import ops

B, H, T, w = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (256, 16, 37, 10)))
c = w * H
qkv = (torch.randn(B, T, 3 * 8 * c, device="cuda") * 0.7).bfloat16()
do = torch.randn(B, T, 8 * c, device="cuda").bfloat16()
scale = (8 * w) ** -0.5
o, lse = ops.attn_fwd_packed(qkv, H, c, scale)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tf = timeit(lambda: ops.attn_fwd_packed(qkv, H, c, scale))
tb = timeit(lambda: ops.attn_bwd_packed(qkv, o, do, lse, H, c, scale))
old = ops.ATTN_BWD_FUSED
ops.ATTN_BWD_FUSED = False
tp = timeit(lambda: ops.attn_bwd_packed(qkv, o, do, lse, H, c, scale))
ops.ATTN_BWD_FUSED = old
print(f"packed rows (B {B} H {H} T {T} hd {8 * w}): fwd {tf:.1f} us   bwd single-pass {tb:.1f} us   bwd dq + dkv pair {tp:.1f} us")
