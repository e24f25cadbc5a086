"""Developer micro-benchmark: LinearD8 weight gradient (ring kernel + finish) at the ViT-H shapes, device time in a hipGraph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octic_vits_amd import ops

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        fn(); st.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best

B, T, c = 64, 257, 160
M = B * T
bf = torch.bfloat16
out = []
for name, cin, cout in (("qkv", c, 3 * c), ("proj", c, c), ("fc1", c, 4 * c), ("fc2", 4 * c, c)):
    x = torch.randn(M, 8 * cin, device="cuda").to(bf)
    dy = torch.randn(M, 8 * cout, device="cuda").to(bf)
    sp = ops.lib().octic_linear_d8_wgrad_splits(M, cin, cout)
    t = timeit(lambda: ops.linear_wgrad(ops.pview(x, cin), ops.pview(dy, cout), M, cin, cout, bf, x))
    out.append(f"{name} (S={sp}) {t:6.1f} us")
print("wgrad+finish: " + "   ".join(out))
