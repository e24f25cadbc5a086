"""Developer micro-benchmark: hand-written TN weight-gradient kernel (csrc/dense_wgrad.hip) over row-slab counts vs the
library paths (single GEMM + cast; batched row-slab GEMM + f32 slab sum = what functional._wgrad_lib ships)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octic_vits_amd import ops, functional as OF, _lib

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

raw = ctypes.CDLL(_lib.LIB_PATH)
M = 16448
sweep = [int(s) for s in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,2,3,4,5,6,8,10".split(","))]
for (N, K) in [(5120, 1280), (1280, 5120), (3840, 1280), (1280, 1280)]:
    dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    xx = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    fl = 2.0 * M * N * K
    t_lib = timeit(lambda: (dy.t() @ xx).float())
    t_slab = timeit(lambda: OF._wgrad_lib(dy, xx))
    line = f"dW {N:5d}x{K:5d}: lib {t_lib:6.1f} us ({fl / t_lib / 1e6:5.0f} TF) | lib slabs {t_slab:6.1f} us ({fl / t_slab / 1e6:5.0f} TF) | mine"
    for S in sweep:
        raw.octic_route_override(2, S)
        ops._DW_WS.clear()
        t = timeit(lambda: ops.dense_wgrad_tn(dy, xx))
        line += f"  S={S or 'auto'}: {t:6.1f} ({fl / t / 1e6:5.0f})"
    raw.octic_route_override(2, 0)
    print(line, flush=True)
