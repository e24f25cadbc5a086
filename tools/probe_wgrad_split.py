"""Developer probe: library weight gradient dW = dY^T X as a batched GEMM over S row slabs (+ a sum over the slabs) against
the single GEMM (few output tiles -> partly idle chip).  ViT-H standard half, M = 16448."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octic_vits_amd.train import use_tuned_gemms

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

if "--tuned" in sys.argv:
    use_tuned_gemms()
M = 16448
for (N, K) in [(5120, 1280), (1280, 5120), (3840, 1280), (1280, 1280)]:
    dy = torch.randn(M, N, device="cuda").bfloat16()
    x = torch.randn(M, K, device="cuda").bfloat16()
    ref = (dy.t() @ x).float()
    t0 = timeit(lambda: (dy.t() @ x).float())
    line = f"dW {N}x{K}: mm+cast {t0:6.1f} us |"
    for S in (2, 4, 8, 16):
        dys, xs = dy.view(S, M // S, N), x.view(S, M // S, K)
        f = lambda: torch.bmm(dys.transpose(1, 2), xs).sum(0, dtype=torch.float32)
        out = f()
        err = ((out - ref).norm() / ref.norm()).item()
        f32 = lambda: torch.bmm(dys.transpose(1, 2), xs, out_dtype=torch.float32).sum(0)
        try:
            t32 = timeit(f32)
        except Exception:
            t32 = float("nan")
        line += f"  S={S}: {timeit(f):6.1f} (f32 partials {t32:6.1f}) err {err:.1e}"
    print(line, flush=True)
