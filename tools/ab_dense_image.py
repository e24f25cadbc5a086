"""Developer A/B (round 6): per-image row panels + class-token kernel against the classic panels of csrc/dense_gemm.hip on the
ViT-H shapes (B = 64 images x 257 tokens), all modes the step uses, interleaved rounds in ONE process on random data."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octic_vits_amd import _lib, ops

_lib.lib()


def one(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B = int(os.environ.get("B", 64))
M = B * 257
cases = [(1280, 1280, 0), (1280, 5120, 0), (1280, 3840, 0), (3840, 1280, 0), (5120, 1280, 4), (5120, 1280, 5), (5120, 1280, 6)]
if os.environ.get("CLS2"):        # class-token rows as two launches (K split over workgroups): never / wherever legal
    _lib.route_override(_lib.ROUTE_DENSE_CLS2, int(os.environ["CLS2"]))
med = lambda x: sorted(x)[len(x) // 2]
for (N, K, mode) in cases:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    h = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    if mode == 5:
        fn = lambda: ops.dense_gemm_nt(a, b, 5, h=h, want_colsum=True, tokens=257)
    else:
        fn = lambda: ops.dense_gemm_nt(a, b, mode, bias=bias, tokens=257)
    res = {1: [], 2: [], 0: []}
    for r in range(5):
        for v in res:
            _lib.route_override(_lib.ROUTE_DENSE_IMAGE, v)
            fn(); fn()
            torch.cuda.synchronize()
            res[v].append(one(fn))
    _lib.route_override(_lib.ROUTE_DENSE_IMAGE, 0)
    fl = 2.0 * M * N * K
    print(f"N={N:5d} K={K:5d} mode {mode}: per-image {med(res[1]):6.1f} us {fl / med(res[1]) / 1e6:5.0f} TF | classic {med(res[2]):6.1f} us "
          f"{fl / med(res[2]) / 1e6:5.0f} TF | model's choice {med(res[0]):6.1f} us (image={ops.dense_plan(M, N, K, mode, 257)[2]})", flush=True)
