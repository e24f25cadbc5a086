"""Developer A/B: 256-wide vs 320-wide tile of csrc/dense_gemm.hip (plain epilogue) on the ViT-H shapes, interleaved rounds
in ONE process on random data (cdna guide rule 24 / 25).  Variants: (tile, split) with tile 4 = 256-wide, 5 = 320-wide,
0 = the cost model's choice; split 0 = the plan's choice, 1 = remaining tiles unsplit in front of the grid, n = forced."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octic_vits_amd import _lib, ops

_lib.lib()
raw = ctypes.CDLL(_lib.LIB_PATH)


def setv(nt, split):
    raw.octic_route_override(0, nt)
    raw.octic_route_override(1, split)
    ops._DG_WS.clear()          # the workspace size depends on the split


def one(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


M = int(os.environ.get("M", 16448))
shapes = [(1280, 1280), (1280, 5120), (1280, 3840), (3840, 1280), (5120, 1280)]
for (N, K) in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    fn = lambda: ops.dense_gemm_nt(a, b, 0, bias=bias)
    variants = [(4, 0), (5, 0), (0, 0)] + ([(5, 1), (5, 2), (5, 3), (5, 8), (4, 1)] if N == 1280 else [])
    res = {v: [] for v in variants}
    for r in range(5):
        for v in variants:
            setv(*v)
            fn(); fn()
            torch.cuda.synchronize()
            res[v].append(one(fn))
    setv(0, 0)
    fl = 2.0 * M * N * K
    med = lambda x: sorted(x)[len(x) // 2]
    print(f"N={N:5d} K={K:5d}: " + " | ".join(f"t{v[0]}s{v[1]} {med(res[v]):6.1f} us {fl / med(res[v]) / 1e6:5.0f} TF" for v in variants), flush=True)
