"""Developer A/B: TN weight-gradient kernel at the 256- and 320-wide tile, interleaved rounds in one process.
    python tools/ab_tn_tile.py [--rounds 3]"""
import argparse, ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octic_vits_amd import _lib, ops  # noqa: E402

L = _lib.lib()
raw = ctypes.CDLL(_lib.LIB_PATH)
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--slabs", type=int, nargs="*", default=[0])
a = ap.parse_args()
M = 64 * 257


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
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


for N, K in ((5120, 1280), (1280, 5120), (3840, 1280), (1280, 1280)):
    dy = (torch.randn(M, N, device="cuda") * 0.5).to(torch.bfloat16)
    x = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    for r in range(a.rounds):
        line = []
        for width in (256, 320):
            for S in a.slabs:
                raw.octic_route_override(3, width)
                raw.octic_route_override(2, S)
                t = timeit(lambda: ops.dense_wgrad_tn(dy, x))
                line.append(f"{width}{'/S' + str(S) if S else ''}: {t:6.1f} us ({2.0 * M * N * K / t / 1e6:5.0f} TF)")
        print(f"{N}x{K} round {r}: " + "   ".join(line), flush=True)
raw.octic_route_override(3, 0)
raw.octic_route_override(2, 0)
