"""Developer probe: where does the TN weight-gradient kernel's time go?  Same launch with (a) real operands, (b) row
stride 0 = zero-record descriptors: every DMA is dropped by the range check (no memory traffic, pipeline only),
(c) a tiny row stride: all rows inside 0.5 MB (L2-resident)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octic_vits_amd import ops, _lib
from octic_vits_amd.ops import _p, _stream, check, lib

def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

raw = ctypes.CDLL(_lib.LIB_PATH)
M = 16448
N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5120, 1280)
Ss = [int(s) for s in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["2", "5"])]
dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
xx = torch.randn(M, K, device="cuda").to(torch.bfloat16)
dw = torch.empty(N, K, device="cuda")
for S in Ss:
    raw.octic_dbg_dense_wgrad_slabs(S)
    ws = torch.zeros(int(lib().octic_dense_wgrad_workspace_bytes(M, N, K)), dtype=torch.uint8, device="cuda")
    def run(ldy, ldx):
        check(lib().octic_dense_wgrad_tn(_p(dy), _p(xx), M, N, K, ldy, ldx, _p(dw), _p(ws), _stream(dy)))
    fl = 2.0 * M * N * K
    for name, ldy, ldx in (("real", N, K), ("no-mem (ld 0)", 0, 0), ("L2-resident (ld 8)", 8, 8), ("Y real, X none", N, 0), ("Y none, X real", 0, K)):
        t = timeit(lambda: run(ldy, ldx))
        print(f"dW {N}x{K} S={S} {name:20s}: {t:7.1f} us ({fl / t / 1e6:6.0f} TF-equivalent)", flush=True)
