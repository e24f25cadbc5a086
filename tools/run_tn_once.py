import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octic_vits_amd import ops, _lib
S = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N, K = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (5120, 1280)
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.octic_dbg_dense_wgrad_slabs(S)
M = 16448
dy = torch.randn(M, N, device="cuda").to(torch.bfloat16)
xx = torch.randn(M, K, device="cuda").to(torch.bfloat16)
for _ in range(6):
    ops.dense_wgrad_tn(dy, xx)
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
for _ in range(6):
    ops.dense_gemm_nt(a, b, 0)
torch.cuda.synchronize()
