import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octic_vits_amd import ops
M, N, K = 16448, int(sys.argv[1]) if len(sys.argv) > 1 else 5120, int(sys.argv[2]) if len(sys.argv) > 2 else 1280
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
b = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
for _ in range(5):
    ops.dense_gemm_nt(a, b, 0)
    torch.nn.functional.linear(a, b)
torch.cuda.synchronize()
