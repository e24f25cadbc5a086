"""Launch one LinearD8 GEMM shape a few times (for rocprofv3 --pmc passes).  usage: one_kernel.py [fc1|qkv|fc2|dgrad_fc1]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octic_vits_amd import ops
B, T, c = 64, 257, 160
M = B * T
which = sys.argv[1] if len(sys.argv) > 1 else "fc1"
cin, cout = {"fc1": (c, 4 * c), "qkv": (c, 3 * c), "fc2": (4 * c, c), "dgrad_fc1": (4 * c, c)}[which]
bf = torch.bfloat16
x = torch.randn(B, T, 8 * cin, device="cuda").to(bf)
w = [(torch.randn(cout, cin, device="cuda") * 0.05).to(bf) for _ in range(4)] + [(torch.randn(2 * cout, 2 * cin, device="cuda") * 0.05).to(bf)]
y = torch.empty(B, T, 8 * cout, device="cuda", dtype=bf)
for _ in range(5):
    ops.linear_fwd(ops.pview(x, cin), w, None, ops.pview(y, cout), M, cin, cout, bf, bf, x)
torch.cuda.synchronize()
