"""Which items does the planned ring launch miss?  (NaN-filled output, fc2 shape)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from octic_vits_amd import _lib, ops
L = _lib.lib()
bf = torch.bfloat16
M, cin, cout = 64 * 257, 640, 160
x = torch.randn(M, 8 * cin, device="cuda").to(bf)
w = [(torch.randn(cout, cin, device="cuda") * cin ** -0.5).to(bf) for _ in range(4)] + [(torch.randn(2 * cout, 2 * cin, device="cuda") * (2 * cin) ** -0.5).to(bf)]
outs = []
for off in (1, 0):
    L.octic_route_override(5, off)
    y = torch.full((M, 8 * cout), float("nan"), device="cuda", dtype=bf)
    ops.linear_fwd(ops.pview(x, cin), w, None, ops.pview(y, cout), M, cin, cout, bf, bf, x)
    torch.cuda.synchronize()
    outs.append(y)
ref, got = outs
print("ref nan", torch.isnan(ref).sum().item(), "got nan", torch.isnan(got).sum().item())
bad = torch.isnan(got)
seg = [("A1", 0, 160), ("A2", 160, 320), ("B1", 320, 480), ("B2", 480, 640), ("E0", 640, 960), ("E1", 960, 1280)]
for name, c0, c1 in seg:
    b = bad[:, c0:c1]
    rows = b.any(1).nonzero().flatten()
    print(name, "bad rows", rows.numel(), "tiles", sorted(set((rows // 128).tolist()))[:20], "cols bad", b.any(0).sum().item())
ok = ~bad
print("max diff where written", (ref.float() - got.float())[ok].abs().max().item())
