"""Developer probe: single-pass attention backward vs float64 on one head; prints where the errors are."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from octic_vits_amd import ops
B, H, T, hd = 1, 1, 257, 80
g = torch.Generator().manual_seed(1)
q, k, v = (torch.randn(B, H, T, hd, generator=g).to(torch.bfloat16).cuda() for _ in range(3))
do = torch.randn(B, H, T, hd, generator=g).to(torch.bfloat16).cuda()
scale = hd ** -0.5
o, lse = ops.attn_fwd(q, k, v, scale)
qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
ro = torch.softmax((qd @ kd.transpose(-1, -2)) * scale, dim=-1) @ vd
ro.backward(do.double())
for fused in (True, False):
    ops.ATTN_BWD_FUSED = fused
    dq, dk, dv = (torch.full_like(q, float("nan")) for _ in range(3))
    ops.attn_bwd(q, k, v, o, do, lse, scale, dq, dk, dv)
    torch.cuda.synchronize()
    print("fused" if fused else "pair")
    for name, got, want in (("dq", dq, qd.grad), ("dk", dk, kd.grad), ("dv", dv, vd.grad)):
        e = (got.double() - want).abs()[0, 0]          # [T, hd]
        print(f"  {name}: max err {float(e.max()):.3e}  nan {int(torch.isnan(got).sum())}  ref max {float(want.abs().max()):.3g}")
        if float(e.max()) > 3e-2 or torch.isnan(got).any():
            rows = e.max(dim=1).values
            bad = (rows > 3e-2) | torch.isnan(rows)
            print("    bad rows:", bad.nonzero().flatten().tolist()[:40], "... count", int(bad.sum()))
            cols = e.max(dim=0).values
            badc = (cols > 3e-2) | torch.isnan(cols)
            print("    bad cols:", badc.nonzero().flatten().tolist())
            r0 = int(bad.nonzero()[0])
            print("    row", r0, "got", got[0, 0, r0, :8].float().tolist(), "want", want[0, 0, r0, :8].float().tolist())
            ratio = (got.double()[0, 0] / want[0, 0])
            print("    median ratio", float(ratio[~torch.isnan(ratio)].median()))

# delta written by the kernels vs <dO, O>
from octic_vits_amd.ops import lib, check, _p, _stream
for phase in (3, 1):
    delta = torch.full((B, H, T), float("nan"), dtype=torch.float32, device="cuda")
    dq, dk, dv = (torch.empty_like(q) for _ in range(3))
    st, so, sg = q.stride(), o.stride(), dq.stride()
    check(lib().octic_attn_bwd(_p(q), _p(k), _p(v), _p(o), _p(do), _p(lse), _p(delta), _p(dq), _p(dk), _p(dv), B, H, T, hd,
                               st[0], st[1], st[2], so[0], so[1], so[2], sg[0], sg[1], sg[2], float(scale), phase, _stream(q)))
    torch.cuda.synchronize()
    want = (do.double() * o.double()).sum(-1)
    e = (delta.double() - want).abs()[0, 0]
    print("phase", phase, "delta max err", float(e.max()), "bad", (e > 1e-3).nonzero().flatten().tolist()[:40])
    print("   got", delta[0, 0, :6].tolist(), "want", want[0, 0, :6].tolist())
