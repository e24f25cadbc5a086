"""Developer check + A/B: the W-stationary LinearD8 kernel (csrc/gemm_wreg.hip) against a torch f32 restatement and
against the ring kernel (octic_route_override), over shapes with ragged row / column tails and every epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octic_vits_amd import _lib, ops

L = _lib.lib()
bf, f32 = torch.bfloat16, torch.float32


def ref_linear(x, w, bias, cin, cout, resid=None, cs=None, rs=None, rps=1):
    M = x.shape[0]
    xs = x.float()
    outs = []
    for g in range(4):
        y = xs[:, g * cin:(g + 1) * cin] @ w[g].float().t()
        if g == 0 and bias is not None:
            y = y + bias
        if cs is not None:
            y = y * cs[g]
        outs.append(y)
    e = xs[:, 4 * cin:].reshape(M, 2, 2 * cin) @ w[4].float().t()
    if cs is not None:
        e = e * cs[4]
    outs.append(e.reshape(M, 4 * cout))
    y = torch.cat(outs, dim=1)
    if rs is not None:
        y = y * rs[torch.arange(M, device=x.device) // rps][:, None]
    if resid is not None:
        y = y + resid.float()
    return y


def timeit(fn, n=20):
    """Device time per call: n calls captured in one hipGraph (no host launch gaps), replayed three times."""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        fn()
        st.synchronize()
        with torch.cuda.graph(g, stream=st):
            for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


def run(M, cin, cout, out_dt, fused, bias_on=True, time_it=False):
    torch.manual_seed(M + cin + cout)
    x = torch.randn(M, 8 * cin, device="cuda").to(bf)
    w = [(torch.randn(cout, cin, device="cuda") * cin ** -0.5).to(bf) for _ in range(4)] + [(torch.randn(2 * cout, 2 * cin, device="cuda") * (2 * cin) ** -0.5).to(bf)]
    bias = torch.randn(cout, device="cuda") if bias_on else None
    resid = cs = rs = None
    rps = 1
    if fused:
        resid = torch.randn(M, 8 * cout, device="cuda").to(out_dt)
        cs = [torch.rand(cout, device="cuda") + 0.5 for _ in range(4)] + [torch.rand(2 * cout, device="cuda") + 0.5]
        rps = 37 if M % 37 == 0 else (257 if M % 257 == 0 else M)
        rs = torch.rand((M + rps - 1) // rps, device="cuda") + 0.5
    want = ref_linear(x, w, bias, cin, cout, resid, cs, rs, rps)
    res = {}
    for off in (0, 1):
        L.octic_route_override(4, off)
        y = torch.full((M, 8 * cout), float("nan"), device="cuda", dtype=out_dt)
        call = lambda: ops.linear_fwd(ops.pview(x, cin), w, bias, ops.pview(y, cout), M, cin, cout, bf, out_dt, x,
                                      resid_v=ops.pview(resid, cout) if fused else None, rs=rs, rps=rps, cs5=cs)
        call()
        torch.cuda.synchronize()
        err = (y.float() - want).abs().max().item()
        t = timeit(call) if time_it else 0.0
        res[off] = (err, t, y.clone())
    L.octic_route_override(4, 0)
    scale = want.abs().max().item()
    same = (res[0][2].float() - res[1][2].float()).abs().max().item()
    tag = f"M={M:6d} cin={cin:4d} cout={cout:4d} out={'bf16' if out_dt == bf else 'f32 '} fused={int(fused)}"
    print(f"{tag}: wreg err {res[0][0]:.3e}  ring err {res[1][0]:.3e}  (scale {scale:.2f}, wreg-ring {same:.3e})"
          + (f"   wreg {res[0][1]:6.1f} us  ring {res[1][1]:6.1f} us" if time_it else ""), flush=True)
    tol = (2e-2 if out_dt == bf else 2e-3) * max(scale, 1.0)
    return res[0][0] <= tol and not torch.isnan(res[0][2]).any().item()


ok = True
if "--bench" not in sys.argv and "--long" not in sys.argv:
    for (M, cin, cout) in [(96, 32, 32), (77, 64, 24), (500, 96, 40), (640, 128, 128), (37 * 9, 160, 160), (1001, 160, 480), (257 * 4, 160, 640), (33, 32, 8)]:
        for out_dt, fused in ((bf, False), (f32, True), (bf, True), (f32, False)):
            ok = run(M, cin, cout, out_dt, fused) and ok
M = 64 * 257
SHORT = (("qkv", 160, 480, bf, False), ("fc1", 160, 640, bf, False), ("proj+res", 160, 160, f32, True),
         ("dgrad fc2", 160, 640, bf, False), ("proj bf16 res", 160, 160, bf, True))
LONG = (("fc2+res", 640, 160, f32, True), ("dgrad fc1", 640, 160, bf, False), ("dgrad qkv", 480, 160, bf, False))
if "--no-plan" in sys.argv:
    L.octic_route_override(5, 1)
for name, cin, cout, out_dt, fused in (LONG if "--long" in sys.argv else SHORT):
    ok = run(M, cin, cout, out_dt, fused, bias_on=name != "dgrad fc2", time_it=True) and ok
print("ALL OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
