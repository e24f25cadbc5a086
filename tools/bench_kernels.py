"""Micro-benchmark of the HIP kernels at the BASELINE shapes (ViT-H/14, B=64, T=257, c=160), HIP-event timed.
Prints avg microseconds and algorithmic GB/s per launch.  Usage: python tools/bench_kernels.py [filter]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octic_vits_amd import ops  # noqa: E402

dev = "cuda"
B, T, c, H = 64, int(os.environ.get("OCTIC_BENCH_T", "257")), 160, 16
M = B * T
flt = sys.argv[1] if len(sys.argv) > 1 else ""
bf, f32 = torch.bfloat16, torch.float32


def timeit(name, fn, nbytes, flops=0, reps=20):
    if flt and flt not in name:
        return
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{name:34s} {us:9.1f} us  {nbytes / us / 1e3:8.1f} GB/s  {flops / us / 1e6:8.1f} TFLOP/s", flush=True)


def rnd(*shape, dtype=bf):
    return torch.randn(*shape, device=dev, dtype=torch.float32).to(dtype)


def weights(cin, cout, dtype=bf):
    return [rnd(cout, cin, dtype=dtype) * 0.05 for _ in range(4)] + [rnd(2 * cout, 2 * cin, dtype=dtype) * 0.05]


for name, cin, cout, fused in [("qkv", c, 3 * c, False), ("proj+res", c, c, True), ("fc1", c, 4 * c, False),
                               ("fc2+res", 4 * c, c, True), ("dgrad qkv", 3 * c, c, False), ("dgrad fc1", 4 * c, c, False),
                               ("dgrad fc2", c, 4 * c, False)]:
    x = rnd(B, T, 8 * cin)
    w = weights(cin, cout)
    od = f32 if fused else bf
    y = torch.empty(B, T, 8 * cout, device=dev, dtype=od)
    bias = rnd(cout, dtype=f32)
    kw = {}
    if fused:
        r = rnd(B, T, 8 * cout, dtype=f32)
        rs = torch.ones(B, device=dev)
        cs = [torch.ones(cout, device=dev) for _ in range(4)] + [torch.ones(2 * cout, device=dev)]
        kw = dict(resid_v=ops.pview(r, cout), rs=rs, rps=T, cs5=cs)
    xv, yv = ops.pview(x, cin), ops.pview(y, cout)
    nb = M * 8 * cin * 2 + M * 8 * cout * y.element_size() * (2 if fused else 1) + 8 * cin * cout * 2
    timeit(f"linear {name} [{cin}->{cout}]", lambda: ops.linear_fwd(xv, w, bias, yv, M, cin, cout, bf, od, x, **kw), nb,
           24.0 * M * cin * cout)
    dy = rnd(B, T, 8 * cout)
    dv = ops.pview(dy, cout)
    L = ops.lib()
    import ctypes
    splits = L.octic_linear_d8_wgrad_splits(M, cin, cout)
    ws = torch.empty(L.octic_linear_d8_wgrad_workspace_bytes(cin, cout, splits) // 4, device=dev)
    nbw = M * 8 * (cin + cout) * 2 + splits * 8 * cin * cout * 4
    if not name.startswith("dgrad"):
        timeit(f"wgrad {name} (splits {splits})", lambda: ops.check(L.octic_linear_d8_wgrad(
            ctypes.byref(xv), ctypes.byref(dv), M, cin, cout, 1, ops._p(ws), splits, ops._stream(x))), nbw,
            24.0 * M * cin * cout)

x32 = rnd(B, T, 8 * c, dtype=f32)
alpha = [torch.ones(c, device=dev) for _ in range(4)] + [torch.ones(2 * c, device=dev)]
beta = torch.zeros(c, device=dev)
timeit("ln_fwd f32->bf16", lambda: ops.layernorm_fwd(x32, alpha, beta, 1e-5, bf, c), M * 8 * c * 6)
y, stats = ops.layernorm_fwd(x32, alpha, beta, 1e-5, bf, c)
g = rnd(B, T, 8 * c)
timeit("ln_bwd (+finish)", lambda: ops.layernorm_bwd(g, x32, stats, alpha, None, c), M * 8 * c * 10)
dres = rnd(B, T, 8 * c, dtype=f32)
timeit("ln_bwd +dres (+finish)", lambda: ops.layernorm_bwd(g, x32, stats, alpha, dres, c), M * 8 * c * 14)
wd, bd = rnd(8 * c, dtype=f32), rnd(8 * c, dtype=f32)
yd, std = ops.dense_layernorm_fwd(x32, wd, bd, 1e-6, bf)
timeit("dense_ln_fwd f32->bf16", lambda: ops.dense_layernorm_fwd(x32, wd, bd, 1e-6, bf), M * 8 * c * 6)
timeit("dense_ln_bwd +dres (+finish)", lambda: ops.dense_layernorm_bwd(g, x32, wd, std, dres), M * 8 * c * 14)
timeit("scale_residual_fwd", lambda: ops.scale_residual_fwd(x32, g, wd, None, T), M * 8 * c * 10)
timeit("scale_residual_bwd (+finish)", lambda: ops.scale_residual_bwd(x32, g, wd, None, T), M * 8 * c * 8)
h = rnd(B, T, 32 * c)
hy = torch.empty_like(h)
timeit("gelu_fwd", lambda: ops.gelu_fwd(ops.pview(h, 4 * c), ops.pview(hy, 4 * c), M, 4 * c, bf, h), 2 * h.numel() * 2)
timeit("gelu_bwd", lambda: ops.gelu_bwd(ops.pview(h, 4 * c), ops.pview(h, 4 * c), ops.pview(hy, 4 * c), M, 4 * c, bf, h),
       3 * h.numel() * 2)
qkv = rnd(B, T, 24 * c)
timeit("pack_heads", lambda: ops.pack_heads(qkv, B, T, H, c, 3), 2 * qkv.numel() * 2)
hs = ops.pack_heads(qkv, B, T, H, c, 3)
timeit("unpack_heads n_s=3", lambda: ops.unpack_heads(hs, B, T, H, c), 2 * qkv.numel() * 2)
timeit("unpack_heads n_s=1", lambda: ops.unpack_heads(hs[:1], B, T, H, c), 2 * hs[0].numel() * 2)
timeit("cast_rowscale f32->bf16", lambda: ops.cast_rowscale(x32, None, 1, bf, c), M * 8 * c * 6)
qh, kh, vh = (rnd(B, H, T, 80) for _ in range(3))
timeit("attn_fwd hip", lambda: ops.attn_fwd(qh, kh, vh, 80 ** -0.5), 4 * qh.numel() * 2, 4.0 * B * H * T * T * 80)
import torch.nn.functional as F
timeit("attn_fwd torch sdpa", lambda: F.scaled_dot_product_attention(qh, kh, vh), 4 * qh.numel() * 2, 4.0 * B * H * T * T * 80)
from octic_vits_amd.functional import AttnFn
qg, kg, vg = (rnd(B, H, T, 80).requires_grad_(True) for _ in range(3))
og = AttnFn.apply(qg, kg, vg, 80 ** -0.5)
dog = rnd(B, H, T, 80)
timeit("attn_bwd hip", lambda: torch.autograd.grad(og, (qg, kg, vg), dog, retain_graph=True), 8 * qg.numel() * 2, 14.0 * B * H * T * T * 80)
ot = F.scaled_dot_product_attention(qg, kg, vg)
timeit("attn_bwd torch sdpa", lambda: torch.autograd.grad(ot, (qg, kg, vg), dog, retain_graph=True), 8 * qg.numel() * 2, 14.0 * B * H * T * T * 80)
