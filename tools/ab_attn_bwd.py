"""Developer A/B of csrc/attn80_bwd.hip builds: side libraries with extra -D flags, interleaved timing in one process is not
possible across libraries, so each variant runs in a child process on the same device, three rounds, median of the per-round
minima.  usage: python tools/ab_attn_bwd.py "-DBW_HI_PRIO=0" "-DBW_HI_PRIO=1" """
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    from octic_vits_amd import _lib
    _lib.LIB_PATH = sys.argv[2]
    from octic_vits_amd import functional as OF
    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / n * 1e3)
        return best
    B, H, T, hd = 64, 16, 257, 80
    qkv = torch.randn(B, T, 3, H, hd, device="cuda").bfloat16().requires_grad_(True)
    do = torch.randn(B, T, H * hd, device="cuda").bfloat16()
    o = OF.AttnFusedQKVFn.apply(qkv, hd ** -0.5)
    t_plain = timeit(lambda: torch.autograd.grad(o, qkv, do, retain_graph=True))
    c = 10 * H
    qp = (torch.randn(B, T, 3 * 8 * c, device="cuda") * 0.7).bfloat16().requires_grad_(True)
    dop = torch.randn(B, T, 8 * c, device="cuda").bfloat16()
    op = OF.AttnPackedFn.apply(qp, H, c, hd ** -0.5)
    t_packed = timeit(lambda: torch.autograd.grad(op, qp, dop, retain_graph=True))
    print(f"{t_plain:.1f} {t_packed:.1f}")
    sys.exit(0)
from octic_vits_amd import build as Bd
variants = sys.argv[1:] or [""]
libs = []
for i, flags in enumerate(variants):
    out = os.path.join(ROOT, "gpurun_out", f"liboctic_ab{i}.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call([Bd.HIPCC, *Bd.FLAGS, "-shared", *flags.split(), "-o", out] + [os.path.join(Bd.CSRC, s) for s in Bd.SOURCES])
    libs.append(out)
res = {v: [] for v in variants}
for rnd in range(3):
    for v, lib in zip(variants, libs):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib], capture_output=True, text=True).stdout.strip().split("\n")[-1]
        res[v].append(tuple(float(x) for x in out.split()))
for v in variants:
    pl = sorted(r[0] for r in res[v]); pk = sorted(r[1] for r in res[v])
    print(f"{v or '(default)':40s} strided rows {pl[1]:6.1f} us (min {pl[0]:6.1f})   packed rows {pk[1]:6.1f} us (min {pk[0]:6.1f})")
