"""Developer tool: timeline of the X-stationary GEMM from s_memtime stamps (library built with -DOCTIC_XREG_TRACE,
path in OCTIC_LIB).  Prints per-phase cycle statistics for fc1 at the ViT-H shape."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octic_vits_amd import _lib, ops  # noqa: E402

_lib.LIB_PATH = os.environ.get("OCTIC_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "liboctic_trace.so"))
L = _lib.lib()
L.octic_dbg_xreg_trace.restype = ctypes.c_void_p
B, T, c = 64, 257, 160
cin, cout = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (c, 4 * c)
M = B * T
bf = torch.bfloat16
x = torch.randn(B, T, 8 * cin, device="cuda").to(bf)
w = [torch.randn(cout, cin, device="cuda").to(bf) * 0.05 for _ in range(4)] + [torch.randn(2 * cout, 2 * cin, device="cuda").to(bf) * 0.05]
y = torch.empty(B, T, 8 * cout, device="cuda", dtype=bf)
for _ in range(3):
    ops.linear_fwd(ops.pview(x, cin), w, None, ops.pview(y, cout), M, cin, cout, bf, bf, x)
torch.cuda.synchronize()
n = 256 * 4 * 64
buf = torch.zeros(n, dtype=torch.int64, device="cuda")
hip = ctypes.CDLL("libamdhip64.so")
src = L.octic_dbg_xreg_trace()
hip.hipMemset(ctypes.c_void_p(src), 0, n * 8)
ops.linear_fwd(ops.pview(x, cin), w, None, ops.pview(y, cout), M, cin, cout, bf, bf, x)
torch.cuda.synchronize()
hip.hipMemcpy(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(src), n * 8, 3)
tr = buf.cpu().numpy().reshape(256, 4, 64)
for wg in (0, 1, 100, 200, 255):
    t = tr[wg, 0]
    nz = np.nonzero(t)[0]
    if len(nz) < 3:
        continue
    t0 = t[0]
    steps = (nz.max() - 1) // 3
    print(f"WG {wg}: events {len(nz)}, steps {steps}, total {t[nz.max()] - t0} cycles")
    print("   X landed  +%d" % (t[1] - t0))
    for s in range(steps):
        a, b_, c_ = t[2 + 3 * s], t[3 + 3 * s], t[4 + 3 * s]
        prev = t[1] if s == 0 else t[4 + 3 * (s - 1)]
        print(f"   step {s:2d}: gap(prev end->wait done) {a - prev:6d}  barrier {b_ - a:6d}  issue+mfma {c_ - b_:6d}")
# aggregate over all traced WGs / waves
gaps, bars, comp, xl, tot = [], [], [], [], []
for wg in range(256):
    for wv in range(4):
        t = tr[wg, wv]
        nz = np.nonzero(t)[0]
        if len(nz) < 5:
            continue
        steps = (nz.max() - 1) // 3
        xl.append(t[1] - t[0])
        tot.append(t[nz.max()] - t[0])
        for s in range(steps):
            prev = t[1] if s == 0 else t[4 + 3 * (s - 1)]
            gaps.append(t[2 + 3 * s] - prev)
            bars.append(t[3 + 3 * s] - t[2 + 3 * s])
            comp.append(t[4 + 3 * s] - t[3 + 3 * s])
f = lambda v: f"mean {np.mean(v):8.0f}  median {np.median(v):8.0f}  p90 {np.percentile(v, 90):8.0f}"
print("X load      ", f(xl))
print("wait (gap)  ", f(gaps))
print("barrier     ", f(bars))
print("issue+mfma  ", f(comp))
print("wave total  ", f(tot), " steps/wave", len(gaps) / max(1, len(tot)))
