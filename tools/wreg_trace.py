"""Developer tool: timeline of the W-stationary LinearD8 kernel from clock stamps (library built with -DOCTIC_WREG_TRACE,
path in OCTIC_LIB; tools/wreg_variants.py builds one as "trace:-DOCTIC_WREG_TRACE").  fc1 at the ViT-H shape by default."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octic_vits_amd import _lib, ops  # noqa: E402

L = _lib.lib()
L.octic_dbg_wreg_trace.restype = ctypes.c_void_p
B, T, c = 64, 257, 160
cin, cout = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (c, 4 * c)
M = B * T
bf = torch.bfloat16
x = torch.randn(B, T, 8 * cin, device="cuda").to(bf)
w = [torch.randn(cout, cin, device="cuda").to(bf) * 0.05 for _ in range(4)] + [torch.randn(2 * cout, 2 * cin, device="cuda").to(bf) * 0.05]
y = torch.empty(B, T, 8 * cout, device="cuda", dtype=bf)
call = lambda: ops.linear_fwd(ops.pview(x, cin), w, None, ops.pview(y, cout), M, cin, cout, bf, bf, x)
for _ in range(3): call()
torch.cuda.synchronize()
n = 1024 * 4 * 128
buf = torch.zeros(n, dtype=torch.int64, device="cuda")
hip = ctypes.CDLL("libamdhip64.so")
src = L.octic_dbg_wreg_trace()
hip.hipMemset(ctypes.c_void_p(src), 0, n * 8)
call()
torch.cuda.synchronize()
hip.hipMemcpy(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(src), n * 8, 3)
tr = buf.cpu().numpy().reshape(1024, 4, 128)
live = [i for i in range(1024) if tr[i, 0, 0] != 0]
t00 = min(tr[i, 0, 0] for i in live)
print(f"{len(live)} workgroups traced")
def show(wg):
    t = tr[wg, 0]
    nz = np.nonzero(t)[0]
    steps = (nz.max() - 1) // 3
    print(f"WG {wg}: start +{t[0] - t00}, prologue {t[1] - t[0]}, {steps} steps traced, end of last traced +{t[nz.max()] - t00}")
    for s in range(min(steps, 8)):
        a, b_, c_ = t[2 + 3 * s], t[3 + 3 * s], t[4 + 3 * s]
        prev = t[1] if s == 0 else t[4 + 3 * (s - 1)]
        print(f"   step {s:2d}: wait {a - prev:6d}  barrier {b_ - a:6d}  work {c_ - b_:6d}")
for wg in (live[0], live[len(live) // 3], live[-1]):
    show(wg)
st, pro, wa, ba, wo, end = [], [], [], [], [], []
for wg in live:
    for wv in range(4):
        t = tr[wg, wv]
        nz = np.nonzero(t)[0]
        if len(nz) < 5: continue
        steps = (nz.max() - 1) // 3
        st.append(t[0] - t00); pro.append(t[1] - t[0]); end.append(t[nz.max()] - t00)
        for s in range(steps):
            prev = t[1] if s == 0 else t[4 + 3 * (s - 1)]
            wa.append(t[2 + 3 * s] - prev); ba.append(t[3 + 3 * s] - t[2 + 3 * s]); wo.append(t[4 + 3 * s] - t[3 + 3 * s])
f = lambda v: f"mean {np.mean(v):8.0f}  median {np.median(v):8.0f}  p90 {np.percentile(v, 90):8.0f}  max {np.max(v):8.0f}"
print("start      ", f(st)); print("prologue   ", f(pro)); print("wait       ", f(wa)); print("barrier    ", f(ba)); print("work       ", f(wo))
print("last stamp ", f(end))

# per-XCD span (the counters of different XCDs have different origins): items are XCD-contiguous
nl = len(live)
q8, r8 = nl // 8, nl % 8
lo = 0
for xcd in range(8):
    cnt = q8 + (1 if xcd < r8 else 0)
    items = live[lo:lo + cnt]; lo += cnt
    s0 = np.array([tr[i, :, 0].min() for i in items]); e1 = []
    for i in items:
        t = tr[i]; e1.append(t.max())
    e1 = np.array(e1)
    base = s0.min()
    print(f"XCD {xcd}: {cnt} WGs, start spread {s0.max() - base}, first end +{e1.min() - base}, last stamp +{e1.max() - base}")

# workgroup life (first stamp -> last stamp, 20 steps traced at most: E workgroups are truncated) by item decile
life = np.array([tr[i].max() - tr[i, :, 0].min() for i in live])
nst = np.array([(np.nonzero(tr[i, 0])[0].max() - 1) // 3 for i in live])
for d in range(10):
    sl = slice(d * nl // 10, (d + 1) * nl // 10)
    print(f"items {sl.start:4d}-{sl.stop:4d}: traced life mean {life[sl].mean():8.0f}  max {life[sl].max():8.0f}  steps traced {nst[sl].mean():.1f}")
