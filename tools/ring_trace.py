"""Developer tool: timeline of the ring (long-K) LinearD8 kernel from clock stamps (library built with -DOCTIC_RING_TRACE,
path in OCTIC_LIB; SRC=gemm tools/wreg_variants.py "trace:-DOCTIC_RING_TRACE" runs it).  fc2 at the ViT-H shape by default."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octic_vits_amd import _lib, ops  # noqa: E402

L = _lib.lib()
L.octic_dbg_ring_trace.restype = ctypes.c_void_p
B, T, c = 64, 257, 160
cin, cout = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4 * c, c)
M = B * T
bf = torch.bfloat16
x = torch.randn(B, T, 8 * cin, device="cuda").to(bf)
w = [torch.randn(cout, cin, device="cuda").to(bf) * 0.05 for _ in range(4)] + [torch.randn(2 * cout, 2 * cin, device="cuda").to(bf) * 0.05]
y = torch.empty(B, T, 8 * cout, device="cuda", dtype=bf)
call = lambda: ops.linear_fwd(ops.pview(x, cin), w, None, ops.pview(y, cout), M, cin, cout, bf, bf, x)
for _ in range(3): call()
torch.cuda.synchronize()
n = 2048 * 4 * 64
buf = torch.zeros(n, dtype=torch.int64, device="cuda")
hip = ctypes.CDLL("libamdhip64.so")
src = L.octic_dbg_ring_trace()
hip.hipMemset(ctypes.c_void_p(src), 0, n * 8)
call()
torch.cuda.synchronize()
hip.hipMemcpy(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(src), n * 8, 3)
tr = buf.cpu().numpy().reshape(2048, 4, 64).copy()
ids = tr[:, 0, 63].copy()
tr[:, :, 63] = 0
end = tr[:, :, 62].copy()          # stamp after the epilogue
tr[:, :, 62] = 0
live = [i for i in range(2048) if tr[i, 0, 0] != 0]
print(f"{len(live)} workgroups traced")
# where each workgroup ran: XCC id, and (se, sh, cu) from HW_ID
from collections import defaultdict
percu = defaultdict(list)
for i in live:
    xcc = int(ids[i]) & 0xf
    hw = int(ids[i]) >> 32
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
    percu[(xcc, se, sh, cu)].append(i)
print(f"{len(percu)} distinct (xcc, se, sh, cu); workgroups per CU: min {min(len(v) for v in percu.values())} max {max(len(v) for v in percu.values())}")
xcc_of_items = [int(ids[i]) & 0xf for i in live[:24]]
print("xcc of the first 24 items:", xcc_of_items)
for key in list(sorted(percu))[:3] + list(sorted(percu))[-2:]:
    v = percu[key]
    base = min(tr[i, :, 0].min() for i in v)
    print(key, [(i, int(tr[i, :, 0].min() - base), int(tr[i].max() - base)) for i in sorted(v, key=lambda i: tr[i, :, 0].min())])
spans = []
for key, v in percu.items():
    base = min(tr[i, :, 0].min() for i in v)
    spans.append(max(max(tr[i].max(), end[i].max()) for i in v) - base)
print(f"per-CU span (first start -> last stamp): mean {np.mean(spans):.0f}  max {np.max(spans):.0f}")
# per-step statistics (wave 0 of every workgroup)
wa, ba, wo, pro = [], [], [], []
for wg in live:
    t = tr[wg, 0]
    nz = np.nonzero(t)[0]
    if len(nz) < 5: continue
    steps = (nz.max() - 1) // 3
    pro.append(t[1] - t[0])
    for s_ in range(steps):
        prev = t[1] if s_ == 0 else t[4 + 3 * (s_ - 1)]
        wa.append(t[2 + 3 * s_] - prev); ba.append(t[3 + 3 * s_] - t[2 + 3 * s_]); wo.append(t[4 + 3 * s_] - t[3 + 3 * s_])
f = lambda v: f"mean {np.mean(v):8.0f}  median {np.median(v):8.0f}  p90 {np.percentile(v, 90):8.0f}  max {np.max(v):8.0f}"
print("prologue   ", f(pro)); print("wait       ", f(wa)); print("barrier    ", f(ba)); print("work       ", f(wo))
epi = [int(end[wg, 0] - tr[wg, 0].max()) for wg in live if end[wg, 0]]
if epi: print("epilogue   ", f(epi))
for wg in (live[0], live[-1]):
    t = tr[wg, 0]; nz = np.nonzero(t)[0]; steps = (nz.max() - 1) // 3
    print(f"WG {wg}: {steps} steps; (wait, barrier, work):", [(int(t[2 + 3 * s_] - (t[1] if s_ == 0 else t[4 + 3 * (s_ - 1)])), int(t[3 + 3 * s_] - t[2 + 3 * s_]), int(t[4 + 3 * s_] - t[3 + 3 * s_])) for s_ in range(steps)])
