"""Developer tool: where a dense NT tile's time goes.  Builds a side library with -DDG_TRACE2 (wave 0 of every workgroup
stamps s_memrealtime at start / first K-tile / end of K loop / stores issued / stores acknowledged) and prints the
distribution of the phases plus the per-CU timeline (idle gaps between consecutive workgroups of a CU)."""
import ctypes, os, subprocess, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from octic_vits_amd import build as B
CS = B.CSRC
out = os.path.join(ROOT, "gpurun_out", "liboctic_dgt2.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
extra = os.environ.get("FLAGS", "").split()
subprocess.check_call([B.HIPCC, *B.FLAGS, "-shared", "-DDG_TRACE2", *extra, "-o", out] + [os.path.join(CS, s) for s in B.SOURCES])
os.environ["OCTIC_LIB"] = out
import numpy as np
import torch
from octic_vits_amd import ops, _lib
L = _lib.lib()
L.octic_dbg_dense_trace2.restype = ctypes.c_void_p
hip = ctypes.CDLL("libamdhip64.so")
M = 16448
SHAPES = [tuple(int(v) for v in sh.split("x")) for sh in os.environ.get("SHAPES", "5120x1280x0,1280x5120x0,1280x1280x0").split(",")]
for (N, K, mode) in SHAPES:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = (torch.randn(N, K, device="cuda") / 36).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    for _ in range(3):
        ops.dense_gemm_nt(a, b, mode, bias=bias)
    torch.cuda.synchronize()
    src = L.octic_dbg_dense_trace2()
    hip.hipMemset(ctypes.c_void_p(src), 0, 4096 * 8 * 8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.dense_gemm_nt(a, b, mode, bias=bias); e1.record()
    torch.cuda.synchronize()
    buf = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
    hip.hipMemcpy(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(src), 4096 * 8 * 8, 3)
    t = buf.cpu().numpy().reshape(4096, 8)
    live = np.nonzero(t[:, 0])[0]
    t = t[live]
    t0 = t[:, 0].min()
    us = lambda v: v / 100.0
    f = lambda v: f"mean {us(np.mean(v)):6.2f}  p10 {us(np.percentile(v, 10)):6.2f}  median {us(np.median(v)):6.2f}  p90 {us(np.percentile(v, 90)):6.2f}  max {us(np.max(v)):6.2f}"
    print(f"== N {N} K {K} mode {mode}: {len(live)} workgroups, event time {e0.elapsed_time(e1) * 1e3:.1f} us, stamps span {us(t[:, 4].max() - t0):.1f} us")
    print("  prologue (start -> first K-tile)   ", f(t[:, 1] - t[:, 0]))
    print("  K loop                             ", f(t[:, 2] - t[:, 1]))
    print("  epilogue until stores issued       ", f(t[:, 3] - t[:, 2]))
    print("  stores acknowledged (vmcnt 0)      ", f(t[:, 4] - t[:, 3]))
    print("  workgroup total                    ", f(t[:, 4] - t[:, 0]))
    tail = t[:, 5] != 0
    if tail.any():
        tt = t[tail]
        print(f"  split-K tail: {int(tail.sum())} workgroups; first start {us(tt[:, 0].min() - t0):.1f} us, last end {us(tt[:, 4].max() - t0):.1f} us")
        print("    prologue                         ", f(tt[:, 1] - tt[:, 0]))
        print("    K loop                           ", f(tt[:, 2] - tt[:, 1]))
        print("    publish partial (stores drained) ", f(tt[:, 5] - tt[:, 2]))
        print("    wait for the other parts         ", f(tt[:, 6] - tt[:, 5]))
        print("    reduce + epilogue (stores issued)", f(tt[:, 3] - tt[:, 6]))
        print("    start time                       ", f(tt[:, 0] - t0))
    percu = defaultdict(list)
    for i in range(len(live)):
        xcc = int(t[i, 7]) & 0xf
        hw = int(t[i, 7]) >> 32
        percu[(xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf)].append(i)
    gaps, first, last, cnt = [], [], [], []
    for key, v in percu.items():
        v = sorted(v, key=lambda i: t[i, 0])
        first.append(t[v[0], 0] - t0); last.append(t[v[-1], 4] - t0); cnt.append(len(v))
        for x, y in zip(v[:-1], v[1:]):
            gaps.append(t[y, 0] - t[x, 4])
    print(f"  {len(percu)} CUs; workgroups per CU min {min(cnt)} max {max(cnt)}; first start {f(np.array(first))}")
    print(f"  CU finish time                     ", f(np.array(last)))
    if gaps:
        print(f"  gap between workgroups on a CU     ", f(np.array(gaps)))
