"""Do the weight-gradient launches of a standard block's MLP backward overlap with the input-gradient GEMMs when
they run on a second stream?  (dense_tn_kernel leaves 22 % of the CUs without a workgroup on the S = 2 shapes; the NT
kernels end in partial rounds.)  Same four kernels, back to back on one stream vs the two TN launches on a side stream
with event dependencies; HIP-event time of REPS rounds.

    python tools/probe_overlap.py [--reps 20] [--graph]
"""
import argparse
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octic_vits_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--graph", action="store_true", help="time a captured graph of the rounds instead of eager launches")
    ap.add_argument("--rows", action="store_true", help="add a row kernel (LayerNorm backward) after the MLP GEMMs")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    M, D, F = 64 * 257, 1280, 5120
    g = torch.Generator(device=dev).manual_seed(1)
    rn = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    dy, hpre, hact, x = rn(M, D), rn(M, F), rn(M, F), rn(M, D)
    w2t, w1t = rn(F, D), rn(D, F)          # the transposed copies the input-gradient NT problems read
    xs = torch.randn(M, D, device=dev, generator=g)
    lnw = torch.ones(D, device=dev)
    side = torch.cuda.Stream()
    stats = ops.dense_layernorm_fwd(xs, lnw, lnw, 1e-6, torch.bfloat16)[1]

    def rows(gy):
        if a.rows:
            return ops.dense_layernorm_bwd(gy, xs, lnw, stats, None)
        return None

    def serial():
        dw2 = ops.dense_wgrad_tn(dy, hact)
        dh = ops.dense_gemm_nt(dy, w2t, mode=3, h=hpre)
        dw1 = ops.dense_wgrad_tn(dh, x)
        dx = ops.dense_gemm_nt(dh, w1t)
        rows(dx)
        return dw2, dw1, dx

    def forked():
        main_s = torch.cuda.current_stream()
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            dw2 = ops.dense_wgrad_tn(dy, hact)
        dh = ops.dense_gemm_nt(dy, w2t, mode=3, h=hpre)
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            dw1 = ops.dense_wgrad_tn(dh, x)
        dx = ops.dense_gemm_nt(dh, w1t)
        rows(dx)
        main_s.wait_stream(side)
        return dw2, dw1, dx

    ref = serial()
    got = forked()
    torch.cuda.synchronize()
    for r, o in zip(ref, got):
        assert torch.equal(r, o), "forked result differs"

    def timed(fn):
        run = fn
        if a.graph:
            gr = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                fn()
                torch.cuda.synchronize()
                with torch.cuda.graph(gr, stream=s):
                    for _ in range(a.reps):
                        fn()
            run = gr.replay
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if a.graph:
            run()
        else:
            for _ in range(a.reps):
                run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / a.reps

    for rnd in range(3):
        ts, tf = timed(serial), timed(forked)
        print(f"round {rnd}: one stream {ts:8.1f} us   TN on a side stream {tf:8.1f} us   ({(tf / ts - 1) * 100:+.1f} %)", flush=True)


if __name__ == "__main__":
    main()
