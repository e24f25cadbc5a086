"""Does an MFMA-bound GEMM stream overlap with an HBM-bound row-kernel stream?  Stream A: NW weight-gradient launches
(dense_wgrad_tn 5120 x 1280 over 16448 token rows); stream B: NR LayerNorm-backward row passes over the f32 residual stream
(and optionally the LAMB-like triad of elementwise passes).  HIP-event time of A alone, B alone, A then B on one stream, and A
beside B on two streams - eagerly and as one captured graph.

    python tools/probe_overlap2.py [--nw 4] [--nr 12] [--reps 10]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octic_vits_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nw", type=int, default=4)
    ap.add_argument("--nr", type=int, default=12)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--nt", action="store_true", help="stream A runs forward-type NT GEMMs instead of weight gradients")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    M, D, F = 64 * 257, 1280, 5120
    g = torch.Generator(device=dev).manual_seed(1)
    rn = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    dy, hact, w1 = rn(M, D), rn(M, F), rn(F, D)
    xs = torch.randn(M, D, device=dev, generator=g)
    gy = rn(M, D)
    lnw = torch.ones(D, device=dev)
    stats = ops.dense_layernorm_fwd(xs, lnw, lnw, 1e-6, torch.bfloat16)[1]
    side = torch.cuda.Stream()

    def A():
        for _ in range(a.nw):
            if a.nt:
                ops.dense_gemm_nt(dy, w1)
            else:
                ops.dense_wgrad_tn(dy, hact)

    def B():
        for _ in range(a.nr):
            ops.dense_layernorm_bwd(gy, xs, lnw, stats, None)

    def serial():
        A()
        B()

    def forked():
        main_s = torch.cuda.current_stream()
        side.wait_stream(main_s)
        with torch.cuda.stream(side):
            B()
        A()
        main_s.wait_stream(side)

    def timed(fn, graph):
        run = fn
        if graph:
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
        if graph:
            run()
        else:
            for _ in range(a.reps):
                run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / a.reps

    for graph in (False, True):
        for rnd in range(2):
            ta, tb, ts, tf = (timed(f, graph) for f in (A, B, serial, forked))
            print(f"{'graph' if graph else 'eager'} round {rnd}: A {ta:7.1f} us  B {tb:7.1f} us  A;B {ts:7.1f} us  A||B {tf:7.1f} us "
                  f"({(tf / ts - 1) * 100:+.1f} %)", flush=True)


if __name__ == "__main__":
    main()
