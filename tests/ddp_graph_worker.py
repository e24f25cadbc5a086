"""Child process of two tests in tests/test_train_gpu.py (not a test module): the captured data-parallel steps on a one-rank
RCCL group.  They run in a process of their own because a fault in the process group's watchdog thread terminates the whole
process (std::terminate from a C++ thread): that must cost one test, not the pytest run.

    python tests/ddp_graph_worker.py captured_data_parallel | captured_accumulated
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import pytest  # noqa: E402
import torch  # noqa: E402


def run_captured_data_parallel_step_on_a_one_rank_rccl_group():
    """The N > 1 step of bench.py: Trainer(distributed=True) with train.GradReducer (gradients written into flat buckets, one
    RCCL all-reduce per bucket on the group's stream as it fills, no DistributedDataParallel), eagerly and captured as ONE
    hipGraph with the collectives inside.  On a one-rank group the average is the identity, so all three - plain trainer,
    eager reducer step, replayed graph - must agree BITWISE in losses, weights and EMA (same kernels, same order; the
    reducer only changes where gradients are written); and the large gradients must really have landed in the buckets, with
    buckets reduced while the backward pass was still being issued."""
    import os
    import socket

    import torch.distributed as dist

    from octic_vits_amd import train as TR
    from octic_vits_amd.train import GradReducer, Trainer, synthetic_batch
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    old = TR.DDP_FLAT_SMALL_NUMEL
    TR.DDP_FLAT_SMALL_NUMEL = 20_000          # this model's 768 x 256 ... 1024 x 256 weights become bucket members
    try:
        from octic_vits_amd.d8_layers import Layer_scale_init_BlockD8
        from octic_vits_amd.model import OcticVisionTransformer
        from octic_vits_amd.vit import Layer_scale_init_Block

        def make():        # embed_dim 256: the standard half's weight gradients run on csrc/dense_wgrad.hip (destinations honoured)
            torch.manual_seed(5)
            return OcticVisionTransformer(img_size=56, patch_size=14, num_classes=100, embed_dim=256, depth=4, num_heads=4,
                                          qkv_bias=True, drop_path_rate=0.0, octic_block_layers=Layer_scale_init_BlockD8,
                                          standard_block_layers=Layer_scale_init_Block).cuda()
        ma, mb, mc = make(), make(), make()
        ta = Trainer(ma, lr=1e-3)
        tb = Trainer(mb, lr=1e-3, distributed=True, local_rank=0, bucket_cap_mb=1)
        tc = Trainer(mc, lr=1e-3, distributed=True, local_rank=0, bucket_cap_mb=1)
        assert isinstance(tb._reducer, GradReducer) and tb.model is mb and len(tb._reducer.buckets) >= 2
        batches = [synthetic_batch(8, 100, "cuda", seed=s, img_size=56) for s in range(5)]
        warm = batches[0]
        gs = tc.capture(*warm, warmup=2)
        for _ in range(2):
            ta.step(*warm)
            tb.step(*warm)
        la, lb, lc = [], [], []
        for x, y in batches[1:] + batches[1:]:
            la.append(float(ta.step(x, y).detach()))
            lb.append(float(tb.step(x, y).detach()))
            lc.append(float(gs.replay(x, y)))
        assert la == lb == lc, (la, lb, lc)
        assert len(set(la)) == len(la)
        for (n, pa), pb, pc in zip(ma.named_parameters(), mb.parameters(), mc.parameters()):
            assert torch.equal(pa, pb) and torch.equal(pa, pc), n
        for ea, eb, ec in zip(ta.optimizer.ema_state(), tb.optimizer.ema_state(), tc.optimizer.ema_state()):
            assert torch.equal(ea, eb) and torch.equal(ea, ec)
        r = tb._reducer
        assert r.early >= 1                    # at least one bucket was handed to RCCL before the end of the pass
        flats = [b[0] for b in r.buckets] + [r._misc]
        for p in mb.parameters():
            if p.requires_grad:
                assert any(f.data_ptr() <= p.grad.data_ptr() < f.data_ptr() + 4 * f.numel() for f in flats)
        # an eager reducer step after the replays continues from the same state
        x, y = batches[0]
        assert float(ta.step(x, y).detach()) == float(tc.step(x, y).detach())
    finally:
        TR.DDP_FLAT_SMALL_NUMEL = old
        dist.destroy_process_group()


def run_captured_accumulated_step_also_under_the_own_reducer():
    """accum_steps = 2 (BASELINE configs[2]'s shape in miniature): the two forward / backward passes and the optimizer step as
    ONE hipGraph, without and with train.GradReducer on a one-rank RCCL group (first micro-batch written into the buckets, the
    second added in place, the collectives after it) - all bitwise equal to the eager accumulated step of the plain trainer."""
    import os
    import socket

    import torch.distributed as dist

    from octic_vits_amd import train as TR
    from octic_vits_amd.train import Trainer, synthetic_batch
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    old = TR.DDP_FLAT_SMALL_NUMEL
    TR.DDP_FLAT_SMALL_NUMEL = 20_000
    try:
        from octic_vits_amd.d8_layers import Layer_scale_init_BlockD8
        from octic_vits_amd.model import OcticVisionTransformer
        from octic_vits_amd.vit import Layer_scale_init_Block

        def make():
            torch.manual_seed(5)
            return OcticVisionTransformer(img_size=56, patch_size=14, num_classes=100, embed_dim=256, depth=4, num_heads=4,
                                          qkv_bias=True, drop_path_rate=0.0, octic_block_layers=Layer_scale_init_BlockD8,
                                          standard_block_layers=Layer_scale_init_Block).cuda()
        ma, mb, mc = make(), make(), make()
        ta = Trainer(ma, lr=1e-3, accum_steps=2)
        tb = Trainer(mb, lr=1e-3, accum_steps=2)
        tc = Trainer(mc, lr=1e-3, accum_steps=2, distributed=True, local_rank=0, bucket_cap_mb=1)
        assert tc._reducer is not None and tc.model is mc
        batches = [synthetic_batch(8, 100, "cuda", seed=s, img_size=56) for s in range(4)]
        gb = tb.capture(*batches[0], warmup=2)
        gc = tc.capture(*batches[0], warmup=2)
        for _ in range(2):
            ta.step(*batches[0])
        la, lb, lc = [], [], []
        for x, y in batches[1:] + batches[1:]:
            la.append(float(ta.step(x, y).detach()))
            lb.append(float(gb.replay(x, y)))
            lc.append(float(gc.replay(x, y)))
        assert la == lb == lc, (la, lb, lc)
        for (n, pa), pb, pc in zip(ma.named_parameters(), mb.parameters(), mc.parameters()):
            assert torch.equal(pa, pb) and torch.equal(pa, pc), n
        # and the accumulated step equals the one-micro-batch step on the whole batch up to f32 summation order
        md = make()
        td = Trainer(md, lr=1e-3)
        for _ in range(2):
            td.step(*batches[0])
        ld = [float(td.step(x, y).detach()) for x, y in batches[1:] + batches[1:]]
        assert la == pytest.approx(ld, rel=2e-2)
    finally:
        TR.DDP_FLAT_SMALL_NUMEL = old
        dist.destroy_process_group()


if __name__ == "__main__":
    {"captured_data_parallel": run_captured_data_parallel_step_on_a_one_rank_rccl_group,
     "captured_accumulated": run_captured_accumulated_step_also_under_the_own_reducer}[sys.argv[1]]()
    print("worker ok")
