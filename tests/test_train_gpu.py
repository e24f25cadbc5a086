"""GPU checks of the training-step plumbing: fused LAMB+EMA HIP step vs the foreach reference implementation,
and one full train step of a small hybrid model (loss finite, parameters move, EMA follows)."""
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu


def _toy_params(dev):
    g = torch.Generator().manual_seed(5)
    shapes = [(7,), (33, 17), (160, 160), (1000, 130), (3,), (70001,)]
    return [torch.randn(*s, generator=g).to(dev).requires_grad_(True) for s in shapes]


def test_fused_lamb_matches_foreach_lamb_and_ema():
    from octic_vits_amd.train import FusedLamb, Lamb, ModelEma
    dev = "cuda"
    pa, pb = _toy_params(dev), _toy_params(dev)

    def groups(ps):
        return [{"params": [p for p in ps if p.ndim <= 1], "weight_decay": 0.0},
                {"params": [p for p in ps if p.ndim > 1], "weight_decay": 0.02}]

    ref = Lamb(groups(pa), lr=3e-3, weight_decay=0.02)
    order_a = [p for g in ref.param_groups for p in g["params"]]

    class _M:  # ModelEma only needs .parameters()
        def __init__(self, ps): self.ps = ps
        def parameters(self): return self.ps
    ema = ModelEma(_M(order_a), decay=0.9)
    fused = FusedLamb(groups(pb), lr=3e-3, ema_decay=0.9)
    gen = torch.Generator().manual_seed(11)
    for step in range(3):
        for a, b in zip(order_a, fused.params):
            gr = (torch.randn(a.shape, generator=gen) * (5.0 if step == 0 else 0.1)).to(dev)   # step 0 triggers the clip
            a.grad, b.grad = gr.clone(), gr.clone()
        ref.step()
        ema.update(_M(order_a))
        fused.step()
        assert torch.allclose(fused.last_grad_norm, ref.last_grad_norm, rtol=1e-5)
        for a, b in zip(order_a, fused.params):
            assert torch.allclose(a, b, rtol=2e-5, atol=1e-6), f"step {step}: param mismatch {(a - b).abs().max()}"
        for e_ref, e_f in zip(ema.params, fused.ema_state()):
            assert torch.allclose(e_ref, e_f, rtol=2e-5, atol=1e-6)


def test_train_step_small_hybrid_model():
    from octic_vits_amd.d8_layers import Layer_scale_init_BlockD8
    from octic_vits_amd.model import OcticVisionTransformer
    from octic_vits_amd.train import Trainer, synthetic_batch
    from octic_vits_amd.vit import Layer_scale_init_Block
    torch.manual_seed(0)
    net = OcticVisionTransformer(img_size=56, patch_size=14, num_classes=100, embed_dim=128, depth=4, num_heads=4,
                                 qkv_bias=True, drop_path_rate=0.5, octic_block_layers=Layer_scale_init_BlockD8,
                                 standard_block_layers=Layer_scale_init_Block).cuda()
    before = [p.detach().clone() for p in net.parameters()]
    tr = Trainer(net)
    x, y = synthetic_batch(8, 100, "cuda", 3, img_size=56)
    losses = [float(tr.step(x, y).detach()) for _ in range(5)]
    assert all(l == l and l < 10 for l in losses)
    assert losses[-1] < losses[0]          # lr 3e-3 LAMB on a fixed batch must make progress
    # every weight matrix / embedding must have moved (1-D LayerNorm scales of the octic MLP branch have ~1e-14
    # gradients at init — layer-scale 1e-4 enters twice — which is below the f32 resolution of their value 1.0)
    moved = [not torch.equal(a, b) for a, b in zip(before, net.parameters()) if b.requires_grad and b.ndim >= 2]
    assert all(moved) and len(moved) > 50
    frozen = [torch.equal(a, b) for a, b in zip(before, net.parameters()) if not b.requires_grad]
    assert all(frozen)                     # cls_token.1-4 stay frozen zeros (reference model.py:99-105)


def test_weight_caches_follow_fused_optimizer():
    """The fused LAMB step writes parameters through raw pointers; the compute-dtype weight caches must notice."""
    from octic_vits_amd import functional as OF
    from octic_vits_amd.d8_layers import Layer_scale_init_BlockD8
    from octic_vits_amd.model import OcticVisionTransformer
    from octic_vits_amd.train import Trainer, synthetic_batch
    from octic_vits_amd.vit import Layer_scale_init_Block
    torch.manual_seed(1)
    net = OcticVisionTransformer(img_size=56, patch_size=14, num_classes=100, embed_dim=128, depth=4, num_heads=4,
                                 qkv_bias=True, drop_path_rate=0.0, octic_block_layers=Layer_scale_init_BlockD8,
                                 standard_block_layers=Layer_scale_init_Block).cuda()
    tr = Trainer(net)
    x, y = synthetic_batch(8, 100, "cuda", 5, img_size=56)
    for _ in range(3):
        tr.step(x, y)
    net.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        cached = net(x).float()
        for m in net.modules():                      # drop every cache: the next forward re-derives from the masters
            for k, v in vars(m).items():
                if isinstance(v, (OF.WeightPrep, OF.DenseWeightCache)):
                    setattr(m, k, type(v)())
        fresh = net(x).float()
    assert torch.equal(cached, fresh)


def test_ddp_wrapped_trainer_matches_plain_trainer():
    """The data-parallel path (DistributedDataParallel over RCCL, gradients as bucket views, fused LAMB reading the
    bucket views, bf16 weight copies handed to the caches) on a one-rank process group: same losses as the plain
    trainer.  More than one rank cannot run on a one-GPU box; tests/test_distributed_cpu.py covers world_size 2."""
    import os
    import socket

    import torch.distributed as dist

    from octic_vits_amd.d8_layers import Layer_scale_init_BlockD8
    from octic_vits_amd.model import OcticVisionTransformer
    from octic_vits_amd.train import Trainer, synthetic_batch
    from octic_vits_amd.vit import Layer_scale_init_Block

    def make():
        torch.manual_seed(7)
        return OcticVisionTransformer(img_size=56, patch_size=14, num_classes=100, embed_dim=128, depth=4, num_heads=4,
                                      qkv_bias=True, drop_path_rate=0.0, octic_block_layers=Layer_scale_init_BlockD8,
                                      standard_block_layers=Layer_scale_init_Block).cuda()

    x, y = synthetic_batch(8, 100, "cuda", 11, img_size=56)
    plain = Trainer(make())
    want = [float(plain.step(x, y).detach()) for _ in range(4)]

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        ddp = Trainer(make(), distributed=True, local_rank=0)
        got = [float(ddp.step(x, y).detach()) for _ in range(4)]
    finally:
        dist.destroy_process_group()
    assert got == pytest.approx(want, rel=1e-5, abs=1e-6)
    assert want[-1] < want[0]


def test_fused_lamb_matches_independent_restatement():
    """HIP LAMB+EMA against oracle/lamb_ref.py (numpy float64, element by element, no shared code) — three steps, the
    first one clipped; f32 state vs f64 reference: 2e-5."""
    import numpy as np

    from octic_vits_amd.train import FusedLamb
    from oracle.lamb_ref import LambRef, ema_update
    ps = _toy_params("cuda")
    groups = [{"params": [p for p in ps if p.ndim <= 1], "weight_decay": 0.0},
              {"params": [p for p in ps if p.ndim > 1], "weight_decay": 0.02}]
    fused = FusedLamb(groups, lr=3e-3, eps=1e-8, ema_decay=0.9)
    order = fused.params
    ref = LambRef([tuple(p.shape) for p in order], [0.0 if p.ndim <= 1 else 0.02 for p in order], lr=3e-3, eps=1e-8)
    cur = [p.detach().double().cpu().numpy() for p in order]
    ema = [c.copy() for c in cur]
    gen = torch.Generator().manual_seed(11)
    for step in range(3):
        grads = [torch.randn(p.shape, generator=gen) * (5.0 if step == 0 else 0.1) for p in order]
        for p, g in zip(order, grads):
            p.grad = g.cuda()
        fused.step()
        cur = ref.step(cur, [g.double().numpy() for g in grads])
        ema = ema_update(ema, cur, 0.9)
        assert abs(float(fused.last_grad_norm) - ref.last_grad_norm) <= 1e-5 * ref.last_grad_norm
        for p, q, e_f, e_r in zip(order, cur, fused.ema_state(), ema):
            assert np.allclose(p.detach().cpu().numpy(), q, rtol=2e-5, atol=1e-6), f"step {step}"
            assert np.allclose(e_f.cpu().numpy(), e_r, rtol=2e-5, atol=1e-6), f"step {step} ema"


def test_fused_lamb_refuses_non_finite_step():
    """A NaN/Inf gradient norm must leave parameters, moments, EMA and bf16 copies untouched (the reference exits
    before optimizer.step(), deit/engine.py:67-71) and must not advance the bias-correction step."""
    from octic_vits_amd.train import FusedLamb
    ps = _toy_params("cuda")
    fused = FusedLamb([{"params": ps, "weight_decay": 0.02}], lr=3e-3, ema_decay=0.9)
    gen = torch.Generator().manual_seed(2)
    for p in ps:
        p.grad = torch.randn(p.shape, generator=gen).cuda()
    fused.step()
    before = [p.detach().clone() for p in ps]
    m0, v0, e0 = fused.m.clone(), fused.v.clone(), fused.ema.clone()
    for i, p in enumerate(ps):
        p.grad = torch.randn(p.shape, generator=gen).cuda()
        if i == 2:
            p.grad[3, 5] = float("nan")
    fused.step()
    assert fused.skipped_steps == 1 and float(fused.ws[2]) == 1.0 and float(fused.ws[3]) == 1.0
    assert all(torch.equal(a, b) for a, b in zip(before, ps))
    assert torch.equal(m0, fused.m) and torch.equal(v0, fused.v) and torch.equal(e0, fused.ema)
    for p in ps:
        p.grad = torch.randn(p.shape, generator=gen).cuda()
    fused.step()
    assert float(fused.ws[2]) == 0.0 and float(fused.ws[3]) == 2.0 and not torch.equal(before[1], ps[1])


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_trainer_reproduces_reference_train_fixture(mode):
    """the HIP Trainer (engine kernels + fused LAMB/EMA) against tests/golden/train_hybrid.npz — losses
    before each of three optimizer steps and after the last, gradient norms, parameter and EMA samples, produced by the
    REAL reference model on CPU (optimizer: restated apex LAMB).  f32 path: 1e-3 (north-star forward tolerance);
    bf16 autocast: 3e-2 — three compounding steps of bf16-rounded activations (rel 2^-9 each) through a depth-4 model."""
    import train_case
    from test_lamb_oracle import check_against_fixture
    from octic_vits_amd import d8_layers, model as M, vit
    import types
    ns = types.SimpleNamespace(OcticVisionTransformer=M.OcticVisionTransformer,
                               Layer_scale_init_BlockD8=d8_layers.Layer_scale_init_BlockD8,
                               Layer_scale_init_Block=vit.Layer_scale_init_Block)
    from octic_vits_amd.train import Trainer
    net = train_case.build(ns).cuda()
    names = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
    tr = Trainer(net, lr=train_case.LR, weight_decay=train_case.WD, ema_decay=train_case.EMA_DECAY, opt_eps=train_case.EPS,
                 autocast=(mode == "bf16"), tuned_gemms=False)
    x, y = train_case.batch()
    x, y = x.cuda(), y.cuda()
    losses, gnorms = [], []
    for _ in range(train_case.STEPS):
        losses.append(float(tr.step(x, y).detach()))
        gnorms.append(float(tr.optimizer.last_grad_norm))
    net.train()
    with torch.no_grad():
        if mode == "bf16":
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = net(x).float()
        else:
            out = net(x)
        losses.append(float(torch.nn.functional.binary_cross_entropy_with_logits(out, y)))
    ema_by_id = {id(p): e for p, e in zip(tr.optimizer.params, tr.optimizer.ema_state())}
    res = train_case.summarize(names, losses, gnorms, [ema_by_id[id(p)].cpu().numpy() for _, p in names])
    check_against_fixture(res, 1e-3 if mode == "f32" else 3e-2)


def test_weight_caches_need_invalidation_after_raw_data_writes():
    """an optimizer that writes through p.data (apex FusedLAMB does) leaves `_version` alone, so the
    compute-dtype weight caches cannot see the update; `invalidate_weight_caches` / `track_optimizer` is the documented
    hook for such optimizers."""
    from octic_vits_amd import functional as OF
    from octic_vits_amd.d8_layers import Layer_scale_init_BlockD8
    from octic_vits_amd.model import OcticVisionTransformer
    from octic_vits_amd.vit import Layer_scale_init_Block
    torch.manual_seed(3)
    net = OcticVisionTransformer(img_size=56, patch_size=14, num_classes=10, embed_dim=128, depth=4, num_heads=4,
                                 qkv_bias=True, octic_block_layers=Layer_scale_init_BlockD8,
                                 standard_block_layers=Layer_scale_init_Block).cuda().eval()
    x = torch.randn(2, 3, 56, 56, device="cuda")

    def fwd():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return net(x).float()
    y0 = fwd()
    with torch.no_grad():
        for p in net.parameters():
            if p.ndim >= 2:
                p.data.mul_(1.5)                 # raw write: no version bump
    n = net.invalidate_weight_caches()
    assert n >= 16
    y1 = fwd()
    assert not torch.allclose(y0, y1)
    # and the hook form: a torch.optim optimizer stepping through .data
    opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=0.0)
    OF.track_optimizer(net, opt)
    with torch.no_grad():
        for p in net.parameters():
            if p.ndim >= 2:
                p.data.mul_(1.0 / 1.5)
    for p in opt.param_groups[0]["params"]:
        p.grad = torch.zeros_like(p)
    opt.step()
    assert torch.allclose(fwd(), y0, atol=2e-2, rtol=2e-2)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("buckets", ["f32", "bf16"])
def test_two_rank_ddp_on_one_gpu(tmp_path, buckets):
    """Two fresh child processes (never an exec of this GPU-initialised process), both on cuda:0, gloo between them:
    Trainer(distributed=True) with the fused LAMB reading DDP's bucket views.  Replicas must stay identical and equal
    single-process training on the concatenated batch (f32 buckets: 1e-4; bf16-compressed buckets: 2e-2)."""
    import os
    import socket
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    worker = os.path.join(here, "ddp_gpu_worker.py")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    single, ddp = str(tmp_path / "single.pt"), str(tmp_path / "ddp.pt")
    r = subprocess.run([sys.executable, worker, "1", "0", str(port), single], env=env, capture_output=True, text=True,
                       timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    extra = ["bf16"] if buckets == "bf16" else []
    procs = [subprocess.Popen([sys.executable, worker, "2", str(rk), str(port), ddp] + extra, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for rk in range(2)]
    outs = [p.communicate(timeout=400) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    a, b = torch.load(single), torch.load(ddp)
    tol = 1e-4 if buckets == "f32" else 2e-2
    assert b["skipped"] == 0
    assert a["losses"] == pytest.approx(b["losses"], rel=tol, abs=tol)
    # bf16-compressed buckets round every gradient to 8 bits: elements whose gradient is rounding noise (the K third
    # of the qkv biases: exactly zero in exact arithmetic) are moved by Adam's normalisation by up to lr per step in a
    # rounding-dependent direction -> 2 * lr * steps absolute for that mode
    atol = 1e-5 if buckets == "f32" else 2 * 3e-3 * 3
    assert torch.allclose(a["params"], b["params"], rtol=tol, atol=atol), (a["params"] - b["params"]).abs().max()


@pytest.mark.timeout(900)
def test_invariant_vit_huge_full_size():
    """BASELINE configs[3] at its real size: d8_inv_early_deit_huge_patch14 (16 octic blocks + PowerSpectrum hand-off +
    16 standard blocks, 357 M parameters), one bf16 forward + backward on 8 images: finite, and the logits are
    invariant under all 8 group elements acting on the image (reference test_invariance_deit_inv_early,
    experiments/test_equivariance.py:302-322, at bf16 tolerance: 3e-2 of the logit scale)."""
    from octic_vits_amd.deit_models import create_model
    from oracle import octic_ref as R
    torch.manual_seed(5)
    net = create_model("d8_inv_early_deit_huge_patch14", num_classes=1000, drop_path_rate=0.0, img_size=224).cuda()
    x = torch.randn(8, 3, 224, 224, device="cuda")
    net.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = net(x)
        loss = out.float().square().mean()
    loss.backward()
    assert torch.isfinite(out).all()
    gn = torch.stack([p.grad.float().norm() for p in net.parameters() if p.grad is not None])
    assert torch.isfinite(gn).all() and float(gn.max()) > 0
    net.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        base = net(x[:2]).float()
        scale = max(1e-3, float(base.abs().max()))
        for g in ("r", "rr", "rrr", "m", "mr", "mrr", "mrrr"):
            got = net(R.image_space_group_action(g, x[:2]).contiguous()).float()
            err = float((got - base).abs().max())
            assert err <= 3e-2 * scale, f"invariance under {g}: {err:.3e} vs scale {scale:.3e}"


def test_captured_step_replays_like_eager_steps():
    """Trainer.capture: forward + backward + LAMB/EMA as one hipGraph.  Two identically initialised trainers, one
    stepping eagerly and one replaying its captured graph on the same batches, must produce the same losses and weights (same kernels,
    same order); the device-side step counter keeps the bias correction advancing."""
    from octic_vits_amd.model import OcticVisionTransformer
    from octic_vits_amd.train import Trainer, synthetic_batch
    kw = dict(img_size=32, patch_size=4, in_chans=3, num_classes=10, embed_dim=128, depth=4, num_heads=2,
              mlp_ratio=4.0, drop_path_rate=0.0, octic_equi_break_layer=2)
    torch.manual_seed(0)
    ma = OcticVisionTransformer(**kw).cuda()
    torch.manual_seed(0)
    mb = OcticVisionTransformer(**kw).cuda()
    ta, tb = Trainer(ma, lr=1e-3), Trainer(mb, lr=1e-3)
    batches = [synthetic_batch(8, 10, "cuda", seed=s, img_size=32) for s in range(5)]
    warm = batches[0]
    gs = tb.capture(*warm, warmup=2)
    for _ in range(2):
        ta.step(*warm)
    la, lb = [], []
    for x, y in batches[1:] + batches[1:]:
        la.append(float(ta.step(x, y).detach()))
        lb.append(float(gs.replay(x, y)))
    # BITWISE: every kernel of the step sums in a fixed order (slab reductions, split-K tails, the pos-embed gradient is
    # a gather over inverse index tables, d8_utils._GatherUnfoldFn) - graph replay and eager launches run the same kernels
    assert la == lb, (la, lb)
    assert len(set(la)) == len(la)                 # the weights did move between the steps
    for (n, pa), pb in zip(ma.named_parameters(), mb.parameters()):
        assert torch.equal(pa, pb), n
    for ea, eb in zip(ta.optimizer.ema_state(), tb.optimizer.ema_state()):
        assert torch.equal(ea, eb)
    # an eager step after the replays continues from the same state (the caches were refreshed on the device)
    x, y = batches[0]
    assert float(ta.step(x, y).detach()) == float(tb.step(x, y).detach())
    assert float(ta.step(x, y).detach()) == float(gs.replay(x, y))


def _run_ddp_graph_worker(which):
    """tests/ddp_graph_worker.py in a child process; ONE retry if the child was terminated by a signal (abort / segfault from a
    runtime thread), none on an ordinary failure."""
    import os
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ddp_graph_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for attempt in range(2):
        r = subprocess.run([sys.executable, worker, which], env=env, capture_output=True, text=True, timeout=500)
        if r.returncode == 0:
            assert "worker ok" in r.stdout
            return
        if r.returncode not in (-6, -11, 134, 139) or attempt == 1:
            raise AssertionError(f"{which}: exit code {r.returncode}\n" + (r.stdout or "")[-1500:] + (r.stderr or "")[-3000:])


@pytest.mark.timeout(1100)
def test_captured_data_parallel_step_on_a_one_rank_rccl_group():
    """The N > 1 step of bench.py: Trainer(distributed=True) with train.GradReducer (gradients written into flat buckets, one
    RCCL all-reduce per bucket on the group's stream as it fills, no DistributedDataParallel), eagerly and captured as ONE
    hipGraph with the collectives inside.  On a one-rank group the average is the identity, so all three - plain trainer,
    eager reducer step, replayed graph - must agree BITWISE in losses, weights and EMA (same kernels, same order; the
    reducer only changes where gradients are written); and the large gradients must really have landed in the buckets, with
    buckets reduced while the backward pass was still being issued."""
    _run_ddp_graph_worker("captured_data_parallel")


@pytest.mark.timeout(1100)
def test_captured_accumulated_step_also_under_the_own_reducer():
    """accum_steps = 2 (BASELINE configs[2]'s shape in miniature): the two forward / backward passes and the optimizer step as
    ONE hipGraph, without and with train.GradReducer on a one-rank RCCL group (first micro-batch written into the buckets, the
    second added in place, the collectives after it) - all bitwise equal to the eager accumulated step of the plain trainer."""
    _run_ddp_graph_worker("captured_accumulated")


def test_captured_step_draws_fresh_drop_path_masks():
    """The drop-path masks come from the device generator: replays of one graph must not repeat the captured draw."""
    from octic_vits_amd.model import OcticVisionTransformer
    from octic_vits_amd.train import Trainer, synthetic_batch
    torch.manual_seed(0)
    m = OcticVisionTransformer(img_size=32, patch_size=4, in_chans=3, num_classes=10, embed_dim=128, depth=4,
                               num_heads=2, mlp_ratio=4.0, drop_path_rate=0.5, octic_equi_break_layer=2).cuda()
    t = Trainer(m, lr=0.0)                      # lr = 0: the weights stay put, the loss varies with the masks only
    x, y = synthetic_batch(8, 10, "cuda", seed=1, img_size=32)
    gs = t.capture(x, y, warmup=1)
    losses = {float(gs.replay(x, y)) for _ in range(6)}
    assert len(losses) > 1


def _small_hybrid(seed=5, drop_path=0.0):
    from octic_vits_amd.d8_layers import Layer_scale_init_BlockD8
    from octic_vits_amd.model import OcticVisionTransformer
    from octic_vits_amd.vit import Layer_scale_init_Block
    torch.manual_seed(seed)
    return OcticVisionTransformer(img_size=56, patch_size=14, num_classes=100, embed_dim=128, depth=4, num_heads=4,
                                  qkv_bias=True, drop_path_rate=drop_path, octic_block_layers=Layer_scale_init_BlockD8,
                                  standard_block_layers=Layer_scale_init_Block).cuda()


def test_accum_steps_equal_one_step_on_the_concatenated_batch():
    """Trainer(accum_steps=4): four micro-batches, gradients of the micro-batch mean losses averaged, ONE optimizer step
    (BASELINE configs[2]: global 2048 = 8 GPUs x 64 x 4; reference experiments/train_deit.py:7-12,66) must equal one
    step on the whole batch.  f32 arithmetic (autocast off) so that "mean of four means" == "mean over the batch" to
    rounding; three steps so the optimizer state carries over."""
    from octic_vits_amd.train import Trainer, synthetic_batch
    x, y = synthetic_batch(8, 100, "cuda", 21, img_size=56)
    ta = Trainer(_small_hybrid(), autocast=False, accum_steps=4)
    tb = Trainer(_small_hybrid(), autocast=False, accum_steps=1)
    for _ in range(3):
        la, lb = float(ta.step(x, y)), float(tb.step(x, y))
        assert abs(la - lb) <= 2e-6 * max(1.0, abs(lb)), (la, lb)
    assert ta.optimizer.step_count == 3 and ta.optimizer.skipped_steps == 0
    for (n, pa), pb in zip(ta.raw_model.named_parameters(), tb.raw_model.parameters()):
        assert torch.allclose(pa, pb, rtol=2e-4, atol=2e-6), f"{n}: {float((pa - pb).abs().max()):.3e}"
    # and under bf16 autocast it runs and stays close (micro-batch GEMMs round differently: loose bound)
    tc = Trainer(_small_hybrid(), accum_steps=2)
    td = Trainer(_small_hybrid(), accum_steps=1)
    lc, ld = float(tc.step(x, y)), float(td.step(x, y))
    assert abs(lc - ld) < 2e-2 * max(1.0, abs(ld))


def test_raw_data_writes_invalidate_the_transposed_weight_copy_too():
    """after a raw p.data write + invalidate_weight_caches the qkv INPUT gradient of a standard block (routed
    to the hand-written GEMM, which multiplies by a cached W^T) must be computed against the NEW weights."""
    from octic_vits_amd import functional as OF
    from octic_vits_amd.vit import Layer_scale_init_Block
    torch.manual_seed(2)
    blk = Layer_scale_init_Block(dim=256, num_heads=4, qkv_bias=True, init_values=0.5).cuda()
    x = torch.randn(4, 33, 256, device="cuda")

    def in_grad():
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = blk(xi)
        y.float().square().sum().backward()
        return xi.grad.clone()
    g0 = in_grad()                                   # fills the bf16 W and W^T caches
    with torch.no_grad():
        blk.attn.qkv.weight.data.mul_(-1.0)          # raw write: no version bump; q,k flip sign with v -> attention
        blk.attn.qkv.bias.data.mul_(-1.0)            # probabilities unchanged, the branch output and dX flip sign
    OF.invalidate_weight_caches(blk)
    g1 = in_grad()
    # reference: the same block with the flipped weights, caches built from scratch
    ref = Layer_scale_init_Block(dim=256, num_heads=4, qkv_bias=True, init_values=0.5).cuda()
    ref.load_state_dict(blk.state_dict())
    xi = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = ref(xi)
    y.float().square().sum().backward()
    assert torch.equal(g1, xi.grad), float((g1 - xi.grad).abs().max())
    assert not torch.allclose(g0, g1)


def test_fused_lamb_state_dict_round_trip():
    """FusedLamb.state_dict / load_state_dict (the reference saves optimizer.state_dict() every epoch,
    deit/main.py:414-423): a resumed trainer continues bit for bit."""
    from octic_vits_amd.train import Trainer, synthetic_batch
    x, y = synthetic_batch(8, 100, "cuda", 33, img_size=56)
    ta = Trainer(_small_hybrid(9), autocast=False)
    for _ in range(2):
        ta.step(x, y)
    sd_model = {k: v.clone() for k, v in ta.raw_model.state_dict().items()}
    sd_opt = ta.optimizer.state_dict()
    tb = Trainer(_small_hybrid(10), autocast=False)            # different init, then resume
    tb.raw_model.load_state_dict(sd_model)
    from octic_vits_amd import functional as OF
    OF.invalidate_weight_caches(tb.raw_model)
    tb.optimizer.load_state_dict(sd_opt)
    assert tb.optimizer.step_count == 2 and float(tb.optimizer.ws[3]) == 2.0
    for _ in range(2):
        la, lb = float(ta.step(x, y)), float(tb.step(x, y))
        assert la == lb
    for pa, pb in zip(ta.raw_model.parameters(), tb.raw_model.parameters()):
        assert torch.equal(pa, pb)
    for ea, eb in zip(ta.optimizer.ema_state(), tb.optimizer.ema_state()):
        assert torch.equal(ea, eb)


def test_segment_graphs_replay_like_the_eager_segmented_step_also_under_ddp_hooks_and_accumulation():
    """Trainer(segment_graphs=n) (round 4, review item 4): forward and backward as 2 n hipGraph replays
    (train.SegmentedModel over torch.cuda.make_graphed_callables), loss / gradient hooks / fused optimizer eager - the
    host-cheap step for N > 1 GPUs, where the whole-step graph does not apply.  (a) graphed slices == the same slices run
    eagerly, BITWISE, over several steps with drop-path on (the masks come from the device generator inside the captured
    slices); (b) the same under DistributedDataParallel (one-rank gloo group in this process: DDP's bucket hooks fire on the
    gradients the backward graphs return, FusedLamb reads the bucket views) and (c) with accum_steps = 2 (no_sync +
    accumulation into existing .grad)."""
    import os
    import torch.distributed as dist
    from octic_vits_amd.train import Trainer, synthetic_batch
    x, y = synthetic_batch(8, 100, "cuda", 33, img_size=56)
    batches = [synthetic_batch(8, 100, "cuda", 40 + i, img_size=56) for i in range(3)]

    def run(graphed, **kw):
        torch.manual_seed(1234)
        torch.cuda.manual_seed(1234)
        ddp = kw.pop("distributed", False)
        tr = Trainer(_small_hybrid(seed=9, drop_path=0.2), lr=1e-3, segment_graphs=3, distributed=ddp and graphed, **kw)
        if graphed:                                     # (a distributed trainer captures first: graphs, then the DDP wrapper)
            tr.capture_segments(x[: x.shape[0] // kw.get("accum_steps", 1)])     # static shape = one micro-batch
            assert tr.segmented.graphed
        elif ddp:
            tr._ddp_args = (0, None, False)
            tr._wrap_ddp()                              # eager slices under DDP
        torch.manual_seed(77)
        torch.cuda.manual_seed(77)                      # same device RNG stream for the drop-path masks from here on
        losses = [float(tr.step(bx, by)) for bx, by in [(x, y)] + batches + batches]
        return losses, [p.detach().clone() for p in tr.raw_model.parameters()]

    la, pa = run(False)
    lb, pb = run(True)
    assert la == lb, (la, lb)
    assert len(set(la)) == len(la)
    for a_, b_ in zip(pa, pb):
        assert torch.equal(a_, b_)
    # (c) accumulation
    lc, pc = run(False, accum_steps=2)
    ld, pd = run(True, accum_steps=2)
    assert lc == ld
    for a_, b_ in zip(pc, pd):
        assert torch.equal(a_, b_)
    # (b) one-rank process group: DDP wraps the segmented model
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("gloo", rank=0, world_size=1)
        made = True
    else:
        made = False
    try:
        le, pe = run(False, distributed=True)
        lf, pf = run(True, distributed=True)
        assert le == lf, (le, lf)
        for a_, b_ in zip(pe, pf):
            assert torch.equal(a_, b_)
        assert le == la                                 # and DDP at world size 1 changes nothing
    finally:
        if made:
            dist.destroy_process_group()


def test_compacted_stochastic_depth_equals_the_masked_full_batch():
    """d8_layers.COMPACT_DROP_PATH: every branch on the samples its per-sample Bernoulli mask keeps, against the reference
    formulation (every branch on every sample, dropped ones multiplied by zero) with the SAME masks: same logits and the same
    parameter gradients up to bf16 GEMM summation order (the kept rows see identical operands; weight gradients sum over fewer,
    differently tiled rows).  Also covers a branch that keeps nobody and one that keeps everybody."""
    import octic_vits_amd.d8_layers as L
    from octic_vits_amd.d8_layers import Layer_scale_init_BlockD8
    from octic_vits_amd.model import OcticVisionTransformer
    from octic_vits_amd.vit import Layer_scale_init_Block
    torch.manual_seed(0)
    net = OcticVisionTransformer(img_size=56, patch_size=14, num_classes=100, embed_dim=128, depth=4, num_heads=4,
                                 qkv_bias=True, drop_path_rate=0.5, octic_block_layers=Layer_scale_init_BlockD8,
                                 standard_block_layers=Layer_scale_init_Block).cuda().train()
    with torch.no_grad():                   # layer scale 1e-4 would hide the branches below the bf16 noise of the stream
        for n, p in net.named_parameters():
            if "gamma" in n:
                p.fill_(0.5)
    B = 12
    g = torch.Generator().manual_seed(5)
    masks = [(torch.rand(B, generator=g) < 0.5).float() for _ in range(8)]
    masks[2] = torch.zeros(B)               # nobody kept
    masks[5] = torch.ones(B)                # everybody kept
    x = torch.randn(B, 3, 56, 56, device="cuda")
    cot = torch.randn(B, 100, device="cuda")
    res = {}
    for compact in (False, True):
        it = iter(masks)
        L.drop_path_mask_source = lambda Bn, keep, device: next(it).to(device)
        L.compact_mask_source = lambda Bn, keep: next(it)
        L.COMPACT_DROP_PATH = compact
        try:
            net.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = net(x)
            (out.float() * cot).sum().backward()
        finally:
            L.drop_path_mask_source = L.compact_mask_source = None
            L.COMPACT_DROP_PATH = False
        assert next(it, None) is None       # both formulations drew all eight masks
        res[compact] = (out.detach().float(), {n: p.grad.detach().float().clone() for n, p in net.named_parameters()
                                              if p.grad is not None})
    out_f, g_f = res[False]
    out_c, g_c = res[True]
    scale = float(out_f.abs().max())
    assert float((out_f - out_c).abs().max()) <= 2e-2 * scale
    assert g_f.keys() == g_c.keys()
    worst = max((float((g_c[n] - g_f[n]).norm() / g_f[n].norm().clamp_min(1e-6)), n) for n in g_f)
    assert worst[0] <= 3e-2, worst


def test_graphed_inference_forward_equals_eager_and_refuses_stale_weights():
    """serve.GraphedForward: the eval forward of a hybrid model as one hipGraph replay - bitwise equal to the eager forward on
    fresh inputs, shape changes and parameter updates behind the graph are refused."""
    from octic_vits_amd.model import OcticVisionTransformer
    from octic_vits_amd.serve import GraphedForward
    torch.manual_seed(0)
    net = OcticVisionTransformer(img_size=32, patch_size=4, embed_dim=256, depth=4, num_heads=4, num_classes=10).cuda().eval()
    g = torch.Generator(device="cuda").manual_seed(1)
    x0 = torch.randn(8, 3, 32, 32, generator=g, device="cuda")
    gf = GraphedForward(net, x0)
    for seed in (2, 3):
        x = torch.randn(8, 3, 32, 32, generator=torch.Generator(device="cuda").manual_seed(seed), device="cuda")
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            want = net(x)
        got = gf(x)
        assert torch.equal(got, want)
    with pytest.raises(ValueError):
        gf(x[:4])
    with torch.no_grad():
        net.head.weight.add_(1.0)
    with pytest.raises(RuntimeError):
        gf(x)
