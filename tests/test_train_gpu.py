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
