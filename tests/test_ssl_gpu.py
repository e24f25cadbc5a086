"""DINOv2 SSL student/teacher step on the HIP engine (octic_vits_amd/ssl.py) against the CPU oracle's restatement
(oracle/ssl_ref.py, whose pieces are pinned to the real reference by tests/golden/ssl_pieces.npz): same parameters, same
multi-crop batch (2 global + 2 local crops at a non-native resolution, iBOT masks), every loss term and a sample of the
student gradients.  f32 path 1e-3 (north-star forward tolerance), bf16 autocast 5e-2 on the losses."""
import random

import pytest
import torch

import cases
from oracle import octic_ref as R
from oracle import ssl_ref as SR

pytestmark = pytest.mark.gpu

KW = dict(head_n_prototypes=64, head_hidden_dim=48, head_bottleneck_dim=16, local_crops_number=2)
DIM = 128


def _oracle_backbone():
    m = R.OcticDinoVisionTransformer(img_size=32, patch_size=4, embed_dim=DIM, depth=4, num_heads=4,
                                     octic_block_layers=lambda **kw: R.NestedTensorBlockD8(init_values=0.1, **{k: v for k, v in kw.items() if k != "init_values"}),
                                     standard_block_layers=lambda **kw: R.NestedTensorBlock(init_values=0.1, **{k: v for k, v in kw.items() if k != "init_values"}))
    m.patch_embed.strict_img_size = False
    return m


def _product_backbone():
    from functools import partial
    from octic_vits_amd import d8_layers, dinov2_models, vit
    return dinov2_models.OcticDinoVisionTransformer(
        img_size=32, patch_size=4, embed_dim=DIM, depth=4, num_heads=4,
        octic_block_layers=lambda **kw: d8_layers.NestedTensorBlockD8(init_values=0.1, **{k: v for k, v in kw.items() if k != "init_values"}),
        standard_block_layers=lambda **kw: vit.NestedTensorBlock(attn_class=vit.MemEffAttention, init_values=0.1,
                                                                 **{k: v for k, v in kw.items() if k != "init_values"}))


def _batch():
    random.seed(11)
    gc, lc = cases.randn("ssl.step.g", 8, 3, 32, 32), cases.randn("ssl.step.l", 8, 3, 16, 16)
    return SR.collate(gc, lc, (0.1, 0.5), 0.5, 64, SR.MaskingGenerator((8, 8), max_num_patches=32))


def _to(images, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in images.items()}


def _pair():
    from octic_vits_amd import ssl as S
    torch.manual_seed(0)
    ref = SR.SSLMetaArch(_oracle_backbone, DIM, **KW)
    cases.fill_parameters(ref.student, salt="ssl.")
    for k in ref.student:
        ref.teacher[k].load_state_dict(ref.student[k].state_dict())
    mine = S.SSLMetaArch(_product_backbone, DIM, **KW)
    mine.student.load_state_dict(ref.student.state_dict(), strict=True)
    mine.teacher.load_state_dict(ref.teacher.state_dict(), strict=True)
    return ref.train(), mine.cuda().train()


def test_non_native_resolution_tokens_match_oracle():
    """Local crops: pos-embed resize branch (raises TypeError in the reference as shipped; oracle = evident intent)."""
    ref, mine = _pair()
    x = cases.randn("ssl.res.x", 2, 3, 16, 16)
    with torch.no_grad():
        a = mine.student.backbone.forward_features(x.cuda())
        b = ref.student.backbone.forward_features(x)
    for k in ("x_norm_clstoken", "x_norm_patchtokens"):
        scale = max(1.0, float(b[k].abs().max()))
        assert float((a[k].cpu() - b[k]).abs().max()) <= 1e-3 * scale, k


@pytest.mark.parametrize("centering", ["centering", "sinkhorn_knopp"])
def test_ssl_forward_backward_matches_oracle_f32(centering):
    ref, mine = _pair()
    ref.centering = mine.centering = centering
    images = _batch()
    want = ref.forward_backward(images, teacher_temp=0.05)
    got = mine.forward_backward(_to(images, "cuda"), teacher_temp=0.05)
    assert set(got) == set(want)
    for k in want:
        assert float(got[k]) == pytest.approx(float(want[k]), rel=1e-3, abs=1e-4), k
    # centres queued for the next step agree, too
    ref.dino_loss.apply_center_update(); mine.dino_loss.apply_center_update()
    assert torch.allclose(mine.dino_loss.center.cpu(), ref.dino_loss.center, atol=1e-5)
    n = 0
    mine_by_name = dict(mine.student.named_parameters())
    for name, p in ref.student.named_parameters():
        q = mine_by_name[name]
        if p.grad is None:
            assert q.grad is None or float(q.grad.abs().max()) == 0.0, name
            continue
        scale = max(1e-6, float(p.grad.abs().max()))
        err = float((q.grad.cpu() - p.grad).abs().max())
        if ".qkv." in name and name.endswith("bias"):
            continue        # K third: exactly zero gradient, rounding noise only
        assert err <= 2e-3 * scale + 1e-7, f"{name}: {err:.3e} vs scale {scale:.3e}"
        n += 1
    assert n > 60


def test_ssl_trainer_steps_bf16():
    """Two full iterations (bf16 autocast student/teacher forward, clip, AdamW, teacher EMA): losses finite and within
    5e-2 of the f32 oracle on the first step, the teacher moves towards the student, its bf16 weight caches follow."""
    from octic_vits_amd import ssl as S
    ref, mine = _pair()
    images = _batch()
    want = ref.forward_backward(images, teacher_temp=0.05, backward=False)
    tr = S.SSLTrainer(mine, lr=1e-3)
    t0 = [p.detach().clone() for p in mine.teacher.parameters()]
    out1 = tr.step(_to(images, "cuda"), teacher_temp=0.05, momentum=0.9)
    for k in want:
        assert float(out1[k]) == pytest.approx(float(want[k]), rel=5e-2, abs=5e-3), k
    out2 = tr.step(_to(images, "cuda"), teacher_temp=0.05, momentum=0.9)
    assert all(torch.isfinite(v).all() for v in out2.values())
    trainable = [p.requires_grad for p in mine.student.parameters()]     # frozen cls/mask token copies stay zeros
    moved = [not torch.equal(a, b) for a, b, tr_ in zip(t0, mine.teacher.parameters(), trainable) if tr_ and a.ndim >= 2]
    assert all(moved) and len(moved) > 40 and all(p.grad is None for p in mine.teacher.parameters())
    for (nt, t), (ns_, s_) in zip(mine.teacher.named_parameters(), mine.student.named_parameters()):
        assert nt == ns_ and t.shape == s_.shape


def test_fused_adamw_and_teacher_ema_match_torch_adamw_and_the_foreach_ema():
    """SSLTrainer(fused_optimizer=True): clip_grad_norm_ per sub-model + AdamW + the teacher's EMA as octic_adamw_step
    (csrc/lamb.hip without the trust ratio; the EMA is kept in the teacher's own tensors) against torch.optim.AdamW +
    clip_grad_norm_ + the foreach EMA on an identically initialised twin, three f32 steps: student, teacher and the losses."""
    from octic_vits_amd import ssl as S
    _, a = _pair()
    _, b = _pair()
    images = _to(_batch(), "cuda")
    ta = S.SSLTrainer(a, lr=2e-3, autocast=False, fused_optimizer=True, clip_grad=0.5)      # (a clip that bites)
    tb = S.SSLTrainer(b, lr=2e-3, autocast=False, fused_optimizer=False, clip_grad=0.5)
    for i in range(3):
        la = ta.step(images, teacher_temp=0.05, momentum=0.9)
        lb = tb.step(images, teacher_temp=0.05, momentum=0.9)
        for k in lb:
            assert float(la[k].detach()) == pytest.approx(float(lb[k].detach()), rel=2e-4, abs=1e-5), (i, k)
    moved = 0
    # the K third of a qkv bias has an exactly-zero gradient (softmax is invariant to a per-query shift): Adam normalises its
    # rounding noise to +-lr per step, so those tensors are held to 2 lr steps (the exception of the DeiT train fixture too)
    tol = lambda n: 2 * 2e-3 * 3 if n.endswith(("qkv.lin_A1.bias", "qkv.bias")) else 2e-6
    for (n, pa), (_, pb) in zip(a.student.named_parameters(), b.student.named_parameters()):
        assert torch.allclose(pa, pb, rtol=2e-4, atol=tol(n)), n
    for (n, pa), (_, pb), (_, ps) in zip(a.teacher.named_parameters(), b.teacher.named_parameters(), a.student.named_parameters()):
        assert torch.allclose(pa, pb, rtol=2e-4, atol=tol(n)), n
        moved += int(ps.requires_grad and not torch.equal(pa, ps))
    assert moved > 40                                            # the teacher trails the student, it is not a copy


def _hd64_backbone(drop_path=0.0):
    from octic_vits_amd import d8_layers, dinov2_models, vit
    return dinov2_models.OcticDinoVisionTransformer(
        img_size=32, patch_size=4, embed_dim=256, depth=4, num_heads=4, drop_path_rate=drop_path,
        octic_block_layers=lambda **kw: d8_layers.NestedTensorBlockD8(init_values=0.3, **{k: v for k, v in kw.items() if k != "init_values"}),
        standard_block_layers=lambda **kw: vit.NestedTensorBlock(attn_class=vit.MemEffAttention, init_values=0.3,
                                                                 **{k: v for k, v in kw.items() if k != "init_values"}))


def test_crop_sets_as_one_row_tensor_equal_the_set_by_set_loop():
    """ragged.py (round 5): a list of crop sets (global 32 x 32 = 65 tokens, local 16 x 16 = 17 tokens, head_dim 64) through the
    backbone as ONE token-row tensor - every row-wise kernel once on all rows, attention per set - against the reference's
    set-by-set loop (dinov2_models.RAGGED_LISTS = False), bf16 autocast: tokens of both sets and every parameter gradient,
    which each parameter now receives from ONE graph node instead of two.  Then training with drop_path 0.4: stochastic-depth
    masks per sample of every set (octic half) and the batch-subset stochastic depth per set (standard half) run and train."""
    from octic_vits_amd import dinov2_models as DM
    torch.manual_seed(0)
    net = _hd64_backbone().cuda().train()
    net.patch_embed.strict_img_size = False
    g = torch.Generator(device="cuda").manual_seed(1)
    xg = torch.randn(4, 3, 32, 32, generator=g, device="cuda")
    xl = torch.randn(8, 3, 16, 16, generator=g, device="cuda")
    masks = torch.rand(4, 64, generator=g, device="cuda") < 0.3
    cg = torch.randn(4, 65, 256, generator=g, device="cuda")
    cl = torch.randn(8, 17, 256, generator=g, device="cuda")
    params = [p for p in net.parameters() if p.requires_grad]

    def run(ragged):
        DM.RAGGED_LISTS = ragged
        for p in params:
            p.grad = None
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                og, ol = net([xg, xl], masks=[masks, None], is_training=True)
        finally:
            DM.RAGGED_LISTS = True
        tg = torch.cat([og["x_norm_clstoken"].unsqueeze(1), og["x_norm_patchtokens"]], 1)
        tl = torch.cat([ol["x_norm_clstoken"].unsqueeze(1), ol["x_norm_patchtokens"]], 1)
        ((tg.float() * cg).sum() + (tl.float() * cl).sum()).backward()
        return tg.detach().float(), tl.detach().float(), [p.grad.detach().float().clone() for p in params]

    g0, l0, gr0 = run(False)
    g1, l1, gr1 = run(True)
    for a_, b_ in ((g0, g1), (l0, l1)):
        assert float((a_ - b_).abs().max()) <= 3e-2 * float(a_.abs().max())
    for a_, b_, p in zip(gr0, gr1, params):
        assert float((a_ - b_).norm()) <= 4e-2 * float(a_.norm()) + 1e-6, tuple(p.shape)

    torch.manual_seed(0)
    net2 = _hd64_backbone(drop_path=0.4).cuda().train()
    net2.patch_embed.strict_img_size = False
    opt = torch.optim.SGD(net2.parameters(), lr=1e-3)
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            og, ol = net2([xg, xl], masks=[masks, None], is_training=True)
        loss = og["x_norm_patchtokens"].float().square().mean() + ol["x_norm_clstoken"].float().square().mean()
        loss.backward()
        assert torch.isfinite(loss) and all(p.grad is None or torch.isfinite(p.grad).all() for p in net2.parameters())
        opt.step()


def test_stochastic_depth_of_the_one_row_tensor_pass_matches_the_oracle_draw_for_draw():
    """The student pass with drop_path 0.4 against the CPU oracle WITH THE SAME RANDOM DRAWS: the oracle (set-by-set loop,
    dinov2_models.py:139-160; per-sample masks of the octic blocks, d8_layers.py:249-271; batch subsets of the standard
    blocks, dinov2/layers/block.py:113-140) runs first and its draws are recorded in draw order (block -> crop set -> branch);
    the engine's pass - both crop sets as one row tensor, masks and subsets drawn once per pass, the subset's rows through row
    maps - replays them through ragged.MASK_SOURCE / PERM_SOURCE.  Outputs within 3e-2 of scale, every parameter gradient
    within max(5e-2, 2.5 x the oracle's own distance under CPU bf16 autocast with the same draws)."""
    import numpy as np
    from octic_vits_amd import ragged as RG
    kw_strip = lambda kw: {k: v for k, v in kw.items() if k != "init_values"}
    ref = R.OcticDinoVisionTransformer(
        img_size=32, patch_size=4, embed_dim=256, depth=4, num_heads=4, drop_path_rate=0.4,
        octic_block_layers=lambda **kw: R.NestedTensorBlockD8(init_values=0.3, **kw_strip(kw)),
        standard_block_layers=lambda **kw: R.NestedTensorBlock(init_values=0.3, **kw_strip(kw)))
    ref.patch_embed.strict_img_size = False
    cases.fill_parameters(ref, salt="ragdp.")
    net = _hd64_backbone(drop_path=0.4)
    net.patch_embed.strict_img_size = False
    net.load_state_dict(ref.state_dict(), strict=True)
    net = net.cuda().train()
    ref.train()
    xg, xl = cases.randn("ragdp.g", 4, 3, 32, 32), cases.randn("ragdp.l", 8, 3, 16, 16)
    cg, cl = cases.randn("ragdp.cg", 4, 65, 256), cases.randn("ragdp.cl", 8, 17, 256)
    names = [n for n, p in ref.named_parameters() if p.requires_grad]

    def tokens(o):
        return torch.cat([o["x_norm_clstoken"].unsqueeze(1), o["x_norm_patchtokens"]], 1)

    rec_masks, rec_subsets = [], []

    def mask_rec(B, keep):
        m = torch.empty((B, 1, 1)).bernoulli_(keep).flatten()       # the reference's own call (d8_layers.py:256-262)
        rec_masks.append(m.clone())
        return m

    def oracle(autocast):
        rec_masks.clear()
        rec_subsets.clear()
        for p in ref.parameters():
            p.grad = None
        torch.manual_seed(77)
        R.drop_path_mask_source, R.subset_observer = mask_rec, lambda b, br: rec_subsets.append((b, br.clone()))
        try:
            if autocast:
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    og, ol = ref([xg, xl], masks=[None, None], is_training=True)
            else:
                og, ol = ref([xg, xl], masks=[None, None], is_training=True)
        finally:
            R.drop_path_mask_source, R.subset_observer = None, None
        tg, tl = tokens(og).float(), tokens(ol).float()
        ((tg * cg).sum() + (tl * cl).sum()).backward()
        rp = dict(ref.named_parameters())
        return tg.detach(), tl.detach(), {n: rp[n].grad.detach().double().numpy().copy() for n in names if rp[n].grad is not None}

    tg0, tl0, g_ref = oracle(False)
    masks0, subsets0 = [m.clone() for m in rec_masks], [(b, s.clone()) for b, s in rec_subsets]
    _, _, g_yard = oracle(True)
    assert all(torch.equal(a, b) for a, b in zip(masks0, rec_masks)) and all(torch.equal(a[1], b[1]) for a, b in zip(subsets0, rec_subsets))
    assert len(masks0) == 2 * 2 * 2 and len(subsets0) == 2 * 2 * 2        # 2 blocks x 2 crop sets x 2 branches, each half

    def mask_source(live, rag):          # live = [blk0.dp1, blk0.dp2, blk1.dp1, blk1.dp2]; oracle order: block, set, branch
        assert len(live) == 4 and rag.samples == 12
        return torch.stack([torch.cat([masks0[(i // 2) * 4 + s * 2 + (i % 2)] for s in range(2)]) for i in range(4)])

    def perm_source(n, rag):             # branch i = 2 * block + branch of the standard half; per set an [n, B] permutation
        assert n == 4
        out = []
        for s, (B, _, _) in enumerate(rag.sets):
            rows = []
            for i in range(n):
                b, br = subsets0[(i // 2) * 4 + s * 2 + (i % 2)]
                assert b == B
                rest = torch.tensor([j for j in range(B) if j not in set(br.tolist())], dtype=torch.int64)
                rows.append(torch.cat([br.to(torch.int64), rest]))
            out.append(torch.stack(rows))
        return out

    RG.MASK_SOURCE, RG.PERM_SOURCE = mask_source, perm_source
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            og, ol = net([xg.cuda(), xl.cuda()], masks=[None, None], is_training=True)
        tg, tl = tokens(og).float(), tokens(ol).float()
        ((tg * cg.cuda()).sum() + (tl * cl.cuda()).sum()).backward()
    finally:
        RG.MASK_SOURCE, RG.PERM_SOURCE = None, None
    assert net._single_use_pass
    for got, want in ((tg, tg0), (tl, tl0)):
        assert float((got.detach().cpu() - want).abs().max()) <= 3e-2 * float(want.abs().max())
    mp = dict(net.named_parameters())
    bad = []
    for n, w in g_ref.items():
        den = max(float(np.linalg.norm(w)), 1e-6)
        rel = float(np.linalg.norm(mp[n].grad.detach().double().cpu().numpy() - w)) / den
        rel_oracle = float(np.linalg.norm(g_yard[n] - w)) / den
        if rel > max(5e-2, 2.5 * rel_oracle):
            bad.append((n, round(rel, 4), round(rel_oracle, 4)))
    assert not bad, bad[:8]


def test_ssl_trainer_batched_finishes_equal_immediate_ones():
    """SSLTrainer on a backbone whose crop sets run as one row tensor (every block module used once per pass): the
    parameter-gradient reductions batched at the end of backward + the paired qkv / proj weight gradients (train.BATCHED_FINISHES)
    against immediate finishes - same seeds, two steps: losses equal, student parameters equal to f32 summation-order noise."""
    from octic_vits_amd import ssl as S, train as TR

    def run(batched):
        TR.BATCHED_FINISHES = batched
        try:
            torch.manual_seed(0)
            arch = S.SSLMetaArch(lambda: _hd64_backbone(drop_path=0.4), 256, head_n_prototypes=512, head_hidden_dim=128,
                                 head_bottleneck_dim=64, local_crops_number=4).cuda()
            for m in (arch.student["backbone"], arch.teacher["backbone"]):
                m.patch_embed.strict_img_size = False
            tr = S.SSLTrainer(arch, lr=1e-3)
            images = S.synthetic_multicrop_batch(4, "cuda", seed=3, global_size=32, local_size=16, n_local=4, patch_size=4)
            losses = []
            for i in range(2):
                torch.manual_seed(10 + i)
                out = tr.step(images, teacher_temp=0.05, momentum=0.99)
                losses.append(float(out["total"]))
            assert arch.student["backbone"]._single_use_pass
            return losses, [p.detach().clone() for p in arch.student.parameters()]
        finally:
            TR.BATCHED_FINISHES = True

    (la, pa), (lb, pb) = run(True), run(False)
    assert la[0] == lb[0] and abs(la[1] - lb[1]) <= 1e-4 * abs(lb[1])
    for a, b in zip(pa, pb):
        assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max())) + 2e-6


def test_row_maps_equal_gather_and_scatter_bitwise():
    """The batch-subset stochastic depth of the ragged pass with the kept rows read / written through row maps inside the
    LayerNorm and residual-tail kernels (vit.ROW_MAPS) against explicit gather / scatter passes: same draws (same seed), same
    arithmetic per element -> outputs and every gradient bit for bit."""
    from octic_vits_amd import vit
    torch.manual_seed(0)
    net = _hd64_backbone(drop_path=0.4).cuda().train()
    net.patch_embed.strict_img_size = False
    g = torch.Generator(device="cuda").manual_seed(1)
    xg = torch.randn(4, 3, 32, 32, generator=g, device="cuda")
    xl = torch.randn(8, 3, 16, 16, generator=g, device="cuda")
    params = [p for p in net.parameters() if p.requires_grad]

    def run(maps):
        vit.ROW_MAPS = maps
        for p in params:
            p.grad = None
        torch.manual_seed(7)
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                og, ol = net([xg, xl], masks=[None, None], is_training=True)
        finally:
            vit.ROW_MAPS = True
        (og["x_norm_patchtokens"].float().square().mean() + ol["x_norm_clstoken"].float().square().mean()).backward()
        return (og["x_prenorm"].detach().clone(), ol["x_prenorm"].detach().clone(),
                [None if p.grad is None else p.grad.detach().clone() for p in params])

    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert sum(g is not None for g in a[2]) > 100
    for ga, gb, p in zip(a[2], b[2], params):
        assert (ga is None) == (gb is None) and (ga is None or torch.equal(ga, gb)), tuple(p.shape)


def test_row_map_kernels_against_index_arithmetic():
    """octic_*_rows through the C ABI: LayerNorm of mapped rows (+ compact copy), tail written to mapped rows, tail backward
    reading mapped rows, LayerNorm backward editing mapped rows in place - against the plain kernels on gathered rows."""
    from octic_vits_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    R, d, k = 300, 1280, 117
    x = torch.randn(1, R, d, generator=g, device="cuda")
    w, b = torch.randn(d, generator=g, device="cuda"), torch.randn(d, generator=g, device="cuda")
    rowmap = torch.randperm(R, generator=g, device="cuda")[:k].to(torch.int32)
    idx = rowmap.long()
    y, stats, xa = ops.dense_layernorm_fwd_rows(x, rowmap, w, b, 1e-6, torch.bfloat16)
    y0, stats0 = ops.dense_layernorm_fwd(x[0, idx].contiguous(), w, b, 1e-6, torch.bfloat16)
    assert torch.equal(y[0], y0) and torch.equal(stats, stats0) and torch.equal(xa[0], x[0, idx])
    f = torch.randn(k, d, generator=g, device="cuda").to(torch.bfloat16)
    gamma, rs = torch.randn(d, generator=g, device="cuda"), torch.rand(k, generator=g, device="cuda")
    stream = x.clone()
    ops.scale_residual_fwd_rows_(stream, rowmap, xa.view(k, d), f, gamma, rs, 1)
    ref = x.clone()
    ref[0, idx] = ops.scale_residual_fwd(xa.view(k, d), f, gamma, rs, 1)
    assert torch.equal(stream, ref)
    gs = torch.randn(1, R, d, generator=g, device="cuda")
    gy, dgamma, colsum = ops.scale_residual_bwd(gs, f, gamma, rs, 1, rowmap=rowmap)
    gy0, dgamma0, colsum0 = ops.scale_residual_bwd(gs[0, idx].contiguous(), f, gamma, rs, 1)
    assert torch.equal(gy, gy0) and torch.equal(dgamma, dgamma0) and torch.equal(colsum, colsum0)
    gln = torch.randn(k, d, generator=g, device="cuda").to(torch.bfloat16)
    g_edit = gs.clone()
    dw, db = ops.dense_layernorm_bwd_rows_(gln, xa, w, stats, g_edit, rowmap)
    dx0, dw0, db0 = ops.dense_layernorm_bwd(gln, xa.view(k, d), w, stats, gs[0, idx].contiguous())
    ref = gs.clone()
    ref[0, idx] = dx0
    assert torch.equal(g_edit, ref) and torch.equal(dw, dw0) and torch.equal(db, db0)
    with pytest.raises(ValueError):
        ops.dense_layernorm_fwd_rows(x, rowmap.long(), w, b, 1e-6, torch.bfloat16)


def test_nested_block_subset_stochastic_depth_on_the_engine_equals_the_eager_composition():
    """vit.NestedTensorBlock in training with drop_path > 0.1 (dinov2/layers/block.py:113-140: the branch on a random batch
    subset, added back scaled by b / keep): the engine path (vit.SUBSET_FUSED: gathered rows through the fused LayerNorm /
    GEMM / attention / residual-tail kernels) against the eager composition with the SAME subsets (one randperm per branch,
    same seed) - output and every gradient within the bf16 tolerances; the caller's tensor is not modified."""
    from octic_vits_amd import vit
    torch.manual_seed(0)
    blk = vit.NestedTensorBlock(dim=256, num_heads=4, qkv_bias=True, init_values=0.5, drop_path=0.4).cuda().train()
    x0 = torch.randn(10, 37, 256, device="cuda")
    cot = torch.randn(10, 37, 256, device="cuda")
    params = list(blk.parameters())

    def run(fused):
        vit.SUBSET_FUSED = fused
        for p in params:
            p.grad = None
        x = x0.clone().requires_grad_(True)
        keep = x.detach().clone()
        torch.manual_seed(123)
        torch.cuda.manual_seed(123)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = blk(x)
        (out.float() * cot).sum().backward()
        assert torch.equal(x.detach(), keep)
        return out.detach().float(), [x.grad.float()] + [p.grad.detach().float().clone() for p in params]

    try:
        oe, ge = run(False)
        of, gf = run(True)
    finally:
        vit.SUBSET_FUSED = True
    changed = ((oe - x0).flatten(1).abs().amax(1) > 0)
    assert 0 < int(changed.sum()) <= 10                      # a strict subset may be kept by both branches' draws
    assert torch.equal(changed, (of - x0).flatten(1).abs().amax(1) > 0)      # the same samples
    assert float((oe - of).abs().max()) <= 3e-2 * float(oe.abs().max())
    for a_, b_ in zip(ge, gf):
        assert float((a_ - b_).norm()) <= 3e-2 * float(a_.norm()) + 1e-6


@pytest.mark.timeout(1800)
def test_ssl_step_hybrid_vit_huge_multicrop():
    """BASELINE configs[4] geometry at the reference's largest octic DINOv2 model (hybrid ViT-H/16; the reference has no
    ViT-g): 2 x 224^2 + 8 x 96^2 crops, 4 images, 65536 prototypes, one bf16-autocast forward_backward against the
    CPU oracle (oracle/ssl_ref.py + oracle/octic_ref.py) with the same parameters and crops: every loss term within 5e-2 and a
    strided sample of the student gradients (backbone of both halves + head) within max(3e-2, 2 x the oracle's own distance
    under CPU bf16 autocast) - the recipe of test_vith_gpu.py.  (Round 4 checked this step for finiteness only.)"""
    import numpy as np
    from octic_vits_amd import ssl as S
    from octic_vits_amd.dinov2_models import hybrid_dinov2_vit_huge_patch16
    torch.manual_seed(0)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    def oracle_backbone():
        m = R.hybrid_dinov2_vit_huge_patch16(img_size=224, drop_path_rate=0.0)
        m.patch_embed.strict_img_size = False            # (the 96 x 96 local crops: see ssl.SSLMetaArch)
        return m

    ref = SR.SSLMetaArch(oracle_backbone, 1280)
    cases.fill_parameters(ref.student, salt="sslh.")
    for k in ref.student:
        ref.teacher[k].load_state_dict(ref.student[k].state_dict())
    arch = S.SSLMetaArch(lambda: hybrid_dinov2_vit_huge_patch16(img_size=224, drop_path_rate=0.0), 1280)
    arch.student.load_state_dict(ref.student.state_dict(), strict=True)
    arch.teacher.load_state_dict(ref.teacher.state_dict(), strict=True)
    arch = arch.cuda().train()
    ref.train()
    images = S.synthetic_multicrop_batch(4, "cpu", seed=5)
    names = [n for n, p in ref.student.named_parameters() if p.requires_grad]
    sample = sorted(set(names[::23] + [n for n in names if n.startswith("dino_head.")]))
    rp = dict(ref.student.named_parameters())

    def oracle(autocast):
        for p in ref.student.parameters():
            p.grad = None
        if autocast:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                ld = ref.forward_backward(images, teacher_temp=0.04)
        else:
            ld = ref.forward_backward(images, teacher_temp=0.04)
        return ({k: float(v.detach()) for k, v in ld.items()},
                {n: rp[n].grad.detach().double().numpy().copy() for n in sample if rp[n].grad is not None})

    want, g_ref = oracle(False)
    _, g_yard = oracle(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = arch.forward_backward(_to(images, "cuda"), teacher_temp=0.04)
    assert all(torch.isfinite(v).all() for v in out.values()), out
    for k in want:
        assert float(out[k]) == pytest.approx(want[k], rel=5e-2, abs=5e-3), k
    gp = dict(arch.student.named_parameters())
    bad = []
    for n, w in g_ref.items():
        den = max(float(np.linalg.norm(w)), 1e-12)
        rel = float(np.linalg.norm(gp[n].grad.detach().float().cpu().double().numpy() - w)) / den
        lim = max(3e-2, 2.0 * float(np.linalg.norm(g_yard[n] - w)) / den)
        if rel > lim:
            bad.append((rel / lim, n, rel))
    bad.sort(reverse=True)
    assert len(g_ref) > 30
    assert not bad, "; ".join(f"{n}: {r:.4f}" for _, n, r in bad[:8])


def test_ssl_trainer_with_the_own_reducer_on_a_one_rank_rccl_group_equals_the_plain_trainer():
    """Round 6: SSLTrainer(distributed=True) averages every student gradient with train.GradReducer (no DistributedDataParallel
    wrapper): on a one-rank RCCL group, on a backbone whose crop sets run as one row tensor (every module used once: the buckets
    go out during the pass, batched finishes and paired weight gradients stay on), losses and student / teacher parameters must
    equal the plain trainer's BITWISE - the reducer only changes where gradients are written."""
    import os
    import socket

    import torch.distributed as dist

    from octic_vits_amd import ssl as S, train as TR

    def run(distributed):
        torch.manual_seed(0)
        arch = S.SSLMetaArch(lambda: _hd64_backbone(drop_path=0.4), 256, head_n_prototypes=512, head_hidden_dim=128,
                             head_bottleneck_dim=64, local_crops_number=4).cuda()
        for m in (arch.student["backbone"], arch.teacher["backbone"]):
            m.patch_embed.strict_img_size = False
        tr = S.SSLTrainer(arch, lr=1e-3, distributed=distributed, local_rank=0)
        assert (tr._reducer is not None) == distributed and tr._ddp is None
        images = S.synthetic_multicrop_batch(4, "cuda", seed=3, global_size=32, local_size=16, n_local=4, patch_size=4)
        losses = []
        for i in range(3):
            torch.manual_seed(10 + i)
            out = tr.step(images, teacher_temp=0.05, momentum=0.99)
            losses.append(float(out["total"]))
        return (losses, [p.detach().clone() for p in arch.student.parameters()],
                [p.detach().clone() for p in arch.teacher.parameters()], tr)

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    old = TR.DDP_FLAT_SMALL_NUMEL
    TR.DDP_FLAT_SMALL_NUMEL = 20_000
    try:
        lb, sb, tb, trb = run(True)
        assert len(trb._reducer.buckets) >= 1
    finally:
        TR.DDP_FLAT_SMALL_NUMEL = old
        dist.destroy_process_group()
    la, sa, ta, _ = run(False)
    assert la == lb, (la, lb)
    for a, b in zip(sa + ta, sb + tb):
        assert torch.equal(a, b)
