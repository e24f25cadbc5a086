"""csrc/ssl_loss.hip through the C ABI: the prototype-axis row kernels of the DINOv2 objective against the reference's eager
composition evaluated in float64 (dinov2/loss/dino_clstoken_loss.py:43-51,78-92; ibot_patch_loss.py:26-34,63-77).
Tolerance: logits / temperature reach magnitudes of a few hundred, where ONE f32 rounding of the argument already moves exp() by
1e-5 relative - so the bar is the f32 eager composition itself: the kernels' error against float64 may not exceed four times the
error of the same formula evaluated by torch in f32 (+ 2e-6 of the scale; the kernels multiply by 1 / temp where torch divides).  bf16 logits are read exactly (the arithmetic is f32); a bf16
gradient is rounded once (2^-8 relative)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _gen(seed):
    return torch.Generator(device="cuda").manual_seed(seed)


@pytest.mark.parametrize("rows,K", [(1, 8), (5, 520), (37, 4096), (64, 65536)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("centered", [True, False])
def test_softmax_center_matches_float64(rows, K, dtype, centered):
    from octic_vits_amd import ops
    g = _gen(rows + K)
    t = (torch.randn(rows, K, generator=g, device="cuda") * 3).to(dtype)
    c = torch.randn(1, K, generator=g, device="cuda") if centered else None
    out = ops.softmax_center(t, c, 1.0 / 0.04)
    ref = F.softmax((t.double() - (c.double() if centered else 0.0)) / 0.04, dim=-1)
    eager = F.softmax((t.float() - (c if centered else 0.0)) / 0.04, dim=-1)
    assert out.dtype == torch.float32 and out.shape == t.shape
    bar = 4 * float((eager.double() - ref).abs().max()) + 2e-6 * float(ref.max())
    assert float((out.double() - ref).abs().max()) <= bar
    assert float((out.double().sum(-1) - 1).abs().max()) <= 1e-5


def test_softmax_center_reads_strided_rows_and_3d():
    from octic_vits_amd import ops
    g = _gen(3)
    big = torch.randn(6, 2 * 1024, generator=g, device="cuda")
    t = big[:, :1024]                                  # row stride 2048
    c = torch.randn(1, 1, 1024, generator=g, device="cuda")
    out = ops.softmax_center(t.unsqueeze(0), c, 10.0)
    ref = F.softmax((t.double().unsqueeze(0) - c.double()) * 10.0, dim=-1)
    assert out.shape == (1, 6, 1024)
    assert float((out.double() - ref).abs().max()) <= 2e-5 * float(ref.max())


@pytest.mark.parametrize("rows,t_rows,K", [(1, 1, 8), (6, 3, 520), (24, 8, 4096), (40, 40, 65536)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_soft_cross_entropy_rows_and_gradient(rows, t_rows, K, dtype):
    """loss[r] = -sum_k t[r % Nt, k] log_softmax(s_r / temp)_k and d loss / d s under a random per-row cotangent."""
    from octic_vits_amd import functional as OF
    g = _gen(rows * 7 + K)
    s = (torch.randn(rows, K, generator=g, device="cuda") * 2).to(dtype).requires_grad_(True)
    t = F.softmax(torch.randn(t_rows, K, generator=g, device="cuda") * 4, dim=-1)
    w = torch.randn(rows, generator=g, device="cuda")
    loss = OF.SoftCrossEntropyFn.apply(s, t, 0.1)
    (loss * w).sum().backward()
    s64 = s.detach().double().requires_grad_(True)
    tt = t.double().repeat(rows // t_rows, 1)
    ref = -(tt * F.log_softmax(s64 / 0.1, dim=-1)).sum(-1)
    (ref * w.double()).sum().backward()
    s32 = s.detach().float().requires_grad_(True)
    eager = -(t.repeat(rows // t_rows, 1) * F.log_softmax(s32 / 0.1, dim=-1)).sum(-1)
    (eager * w).sum().backward()
    assert loss.dtype == torch.float32
    assert float((loss.double() - ref).abs().max()) <= 4 * float((eager.double() - ref).abs().max()) + 2e-6 * float(ref.abs().max())
    assert s.grad.dtype == dtype
    gscale = float(s64.grad.abs().max())
    bar = 4 * float((s32.grad.double() - s64.grad).abs().max()) + (2e-6 if dtype == torch.float32 else 2.0 ** -8) * gscale
    assert float((s.grad.double() - s64.grad).abs().max()) <= bar


def test_unnormalised_targets_use_their_row_sums():
    """Sinkhorn-Knopp targets are not renormalised per row by the loss: the gradient carries sum_k t_k."""
    from octic_vits_amd import functional as OF
    g = _gen(11)
    s = torch.randn(9, 256, generator=g, device="cuda", requires_grad=True)
    t = torch.rand(9, 256, generator=g, device="cuda") * 0.3
    OF.SoftCrossEntropyFn.apply(s, t, 0.5).sum().backward()
    s64 = s.detach().double().requires_grad_(True)
    (-(t.double() * F.log_softmax(s64 / 0.5, dim=-1)).sum(-1)).sum().backward()
    assert float((s.grad.double() - s64.grad).abs().max()) <= 1e-5 * float(s64.grad.abs().max())


def test_bad_shapes_are_refused():
    from octic_vits_amd import ops
    with pytest.raises(RuntimeError):
        ops.softmax_center(torch.zeros(2, 12, device="cuda"), None, 1.0)          # K % 8
    with pytest.raises(ValueError):
        ops.soft_ce_fwd(torch.zeros(2, 16, device="cuda"), torch.zeros(2, 8, device="cuda"), 1.0)


@pytest.mark.parametrize("centering", ["centering", "sinkhorn_knopp"])
def test_dino_and_ibot_terms_equal_the_eager_composition(centering):
    """ssl.DINOLoss / ssl.iBOTPatchLoss with the row kernels against the same modules with FUSED_LOSS_ROWS = False, under
    bf16 autocast (the student logits are bf16): losses and student-logit gradients."""
    from octic_vits_amd import ssl as S
    K, B, n_local, nm = 4096, 6, 4, 37
    g = _gen(5)
    t_cls = (torch.randn(2 * B, K, generator=g, device="cuda") * 0.5).to(torch.bfloat16)       # exp(t / 0.05) stays finite
    t_patch = (torch.randn(nm, K, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
    masks = torch.zeros(2 * B, 16, dtype=torch.bool, device="cuda")
    masks.view(-1)[torch.randperm(masks.numel(), generator=g, device="cuda")[:nm]] = True
    mw = (1 / masks.sum(-1).clamp(min=1.0)).unsqueeze(-1).expand_as(masks)[masks]

    def run(fused):
        S.FUSED_LOSS_ROWS = fused
        try:
            dl, il = S.DINOLoss(K).cuda(), S.iBOTPatchLoss(K).cuda()
            dl.center.copy_(torch.linspace(-1, 1, K, device="cuda").view(1, K))
            il.center.copy_(torch.linspace(1, -1, K, device="cuda").view(1, 1, K))
            gg = _gen(6)
            s_loc = torch.randn(n_local * B, K, generator=gg, device="cuda").to(torch.bfloat16).requires_grad_(True)
            s_glob = torch.randn(2 * B, K, generator=gg, device="cuda").to(torch.bfloat16).requires_grad_(True)
            s_patch = torch.randn(nm, K, generator=gg, device="cuda").to(torch.bfloat16).requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                with torch.no_grad():
                    if centering == "centering":
                        td = dl.softmax_center_teacher(t_cls, 0.05).view(2, B, K)
                        tp = il.softmax_center_teacher(t_patch.unsqueeze(0), 0.05).squeeze(0)
                    else:
                        td = dl.sinkhorn_knopp_teacher(t_cls, 0.05).view(2, B, K)
                        tp = il.sinkhorn_knopp_teacher(t_patch, 0.05, torch.tensor([nm], device="cuda"))
                l1 = dl(s_loc.chunk(n_local), td)
                l2 = dl([s_glob], [td.flatten(0, 1)])
                l3 = il.forward_masked(s_patch, tp, student_masks_flat=masks, n_masked_patches=nm, masks_weight=mw)
            (l1 + l2 + l3).backward()
            return [l1.detach().float(), l2.detach().float(), l3.detach().float()], [s_loc.grad.float(), s_glob.grad.float(), s_patch.grad.float()]
        finally:
            S.FUSED_LOSS_ROWS = True

    (la, ga), (lb, gb) = run(True), run(False)
    for a, b in zip(la, lb):
        assert abs(float(a) - float(b)) <= 4e-3 * abs(float(b)), (float(a), float(b))     # eager rounds s / temp to bf16 first
    for a, b in zip(ga, gb):
        assert float((a - b).norm()) <= 2e-2 * float(b.norm())
