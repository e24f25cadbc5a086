"""world_size-2 `gloo` test (CPU) of the DINOv2 step's data-parallel variant ("FSDP-free DP variant").

``ssl.SSLTrainer(distributed=True)`` is the only code with collectives of its own: the DDP wrapper around the student
backbone, the asynchronous all-reduce of the DINO / iBOT centre statistics (reference dinov2/loss/dino_clstoken_loss.py:75-99,
ibot_patch_loss.py:131-151), the cross-rank sums inside Sinkhorn-Knopp (dino_clstoken_loss.py:43-63, ibot_patch_loss.py:64-95)
and the manual all-reduce of the head gradients (the heads run outside the wrapper).  The test: two ranks on half of the
images each == one process on all of them - losses, centres after ``apply_center_update``, Sinkhorn targets, every student
gradient (backbone through DDP, heads through the manual all-reduce), and the teacher after its EMA update - for both
centerings, over three steps (the centre of step k enters the loss of step k + 1).

The backbone is a small torch stand-in with the product backbone's call surface (the HIP engine refuses CPU tensors by
design; what is under test here is the collective plumbing around a backbone, not the backbone).  KoLeo is switched off in
the equality runs: its nearest neighbour is searched inside the local batch in the reference too (koleo_loss.py:27-48), so a
sharded batch is a different loss by construction."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

N_IMG, N_LOCAL, DIM, PROTO = 8, 2, 32, 48


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _TinyBackbone(nn.Module):
    """patch embedding + mask token + one token-mixing residual layer + norm; the call surface of
    dinov2_models.OcticDinoVisionTransformer.forward (list of crop sets + list of masks -> list of dicts)."""

    def __init__(self, dim=DIM, patch=8):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, patch, patch)
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim) * 0.02)
        self.mask_token = nn.Parameter(torch.randn(1, dim) * 0.02)
        self.mix = nn.Linear(dim, dim)
        self.fc = nn.Linear(dim, dim)
        self.norm = nn.LayerNorm(dim)

    def _one(self, x, masks):
        t = self.proj(x).flatten(2).transpose(1, 2)
        if masks is not None:
            t = torch.where(masks.unsqueeze(-1), self.mask_token.to(t.dtype).unsqueeze(0), t)
        t = torch.cat([self.cls_token.expand(t.shape[0], -1, -1), t], dim=1)
        t = t + torch.tanh(self.mix(t.mean(1, keepdim=True)))
        t = self.norm(t + self.fc(t))
        return {"x_norm_clstoken": t[:, 0], "x_norm_patchtokens": t[:, 1:], "masks": masks}

    def forward(self, x, masks=None, is_training=False):
        if isinstance(x, list):
            return [self._one(xi, mi) for xi, mi in zip(x, masks)]
        return self._one(x, masks)


def _full_batch(seed=11):
    """All images: global crops crop-major [A of every image | B of every image], local crops likewise, iBOT masks on about
    half of the global crops."""
    g = torch.Generator().manual_seed(seed)
    gc = torch.randn(2 * N_IMG, 3, 32, 32, generator=g)
    lc = torch.randn(N_LOCAL * N_IMG, 3, 16, 16, generator=g)
    masks = torch.rand(2 * N_IMG, 16, generator=g) < 0.3
    masks[torch.arange(2 * N_IMG) % 3 == 0] = False          # some crops carry no mask at all
    # Both halves of the images mask the same NUMBER of patches (other positions): the iBOT centre statistic is the mean over
    # a rank's masked patches, averaged over ranks (ibot_patch_loss.py:133-147 - a mean of per-rank means, in the reference
    # too), which equals the mean over all masked patches only when the ranks hold equally many.
    h = N_IMG // 2
    for base in (0, N_IMG):
        masks[base + h:base + 2 * h] = torch.roll(masks[base:base + h], shifts=5, dims=1)
    return gc, lc, masks


def _images_of(rows):
    """The collated dict (ssl.collate's keys) of the images in `rows` (indices into range(N_IMG))."""
    gc, lc, masks = _full_batch()
    rows = torch.as_tensor(rows)
    gi = torch.cat([rows, N_IMG + rows])
    li = torch.cat([j * N_IMG + rows for j in range(N_LOCAL)])
    m = masks[gi]
    idx = m.flatten().nonzero().flatten()
    w = (1 / m.sum(-1).clamp(min=1.0)).unsqueeze(-1).expand_as(m)[m]
    return {"collated_global_crops": gc[gi], "collated_local_crops": lc[li], "collated_masks": m, "mask_indices_list": idx,
            "masks_weight": w, "upperbound": int(idx.shape[0]) + 3,
            "n_masked_patches": torch.full((1,), idx.shape[0], dtype=torch.long)}


def _make_arch(centering):
    from octic_vits_amd.ssl import SSLMetaArch
    torch.manual_seed(5)
    return SSLMetaArch(_TinyBackbone, DIM, koleo_loss_weight=0.0, head_n_prototypes=PROTO, head_hidden_dim=40,
                       head_bottleneck_dim=16, centering=centering, local_crops_number=N_LOCAL)


def _run(centering, rows, distributed, steps=3):
    from octic_vits_amd.ssl import SSLTrainer
    arch = _make_arch(centering)
    tr = SSLTrainer(arch, lr=1e-3, autocast=False, distributed=distributed)
    images = _images_of(rows)
    losses = []
    for _ in range(steps):
        ld = tr.step(images, teacher_temp=0.07, momentum=0.9)
        losses.append({k: float(v.detach()) for k, v in ld.items()})
    arch.dino_loss.apply_center_update()
    arch.ibot_patch_loss.apply_center_update()
    return {"losses": losses,
            "grads": {n: p.grad.detach().clone() for n, p in arch.student.named_parameters() if p.grad is not None},
            "student": {n: p.detach().clone() for n, p in arch.student.named_parameters()},
            "teacher": {n: p.detach().clone() for n, p in arch.teacher.named_parameters()},
            "dino_center": arch.dino_loss.center.clone(), "ibot_center": arch.ibot_patch_loss.center.clone()}, arch


def _sinkhorn_probe(arch, rows):
    """Sinkhorn-Knopp targets of fixed logits: per-image rows, so a rank's result must be its rows of the global one."""
    g = torch.Generator().manual_seed(3)
    all_logits = torch.randn(N_IMG, PROTO, generator=g)
    all_patch = torch.randn(N_IMG * 3, PROTO, generator=g)
    rows = torch.as_tensor(rows)
    prow = torch.cat([3 * rows + j for j in range(3)]).sort().values
    d = arch.dino_loss.sinkhorn_knopp_teacher(all_logits[rows], teacher_temp=0.07)
    i = arch.ibot_patch_loss.sinkhorn_knopp_teacher(all_patch[prow], teacher_temp=0.07,
                                                    n_masked_patches_tensor=torch.full((1,), len(prow), dtype=torch.long))
    return d, i, rows, prow


def _worker(rank, world, port, centering, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = N_IMG // world
    rows = list(range(rank * per, (rank + 1) * per))
    res, arch = _run(centering, rows, distributed=True)
    # replicas stay identical (DDP-averaged backbone gradients, all-reduced head gradients, same AdamW)
    flat = torch.cat([p.flatten() for p in res["student"].values()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert all(torch.equal(gathered[0], g_) for g_ in gathered)
    res["sinkhorn"] = _sinkhorn_probe(arch, rows)
    torch.save(res, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("centering", ["centering", "sinkhorn_knopp"])
def test_ssl_trainer_two_gloo_ranks_equal_one_process_on_all_images(tmp_path, centering):
    out = str(tmp_path / "ssl")
    mp.spawn(_worker, args=(2, _free_port(), centering, out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    single, arch = _run(centering, list(range(N_IMG)), distributed=False)

    # losses: every term is a mean over the local images -> the mean over ranks is the global value, at every step (the
    # centre of step k, all-reduced over both ranks, enters step k + 1)
    for k, want in enumerate(single["losses"]):
        for name, w in want.items():
            got = 0.5 * (r0["losses"][k][name] + r1["losses"][k][name])
            assert abs(got - w) <= 2e-5 * max(1.0, abs(w)), (k, name, got, w)
    assert len({round(l["total"], 6) for l in single["losses"]}) == len(single["losses"])     # the steps do differ

    # centre statistics after the asynchronous all-reduce
    for key in ("dino_center", "ibot_center"):
        assert torch.allclose(r0[key], single[key], rtol=1e-5, atol=1e-7), key
        assert torch.equal(r0[key], r1[key])
    if centering == "centering":
        assert float(single["dino_center"].abs().max()) > 0 and float(single["ibot_center"].abs().max()) > 0

    # gradients of the last step (after the per-sub-model clip): backbone via DDP, heads via the manual all-reduce
    assert any(n.startswith("dino_head") for n in single["grads"]) and any(n.startswith("backbone") for n in single["grads"])
    for n, w in single["grads"].items():
        scale = max(1e-6, float(w.abs().max()))
        for r in (r0, r1):
            assert float((r["grads"][n] - w).abs().max()) <= 2e-4 * scale + 1e-8, n
    # three AdamW steps and three teacher EMA updates later
    for part in ("student", "teacher"):
        for n, w in single[part].items():
            assert torch.allclose(r0[part][n], w, rtol=2e-3, atol=2e-5), (part, n)

    # Sinkhorn-Knopp with the cross-rank sums: a rank's targets are its rows of the single-process targets
    d_all, i_all, _, _ = _sinkhorn_probe(arch, list(range(N_IMG)))
    for r in (r0, r1):
        d, i, rows, prow = r["sinkhorn"]
        assert torch.allclose(d, d_all[rows], rtol=1e-5, atol=1e-8)
        assert torch.allclose(i, i_all[prow], rtol=1e-5, atol=1e-8)
