"""Cases for the pieces of the DINOv2 SSL step (losses, head, masking, collate), shared by the golden generator (real
reference), the oracle test and the product tests.  ``ns`` exposes DINOLoss, iBOTPatchLoss, KoLeoLoss, DINOHead,
MaskingGenerator, collate."""
import random

import numpy as np
import torch

import cases

K = 48          # prototypes
D = 32          # backbone width


def _np(t):
    return t.detach().float().cpu().numpy()


def run_pieces(ns, device="cpu"):
    res = {}
    dev = torch.device(device)
    # ---- DINOLoss: centering, centre EMA (applied lazily at the next call), cross entropy over crop lists
    dl = ns.DINOLoss(K).to(dev)
    t1 = cases.randn("ssl.dino.t1", 12, K).to(dev)
    t2 = cases.randn("ssl.dino.t2", 12, K).to(dev)
    p1 = dl.softmax_center_teacher(t1, teacher_temp=0.05)
    dl.update_center(t1)
    p2 = dl.softmax_center_teacher(t2, teacher_temp=0.07)          # uses the centre updated from t1
    dl.update_center(t2)
    dl.apply_center_update()
    res["dino.p1"], res["dino.p2"], res["dino.center"] = _np(p1), _np(p2), _np(dl.center)
    s = [cases.randn(f"ssl.dino.s{i}", 6, K).to(dev).requires_grad_(True) for i in range(3)]
    loss = dl(s, list(p2.view(2, 6, K)))
    loss.backward()
    res["dino.loss"] = _np(loss).reshape(1)
    res["dino.gs0"] = _np(s[0].grad)
    res["dino.sk"] = _np(dl.sinkhorn_knopp_teacher(t1, teacher_temp=0.05))
    # ---- iBOT: masked form with the collate's weights, centre update quirk ([1, n, K] input)
    il = ns.iBOTPatchLoss(K).to(dev)
    n_img, n_tok = 4, 16
    g = cases._gen("ssl.ibot.mask")
    masks = torch.rand(n_img, n_tok, generator=g) < 0.4
    masks[3] = False
    masks = masks.to(dev)
    nm = int(masks.sum())
    tt = cases.randn("ssl.ibot.t", nm, K).to(dev)
    tc = il.softmax_center_teacher(tt.unsqueeze(0)[:, :nm], teacher_temp=0.06).squeeze(0)
    il.update_center(tt.unsqueeze(0)[:nm])
    il.apply_center_update()
    res["ibot.tc"], res["ibot.center"] = _np(tc), _np(il.center)
    st = cases.randn("ssl.ibot.s", nm + 5, K).to(dev).requires_grad_(True)      # padded buffer: only the first nm count
    mw = (1 / masks.sum(-1).clamp(min=1.0)).unsqueeze(-1).expand_as(masks)[masks]
    loss = il.forward_masked(st[:nm], tc, student_masks_flat=masks, n_masked_patches=nm, masks_weight=mw)
    loss.backward()
    res["ibot.loss"], res["ibot.gs"] = _np(loss).reshape(1), _np(st.grad)
    loss2 = il.forward_masked(st[:nm].detach(), tc, student_masks_flat=masks)                 # weights derived inside
    res["ibot.loss_default_weights"] = _np(loss2).reshape(1)
    full_s = cases.randn("ssl.ibot.fs", n_img, n_tok, K).to(dev)
    full_t = torch.softmax(cases.randn("ssl.ibot.ft", n_img, n_tok, K).to(dev), -1)
    res["ibot.loss_dense"] = _np(il(full_s, full_t, masks)).reshape(1)
    # ---- KoLeo
    kl = ns.KoLeoLoss()
    x = cases.randn("ssl.koleo.x", 10, D).to(dev).requires_grad_(True)
    loss = kl(x)
    loss.backward()
    res["koleo.loss"], res["koleo.gx"] = _np(loss).reshape(1), _np(x.grad)
    # ---- DINOHead (weight-normed last layer), forward + backward with name-keyed parameters
    head = cases.fill_parameters(ns.DINOHead(in_dim=D, out_dim=K, hidden_dim=40, bottleneck_dim=24, nlayers=3)).to(dev)
    hx = cases.randn("ssl.head.x", 7, D).to(dev).requires_grad_(True)
    hy = head(hx)
    (hy * cases.randn("ssl.head.cot", 7, K).to(dev)).sum().backward()
    res["head.y"], res["head.gx"] = _np(hy), _np(hx.grad)
    for n, p in head.named_parameters():
        res["head.gpar." + n] = _np(p.grad)
    res["head.keys"] = np.array([sum(ord(c) for c in k) for k in head.state_dict().keys()], dtype=np.int64)
    # ---- MaskingGenerator + collate under a fixed python RNG
    random.seed(1234)
    mg = ns.MaskingGenerator(input_size=(8, 8), max_num_patches=0.5 * 8 * 8)
    res["mask.a"] = np.asarray(mg(20)).astype(np.uint8)
    res["mask.b"] = np.asarray(mg(5)).astype(np.uint8)
    random.seed(4321)
    gc = cases.randn("ssl.coll.g", 8, 3, 8, 8)
    lc = cases.randn("ssl.coll.l", 12, 3, 4, 4)
    out = ns.collate(gc, lc, (0.1, 0.5), 0.5, 64, mg)
    res["coll.masks"] = out["collated_masks"].numpy().astype(np.uint8)
    res["coll.idx"] = out["mask_indices_list"].numpy()
    res["coll.w"] = out["masks_weight"].numpy()
    res["coll.upper"] = np.array([out["upperbound"]])
    res["coll.n"] = out["n_masked_patches"].numpy()
    res["coll.g0"] = _np(out["collated_global_crops"][5])
    res["coll.l0"] = _np(out["collated_local_crops"][7])
    return res
