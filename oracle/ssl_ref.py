"""TEST INFRASTRUCTURE — CPU restatement of the DINOv2 self-supervised step of the reference (BASELINE
configs[4]).  Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this module.

What is restated, with the lines it follows:
  * ``DINOHead``            dinov2/layers/dino_head.py:14-58  (MLP -> L2-normalise -> weight-normed prototypes)
  * ``DINOLoss``            dinov2/loss/dino_clstoken_loss.py:13-99  (softmax-centering, Sinkhorn-Knopp, cross entropy, EMA centre)
  * ``iBOTPatchLoss``       dinov2/loss/ibot_patch_loss.py:37-151  (masked-patch cross entropy; ``lossfunc`` is the reference's
                            own pure-torch fallback, :26-34 — xformers is not installed here or on the GPU box)
  * ``KoLeoLoss``           dinov2/loss/koleo_loss.py:16-48
  * ``MaskingGenerator``    dinov2/data/masking.py:11-86, ``collate`` = dinov2/data/collate.py:10-49
  * ``forward_backward``    dinov2/train/ssl_meta_arch.py:140-354 and ``update_teacher`` :370-379, without FSDP: the teacher
                            and student are plain modules (the reference shards them with FSDP; the arithmetic is the same),
                            ``fmha.BlockDiagonalMask.from_tensor_list`` + ``split`` around the shared head is a concatenation
                            along the token axis and a split at the same places (:246-261).

Pinned: tests/golden/ssl_*.npz are produced by tests/golden/make_ssl_golden.py from the REAL reference classes for the
losses, the head and the mask generator (importable on CPU).  ``ssl_meta_arch.py`` itself cannot be imported here (it asserts
xformers and builds FSDP wrappers), so the composition in ``forward_backward`` is **parity unpinned** beyond those pieces;
local crops at a non-native resolution go through the pos-embed resize branch that raises TypeError in the reference as
shipped — the oracle implements the evident intent (oracle/octic_ref.py ``interpolate_spatial_tuple``).
All arithmetic is plain torch fp32 on the host; distributed reductions are the identity at world size 1.
"""
import math
import random

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------- head
def _build_mlp(nlayers, in_dim, bottleneck_dim, hidden_dim=None, bias=True):
    if nlayers == 1:
        return nn.Linear(in_dim, bottleneck_dim, bias=bias)
    layers = [nn.Linear(in_dim, hidden_dim, bias=bias), nn.GELU()]
    for _ in range(nlayers - 2):
        layers += [nn.Linear(hidden_dim, hidden_dim, bias=bias), nn.GELU()]
    layers.append(nn.Linear(hidden_dim, bottleneck_dim, bias=bias))
    return nn.Sequential(*layers)


class DINOHead(nn.Module):
    def __init__(self, in_dim, out_dim, use_bn=False, nlayers=3, hidden_dim=2048, bottleneck_dim=256, mlp_bias=True):
        super().__init__()
        if use_bn:
            raise NotImplementedError("use_bn is never set by the reference configs")
        nlayers = max(nlayers, 1)
        self.mlp = _build_mlp(nlayers, in_dim, bottleneck_dim, hidden_dim=hidden_dim, bias=mlp_bias)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        self.last_layer = nn.utils.weight_norm(nn.Linear(bottleneck_dim, out_dim, bias=False))
        self.last_layer.weight_g.data.fill_(1)

    def forward(self, x):
        x = self.mlp(x)
        eps = 1e-6 if x.dtype == torch.float16 else 1e-12
        x = F.normalize(x, dim=-1, p=2, eps=eps)
        return self.last_layer(x)


# --------------------------------------------------------------------------------------------- losses
class DINOLoss(nn.Module):
    def __init__(self, out_dim, student_temp=0.1, center_momentum=0.9):
        super().__init__()
        self.student_temp, self.center_momentum = student_temp, center_momentum
        self.register_buffer("center", torch.zeros(1, out_dim))
        self._pending = None            # (sum over the batch, batch length): applied lazily, as in the reference

    @torch.no_grad()
    def softmax_center_teacher(self, teacher_output, teacher_temp):
        self.apply_center_update()
        return F.softmax((teacher_output - self.center) / teacher_temp, dim=-1)

    @torch.no_grad()
    def sinkhorn_knopp_teacher(self, teacher_output, teacher_temp, n_iterations=3):
        Q = torch.exp(teacher_output.float() / teacher_temp).t()     # K x B
        B, K = Q.shape[1], Q.shape[0]
        Q = Q / Q.sum()
        for _ in range(n_iterations):
            Q = Q / Q.sum(dim=1, keepdim=True) / K
            Q = Q / Q.sum(dim=0, keepdim=True) / B
        return (Q * B).t()

    def forward(self, student_output_list, teacher_out_softmaxed_centered_list):
        total = 0
        for s in student_output_list:
            lsm = F.log_softmax(s / self.student_temp, dim=-1)
            for t in teacher_out_softmaxed_centered_list:
                total = total - torch.sum(t * lsm, dim=-1).mean()
        return total

    @torch.no_grad()
    def update_center(self, teacher_output):
        self._pending = (torch.sum(teacher_output, dim=0, keepdim=True), len(teacher_output))

    @torch.no_grad()
    def apply_center_update(self):
        if self._pending is not None:
            s, n = self._pending
            self.center = self.center * self.center_momentum + (s / n) * (1 - self.center_momentum)
            self._pending = None


class iBOTPatchLoss(nn.Module):
    def __init__(self, patch_out_dim, student_temp=0.1, center_momentum=0.9):
        super().__init__()
        self.student_temp, self.center_momentum = student_temp, center_momentum
        self.register_buffer("center", torch.zeros(1, 1, patch_out_dim))
        self._pending = None

    @torch.no_grad()
    def softmax_center_teacher(self, teacher_patch_tokens, teacher_temp):
        self.apply_center_update()
        return F.softmax((teacher_patch_tokens - self.center) / teacher_temp, dim=-1)

    @torch.no_grad()
    def sinkhorn_knopp_teacher(self, teacher_output, teacher_temp, n_masked_patches_tensor, n_iterations=3):
        Q = torch.exp(teacher_output.float() / teacher_temp).t()
        B, K = n_masked_patches_tensor, Q.shape[0]
        Q = Q / Q.sum()
        for _ in range(n_iterations):
            Q = Q / Q.sum(dim=1, keepdim=True) / K
            Q = Q / Q.sum(dim=0, keepdim=True) / B
        return (Q * B).t()

    def forward(self, student_patch_tokens, teacher_patch_tokens, student_masks_flat):
        loss = torch.sum(teacher_patch_tokens * F.log_softmax(student_patch_tokens / self.student_temp, dim=-1), dim=-1)
        loss = torch.sum(loss * student_masks_flat.float(), dim=-1) / student_masks_flat.sum(dim=-1).clamp(min=1.0)
        return -loss.mean()

    def forward_masked(self, student_patch_tokens_masked, teacher_patch_tokens_masked, student_masks_flat,
                       n_masked_patches=None, masks_weight=None):
        loss = torch.sum(teacher_patch_tokens_masked * F.log_softmax(student_patch_tokens_masked / self.student_temp, dim=-1),
                         dim=-1)
        if masks_weight is None:
            masks_weight = ((1 / student_masks_flat.sum(-1).clamp(min=1.0)).unsqueeze(-1)
                            .expand_as(student_masks_flat)[student_masks_flat])
        if n_masked_patches is not None:
            loss = loss[:n_masked_patches]
        return -(loss * masks_weight).sum() / student_masks_flat.shape[0]

    @torch.no_grad()
    def update_center(self, teacher_patch_tokens):
        self._pending = (torch.sum(teacher_patch_tokens.mean(1), dim=0, keepdim=True), len(teacher_patch_tokens))

    @torch.no_grad()
    def apply_center_update(self):
        if self._pending is not None:
            s, n = self._pending
            self.center = self.center * self.center_momentum + (s / n) * (1 - self.center_momentum)
            self._pending = None


class KoLeoLoss(nn.Module):
    def forward(self, student_output, eps=1e-8):
        x = F.normalize(student_output.float(), eps=eps, p=2, dim=-1)
        dots = x @ x.t()
        n = x.shape[0]
        dots.view(-1)[:: (n + 1)].fill_(-1)
        idx = dots.argmax(dim=1)
        distances = F.pairwise_distance(x, x[idx], p=2, eps=1e-8)
        return -torch.log(distances + eps).mean()


# ------------------------------------------------------------------------------------------ data side
class MaskingGenerator:
    """dinov2/data/masking.py: random rectangles until the requested number of patches is masked (python ``random``)."""

    def __init__(self, input_size, num_masking_patches=None, min_num_patches=4, max_num_patches=None, min_aspect=0.3,
                 max_aspect=None):
        if not isinstance(input_size, tuple):
            input_size = (input_size,) * 2
        self.height, self.width = input_size
        self.num_patches = self.height * self.width
        self.num_masking_patches = num_masking_patches
        self.min_num_patches = min_num_patches
        self.max_num_patches = num_masking_patches if max_num_patches is None else max_num_patches
        max_aspect = max_aspect or 1 / min_aspect
        self.log_aspect_ratio = (math.log(min_aspect), math.log(max_aspect))

    def _mask(self, mask, max_mask_patches):
        delta = 0
        for _ in range(10):
            target_area = random.uniform(self.min_num_patches, max_mask_patches)
            aspect_ratio = math.exp(random.uniform(*self.log_aspect_ratio))
            h = int(round(math.sqrt(target_area * aspect_ratio)))
            w = int(round(math.sqrt(target_area / aspect_ratio)))
            if w < self.width and h < self.height:
                top = random.randint(0, self.height - h)
                left = random.randint(0, self.width - w)
                num_masked = mask[top:top + h, left:left + w].sum()
                if 0 < h * w - num_masked <= max_mask_patches:
                    fresh = ~mask[top:top + h, left:left + w]
                    delta += int(fresh.sum())
                    mask[top:top + h, left:left + w] = True
                if delta > 0:
                    break
        return delta

    def __call__(self, num_masking_patches=0):
        mask = np.zeros((self.height, self.width), dtype=bool)
        count = 0
        while count < num_masking_patches:
            delta = self._mask(mask, min(num_masking_patches - count, self.max_num_patches))
            if delta == 0:
                break
            count += delta
        return mask


def collate(global_crops, local_crops, mask_ratio_tuple, mask_probability, n_tokens, mask_generator):
    """dinov2/data/collate.py on already-stacked crops: global_crops [2*B,3,H,W] (crop-major: all first crops, then all
    second crops), local_crops [n_local*B,3,h,w]."""
    B = len(global_crops)
    n_samples_masked = int(B * mask_probability)
    probs = torch.linspace(*mask_ratio_tuple, n_samples_masked + 1)
    upperbound, masks_list = 0, []
    for i in range(n_samples_masked):
        masks_list.append(torch.BoolTensor(mask_generator(int(n_tokens * random.uniform(float(probs[i]), float(probs[i + 1]))))))
        upperbound += int(n_tokens * probs[i + 1])
    for _ in range(n_samples_masked, B):
        masks_list.append(torch.BoolTensor(mask_generator(0)))
    random.shuffle(masks_list)
    collated_masks = torch.stack(masks_list).flatten(1)
    mask_indices_list = collated_masks.flatten().nonzero().flatten()
    masks_weight = (1 / collated_masks.sum(-1).clamp(min=1.0)).unsqueeze(-1).expand_as(collated_masks)[collated_masks]
    return {"collated_global_crops": global_crops, "collated_local_crops": local_crops, "collated_masks": collated_masks,
            "mask_indices_list": mask_indices_list, "masks_weight": masks_weight, "upperbound": upperbound,
            "n_masked_patches": torch.full((1,), fill_value=mask_indices_list.shape[0], dtype=torch.long)}


# ------------------------------------------------------------------------------------------ the step
class SSLMetaArch(nn.Module):
    """ssl_meta_arch.py without FSDP.  ``make_backbone()`` returns an OcticDinoVisionTransformer-like module whose
    ``forward(x | [xs], masks=..., is_training=True)`` yields the feature dict(s)."""

    def __init__(self, make_backbone, embed_dim, *, dino_loss_weight=1.0, koleo_loss_weight=0.1, ibot_loss_weight=1.0,
                 head_n_prototypes=65536, head_hidden_dim=2048, head_bottleneck_dim=256, head_nlayers=3,
                 ibot_separate_head=False, centering="centering", local_crops_number=8):
        super().__init__()
        head = lambda: DINOHead(embed_dim, head_n_prototypes, hidden_dim=head_hidden_dim,
                                bottleneck_dim=head_bottleneck_dim, nlayers=head_nlayers)
        self.do_dino, self.do_koleo, self.do_ibot = dino_loss_weight > 0, koleo_loss_weight > 0, ibot_loss_weight > 0
        self.dino_loss_weight, self.koleo_loss_weight, self.ibot_loss_weight = dino_loss_weight, koleo_loss_weight, ibot_loss_weight
        self.ibot_separate_head, self.centering, self.n_local_crops = ibot_separate_head, centering, local_crops_number
        student, teacher = {"backbone": make_backbone()}, {"backbone": make_backbone()}
        student["dino_head"], teacher["dino_head"] = head(), head()
        if self.do_ibot and ibot_separate_head:
            student["ibot_head"], teacher["ibot_head"] = head(), head()
        self.student, self.teacher = nn.ModuleDict(student), nn.ModuleDict(teacher)
        self.dino_loss, self.koleo_loss = DINOLoss(head_n_prototypes), KoLeoLoss()
        self.ibot_patch_loss = iBOTPatchLoss(head_n_prototypes)
        for k in self.student:                                   # prepare_for_distributed_training: teacher := student
            self.teacher[k].load_state_dict(self.student[k].state_dict())
        for p in self.teacher.parameters():
            p.requires_grad = False

    def train(self, mode=True):
        super().train(mode)
        self.teacher.eval()
        return self

    def forward_backward(self, images, teacher_temp, backward=True):
        n_global_crops, n_local_crops = 2, self.n_local_crops
        global_crops, local_crops = images["collated_global_crops"], images["collated_local_crops"]
        masks, mask_indices_list = images["collated_masks"], images["mask_indices_list"]
        n_masked_patches, upperbound = mask_indices_list.shape[0], images["upperbound"]
        masks_weight = images["masks_weight"]
        n_local_terms = max(n_local_crops * n_global_crops, 1)
        n_global_terms = (n_global_crops - 1) * n_global_crops
        ibot_loss_scale = 1.0 / n_global_crops

        with torch.no_grad():
            tout = self.teacher.backbone(global_crops, is_training=True)
            tcls = tout["x_norm_clstoken"].chunk(n_global_crops)
            tcls = torch.cat((tcls[1], tcls[0]))                  # reversed: crop A's student is matched with crop B's teacher
            tpatch = tout["x_norm_patchtokens"]
            n_cls = tcls.shape[0]
            masked_teacher = None
            if self.do_ibot and not self.ibot_separate_head:
                buf = tpatch.new_zeros(upperbound + n_cls, tpatch.shape[-1])
                buf[:n_cls] = tcls
                buf[n_cls:n_cls + n_masked_patches] = tpatch.flatten(0, 1)[mask_indices_list]
                after = self.teacher.dino_head(buf)
                tcls_after, masked_teacher = after[:n_cls], after[n_cls:n_cls + n_masked_patches]
            elif self.do_ibot:
                buf = tpatch.new_zeros(upperbound, tpatch.shape[-1])
                buf[:n_masked_patches] = tpatch.flatten(0, 1)[mask_indices_list]
                tcls_after = self.teacher.dino_head(tcls)
                masked_teacher = self.teacher.ibot_head(buf)[:n_masked_patches]
            else:
                tcls_after = self.teacher.dino_head(tcls)
            masked_teacher_centered = None
            if self.centering == "centering":
                t_dino = self.dino_loss.softmax_center_teacher(tcls_after, teacher_temp).view(n_global_crops, -1, tcls_after.shape[-1])
                self.dino_loss.update_center(tcls_after)
                if self.do_ibot:
                    mt = masked_teacher.unsqueeze(0)
                    masked_teacher_centered = self.ibot_patch_loss.softmax_center_teacher(mt[:, :n_masked_patches], teacher_temp).squeeze(0)
                    self.ibot_patch_loss.update_center(mt[:n_masked_patches])
            elif self.centering == "sinkhorn_knopp":
                t_dino = self.dino_loss.sinkhorn_knopp_teacher(tcls_after, teacher_temp).view(n_global_crops, -1, tcls_after.shape[-1])
                if self.do_ibot:
                    masked_teacher_centered = self.ibot_patch_loss.sinkhorn_knopp_teacher(
                        masked_teacher, teacher_temp, n_masked_patches_tensor=images["n_masked_patches"])
            else:
                raise NotImplementedError

        loss_dict, total = {}, 0
        sg, sl = self.student.backbone([global_crops, local_crops], masks=[masks, None], is_training=True)
        pieces = [sl["x_norm_clstoken"], sg["x_norm_clstoken"]]
        student_masked_after = None
        if self.do_ibot:
            sp = sg["x_norm_patchtokens"]
            buf = sp.new_zeros(upperbound, sp.shape[-1])
            buf[:n_masked_patches] = sp.flatten(0, 1)[mask_indices_list]
            if not self.ibot_separate_head:
                pieces.append(buf)
            else:
                student_masked_after = self.student.ibot_head(buf)[:n_masked_patches]
        outs = list(self.student.dino_head(torch.cat(pieces, dim=0)).split([p.shape[0] for p in pieces], dim=0))
        sl_after, sg_after = outs.pop(0), outs.pop(0)
        if self.do_ibot and not self.ibot_separate_head:
            student_masked_after = outs.pop(0)[:n_masked_patches]

        if n_local_crops > 0:
            l = self.dino_loss(sl_after.chunk(n_local_crops), t_dino) / (n_global_terms + n_local_terms)
            loss_dict["dino_local_crops_loss"] = l
            total = total + self.dino_loss_weight * l
        loss_scales = 2
        if self.do_dino:
            l = self.dino_loss([sg_after], [t_dino.flatten(0, 1)]) * loss_scales / (n_global_terms + n_local_terms)
            loss_dict["dino_global_crops_loss"] = l
            total = total + self.dino_loss_weight * l
            if self.do_koleo:
                k = self.koleo_loss_weight * sum(self.koleo_loss(p) for p in sg["x_norm_clstoken"].chunk(2))
                total = total + k
                loss_dict["koleo_loss"] = k / loss_scales
        if self.do_ibot:
            l = self.ibot_patch_loss.forward_masked(student_masked_after, masked_teacher_centered, student_masks_flat=masks,
                                                    n_masked_patches=n_masked_patches, masks_weight=masks_weight) * loss_scales * ibot_loss_scale
            loss_dict["ibot_loss"] = l / 2
            total = total + self.ibot_loss_weight * l
        if backward:
            total.backward()
        loss_dict["total"] = total.detach()
        return loss_dict

    @torch.no_grad()
    def update_teacher(self, m):
        for k in self.student:
            for ps, pt in zip(self.student[k].parameters(), self.teacher[k].parameters()):
                pt.mul_(m).add_(ps.detach(), alpha=1 - m)
