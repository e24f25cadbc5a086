"""DINOv2 self-supervised student/teacher step for the octic backbones (BASELINE configs[4]).

Same classes, constructor arguments, ``state_dict`` keys and arithmetic as the reference's
``dinov2/loss/{dino_clstoken_loss,ibot_patch_loss,koleo_loss}.py``, ``dinov2/layers/dino_head.py``,
``dinov2/data/{masking,collate}.py`` and ``dinov2/train/ssl_meta_arch.py:140-379`` — with plain data parallelism instead of
FSDP (parameters and optimizer state of a ViT-H student + teacher are ~13 GB: nothing to shard on a 288 GB device; the
student is wrapped in DistributedDataParallel by ``SSLTrainer`` when a process group exists, the centre statistics are
all-reduced as in the reference).  The backbones run on the HIP engine (octic blocks, attention, standard half); the
heads and losses are a few small dense ops per step and stay on torch.

Local crops (96 x 96 on a 224 model) need the positional-embedding resize branch, which raises TypeError in the
reference as shipped (tuple patch size); ``OcticDinoVisionTransformer`` here implements the evident intent
(bicubic resize of the unfolded grids, ``d8_utils.py:453-499`` with an integer patch side).
"""
import math
import random

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from . import functional as _OF

FUSED_LOSS_ROWS = True      # GPU: the prototype-axis passes of the DINO / iBOT terms on csrc/ssl_loss.hip (False: the eager composition)


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


# ----------------------------------------------------------------------------------------------- head
def _build_mlp(nlayers, in_dim, bottleneck_dim, hidden_dim=None, use_bn=False, bias=True):
    if nlayers == 1:
        return nn.Linear(in_dim, bottleneck_dim, bias=bias)
    layers = [nn.Linear(in_dim, hidden_dim, bias=bias)]
    if use_bn:
        layers.append(nn.BatchNorm1d(hidden_dim))
    layers.append(nn.GELU())
    for _ in range(nlayers - 2):
        layers.append(nn.Linear(hidden_dim, hidden_dim, bias=bias))
        if use_bn:
            layers.append(nn.BatchNorm1d(hidden_dim))
        layers.append(nn.GELU())
    layers.append(nn.Linear(hidden_dim, bottleneck_dim, bias=bias))
    return nn.Sequential(*layers)


class DINOHead(nn.Module):
    """dinov2/layers/dino_head.py:14-41 (keys mlp.{0,2,4}.{weight,bias}, last_layer.weight_g / weight_v)."""

    def __init__(self, in_dim, out_dim, use_bn=False, nlayers=3, hidden_dim=2048, bottleneck_dim=256, mlp_bias=True):
        super().__init__()
        nlayers = max(nlayers, 1)
        self.mlp = _build_mlp(nlayers, in_dim, bottleneck_dim, hidden_dim=hidden_dim, use_bn=use_bn, bias=mlp_bias)
        self.apply(self._init_weights)
        self.last_layer = nn.utils.weight_norm(nn.Linear(bottleneck_dim, out_dim, bias=False))
        self.last_layer.weight_g.data.fill_(1)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        x = self.mlp(x)
        eps = 1e-6 if x.dtype == torch.float16 else 1e-12
        x = F.normalize(x, dim=-1, p=2, eps=eps)
        return self.last_layer(x)


# --------------------------------------------------------------------------------------------- losses
class _Centered(nn.Module):
    """Shared centre bookkeeping of DINOLoss / iBOTPatchLoss: the batch statistic is reduced asynchronously and applied
    at the next use (dino_clstoken_loss.py:75-99, ibot_patch_loss.py:131-151)."""

    def __init__(self, center_shape, center_momentum):
        super().__init__()
        self.center_momentum = center_momentum
        self.register_buffer("center", torch.zeros(*center_shape))
        self.updated, self.reduce_handle, self._len, self.async_batch_center = True, None, None, None

    def _queue(self, batch_sum, length):
        self.updated, self._len, self.async_batch_center = False, length, batch_sum
        if _world() > 1:
            self.reduce_handle = dist.all_reduce(self.async_batch_center, async_op=True)

    @torch.no_grad()
    def apply_center_update(self):
        if self.updated is False:
            if self.reduce_handle is not None:
                self.reduce_handle.wait()
                self.reduce_handle = None
            t = self.async_batch_center / (self._len * _world())
            self.center = self.center * self.center_momentum + t * (1 - self.center_momentum)
            self.updated = True


def _sinkhorn(teacher_output, teacher_temp, B_total, n_iterations):
    Q = torch.exp(teacher_output.float() / teacher_temp).t()       # K x B
    K = Q.shape[0]
    s = torch.sum(Q)
    if _world() > 1:
        dist.all_reduce(s)
    Q = Q / s
    for _ in range(n_iterations):
        r = torch.sum(Q, dim=1, keepdim=True)
        if _world() > 1:
            dist.all_reduce(r)
        Q = Q / r / K
        Q = Q / torch.sum(Q, dim=0, keepdim=True) / B_total
    return (Q * B_total).t()


class DINOLoss(_Centered):
    def __init__(self, out_dim, student_temp=0.1, center_momentum=0.9):
        super().__init__((1, out_dim), center_momentum)
        self.student_temp = student_temp

    @torch.no_grad()
    def softmax_center_teacher(self, teacher_output, teacher_temp):
        self.apply_center_update()
        if FUSED_LOSS_ROWS and _OF.loss_rows_ok(teacher_output):
            return _OF.ops.softmax_center(teacher_output, self.center, 1.0 / teacher_temp)
        return F.softmax((teacher_output - self.center) / teacher_temp, dim=-1)

    @torch.no_grad()
    def sinkhorn_knopp_teacher(self, teacher_output, teacher_temp, n_iterations=3):
        return _sinkhorn(teacher_output, teacher_temp, teacher_output.shape[0] * _world(), n_iterations)

    def forward(self, student_output_list, teacher_out_softmaxed_centered_list):
        sl, tl = list(student_output_list), list(teacher_out_softmaxed_centered_list)
        if (FUSED_LOSS_ROWS and sl and tl and all(_OF.loss_rows_ok(s) and s.shape == sl[0].shape for s in sl)
                and all(t.shape == sl[0].shape for t in tl)):
            # sum_t sum_k t_k lsm_k is linear in t: every student crop meets the SUM of the teacher crops, all crops in one
            # launch (row r of the stacked crops reads teacher row r % B)
            tsum = tl[0] if len(tl) == 1 else torch.stack([t.float() for t in tl]).sum(0)
            s = sl[0] if len(sl) == 1 else torch.cat(sl, dim=0)
            rows = _OF.SoftCrossEntropyFn.apply(s, tsum, self.student_temp)
            return rows.sum() / sl[0].shape[0]
        total_loss = 0
        for s in student_output_list:
            lsm = F.log_softmax(s / self.student_temp, dim=-1)
            for t in teacher_out_softmaxed_centered_list:
                total_loss = total_loss - torch.sum(t * lsm, dim=-1).mean()
        return total_loss

    @torch.no_grad()
    def update_center(self, teacher_output):
        self._queue(torch.sum(teacher_output, dim=0, keepdim=True), len(teacher_output))


class iBOTPatchLoss(_Centered):
    def __init__(self, patch_out_dim, student_temp=0.1, center_momentum=0.9):
        super().__init__((1, 1, patch_out_dim), center_momentum)
        self.student_temp = student_temp

    @torch.no_grad()
    def softmax_center_teacher(self, teacher_patch_tokens, teacher_temp):
        self.apply_center_update()
        if FUSED_LOSS_ROWS and _OF.loss_rows_ok(teacher_patch_tokens):
            return _OF.ops.softmax_center(teacher_patch_tokens, self.center, 1.0 / teacher_temp)
        return F.softmax((teacher_patch_tokens - self.center) / teacher_temp, dim=-1)

    @torch.no_grad()
    def sinkhorn_knopp_teacher(self, teacher_output, teacher_temp, n_masked_patches_tensor, n_iterations=3):
        B = n_masked_patches_tensor.clone()
        if _world() > 1:
            dist.all_reduce(B)
        return _sinkhorn(teacher_output, teacher_temp, B, n_iterations)

    def forward(self, student_patch_tokens, teacher_patch_tokens, student_masks_flat):
        loss = torch.sum(teacher_patch_tokens * F.log_softmax(student_patch_tokens / self.student_temp, dim=-1), dim=-1)
        loss = torch.sum(loss * student_masks_flat.float(), dim=-1) / student_masks_flat.sum(dim=-1).clamp(min=1.0)
        return -loss.mean()

    def forward_masked(self, student_patch_tokens_masked, teacher_patch_tokens_masked, student_masks_flat,
                       n_masked_patches=None, masks_weight=None):
        # ibot_patch_loss.py:26-34: the pure-torch lossfunc (the xformers cross_entropy twin is CUDA-only)
        if (FUSED_LOSS_ROWS and _OF.loss_rows_ok(student_patch_tokens_masked) and student_patch_tokens_masked.dim() == 2
                and teacher_patch_tokens_masked.shape == student_patch_tokens_masked.shape):
            loss = -_OF.SoftCrossEntropyFn.apply(student_patch_tokens_masked, teacher_patch_tokens_masked, self.student_temp)
            if masks_weight is None:
                masks_weight = ((1 / student_masks_flat.sum(-1).clamp(min=1.0)).unsqueeze(-1)
                                .expand_as(student_masks_flat)[student_masks_flat])
            if n_masked_patches is not None:
                loss = loss[:n_masked_patches]
            return -(loss * masks_weight).sum() / student_masks_flat.shape[0]
        loss = torch.sum(teacher_patch_tokens_masked.float()
                         * F.log_softmax(student_patch_tokens_masked.float() / self.student_temp, dim=-1), dim=-1)
        if masks_weight is None:
            masks_weight = ((1 / student_masks_flat.sum(-1).clamp(min=1.0)).unsqueeze(-1)
                            .expand_as(student_masks_flat)[student_masks_flat])
        if n_masked_patches is not None:
            loss = loss[:n_masked_patches]
        return -(loss * masks_weight).sum() / student_masks_flat.shape[0]

    @torch.no_grad()
    def update_center(self, teacher_patch_tokens):
        self._queue(torch.sum(teacher_patch_tokens.mean(1), dim=0, keepdim=True), len(teacher_patch_tokens))


class KoLeoLoss(nn.Module):
    """koleo_loss.py:16-48 (autocast disabled inside, as in the reference)."""

    def __init__(self):
        super().__init__()
        self.pdist = nn.PairwiseDistance(2, eps=1e-8)

    def pairwise_NNs_inner(self, x):
        dots = torch.mm(x, x.t())
        n = x.shape[0]
        dots.view(-1)[:: (n + 1)].fill_(-1)
        return torch.max(dots, dim=1)[1]

    def forward(self, student_output, eps=1e-8):
        with torch.autocast(device_type=student_output.device.type, enabled=False):
            student_output = F.normalize(student_output.float(), eps=eps, p=2, dim=-1)
            idx = self.pairwise_NNs_inner(student_output)
            distances = self.pdist(student_output, student_output[idx])
            return -torch.log(distances + eps).mean()


# ------------------------------------------------------------------------------------------ data side
class MaskingGenerator:
    """dinov2/data/masking.py:11-86."""

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

    def get_shape(self):
        return self.height, self.width

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
                window = mask[top:top + h, left:left + w]
                if 0 < h * w - window.sum() <= max_mask_patches:
                    delta += int((~window).sum())
                    window[...] = True
                if delta > 0:
                    break
        return delta

    def __call__(self, num_masking_patches=0):
        mask = np.zeros(shape=self.get_shape(), dtype=bool)
        mask_count = 0
        while mask_count < num_masking_patches:
            delta = self._mask(mask, min(num_masking_patches - mask_count, self.max_num_patches))
            if delta == 0:
                break
            mask_count += delta
        return mask


def collate(global_crops, local_crops, mask_ratio_tuple, mask_probability, n_tokens, mask_generator, dtype=None):
    """dinov2/data/collate.py:10-49 on crops that are already stacked crop-major ([2*B,...] / [n_local*B,...])."""
    B = len(global_crops)
    n_samples_masked = int(B * mask_probability)
    probs = torch.linspace(*mask_ratio_tuple, n_samples_masked + 1)
    upperbound, masks_list = 0, []
    for i in range(n_samples_masked):
        prob_min, prob_max = probs[i], probs[i + 1]
        masks_list.append(torch.BoolTensor(mask_generator(int(n_tokens * random.uniform(prob_min, prob_max)))))
        upperbound += int(n_tokens * prob_max)
    for _ in range(n_samples_masked, B):
        masks_list.append(torch.BoolTensor(mask_generator(0)))
    random.shuffle(masks_list)
    collated_masks = torch.stack(masks_list).flatten(1)
    mask_indices_list = collated_masks.flatten().nonzero().flatten()
    masks_weight = (1 / collated_masks.sum(-1).clamp(min=1.0)).unsqueeze(-1).expand_as(collated_masks)[collated_masks]
    return {"collated_global_crops": global_crops if dtype is None else global_crops.to(dtype),
            "collated_local_crops": local_crops if dtype is None else local_crops.to(dtype),
            "collated_masks": collated_masks, "mask_indices_list": mask_indices_list, "masks_weight": masks_weight,
            "upperbound": upperbound,
            "n_masked_patches": torch.full((1,), fill_value=mask_indices_list.shape[0], dtype=torch.long)}


def synthetic_multicrop_batch(batch, device, seed, global_size=224, local_size=96, n_local=8, patch_size=16,
                              mask_ratio_tuple=(0.1, 0.5), mask_probability=0.5):
    """randn crops of the reference's multi-crop geometry (2 global + n_local local crops per image,
    ssl_default_config.yaml:108-117) with iBOT masks drawn like the reference's collate (python RNG seeded)."""
    g = torch.Generator().manual_seed(seed)
    random.seed(seed)
    gc = torch.randn(2 * batch, 3, global_size, global_size, generator=g)
    lc = torch.randn(n_local * batch, 3, local_size, local_size, generator=g)
    side = global_size // patch_size
    mg = MaskingGenerator(input_size=(side, side), max_num_patches=0.5 * side * side)
    out = collate(gc, lc, mask_ratio_tuple, mask_probability, side * side, mg)
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in out.items()}


# ------------------------------------------------------------------------------------------ the step
class SSLMetaArch(nn.Module):
    """ssl_meta_arch.py:30-379 for plain data parallelism.  ``make_backbone()`` builds one OcticDinoVisionTransformer."""

    def __init__(self, make_backbone, embed_dim, *, dino_loss_weight=1.0, koleo_loss_weight=0.1, ibot_loss_weight=1.0,
                 head_n_prototypes=65536, head_hidden_dim=2048, head_bottleneck_dim=256, head_nlayers=3,
                 ibot_separate_head=False, centering="centering", local_crops_number=8):
        super().__init__()
        head = lambda: DINOHead(in_dim=embed_dim, out_dim=head_n_prototypes, hidden_dim=head_hidden_dim,
                                bottleneck_dim=head_bottleneck_dim, nlayers=head_nlayers)
        self.embed_dim, self.dino_out_dim = embed_dim, head_n_prototypes
        self.do_dino, self.do_koleo, self.do_ibot = dino_loss_weight > 0, koleo_loss_weight > 0, ibot_loss_weight > 0
        self.dino_loss_weight, self.koleo_loss_weight, self.ibot_loss_weight = dino_loss_weight, koleo_loss_weight, ibot_loss_weight
        self.ibot_separate_head, self.centering, self.n_local_crops = ibot_separate_head, centering, local_crops_number
        student, teacher = {"backbone": make_backbone()}, {"backbone": make_backbone()}
        for m in (student["backbone"], teacher["backbone"]):
            # the reference builds PatchEmbedD8 with strict_img_size=True (model.py:93, d8_layers.py:422,455), which
            # rejects the 96 x 96 local crops of its own multi-crop recipe before the pos-embed TypeError is even reached;
            # the non-strict check (d8_layers.py:459-462: sides divisible by 2 * patch) is what multi-crop needs
            if hasattr(m, "patch_embed") and hasattr(m.patch_embed, "strict_img_size"):
                m.patch_embed.strict_img_size = False
        if self.do_dino or self.do_ibot:
            student["dino_head"], teacher["dino_head"] = head(), head()
        if self.do_ibot and ibot_separate_head:
            student["ibot_head"], teacher["ibot_head"] = head(), head()
        self.student, self.teacher = nn.ModuleDict(student), nn.ModuleDict(teacher)
        self.dino_loss, self.koleo_loss = DINOLoss(head_n_prototypes), KoLeoLoss()
        self.ibot_patch_loss = iBOTPatchLoss(head_n_prototypes)
        for k in self.student:                                   # prepare_for_distributed_training: teacher := student
            self.teacher[k].load_state_dict(self.student[k].state_dict())
        for p in self.teacher.parameters():                       # no backpropagation through the teacher
            p.requires_grad = False
        self._student_call = None                                 # set by SSLTrainer when the student is DDP-wrapped

    def train(self, mode=True):
        super().train(mode)
        self.teacher.eval()
        return self

    def _student_forward(self, global_crops, local_crops, masks):
        """Backbone on both crop sets + the shared head on [local cls | global cls | masked patch tokens] in one call
        (ssl_meta_arch.py:226-267).  One function so that DistributedDataParallel sees a single forward."""
        sg, sl = self.student.backbone([global_crops, local_crops], masks=[masks, None], is_training=True)
        return sg, sl

    def forward_backward(self, images, teacher_temp, backward=True, backward_scope=None):
        """backward_scope: a context-manager factory entered around total.backward() (SSLTrainer: batched parameter-gradient
        finishes where every block module was used once in the pass)."""
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
            tcls = torch.cat((tcls[1], tcls[0]))                  # reversed so crop A is matched with crop B
            tpatch = tout["x_norm_patchtokens"]
            n_cls = tcls.shape[0]
            masked_teacher = masked_teacher_centered = None
            if self.do_ibot and not self.ibot_separate_head:
                buf = tpatch.new_zeros(upperbound + n_cls, tpatch.shape[-1])
                buf[:n_cls].copy_(tcls)
                torch.index_select(tpatch.flatten(0, 1), dim=0, index=mask_indices_list,
                                   out=buf[n_cls:n_cls + n_masked_patches])
                after = self.teacher.dino_head(buf)
                tcls_after, masked_teacher = after[:n_cls], after[n_cls:n_cls + n_masked_patches]
            elif self.do_ibot:
                buf = tpatch.new_zeros(upperbound, tpatch.shape[-1])
                torch.index_select(tpatch.flatten(0, 1), dim=0, index=mask_indices_list, out=buf[:n_masked_patches])
                tcls_after = self.teacher.dino_head(tcls)
                masked_teacher = self.teacher.ibot_head(buf)[:n_masked_patches]
            else:
                tcls_after = self.teacher.dino_head(tcls)
            if self.centering == "centering":
                t_dino = self.dino_loss.softmax_center_teacher(tcls_after, teacher_temp=teacher_temp).view(
                    n_global_crops, -1, *tcls_after.shape[1:])
                self.dino_loss.update_center(tcls_after)
                if self.do_ibot:
                    mt = masked_teacher.unsqueeze(0)
                    masked_teacher_centered = self.ibot_patch_loss.softmax_center_teacher(
                        mt[:, :n_masked_patches], teacher_temp=teacher_temp).squeeze(0)
                    self.ibot_patch_loss.update_center(mt[:n_masked_patches])
            elif self.centering == "sinkhorn_knopp":
                t_dino = self.dino_loss.sinkhorn_knopp_teacher(tcls_after, teacher_temp=teacher_temp).view(
                    n_global_crops, -1, *tcls_after.shape[1:])
                if self.do_ibot:
                    masked_teacher_centered = self.ibot_patch_loss.sinkhorn_knopp_teacher(
                        masked_teacher, teacher_temp=teacher_temp, n_masked_patches_tensor=images["n_masked_patches"])
            else:
                raise NotImplementedError

        loss_dict, total = {}, 0
        sg, sl = (self._student_call or self._student_forward)(global_crops, local_crops, masks)
        pieces = [sl["x_norm_clstoken"], sg["x_norm_clstoken"]]
        student_masked_after = None
        if self.do_ibot:
            sp = sg["x_norm_patchtokens"]
            buf = sp.new_zeros(upperbound, sp.shape[-1])
            buf[:n_masked_patches].copy_(torch.index_select(sp.flatten(0, 1), dim=0, index=mask_indices_list))
            if not self.ibot_separate_head:
                pieces.append(buf)
            else:
                student_masked_after = self.student.ibot_head(buf)[:n_masked_patches]
        # fmha.BlockDiagonalMask.from_tensor_list / split around the head = concatenate along tokens, split back
        outs = list(self.student.dino_head(torch.cat(pieces, dim=0)).split([p.shape[0] for p in pieces], dim=0))
        sl_after, sg_after = outs.pop(0), outs.pop(0)
        if self.do_ibot and not self.ibot_separate_head:
            student_masked_after = outs.pop(0)[:n_masked_patches]

        if n_local_crops > 0:
            l = self.dino_loss(student_output_list=sl_after.chunk(n_local_crops),
                               teacher_out_softmaxed_centered_list=t_dino) / (n_global_terms + n_local_terms)
            loss_dict["dino_local_crops_loss"] = l
            total = total + self.dino_loss_weight * l
        loss_scales = 2                                           # the two global crops are processed together
        if self.do_dino:
            l = (self.dino_loss(student_output_list=[sg_after], teacher_out_softmaxed_centered_list=[t_dino.flatten(0, 1)])
                 * loss_scales / (n_global_terms + n_local_terms))
            loss_dict["dino_global_crops_loss"] = l
            total = total + self.dino_loss_weight * l
            if self.do_koleo:
                k = self.koleo_loss_weight * sum(self.koleo_loss(p) for p in sg["x_norm_clstoken"].chunk(2))
                total = total + k
                loss_dict["koleo_loss"] = k / loss_scales
        if self.do_ibot:
            l = (self.ibot_patch_loss.forward_masked(student_masked_after, masked_teacher_centered, student_masks_flat=masks,
                                                     n_masked_patches=n_masked_patches, masks_weight=masks_weight)
                 * loss_scales * ibot_loss_scale)
            loss_dict["ibot_loss"] = l / 2
            total = total + self.ibot_loss_weight * l
        if backward:
            if backward_scope is None:
                total.backward()
            else:
                with backward_scope():
                    total.backward()
        loss_dict["total"] = total.detach()
        return loss_dict

    @torch.no_grad()
    def update_teacher(self, m):
        sp = [p for k in self.student for p in self.student[k].parameters()]
        tp = [p for k in self.student for p in self.teacher[k].parameters()]
        torch._foreach_mul_(tp, m)
        torch._foreach_add_(tp, [p.detach() for p in sp], alpha=1 - m)
        from .functional import invalidate_weight_caches
        invalidate_weight_caches(self.teacher)                    # raw writes: the teacher's bf16 weight copies are stale

    @staticmethod
    def _no_decay(name, p):
        return (name.endswith(".bias") or "norm" in name or "gamma" in name or ".ls" in name or "alpha" in name
                or "token" in name or "pos_embed" in name or p.ndim <= 1)

    def get_params_groups(self, weight_decay=0.04):
        """AdamW groups: no weight decay for biases, norms, layer scales (gamma / ls / alpha) and tokens
        (dinov2/utils/param_groups.py:68-94 without the layer-wise lr decay, which only rescales lr per group)."""
        decay, no_decay = [], []
        for name, p in self.student.named_parameters():
            if not p.requires_grad:
                continue
            (no_decay if self._no_decay(name, p) else decay).append(p)
        return [{"params": decay, "weight_decay": weight_decay}, {"params": no_decay, "weight_decay": 0.0}]


class _TeacherRefresh:
    """The teacher's bf16 operand copies, remade in two launches right after the fused step rewrote its parameters (EMA): one
    ``octic_dense_prep_batch`` for every projection of the standard half and one ``octic_linear_d8_prep_batch`` for the LinearD8
    layers, adopted by the layers' caches - instead of ~145 per-layer casts at the teacher's next forward.  The values are the
    round-to-nearest casts the caches would make themselves."""

    def __init__(self, teacher):
        import numpy as np
        from . import _lib
        from .functional import PrepBatch
        from .train import library_gemm_layers, octic_weight_preps
        self._lib = _lib
        self.layers = [(lin, cache) for lin, cache, _ in library_gemm_layers(teacher)
                       if lin.weight.is_cuda and lin.weight.dtype == torch.float32 and lin.weight.is_contiguous()]
        self.items, self.bufs = None, []
        if self.layers:
            dev = self.layers[0][0].weight.device
            dt = np.dtype([("src", "<u8"), ("wb", "<u8"), ("wt", "<u8"), ("N", "<i4"), ("K", "<i4"), ("block_begin", "<i4"),
                           ("pad", "<i4")])
            tab = np.zeros(len(self.layers), dtype=dt)
            blocks = 0
            for i, (lin, _) in enumerate(self.layers):
                N, K = lin.weight.shape
                wb = torch.empty((N, K), dtype=torch.bfloat16, device=dev)
                bb = None if lin.bias is None else torch.empty_like(lin.bias, dtype=torch.bfloat16)
                self.bufs.append((wb, bb))
                tab[i]["src"], tab[i]["wb"], tab[i]["wt"] = lin.weight.data_ptr(), wb.data_ptr(), 0
                tab[i]["N"], tab[i]["K"], tab[i]["block_begin"] = N, K, blocks
                blocks += int(_lib.lib().octic_dense_prep_batch_blocks(N, K))
            self.items = torch.from_numpy(tab.view(np.uint8).copy()).to(dev)
            self.blocks = blocks
        self.prep = PrepBatch(octic_weight_preps(teacher))
        self._ptrs = self._current()

    def _current(self):
        return tuple(t.data_ptr() for lin, _ in self.layers for t in (lin.weight, lin.bias) if t is not None)

    def stale(self):
        return self._current() != self._ptrs or self.prep.stale()

    @torch.no_grad()
    def run(self):
        import ctypes
        if self.items is not None:
            stream = ctypes.c_void_p(torch.cuda.current_stream(self.items.device).cuda_stream)
            self._lib.check(self._lib.lib().octic_dense_prep_batch(ctypes.c_void_p(self.items.data_ptr()), len(self.layers),
                                                                   self.blocks, self._lib.F32, stream))
            pairs = [(bb, lin.bias) for (lin, _), (_, bb) in zip(self.layers, self.bufs) if bb is not None]
            if pairs:
                torch._foreach_copy_([d for d, _ in pairs], [s.detach() for _, s in pairs])
            for (lin, cache), (wb, bb) in zip(self.layers, self.bufs):
                cache.adopt(lin.weight, lin.bias, wb, bb, torch.bfloat16)
        self.prep.run()


class SSLTrainer:
    """One iteration of dinov2/train/train.py:253-296: schedules' values are arguments; zero_grad -> forward_backward (bf16
    autocast student, ssl_default_config.yaml:25-31) -> clip_grad_norm_(3.0) per sub-model -> AdamW -> teacher EMA."""

    def __init__(self, arch: SSLMetaArch, lr=1e-3, weight_decay=0.04, betas=(0.9, 0.999), clip_grad=3.0, autocast=True,
                 distributed=False, local_rank=0, fused_optimizer=None, own_reducer=True):
        """fused_optimizer (default: on the GPU): clip_grad_norm_ per sub-model, AdamW and the teacher's EMA as the fused
        multi-tensor HIP step (train.FusedLamb(adam=True) = octic_adamw_step: two streaming passes per sub-model that also
        leave the bf16 weight copies of the standard half and the prepared LinearD8 weights behind) instead of ~60 foreach
        launches + two more passes over the teacher; False: torch.optim.AdamW + foreach EMA, as the reference runs it.
        own_reducer (distributed only, round 6): average every student gradient (backbone and heads) with train.GradReducer
        instead of a DistributedDataParallel wrapper around the backbone + a flat all-reduce for the rest - no autograd hooks,
        so the step keeps its batched finishes and paired weight gradients at N > 1; a pass that uses its modules more than once
        (the set-by-set crop loop) reduces everything after the backward pass."""
        self.arch, self.clip_grad, self.autocast = arch, clip_grad, autocast
        self.device_type = next(arch.parameters()).device.type
        self.lr, self.weight_decay, self.betas = lr, weight_decay, betas
        self.fused = (self.device_type == "cuda") if fused_optimizer is None else bool(fused_optimizer)
        self._fused_opts = None                                   # built after the first backward (which tensors get gradients)
        self._teacher_refresh = None
        self.optimizer = None
        if not self.fused:
            self.optimizer = torch.optim.AdamW(arch.get_params_groups(weight_decay), lr=lr, betas=betas)
            from .functional import track_optimizer
            track_optimizer(arch.student, self.optimizer)         # AdamW's foreach path updates parameters in place
        self._reducer = None
        self._ddp, self._head_params = None, []
        if distributed and own_reducer:
            from .train import GradReducer
            ps = [p for p in arch.student.parameters() if p.requires_grad]
            if all(p.dtype == torch.float32 for p in ps):
                self._reducer = GradReducer(ps)
                self._reducer.broadcast(list(arch.student.parameters()) + list(arch.student.buffers()))
        if distributed and self._reducer is None:
            class _Fwd(nn.Module):
                def __init__(s, a):
                    super().__init__()
                    s.backbone = a.student.backbone

                def forward(s, g, l, m):
                    return s.backbone([g, l], masks=[m, None], is_training=True)
            from .train import FlatGradReducer, ddp_ignore_small
            fwd = _Fwd(arch)
            # the backbone's small tensors stay out of DistributedDataParallel (one launch per tensor in its reducer) and join
            # the heads - which are outside the wrapper anyway - in ONE flat all-reduce after the backward pass
            small = ddp_ignore_small(fwd)
            self._head_params = [p for k in arch.student if k != "backbone" for p in arch.student[k].parameters()]
            self._flat_reducer = FlatGradReducer(small + [p for p in self._head_params if p.requires_grad])
            FlatGradReducer(small).broadcast()                    # (the heads are built identical: ssl_meta_arch seeds them)
            self._ddp = nn.parallel.DistributedDataParallel(
                fwd, device_ids=[local_rank] if self.device_type == "cuda" else None, bucket_cap_mb=128,
                gradient_as_bucket_view=True)
            arch._student_call = lambda g, l, m: tuple(self._ddp(g, l, m))

    def step(self, images, teacher_temp=0.07, momentum=0.992):
        self.arch.train()
        if self.optimizer is not None:
            self.optimizer.zero_grad(set_to_none=True)
        else:
            for p in self.arch.student.parameters():
                p.grad = None
        if self._reducer is not None:
            # gradients land in the reducer's buckets; buckets go out as they fill only when every module is used once in the
            # pass (a second use ADDS to a gradient that an early collective would already have taken)
            bb = self.arch.student["backbone"] if "backbone" in self.arch.student else None
            self._reducer.begin(hold=not getattr(bb, "_single_use_pass", False))
        try:
            if self.autocast:
                with torch.autocast(self.device_type, dtype=torch.bfloat16):
                    loss_dict = self.arch.forward_backward(images, teacher_temp, backward_scope=self._finish_scope)
            else:
                loss_dict = self.arch.forward_backward(images, teacher_temp, backward_scope=self._finish_scope)
        except BaseException:
            if self._reducer is not None:
                self._reducer.abort()
            raise
        if self._reducer is not None:
            self._reducer.finish()
        if self._ddp is not None and _world() > 1:                # the heads (outside the DDP wrapper) + the backbone's small tensors
            self._flat_reducer.reduce()
        if self.fused:
            self._fused_step(momentum)
            return loss_dict
        if self.clip_grad:
            for k in self.arch.student:
                torch.nn.utils.clip_grad_norm_(self.arch.student[k].parameters(), self.clip_grad)
        self.optimizer.step()
        self.arch.update_teacher(momentum)
        return loss_dict

    def _finish_scope(self):
        """The parameter-gradient slab reductions of the pass as batched launches at its end, and the qkv + proj weight gradients
        of a standard block as one launch (ops.DEFERRED_FINISHES, functional.WGRAD_PAIRED) - only where nothing reads a
        parameter gradient before the end of backward: no DDP hooks, and every block module used ONCE in the pass (the crop
        sets as one row tensor: a module used twice would have autograd add two gradient buffers before they are written)."""
        from . import ops
        from .train import BATCHED_FINISHES, _FinishScope
        bb = self.arch.student["backbone"] if "backbone" in self.arch.student else None
        safe = (BATCHED_FINISHES and self.device_type == "cuda" and self._ddp is None
                and getattr(bb, "_single_use_pass", False))
        return _FinishScope(ops.DEFERRED_FINISHES, safe)

    def _fused_step(self, momentum):
        """One octic_adamw_step per sub-model (dinov2/train/train.py:274-296: clip per sub-model, optimizer step, teacher EMA).
        Like torch.optim.AdamW, tensors that receive no gradient are left alone - the set is fixed by the first step."""
        from .train import FusedLamb, library_gemm_layers, octic_weight_preps
        arch = self.arch
        if self._fused_opts is None:
            self._fused_opts = {}
            for k in arch.student:
                sp = dict(arch.student[k].named_parameters())
                tp = dict(arch.teacher[k].named_parameters())
                live = [(n, p) for n, p in sp.items() if p.requires_grad and p.grad is not None]
                if not live:
                    continue
                groups = [{"params": [p for n, p in live if not arch._no_decay(k + "." + n, p)], "weight_decay": self.weight_decay},
                          {"params": [p for n, p in live if arch._no_decay(k + "." + n, p)], "weight_decay": 0.0}]
                twin = {id(p): tp[n].data for n, p in live}
                sub = arch.student[k]
                self._fused_opts[k] = (FusedLamb(groups, lr=self.lr, betas=self.betas, eps=1e-8,
                                                 max_grad_norm=self.clip_grad or 0.0, ema_decay=momentum, adam=True,
                                                 ema_tensors=twin, shadow_layers=library_gemm_layers(sub),
                                                 prep_source=(lambda s=sub: octic_weight_preps(s))), len(live))
            # the teacher's tensors without a live student twin (frozen tokens) follow the plain EMA rule once here and stay
        for k, (opt, nlive) in self._fused_opts.items():
            have = sum(1 for p in arch.student[k].parameters() if p.requires_grad and p.grad is not None)
            if have != nlive:
                raise RuntimeError(f"SSLTrainer: the set of student tensors with gradients changed ({have} vs {nlive} in '{k}')")
            opt.lr, opt.ema_decay = self.lr, momentum
            opt.step()
        from .functional import invalidate_weight_caches
        invalidate_weight_caches(arch.teacher)                    # the kernel rewrote the teacher's parameters
        if self._teacher_refresh is None or self._teacher_refresh.stale():
            self._teacher_refresh = _TeacherRefresh(arch.teacher)
        self._teacher_refresh.run()
