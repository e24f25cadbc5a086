"""Several crop sets of different sequence lengths as ONE token-row tensor (round 5).

The DINOv2 student runs every block on two crop sets - global crops (B_g samples of T_g tokens) and local crops (B_l of T_l) -
and the reference loops over them block by block (`NestedTensorBlockD8.forward_nested`, octic_vits/d8_layers.py:780-794; DINOv2's
own block packs them with xformers, dinov2/layers/block.py:234-260).  Everything in a block except the softmax core is row-wise,
so the engine keeps the token rows of all sets in one tensor `[1, R, D]` (set i = rows [off_i, off_i + B_i T_i), sample-major):
LayerNorm, the GEMMs, GELU and the residual tails run ONCE on R rows - every parameter enters the graph once (no per-tensor
gradient additions: ~1000 launches per ViT-H step), the GEMMs see one larger M - and only attention walks the sets
(`AttnPackedRaggedFn`, `AttnQKVRaggedFn`: one launch per set into one output / one gradient tensor).  Per-sample factors
(stochastic-depth masks, the b / keep of DINOv2's batch-subset stochastic depth) become per-ROW factors (`rps = 1`).

`functional.RAGGED` holds the description of the tensor the blocks are currently looking at (set by the model's block loop,
`dinov2_models.OcticDinoVisionTransformer.forward_features_list`); the attention modules consult it when they meet a
`[1, R, .]` tensor with R = its row count."""
import torch

from . import ops


_CONST_SCALES = {}         # (sets, per-set constants, device) -> [R] f32 per-row factors
# Test hooks (None = draw): MASK_SOURCE(live drop-path modules, ragged) -> [len(live), samples] 0/1 keep masks;
# PERM_SOURCE(n, ragged) -> per set an [n, B] integer tensor of permutations.  They let a test replay the draws of a
# reference run (tests/test_ssl_gpu.py: the oracle's masks and batch subsets) through the one-shot draws of a pass.
MASK_SOURCE = None
PERM_SOURCE = None


class Ragged:
    __slots__ = ("sets", "rows", "samples", "masks", "perms", "perm_at", "rowmaps", "map_keeps")

    def __init__(self, shapes):
        """shapes: [(B_i, T_i)]"""
        self.sets, off = [], 0
        for B, T in shapes:
            self.sets.append((int(B), int(T), off))
            off += int(B) * int(T)
        self.rows = off
        self.samples = sum(b for b, _, _ in self.sets)
        self.masks, self.perms, self.perm_at = None, None, 0      # draws of one forward pass, made at once (draw_*)
        self.rowmaps, self.map_keeps = None, None

    def matches(self, t):
        return t.dim() == 3 and t.shape[0] == 1 and t.shape[1] == self.rows

    def views(self, t, width=None):
        """Per set: t[0, rows of the set] as [B, T, width]."""
        w = t.shape[-1] if width is None else width
        return [t[0, off:off + B * T].view(B, T, w) for B, T, off in self.sets]

    def split(self, t):
        return self.views(t)

    def row_scale(self, per_sample):
        """[sum B_i] per-sample factors (set-major) -> [R] per-row factors."""
        out, s0 = [], 0
        for B, T, _ in self.sets:
            out.append(per_sample[s0:s0 + B].repeat_interleave(T))
            s0 += B
        return torch.cat(out)

    def row_to_sample(self, device):
        """[R] int64: the (set-major) sample index of every row; cached."""
        key = (tuple(self.sets), "r2s", str(device))
        t = _CONST_SCALES.get(key)
        if t is None:
            t = _CONST_SCALES[key] = self.row_scale(torch.arange(self.samples, device=device))
        return t

    def draw_masks(self, drop_paths, device):
        """Stochastic-depth factors of a whole forward pass in four launches: one Bernoulli(keep) / keep per sample for each
        of the given DropPathD8-like modules (drop_prob, scale_by_keep), expanded to per-row factors.  `masks[id(module)]` is
        what BlockD8 takes instead of drawing its own (timm's drop_path draws per call: the same distribution)."""
        live = [d for d in drop_paths if getattr(d, "drop_prob", 0.) > 0. and d.training]
        if not live:
            return
        keeps = tuple(1. - d.drop_prob for d in live)
        divs = tuple((1. - d.drop_prob) if (d.scale_by_keep and d.drop_prob < 1.) else 1. for d in live)
        key = (keeps, divs, "keep", str(device))
        kd = _CONST_SCALES.get(key)
        if kd is None:
            kd = _CONST_SCALES[key] = (torch.tensor(keeps, dtype=torch.float32, device=device).unsqueeze(1),
                                       torch.tensor(divs, dtype=torch.float32, device=device).unsqueeze(1))
        keep, div = kd
        if MASK_SOURCE is not None:
            m = MASK_SOURCE(live, self).to(device=device, dtype=torch.float32) / div
        else:
            m = (torch.rand(len(live), self.samples, device=device) < keep).float() / div
        rows = m.index_select(1, self.row_to_sample(device))
        self.masks = {id(d): rows[i] for i, d in enumerate(live)}

    def draw_perms(self, n, device, keeps=None):
        """n random permutations of every set's samples in two launches per set (argsort of uniforms) - what the batch-subset
        stochastic depth of n branches would draw with n x sets torch.randperm calls.  keeps (kept samples per set, the same
        for all n branches): also the n ROW MAPS [n, sum_i keeps_i T_i] int32 - compact row (set-major, kept sample, token) ->
        row of the full tensor - that let the LayerNorm / tail kernels read and write the kept rows in place."""
        if PERM_SOURCE is not None:
            self.perms = [p.to(device=device, dtype=torch.int64) for p in PERM_SOURCE(n, self)]
        else:
            self.perms = [torch.rand(n, B, device=device).argsort(dim=1) for B, _, _ in self.sets]
        self.perm_at = 0
        self.rowmaps, self.map_keeps = None, None
        if keeps is not None:
            parts = []
            for p, k, (B, T, off) in zip(self.perms, keeps, self.sets):
                parts.append(((p[:, :k] * T).unsqueeze(2) + torch.arange(T, device=device)).reshape(n, k * T) + off)
            self.rowmaps = torch.cat(parts, dim=1).to(torch.int32).contiguous()
            self.map_keeps = tuple(int(k) for k in keeps)

    def take_subset(self, keeps, device):
        """The next branch's kept samples per set and, when the maps were drawn for these keeps, its row map."""
        i = self.perm_at
        idxs = self.take_perms(keeps, device)
        if self.rowmaps is not None and i < self.rowmaps.shape[0] and tuple(int(k) for k in keeps) == self.map_keeps:
            return idxs, self.rowmaps[i]
        return idxs, None

    def take_perms(self, keeps, device):
        """The next branch's kept samples per set (idx_i = first keeps[i] entries of a random permutation)."""
        if self.perms is not None and self.perm_at < self.perms[0].shape[0]:
            i, self.perm_at = self.perm_at, self.perm_at + 1
            return [p[i, :k] for p, k in zip(self.perms, keeps)]
        return [torch.randperm(B, device=device)[:k] for k, (B, _, _) in zip(keeps, self.sets)]

    def const_row_scale(self, values, device):
        """Per-set constants -> cached [R] per-row factors."""
        key = (tuple(self.sets), tuple(float(v) for v in values), str(device))
        t = _CONST_SCALES.get(key)
        if t is None:
            if len(_CONST_SCALES) > 64:
                _CONST_SCALES.clear()
            t = _CONST_SCALES[key] = torch.cat([torch.full((B * T,), float(v), dtype=torch.float32, device=device)
                                                    for (B, T, _), v in zip(self.sets, values)])
        return t


def concat(tensors):
    """[B_i, T_i, D] tensors -> ([1, R, D] rows, Ragged)."""
    rag = Ragged([(t.shape[0], t.shape[1]) for t in tensors])
    return torch.cat([t.reshape(-1, t.shape[-1]) for t in tensors], dim=0).unsqueeze(0), rag


# ------------------------------------------------------------------------------------------------ attention over the sets
class AttnPackedRaggedFn(torch.autograd.Function):
    """AttentionD8's core (functional.AttnPackedFn) per crop set on the rows of one packed tensor: qkv [1,R,3*8c] -> o [1,R,8c]."""

    @staticmethod
    def forward(ctx, qkv, rag, H, c, scale):
        qkv = qkv if qkv.is_contiguous() else qkv.contiguous()
        o = torch.empty((1, rag.rows, 8 * c), dtype=qkv.dtype, device=qkv.device)
        lses = []
        for qv, ov in zip(rag.views(qkv), rag.views(o)):
            _, lse = ops.attn_fwd_packed(qv, H, c, scale, out=ov)
            lses.append(lse)
        ctx.save_for_backward(qkv, o, *lses)
        ctx.meta = (rag, H, c, scale)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, *lses = ctx.saved_tensors
        rag, H, c, scale = ctx.meta
        do = do if do.is_contiguous() else do.contiguous()
        dqkv = torch.empty_like(qkv)
        for qv, ov, dv, gv, lse in zip(rag.views(qkv), rag.views(o), rag.views(do), rag.views(dqkv), lses):
            ops.attn_bwd_packed(qv, ov, dv, lse, H, c, scale, out=gv)
        return dqkv, None, None, None, None


class AttnQKVRaggedFn(torch.autograd.Function):
    """functional.AttnFusedQKVFn per crop set: qkv [1,R,3*H*hd] (rows = [3,H,hd]) -> o [1,R,H*hd]; one gradient tensor."""

    @staticmethod
    def forward(ctx, qkv, rag, H, scale):
        qkv = qkv if qkv.is_contiguous() else qkv.contiguous()
        hd = qkv.shape[-1] // (3 * H)
        o = torch.empty((1, rag.rows, H * hd), dtype=qkv.dtype, device=qkv.device)
        lses = []
        for (B, T, off), qv, ov in zip(rag.sets, rag.views(qkv), rag.views(o)):
            q5 = qv.view(B, T, 3, H, hd)
            q, k, v = (q5[:, :, i].permute(0, 2, 1, 3) for i in range(3))
            o4 = ov.view(B, T, H, hd).permute(0, 2, 1, 3)
            lse = torch.empty((B, H, T), dtype=torch.float32, device=qkv.device)
            st = q.stride()
            ops.check(ops.lib().octic_attn_fwd(ops._p(q), ops._p(k), ops._p(v), ops._p(o4), ops._p(lse), B, H, T, hd, st[0], st[1],
                                               st[2], o4.stride(0), o4.stride(1), o4.stride(2), float(scale), ops._stream(qkv)))
            lses.append(lse)
        ctx.save_for_backward(qkv, o, *lses)
        ctx.meta = (rag, H, hd, scale)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, *lses = ctx.saved_tensors
        rag, H, hd, scale = ctx.meta
        do = do if do.is_contiguous() else do.contiguous()
        dqkv = torch.empty_like(qkv)
        for (B, T, off), qv, ov, dv, gv, lse in zip(rag.sets, rag.views(qkv), rag.views(o), rag.views(do), rag.views(dqkv), lses):
            q5, g5 = qv.view(B, T, 3, H, hd), gv.view(B, T, 3, H, hd)
            q, k, v = (q5[:, :, i].permute(0, 2, 1, 3) for i in range(3))
            dq, dk, dvv = (g5[:, :, i].permute(0, 2, 1, 3) for i in range(3))
            ops.attn_bwd(q, k, v, ov.view(B, T, H, hd).permute(0, 2, 1, 3), dv.view(B, T, H, hd).permute(0, 2, 1, 3), lse, scale,
                         dq, dk, dvv)
        return dqkv, None, None, None


def attn_sets_supported(rag, hd, dtype):
    return all(ops.attn_supported(T, hd, dtype) for _, T, _ in rag.sets)


# -------------------------------------------------------------------- batch-subset stochastic depth on a ragged stream
class GatherSetsFn(torch.autograd.Function):
    """xa = the kept samples of every set (idx_i into set i) as one compact ragged tensor.  Paired with ScatterSetsFn
    through `link` exactly like d8_layers._GatherRowsFn / _ScatterRowsFn: the stream's cotangent stays ONE tensor edited in place."""

    @staticmethod
    def forward(ctx, x, idxs, rag, sub, link):
        ctx.meta = (idxs, rag, sub, link)
        out = torch.empty((1, sub.rows, x.shape[-1]), dtype=x.dtype, device=x.device)
        for idx, (B, T, off), (k, _, coff) in zip(idxs, rag.sets, sub.sets):
            ops.gather_samples(x[0, off:off + B * T].view(B, T * x.shape[-1]), idx, out=out[0, coff:coff + k * T].view(k, T * x.shape[-1]))
        return out

    @staticmethod
    def backward(ctx, g_xa):
        idxs, rag, sub, link = ctx.meta
        g, link.g = link.g, None
        if g is None:
            raise RuntimeError("GatherSetsFn: the stream cotangent of the paired scatter is missing")
        g_xa = g_xa.to(g.dtype)
        g_xa = g_xa if g_xa.is_contiguous() else g_xa.contiguous()
        D = g.shape[-1]
        for idx, (B, T, off), (k, _, coff) in zip(idxs, rag.sets, sub.sets):
            ops.scatter_samples_(g[0, off:off + B * T].view(B, T * D), idx, g_xa[0, coff:coff + k * T].view(k, T * D))
        return g, None, None, None, None


class ScatterSetsFn(torch.autograd.Function):
    """x[kept samples] = out (in place; out = the compact branch result INCLUDING its residual connection)."""

    @staticmethod
    def forward(ctx, x, idxs, rag, sub, out, link):
        ctx.meta = (idxs, rag, sub, link)
        out = out if out.is_contiguous() else out.contiguous()
        D = x.shape[-1]
        for idx, (B, T, off), (k, _, coff) in zip(idxs, rag.sets, sub.sets):
            ops.scatter_samples_(x[0, off:off + B * T].view(B, T * D), idx, out[0, coff:coff + k * T].view(k, T * D))
        ctx.mark_dirty(x)
        return x

    @staticmethod
    def backward(ctx, g):
        idxs, rag, sub, link = ctx.meta
        g = g if g.is_contiguous() else g.contiguous()
        link.g = g
        D = g.shape[-1]
        gc = torch.empty((1, sub.rows, D), dtype=g.dtype, device=g.device)
        for idx, (B, T, off), (k, _, coff) in zip(idxs, rag.sets, sub.sets):
            ops.gather_samples(g[0, off:off + B * T].view(B, T * D), idx, out=gc[0, coff:coff + k * T].view(k, T * D))
        return None, None, None, None, gc, None
