"""DeiT-III-style training step on synthetic batches (the path the BASELINE metric times).

Mirrors the reference's per-iteration body (deit/engine.py:43-87) and set-up (deit/main.py:340-381 with the
recipe of experiments/train_deit.py:31-51): model.train -> bf16 autocast forward -> BCEWithLogits on multi-hot
targets -> backward (gradients all-reduced over RCCL by DDP bucket hooks, overlapped with backward) -> LAMB step
(lr 3e-3, wd 0.02) -> EMA update -> gradient norm.  Differences, all stated in DESIGN.md:
  * autocast dtype is bf16 (BASELINE.json) instead of the reference's fp16 + loss scaler;
  * the non-finite-loss check (engine.py:67-71) runs TWO steps late on a pinned host copy of the loss (`_LossWatch`):
    the host never waits for work the GPU has not finished long ago, so it can enqueue step k+1 while step k runs
    (the reference calls loss.item() and torch.cuda.synchronize() every iteration);
  * the logged gradient norm is the one LAMB computes anyway (the reference makes a second pass,
    engine.py:84).
"""
import math
import os

import torch
import torch.distributed as dist
import torch.nn as nn


# DDP gradient bucket size.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of a bucket is
# per-link bound, and the LAST bucket's all-reduce cannot overlap any backward work.  1.42 GB of f32 gradients in
# 48 MB buckets = 30 collectives issued in gradient-ready order (DDP rebuilds its buckets in that order after the
# first iteration), so the exposed tail is one 48 MB all-reduce (~0.6 ms at ring speed) instead of a 128 MB one, while
# each collective is still large enough (>> 1 MB) to run at link bandwidth rather than latency.
DDP_BUCKET_MB = 48
# DistributedDataParallel copies EVERY parameter gradient into its bucket with one launch per tensor when autograd accumulates it
# (reducer.cpp mark_variable_ready_dense; 978 launches = 4.5 ms per ViT-H step, profiles NOTES section 9) - for the hundreds of
# small tensors (biases, norm / layer-scale vectors, the 160 x 160 irrep blocks: 721 of the 978, 29 MB of the 1.42 GB) that is
# all launch cost.  Tensors of at most this many elements are kept OUT of DDP and all-reduced as ONE flat buffer after the
# backward pass (a handful of launches: concatenate, one collective, copy back); 0 = everything through DDP's buckets.
DDP_FLAT_SMALL_NUMEL = 100_000
# GradReducer's flat buckets: on a one-rank RCCL group the captured ViT-H step measured 66.0 / 65.8 / 65.8 / 66.3 ms with 48 / 96 /
# 200 / 1500 MB buckets (37 / 19 / 8 / 2 collectives; round 6) - flat between 96 and 200; 96 MB keeps the exposed tail (the last
# bucket + the misc buffer, reduced after the backward pass) at ~0.5 ms of ring time on 8 GPUs.
GRAD_BUCKET_MB = 96
DDP_GRADS_IN_BUCKETS = True     # the weight-gradient kernels write into DDP's bucket views (no per-tensor bucket copy); A/B switch


class Lamb(torch.optim.Optimizer):
    """LAMB as used by the reference recipe (timm/apex 'fusedlamb': grad-norm clipping to 1.0, bias-corrected
    Adam update + weight decay, per-tensor trust ratio).  Multi-tensor (torch._foreach) implementation."""

    def __init__(self, params, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02, max_grad_norm=1.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_grad_norm=max_grad_norm))
        self.last_grad_norm = None

    @torch.no_grad()
    def step(self):
        grads_all = [p.grad for g in self.param_groups for p in g["params"] if p.grad is not None]
        norms = torch._foreach_norm(grads_all)
        gnorm = torch.linalg.vector_norm(torch.stack(norms))
        self.last_grad_norm = gnorm
        if not bool(torch.isfinite(gnorm)):       # same guard as the fused step: a non-finite step changes nothing
            self.skipped_steps = getattr(self, "skipped_steps", 0) + 1
            return
        mg = self.param_groups[0]["max_grad_norm"]
        clip = torch.clamp(gnorm / mg, min=1.0) if mg is not None else None
        for group in self.param_groups:
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            grads = [p.grad for p in params]
            if clip is not None:
                grads = torch._foreach_div(grads, clip)
            b1, b2 = group["betas"]
            for p in params:
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
            group["step"] = group.get("step", 0) + 1
            t = group["step"]
            m = [self.state[p]["exp_avg"] for p in params]
            v = [self.state[p]["exp_avg_sq"] for p in params]
            torch._foreach_lerp_(m, grads, 1 - b1)
            torch._foreach_mul_(v, b2)
            torch._foreach_addcmul_(v, grads, grads, 1 - b2)
            bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
            denom = torch._foreach_sqrt(v)
            torch._foreach_div_(denom, math.sqrt(bc2))
            torch._foreach_add_(denom, group["eps"])
            upd = torch._foreach_div(m, denom)
            torch._foreach_div_(upd, bc1)
            wd = group["weight_decay"]
            if wd != 0:
                torch._foreach_add_(upd, params, alpha=wd)
                wn = torch._foreach_norm(params)
                un = torch._foreach_norm(upd)
                wn_t, un_t = torch.stack(wn), torch.stack(un)
                ratio = torch.where((wn_t > 0) & (un_t > 0), wn_t / un_t, torch.ones_like(wn_t))
                torch._foreach_mul_(upd, list(ratio.unbind(0)))
            torch._foreach_add_(params, upd, alpha=-group["lr"])


class FusedLamb:
    """LAMB + EMA as ONE fused multi-tensor HIP step (octic_lamb_step, csrc/lamb.hip): same math as ``Lamb`` /
    ``ModelEma`` above, 5 launches per step instead of ~100 foreach kernels, 52 B/param of HBM traffic.
    Gradients are consumed (overwritten) by the step."""

    CHUNK = 65536

    def __init__(self, param_groups, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0, ema_decay=None,
                 shadow_layers=(), prep_source=None, adam=False, ema_tensors=None):
        """shadow_layers: (nn.Linear, functional.DenseWeightCache) pairs; their weights/biases get a bf16 copy
        written by the update kernel itself, handed to the cache after every step (no per-step cast launches).
        adam=True: the same fused passes without the layer-wise trust ratio = AdamW with decoupled weight decay
        (octic_adamw_step; the DINOv2 recipe's optimizer).  ema_tensors: {id(param): tensor} - the EMA is kept IN these tensors
        (the DINOv2 teacher's parameters) instead of a flat buffer of this object; ema_decay is then the per-step momentum."""
        from . import _lib
        self._lib = _lib
        self.adam = bool(adam)
        self.lr, self.betas, self.eps, self.max_grad_norm, self.ema_decay = lr, betas, eps, max_grad_norm, ema_decay
        self.params, wds = [], []
        for g in param_groups:
            for p in g["params"]:
                if p.requires_grad:
                    self.params.append(p)
                    wds.append(float(g.get("weight_decay", 0.0)))
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FusedLamb runs on the GPU only")
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4          # keep every tensor 16-byte aligned in the flat state
        self.m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.ema = None
        self._ema_ext = None
        if ema_tensors is not None:
            self._ema_ext = [ema_tensors[id(p)] for p in self.params]
            for p, e in zip(self.params, self._ema_ext):
                if e.shape != p.shape or e.dtype != torch.float32 or not e.is_contiguous() or e.device != p.device:
                    raise ValueError("FusedLamb: an external EMA tensor must be a contiguous f32 twin of its parameter")
        elif ema_decay:
            self.ema = torch.zeros(total, dtype=torch.float32, device=dev)
            for p, o in zip(self.params, offs):
                self.ema[o:o + p.numel()].copy_(p.detach().flatten())
        self._offs = offs
        ct, co, cl, tb = [], [], [], [0]
        for i, p in enumerate(self.params):
            n = p.numel()
            for o in range(0, n, self.CHUNK):
                ct.append(i); co.append(o); cl.append(min(self.CHUNK, n - o))
            tb.append(len(ct))
        i32, i64 = torch.int32, torch.int64
        self.chunk_tensor = torch.tensor(ct, dtype=i32, device=dev)
        self.chunk_off = torch.tensor(co, dtype=i64, device=dev)
        self.chunk_len = torch.tensor(cl, dtype=i32, device=dev)
        self.tensor_chunk_begin = torch.tensor(tb, dtype=i32, device=dev)
        self.wd = torch.tensor(wds, dtype=torch.float32, device=dev)
        base = lambda t: [t.data_ptr() + 4 * o for o in offs]
        self.p_ptrs = torch.tensor([p.data_ptr() for p in self.params], dtype=i64, device=dev)
        self.m_ptrs = torch.tensor(base(self.m), dtype=i64, device=dev)
        self.v_ptrs = torch.tensor(base(self.v), dtype=i64, device=dev)
        self.e_ptrs = torch.tensor(base(self.ema), dtype=i64, device=dev) if self.ema is not None else None
        if self._ema_ext is not None:
            self.e_ptrs = torch.tensor([e.data_ptr() for e in self._ema_ext], dtype=i64, device=dev)
        self.g_ptrs = torch.zeros(len(self.params), dtype=i64, device=dev)
        index = {id(p): i for i, p in enumerate(self.params)}
        sh_ptrs, self._shadows = [0] * len(self.params), []
        from . import functional as _OF
        for item in shadow_layers:
            lin, cache = item[0], item[1]
            # a transposed copy only where the input gradient runs on csrc/dense_gemm.hip (role "qkv" -> "dqkv" routed)
            need_wt = len(item) < 3 or ("d" + item[2]) in _OF.DENSE_HIP
            w, b = lin.weight, lin.bias
            if id(w) not in index or (b is not None and id(b) not in index):
                continue
            wb = torch.empty_like(w, dtype=torch.bfloat16)
            bb = None if b is None else torch.empty_like(b, dtype=torch.bfloat16)
            sh_ptrs[index[id(w)]] = wb.data_ptr()
            if b is not None:
                sh_ptrs[index[id(b)]] = bb.data_ptr()
            self._shadows.append((lin, cache, wb, bb, need_wt))
        self.s_ptrs = torch.tensor(sh_ptrs, dtype=i64, device=dev) if self._shadows else None
        # transposed bf16 copies (operand of the hand-written input-gradient GEMMs): one batched launch per step
        self._wt, self._wt_items, self._wt_blocks = [], None, 0
        wt_layers = [sh for sh in self._shadows if sh[4]]
        self._wt = [None] * len(self._shadows)
        if wt_layers:
            import numpy as np
            dt = np.dtype([("src", "<u8"), ("wb", "<u8"), ("wt", "<u8"), ("N", "<i4"), ("K", "<i4"),
                           ("block_begin", "<i4"), ("pad", "<i4")])
            tab = np.zeros(len(wt_layers), dtype=dt)
            blocks, j = 0, 0
            for i, (lin, cache, wb, bb, need_wt) in enumerate(self._shadows):
                if not need_wt:
                    continue
                N, K = wb.shape
                wt = torch.empty((K, N), dtype=torch.bfloat16, device=dev)
                self._wt[i] = wt
                tab[j]["src"], tab[j]["wb"], tab[j]["wt"] = wb.data_ptr(), 0, wt.data_ptr()
                tab[j]["N"], tab[j]["K"], tab[j]["block_begin"] = N, K, blocks
                blocks += int(_lib.lib().octic_dense_prep_batch_blocks(N, K))
                j += 1
            self._wt_items = torch.from_numpy(tab.view(np.uint8).copy()).to(dev)
            self._wt_blocks, self._wt_count = blocks, len(wt_layers)
        self._g_key = None
        self._g_pinned, self._g_capture_host = [], None
        self.ntensors, self.nchunks = len(self.params), len(ct)
        self.ws = torch.zeros(int(_lib.lib().octic_lamb_workspace_floats(self.ntensors, self.nchunks)),
                              dtype=torch.float32, device=dev)
        self.step_count = 0
        # bias-correction step counted on the device (skipped steps do not advance it; replayable from a hipGraph)
        self.device_step = True
        # prep_source(): the functional.WeightPrep caches of the model's LinearD8 layers; once they have all been used
        # (first forward) their per-layer preparation launches are replaced by one batched launch after each step
        self._prep_source, self._prep_batch = prep_source, None

    @property
    def last_grad_norm(self):
        return self.ws[1]

    @property
    def skipped_steps(self):
        """Steps the kernels refused because the gradient norm was not finite (device counter; reading it syncs)."""
        return int(self.ws[6].item())

    def ema_state(self):
        """EMA weights as {param index: tensor view} (same order as the parameters)."""
        if self._ema_ext is not None:
            return list(self._ema_ext)
        return [self.ema[o:o + p.numel()].view_as(p) for p, o in zip(self.params, self._offs)]

    # ---- checkpoint / resume (the reference saves optimizer.state_dict() every epoch, deit/main.py:414-423) ----------
    def state_dict(self):
        """Everything a resumed run needs: first / second moments and EMA as flat f32 tensors (the layout is the
        parameter order of the param groups, each tensor padded to a multiple of 4 elements), the device-side
        bias-correction step (ws[3]) and skipped-step counter (ws[6]), and the hyper-parameters."""
        return {"m": self.m.detach().clone(), "v": self.v.detach().clone(),
                "ema": None if self.ema is None else self.ema.detach().clone(),
                "device_step": float(self.ws[3].item()), "skipped_steps": float(self.ws[6].item()),
                "step_count": int(self.step_count), "numel": [int(p.numel()) for p in self.params],
                "hyper": {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps,
                          "max_grad_norm": self.max_grad_norm, "ema_decay": self.ema_decay}}

    def load_state_dict(self, sd):
        if list(sd["numel"]) != [int(p.numel()) for p in self.params]:
            raise ValueError("FusedLamb.load_state_dict: the parameter list differs from the checkpoint's")
        if (sd["ema"] is None) != (self.ema is None):
            raise ValueError("FusedLamb.load_state_dict: EMA present on one side only")
        with torch.no_grad():
            self.m.copy_(sd["m"])
            self.v.copy_(sd["v"])
            if self.ema is not None:
                self.ema.copy_(sd["ema"])
            self.ws[3] = float(sd["device_step"])
            self.ws[6] = float(sd["skipped_steps"])
        self.step_count = int(sd["step_count"])
        h = sd.get("hyper", {})
        self.lr, self.eps = h.get("lr", self.lr), h.get("eps", self.eps)
        self.betas = tuple(h.get("betas", self.betas))
        self.max_grad_norm, self.ema_decay = h.get("max_grad_norm", self.max_grad_norm), h.get("ema_decay", self.ema_decay)

    @torch.no_grad()
    def prime(self):
        """Bring every buffer a step refreshes on the device to its post-step form WITHOUT stepping: the bf16 shadow copies
        of the standard half's weights (+ their transposed copies) and the batched LinearD8 weight preparation, adopted by
        the layers' caches.  After this, the addresses a forward pass reads its compute-dtype weights from are the ones
        every later optimizer step rewrites - what a hipGraph captured BEFORE the first step needs (segment graphs under
        DDP are captured before the data-parallel wrapper exists, i.e. before any step).  Values: the same round-to-nearest
        casts the caches would make themselves.  Needs one forward pass first (the LinearD8 caches record their operands)."""
        import ctypes
        vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.m.device).cuda_stream)
        for (lin, cache, wb, bb, _) in self._shadows:
            wb.copy_(lin.weight)
            if bb is not None:
                bb.copy_(lin.bias)
        if self._wt_items is not None:
            self._lib.check(self._lib.lib().octic_dense_prep_batch(vp(self._wt_items), self._wt_count, self._wt_blocks,
                                                                   self._lib.BF16, stream))
        for (lin, cache, wb, bb, _), wt in zip(self._shadows, self._wt):
            cache.adopt(lin.weight, lin.bias, wb, bb, torch.bfloat16, wt_copy=wt)
        if self._prep_source is not None:
            if self._prep_batch is None or self._prep_batch.stale():
                # (a second prime() - after DDP's parameter broadcast - must refill the SAME buffers: graphs captured in
                # between read their addresses)
                from .functional import PrepBatch
                self._prep_batch = PrepBatch(self._prep_source())
            self._prep_batch.run()

    def prepare_capture(self):
        """Page-locked staging for the gradient address table of a step that is about to be captured (host
        allocations are not allowed while a stream is capturing)."""
        self._g_capture_host = torch.empty(len(self.params), dtype=torch.int64).pin_memory()

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self):
        import ctypes
        grads = []
        for p in self.params:
            if p.grad is None:
                raise RuntimeError("FusedLamb: a trainable parameter received no gradient")
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = p.grad = g.float().contiguous()
            grads.append(g)
        capturing = torch.cuda.is_current_stream_capturing()
        if (self.step_count == 0 or (self.step_count & 63) == 0) and not capturing:   # the tables hold raw parameter addresses
            if tuple(p.data_ptr() for p in self.params) != tuple(self.p_ptrs.tolist()):
                raise RuntimeError("FusedLamb: a parameter was re-allocated after the optimizer was built "
                                   "(move the model to its device before constructing the optimizer)")
        key = tuple(g.data_ptr() for g in grads)
        if capturing:
            # the upload becomes a node of the graph: its source must stay valid (and unchanged) for every replay, so
            # it is a pinned buffer owned by this capture; eager steps afterwards upload their own table again
            if self._g_capture_host is None:
                raise RuntimeError("FusedLamb: call prepare_capture() before capturing a step")
            host, self._g_capture_host = self._g_capture_host, None
            host.copy_(torch.tensor(key, dtype=torch.int64))
            self._g_pinned.append(host)
            self.g_ptrs.copy_(host, non_blocking=True)
            self._g_key = None
        elif key != self._g_key:
            # gradient buffers moved (allocation pattern changed: first steps, varying batch shapes): upload the new table
            # WITHOUT blocking the host - a pageable-memory copy would wait for the whole backward pass in front of it.
            # Four pinned tables in rotation; a table is reused only after its upload has finished.
            if getattr(self, "_g_ring", None) is None:
                self._g_ring = [[torch.empty(len(key), dtype=torch.int64).pin_memory(), None] for _ in range(4)]
                self._g_ring_i = 0
            slot = self._g_ring[self._g_ring_i]
            self._g_ring_i = (self._g_ring_i + 1) % len(self._g_ring)
            if slot[1] is not None:
                slot[1].synchronize()
            slot[0].copy_(torch.tensor(key, dtype=torch.int64))
            self.g_ptrs.copy_(slot[0], non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record()
            self._g_key = key
        self.step_count += 1
        vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.m.device).cuda_stream)
        fn = self._lib.lib().octic_adamw_step if self.adam else self._lib.lib().octic_lamb_step
        from . import ops as _ops
        nparam = int(self.m.numel())
        kt = _ops.KERNEL_TIMER.start()
        self._lib.check(fn(
            vp(self.p_ptrs), vp(self.g_ptrs), vp(self.m_ptrs), vp(self.v_ptrs), vp(self.e_ptrs), vp(self.wd),
            vp(self.chunk_tensor), vp(self.chunk_off), vp(self.chunk_len), vp(self.tensor_chunk_begin),
            self.ntensors, self.nchunks, vp(self.ws), float(self.lr), float(self.betas[0]), float(self.betas[1]),
            float(self.eps), float(self.max_grad_norm or 0.0), 0 if self.device_step else self.step_count,
            float(self.ema_decay or 0.0),
            vp(self.s_ptrs), stream))
        # algorithmic bytes per parameter: gradient norm 4 + (g, m, v, p in; m, v, u out) 28 + (u, p, ema in; p, ema out) 20
        # + the bf16 copies of the standard half's weights 2 (csrc/lamb.hip)
        _ops.KERNEL_TIMER.stop(kt, "adamw_step<gradsq+stage1+stage2>" if self.adam else "lamb_step<gradsq+stage1+ratio+stage2>",
                               (54 if self.ema is not None or self._ema_ext is not None else 46) * nparam)
        # the kernels wrote the parameters behind autograd's back: advance their version counters so that every
        # cache keyed on them (the compute-dtype weight copies of functional.WeightPrep) is refreshed
        torch._C._autograd._unsafe_set_version_counter(
            tuple(self.params), tuple(p._version + 1 for p in self.params))
        if self._ema_ext is not None:                 # the EMA twins (teacher parameters) were rewritten too
            torch._C._autograd._unsafe_set_version_counter(
                tuple(self._ema_ext), tuple(e._version + 1 for e in self._ema_ext))
        if self._wt_items is not None:
            kt = _ops.KERNEL_TIMER.start()
            self._lib.check(self._lib.lib().octic_dense_prep_batch(vp(self._wt_items), self._wt_count, self._wt_blocks,
                                                                   self._lib.BF16, stream))
            _ops.KERNEL_TIMER.stop(kt, "dense_prep_batch_kernel", 4 * sum(w.numel() for w in self._wt if w is not None))
        for (lin, cache, wb, bb, _), wt in zip(self._shadows, self._wt):   # the bf16 copies are current for the new versions
            cache.adopt(lin.weight, lin.bias, wb, bb, torch.bfloat16, wt_copy=wt)
        if self._prep_source is not None:
            if self._prep_batch is None or self._prep_batch.stale():
                from .functional import PrepBatch
                self._prep_batch = PrepBatch(self._prep_source())
            kt = _ops.KERNEL_TIMER.start()
            self._prep_batch.run()
            _ops.KERNEL_TIMER.stop(kt, "linear_prep_batch_kernel", 0)


def param_groups_weight_decay(model, weight_decay, no_decay_names=()):
    """timm's rule: no weight decay for 1-D tensors, biases and the model's no_weight_decay() names."""
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if p.ndim <= 1 or name.endswith(".bias") or name in no_decay_names:
            no_decay.append(p)
        else:
            decay.append(p)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


class ModelEma:
    """Exponential moving average of the parameters (timm ModelEma, decay 0.99996: deit/main.py:344-351)."""

    def __init__(self, model, decay=0.99996):
        self.decay = decay
        self.params = [p.detach().clone() for p in model.parameters()]

    @torch.no_grad()
    def update(self, model):
        torch._foreach_lerp_(self.params, [p.detach() for p in model.parameters()], 1.0 - self.decay)


def synthetic_batch(batch, num_classes, device, seed, img_size=224):
    """samples ~ randn like the reference's own benchmark (experiments/complexity.py:70); targets = multi-hot
    (1-2 classes per row) as after Mixup + targets.gt(0) (deit/engine.py:47-54)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    samples = torch.randn(batch, 3, img_size, img_size, generator=g)
    targets = torch.zeros(batch, num_classes)
    idx = torch.randint(0, num_classes, (batch, 2), generator=g)
    targets.scatter_(1, idx, 1.0)
    return samples.to(device), targets.to(device)


def use_tuned_gemms(table=None):
    """Library GEMMs (hipBLASLt / rocBLAS) serve the standard half wherever the hand-written kernels are de-selected or
    refuse a shape (functional.DENSE_HIP / WGRAD_HIP; on the ViT-H bench path none is left since round 3).
    Which library solution is fastest per (layout, m, n, k) was searched once on an MI355X with PyTorch's TunableOp and
    is shipped as a lookup table; this only loads it (tuning itself stays off, so a step never searches).  Shapes that
    are not in the table, or a library build whose validators differ, fall back to the library default."""
    import torch.cuda.tunable as tunable
    table = table or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunable", "gfx950_vit_huge_b64.csv")
    if not os.path.exists(table):
        return False
    tunable.enable(True)
    tunable.tuning_enable(False)
    tunable.record_untuned_enable(False)
    return bool(tunable.read_file(table))


def library_gemm_layers(model):
    """(nn.Linear, DenseWeightCache, role) of every projection of the standard half (vit.Attention / vit.Mlp)."""
    from . import vit
    out = []
    for m in model.modules():
        if isinstance(m, vit.Attention):
            out += [(m.qkv, m._c1, "qkv"), (m.proj, m._c2, "proj")]
        elif isinstance(m, vit.Mlp):
            out += [(m.fc1, m._c1, "fc1"), (m.fc2, m._c2, "fc2")]
    return out


def octic_weight_preps(model):
    """The functional.WeightPrep cache of every LinearD8 of the model."""
    from .d8_layers import LinearD8
    return [m._prep for m in model.modules() if isinstance(m, LinearD8)]


class _Segment(nn.Module):
    """A tensor-in / tensor-out slice of an OcticVisionTransformer: blocks[lo:hi], plus the patch embedding in front of
    the first slice and the final norm + pooling + head behind the last one.  The blocks (and, for the first / last slice,
    the embedding parameters and the head) are the MODEL's own modules, registered here a second time only so that
    ``parameters()`` lists exactly what this slice touches - ``torch.cuda.make_graphed_callables`` captures one forward
    and one backward hipGraph per slice with those parameters as the static input surface."""

    def __init__(self, model, lo, hi, first, last):
        super().__init__()
        self.blocks = nn.ModuleList(model.blocks[lo:hi])
        self.first, self.last, self.lo, self.hi = first, last, lo, hi
        self.brk = model.octic_equi_break_layer
        object.__setattr__(self, "_model", model)       # not a sub-module (the model owns these modules)
        if first:
            self.patch_embed, self.pos_embed, self.cls_token = model.patch_embed, model.pos_embed, model.cls_token
        if lo < self.brk <= hi and model.invariant:
            self.invariantization, self.invariant_proj = model.invariantization, model.invariant_proj
        if last:
            self.norm, self.head = model.norm, model.head

    def forward(self, x):
        from .d8_layers import arm_drop_path_pool
        from .d8_utils import packed_pos_embed
        from . import functional as OF
        m = self._model
        c = m.embed_dim // 8
        # autocast without the weight cache (a captured slice must not depend on a cache another slice filled)
        with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
            arm_drop_path_pool(True)                    # every slice draws its own pool of drop-path masks
            try:
                if self.first:
                    pos = packed_pos_embed(m.pos_embed)
                    cls_row = None if m.global_pool else m._cls_row()
                    x = m.patch_embed.tokens(x, pos, cls_row)
                for i, blk in enumerate(self.blocks):
                    j = self.lo + i
                    if j < self.brk:
                        xs = blk(x if isinstance(x, OF.Octic) else OF.Octic(x, c))
                        x = xs
                        if j + 1 == self.brk:           # hand-off to the standard half (model.py:196-200)
                            if m.invariant:
                                x = m.invariant_proj(m.invariantization(xs, _out_dtype=OF.compute_dtype(xs.packed)))
                            else:
                                x = OF.HandoffCatFn.apply(xs.packed, c, xs.packed.dtype)
                    else:
                        x = blk(x)
                if isinstance(x, OF.Octic):
                    x = x.packed
                if self.last:
                    x = m.norm(x)
                    x = x.mean(dim=1) if m.global_pool else x[:, 0]
                    x = m.head(x)
            finally:
                arm_drop_path_pool(False)
        return x


class SegmentedModel(nn.Module):
    """The model as a chain of ``n_segments`` slices (``_Segment``) whose forward and backward are hipGraphs.

    Why: a captured WHOLE step (``Trainer.capture``) cannot contain DDP's bucketed all-reduce, so with N > 1 the step was
    issued eagerly - ~45 ms of Python / ctypes / autograd per step against ~66 ms of kernels (round-3 review, item 4).
    Here forward and backward are 2 x n_segments graph launches; what stays eager is what has to: the loss, autograd's
    gradient accumulation (where DDP's hooks fire, slice by slice, so the all-reduce of a slice's buckets overlaps the
    backward graphs of the slices in front of it) and the fused optimizer.  The fused residual + next-LayerNorm links
    (vit.link_blocks / d8_layers.link_octic_blocks) are cut at the slice boundaries (the carried norm is not a tensor in
    the slice's signature): one extra LayerNorm pass per boundary; the eager run of a SegmentedModel has the same links,
    so graphed and eager steps are bitwise equal."""

    def __init__(self, model, n_segments=8):
        super().__init__()
        from .vit import link_blocks
        from .d8_layers import link_octic_blocks
        self.model = model
        nb, brk = len(model.blocks), model.octic_equi_break_layer
        if brk < 1:
            # (model._forward_features hands the lifted tokens to the standard half before any block; the slices only do
            # the hand-off behind an octic block)
            raise ValueError("SegmentedModel: octic_equi_break_layer must be >= 1 (a model without octic blocks has no "
                             "octic -> standard hand-off inside a slice)")
        if model.dropout_rate:
            raise ValueError("SegmentedModel: dropout in front of the head (dropout_rate > 0) is not part of a slice")
        n_segments = max(1, min(int(n_segments), nb))
        # boundaries: evenly spaced, and the octic / standard hand-off is always one of them
        cuts = sorted(set([0, nb] + [round(i * nb / n_segments) for i in range(1, n_segments)] + ([brk] if 0 < brk < nb else [])))
        segs = [_Segment(model, lo, hi, lo == 0, hi == nb) for lo, hi in zip(cuts[:-1], cuts[1:])]
        object.__setattr__(self, "_segments", segs)     # share the model's modules: not registered twice under this module
        for b in model.blocks:                          # re-link inside the slices only
            if hasattr(b, "_next_norm"):
                del b._next_norm
        for sg in segs:
            link_octic_blocks([b for j, b in zip(range(sg.lo, sg.hi), sg.blocks) if j < brk])
            link_blocks([b for j, b in zip(range(sg.lo, sg.hi), sg.blocks) if j >= brk])
        self.graphed = False

    def no_weight_decay(self):
        return {"model." + n for n in self.model.no_weight_decay()} | set(self.model.no_weight_decay())

    def train(self, mode=True):
        super().train(mode)
        for sg in self._segments:                       # (a graphed slice replays its graphs only in the mode it was captured in)
            sg.training = mode
        return self

    def forward(self, x):
        for sg in self._segments:
            x = sg(x)
        return x

    def capture(self, samples, warmup=3):
        """Replace every slice's forward / backward by hipGraph replays (static shapes: those of ``samples``)."""
        if self.graphed:
            return self
        self.train()
        for sg in self._segments:
            sg.train()
        args, x = [], samples
        with torch.no_grad():
            for sg in self._segments:
                args.append((x.detach().clone().requires_grad_(not sg.first),))
                x = sg(x)
        torch.cuda.synchronize()
        graphed = torch.cuda.make_graphed_callables(tuple(self._segments), tuple(args), num_warmup_iters=warmup)
        object.__setattr__(self, "_segments", list(graphed))
        self.graphed = True
        return self


def ddp_ignore_small(module, numel=None):
    """Put the trainable tensors of `module` with at most `numel` elements (default DDP_FLAT_SMALL_NUMEL) on
    DistributedDataParallel's ignore list - call BEFORE wrapping - and return them ([] when that would leave DDP nothing)."""
    numel = DDP_FLAT_SMALL_NUMEL if numel is None else numel
    if numel <= 0:
        return []
    small = [p for p in module.parameters() if p.requires_grad and p.numel() <= numel]
    big = sum(1 for p in module.parameters() if p.requires_grad) - len(small)
    if not small or big <= 0:
        return []
    ids = {id(p) for p in small}
    # every name a small parameter can be reached by (a module registered twice - train._Segment - has two)
    names = [f"{mn}.{pn}" if mn else pn for mn, mod in module.named_modules(remove_duplicate=False)
             for pn, p in mod.named_parameters(recurse=False) if id(p) in ids]
    nn.parallel.DistributedDataParallel._set_params_and_buffers_to_ignore_for_model(module, names)
    return small


class FlatGradReducer:
    """Parameters outside DistributedDataParallel: rank 0's values brought over once, gradients averaged over the ranks through
    ONE persistent flat buffer after the backward pass (two multi-tensor copies + one collective)."""

    def __init__(self, params):
        self.params = list(params)
        self._key = self._flat = self._views = None

    @torch.no_grad()
    def broadcast(self):
        if not self.params or dist.get_world_size() == 1:
            return
        flat = torch.cat([p.detach().reshape(-1) for p in self.params])
        dist.broadcast(flat, 0)
        torch._foreach_copy_([p.data for p in self.params],
                             [t.view_as(p) for t, p in zip(flat.split([p.numel() for p in self.params]), self.params)])

    @torch.no_grad()
    def reduce(self, proxy=None):
        world = dist.get_world_size()
        if not self.params or (world == 1 and proxy is None):
            return                                      # one rank: the gradients are final as they are
        live = [p for p in self.params if p.grad is not None]
        if not live:
            return
        key = tuple(id(p) for p in live)
        if self._key != key:                            # (built once: the set of tensors with gradients does not change)
            n = sum(p.numel() for p in live)
            self._flat = torch.empty(n, dtype=live[0].grad.dtype, device=live[0].grad.device)
            self._views = [t.view_as(p) for t, p in zip(self._flat.split([p.numel() for p in live]), live)]
            self._key = key
        grads = [p.grad for p in live]
        torch._foreach_copy_(self._views, grads)
        if world > 1:
            self._flat.div_(world)
            dist.all_reduce(self._flat)
        else:
            proxy.traffic(self._flat)                   # the 1-GPU stand-in for the collective (DdpTrafficProxy)
        torch._foreach_copy_(grads, self._views)


class GradReducer:
    """Gradient averaging over the ranks WITHOUT DistributedDataParallel - for the one-micro-batch step of ``Trainer``, eager or
    captured (deit/main.py:354-359 wraps the model in DDP; deit/engine.py:62-75 is the backward it hooks into).

    Why not DDP: its reducer copies / scales gradients from autograd hooks (one launch per tensor), so nothing may write a
    parameter gradient late - the batched finishes and the paired weight gradients of the one-GPU step are off under it - and a
    step that contains it cannot be one hipGraph.  Here the large gradients (more than ``small_numel`` elements) are WRITTEN by
    their kernels into flat f32 buckets (``ops.GRAD_DEST`` protocol: ``get`` hands out the bucket view, ``written`` is called
    once the launch that fills it is on the stream), filled in gradient-ready order (reverse registration order) up to
    ``bucket_mb``; a bucket whose last tensor has been written is all-reduced at once on the process group's own stream
    (``async_op``: RCCL kernels beside the rest of the backward pass).  Everything else - vectors, the small irrep blocks,
    whatever did not honour its destination - is gathered into the buckets / one "misc" buffer by multi-tensor copies at the end
    of the pass.  ``finish`` leaves every ``.grad`` a view of reduced memory and makes the compute stream wait for the
    collectives: the optimizer reads the views.  No autograd hooks, no per-parameter launches, nothing host-synchronous: the
    whole sequence is capturable (``Trainer.capture``; torch's NCCL = RCCL process group records its collectives into the
    capturing graph) and, eagerly, costs the host ~30 collective calls per step.
    Averaging: ``ReduceOp.AVG`` where the backend has it (RCCL), else pre-divided sums (gloo: the CPU tests)."""

    def __init__(self, params, bucket_mb=None, small_numel=None, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.avg = dist.get_backend(group) == "nccl"
        params = [p for p in params if p.requires_grad]
        if not params:
            raise ValueError("GradReducer: no trainable parameter")
        dev = params[0].device
        cap = int((bucket_mb or GRAD_BUCKET_MB) * (1 << 20) // 4)
        small_numel = DDP_FLAT_SMALL_NUMEL if small_numel is None else small_numel
        big = [p for p in reversed(params) if p.numel() > small_numel and p.dtype == torch.float32]
        big_ids = {id(p) for p in big}
        self.small = [p for p in params if id(p) not in big_ids]
        groups, cur, n = [], [], 0
        for p in big:
            if cur and n + p.numel() > cap:
                groups.append(cur)
                cur, n = [], 0
            cur.append(p)
            n += (p.numel() + 3) // 4 * 4
        if cur:
            groups.append(cur)
        self.buckets = []                               # [flat, [(param, view)], pending]
        self._dest, self._bucket_of = {}, {}
        for k, g in enumerate(groups):
            flat = torch.zeros(sum((p.numel() + 3) // 4 * 4 for p in g), dtype=torch.float32, device=dev)
            o, pv = 0, []
            for p in g:
                v = flat[o:o + p.numel()].view_as(p)
                pv.append((p, v))
                self._dest[p.data_ptr()] = v
                self._bucket_of[p.data_ptr()] = k
                o += (p.numel() + 3) // 4 * 4
            self.buckets.append([flat, pv, 0])
        self.big = big
        self._misc = self._misc_views = self._misc_key = None
        self._works, self._fired, self._seen = [], set(), set()
        self._captured_works = []
        self.capturing = False                          # set by Trainer.capture around the captured pass
        self.active = False
        self._hold = False
        self.early = 0                                  # buckets of the last pass whose collective was issued before its end

    # ---- the ops.GRAD_DEST protocol ----------------------------------------------------------------------------------
    def get(self, ptr):
        return self._dest.get(ptr) if self.active else None

    def written(self, ptr):
        k = self._bucket_of.get(ptr)
        if k is None or not self.active or self._hold or ptr in self._seen:
            return
        self._seen.add(ptr)
        b = self.buckets[k]
        b[2] -= 1
        if b[2] == 0:
            self._fire(k)

    # ---- one backward pass ---------------------------------------------------------------------------------------------
    def begin(self, hold=False):
        """Call with every .grad None, right before backward(); installs this object as ops.GRAD_DEST.
        hold: more micro-batches follow (gradient accumulation) - gradients still land in the buckets, but nothing is reduced
        before ``reduce_all``."""
        from . import ops
        self._works, self._fired, self._seen = [], set(), set()
        for b in self.buckets:
            b[2] = len(b[1])
        self.active = True
        self._hold = bool(hold)
        self._prev_dest, ops.GRAD_DEST = ops.GRAD_DEST, self

    def _fire(self, k):
        if k in self._fired:
            return
        if self.capturing and not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            # Inside Trainer.capture, on a thread / stream that is not (yet) part of the capture - autograd runs a node on the
            # stream it was created on, and an AccumulateGrad node kept alive from an eager iteration is bound to the eager
            # stream.  A collective issued from here would count as an EAGER one (its work object goes to the process group's
            # watchdog thread) while the group's stream is capturing: the watchdog's poll of its end event then fails with
            # hipErrorCapturedEvent and terminates the process (seen once in ~30 captured runs, round 6).  Leave the bucket to
            # reduce_all(), which runs on the capturing stream.
            return
        self._fired.add(k)
        flat = self.buckets[k][0] if k >= 0 else self._misc
        if not self.avg and self.world > 1:
            flat.div_(self.world)
        self._works.append(dist.all_reduce(flat, op=dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM,
                                           group=self.group, async_op=True))

    def abort(self):
        """backward() raised: drop the registry without touching the gradients (collectives already issued are waited for)."""
        from . import ops
        if self.active:
            ops.GRAD_DEST = self._prev_dest
            self.active = False
        for w in self._works:
            w.wait()
        self._works = []

    @torch.no_grad()
    def broadcast(self, params=None):
        """Rank 0's parameter values to every rank (what DDP's constructor does), as one flat buffer per dtype."""
        ps = [p for p in (params if params is not None else self.big + self.small)]
        if self.world == 1 or not ps:
            return
        for dt in {p.dtype for p in ps}:
            sel = [p for p in ps if p.dtype == dt]
            flat = torch.cat([p.detach().reshape(-1) for p in sel])
            dist.broadcast(flat, 0, group=self.group)
            torch._foreach_copy_([p.data for p in sel],
                                 [t.view_as(p) for t, p in zip(flat.split([p.numel() for p in sel]), sel)])

    @torch.no_grad()
    def settle(self):
        """After the (first) backward(): gather what was not written in place and re-point every .grad at bucket / misc memory -
        no collective.  With gradient accumulation the later micro-batches then ADD into those views (autograd accumulates in
        place into an existing .grad)."""
        from . import ops
        ops.GRAD_DEST = self._prev_dest
        self.active = False
        self.early = len(self._fired)
        # large gradients that did not land in their bucket (a route without a destination: library GEMM, CPU, f32 paths)
        src, dst = [], []
        for k, (flat, pv, _pending) in enumerate(self.buckets):
            for p, v in pv:
                g = p.grad
                if g is None:
                    # a parameter that took no part in this pass: its slot carries zeros through the collective (the same on
                    # every rank) and its .grad STAYS None - optimizers skip such tensors (no weight decay on them either)
                    v.zero_()
                    continue
                elif g.data_ptr() != v.data_ptr():
                    if k in self._fired:
                        raise RuntimeError("GradReducer: a bucket was reduced before one of its gradients was written")
                    src.append(g)
                    dst.append(v)
                p.grad = v
        if src:
            torch._foreach_copy_(dst, src)
        live = [p for p in self.small if p.grad is not None]
        self._misc_live = bool(live)
        if live:
            key = tuple(id(p) for p in live)
            if self._misc_key != key:                   # (built once: the set of tensors with gradients does not change)
                n = sum(p.numel() for p in live)
                self._misc = torch.empty(n, dtype=torch.float32, device=live[0].device)
                self._misc_views = [t.view_as(p) for t, p in zip(self._misc.split([p.numel() for p in live]), live)]
                self._misc_key = key
            torch._foreach_copy_(self._misc_views, [p.grad for p in live])
            for p, v in zip(live, self._misc_views):
                p.grad = v

    @torch.no_grad()
    def reduce_all(self):
        """Reduce what has not been reduced yet and make the current stream wait for every collective of the pass."""
        for k in range(len(self.buckets)):
            self._fire(k)
        if self._misc_live:
            self._fire(-1)
        if len(self._fired) != len(self.buckets) + (1 if self._misc_live else 0):
            raise RuntimeError("GradReducer.reduce_all: called inside Trainer.capture from a stream that is not capturing")
        for w in self._works:
            w.wait()
        if self._works and self._works[0] is not None and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            # Work objects made under capture are kept for the life of this reducer.  Dropping them returns their HIP events
            # to ProcessGroupNCCL's event cache; the next EAGER collective re-records such an event, the group's watchdog
            # thread polls it - and HIP still answers "operation not permitted on an event last recorded in a capturing
            # stream" (hipErrorCapturedEvent), which terminates the process from the watchdog thread.  Seen once in ~30
            # captured runs in round 6, only in flows with eager steps after a capture.
            self._captured_works.extend(self._works)
        self._works = []

    def finish(self):
        """After backward() of a one-micro-batch step: ``settle`` + ``reduce_all`` - every .grad a view of reduced memory, the
        current stream behind the collectives."""
        self.settle()
        self.reduce_all()

class DdpTrafficProxy:
    """A one-GPU stand-in for what the gradient all-reduce of an N-GPU step does to the step (round-4 review item 5): a DDP
    communication hook that, for every gradient bucket that becomes ready during the backward pass, moves the bucket's bytes
    device-to-device on a SIDE stream (`copies` times; a ring all-reduce over 8 GPUs reads and writes about 1.75 x the bucket
    on each GPU) and reports the bucket as reduced.  The optimizer waits for the side stream (`join`), as it would for the last
    bucket's collective.  This prices two things a single GPU can measure - HBM / fabric contention between a concurrent
    byte-moving stream and the backward kernels (the dense GEMMs re-read their operand panels 3.4-3.6 x through the fabric),
    and the exposed tail of the last bucket - and nothing else: no xGMI link, no RCCL kernel occupancy, no cross-rank skew.
    ``copies = 0`` gives the no-traffic baseline with the same hooks."""

    def __init__(self, device, copies=1, halve=False):
        self.stream = torch.cuda.Stream(device)
        self.copies, self.halve = int(copies), bool(halve)
        self.scratch = None
        self.bytes = 0                                  # bytes read (= bytes written) by the side stream so far

    def traffic(self, buf, halve=False):
        """Move `buf`'s bytes `copies` times on the side stream (what a collective over it would read and write)."""
        n = buf.numel() // 2 if halve else buf.numel()
        if self.copies > 0 and n > 0:
            if self.scratch is None or self.scratch.numel() < n:
                self.scratch = torch.empty(n, dtype=buf.dtype, device=buf.device)
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                for _ in range(self.copies):
                    self.scratch[:n].copy_(buf[:n], non_blocking=True)
            self.bytes += n * buf.element_size() * self.copies

    def hook(self, state, bucket):
        buf = bucket.buffer()
        self.traffic(buf, self.halve)                   # bf16 buckets: half the payload
        fut = torch.futures.Future()
        fut.set_result(buf)
        return fut

    def join(self):
        torch.cuda.current_stream().wait_stream(self.stream)


class _LossWatch:
    """The non-finite-loss check of deit/engine.py:67-71 without a per-step stream drain.

    `loss.item()` on the previous step's loss waits for everything queued before it - in graph mode the whole previous
    replay - so the host could never enqueue step k+1 while step k runs (round-3 review: `host_issue_ms_per_step` read
    65 ms for a 0.5 ms graph launch).  Here every step ends with an asynchronous copy of its loss into one of `lag + 1`
    pinned host slots plus an event; the check of step k happens when step k + lag is about to be issued (lag 2): step k
    finished a whole step ago and step k+1 is keeping the GPU busy, so waiting on k's event costs nothing.  The optimizer
    kernels refuse a non-finite step on their own, so the weights are intact when this raises.  On the CPU the loss is
    read directly (nothing is asynchronous there)."""

    def __init__(self, device_type, lag=2):
        self.cuda = device_type == "cuda"
        self.lag = lag
        self.n = 0
        self.wait_s = 0.0                               # host time spent waiting for a two-steps-old event (throttle, not issue cost)
        if self.cuda:
            self.host = torch.zeros(lag + 1, dtype=torch.float32).pin_memory()
            self.events = [torch.cuda.Event() for _ in range(lag + 1)]
        else:
            self.host = torch.zeros(lag + 1, dtype=torch.float32)

    def push(self, loss):
        """Call after a step has been issued; `loss` is its (device-resident) scalar loss."""
        slot = self.n % (self.lag + 1)
        if self.cuda:
            self.host[slot:slot + 1].copy_(loss.detach().reshape(1).float(), non_blocking=True)
            self.events[slot].record()
        else:
            self.host[slot] = float(loss.detach())
        self.n += 1

    def check(self, block=True):
        """Call before issuing a step: raises if the loss of the step `lag` steps back was not finite."""
        k = self.n - self.lag
        if k < 0:
            return
        slot = k % (self.lag + 1)
        if self.cuda:
            if block:
                if not self.events[slot].query():
                    import time
                    t0 = time.perf_counter()
                    self.events[slot].synchronize()
                    self.wait_s += time.perf_counter() - t0
            elif not self.events[slot].query():
                return
        if not math.isfinite(float(self.host[slot])):
            raise FloatingPointError("Loss is not finite, stopping training")

    def drain(self):
        """Check every loss that has not been looked at yet (end of an epoch / before a checkpoint)."""
        if self.cuda:
            torch.cuda.current_stream().synchronize()
        for k in range(max(0, self.n - self.lag), self.n):
            if not math.isfinite(float(self.host[k % (self.lag + 1)])):
                raise FloatingPointError("Loss is not finite, stopping training")


class Trainer:
    """One DeiT-III iteration (deit/engine.py:43-87).  ``accum_steps`` micro-batches per optimizer step express the
    reference's global batch on fewer GPUs (configs[2]: 2048 = 8 GPUs x 64 x 4 micro-batches; the reference ran
    32 GPUs x 64, experiments/train_deit.py:7-12,66): gradients of the micro-batch mean losses are averaged, the
    all-reduce runs once, in the last micro-batch's backward (``no_sync`` before).  ``opt_eps`` is the recipe's
    ``--opt-eps`` (deit/main.py:68, default 1e-8, which timm forwards to apex FusedLAMB).  ``bf16_buckets`` halves
    the all-reduce payload (DDP's bf16 compression hook: buckets are cast to bf16, summed, cast back)."""

    def __init__(self, model, lr=3e-3, weight_decay=0.02, ema_decay=0.99996, distributed=False, local_rank=0,
                 fused_optimizer=True, tuned_gemms=True, opt_eps=1e-8, accum_steps=1, bf16_buckets=False,
                 autocast=True, check_every=1, device_type=None, bucket_cap_mb=None, segment_graphs=0, ddp_proxy=None,
                 own_reducer=None):
        """segment_graphs = n > 0 (GPU only): forward and backward run as 2 n hipGraph replays (``SegmentedModel``), the
        loss, the gradient all-reduce hooks and the optimizer stay eager - the cheap-on-the-host step for gradient
        accumulation, where the whole-step graph of ``capture`` does not apply.
        own_reducer (distributed only): average the gradients with ``GradReducer`` instead of DistributedDataParallel - the
        step keeps its batched finishes and paired weight gradients and can be captured as ONE hipGraph with the RCCL
        collectives inside (``capture``), gradient accumulation included (the first micro-batch lands in the buckets, the
        others add to them, the collectives follow the last one).  None = wherever it applies: f32 buckets, f32 master
        parameters, no segment graphs; everything else goes through DistributedDataParallel as before."""
        self.raw_model = model
        self.segmented = None
        if segment_graphs:
            self.segmented = SegmentedModel(model, segment_graphs)
        self.tuned_gemms = use_tuned_gemms() if (tuned_gemms and torch.cuda.is_available()) else False
        self.model = self.segmented if self.segmented is not None else model
        self.accum_steps = int(accum_steps)
        self.device_type = device_type or next(model.parameters()).device.type
        self.autocast = autocast
        self.check_every = max(1, int(check_every))
        self._steps = 0
        self._ddp_args = None
        self._proxy = ddp_proxy                         # a DdpTrafficProxy: its hook replaces the all-reduce (measurement only)
        self._reducer = None
        if distributed:
            can_own = (self.segmented is None and not bf16_buckets and ddp_proxy is None
                       and all(p.dtype == torch.float32 for p in model.parameters() if p.requires_grad))
            if own_reducer and not can_own:
                raise ValueError("Trainer(own_reducer=True) needs f32 buckets, f32 master parameters and no segment graphs")
            if can_own if own_reducer is None else own_reducer:
                self._reducer = GradReducer(list(model.parameters()), bucket_mb=bucket_cap_mb)
                self._reducer.broadcast(list(model.parameters()) + list(model.buffers()))
                distributed = False                     # (no DistributedDataParallel wrapper below)
        if distributed:
            self._ddp_args = (local_rank, bucket_cap_mb, bf16_buckets)
            # With segment graphs the slices are captured FIRST and wrapped in DistributedDataParallel afterwards
            # (capture_segments): capturing the backward graphs of parameters that a DDP reducer already manages crashed
            # inside torch.autograd.grad (segfault on ROCm 7 / torch 2.10); the documented order is graphs, then DDP.
            if self.segmented is None:
                self._wrap_ddp()
        groups = param_groups_weight_decay(model, weight_decay, model.no_weight_decay())
        if fused_optimizer:
            # LAMB and EMA in one fused step; it also refreshes the bf16 weights of the library-GEMM layers
            self.optimizer = FusedLamb(groups, lr=lr, eps=opt_eps, ema_decay=ema_decay,
                                       shadow_layers=library_gemm_layers(model),
                                       prep_source=lambda: octic_weight_preps(model))
            self.ema = None
        else:
            self.optimizer = Lamb(groups, lr=lr, eps=opt_eps, weight_decay=weight_decay)
            self.ema = ModelEma(model, ema_decay) if ema_decay else None
        self.criterion = nn.BCEWithLogitsLoss()
        self._watch = _LossWatch(self.device_type)

    def _wrap_ddp(self):
        local_rank, bucket_cap_mb, bf16_buckets = self._ddp_args
        # every trainable parameter is used each step (frozen cls_token.1-4 are not registered for grads)
        self._small = ddp_ignore_small(self.model)
        self._flat_reducer = FlatGradReducer(self._small)
        self._flat_reducer.broadcast()          # DDP broadcasts rank 0's values of the parameters IT manages only
        self.model = nn.parallel.DistributedDataParallel(
            self.model, device_ids=[local_rank] if self.device_type == "cuda" else None,
            bucket_cap_mb=bucket_cap_mb or DDP_BUCKET_MB,
            gradient_as_bucket_view=True, find_unused_parameters=False, static_graph=False)
        from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
        if self._proxy is not None:
            self.model.register_comm_hook(None, self._proxy.hook)
        elif bf16_buckets:
            self.model.register_comm_hook(None, default_hooks.bf16_compress_hook)
        else:
            # the reducer's own path divides every gradient that already lives in its bucket by the world size with one launch
            # per tensor; with a communication hook the division is one launch per bucket.  The built-in (C++) hook: no Python
            # call per bucket on the autograd thread
            try:
                self.model._register_builtin_comm_hook(dist.BuiltinCommHookType.ALLREDUCE)
            except Exception:
                self.model.register_comm_hook(None, default_hooks.allreduce_hook)
        # Weight gradients written straight into DDP's bucket views (ops.GRAD_DEST): the views a step's backward leaves in
        # .grad are where the next step's weight-gradient kernels write - the reducer sees an alias and skips its copy.
        # Eager single-micro-batch steps on the GPU only (accumulation adds into .grad; graphed slices own their outputs).
        self._grad_dest = {} if (DDP_GRADS_IN_BUCKETS and self.device_type == "cuda" and self.accum_steps == 1
                                 and self.segmented is None) else None
        small_ids = {id(p) for p in self._small}
        self._ddp_params = [p for p in self.raw_model.parameters() if p.requires_grad and id(p) not in small_ids]
        # every parameter whose gradient comes out of a "finish" reduction (vectors: norm, layer scale, bias) is outside DDP
        self._vectors_outside_ddp = bool(self._small) and all(
            id(p) in small_ids for p in self.raw_model.parameters() if p.requires_grad and p.dim() <= 1)

    def _reduce_small_grads(self):
        """The gradients DDP does not manage (DDP_FLAT_SMALL_NUMEL): mean over the ranks through ONE flat buffer."""
        r = getattr(self, "_flat_reducer", None)
        if r is not None:
            r.reduce(self._proxy)

    def capture_segments(self, samples, warmup=3):
        """hipGraphs for the slices of a ``segment_graphs`` trainer (call once, with ONE micro-batch of the training
        shape).  A distributed trainer must call this before its first step: the data-parallel wrapper is built here,
        around the graphed slices."""
        if self.segmented is None:
            raise RuntimeError("Trainer.capture_segments: construct the trainer with segment_graphs=n")
        from . import d8_layers as _L
        if _L.COMPACT_DROP_PATH:
            raise RuntimeError("Trainer.capture_segments: d8_layers.COMPACT_DROP_PATH changes the launch shapes from step to "
                               "step; graphed slices need static shapes")
        if not isinstance(self.optimizer, FusedLamb):
            # The graphs hold no weight casts: they read the static bf16 / prepared copies that only FusedLamb.step rewrites
            # in place.  With another optimizer the replays would use the capture-time weights for ever.
            raise RuntimeError("Trainer.capture_segments needs fused_optimizer=True: graphed slices read the compute-dtype "
                               "weight copies that the fused optimizer step refreshes in place")
        if self.optimizer.step_count == 0:
            # no step yet: one forward (the weight caches record their operands), then the static weight buffers
            self.segmented.train()
            with torch.no_grad():
                self.segmented(samples)
            self.optimizer.prime()
        self.segmented.capture(samples, warmup=warmup)
        if self._ddp_args is not None and self.model is self.segmented:
            self._wrap_ddp()
            if self.optimizer.step_count == 0:
                # the wrapper has just broadcast rank 0's parameters: ranks whose initial weights differed must not keep
                # compute-dtype copies of their pre-broadcast values for the first step
                self.optimizer.prime()
        return self

    def finish(self):
        """Look at every loss the two-steps-late watch has not checked yet (raises FloatingPointError like the reference's
        per-iteration check, deit/engine.py:67-71).  Call at the end of an epoch and before writing a checkpoint."""
        self._watch.drain()

    def _forward_loss(self, samples, targets):
        if self.autocast:
            with torch.autocast(self.device_type, dtype=torch.bfloat16):
                outputs = self.model(samples)
                return self.criterion(outputs.float(), targets)
        return self.criterion(self.model(samples).float(), targets)

    def _batched_finishes(self, first_of_many=False):
        """The parameter-gradient slab reductions of the backward pass as one batched launch at its end
        (ops._DeferredFinishes) - only where nothing can read those gradients earlier: one micro-batch into .grad = None,
        no DDP bucket hooks, no graphed slices.  first_of_many: the FIRST micro-batch of an accumulated step under the own
        reducer also qualifies (every .grad is None; the later ones add into existing gradients and do not)."""
        from . import ops
        safe = (BATCHED_FINISHES and self.device_type == "cuda" and self.model is self.raw_model
                and (self.accum_steps == 1 or first_of_many))
        return _FinishScope(ops.DEFERRED_FINISHES, safe)

    def _reduced_backward(self, loss, hold=False):
        """backward() with the gradients landing in GradReducer's buckets, bucket all-reduces issued as they fill, and the
        same batched finishes / paired weight gradients as the one-GPU step (nothing reads a gradient before ``finish``)."""
        r = self._reducer
        r.begin(hold=hold)
        try:
            with self._batched_finishes(first_of_many=hold):
                loss.backward()
        except BaseException:
            r.abort()
            raise
        if hold:
            r.settle()
        else:
            r.finish()

    def _accumulate(self, samples, targets):
        """k micro-batches into one gradient (mean of the micro-batch mean losses).  Own reducer: the first backward writes into
        the buckets, the others add to them in place, ONE round of collectives follows the last; DDP: ``no_sync`` before the last."""
        k = self.accum_steps
        xs, ys = samples.chunk(k), targets.chunk(k)
        loss = None
        for i, (x, y) in enumerate(zip(xs, ys)):
            last = i == len(xs) - 1
            ctx = self.model.no_sync() if (hasattr(self.model, "no_sync") and not last) else _null()
            with ctx:
                li = self._forward_loss(x, y) / len(xs)
                if self._reducer is not None and i == 0:
                    self._reduced_backward(li, hold=True)
                elif i == 0:
                    with self._batched_finishes(first_of_many=True):   # (every .grad is None: as in a one-micro-batch step)
                        li.backward()
                else:
                    li.backward()
            loss = li.detach() if loss is None else loss + li.detach()
        if self._reducer is not None:
            self._reducer.reduce_all()
        return loss

    def step(self, samples, targets):
        if self._ddp_args is not None and self.model is self.segmented:
            raise RuntimeError("Trainer(distributed=True, segment_graphs=n): call capture_segments(micro_batch) before the "
                               "first step (the slices are captured, then wrapped in DistributedDataParallel)")
        self.model.train()
        # engine.py:67-71 two steps late on a pinned host copy (_LossWatch): no stream drain per step
        self._steps += 1
        if self._steps % self.check_every == 0:
            self._watch.check()
        k = self.accum_steps
        if k > 1 and self.segmented is not None and self.segmented.graphed:
            # Graphed slices hand autograd the SAME static gradient buffers every backward: with .grad = None the first
            # micro-batch's gradient would be adopted by reference and overwritten by the second replay before it is added.
            # Accumulate into gradients of our own instead (zeroed in place, one multi-tensor launch).
            params = [p for p in self.raw_model.parameters() if p.requires_grad]
            for p in params:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            torch._foreach_zero_([p.grad for p in params])
        else:
            self.optimizer.zero_grad(set_to_none=True)
        if k == 1:
            loss = self._forward_loss(samples, targets)
            dest = getattr(self, "_grad_dest", None)
            if dest is not None:
                from . import ops as _ops
                _ops.GRAD_DEST = dest
                try:
                    # vector gradients (all outside DDP) may still be finished in batches at the end of the pass
                    with _FinishScope(_ops.DEFERRED_FINISHES, BATCHED_FINISHES and self._vectors_outside_ddp, vectors_only=True):
                        loss.backward()
                finally:
                    _ops.GRAD_DEST = None
                for p in self._ddp_params:              # (bucket views: DDP re-points .grad at them in its hooks)
                    if p.grad is not None:
                        dest[p.data_ptr()] = p.grad
            elif self._reducer is not None:
                self._reduced_backward(loss)
            else:
                with self._batched_finishes():
                    loss.backward()
        else:
            loss = self._accumulate(samples, targets)
        self._reduce_small_grads()
        if self._proxy is not None:
            self._proxy.join()                          # the optimizer reads the "reduced" buckets
        self.optimizer.step()
        if self.ema is not None:
            self.ema.update(self.raw_model)
        self._watch.push(loss)
        # DETACHED: a caller that keeps the returned loss must not keep this iteration's autograd graph - and with it the
        # AccumulateGrad nodes bound to this iteration's stream - alive: a later Trainer.capture on another stream then re-uses
        # those nodes across streams ("AccumulateGrad node's stream does not match") and hipGraphInstantiate crashed (round 6:
        # bench.py held the last loss of its eager data-parallel measurement while capturing)
        return loss.detach()

    # ---- the whole iteration as one hipGraph -----------------------------------------------------------------------
    def capture(self, samples, targets, warmup=3):
        """Record forward + backward + optimizer of ``step`` once (static shapes) and return a ``GraphedStep`` whose
        ``replay(samples, targets)`` re-launches it with one host call: the ~3000 kernel launches of a ViT-H step
        cost the host nothing any more.  ``accum_steps`` = k: the k forward / backward passes over the chunks of the static batch
        and the one optimizer step are ONE graph.  Data parallel (``own_reducer``): the bucket all-reduces are recorded into the same
        graph on the process group's stream (fork at the event of the bucket's last write, join in front of the optimizer),
        so every rank replays ONE graph per step and the collectives overlap the backward kernels exactly as captured.  Everything the step needs is already device-side: the bias-correction step
        and the non-finite-gradient guard live in the optimizer kernels, the drop-path masks come from the device
        generator (advanced per replay by torch's graph support), the prepared weights are refreshed by the captured
        launches themselves.  ``warmup`` eager steps run first (they also build every lazily created cache).
        Under DistributedDataParallel (accumulation, bf16 buckets) the step is not capturable."""
        if self.device_type != "cuda" or not isinstance(self.optimizer, FusedLamb):
            raise RuntimeError("Trainer.capture needs the GPU and the fused optimizer")
        if self.model is not self.raw_model:
            raise RuntimeError("Trainer.capture: DistributedDataParallel / segmented steps are not captured as one graph "
                               "(own_reducer=True, or segment_graphs=n + capture_segments)")
        from . import d8_layers as _L
        if _L.COMPACT_DROP_PATH:
            raise RuntimeError("Trainer.capture: d8_layers.COMPACT_DROP_PATH launches every branch on the samples its mask "
                               "keeps - the shapes change from step to step and cannot be one captured graph; run eagerly")
        from . import ops
        if ops.KERNEL_TIMER.on:
            raise RuntimeError("Trainer.capture: disable the kernel timer first")
        sx, sy = samples.clone(), targets.clone()
        for _ in range(max(1, warmup)):
            self.step(sx, sy)
        torch.cuda.synchronize()
        self.model.train()
        graph = torch.cuda.CUDAGraph()
        self.optimizer.prepare_capture()
        self.optimizer.zero_grad(set_to_none=True)
        # with collectives inside, other threads (the process group's watchdog polling its events) must stay legal while the
        # capture runs: thread-local error mode, as torch documents for whole-network capture with NCCL
        mode = {"capture_error_mode": "thread_local"} if self._reducer is not None else {}
        if self._reducer is not None:
            self._reducer.capturing = True
        try:
            loss = self._capture_body(graph, mode, sx, sy)
        finally:
            if self._reducer is not None:
                self._reducer.capturing = False
        return GraphedStep(self, graph, sx, sy, loss.detach())

    def _capture_body(self, graph, mode, sx, sy):
        with torch.cuda.graph(graph, **mode):
            if self.accum_steps > 1:
                loss = self._accumulate(sx, sy)           # (k forward / backward passes over the chunks of the static batch)
            else:
                loss = self._forward_loss(sx, sy)
                if self._reducer is not None:
                    self._reduced_backward(loss)
                else:
                    with self._batched_finishes():
                        loss.backward()
            self.optimizer.step()
        return loss


BATCHED_FINISHES = True    # `bench.py --no-batched-finishes` for the A/B


class _FinishScope:
    def __init__(self, deferred, on, vectors_only=False):
        """vectors_only: postpone only the reductions that produce VECTOR gradients (norm / layer-scale / bias parameters) -
        the octic weight-gradient slab reductions and the paired weight gradients run at once (DDP reads weight gradients in
        its hooks; the vectors are outside DDP: train.DDP_FLAT_SMALL_NUMEL)."""
        self.deferred, self.on, self.vectors_only = deferred, on, vectors_only

    def __enter__(self):
        from . import d8_layers as _L
        self.prev = (self.deferred.enabled, self.deferred.allow_pairs)
        self.deferred.enabled = self.on
        # compacted stochastic depth: slab sizes change from step to step, and 3 GB of them held to the end of the backward
        # pass turn the caching allocator's reuse into fresh allocations (measured: +10 ms of host time per step)
        self.deferred.slabs_too = not _L.COMPACT_DROP_PATH and not self.vectors_only
        self.deferred.allow_pairs = not self.vectors_only
        return self

    def __exit__(self, *exc):
        self.deferred.enabled, self.deferred.allow_pairs = self.prev
        self.deferred.flush()            # (the engine's end-of-backward callback has normally emptied the list already)
        return False


class GraphedStep:
    """A captured training iteration (see ``Trainer.capture``).  ``replay`` copies the batch into the graph's input
    buffers, launches the graph and returns the (device-resident) loss of this iteration."""

    def __init__(self, trainer, graph, samples, targets, loss):
        self.trainer, self.graph, self.samples, self.targets, self.loss = trainer, graph, samples, targets, loss

    def replay(self, samples=None, targets=None):
        t = self.trainer
        t._steps += 1
        # the same late check as Trainer.step (the graph's loss buffer is overwritten by every replay: _LossWatch
        # keeps a pinned host copy per step, taken by a stream-ordered copy behind the replay)
        if t._steps % t.check_every == 0:
            t._watch.check()
        if samples is not None and samples.data_ptr() != self.samples.data_ptr():
            self.samples.copy_(samples, non_blocking=True)
        if targets is not None and targets.data_ptr() != self.targets.data_ptr():
            self.targets.copy_(targets, non_blocking=True)
        self.graph.replay()
        t.optimizer.step_count += 1
        t._watch.push(self.loss)
        return self.loss


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def init_distributed(force=False):
    """One process per GPU, RCCL via torch.distributed (backend name 'nccl' on ROCm).  force: also at world size 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # no re-use of HIP events between collectives (see GradReducer.reduce_all: an event once recorded under capture stays
        # "captured" for hipEventQuery on this stack)
        os.environ.setdefault("TORCH_NCCL_CUDA_EVENT_CACHE", "0")
        backend = "nccl" if torch.cuda.is_available() else "gloo"
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return world, rank, local_rank
