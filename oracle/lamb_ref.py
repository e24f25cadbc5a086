"""TEST INFRASTRUCTURE — CPU restatement of the optimizer step of the reference recipe.  Only tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() may import this module.

The reference trains with ``--opt fusedlamb`` (experiments/train_deit.py:42) through timm's
``create_optimizer(args, model)`` (deit/main.py:365), i.e. NVIDIA apex ``FusedLAMB`` at commit 2386a912164
(DEIT_ENV.md:3-15), constructed by timm 1.0.12 as ``FusedLAMB(param_groups, lr=args.lr,
weight_decay=args.weight_decay, eps=args.opt_eps)`` — ``--opt-eps`` defaults to 1e-8 (deit/main.py:68) and
``--opt-betas`` to None (apex default (0.9, 0.999), deit/main.py:70-71); timm's factory puts 1-D tensors, ``.bias``
and the model's ``no_weight_decay()`` names into a weight_decay = 0 group.

apex is NOT part of /root/reference (external dependency, not vendored) and cannot be built here (CUDA extension):
**parity unpinned** against an apex binary.  What is restated is apex's published algorithm
(``apex/optimizers/fused_lamb.py`` + ``csrc/multi_tensor_lamb.cu``, FusedLAMB defaults: bias_correction=True,
adam_w_mode=True, grad_averaging=True, max_grad_norm=1.0, use_nvlamb=False), one parameter element at a time in
numpy float64 so that it shares no code with the product's torch/HIP implementations:

    g_norm  = sqrt(sum over ALL tensors of sum(g^2))                       (multi_tensor_l2norm over every grad)
    clip    = g_norm / max_grad_norm  if g_norm > max_grad_norm else 1     (stage 1: ``clipped_global_grad_norm``)
    g'      = g / clip
    m       = beta1 m + (1 - beta1) g'           (grad_averaging: beta3 = 1 - beta1)
    v       = beta2 v + (1 - beta2) g'^2
    u       = (m / (1 - beta1^t)) / (sqrt(v / (1 - beta2^t)) + eps) + wd * p      (adam_w_mode = MOMENT_MODE_1)
    ratio   = |p| / |u| per tensor if (wd != 0 and |p| > 0 and |u| > 0) else 1   (stage 2; use_nvlamb = False)
    p       = p - lr * ratio * u

tests/test_lamb_oracle.py pins this restatement to values worked out by hand from those formulas.
timm's ModelEma (deit/main.py:344-351): ema = decay * ema + (1 - decay) * p after every step.
"""
import numpy as np


class LambRef:
    def __init__(self, shapes, weight_decays, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=1.0):
        self.lr, self.b1, self.b2, self.eps, self.max_grad_norm = float(lr), float(betas[0]), float(betas[1]), float(eps), max_grad_norm
        self.wd = [float(w) for w in weight_decays]
        self.m = [np.zeros(s, dtype=np.float64) for s in shapes]
        self.v = [np.zeros(s, dtype=np.float64) for s in shapes]
        self.t = 0
        self.last_grad_norm = None

    def step(self, params, grads):
        """params, grads: lists of float64 ndarrays; returns the updated parameter list (inputs are not modified)."""
        gn = float(np.sqrt(sum(float(np.sum(np.square(g, dtype=np.float64))) for g in grads)))
        self.last_grad_norm = gn
        if not np.isfinite(gn):
            return [p.copy() for p in params]
        clip = gn / self.max_grad_norm if (self.max_grad_norm and gn > self.max_grad_norm) else 1.0
        self.t += 1
        bc1, bc2 = 1.0 - self.b1 ** self.t, 1.0 - self.b2 ** self.t
        out = []
        for i, (p, g) in enumerate(zip(params, grads)):
            p = np.asarray(p, dtype=np.float64)
            gs = np.asarray(g, dtype=np.float64) / clip
            self.m[i] = self.b1 * self.m[i] + (1.0 - self.b1) * gs
            self.v[i] = self.b2 * self.v[i] + (1.0 - self.b2) * gs * gs
            u = (self.m[i] / bc1) / (np.sqrt(self.v[i] / bc2) + self.eps) + self.wd[i] * p
            ratio = 1.0
            if self.wd[i] != 0.0:
                pn, un = float(np.sqrt(np.sum(p * p))), float(np.sqrt(np.sum(u * u)))
                if pn > 0.0 and un > 0.0:
                    ratio = pn / un
            out.append(p - self.lr * ratio * u)
        return out


def ema_update(ema, params, decay):
    return [decay * e + (1.0 - decay) * p for e, p in zip(ema, params)]


def weight_decay_of(names_shapes, weight_decay, no_decay_names=()):
    """timm ``param_groups_weight_decay`` (optim_factory): no decay for ndim <= 1, ``.bias`` and listed names."""
    return [0.0 if (len(shape) <= 1 or name.endswith(".bias") or name in no_decay_names) else float(weight_decay)
            for name, shape in names_shapes]
