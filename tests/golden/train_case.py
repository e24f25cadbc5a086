"""The §8a-13 train-step parity case, shared by the golden generator and the tests.

Depth-4, D=128 hybrid (2 octic + 2 standard DeiT-III blocks, drop_path 0, name-keyed parameters), batch 8 of 32x32
images, multi-hot targets, BCE-with-logits (deit/engine.py:53-64), K LAMB steps (lr 3e-3, wd 0.02, eps 1e-8, clip 1.0,
timm's no-decay rule) + EMA.  ``run_train_case`` drives any namespace exposing the reference's class names with the
numpy LAMB restatement of oracle/lamb_ref.py, so reference / oracle runs share the optimizer arithmetic."""
import numpy as np
import torch

import cases

STEPS = 3
LR, WD, EPS, EMA_DECAY = 3e-3, 0.02, 1e-8, 0.9
SPEC = dict(img_size=32, patch_size=4, num_classes=10, embed_dim=128, depth=4, num_heads=4, qkv_bias=True,
            blocks="deit", init_scale=0.1, drop_path_rate=0.0)


def batch():
    x = cases.randn("train.x", 8, 3, 32, 32)
    g = cases._gen("train.y")
    y = torch.zeros(8, 10)
    y.scatter_(1, torch.randint(0, 10, (8, 2), generator=g), 1.0)
    return x, y


def build(ns):
    return cases.fill_parameters(cases.build_model(ns, SPEC)).train()


def sample(t, n=64):
    flat = t.detach().double().flatten().cpu()
    return flat[:: max(1, flat.numel() // n)][:n].numpy()


def summarize(named_params, losses, gnorms, ema=None):
    res = {"losses": np.asarray(losses, dtype=np.float64), "grad_norms": np.asarray(gnorms, dtype=np.float64)}
    for i, (name, p) in enumerate(named_params):
        res[f"p_sample.{name}"] = sample(p)
        res[f"p_norm.{name}"] = np.array([float(p.detach().double().norm())])
        if ema is not None:
            res[f"ema_sample.{name}"] = sample(torch.as_tensor(ema[i]))
    return res


def run_train_case(ns):
    """Reference-style run: model of `ns` on CPU in f32, optimizer = oracle.lamb_ref.LambRef (float64 numpy)."""
    from oracle.lamb_ref import LambRef, ema_update, weight_decay_of
    model = build(ns)
    names = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    nd = model.no_weight_decay()
    opt = LambRef([tuple(p.shape) for _, p in names], weight_decay_of([(n, tuple(p.shape)) for n, p in names], WD, nd),
                  lr=LR, eps=EPS)
    ema = [p.detach().double().numpy().copy() for _, p in names]
    x, y = batch()
    crit = torch.nn.BCEWithLogitsLoss()
    losses, gnorms = [], []
    for _ in range(STEPS):
        model.zero_grad(set_to_none=True)
        loss = crit(model(x), y)
        loss.backward()
        losses.append(float(loss.detach()))
        new = opt.step([p.detach().double().numpy() for _, p in names], [p.grad.double().numpy() for _, p in names])
        gnorms.append(opt.last_grad_norm)
        with torch.no_grad():
            for (_, p), q in zip(names, new):
                p.copy_(torch.from_numpy(q).to(p.dtype))
        ema = ema_update(ema, [p.detach().double().numpy() for _, p in names], EMA_DECAY)
    with torch.no_grad():
        losses.append(float(crit(model(x), y)))
    return summarize(names, losses, gnorms, ema)
