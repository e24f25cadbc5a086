"""Dispatcher-visible ops (torch.library) for the one custom-op boundary the reference really has:
``TritonGeluD8Function`` (octic_vits/d8_gelu.py:456-478, module ``TritonGeluD8`` d8_gelu.py:480-482).

The engine's other entry points are ``torch.autograd.Function``s over ctypes - correct, but opaque to ``torch.compile``
(Dynamo breaks the graph at every ctypes call).  These ops are registered with schemas, fake-tensor (meta) kernels and
``register_autograd`` formulas, so ``torch.compile`` (the reference recipe's default: deit/main.py:341-342,
experiments/complexity.py:79-82) traces THROUGH the D8-GELU instead of around it, ``torch.ops.octic.*`` shows them to
anyone who inspects the dispatcher, and they can be captured, functionalized and exported like any ATen op:

    torch.ops.octic.gelu_d8(x_packed, c)                       packed rows [.., 8c]  (the engine's native layout)
    torch.ops.octic.gelu_d8_bwd(grad, x_packed, c)
    torch.ops.octic.gelu_d8_tuple(A1, A2, B1, B2, E)           the reference's five tensors (strides go to the kernel)
    torch.ops.octic.gelu_d8_tuple_bwd(gA1, .., gE, A1, .., E)

The kernels are the same HIP launches as before (csrc/elementwise.hip through the C ABI); there is no CPU kernel: a CPU
tensor raises, like every product op."""
import torch

from . import ops


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------------------------- packed rows
@torch.library.custom_op("octic::gelu_d8", mutates_args=())
def gelu_d8(x: torch.Tensor, c: int) -> torch.Tensor:
    x = _c(x)
    y = torch.empty_like(x)
    ops.gelu_fwd(ops.pview(x, c), ops.pview(y, c), x.numel() // (8 * c), c, x.dtype, x)
    return y


@gelu_d8.register_fake
def _(x, c):
    if x.shape[-1] != 8 * c:
        raise ValueError(f"gelu_d8: last dimension {x.shape[-1]} is not 8 c = {8 * c}")
    return torch.empty_like(x, memory_format=torch.contiguous_format)


@torch.library.custom_op("octic::gelu_d8_bwd", mutates_args=())
def gelu_d8_bwd(g: torch.Tensor, x: torch.Tensor, c: int) -> torch.Tensor:
    x = _c(x)
    g = _c(g.to(x.dtype))
    gi = torch.empty_like(x)
    ops.gelu_bwd(ops.pview(g, c), ops.pview(x, c), ops.pview(gi, c), x.numel() // (8 * c), c, x.dtype, x)
    return gi


@gelu_d8_bwd.register_fake
def _(g, x, c):
    return torch.empty_like(x, memory_format=torch.contiguous_format)


def _gelu_setup(ctx, inputs, output):
    x, c = inputs
    ctx.save_for_backward(x)
    ctx.c = c


def _gelu_backward(ctx, g):
    (x,) = ctx.saved_tensors
    return torch.ops.octic.gelu_d8_bwd(g, x, ctx.c), None


gelu_d8.register_autograd(_gelu_backward, setup_context=_gelu_setup)


# ---------------------------------------------------------------------------- the reference's five separate tensors
@torch.library.custom_op("octic::gelu_d8_tuple", mutates_args=())
def gelu_d8_tuple(x_A1: torch.Tensor, x_A2: torch.Tensor, x_B1: torch.Tensor, x_B2: torch.Tensor,
                  x_2d: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    xs = tuple(_c(t) for t in (x_A1, x_A2, x_B1, x_B2, x_2d))
    c = xs[0].shape[-1]
    ys = tuple(torch.empty_like(t) for t in xs)
    xv, _k1 = ops.tview(xs, c)
    yv, _k2 = ops.tview(ys, c)
    ops.gelu_fwd(xv, yv, xs[0].numel() // c, c, xs[0].dtype, xs[0])
    return ys


@gelu_d8_tuple.register_fake
def _(x_A1, x_A2, x_B1, x_B2, x_2d):
    c = x_A1.shape[-1]
    if tuple(x_2d.shape[-2:]) != (2, 2 * c):
        raise ValueError(f"gelu_d8_tuple: E irrep must be [..., 2, {2 * c}], got {tuple(x_2d.shape)}")
    return tuple(torch.empty_like(t, memory_format=torch.contiguous_format) for t in (x_A1, x_A2, x_B1, x_B2, x_2d))


@torch.library.custom_op("octic::gelu_d8_tuple_bwd", mutates_args=())
def gelu_d8_tuple_bwd(g_A1: torch.Tensor, g_A2: torch.Tensor, g_B1: torch.Tensor, g_B2: torch.Tensor, g_2d: torch.Tensor,
                      x_A1: torch.Tensor, x_A2: torch.Tensor, x_B1: torch.Tensor, x_B2: torch.Tensor,
                      x_2d: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    xs = tuple(_c(t) for t in (x_A1, x_A2, x_B1, x_B2, x_2d))
    c = xs[0].shape[-1]
    gs = tuple(_c(g.to(xs[0].dtype)) for g in (g_A1, g_A2, g_B1, g_B2, g_2d))
    outs = tuple(torch.empty_like(t) for t in xs)
    gv, _k1 = ops.tview(gs, c)
    xv, _k2 = ops.tview(xs, c)
    ov, _k3 = ops.tview(outs, c)
    ops.gelu_bwd(gv, xv, ov, xs[0].numel() // c, c, xs[0].dtype, xs[0])
    return outs


@gelu_d8_tuple_bwd.register_fake
def _(g_A1, g_A2, g_B1, g_B2, g_2d, x_A1, x_A2, x_B1, x_B2, x_2d):
    return tuple(torch.empty_like(t, memory_format=torch.contiguous_format) for t in (x_A1, x_A2, x_B1, x_B2, x_2d))


def _gelu_tuple_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _gelu_tuple_backward(ctx, g_A1, g_A2, g_B1, g_B2, g_2d):
    return torch.ops.octic.gelu_d8_tuple_bwd(g_A1, g_A2, g_B1, g_B2, g_2d, *ctx.saved_tensors)


gelu_d8_tuple.register_autograd(_gelu_tuple_backward, setup_context=_gelu_tuple_setup)
