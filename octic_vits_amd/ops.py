"""Raw (non-autograd) launches of the HIP engine on torch tensors.

torch is plumbing here: it owns device memory and the current HIP stream; every call goes straight
through the C ABI (include/octic_hip.h) via ctypes.  Tensors are described to the ABI as
``octic_view`` (5 base pointers + row strides), so both the reference's 5-tuple of separate tensors
and the engine's packed ``[B, T, 8c]`` rows are handled by the same kernels without a copy.
"""
import ctypes

import torch

from . import _lib
from ._lib import BF16, F32, OcticView, PtrArray5, check, lib

_DT = {torch.float32: F32, torch.bfloat16: BF16}
_DTN = {torch.float32: "f32", torch.bfloat16: "bf16"}


class KernelTimer:
    """Optional per-launch timing with HIP events on the stream the kernels are launched on (torch's current
    stream).  Disabled by default: zero overhead.  bench.py enables it over the timed steps to report the
    dominant kernel's achieved bytes/s against the HBM roofline."""

    def __init__(self):
        self.on = False
        self.records = []
        self.overhead_us = 0.0

    def enable(self):
        self.on, self.records = True, []
        # What an event pair reads with NOTHING between its two records (the second marker's own cost on the queue, ~1 us on
        # MI355X): subtracted from every bracket, so that ~1800 brackets per step do not add up to a millisecond of phantom
        # kernel time.  Median of 64 empty pairs on the current stream.
        pairs = []
        for _ in range(64):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        self.overhead_us = sorted(a.elapsed_time(b) * 1e3 for a, b in pairs)[len(pairs) // 2]

    def disable(self):
        self.on = False

    def start(self):
        if not self.on:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def stop(self, e0, name, alg_bytes, flops=0):
        if e0 is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.records.append((name, e0, e1, alg_bytes, flops))

    def _agg(self):
        torch.cuda.synchronize()
        agg = {}
        for name, e0, e1, b, f in self.records:
            a = agg.setdefault(name, {"name": name, "launches": 0, "total_us": 0.0, "bytes": 0.0, "flops": 0.0})
            a["launches"] += 1
            a["total_us"] += max(0.0, e0.elapsed_time(e1) * 1e3 - self.overhead_us)
            a["bytes"] += b
            a["flops"] += f
        for a in agg.values():
            a["avg_us"] = a["total_us"] / a["launches"]
            a["alg_bytes_per_launch"] = a["bytes"] / a["launches"]
            a["flops_per_launch"] = a["flops"] / a["launches"]
        return agg

    @staticmethod
    def bound_of(name):
        """Roofline that bounds a kernel family: the attention kernels are MFMA/VALU-bound (q,k,v,o are read once,
        14 T^2 hd flops per head), every other kernel of the engine moves more bytes than it can compute on."""
        return "mfma" if name.startswith(("attn_", "dense_nt_kernel", "dense_tn_kernel", "library_gemm")) else "hbm"

    def dominant(self, bound=None):
        agg = [a for a in self._agg().values() if bound is None or self.bound_of(a["name"]) == bound]
        return max(agg, key=lambda a: a["total_us"]) if agg else None

    def summary(self):
        out = {}
        for k, a in sorted(self._agg().items(), key=lambda kv: -kv[1]["total_us"]):
            out[k] = {"launches": a["launches"], "total_us": round(a["total_us"], 1), "avg_us": round(a["avg_us"], 2),
                      "GBps": round(a["alg_bytes_per_launch"] / a["avg_us"] / 1e3, 1)}
            if a["flops_per_launch"]:
                out[k]["TFLOPs"] = round(a["flops_per_launch"] / a["avg_us"] / 1e6, 1)
        return out


KERNEL_TIMER = KernelTimer()


def dt_code(dtype):
    try:
        return _DT[dtype]
    except KeyError:
        raise TypeError(f"octic engine supports float32 and bfloat16, got {dtype}") from None


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _require_cuda(t):
    if not t.is_cuda:
        raise RuntimeError("octic_vits_amd ops run on the GPU only (no CPU fallback); got a CPU tensor")


def pview(t: torch.Tensor, c: int) -> OcticView:
    """View of a packed tensor [..., 8c] = [A1|A2|B1|B2|E_row0|E_row1]."""
    _require_cuda(t)
    if t.shape[-1] != 8 * c or not t.is_contiguous():
        raise ValueError(f"expected a contiguous packed tensor with last dim {8 * c}, got {tuple(t.shape)}")
    es, base, D = t.element_size(), t.data_ptr(), 8 * c
    v = OcticView()
    for i in range(4):
        v.ptr[i] = base + i * c * es
        v.ld[i] = D
    v.ptr[4] = base + 4 * c * es
    v.ld[4] = D
    return v


def _rows_ok(t, inner):
    """t viewed as rows of `inner` contiguous elements with one uniform row stride?"""
    lead = t.shape[:-len(inner)]
    st = t.stride()
    exp = 1
    for k in range(1, len(inner) + 1):
        if t.shape[-k] != inner[-k] or (st[-k] != exp and t.shape[-k] != 1):
            return False
        exp *= inner[-k]
    # leading dims must collapse onto a single stride
    ld = st[len(lead) - 1] if lead else exp
    acc = ld
    for k in range(len(lead) - 1, -1, -1):
        if t.shape[k] != 1 and st[k] != acc:
            return False
        acc *= t.shape[k]
    return True


def tview(xs, c: int):
    """View of a reference-style 5-tuple (A1..B2: [..., c]; E: [..., 2, 2c]).  Returns (view, keepalive)."""
    keep = []
    v = OcticView()
    for i in range(5):
        t = xs[i]
        _require_cuda(t)
        inner = (c,) if i < 4 else (2, 2 * c)
        if tuple(t.shape[-len(inner):]) != inner:
            raise ValueError(f"irrep {i}: expected trailing shape {inner}, got {tuple(t.shape)}")
        if not _rows_ok(t, inner):
            t = t.contiguous()
        keep.append(t)
        nlead = t.dim() - len(inner)
        v.ptr[i] = t.data_ptr()
        v.ld[i] = t.stride(nlead - 1) if nlead > 0 else (c if i < 4 else 4 * c)
    return v, keep


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _arr5(ts):
    a = PtrArray5()
    for i in range(5):
        a[i] = ts[i].data_ptr() if (ts is not None and ts[i] is not None) else 0
    return a


def split_packed(t, c):
    """The reference's 5-tuple as zero-copy views of a packed [B,T,8c] tensor."""
    lead = t.shape[:-1]
    return (t[..., 0:c], t[..., c:2 * c], t[..., 2 * c:3 * c], t[..., 3 * c:4 * c],
            t[..., 4 * c:].unflatten(-1, (2, 2 * c)))


# ------------------------------------------------------------------------------------------ launches
def gelu_fwd(xv, yv, M, c, dtype, ref):
    t = KERNEL_TIMER.start()
    check(lib().octic_gelu_d8_fwd(ctypes.byref(xv), ctypes.byref(yv), M, c, dt_code(dtype), _stream(ref)))
    KERNEL_TIMER.stop(t, f"gelu_fwd_kernel<{_DTN[dtype]}>", 2 * M * 8 * c * ref.element_size())


def gelu_bwd(gv, xv, ov, M, c, dtype, ref):
    t = KERNEL_TIMER.start()
    check(lib().octic_gelu_d8_bwd(ctypes.byref(gv), ctypes.byref(xv), ctypes.byref(ov), M, c, dt_code(dtype), _stream(ref)))
    KERNEL_TIMER.stop(t, f"gelu_bwd_kernel<{_DTN[dtype]}>", 3 * M * 8 * c * ref.element_size())


def layernorm_fwd(x, alpha5, beta, eps, out_dtype, c, want_stats=True):
    """x: packed f32 [..., 8c] -> (y packed out_dtype, stats [M,8] f32)."""
    M = x.numel() // (8 * c)
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    stats = torch.empty((M, 8), dtype=torch.float32, device=x.device) if want_stats else None
    xv, yv = pview(x, c), pview(y, c)
    t = KERNEL_TIMER.start()
    check(lib().octic_layernorm_d8_fwd(ctypes.byref(xv), ctypes.byref(yv), _arr5(alpha5), _p(beta), _p(stats), M, c,
                                       float(eps), dt_code(out_dtype), _stream(x)))
    KERNEL_TIMER.stop(t, f"ln_fwd_kernel<{_DTN[out_dtype]}>", M * 8 * c * (4 + y.element_size()))
    return y, stats


def layernorm_bwd(g, x, stats, alpha5, dres, c, want_param_grads=True):
    """Returns (dx f32 packed, dalpha5 or None, dbeta or None).  dx = dres + LN'(g)."""
    M = x.numel() // (8 * c)
    dx = torch.empty_like(x)
    nblk = lib().octic_layernorm_d8_bwd_blocks(M)
    partials = torch.empty((nblk, 2, 8 * c), dtype=torch.float32, device=x.device)
    gv, xv, dv = pview(g, c), pview(x, c), pview(dx, c)
    rv = pview(dres, c) if dres is not None else None
    t = KERNEL_TIMER.start()
    check(lib().octic_layernorm_d8_bwd(ctypes.byref(gv), ctypes.byref(xv), _p(stats), _arr5(alpha5),
                                       ctypes.byref(rv) if rv is not None else None, ctypes.byref(dv), _p(partials),
                                       M, c, dt_code(g.dtype), _stream(x)))
    KERNEL_TIMER.stop(t, f"ln_bwd_kernel<{_DTN[g.dtype]}>", M * 8 * c * (g.element_size() + 8 + (4 if dres is not None else 0)))
    if not want_param_grads or alpha5 is None:
        return dx, None, None
    dal = [torch.empty_like(a) for a in alpha5]
    dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
    _ln_finish(partials, nblk, c, dal, dbeta, _stream(x))
    return dx, dal, dbeta


def sample_blocks_ok(full, compact, idx):
    return (full.is_cuda and full.is_contiguous() and compact.is_contiguous() and full.dtype == compact.dtype
            and idx.dtype == torch.int64 and idx.is_cuda and full.dim() >= 2 and full.shape[1:] == compact.shape[1:]
            and (full[0].numel() * full.element_size()) % 16 == 0 and 0 < idx.numel() == compact.shape[0] <= 65535)


def gather_samples(full, idx, out=None):
    """full[idx] for whole samples (leading dimension) on csrc/elementwise.hip sample_blocks_kernel."""
    if out is None:
        out = torch.empty((idx.numel(),) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
    check(lib().octic_sample_blocks(_p(full), _p(out), _p(idx), idx.numel(), full[0].numel() * full.element_size(), 0,
                                    _stream(full)))
    return out


def scatter_samples_(full, idx, compact):
    """full[idx] = compact, in place (idx distinct)."""
    check(lib().octic_sample_blocks(_p(compact), _p(full), _p(idx), idx.numel(), full[0].numel() * full.element_size(), 1,
                                    _stream(full)))
    return full


def _ln_finish(partials, nblk, c, dal, dbeta, stream):
    """octic_layernorm_d8_bwd_finish now, or batched at the end of the running backward pass (DEFERRED_FINISHES)."""
    if DEFERRED_FINISHES.enabled and _in_backward():
        DEFERRED_FINISHES.add_ln(partials, nblk, c, dal, dbeta, stream)
        return
    check(lib().octic_layernorm_d8_bwd_finish(_p(partials), nblk, c, _arr5(dal), _p(dbeta), stream))


def layernorm_bwd_cast_ok(g, x, c):
    """Shapes of octic_layernorm_d8_bwd_cast: bf16 cotangent, one packed tensor, c = 32 ... 160 in steps of 32."""
    return g.dtype == torch.bfloat16 and x.dtype == torch.float32 and c % 32 == 0 and c <= 160


def layernorm_bwd_cast(g, x, stats, alpha5, dres, c, rs, rps, want_param_grads=True):
    """layernorm_bwd that also returns bf16(rs[row // rps] * dx): (dx, dalpha5, dbeta, gcast)."""
    M = x.numel() // (8 * c)
    dx = torch.empty_like(x)
    gc = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    nblk = lib().octic_layernorm_d8_bwd_blocks(M)
    partials = torch.empty((nblk, 2, 8 * c), dtype=torch.float32, device=x.device)
    gv, xv, dv = pview(g, c), pview(x, c), pview(dx, c)
    rv = pview(dres, c) if dres is not None else None
    t = KERNEL_TIMER.start()
    check(lib().octic_layernorm_d8_bwd_cast(ctypes.byref(gv), ctypes.byref(xv), _p(stats), _arr5(alpha5),
                                            ctypes.byref(rv) if rv is not None else None, ctypes.byref(dv), _p(partials),
                                            M, c, _p(rs), int(rps), _p(gc), _stream(x)))
    KERNEL_TIMER.stop(t, "ln_bwd_kernel<bf16>", M * 8 * c * (2 + 8 + (4 if dres is not None else 0) + 2))
    if not want_param_grads or alpha5 is None:
        return dx, None, None, gc
    dal = [torch.empty_like(a) for a in alpha5]
    dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
    _ln_finish(partials, nblk, c, dal, dbeta, _stream(x))
    return dx, dal, dbeta, gc


_DW_WS = {}


def dense_wgrad_ok(M, N, K):
    return N % 256 == 0 and (K % 256 == 0 or K % 320 == 0) and (N // 256) * (K // 256) <= 256 and M > 0


# Where a weight gradient is to be WRITTEN (train.Trainer under DistributedDataParallel, one micro-batch): parameter address ->
# its bucket view.  A gradient produced inside the bucket is an alias of the view: the reducer recognises it and skips its
# per-tensor copy (reducer.cpp mark_variable_ready_dense).  None = fresh tensors (every other mode).
GRAD_DEST = None


def grad_dest(param, shape):
    """A fresh tensor object aliasing the registered destination of `param`'s gradient, or None."""
    if GRAD_DEST is None or param is None:
        return None
    d = GRAD_DEST.get(param.data_ptr())
    if d is None or param.grad is not None or tuple(d.shape) != tuple(shape) or d.dtype != torch.float32 or not d.is_contiguous():
        return None
    return d.detach()


def grad_written(*params):
    """The launches that write these parameters' gradients into their registered destinations are on the stream: a registry
    that reduces whole buckets as they fill (train.GradReducer) is told; a plain dict (DDP's bucket views) is not."""
    w = getattr(GRAD_DEST, "written", None)
    if w is not None:
        for prm in params:
            if prm is not None:
                w(prm.data_ptr())


def dense_wgrad_tn(dy, x, name=None, out=None):
    """dW[N,K] = dy[M,N]^T @ x[M,K] in f32 on csrc/dense_wgrad.hip (out: write it there)."""
    _require_cuda(dy)
    M, N = dy.shape
    K = x.shape[1]
    if dy.stride(1) != 1 or x.stride(1) != 1 or x.shape[0] != M:
        raise ValueError("dense_wgrad_tn: operands must be [M,N] / [M,K] row-major")
    need = int(lib().octic_dense_wgrad_workspace_bytes(M, N, K))
    ws = _DW_WS.get(dy.device)
    if ws is None or ws.numel() < need:          # one workspace per device: launches on a stream are serial
        ws = _DW_WS[dy.device] = torch.zeros(need, dtype=torch.uint8, device=dy.device)
    dw = out if out is not None else torch.empty((N, K), dtype=torch.float32, device=dy.device)
    t = KERNEL_TIMER.start()
    check(lib().octic_dense_wgrad_tn(_p(dy), _p(x), M, N, K, dy.stride(0), x.stride(0), _p(dw), _p(ws), _stream(dy)))
    KERNEL_TIMER.stop(t, name or f"dense_tn_kernel<{N}x{K}>", 2 * (M * N + M * K) + 4 * N * K, 2.0 * M * N * K)
    return dw


def dense_wgrad_pair_ok(M, N0, N1, K):
    return (N0 % 256 == 0 and N1 % 256 == 0 and (K % 256 == 0 or K % 320 == 0) and M > 0
            and ((N0 + N1) // 256) * (K // 256) <= 256)


def dense_wgrad_tn_pair(dy0, x0, dy1, x1, dw1=None, dw0=None):
    """(dW0 [N0,K], dW1 [N1,K]) = (dy0^T x0, dy1^T x1) as ONE launch of csrc/dense_wgrad.hip (same M, same K): the qkv and proj
    weight gradients of a standard block.  dw1: write the second result into this (already handed-out) tensor; dw0: a
    registered destination of the first (grad_dest)."""
    _require_cuda(dy0)
    M, N0 = dy0.shape
    N1, K = dy1.shape[1], x0.shape[1]
    if not (dy1.shape[0] == M and x0.shape[0] == M and x1.shape == (M, K) and all(t.stride(1) == 1 for t in (dy0, x0, dy1, x1))):
        raise ValueError("dense_wgrad_tn_pair: operands must be [M,N0] / [M,K] / [M,N1] / [M,K] row-major")
    need = int(lib().octic_dense_wgrad_pair_workspace_bytes(M, N0, N1, K))
    ws = _DW_WS.get(dy0.device)
    if ws is None or ws.numel() < need:
        ws = _DW_WS[dy0.device] = torch.zeros(need, dtype=torch.uint8, device=dy0.device)
    if dw0 is None:
        dw0 = torch.empty((N0, K), dtype=torch.float32, device=dy0.device)
    if dw1 is None:
        dw1 = torch.empty((N1, K), dtype=torch.float32, device=dy0.device)
    t = KERNEL_TIMER.start()
    check(lib().octic_dense_wgrad_tn_pair(_p(dy0), _p(x0), N0, dy0.stride(0), x0.stride(0), _p(dw0), _p(dy1), _p(x1), N1,
                                          dy1.stride(0), x1.stride(0), _p(dw1), M, K, _p(ws), _stream(dy0)))
    KERNEL_TIMER.stop(t, f"dense_tn_kernel<{N0}+{N1}x{K}>", 2 * M * (N0 + N1 + 2 * K) + 4 * (N0 + N1) * K, 2.0 * M * (N0 + N1) * K)
    return dw0, dw1


def linear_fwd(xv, w5, bias, yv, M, cin, cout, dtype, out_dtype, ref, resid_v=None, rs=None, rps=1, cs5=None):
    t = KERNEL_TIMER.start()
    check(lib().octic_linear_d8_fwd(ctypes.byref(xv), _arr5(w5), _p(bias), ctypes.byref(yv),
                                    ctypes.byref(resid_v) if resid_v is not None else None, _p(rs), int(rps),
                                    _arr5(cs5) if cs5 is not None else None, M, cin, cout, dt_code(dtype),
                                    dt_code(out_dtype), _stream(ref)))
    if t is not None:
        es, eo = (2 if dtype == torch.bfloat16 else 4), (2 if out_dtype == torch.bfloat16 else 4)
        nbytes = M * 8 * cin * es + M * 8 * cout * eo * (2 if resid_v is not None else 1) + 8 * cin * cout * es
        fused = int(resid_v is not None or rs is not None or cs5 is not None)
        KERNEL_TIMER.stop(t, linear_kernel_name(cin, dtype, out_dtype, fused), nbytes, 24.0 * M * cin * cout)


def linear_kernel_name(cin, dtype, out_dtype, fused):
    """Name of the kernel instantiation octic_linear_d8_fwd dispatches to (mirrors dispatch_gemm in csrc/gemm.hip)."""
    if dtype == torch.bfloat16 and cin % 32 == 0 and cin <= 160:
        return f"linear_d8_wreg_kernel<{_DTN[out_dtype]},{fused}>"
    kstep = 32 if dtype == torch.bfloat16 else 16
    if cin % kstep == 0:
        return f"linear_d8_ring_kernel<{_DTN[dtype]},{_DTN[out_dtype]},{fused}>"
    return f"linear_d8_kernel<{_DTN[dtype]},{_DTN[out_dtype]}>"


def linear_wgrad(xv, dyv, M, cin, cout, dtype, ref, w32=None, cs5=None, bias=None, dysum=None, want_bias=False,
                 may_defer=True, wparams=None):
    """Returns (dw5 [f32], dcs5 or None, dbias or None).  may_defer=False: the caller reads the results at once (e.g. casts
    them for non-f32 master weights), so the slab reduction must not be postponed to the end of the backward pass."""
    L = lib()
    dev = ref.device
    splits = L.octic_linear_d8_wgrad_splits(M, cin, cout)
    ws = torch.empty(L.octic_linear_d8_wgrad_workspace_bytes(cin, cout, splits) // 4, dtype=torch.float32, device=dev)
    t = KERNEL_TIMER.start()
    check(L.octic_linear_d8_wgrad(ctypes.byref(xv), ctypes.byref(dyv), M, cin, cout, dt_code(dtype), _p(ws), splits,
                                  _stream(ref)))
    if t is not None:
        es = 2 if dtype == torch.bfloat16 else 4
        tt = L.octic_linear_d8_wgrad_tile(M, cin, cout) // 32
        ring = dtype == torch.bfloat16 and cin % 160 == 0 and cout % 160 == 0      # wgrad.hip: wgrad_ring_ok
        name = "wgrad_ring_kernel<bf16>" if ring else f"wgrad_kernel<{_DTN[dtype]},{tt}>"
        # slabs: the two-dimensional irrep (half of the 8*cin*cout weights) uses `splits`, the others splits/2
        slab = (splits + (splits + 1) // 2) * 4 * cin * cout * 4
        KERNEL_TIMER.stop(t, name, M * 8 * (cin + cout) * es + slab, 24.0 * M * cin * cout)
    shapes = [(cout, cin)] * 4 + [(2 * cout, 2 * cin)]
    dests = [grad_dest(w, sh) for w, sh in zip(wparams, shapes)] if wparams is not None else [None] * 5
    dw = [d if d is not None else torch.empty(sh, dtype=torch.float32, device=dev) for d, sh in zip(dests, shapes)]
    dcs = None
    if cs5 is not None:
        dcs = [torch.empty(cout, dtype=torch.float32, device=dev) for _ in range(4)]
        dcs.append(torch.empty(2 * cout, dtype=torch.float32, device=dev))
    dbias = torch.empty(cout, dtype=torch.float32, device=dev) if want_bias else None
    landed = [w for w, d in zip(wparams, dests) if d is not None] if wparams is not None else []
    if may_defer and DEFERRED_FINISHES.enabled and DEFERRED_FINISHES.slabs_too and _in_backward():
        DEFERRED_FINISHES.add_wg(ws, splits, cin, cout, w32, cs5, bias, dysum, dw, dcs, dbias, _stream(ref), landed)
        return dw, dcs, dbias
    check(L.octic_linear_d8_wgrad_finish(_p(ws), splits, cin, cout, _arr5(w32) if cs5 is not None else None,
                                         _arr5(cs5) if cs5 is not None else None, _p(bias), _p(dysum), _arr5(dw),
                                         _arr5(dcs) if dcs is not None else None, _p(dbias), _stream(ref)))
    grad_written(*landed)
    return dw, dcs, dbias


def wgrad_has_colsum(cin, cout, dtype):
    return bool(lib().octic_linear_d8_wgrad_has_colsum(cin, cout, dt_code(dtype)))


def colsum_a1(dyv, M, c, dtype, ref):
    L = lib()
    nblk = L.octic_colsum_blocks(M)
    partials = torch.empty((nblk, c), dtype=torch.float32, device=ref.device)
    out = torch.empty(c, dtype=torch.float32, device=ref.device)
    check(L.octic_colsum_a1(ctypes.byref(dyv), M, c, dt_code(dtype), _p(partials), _p(out), _stream(ref)))
    return out


def cast_rowscale(x, rs, rps, out_dtype, c):
    M = x.numel() // (8 * c)
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    xv, yv = pview(x, c), pview(y, c)
    t = KERNEL_TIMER.start()
    check(lib().octic_cast_rowscale(ctypes.byref(xv), ctypes.byref(yv), _p(rs), int(rps), M, c, dt_code(out_dtype),
                                    _stream(x)))
    KERNEL_TIMER.stop(t, f"cast_rowscale_kernel<{_DTN[out_dtype]}>", M * 8 * c * (4 + y.element_size()))
    return y


def _arr3(ts):
    a = (ctypes.c_void_p * 3)()
    for i in range(3):
        a[i] = ts[i].data_ptr() if i < len(ts) else 0
    return a


def pack_heads(x, B, T, H, c, n_s):
    """x packed [B,T,n_s*8c] -> n_s separate tensors [B,H,T,8c/H]"""
    outs = [torch.empty((B, H, T, 8 * c // H), dtype=x.dtype, device=x.device) for _ in range(n_s)]
    xv = pview(x, n_s * c)
    t = KERNEL_TIMER.start()
    check(lib().octic_attn_pack_heads(ctypes.byref(xv), _arr3(outs), B, T, H, c, n_s, dt_code(x.dtype), _stream(x)))
    KERNEL_TIMER.stop(t, f"heads_permute_kernel<{_DTN[x.dtype]},pack>", 2 * x.numel() * x.element_size())
    return outs


def unpack_heads(heads, B, T, H, c):
    """list of n_s tensors [B,H,T,8c/H] -> packed [B,T,n_s*8c]"""
    heads = [h if h.is_contiguous() else h.contiguous() for h in heads]
    n_s = len(heads)
    y = torch.empty((B, T, n_s * 8 * c), dtype=heads[0].dtype, device=heads[0].device)
    yv = pview(y, n_s * c)
    t = KERNEL_TIMER.start()
    check(lib().octic_attn_unpack_heads(_arr3(heads), ctypes.byref(yv), B, T, H, c, n_s, dt_code(y.dtype), _stream(y)))
    KERNEL_TIMER.stop(t, f"heads_permute_kernel<{_DTN[y.dtype]},unpack>", 2 * y.numel() * y.element_size())
    return y


def linear_prep(w5, cs5, cin, cout, dtype, want_wb=True):
    """One launch: (wb list of 5 views | None, wt list of 5 views) in `dtype` (see octic_linear_d8_prep)."""
    dev = w5[0].device
    n = 8 * cin * cout
    wb = torch.empty(n, dtype=dtype, device=dev) if want_wb else None
    wt = torch.empty(n, dtype=dtype, device=dev)
    check(lib().octic_linear_d8_prep(_arr5(w5), _arr5(cs5) if cs5 is not None else None, cin, cout, _p(wb), _p(wt),
                                     dt_code(dtype), _stream(w5[0])))
    return prep_views(wb, cin, cout, False), prep_views(wt, cin, cout, True)


def prep_views(flat, cin, cout, transposed):
    """The five per-irrep matrices inside a flat prepared-weight buffer ([cout,cin] or, transposed, [cin,cout])."""
    if flat is None:
        return None
    small = cin * cout
    out = [flat[i * small:(i + 1) * small].view((cin, cout) if transposed else (cout, cin)) for i in range(4)]
    out.append(flat[4 * small:].view((2 * cin, 2 * cout) if transposed else (2 * cout, 2 * cin)))
    return out


def attn_fwd(q, k, v, scale):
    """q,k,v: [B,H,T,hd] bf16 views with a common stride set (last dim contiguous) -> (o [B,H,T,hd], lse [B,H,T])"""
    B, H, T, hd = q.shape
    st = q.stride()
    if st[3] != 1 or k.stride() != st or v.stride() != st:
        raise ValueError("attn_fwd: q, k, v must share strides and be contiguous in the last dim")
    o = torch.empty((B, H, T, hd), dtype=q.dtype, device=q.device)
    lse = torch.empty((B, H, T), dtype=torch.float32, device=q.device)
    t = KERNEL_TIMER.start()
    check(lib().octic_attn_fwd(_p(q), _p(k), _p(v), _p(o), _p(lse), B, H, T, hd, st[0], st[1], st[2],
                               o.stride(0), o.stride(1), o.stride(2), float(scale), _stream(q)))
    KERNEL_TIMER.stop(t, "attn_fwd_kernel", 4 * q.numel() * 2, 4.0 * B * H * T * T * hd)
    return o, lse


def attn_supported(T, hd, dtype):
    """Shapes the HIP attention core handles (others keep torch SDPA): bf16, T <= 320, hd % 16 == 0, LDS fits."""
    if dtype != torch.bfloat16 or T > 320 or hd % 16 or hd > 128:
        return False
    tp = (T + 31) // 32 * 32
    cols = max((hd + 31) // 32 * 32, hd)
    return 2 * tp * (cols * 2 + 16) + 2 * tp * 4 <= 160 * 1024


# Single-pass attention backward (csrc/attn80_bwd.hip: P and dS computed once, 10 T^2 hd FLOP) for head_dim 80, T = 257;
# False = the dq + dkv pair (14 T^2 hd) for every shape (bench --no-fused-attn-bwd)
ATTN_BWD_FUSED = True


def _attn_bwd_phases(T, hd):
    """(phase, timer name, algorithmic bytes per element of q, flops per B H T^2 hd) of the backward launches."""
    if ATTN_BWD_FUSED and hd == 80 and (T == 257 or 192 < T <= 256 or T <= 64):    # csrc/attn80_bwd.hip: attn80_bwd_ok
        return ((3, "attn_bwd_kernel", 8, 10.0),)         # reads q k v o dO, writes dq dk dv
    return ((1, "attn_bwd_dq_kernel", 6, 6.0), (2, "attn_bwd_dkv_kernel", 6, 8.0))


def attn_bwd(q, k, v, o, dout, lse, scale, dq, dk, dv):
    """All tensors are [B,H,T,hd] views; q/k/v share strides, o/dout share strides, dq/dk/dv share strides."""
    B, H, T, hd = q.shape
    st, so, sg = q.stride(), o.stride(), dq.stride()
    if k.stride() != st or v.stride() != st or dout.stride() != so or dk.stride() != sg or dv.stride() != sg:
        raise ValueError("attn_bwd: stride sets differ")
    delta = torch.empty((B, H, T), dtype=torch.float32, device=q.device)
    for phase, name, nbytes, flops in _attn_bwd_phases(T, hd):
        t = KERNEL_TIMER.start()
        check(lib().octic_attn_bwd(_p(q), _p(k), _p(v), _p(o), _p(dout), _p(lse), _p(delta), _p(dq), _p(dk), _p(dv), B, H,
                                   T, hd, st[0], st[1], st[2], so[0], so[1], so[2], sg[0], sg[1], sg[2], float(scale),
                                   phase, _stream(q)))
        KERNEL_TIMER.stop(t, name, nbytes * q.numel() * 2, flops * B * H * T * T * hd)


def attn_packed_ok(T, c, H, dtype):
    """Shapes of octic_attn_{fwd,bwd}_packed: bf16, head_dim 80 (c = 10 H: ViT-H/14) or 64 (c = 8 H: ViT-L/16), T <= 320."""
    return dtype == torch.bfloat16 and c in (10 * H, 8 * H) and 0 < T <= 320 and attn_supported(T, 8 * (c // H), dtype)


def attn_fwd_packed(qkv, H, c, scale, out=None):
    """qkv packed [B,T,3*8c] bf16 -> (o packed [B,T,8c], lse [B,H,T]); no head pack / unpack copies."""
    B, T = qkv.shape[0], qkv.shape[1]
    o = out if out is not None else torch.empty((B, T, 8 * c), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, H, T), dtype=torch.float32, device=qkv.device)
    t = KERNEL_TIMER.start()
    check(lib().octic_attn_fwd_packed(_p(qkv), _p(o), _p(lse), B, H, T, c, qkv.stride(1), o.stride(1), float(scale),
                                      _stream(qkv)))
    KERNEL_TIMER.stop(t, "attn_fwd_kernel", 4 * B * T * 8 * c * 2, 4.0 * B * T * T * 8 * c)
    return o, lse


def attn_bwd_packed(qkv, o, dout, lse, H, c, scale, out=None):
    """-> dqkv packed [B,T,3*8c] (dq | dk | dv in the layout of qkv)."""
    B, T = qkv.shape[0], qkv.shape[1]
    dqkv = out if out is not None else torch.empty_like(qkv)
    delta = torch.empty((B, H, T), dtype=torch.float32, device=qkv.device)
    for phase, name, nbytes, flops in _attn_bwd_phases(T, 8 * (c // H)):
        t = KERNEL_TIMER.start()
        check(lib().octic_attn_bwd_packed(_p(qkv), _p(o), _p(dout), _p(lse), _p(delta), _p(dqkv), B, H, T, c, qkv.stride(1),
                                          o.stride(1), dqkv.stride(1), float(scale), phase, _stream(qkv)))
        KERNEL_TIMER.stop(t, name, nbytes * B * T * 8 * c * 2, flops * B * T * T * 8 * c)
    return dqkv


def handoff_cat_fwd(x, c, out_dtype):
    M = x.numel() // (8 * c)
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    xv = pview(x, c)
    check(lib().octic_handoff_cat_fwd(ctypes.byref(xv), _p(y), M, c, dt_code(out_dtype), _stream(x)))
    return y


def handoff_cat_bwd(dd, c):
    dd = dd.contiguous().float()
    M = dd.numel() // (8 * c)
    dx = torch.empty_like(dd)
    dv = pview(dx, c)
    check(lib().octic_handoff_cat_bwd(_p(dd), ctypes.byref(dv), M, c, _stream(dd)))
    return dx


def power_spectrum_fwd(x, c, out_dtype):
    M = x.numel() // (8 * c)
    y = torch.empty(x.shape[:-1] + (6 * c,), dtype=out_dtype, device=x.device)
    xv = pview(x, c)
    check(lib().octic_power_spectrum_fwd(ctypes.byref(xv), _p(y), M, c, dt_code(out_dtype), _stream(x)))
    return y


def power_spectrum_bwd(dd, x, c):
    dd = dd.contiguous().float()
    M = x.numel() // (8 * c)
    dx = torch.empty_like(x)
    xv, dv = pview(x, c), pview(dx, c)
    check(lib().octic_power_spectrum_bwd(_p(dd), ctypes.byref(xv), ctypes.byref(dv), M, c, _stream(x)))
    return dx


def im2col(img, p, Kpad, dtype):
    B, Cin, Hh, Ww = img.shape
    img = img.contiguous().float()
    rows = B * (Hh // p) * (Ww // p)
    out = torch.empty((rows, Kpad), dtype=dtype, device=img.device)
    check(lib().octic_im2col_patches(_p(img), _p(out), B, Cin, Hh, Ww, p, Kpad, dt_code(dtype), _stream(img)))
    return out


def lift_gemm(patches, w, bias_full, pos, out, B, n_patches, tok0, Kpad, D):
    check(lib().octic_lift_gemm(_p(patches), _p(w), _p(bias_full), _p(pos), _p(out), B, n_patches, tok0, Kpad, D,
                                dt_code(patches.dtype), _stream(patches)))


def lift_wgrad(patches, dout, Kpad, D):
    L = lib()
    rows = patches.shape[0]
    splits = max(1, min(16, rows // 512))
    ws = torch.empty(L.octic_lift_wgrad_workspace_bytes(Kpad, D, splits) // 4, dtype=torch.float32, device=patches.device)
    dw = torch.empty((D, Kpad), dtype=torch.float32, device=patches.device)
    check(L.octic_lift_wgrad(_p(patches), _p(dout), _p(dw), _p(ws), splits, rows, Kpad, D, dt_code(patches.dtype),
                             _stream(patches)))
    return dw


# ------------------------------------------------------------------------------------------ standard half
class _FinishJob(ctypes.Structure):
    _fields_ = [("partials", ctypes.c_void_p), ("out0", ctypes.c_void_p), ("out1", ctypes.c_void_p),
                ("scale1", ctypes.c_void_p), ("nblocks", ctypes.c_int), ("d", ctypes.c_int)]


class _LnFinishJob(ctypes.Structure):
    _fields_ = [("partials", ctypes.c_void_p), ("dalpha", ctypes.c_void_p * 5), ("dbeta", ctypes.c_void_p),
                ("nblk", ctypes.c_int), ("c", ctypes.c_int)]


class _WgFinishJob(ctypes.Structure):
    _fields_ = [("workspace", ctypes.c_void_p), ("w32", ctypes.c_void_p * 5), ("cs", ctypes.c_void_p * 5),
                ("bias", ctypes.c_void_p), ("dysum", ctypes.c_void_p), ("dw", ctypes.c_void_p * 5),
                ("dcs", ctypes.c_void_p * 5), ("dbias", ctypes.c_void_p),
                ("splits", ctypes.c_int), ("cin", ctypes.c_int), ("cout", ctypes.c_int), ("has_cs", ctypes.c_int)]


class _DeferredFinishes:
    """Parameter-gradient slab reductions (octic_dense_finish) postponed to the end of the running backward pass and issued
    as ONE batched launch (octic_dense_finish_batch: same summation order, bit-identical results).  Only the caller knows that
    nothing reads those gradients earlier (no gradient accumulation into an existing .grad - which includes a parameter used
    twice in one pass: autograd adds its second gradient to the first at once -, no DDP bucket hooks, no tensor hooks):
    `train.Trainer` switches this on for its single-GPU, single-micro-batch step of a model whose parameters each enter the
    graph once; the default is immediate launches."""

    def __init__(self):
        self.enabled = False
        self.slabs_too = True   # also postpone the octic weight-gradient slab reductions (their 50 MB slabs stay allocated
                                # until the end of the pass: off when the launch shapes vary from step to step)
        self.jobs = []          # (partials, nblk, d, out0_ptr, out1_ptr, scale1, keep-alive tensors, stream)
        self.ln_jobs = []       # (partials, nblk, c, [5 dalpha ptrs], dbeta_ptr, keep-alive storages, stream)
        self.wg_jobs = []       # (filled _WgFinishJob, keep-alive tensors / storages, stream)
        self.pairs = []         # functional.WgradPair objects holding a postponed weight gradient (see there)
        self.allow_pairs = True  # False: weight gradients are never postponed (they are read early: DDP's reducer hooks)
        self.armed = False

    def add_pair(self, pair):
        self.pairs.append(pair)
        if not self.armed:
            self.armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)

    def add(self, partials, nblk, d, out0_ptr, out1_ptr, scale1, keep, stream):
        self.jobs.append((partials, nblk, d, out0_ptr, out1_ptr, scale1, keep, stream))
        if not self.armed:
            self.armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)

    def add_ln(self, partials, nblk, c, dal, dbeta, stream):
        keep = tuple(t.untyped_storage() for t in list(dal) + [dbeta] if t is not None)
        self.ln_jobs.append((partials, nblk, c, [t.data_ptr() if t is not None else None for t in dal],
                             dbeta.data_ptr() if dbeta is not None else None, keep, stream))
        if not self.armed:
            self.armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)

    def add_wg(self, ws, splits, cin, cout, w32, cs5, bias, dysum, dw, dcs, dbias, stream, landed=()):
        j = _WgFinishJob()
        j.workspace = ws.data_ptr()
        j.has_cs = 1 if cs5 is not None else 0
        for k in range(5):
            j.w32[k] = w32[k].data_ptr() if cs5 is not None else None
            j.cs[k] = cs5[k].data_ptr() if cs5 is not None else None
            j.dw[k] = dw[k].data_ptr()
            j.dcs[k] = dcs[k].data_ptr() if dcs is not None else None
        j.bias = bias.data_ptr() if bias is not None else None
        j.dysum = dysum.data_ptr() if dysum is not None else None
        j.dbias = dbias.data_ptr() if dbias is not None else None
        j.splits, j.cin, j.cout = splits, cin, cout
        outs = list(dw) + (list(dcs) if dcs is not None else []) + ([dbias] if dbias is not None else [])
        keep = (ws, w32 if cs5 is not None else None, cs5, bias, dysum, tuple(t.untyped_storage() for t in outs))
        self.wg_jobs.append((j, keep, stream, tuple(landed)))
        if not self.armed:
            self.armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)

    def flush(self):
        pairs, self.pairs = self.pairs, []
        for pr in pairs:                                # a postponed weight gradient whose partner never came: on its own now
            pr.flush()
        jobs, self.jobs, self.armed = self.jobs, [], False
        ln_jobs, self.ln_jobs = self.ln_jobs, []
        wg_jobs, self.wg_jobs = self.wg_jobs, []
        if wg_jobs:
            arr = (_WgFinishJob * len(wg_jobs))(*[j[0] for j in wg_jobs])
            t = KERNEL_TIMER.start()
            check(lib().octic_linear_d8_wgrad_finish_batch(ctypes.cast(arr, ctypes.c_void_p), len(wg_jobs), wg_jobs[0][2]))
            KERNEL_TIMER.stop(t, "wgrad_finish_batch_kernel", 0)
            for j in wg_jobs:                           # gradients that went into registered destinations are there now
                grad_written(*j[3])
        if ln_jobs:
            arr = (_LnFinishJob * len(ln_jobs))()
            for i, (partials, nblk, c, dal, dbeta, _keep, _s) in enumerate(ln_jobs):
                arr[i].partials = partials.data_ptr()
                for k in range(5):
                    arr[i].dalpha[k] = dal[k]
                arr[i].dbeta = dbeta
                arr[i].nblk, arr[i].c = nblk, c
            t = KERNEL_TIMER.start()
            check(lib().octic_layernorm_d8_bwd_finish_batch(ctypes.cast(arr, ctypes.c_void_p), len(ln_jobs), ln_jobs[0][6]))
            KERNEL_TIMER.stop(t, "ln_bwd_finish_batch_kernel", sum(j[1] * 2 * 8 * j[2] * 4 for j in ln_jobs))
        if not jobs:
            return
        arr = (_FinishJob * len(jobs))()
        for i, (partials, nblk, d, o0, o1, sc, _keep, _s) in enumerate(jobs):
            arr[i].partials = partials.data_ptr()
            arr[i].out0 = o0
            arr[i].out1 = o1
            arr[i].scale1 = sc.data_ptr() if sc is not None else None
            arr[i].nblocks = nblk
            arr[i].d = d
        t = KERNEL_TIMER.start()
        check(lib().octic_dense_finish_batch(ctypes.cast(arr, ctypes.c_void_p), len(jobs), jobs[0][7]))
        KERNEL_TIMER.stop(t, "dense_finish_batch_kernel", sum(j[1] * 2 * j[2] * 4 for j in jobs))


DEFERRED_FINISHES = _DeferredFinishes()


def _in_backward():
    try:
        return torch._C._current_graph_task_id() != -1
    except Exception:
        return False


def _finish(partials, nblk, d, out0, out1, scale1, stream, out1_ptr=None):
    """out0[j] = sum_b partials[b][0][j], out1[j] = scale1[j] * sum_b partials[b][1][j] - now, or (see _DeferredFinishes) at the
    end of the running backward pass.  out1_ptr: raw address for an out1 that is the second half of out0's storage."""
    o0 = out0.data_ptr() if out0 is not None else None
    o1 = out1_ptr if out1_ptr is not None else (out1.data_ptr() if out1 is not None else None)
    if DEFERRED_FINISHES.enabled and _in_backward():
        # keep the outputs' STORAGE alive, not the tensors: a second reference to the tensor would make AccumulateGrad clone
        # the (not yet written) gradient instead of adopting it
        keep = tuple(t.untyped_storage() for t in (out0, out1) if t is not None)
        DEFERRED_FINISHES.add(partials, nblk, d, o0, o1, scale1, keep, stream)
        return
    check(lib().octic_dense_finish(_p(partials), nblk, d, ctypes.c_void_p(o0) if o0 else None,
                                   ctypes.c_void_p(o1) if o1 else None, _p(scale1), stream))


def dense_layernorm_fwd(x, w, b, eps, out_dtype):
    """x: f32 [..., d] contiguous -> (y out_dtype, stats [rows, 2] f32 = (mean, rstd))."""
    _require_cuda(x)
    d = x.shape[-1]
    rows = x.numel() // d
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
    t = KERNEL_TIMER.start()
    check(lib().octic_dense_layernorm_fwd(_p(x), _p(y), dt_code(out_dtype), _p(w), _p(b), _p(stats), rows, d, float(eps),
                                          _stream(x)))
    KERNEL_TIMER.stop(t, f"dense_ln_fwd_kernel<{_DTN[out_dtype]}>", rows * d * (4 + y.element_size()))
    return y, stats


def _check_rowmap(rowmap, rows):
    if rowmap.dtype != torch.int32 or not rowmap.is_contiguous() or rowmap.numel() != rows:
        raise ValueError("row map: a contiguous int32 tensor with one entry per compact row")


def dense_layernorm_fwd_rows(x, rowmap, w, b, eps, out_dtype):
    """LayerNorm of rows rowmap[r] of the f32 stream x [..., d] -> (y [1, rows, d] out_dtype, stats, xa = those rows, compact)."""
    _require_cuda(x)
    d = x.shape[-1]
    rows = rowmap.numel()
    _check_rowmap(rowmap, rows)
    y = torch.empty((1, rows, d), dtype=out_dtype, device=x.device)
    xa = torch.empty((1, rows, d), dtype=torch.float32, device=x.device)
    stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
    t = KERNEL_TIMER.start()
    check(lib().octic_dense_layernorm_fwd_rows(_p(x), _p(y), dt_code(out_dtype), _p(w), _p(b), _p(stats), rows, d, float(eps),
                                               _p(rowmap), _p(xa), _stream(x)))
    KERNEL_TIMER.stop(t, f"dense_ln_fwd_kernel<{_DTN[out_dtype]},rows>", rows * d * (8 + y.element_size()))
    return y, stats, xa


def dense_layernorm_bwd_rows_(gy, xa, w, stats, g, rowmap, want_param_grads=True):
    """In place on the stream's cotangent g: g[rowmap[r]] = LN'(gy[r]; xa[r]) + g[rowmap[r]].  Returns (dw, db)."""
    d = xa.shape[-1]
    rows = rowmap.numel()
    _check_rowmap(rowmap, rows)
    nblk = lib().octic_dense_blocks(rows)
    partials = torch.empty((nblk, 2, d), dtype=torch.float32, device=xa.device) if want_param_grads else None
    t = KERNEL_TIMER.start()
    check(lib().octic_dense_layernorm_bwd_rows(_p(gy), dt_code(gy.dtype), _p(xa), _p(w), _p(stats), _p(g), _p(g), _p(partials),
                                               rows, d, _p(rowmap), _stream(xa)))
    KERNEL_TIMER.stop(t, f"dense_ln_bwd_kernel<{_DTN[gy.dtype]},rows>", rows * d * (gy.element_size() + 12))
    if not want_param_grads:
        return None, None
    dw = torch.empty(d, dtype=torch.float32, device=xa.device)
    db = torch.empty(d, dtype=torch.float32, device=xa.device)
    _finish(partials, nblk, d, dw, db, None, _stream(xa))
    return dw, db


def dense_resid_layernorm_fwd(x, yb, gamma, rs, rps, w, b, eps, out_dtype):
    """xout = x + rs[row // rps] * gamma * yb ; y = LayerNorm(xout) in one row pass -> (xout f32, y out_dtype, stats)."""
    _require_cuda(x)
    d = x.shape[-1]
    rows = x.numel() // d
    xout = torch.empty_like(x)
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
    t = KERNEL_TIMER.start()
    check(lib().octic_dense_resid_layernorm_fwd(_p(x), _p(yb), dt_code(yb.dtype), _p(gamma), _p(rs), int(rps), _p(xout), _p(y),
                                                dt_code(out_dtype), _p(w), _p(b), _p(stats), rows, d, float(eps), _stream(x)))
    KERNEL_TIMER.stop(t, f"dense_resid_ln_fwd_kernel<{_DTN[out_dtype]}>", rows * d * (8 + yb.element_size() + y.element_size()))
    return xout, y, stats


def dense_layernorm_bwd(gy, x, w, stats, dres, want_param_grads=True):
    """Returns (dx f32 = LN'(gy) + dres, dw, db)."""
    d = x.shape[-1]
    rows = x.numel() // d
    dx = torch.empty_like(x)
    nblk = lib().octic_dense_blocks(rows)
    partials = torch.empty((nblk, 2, d), dtype=torch.float32, device=x.device) if want_param_grads else None
    t = KERNEL_TIMER.start()
    check(lib().octic_dense_layernorm_bwd(_p(gy), dt_code(gy.dtype), _p(x), _p(w), _p(stats), _p(dres), _p(dx),
                                          _p(partials), rows, d, _stream(x)))
    KERNEL_TIMER.stop(t, f"dense_ln_bwd_kernel<{_DTN[gy.dtype]}>",
                      rows * d * (gy.element_size() + 8 + (4 if dres is not None else 0)))
    if not want_param_grads:
        return dx, None, None
    dw = torch.empty(d, dtype=torch.float32, device=x.device)
    db = torch.empty(d, dtype=torch.float32, device=x.device)
    _finish(partials, nblk, d, dw, db, None, _stream(x))
    return dx, dw, db


def dense_ln_bwd_tail_ok(gy, yb, d):
    """Shapes of octic_dense_layernorm_bwd_tail: bf16 cotangent and branch, rows of 256, 512, ... 1280 columns."""
    return gy.dtype == torch.bfloat16 and yb.dtype == torch.bfloat16 and d % 256 == 0 and d <= 1280


def dense_layernorm_bwd_tail(gy, x, w, stats, dres, yb, gamma, rs, rps, want_param_grads=True, want_gamma=True,
                             want_colsum=True):
    """dense_layernorm_bwd followed by scale_residual_bwd on its result, one row pass.
    Returns (dx f32, dw, db, gyb bf16 = rs*gamma*dx, dgamma, gamma * colsum(rs*dx))."""
    d = x.shape[-1]
    rows = x.numel() // d
    dx = torch.empty_like(x)
    gyb = torch.empty(x.shape, dtype=yb.dtype, device=x.device)
    nblk = lib().octic_dense_blocks(rows)
    p1 = torch.empty((nblk, 2, d), dtype=torch.float32, device=x.device) if want_param_grads else None
    want2 = want_gamma or want_colsum
    p2 = torch.empty((nblk, 2, d), dtype=torch.float32, device=x.device) if want2 else None
    t = KERNEL_TIMER.start()
    check(lib().octic_dense_layernorm_bwd_tail(_p(gy), _p(x), _p(w), _p(stats), _p(dres), _p(dx), _p(p1), _p(yb), _p(gamma),
                                               _p(rs), int(rps), _p(gyb), _p(p2), rows, d, _stream(x)))
    KERNEL_TIMER.stop(t, "dense_ln_bwd_tail_kernel<bf16>", rows * d * (2 + 8 + (4 if dres is not None else 0) + 4))
    dw = db = dgamma = colsum = None
    if want_param_grads:
        dw = torch.empty(d, dtype=torch.float32, device=x.device)
        db = torch.empty(d, dtype=torch.float32, device=x.device)
        _finish(p1, nblk, d, dw, db, None, _stream(x))
    if want2:
        dgamma = torch.empty(d, dtype=torch.float32, device=x.device) if want_gamma else None
        colsum = torch.empty(d, dtype=torch.float32, device=x.device) if want_colsum else None
        _finish(p2, nblk, d, dgamma, colsum, gamma, _stream(x))
    return dx, dw, db, gyb, dgamma, colsum


def _rows2d(t):
    """[..., K] with contiguous rows -> (2-D view, rows, K, row stride)."""
    K = t.shape[-1]
    t2 = t.reshape(-1, K)
    if t2.stride(1) != 1 or (t2.shape[0] > 1 and t2.stride(0) % 8):
        t2 = t2.contiguous()
    return t2, t2.shape[0], K, (t2.stride(0) if t2.shape[0] > 1 else K)


def softmax_center(t, center, inv_temp):
    """softmax((t - center) * inv_temp) over the last dim, f32 (t f32 / bf16 [..., K], center f32 [K] or None)."""
    _require_cuda(t)
    t2, rows, K, ld = _rows2d(t)
    out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
    c = None if center is None else center.reshape(-1).float().contiguous()
    tk = KERNEL_TIMER.start()
    check(lib().octic_softmax_center(_p(t2), dt_code(t2.dtype), ld, _p(c), float(inv_temp), _p(out), rows, K, _stream(t)))
    KERNEL_TIMER.stop(tk, f"softmax_center_kernel<{_DTN[t2.dtype]}>", rows * K * (2 * t2.element_size() + 4))
    return out


def soft_ce_fwd(s, tprob, inv_temp):
    """Per row r of s [N, K]: -sum_k t_k log_softmax(s_r * inv_temp)_k with t = tprob[r % Nt] (tprob f32 [Nt, K] contiguous).
    Returns (loss [N], lse [N], tsum [N]) f32."""
    _require_cuda(s)
    s2, rows, K, ld = _rows2d(s)
    if tprob.dtype != torch.float32 or not tprob.is_contiguous() or tprob.shape[-1] != K:
        raise ValueError("soft_ce_fwd: tprob must be a contiguous f32 [Nt, K] tensor")
    nt = tprob.numel() // K
    loss, lse, tsum = (torch.empty(rows, dtype=torch.float32, device=s.device) for _ in range(3))
    tk = KERNEL_TIMER.start()
    check(lib().octic_soft_ce_fwd(_p(s2), dt_code(s2.dtype), ld, _p(tprob), nt, float(inv_temp), _p(loss), _p(lse), _p(tsum),
                                  rows, K, _stream(s)))
    KERNEL_TIMER.stop(tk, f"soft_ce_fwd_kernel<{_DTN[s2.dtype]}>", rows * K * (s2.element_size() + 4))
    return loss, lse, tsum


def soft_ce_bwd(s, tprob, inv_temp, g, lse, tsum):
    """d loss / d s for soft_ce_fwd, in s's dtype ([N, K] contiguous)."""
    s2, rows, K, ld = _rows2d(s)
    nt = tprob.numel() // K
    ds = torch.empty((rows, K), dtype=s2.dtype, device=s.device)
    g = g.reshape(-1).float().contiguous()
    tk = KERNEL_TIMER.start()
    check(lib().octic_soft_ce_bwd(_p(s2), dt_code(s2.dtype), ld, _p(tprob), nt, float(inv_temp), _p(g), _p(lse), _p(tsum),
                                  _p(ds), K, rows, K, _stream(s)))
    KERNEL_TIMER.stop(tk, f"soft_ce_bwd_kernel<{_DTN[s2.dtype]}>", rows * K * (2 * s2.element_size() + 4))
    return ds.view(s.shape)


def scale_residual_fwd(x, y, gamma, rs, rps):
    """out = x + rs[row // rps] * gamma * y   (x f32, y f32/bf16, same shape [..., d])."""
    _require_cuda(x)
    d = x.shape[-1]
    rows = x.numel() // d
    out = torch.empty_like(x)
    t = KERNEL_TIMER.start()
    check(lib().octic_scale_residual_fwd(_p(x), _p(y), dt_code(y.dtype), _p(gamma), _p(rs), int(rps), _p(out), rows, d,
                                         _stream(x)))
    KERNEL_TIMER.stop(t, f"scale_residual_fwd_kernel<{_DTN[y.dtype]}>", rows * d * (8 + y.element_size()))
    return out


def scale_residual_fwd_rows_(stream, rowmap, x, y, gamma, rs, rps):
    """In place on the f32 stream: stream[rowmap[r]] = x[r] + rs[r // rps] * gamma * y[r]  (x, y compact)."""
    _require_cuda(x)
    d = x.shape[-1]
    rows = rowmap.numel()
    _check_rowmap(rowmap, rows)
    t = KERNEL_TIMER.start()
    check(lib().octic_scale_residual_fwd_rows(_p(x), _p(y), dt_code(y.dtype), _p(gamma), _p(rs), int(rps), _p(stream), rows, d,
                                              _p(rowmap), _stream(x)))
    KERNEL_TIMER.stop(t, f"scale_residual_fwd_kernel<{_DTN[y.dtype]},rows>", rows * d * (8 + y.element_size()))
    return stream


def scale_residual_bwd(gout, y, gamma, rs, rps, want_gamma=True, want_colsum=True, rowmap=None):
    """Returns (gy in y's dtype, dgamma, gamma * colsum(rs*gout) = bias gradient of the producer of y).  rowmap: the compact
    rows of y are rows rowmap[r] of gout (the cotangent of a stream of which the branch saw a subset)."""
    d = gout.shape[-1]
    rows = gout.numel() // d if rowmap is None else rowmap.numel()
    if rowmap is not None:
        _check_rowmap(rowmap, rows)
    gy = torch.empty(gout.shape if rowmap is None else y.shape, dtype=y.dtype, device=gout.device)
    nblk = lib().octic_dense_blocks(rows)
    want = want_gamma or want_colsum
    partials = torch.empty((nblk, 2, d), dtype=torch.float32, device=gout.device) if want else None
    t = KERNEL_TIMER.start()
    check(lib().octic_scale_residual_bwd_rows(_p(gout), _p(y), dt_code(y.dtype), _p(gamma), _p(rs), int(rps), _p(gy),
                                              _p(partials), rows, d, _p(rowmap), _stream(gout)))
    KERNEL_TIMER.stop(t, f"scale_residual_bwd_kernel<{_DTN[y.dtype]}>", rows * d * (4 + 2 * y.element_size()))
    if not want:
        return gy, None, None
    dgamma = torch.empty(d, dtype=torch.float32, device=gout.device) if want_gamma else None
    colsum = torch.empty(d, dtype=torch.float32, device=gout.device) if want_colsum else None
    _finish(partials, nblk, d, dgamma, colsum, gamma, _stream(gout))
    return gy, dgamma, colsum


def dense_gelu_bwd(h, g, want_colsum=True):
    """dh = gelu'(h) * g for bf16 [..., d]; also the column sums of dh (f32 [d]) when asked."""
    _require_cuda(h)
    d = h.shape[-1]
    rows = h.numel() // d
    dh = torch.empty_like(h)
    nblk = lib().octic_dense_gelu_blocks()
    partials = torch.empty((nblk, d), dtype=torch.float32, device=h.device) if want_colsum else None
    t = KERNEL_TIMER.start()
    check(lib().octic_dense_gelu_bwd(_p(h), _p(g), _p(dh), _p(partials), rows, d, _stream(h)))
    KERNEL_TIMER.stop(t, "dense_gelu_bwd_kernel", rows * d * 6)
    if not want_colsum:
        return dh, None
    out = torch.empty(d, dtype=torch.float32, device=h.device)
    half = d // 2
    _finish(partials, nblk, half, out, None, None, _stream(h), out1_ptr=out.data_ptr() + 4 * half)
    return dh, out


# ------------------------------------------------------------------------------------------ dense MFMA GEMMs
_DG_WS = {}


def _dense_ws(M, N, K, dev):
    """Split-K workspace of one (M,N,K) problem, cached per stream-ordered use (launches on one stream are serial)."""
    key = (M, N, K, dev)
    ws = _DG_WS.get(key)
    if ws is None:
        ws = _DG_WS[key] = torch.zeros(int(lib().octic_dense_gemm_workspace_bytes(M, N, K)), dtype=torch.uint8, device=dev)
    return ws


def dense_colsum(g):
    """f32 column sums of a bf16 [rows, d] tensor (unit column stride): the bias gradient of a dense nn.Linear."""
    _require_cuda(g)
    rows, d = g.shape
    if g.stride(1) != 1:
        raise ValueError("dense_colsum: rows must be contiguous along d")
    nblk = lib().octic_dense_gelu_blocks()
    partials = torch.empty((nblk, d), dtype=torch.float32, device=g.device)
    t = KERNEL_TIMER.start()
    check(lib().octic_dense_colsum(_p(g), rows, d, g.stride(0), _p(partials), _stream(g)))
    KERNEL_TIMER.stop(t, "dense_colsum_kernel", rows * d * 2)
    out = torch.empty(d, dtype=torch.float32, device=g.device)
    half = d // 2
    _finish(partials, nblk, half, out, None, None, _stream(g), out1_ptr=out.data_ptr() + 4 * half)
    return out


def dense_plan(M, N, K, mode, tokens):
    """(tile width, colsum slab rows, per-image panels?, main-launch workgroups) of octic_dense_gemm_nt_tokens."""
    out = (ctypes.c_int * 4)()
    check(lib().octic_dense_gemm_plan(M, N, K, mode, int(tokens), out))
    return out[0], out[1], bool(out[2]), out[3]


def dense_gemm_nt(a, b, mode=0, bias=None, gamma=None, rs=None, rps=1, x=None, h=None, name=None, want_colsum=False,
                  tokens=0):
    """C[M,N] = a[M,K] @ b[N,K]^T on the hand-written MFMA kernel (csrc/dense_gemm.hip) with a fused tail:
    mode 0 -> c ; 1 -> (c, gelu(c)) ; 2 -> (c, x + rs*gamma*c) ; 3 -> gelu'(h) * c (want_colsum: also the f32 column
    sums of that result) ; 4 -> (gelu'(c), gelu(c)) ; 5 -> h * c with h = the factor of mode 4 (want_colsum as 3) ;
    6 -> gelu(c) only.
    a, b bf16 2-D, K contiguous.  tokens: the rows are whole images of that many tokens ([B, tokens, K] flattened) - 257 lets the
    launch use per-image row panels + the class-token kernel (octic_dense_gemm_nt_tokens); 0 = unknown."""
    _require_cuda(a)
    M, K = a.shape
    N = b.shape[0]
    if a.stride(1) != 1 or b.stride(1) != 1 or b.shape[1] != K:
        raise ValueError("dense_gemm_nt: operands must be [M,K] / [N,K] with contiguous K")
    c = torch.empty((M, N), dtype=torch.bfloat16, device=a.device)
    c2 = torch.empty_like(c) if mode in (1, 4) else None
    out = torch.empty((M, N), dtype=torch.float32, device=a.device) if mode == 2 else None
    ws = _dense_ws(M, N, K, a.device)
    tokens = int(tokens) if (tokens and M % int(tokens) == 0) else 0
    cs_rows = dense_plan(M, N, K, mode, tokens)[1] if (mode in (3, 5) and want_colsum) else 0
    cs = torch.empty((cs_rows, N), dtype=torch.float32, device=a.device) if cs_rows else None
    t = KERNEL_TIMER.start()
    check(lib().octic_dense_gemm_nt_tokens(_p(a), _p(b), M, N, K, a.stride(0), b.stride(0), mode, _p(c), _p(c2), N, _p(bias),
                                           _p(gamma), _p(rs), int(rps), _p(x), _p(out), _p(h), _p(cs), _p(ws), tokens,
                                           _stream(a)))
    if t is not None:
        nb = 2 * (M * K + N * K + M * N * (2 if mode in (1, 3, 4, 5) else 1)) + (8 * M * N if mode == 2 else 0)
        # "@320": the launch ran the 256 x 320 tile (kernel symbol dense_nt_kernel<0, 5>), else <mode, 4>
        # (with per-image panels the timed interval also holds the class-token launch behind the panels')
        wide = "@320" if dense_plan(M, N, K, mode, tokens)[0] == 320 else ""
        KERNEL_TIMER.stop(t, (name or f"dense_nt_kernel<{mode}>") + wide, nb, 2.0 * M * N * K)
    if mode in (1, 4):
        return c, c2
    if mode == 2:
        return c, out
    if cs is not None:
        colsum = torch.empty(N, dtype=torch.float32, device=a.device)
        _finish(cs, cs_rows, N // 2, colsum, None, None, _stream(a), out1_ptr=colsum.data_ptr() + 2 * N)
        return c, colsum
    return c
