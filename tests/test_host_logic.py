"""CPU-only checks of the product's host logic (no GPU compute): index-table unfolding, group tables,
transforms, invariants' composite maps, model construction / state_dict compatibility."""
import os
import zlib

import numpy as np
import pytest
import torch

import cases
from octic_vits_amd import d8_utils as U

GOLD = os.path.join(os.path.dirname(__file__), "golden")


class _NS:
    """product namespace for the function-level cases (expand_weight needs the conv class)"""
    pass


def _product_ns():
    import octic_vits_amd.d8_layers as L
    ns = _NS()
    for mod in (U, L):
        for k, v in vars(mod).items():
            if not k.startswith("_"):
                setattr(ns, k, v)
    return ns


@pytest.mark.parametrize("name", ["transforms", "pos_unfold", "expand_weight", "converters", "group_actions"])
def test_function_cases_match_reference_golden(name):
    got = cases.run_func_case(_product_ns(), name)
    want = np.load(os.path.join(GOLD, name + ".npz"))
    assert set(got) == set(want.files)
    for k in want.files:
        np.testing.assert_allclose(got[k], want[k], rtol=1e-6, atol=1e-6, err_msg=f"{name}:{k}")


def test_mult_table_known_answers():
    t = {(a, b): c for a, b, c in U.mult_table}
    assert len(t) == 49
    assert t[("r", "m")] == "mrrr" and t[("m", "r")] == "mr" and t[("mr", "mr")] == "e" and t[("rrr", "mrrr")] == "m"


@pytest.mark.parametrize("name", ["inv_linear", "inv_polynomial", "inv_thirdorder", "inv_maxfilter", "inv_canonization",
                                  "inv_noninvariant"])
def test_composite_invariants_match_reference_golden(name):
    import octic_vits_amd.d8_invariantization as I
    got = cases.run_module_case(I, name)
    want = np.load(os.path.join(GOLD, name + ".npz"))
    tol = 1e-3 if name == "inv_polynomial" else 1e-4
    for k in want.files:
        np.testing.assert_allclose(got[k], want[k], rtol=tol, atol=tol, err_msg=f"{name}:{k}")


def test_model_facts_and_state_dict_keys():
    from octic_vits_amd.deit_models import create_model
    facts = np.load(os.path.join(GOLD, "model_facts.npz"))
    for mname in ("hybrid_deit_huge_patch14", "d8_inv_early_deit_huge_patch14",
                  "hybrid_deit_large_patch16", "d8_inv_early_deit_large_patch16"):
        with torch.device("meta"):
            m = create_model(mname, num_classes=1000)
        assert sum(p.numel() for p in m.parameters()) == int(facts[mname + ".params"][0])
        assert len(list(m.parameters())) == int(facts[mname + ".tensors"][0])
        keys = sorted(m.state_dict().keys())
        assert zlib.crc32("\n".join(keys).encode()) == int(facts[mname + ".keys_crc"][0])


def test_constructor_errors_match_reference():
    import octic_vits_amd.d8_layers as L
    with pytest.raises(ValueError):
        L.LinearD8(60, 64)
    with pytest.raises(ValueError):
        L.AffineD8(12)
    with pytest.raises(AssertionError):
        L.AttentionD8(dim=64, num_heads=3)
    with pytest.raises(NotImplementedError):
        L.AttentionD8(dim=64, num_heads=2, rope=object())
    with pytest.raises(NotImplementedError):
        L.LiftIrrepD8Conv2d(3, 8, 5, 5, bias=False)
    with pytest.raises(ValueError):
        L.LiftIrrepD8Conv2d(3, 8, 2, 2, bias=False, irrep="A2")


def test_product_refuses_cpu_execution():
    """No CPU fallback: running a hot op on CPU tensors fails loudly."""
    import octic_vits_amd.d8_layers as L
    lin = L.LinearD8(64, 64)
    xs = cases.tuple5("cpu", 2, 3, 8)
    with pytest.raises(RuntimeError, match="GPU only"):
        lin(xs)


def test_dinov2_entrypoints_keep_the_reference_state_dict():
    """OcticDinoVisionTransformer (dinov2_models.py:41-267): same parameter names / shapes as the oracle restatement
    (itself pinned to the real class by tests/golden/dino_*.npz), factories registered under the reference's names,
    hybrid_dinov2_vit_large_patch16 = 170,296,704 parameters (measured on the reference)."""
    import torch
    from functools import partial

    from oracle import octic_ref as R
    from octic_vits_amd import d8_layers, deit_models, dinov2_models, vit
    kw = dict(img_size=32, patch_size=4, embed_dim=64, depth=4, num_heads=2, num_register_tokens=2)
    a = dinov2_models.OcticDinoVisionTransformer(
        **kw, octic_block_layers=partial(d8_layers.NestedTensorBlockD8, init_values=1e-5),
        standard_block_layers=partial(vit.NestedTensorBlock, attn_class=vit.MemEffAttention, init_values=1e-5))
    b = R.OcticDinoVisionTransformer(**kw, octic_block_layers=partial(R.NestedTensorBlockD8, init_values=1e-5),
                                     standard_block_layers=partial(R.NestedTensorBlock, init_values=1e-5))
    sa, sb = a.state_dict(), b.state_dict()
    assert sorted(sa) == sorted(sb)
    assert all(sa[k].shape == sb[k].shape for k in sa)
    assert sorted(n for n, p in a.named_parameters() if p.requires_grad) == \
        sorted(n for n, p in b.named_parameters() if p.requires_grad)
    for name in ("hybrid_dinov2_vit_large_patch16", "hybrid_dinov2_vit_huge_patch16",
                 "d8_inv_early_dinov2_vit_large_patch16", "d8_inv_early_dinov2_vit_huge_patch16"):
        assert name in deit_models._LOCAL_REGISTRY
    with torch.device("meta"):
        big = deit_models.create_model("hybrid_dinov2_vit_large_patch16")
    assert sum(p.numel() for p in big.parameters()) == 170296704


def test_drop_path_pool_is_armed_only_inside_a_model_forward_and_is_seed_reproducible():
    """d8_layers._drop_path_mask: outside a model forward every call draws on its own (reference behaviour,
    octic_vits/d8_layers.py:140-152); inside (pool armed by OcticVisionTransformer.forward_features) masks come 64 at a
    time from one draw, a new draw per forward, identical for identical seeds, and scaled by 1 / keep."""
    import torch
    from octic_vits_amd import d8_layers as L
    torch.manual_seed(3)
    a = L._drop_path_mask(8, 0.25, torch.device("cpu"))
    torch.manual_seed(3)
    b = L._drop_path_mask(8, 0.25, torch.device("cpu"))
    assert torch.equal(a, b) and set(a.tolist()) <= {0.0, float(torch.tensor(1.0) / 0.75)}
    runs = []
    for _ in range(2):
        torch.manual_seed(11)
        L.arm_drop_path_pool(True)
        try:
            runs.append(torch.stack([L._drop_path_mask(8, 0.25, torch.device("cpu")) for _ in range(70)]))
        finally:
            L.arm_drop_path_pool(False)
    assert torch.equal(runs[0], runs[1])                      # same seed -> same masks, across the pool refill at 64
    assert len({tuple(r.tolist()) for r in runs[0]}) > 8      # rows of the pool differ
    assert not L._pool_armed


def test_packed_lift_weight_equals_the_composed_construction():
    """LiftD8.packed_weight as one signed gather (d8_utils._LiftWeightFn) against the reference's composition
    (expand_lift_kernel per irrep, rot90 for the E rows, flatten, cat: octic_vits/d8_layers.py:329-411): bitwise equal
    forward (same products, same order of additions), gradients to 1e-6 (fixed-order gather instead of autograd's chain)."""
    import torch
    from octic_vits_amd.d8_layers import LiftD8
    torch.manual_seed(5)
    for cin, cout, p in ((3, 16, 4), (3, 24, 14), (2, 8, 2)):
        try:
            m = LiftD8(cin, cout, p, p, bias=True)
        except ValueError:
            continue                      # p = 2 has no A2 / B1 kernels
        fused, composed = m.packed_weight(), m.packed_weight_composed()
        assert fused.shape == composed.shape and torch.equal(fused, composed)
        g = torch.randn_like(fused)
        ws = [c.weight for c in (m.conv_A1, m.conv_A2, m.conv_B1, m.conv_B2, m.conv_E_left, m.conv_E_right)]
        ga = torch.autograd.grad(m.packed_weight(), ws, g)
        gb = torch.autograd.grad(m.packed_weight_composed(), ws, g)
        for a, b in zip(ga, gb):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-6)


def test_packed_pos_embed_equals_the_composed_construction():
    """packed_pos_embed as one signed row gather (d8_utils._PosEmbedFn) against the reference's composition
    (isotypic_dim_interpolation, octic_vits/d8_utils.py:388-451, + packing): bitwise forward, gradients to 1e-6."""
    import torch
    from octic_vits_amd import d8_utils as U
    torch.manual_seed(9)
    for h, c in ((2, 8), (8, 20), (7, 3)):
        ps = [torch.randn(h, h, c, requires_grad=True) for _ in range(6)]
        fused, composed = U.packed_pos_embed(ps), U.packed_pos_embed_composed(ps)
        assert fused.shape == composed.shape == (4 * h * h, 8 * c) and torch.equal(fused, composed)
        g = torch.randn_like(fused)
        for a, b in zip(torch.autograd.grad(U.packed_pos_embed(ps), ps, g), torch.autograd.grad(U.packed_pos_embed_composed(ps), ps, g)):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-6)


def test_drop_path_pool_is_bypassed_where_a_recompute_could_replay_the_block():
    """Round-3 advisor finding: a checkpoint recompute runs outside the armed window, so the forward must not have used
    the pooled draw.  Under torch.utils.checkpoint (both flavours) masks are drawn per call from torch's generator - whose
    state checkpoint saves and restores - so the recompute sees exactly the forward's mask."""
    import torch
    from torch.utils.checkpoint import checkpoint
    from octic_vits_amd import d8_layers as L
    seen = []

    def branch(x):
        m = L._drop_path_mask(x.shape[0], 0.5, x.device)
        seen.append(m.clone())
        return x * m.view(-1, 1)

    for reentrant in (False, True):
        seen.clear()
        x = torch.ones(16, 4, requires_grad=True)
        torch.manual_seed(21)
        L.arm_drop_path_pool(True)
        try:
            y = checkpoint(branch, x, use_reentrant=reentrant)
        finally:
            L.arm_drop_path_pool(False)
        y.sum().backward()                                    # recompute happens here, pool disarmed
        assert len(seen) == 2 and torch.equal(seen[0], seen[1]), reentrant
        assert torch.equal(x.grad, seen[0].view(-1, 1).expand(16, 4))
    # and without a checkpoint the armed pool is used (one [64, B] draw)
    L.arm_drop_path_pool(True)
    try:
        L._drop_path_mask(16, 0.5, torch.device("cpu"))
        assert any(k[1] == 16 for k in L._mask_pool)
    finally:
        L.arm_drop_path_pool(False)


def test_lift_weight_gather_is_exact_in_float64():
    """The gather factors of d8_utils._lift_tables_on are built in float64 (an f32 table cost a float64 caller 1e-9 in
    the forward, advisor r3): in float64 the fused weight equals the composed construction to the last bit."""
    import torch
    from octic_vits_amd.d8_layers import LiftD8
    torch.manual_seed(6)
    m = LiftD8(3, 16, 4, 4, bias=True).double()
    assert torch.equal(m.packed_weight(), m.packed_weight_composed())


def test_loss_watch_raises_two_steps_late_and_never_on_finite_losses():
    """train._LossWatch (deit/engine.py:67-71 without a per-step stream drain): the loss of step k is examined when step
    k + 2 is about to be issued; drain() looks at everything that is left."""
    import torch
    from octic_vits_amd.train import _LossWatch
    w = _LossWatch("cpu")
    for v in (0.7, 0.6, 0.5):
        w.check()
        w.push(torch.tensor(v))
    w.check()
    w.push(torch.tensor(float("nan")))       # step 3
    w.check()                                # looks at step 2: fine
    w.push(torch.tensor(0.4))                # step 4
    with pytest.raises(FloatingPointError):
        w.check()                            # looks at step 3
    w2 = _LossWatch("cpu")
    w2.push(torch.tensor(float("inf")))
    with pytest.raises(FloatingPointError):
        w2.drain()


# ------------------------------------------------------------------------------------------------ ring dispatch plan
def _ring_order(items, ksteps, slots=64):
    """csrc/gemm.hip plan_ring + ring_item_of through the host-only query octic_linear_d8_ring_order (no device call)."""
    import ctypes
    from octic_vits_amd import _lib
    raw = _lib.lib()
    n = sum(items)
    arr = lambda v: (ctypes.c_int * len(v))(*v)
    og, oi = (ctypes.c_int * n)(), (ctypes.c_int * n)()
    mode = raw.octic_linear_d8_ring_order(len(items), arr(items), arr(ksteps), slots, og, oi)
    return mode, list(zip(og, oi))


@pytest.mark.parametrize("items,ksteps", [
    ([514, 129, 129, 129, 129], [20, 10, 10, 10, 10]),       # ViT-H fc2 / dgrad fc1 at M = 16 448: six items past two rounds
    ([514, 129, 129, 129, 129], [15, 8, 8, 8, 8]),            # dgrad qkv
    ([394, 99, 99, 99, 99], [16, 8, 8, 8, 8]),                # ViT-L
    ([40, 10, 10, 10, 10], [20, 10, 10, 10, 10]),             # fewer items than slots
    ([7, 3, 3], [4, 2, 2]),                                   # tiny launch: the even spread
    ([100, 30, 20], [20, 10, 10]),                            # unequal short groups: the even spread
    ([9000, 2250, 2250, 2250, 2250], [20, 10, 10, 10, 10]),   # many rounds: no plan is computed, the even spread
    ([64], [10])])
def test_ring_dispatch_plan_runs_every_item_exactly_once(items, ksteps):
    """Whatever order the planner picks, the workgroups of a launch cover every (group, item) once."""
    mode, order = _ring_order(items, ksteps)
    assert mode in (0, 1)
    want = {(g, i) for g, n in enumerate(items) for i in range(n)}
    assert len(order) == len(want) and set(order) == want


def test_ring_dispatch_plan_at_vit_h_spreads_the_long_items():
    """ViT-H long-K launch: 1030 items on 8 x 64 slots.  XCDs with an odd item (129 workgroups) get at most 63 long items, so
    the slot that takes a third item is one that never ran a long one (csrc/gemm.hip plan_ring)."""
    items, ksteps = [514, 129, 129, 129, 129], [20, 10, 10, 10, 10]
    mode, order = _ring_order(items, ksteps)
    assert mode == 1
    n = len(order)
    for x in range(8):
        mine = [order[b] for b in range(x, n, 8)]
        longs = sum(1 for g, _ in mine if g == 0)
        if len(mine) == 129:
            assert longs <= 63, (x, longs)
        first_long = next(i for i, (g, _) in enumerate(mine) if g == 0)
        last_long = max(i for i, (g, _) in enumerate(mine) if g == 0)
        assert last_long - first_long + 1 == longs                # the long items of an XCD are one contiguous run


# ------------------------------------------------------------------------------------------------ compacted stochastic depth
def test_gather_scatter_pair_gives_the_gradients_of_the_plain_composition():
    """d8_layers._GatherRowsFn / _ScatterRowsFn (in-place stream, one cotangent tensor edited in place on the way back) against
    the same computation written with index_select / index_copy, in float64 on the CPU: a chain of four "blocks" whose
    branches see random subsets of the samples, a parameter per branch, and a loss that reads only row 0 of every sample (the
    cls-token situation: most rows of the last cotangent are zero)."""
    import octic_vits_amd.d8_layers as L
    torch.manual_seed(0)
    B, T, D = 7, 3, 5
    x0 = torch.randn(B, T, D, dtype=torch.float64)
    ws = [torch.randn(D, D, dtype=torch.float64) * 0.3 for _ in range(4)]
    subsets = [torch.tensor(s) for s in ([0, 2, 3, 6], [1], [0, 1, 2, 3, 4, 5, 6], [2, 5])]

    def branch(xa, w):                      # residual inside, as the fused branch functions have it
        return xa + torch.tanh(xa @ w) * 1.7

    def run(pair):
        x = x0.clone().requires_grad_(True)
        params = [w.clone().requires_grad_(True) for w in ws]
        s = x * 1.0                          # (the stream the blocks edit in place must not be a leaf)
        for idx, w in zip(subsets, params):
            if pair:
                link = L._RowLink()
                xa = L._GatherRowsFn.apply(s, idx, link)
                s = L._ScatterRowsFn.apply(s, idx, branch(xa, w), link)
            else:
                s = s.index_copy(0, idx, branch(s.index_select(0, idx), w))
        loss = (s[:, 0] ** 2).sum()
        loss.backward()
        return float(loss.detach()), x.grad.clone(), [p.grad.clone() for p in params]

    la, gxa, gpa = run(False)
    lb, gxb, gpb = run(True)
    assert la == lb
    assert torch.allclose(gxa, gxb, rtol=0, atol=1e-13)
    for a, b in zip(gpa, gpb):
        assert torch.allclose(a, b, rtol=0, atol=1e-13)


def test_compact_index_pool_matches_its_masks():
    """_compact_indices: a pool of host-drawn Bernoulli masks per forward; every branch gets the kept sample ids of the next mask
    (ascending), the count on the host, and a fresh draw once the pool is used up or a new forward arms it."""
    import octic_vits_amd.d8_layers as L
    torch.manual_seed(3)
    L._compact_pool.clear()
    dev = torch.device("cpu")
    L.arm_drop_path_pool(True)
    try:
        seen = []
        for _ in range(L.DROP_PATH_POOL + 3):
            idx, n = L._compact_indices(16, 0.5, dev)
            assert idx.dtype == torch.int64 and idx.numel() == n and 0 <= n <= 16
            assert torch.equal(idx, idx.sort().values) and len(set(idx.tolist())) == n
            seen.append(n)
        assert 4 < sum(seen) / len(seen) < 12                       # mean kept count 8
        L.arm_drop_path_pool(True)                                   # a new forward starts a new pool
        key = next(iter(L._compact_pool))
        assert L._compact_pool[key][2] >= L.DROP_PATH_POOL
        L._compact_indices(16, 0.5, dev)
        assert L._compact_pool[key][2] == 1
    finally:
        L.arm_drop_path_pool(False)
        L._compact_pool.clear()
    idx, n, scale = L._compact_plan(4, type("DP", (), {"drop_prob": 1.0, "scale_by_keep": True})(), dev)
    assert n == 1 and scale == 0.0 and idx.tolist() == [0]         # a mask that keeps nobody: one sample, scale 0


# ------------------------------------------------------------------------------------------------ trainer guards (round 5)
def _tiny_hybrid(**kw):
    from octic_vits_amd.d8_layers import Layer_scale_init_BlockD8
    from octic_vits_amd.model import OcticVisionTransformer
    from octic_vits_amd.vit import Layer_scale_init_Block
    args = dict(img_size=32, patch_size=16, num_classes=5, embed_dim=64, depth=4, num_heads=2, qkv_bias=True, init_scale=0.1,
                octic_block_layers=Layer_scale_init_BlockD8, standard_block_layers=Layer_scale_init_Block)
    args.update(kw)
    return OcticVisionTransformer(**args)


def test_segment_graph_trainer_refuses_what_its_graphs_cannot_follow():
    """Graphed slices read static compute-dtype weight copies that only FusedLamb refreshes (advisor, round 4): another
    optimizer must be refused, as Trainer.capture does; a model without an octic block in front of the hand-off and head
    dropout are rejected by the slicer instead of being computed differently from model.forward."""
    from octic_vits_amd.train import SegmentedModel, Trainer
    t = Trainer(_tiny_hybrid(), fused_optimizer=False, segment_graphs=2, tuned_gemms=False, device_type="cpu")
    with pytest.raises(RuntimeError, match="fused_optimizer=True"):
        t.capture_segments(torch.zeros(2, 3, 32, 32))
    with pytest.raises(ValueError, match="octic_equi_break_layer"):
        SegmentedModel(_tiny_hybrid(octic_equi_break_layer=0), 2)
    with pytest.raises(ValueError, match="dropout"):
        SegmentedModel(_tiny_hybrid(drop_rate=0.1), 2)


def test_trainer_finish_checks_the_losses_the_late_watch_has_not_seen():
    from octic_vits_amd.train import Trainer
    t = Trainer(_tiny_hybrid(), fused_optimizer=False, tuned_gemms=False, device_type="cpu")
    t._watch.push(torch.tensor(0.5))
    t.finish()                                   # finite: nothing happens
    t._watch.push(torch.tensor(float("inf")))    # the last step of a run: never reached by check()
    with pytest.raises(FloatingPointError):
        t.finish()


def test_ragged_row_bookkeeping():
    """ragged.Ragged (several crop sets in one row tensor): offsets, per-set views, per-sample -> per-row factors, the one-shot
    draws of a pass - every row map is a set of DISTINCT rows made of whole kept samples, set-major, in the order the compact
    tensor of the subset branch uses."""
    import torch
    from octic_vits_amd import ragged as R
    a, b = torch.arange(2 * 5 * 4.0).view(2, 5, 4), 100 + torch.arange(3 * 2 * 4.0).view(3, 2, 4)
    rows, rag = R.concat([a, b])
    assert rag.sets == [(2, 5, 0), (3, 2, 10)] and rag.rows == 16 and rag.samples == 5 and rows.shape == (1, 16, 4)
    assert rag.matches(rows) and not rag.matches(rows[:, :15])
    va, vb = rag.views(rows)
    assert torch.equal(va, a) and torch.equal(vb, b)
    per_sample = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0])
    assert rag.row_scale(per_sample).tolist() == [1.0] * 5 + [2.0] * 5 + [3.0] * 2 + [4.0] * 2 + [5.0] * 2
    assert rag.row_to_sample("cpu").tolist() == [0] * 5 + [1] * 5 + [2] * 2 + [3] * 2 + [4] * 2
    assert rag.const_row_scale([2.0, 1.5], "cpu").tolist() == [2.0] * 10 + [1.5] * 6
    torch.manual_seed(0)
    keeps = [1, 2]
    rag.draw_perms(4, "cpu", keeps)
    assert rag.rowmaps.shape == (4, 1 * 5 + 2 * 2) and rag.rowmaps.dtype == torch.int32
    for i in range(4):
        idxs, rowmap = rag.take_subset(keeps, "cpu")
        assert [len(ix) for ix in idxs] == keeps
        m = rowmap.tolist()
        assert len(set(m)) == len(m)
        want = [int(idxs[0][0]) * 5 + t for t in range(5)] + [10 + int(j) * 2 + t for j in idxs[1] for t in range(2)]
        assert m == want
    idxs, rowmap = rag.take_subset(keeps, "cpu")          # the pool is used up: fresh permutations, no map
    assert rowmap is None and [len(ix) for ix in idxs] == keeps
    rag.draw_perms(2, "cpu", keeps)
    assert rag.take_subset([2, 2], "cpu")[1] is None      # other keep counts than the maps were drawn for


def test_ragged_mask_pool_statistics():
    import torch
    from octic_vits_amd import ragged as R
    from octic_vits_amd.d8_layers import DropPathD8

    rag = R.Ragged([(400, 3), (600, 2)])
    dps = [DropPathD8(0.25), DropPathD8(0.0), DropPathD8(0.5)]
    for d in dps:
        d.train()
    torch.manual_seed(1)
    rag.draw_masks(dps, "cpu")
    assert set(rag.masks) == {id(dps[0]), id(dps[2])}       # inactive modules draw nothing
    for d in (dps[0], dps[2]):
        m = rag.masks[id(d)]
        keep = 1 - d.drop_prob
        assert m.shape == (rag.rows,) and all(v == 0.0 or abs(v - 1 / keep) < 1e-6 for v in m.unique().tolist())
        per_sample = torch.cat([m[:1200].view(400, 3)[:, 0], m[1200:].view(600, 2)[:, 0]])
        assert torch.equal(rag.row_scale(per_sample), m)     # one value per sample, repeated over its rows
        assert abs(float((per_sample > 0).float().mean()) - keep) < 0.06
