"""world_size-2 `gloo` checks (CPU) of the data-parallel plumbing used by bench.py / train.py for N > 1:
process-group init from the torchrun environment, DDP gradient averaging, identical replicas after LAMB steps,
equivalence with single-process training on the concatenated batch, and the max-over-ranks timing reduction."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_model():
    from octic_vits_amd.vit import Layer_scale_init_Block
    torch.manual_seed(7)
    return torch.nn.Sequential(Layer_scale_init_Block(dim=32, num_heads=4, qkv_bias=True, init_values=0.1),
                               torch.nn.Linear(32, 5))


def _batch(rank, world):
    g = torch.Generator().manual_seed(100)
    x = torch.randn(8, 6, 32, generator=g)
    y = torch.randn(8, 6, 5, generator=g)
    per = 8 // world
    return x[rank * per:(rank + 1) * per], y[rank * per:(rank + 1) * per]


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from octic_vits_amd.train import Lamb, init_distributed
    w, r, lr_ = init_distributed()
    assert (w, r) == (world, rank) and dist.get_backend() == "gloo"
    model = torch.nn.parallel.DistributedDataParallel(_make_model())
    opt = Lamb([{"params": list(model.parameters()), "weight_decay": 0.02}], lr=3e-3, weight_decay=0.02)
    x, y = _batch(rank, world)
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        ((model(x) - y) ** 2).mean().backward()
        opt.step()
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    # replicas identical
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert all(torch.equal(gathered[0], g) for g in gathered)
    # bench.py's timing reduction: max over ranks
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == float(world)
    if rank == 0:
        torch.save(flat, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ddp_gloo_two_ranks_matches_single_process(tmp_path):
    out = str(tmp_path / "ddp.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    ddp = torch.load(out)
    from octic_vits_amd.train import Lamb
    model = _make_model()
    opt = Lamb([{"params": list(model.parameters()), "weight_decay": 0.02}], lr=3e-3, weight_decay=0.02)
    x, y = _batch(0, 1)
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        ((model(x) - y) ** 2).mean().backward()
        opt.step()
    single = torch.cat([p.detach().flatten() for p in model.parameters()])
    assert torch.allclose(ddp, single, rtol=1e-4, atol=1e-6), (ddp - single).abs().max()


def test_synthetic_batches_differ_per_rank_and_are_reproducible():
    from octic_vits_amd.train import synthetic_batch
    a, ta = synthetic_batch(2, 10, "cpu", 4242 + 0, img_size=16)
    b, tb = synthetic_batch(2, 10, "cpu", 4242 + 1, img_size=16)
    a2, _ = synthetic_batch(2, 10, "cpu", 4242 + 0, img_size=16)
    assert not torch.equal(a, b) and torch.equal(a, a2)
    assert ta.shape == (2, 10) and set(ta.unique().tolist()) <= {0.0, 1.0} and (ta.sum(1) >= 1).all()


class _Net(torch.nn.Module):
    """standard block + head with the model surface Trainer needs (no_weight_decay)."""

    def __init__(self):
        super().__init__()
        self.body = _make_model()

    def no_weight_decay(self):
        return set()

    def forward(self, x):
        return self.body(x).mean(1)


def _accum_batch():
    g = torch.Generator().manual_seed(321)
    x = torch.randn(8, 6, 32, generator=g)
    y = (torch.rand(8, 5, generator=g) > 0.6).float()
    return x, y


def _accum_worker(rank, world, port, out, flat_numel=None, own=False):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from octic_vits_amd import train as TR
    from octic_vits_amd.train import Trainer, init_distributed
    init_distributed()
    x, y = _accum_batch()
    per = 8 // world
    net = _Net()
    if flat_numel is not None:
        # small tensors (biases, norm / layer-scale vectors) outside DDP, reduced as one flat buffer (train.DDP_FLAT_SMALL_NUMEL);
        # rank 1 starts from other values: the wrapper must bring rank 0's over for the tensors DDP does not manage too
        TR.DDP_FLAT_SMALL_NUMEL = flat_numel
        if rank == 1:
            with torch.no_grad():
                for p in net.parameters():
                    p.add_(0.37)
    tr = Trainer(net, distributed=True, fused_optimizer=False, tuned_gemms=False, autocast=False, accum_steps=2,
                 ema_decay=None, device_type="cpu", bucket_cap_mb=0.004 if own else 1, own_reducer=own)
    if own:
        assert tr._reducer is not None and tr.model is tr.raw_model and len(tr._reducer.buckets) >= 2
    elif flat_numel is not None:
        assert 0 < len(tr._small) < sum(1 for p in net.parameters() if p.requires_grad)
    for _ in range(3):
        tr.step(x[rank * per:(rank + 1) * per], y[rank * per:(rank + 1) * per])
    flat = torch.cat([p.detach().flatten() for p in tr.raw_model.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert all(torch.equal(gathered[0], g) for g in gathered)
    if rank == 0:
        torch.save(flat, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_small_gradients_outside_ddp_as_one_flat_all_reduce(tmp_path):
    """train.DDP_FLAT_SMALL_NUMEL: tensors of at most that many elements are not handed to DistributedDataParallel (which
    copies every gradient into its bucket with one launch per tensor) but averaged over the ranks as ONE flat buffer after the
    backward pass - 2 ranks x 2 accumulated micro-batches with a threshold that splits this model's tensors, rank 1 starting
    from different values: equal to one process on all 8 samples."""
    out = str(tmp_path / "flat.pt")
    mp.spawn(_accum_worker, args=(2, _free_port(), out, 64), nprocs=2, join=True)
    ddp = torch.load(out)
    from octic_vits_amd.train import Trainer
    x, y = _accum_batch()
    tr = Trainer(_Net(), distributed=False, fused_optimizer=False, tuned_gemms=False, autocast=False, accum_steps=1,
                 ema_decay=None, device_type="cpu")
    for _ in range(3):
        tr.step(x, y)
    single = torch.cat([p.detach().flatten() for p in tr.raw_model.parameters()])
    assert torch.allclose(ddp, single, rtol=1e-4, atol=1e-5), float((ddp - single).abs().max())


@pytest.mark.timeout(300)
def test_trainer_accumulation_under_ddp_matches_single_process(tmp_path):
    """configs[2]'s shape in miniature: 2 ranks x 2 accumulated micro-batches x 2 samples == one process on all 8
    samples.  The first micro-batch runs under DDP.no_sync (no all-reduce), the second one all-reduces the SUM of both
    micro-batch gradients; Trainer divides each micro-batch loss by the number of micro-batches."""
    out = str(tmp_path / "accum.pt")
    mp.spawn(_accum_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    ddp = torch.load(out)
    from octic_vits_amd.train import Trainer
    x, y = _accum_batch()
    for accum in (1, 4):
        tr = Trainer(_Net(), distributed=False, fused_optimizer=False, tuned_gemms=False, autocast=False,
                     accum_steps=accum, ema_decay=None, device_type="cpu")
        for _ in range(3):
            tr.step(x, y)
        single = torch.cat([p.detach().flatten() for p in tr.raw_model.parameters()])
        assert torch.allclose(ddp, single, rtol=1e-4, atol=1e-5), (accum, float((ddp - single).abs().max()))


def _reducer_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from octic_vits_amd import train as TR
    from octic_vits_amd.train import GradReducer, Trainer, init_distributed
    init_distributed()
    x, y = _accum_batch()
    per = 8 // world
    net = _Net()
    TR.DDP_FLAT_SMALL_NUMEL = 64            # splits this model's tensors into bucket members and "misc" members
    if rank == 1:
        with torch.no_grad():
            for p in net.parameters():
                p.add_(0.37)
    tr = Trainer(net, distributed=True, fused_optimizer=False, tuned_gemms=False, autocast=False, ema_decay=None,
                 device_type="cpu", bucket_cap_mb=0.004)           # ~1000 floats per bucket: several buckets
    assert isinstance(tr._reducer, GradReducer) and tr.model is tr.raw_model      # no DistributedDataParallel wrapper
    assert len(tr._reducer.buckets) >= 2 and tr._reducer.small
    for _ in range(3):
        tr.step(x[rank * per:(rank + 1) * per], y[rank * per:(rank + 1) * per])
        # every gradient the optimizer read is a view of reduced bucket / misc memory
        flats = [b[0] for b in tr._reducer.buckets] + [tr._reducer._misc]
        for p in net.parameters():
            assert any(f.data_ptr() <= p.grad.data_ptr() < f.data_ptr() + 4 * f.numel() for f in flats)
    flat = torch.cat([p.detach().flatten() for p in tr.raw_model.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert all(torch.equal(gathered[0], g) for g in gathered)
    if rank == 0:
        torch.save(flat, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_own_reducer_two_ranks_matches_single_process(tmp_path):
    """train.GradReducer (the reducer behind Trainer(distributed=True) for one-micro-batch steps, the one a captured data-parallel
    step records its collectives through): 2 ranks x 4 samples, rank 1 starting from other values, buckets of ~1000 floats +
    the misc buffer == one process on all 8 samples; replicas identical."""
    out = str(tmp_path / "own.pt")
    mp.spawn(_reducer_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    from octic_vits_amd.train import Trainer
    x, y = _accum_batch()
    tr = Trainer(_Net(), distributed=False, fused_optimizer=False, tuned_gemms=False, autocast=False, ema_decay=None,
                 device_type="cpu")
    for _ in range(3):
        tr.step(x, y)
    single = torch.cat([p.detach().flatten() for p in tr.raw_model.parameters()])
    assert torch.allclose(got, single, rtol=1e-4, atol=1e-5), float((got - single).abs().max())


def test_own_reducer_destination_protocol():
    """The ops.GRAD_DEST protocol of train.GradReducer on one rank: a gradient written INTO its bucket view (and announced
    with ops.grad_written) is adopted without a copy and its bucket is reduced as soon as its last member is announced; an
    unannounced one is gathered at the end; results equal the plain gradients."""
    from octic_vits_amd import ops
    from octic_vits_amd.train import GradReducer
    made = not dist.is_initialized()
    if made:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        torch.manual_seed(3)
        ps = [torch.nn.Parameter(torch.randn(40, 30)), torch.nn.Parameter(torch.randn(7)),
              torch.nn.Parameter(torch.randn(50, 20)), torch.nn.Parameter(torch.randn(30, 30))]
        r = GradReducer(ps, bucket_mb=1500 * 4 / (1 << 20), small_numel=10)
        assert [len(b[1]) for b in r.buckets] == [1, 1, 1] and r.small == [ps[1]]
        want = [torch.randn_like(p) for p in ps]

        class InPlace(torch.autograd.Function):          # a weight-gradient kernel that honours its destination
            @staticmethod
            def forward(ctx, w, g):
                ctx.w, ctx.g = w, g
                return w.sum() * 0

            @staticmethod
            def backward(ctx, _):
                d = ops.grad_dest(ctx.w, ctx.w.shape)
                assert d is not None
                d.copy_(ctx.g)
                ops.grad_written(ctx.w)
                return d, None

        r.begin()
        assert ops.GRAD_DEST is r
        loss = InPlace.apply(ps[3], want[3]) + (ps[0] * want[0]).sum() + (ps[1] * want[1]).sum() + (ps[2] * want[2]).sum()
        loss.backward()
        assert r._fired == {0}                           # ps[3] is bucket 0 (reverse registration order): reduced mid-pass
        r.finish()
        assert ops.GRAD_DEST is None and r.early == 1
        for p, w in zip(ps, want):
            assert torch.equal(p.grad, w)
        assert ps[3].grad.data_ptr() == r.buckets[0][0].data_ptr()
        assert ps[1].grad.data_ptr() == r._misc.data_ptr()
    finally:
        if made:
            dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_own_reducer_with_accumulation_matches_single_process(tmp_path):
    """configs[2]'s shape in miniature through train.GradReducer: 2 ranks x 2 accumulated micro-batches x 2 samples, rank 1
    starting from other values: the first micro-batch lands in the buckets, the second adds to them in place, ONE round of
    collectives follows - equal to one process on all 8 samples."""
    out = str(tmp_path / "own_accum.pt")
    mp.spawn(_accum_worker, args=(2, _free_port(), out, 64, True), nprocs=2, join=True)
    got = torch.load(out)
    from octic_vits_amd.train import Trainer
    x, y = _accum_batch()
    tr = Trainer(_Net(), distributed=False, fused_optimizer=False, tuned_gemms=False, autocast=False, accum_steps=1,
                 ema_decay=None, device_type="cpu")
    for _ in range(3):
        tr.step(x, y)
    single = torch.cat([p.detach().flatten() for p in tr.raw_model.parameters()])
    assert torch.allclose(got, single, rtol=1e-4, atol=1e-5), float((got - single).abs().max())


def test_own_reducer_edge_cases():
    """train.GradReducer on one rank: (a) a bucket member that takes no part in a pass contributes zeros and keeps .grad = None
    (optimizers skip it); (b) backward() raising leaves ops.GRAD_DEST as it was and the reducer usable for the next pass; (c) with
    hold=True (gradient accumulation) nothing is reduced before reduce_all, announcements included; (d) inside a capture-marked
    pass a collective is never issued from a non-capturing context (CPU: no stream is ever capturing), and reduce_all then
    refuses instead of leaving gradients unreduced."""
    from octic_vits_amd import ops
    from octic_vits_amd.train import GradReducer
    made = not dist.is_initialized()
    if made:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        torch.manual_seed(1)
        ps = [torch.nn.Parameter(torch.randn(30, 30)), torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(40, 20))]
        r = GradReducer(ps, bucket_mb=1000 * 4 / (1 << 20), small_numel=10)
        assert len(r.buckets) == 2
        # (a)
        r.begin()
        (ps[0].sum() * 2 + ps[1].sum() * 3).backward()
        r.finish()
        assert ps[2].grad is None and torch.equal(ps[0].grad, torch.full_like(ps[0], 2.0))
        assert float(r.buckets[0][0].abs().max()) == 0.0                    # its slot went through the collective as zeros
        # (b)
        for p in ps:
            p.grad = None

        class Boom(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x):
                return x.sum()

            @staticmethod
            def backward(ctx, g):
                raise ValueError("boom")
        r.begin()
        with pytest.raises(ValueError):
            try:
                Boom.apply(ps[0]).backward()
            except BaseException:
                r.abort()
                raise
        assert ops.GRAD_DEST is None and not r.active
        for p in ps:
            p.grad = None
        r.begin()
        sum(p.sum() for p in ps).backward()
        r.finish()
        assert all(torch.equal(p.grad, torch.ones_like(p)) for p in ps)
        # (c)
        for p in ps:
            p.grad = None
        r.begin(hold=True)
        d = ops.grad_dest(ps[2], ps[2].shape)
        assert d is not None
        ops.grad_written(ps[2])
        assert not r._fired
        sum(p.sum() for p in ps).backward()
        r.settle()
        assert not r._fired and ps[0].grad.data_ptr() == r.buckets[1][0].data_ptr()
        sum(p.sum() for p in ps).backward()                  # second micro-batch: added in place
        r.reduce_all()
        assert all(torch.equal(p.grad, torch.full_like(p, 2.0)) for p in ps)
        # (d)
        for p in ps:
            p.grad = None
        r.capturing = True
        try:
            r.begin()
            sum(p.sum() for p in ps).backward()
            r.settle()
            with pytest.raises(RuntimeError, match="not capturing"):
                r.reduce_all()
            assert not r._fired
        finally:
            r.capturing = False
            r.abort()
    finally:
        if made:
            dist.destroy_process_group()


def test_bench_spawns_its_own_ranks_when_launched_bare(monkeypatch, capsys):
    """`python bench.py --gpus N` with no WORLD_SIZE must start N ranks itself (fresh children, before any GPU call) and
    pass the exit code through; here the child launcher is intercepted."""
    import importlib
    import subprocess
    import sys
    bench = importlib.import_module("bench")
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1", "--accum", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-8:] == ["--gpus", "4", "--steps", "3", "--warmup", "1", "--accum", "4"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
