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
