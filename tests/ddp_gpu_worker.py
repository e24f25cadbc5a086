"""Child process of tests/test_train_gpu.py::test_two_rank_ddp_on_one_gpu (not a test module).

    python tests/ddp_gpu_worker.py <world> <rank> <port> <out.pt> [bf16_buckets]

world 2: both ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one device); each runs
Trainer(distributed=True) — DistributedDataParallel with gradients as bucket views, the fused HIP LAMB+EMA step reading
those views, bf16 weight copies handed to the caches — on its half of a fixed batch.  world 1: the same model on the
whole batch without a process group.  Rank 0 writes losses and a parameter checksum vector."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    world, rank, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    bf16_buckets = len(sys.argv) > 5 and sys.argv[5] == "bf16"
    from octic_vits_amd.d8_layers import Layer_scale_init_BlockD8
    from octic_vits_amd.model import OcticVisionTransformer
    from octic_vits_amd.train import Trainer, synthetic_batch
    from octic_vits_amd.vit import Layer_scale_init_Block
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.manual_seed(7)
    net = OcticVisionTransformer(img_size=56, patch_size=14, num_classes=100, embed_dim=128, depth=4, num_heads=4,
                                 qkv_bias=True, drop_path_rate=0.0, octic_block_layers=Layer_scale_init_BlockD8,
                                 standard_block_layers=Layer_scale_init_Block).cuda()
    x, y = synthetic_batch(8, 100, "cuda", 11, img_size=56)
    per = 8 // world
    x, y = x[rank * per:(rank + 1) * per], y[rank * per:(rank + 1) * per]
    # f32 arithmetic so that "mean over two half batches" equals "mean over the whole batch" to rounding
    tr = Trainer(net, distributed=world > 1, local_rank=0, autocast=False, bf16_buckets=bf16_buckets)
    losses = []
    for _ in range(3):
        loss = tr.step(x, y).detach()
        if world > 1:
            dist.all_reduce(loss)
            loss = loss / world
        losses.append(float(loss))
    flat = torch.cat([p.detach().flatten()[:: max(1, p.numel() // 64)][:64] for p in net.parameters()]).cpu()
    if world > 1:
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], g) for g in gathered), "replicas diverged"
    if rank == 0:
        torch.save({"losses": losses, "params": flat, "skipped": tr.optimizer.skipped_steps}, out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
