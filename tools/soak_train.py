"""Developer soak: 120 graph-replayed train steps of the flagship model on four rotating synthetic batches (loss must fall,
no step may be refused by the optimizer)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octic_vits_amd.deit_models import create_model
from octic_vits_amd.train import Trainer, synthetic_batch
COMPACT = "--compact" in sys.argv          # eager steps with stochastic depth as batch compaction (d8_layers.COMPACT_DROP_PATH)
torch.manual_seed(1337)
m = create_model("hybrid_deit_huge_patch14", num_classes=1000, drop_path_rate=0.5, img_size=224).cuda()
tr = Trainer(m, check_every=10)
bs = [synthetic_batch(64, 1000, "cuda", 100 + i) for i in range(4)]
if COMPACT:
    from octic_vits_amd import d8_layers as L
    L.COMPACT_DROP_PATH = True
    run = tr.step
else:
    run = tr.capture(*bs[0], warmup=2).replay
out = []
for i in range(int(os.environ.get("SOAK_STEPS", "120"))):
    l = run(*bs[i % 4])
    if i % 10 == 9:
        out.append(round(float(l), 4))
print("losses every 10 replays:", out, "skipped steps:", tr.optimizer.skipped_steps, "grad norm", float(tr.optimizer.last_grad_norm))
