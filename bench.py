"""Headline benchmark: images/sec of a Hybrid Octic ViT-H/14 224² bf16 TRAINING step (fwd + bwd + LAMB + EMA)
on synthetic batches, 1/2/4/8 MI355X (BASELINE.json `metric`, configs[1]: batch 64 per GPU).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `value` is whole-job images/s (weak scaling: 64 images per GPU).
`roofline` is for the dominant kernel of the WHOLE step — the engine's launches and the BLAS-library GEMMs of the
standard half alike — timed live with HIP events on the stream they are launched on during the timed steps
(`roofline_hbm_kernel`: the dominant HBM-bound one when the dominant one is MFMA-bound); `cpu_baseline` is the CPU oracle timed on this box's
host cores on a bounded sample (rank 0, N=1 only).  See DESIGN.md §measurement.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_IMG_STEP = 609.7e9          # 3 x 203.2 GFLOP (matmul FLOPs, fwd+bwd)
HBM_PEAK = 8.0e12                    # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_BF16 = 2.5e15              # dense bf16


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def usable_cores():
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(batch=8, steps=3):
    """The oracle (CPU restatement of the reference, pinned by tests/golden) on the host cores: same model,
    fp32, fwd + bwd + LAMB step, bounded sample: SURVEY 8d's protocol (batch 8, 3 timed iterations after 1 warm-up; ~25 s)."""
    from oracle import octic_ref as R
    from octic_vits_amd.train import Lamb, param_groups_weight_decay, synthetic_batch
    torch.manual_seed(0)
    # torch's CPU ops scale NEGATIVELY on the 256-thread GPU host for this op mix (measured: 16 threads
    # 0.84 img/s, 32 -> 0.60, 64 -> 0.33, all 256 did not finish in 40 min), so the baseline uses 16.
    # `cores` reports the threads actually used.
    cores = min(usable_cores(), int(os.environ.get("OCTIC_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    log(f"cpu_baseline: {cores} usable cores (os.cpu_count()={os.cpu_count()}), batch {batch}")
    model = R.create_model("hybrid_deit_huge_patch14", num_classes=1000, drop_path_rate=0.5, img_size=224).train()
    opt = Lamb(param_groups_weight_decay(model, 0.02, model.no_weight_decay()))
    x, y = synthetic_batch(batch, 1000, "cpu", 123)
    crit = torch.nn.BCEWithLogitsLoss()

    def one():
        opt.zero_grad(set_to_none=True)
        crit(model(x), y).backward()
        opt.step()

    t = time.perf_counter()
    one()  # warm-up
    log(f"cpu_baseline: warm-up step took {time.perf_counter() - t:.1f} s")
    t = time.perf_counter()
    for _ in range(steps):
        one()
    dt = (time.perf_counter() - t) / steps
    return {"value": batch / dt, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"Hybrid Octic ViT-H/14 224² fp32 train step (fwd+bwd+LAMB), batch {batch}, "
                      f"{steps} timed step(s) after 1 warm-up, CPU oracle (oracle/octic_ref.py)"}


def step_variants(trainer, model, samples, targets, args, n=8):
    """How the SAME step runs when it cannot be one hipGraph (N > 1 GPUs, gradient accumulation): launched eagerly, and as
    segment graphs (train.SegmentedModel: 2 x 8 graph replays + eager loss / hooks / optimizer).  Measured after the timed
    region on the same weights, so that the 1-GPU line says what the N > 1 lines should be compared with; `*_host_issue` is the
    host time to enqueue a step (no sync inside the loop): a step is host-bound when it approaches the step time."""
    from octic_vits_amd.train import Trainer

    def timed(tr):
        for _ in range(2):
            tr.step(samples, targets)
        torch.cuda.synchronize()
        w0 = tr._watch.wait_s
        t0 = time.perf_counter()
        for _ in range(n):
            tr.step(samples, targets)
        issued = time.perf_counter() - t0 - (tr._watch.wait_s - w0)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, issued / n * 1e3

    out = {}
    ms, host = timed(trainer)
    out["eager_ms_per_step"], out["eager_host_issue_ms_per_step"] = round(ms, 3), round(host, 2)
    # Stochastic depth as batch compaction (d8_layers.COMPACT_DROP_PATH, opt-in): every branch on the samples its per-sample
    # Bernoulli(1 - drop_path) mask keeps, instead of on every sample with half the results multiplied by zero - the same loss
    # and gradients (tests/test_train_gpu.py::test_compacted_stochastic_depth_equals_the_masked_full_batch), eager launches
    # because the shapes follow the draws.  Reported beside `value`, never as it.
    from octic_vits_amd import d8_layers as _L
    if True:
        _L.COMPACT_DROP_PATH = True
        try:
            for _ in range(40):                       # the per-shape plans / workspaces of ~30 distinct kept counts
                trainer.step(samples, targets)
            ms, host = timed(trainer)
            out["compact_drop_path_ms_per_step"] = round(ms, 3)
            out["compact_drop_path_host_issue_ms_per_step"] = round(host, 2)
            out["compact_drop_path_images_per_s"] = round(samples.shape[0] / ms * 1e3, 2)
            out["compact_drop_path_note"] = ("opt-in, NOT `value`: every branch on the samples its per-sample Bernoulli(0.5) "
                                             "stochastic-depth mask keeps (d8_layers.COMPACT_DROP_PATH; same loss and gradients as "
                                             "the masked full batch: tests/test_train_gpu.py::test_compacted_stochastic_depth_"
                                             "equals_the_masked_full_batch); eager launches, DESIGN.md 3.7")
        except Exception as e:                        # a side figure must never cost the line its `value`
            out["compact_drop_path_error"] = f"{type(e).__name__}: {e}"[:300]
        finally:
            _L.COMPACT_DROP_PATH = False
    # What a concurrent gradient all-reduce would do to this step, as far as ONE GPU can tell (train.DdpTrafficProxy): the eager
    # step under DistributedDataParallel on a one-rank group, its bucket hooks (a) reporting the buckets as reduced at once and
    # (b) first moving every bucket's bytes device-to-device on a side stream - 1.42 GB per step in 48 MB pieces (0.71 GB with
    # --bf16-buckets), twice each (a ring all-reduce over 8 GPUs reads and writes ~1.75 x the payload per GPU).
    try:
        import torch.distributed as dist
        from octic_vits_amd.train import DdpTrafficProxy
        made = not dist.is_initialized()
        if made:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29541")
            dist.init_process_group("gloo", rank=0, world_size=1)
        try:
            bkw = {} if args.bucket_mb is None else {"bucket_cap_mb": args.bucket_mb}
            for key, copies in (("ddp_hooks_ms_per_step", 0), ("ddp_proxy_ms_per_step", 2)):
                proxy = DdpTrafficProxy(samples.device, copies=copies, halve=args.bf16_buckets)
                tp = Trainer(model, distributed=True, local_rank=samples.device.index or 0, ddp_proxy=proxy, own_reducer=False, **bkw)
                ms, host = timed(tp)
                out[key] = round(ms, 3)
                if copies:
                    out["ddp_proxy_gb_per_step"] = round(proxy.bytes / (n + 2) / 1e9, 3)
                del tp, proxy
            out["ddp_proxy_note"] = ("one GPU, one-rank group: DDP bucket hooks + a side stream copying each gradient bucket "
                                     "device-to-device twice (ring all-reduce traffic per GPU); prices HBM / fabric contention "
                                     "and the exposed last bucket only - no xGMI, no RCCL kernels, no rank skew")
        finally:
            if made:
                dist.destroy_process_group()
    except Exception as e:
        out["ddp_proxy_error"] = f"{type(e).__name__}: {e}"[:300]
    del trainer
    try:
        seg = Trainer(model, segment_graphs=8)       # (re-links the blocks inside the slices: keep this last)
        seg.capture_segments(samples)
        ms, host = timed(seg)
        out["segment_graph_ms_per_step"], out["segment_graph_host_issue_ms_per_step"] = round(ms, 3), round(host, 2)
    except Exception as e:
        out["segment_graph_error"] = f"{type(e).__name__}: {e}"[:300]
    return out


def _baseline_metric():
    """BASELINE.json's metric string, verbatim (the driver matches on it)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        return "images/sec Hybrid Octic ViT-H/14 224² bf16 train step @1/2/4/8 MI355X"


METRIC = _baseline_metric()


def ddp_side_figure(args):
    """The N > 1 step as `bench.py --gpus N` runs it, on ONE GPU with a one-rank RCCL group, in a CHILD process (an RCCL problem
    there cannot cost this line its `value`): `bench.py --force-ddp` - eager steps of train.GradReducer first, then the step
    captured as one hipGraph with the collectives inside."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    cmd = [sys.executable, os.path.join(here, "bench.py"), "--force-ddp", "--steps", "8", "--warmup", "3", "--no-cpu-baseline",
           "--no-forward-only", "--no-ssl-side", "--no-step-variants", "--no-kernel-timing", "--batch", str(args.batch),
           "--model", args.model]
    if args.bucket_mb is not None:
        cmd += ["--bucket-mb", str(args.bucket_mb)]
    try:
        env = dict(os.environ, MASTER_PORT=str(29600 + os.getpid() % 300))
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env.pop(k, None)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=420, cwd=here, env=env)
        lines = [ln for ln in (r.stdout or "").strip().splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"ddp_graph_error": ((r.stderr or "no output").strip().splitlines() or ["?"])[-1][:300]}
        d = json.loads(lines[-1])
        out = {"ddp_launch": d["config"]["launch"], "ddp_gradient_reduction": d["config"].get("gradient_reduction"),
               "ddp_note": "side figures, never `value`: child process `bench.py --force-ddp` (one-rank RCCL group on this GPU: every "
                           "collective is RCCL's one-rank AVG kernel over the bucket)"}
        if "hipGraph" in d["config"]["launch"]:
            out["ddp_graph_ms_per_step"], out["ddp_graph_host_issue_ms_per_step"] = d["ms_per_step"], d["host_issue_ms_per_step"]
            out["ddp_eager_ms_per_step"] = d.get("eager_reducer_ms_per_step")
            out["ddp_eager_host_issue_ms_per_step"] = d.get("eager_reducer_host_issue_ms_per_step")
        else:
            out["ddp_eager_ms_per_step"], out["ddp_eager_host_issue_ms_per_step"] = d["ms_per_step"], d["host_issue_ms_per_step"]
        return out
    except Exception as e:
        return {"ddp_graph_error": f"{type(e).__name__}: {e}"[:300]}


def ssl_side_figure():
    """BASELINE configs[4] on one GPU as a side figure (never `value`): tools/bench_ssl.py - the DINOv2 student / teacher step of
    the hybrid ViT-H/16 (the largest DINOv2 factory the reference has; there is no ViT-g/14 one), 32 images per GPU = 64 global
    224^2 + 256 local 96^2 crops, bf16, 65 536 prototypes - in a CHILD process started after the headline measurement: its
    failure or timeout costs the line nothing."""
    import re
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    try:
        r = subprocess.run([sys.executable, os.path.join(here, "tools", "bench_ssl.py"), "32", "6"], capture_output=True,
                           text=True, timeout=420, cwd=here)
        m = re.search(r"([0-9.]+) ms/step, ([0-9.]+) images/s, peak memory ([0-9.]+) GiB", r.stdout or "")
        if r.returncode != 0 or not m:
            return {"ssl_side_error": ((r.stderr or r.stdout or "no output").strip().splitlines() or ["?"])[-1][:300]}
        return {"ssl_step_ms": float(m.group(1)), "ssl_images_per_s": float(m.group(2)), "ssl_peak_gib": float(m.group(3)),
                "ssl_note": "side figure, never `value`: DINOv2 student/teacher step (tools/bench_ssl.py: hybrid ViT-H/16, 32 "
                            "images/GPU = 64 global + 256 local crops, bf16, eager, fused AdamW + EMA), child process"}
    except Exception as e:
        return {"ssl_side_error": f"{type(e).__name__}: {e}"[:300]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (BASELINE configs[1]: 64)")
    ap.add_argument("--model", default="hybrid_deit_huge_patch14")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-forward-only", action="store_true")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every step eagerly (default: the step is captured once with Trainer.capture and replayed as "
                         "one hipGraph - at N > 1 with the bucket all-reduces of train.GradReducer recorded inside it; the "
                         "kernel-timing steps stay eager.  Also: OCTIC_DDP_GRAPH=0 for the data-parallel step only)")
    ap.add_argument("--torch-ddp", action="store_true",
                    help="developer A/B: N > 1 through torch's DistributedDataParallel (eager launches) instead of "
                         "train.GradReducer")
    ap.add_argument("--no-wgrad-slabs", action="store_true",
                    help="developer A/B: library weight gradients as one GEMM instead of a batched GEMM over row slabs")
    ap.add_argument("--lib-wgrad", action="store_true",
                    help="developer A/B: every standard-half weight gradient on the BLAS library (default: csrc/dense_wgrad.hip "
                         "wherever ops.dense_wgrad_ok takes the shape)")
    ap.add_argument("--no-fused-attn-bwd", action="store_true",
                    help="developer A/B: attention backward as the dq + dkv kernel pair instead of the single-pass kernel")
    ap.add_argument("--compact-drop-path", action="store_true",
                    help="opt-in: stochastic depth as batch compaction (d8_layers.COMPACT_DROP_PATH): every branch runs on the "
                         "samples its per-sample Bernoulli mask keeps - same loss and gradients, about half the branch work at "
                         "drop_path 0.5.  Launch shapes vary per step: implies --no-graph")
    ap.add_argument("--no-batched-finishes", action="store_true",
                    help="developer A/B: the parameter-gradient slab reductions of the backward pass as immediate launches "
                         "instead of one batched launch at its end (ops._DeferredFinishes)")
    ap.add_argument("--no-gelu-factor", action="store_true",
                    help="developer A/B: the standard MLP keeps h between fc1 and fc2's input gradient (GELU' evaluated in the "
                         "backward epilogue) instead of the bf16 gelu'(h) factor (functional.GELU_FACTOR)")
    ap.add_argument("--no-paired-wgrad", action="store_true",
                    help="developer A/B: the qkv and proj weight gradients of a standard block as two launches instead of one "
                         "(functional.WGRAD_PAIRED)")
    ap.add_argument("--no-packed-attn", action="store_true",
                    help="developer A/B: AttentionD8 through the pack / unpack kernels instead of the packed-row attention")
    ap.add_argument("--wgrad-f32-out", action="store_true",
                    help="developer A/B: library weight gradients straight to f32 (default: bf16 result + cast, as autocast does)")
    ap.add_argument("--no-octic-next-norm", action="store_true",
                    help="developer A/B: octic proj / fc2 and the following LayerNormD8 as separate autograd nodes (cast pass back)")
    ap.add_argument("--no-ln-tail", action="store_true",
                    help="developer A/B: LayerNorm backward and the residual-tail backward in front of it as two row passes")
    ap.add_argument("--no-next-norm", action="store_true",
                    help="developer A/B: residual add and the next LayerNorm of the standard half as two row passes")
    ap.add_argument("--resid-fused", action="store_true",
                    help="developer A/B: proj / fc2 with the layer-scale + residual tail inside the GEMM epilogue")
    ap.add_argument("--dense-hip", default=None,
                    help="developer A/B: comma list of standard-half GEMMs on csrc/dense_gemm.hip (default: functional.DENSE_HIP; 'none' = library)")
    ap.add_argument("--accum", type=int, default=1,
                    help="micro-batches of --batch images per optimizer step (gradient accumulation; BASELINE configs[2] "
                         "= --gpus 8 --batch 64 --accum 4: global batch 2048)")
    ap.add_argument("--segment-graphs", type=int, default=None,
                    help="forward / backward as 2 n hipGraph replays (train.SegmentedModel), loss + gradient hooks + optimizer "
                         "eager (default 0: off; the step is ONE graph at 1 GPU and eager launches at N > 1)")
    ap.add_argument("--force-ddp", action="store_true",
                    help="developer: wrap the model in DistributedDataParallel at world size 1 (one-rank RCCL group): the N > 1 "
                         "code path on one GPU")
    ap.add_argument("--no-ssl-side", action="store_true",
                    help="skip the DINOv2 student / teacher step side figure (BASELINE configs[4] on one GPU, a child process)")
    ap.add_argument("--no-step-variants", action="store_true",
                    help="skip the extra figures of the 1-GPU line (eager_ms_per_step, segment_graph_ms_per_step)")
    ap.add_argument("--bucket-mb", type=int, default=None, help="DDP gradient bucket size (default: train.DDP_BUCKET_MB)")
    ap.add_argument("--bf16-buckets", action="store_true", help="all-reduce the gradient buckets in bf16")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves as FRESH child processes, before
        # this process has touched the GPU (nothing above initialises HIP), relay rank 0's JSON line and the exit code.
        import socket
        import subprocess
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
        s_.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log("no WORLD_SIZE in the environment: launching " + " ".join(cmd[1:]))
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    # stdout carries ONE JSON line and nothing else: libraries that print to stdout (RCCL prints a version banner when its
    # first communicator is created) are sent to stderr for the rest of the run; the line goes out through the saved fd
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    from octic_vits_amd import ops
    from octic_vits_amd.deit_models import create_model
    from octic_vits_amd.train import Trainer, init_distributed, synthetic_batch
    import torch.distributed as dist

    if os.environ.get("OCTIC_NO_IMAGE"):             # developer A/B: classic row panels for every dense GEMM
        from octic_vits_amd import _lib as _L0
        _L0.route_override(_L0.ROUTE_DENSE_IMAGE, int(os.environ["OCTIC_NO_IMAGE"]) if os.environ["OCTIC_NO_IMAGE"] in ("2", "3") else 2)
    if os.environ.get("OCTIC_CLS2"):                 # developer A/B: class-token rows as two launches: 1 = never, 2 = wherever legal
        from octic_vits_amd import _lib as _L1
        _L1.route_override(_L1.ROUTE_DENSE_CLS2, int(os.environ["OCTIC_CLS2"]))
    if args.resid_fused:
        import octic_vits_amd.functional as _OF2
        _OF2.DENSE_RESID_FUSED = True
    if args.no_octic_next_norm:
        import octic_vits_amd.functional as _OF5
        _OF5.OCTIC_NEXT_NORM = False
    if args.no_ln_tail:
        import octic_vits_amd.functional as _OF4
        _OF4.LN_TAIL_FUSED = False
    if args.no_next_norm:
        import octic_vits_amd.functional as _OF3
        _OF3.NEXT_NORM_FUSED = False
    if args.dense_hip is not None:
        from octic_vits_amd import functional as _OF
        _OF.DENSE_HIP = set() if args.dense_hip == "none" else set(args.dense_hip.split(","))
    if args.no_wgrad_slabs:
        from octic_vits_amd import functional as _OF
        _OF.WGRAD_SLABS = False
    if args.lib_wgrad:
        from octic_vits_amd import functional as _OF
        _OF.WGRAD_HIP = False
    if args.no_gelu_factor:
        from octic_vits_amd import functional as _OF
        _OF.GELU_FACTOR = False
    if args.no_paired_wgrad:
        from octic_vits_amd import functional as _OF
        _OF.WGRAD_PAIRED = False
    if args.no_packed_attn:
        from octic_vits_amd import functional as _OF
        _OF.ATTN_PACKED = False
    if args.no_fused_attn_bwd:
        ops.ATTN_BWD_FUSED = False
    if args.no_batched_finishes:
        from octic_vits_amd import train as _T
        _T.BATCHED_FINISHES = False
    if args.compact_drop_path:
        from octic_vits_amd import d8_layers as _L
        _L.COMPACT_DROP_PATH = True
        args.no_graph = True
    if args.wgrad_f32_out:
        from octic_vits_amd import functional as _OF
        _OF.WGRAD_F32_OUT = True
    if args.force_ddp and "WORLD_SIZE" not in os.environ:
        os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=os.environ.get("MASTER_PORT", "29533"))
    world, rank, local_rank = init_distributed(force=args.force_ddp)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    torch.manual_seed(1337 + rank)   # experiments/train_deit.py:41 ; deit/main.py:222
    log(f"rank {rank}/{world}: building {args.model}")
    model = create_model(args.model, num_classes=1000, drop_path_rate=0.5, img_size=224).to(dev)
    log("model on device")
    tkw = {} if args.bucket_mb is None else {"bucket_cap_mb": args.bucket_mb}
    ddp = world > 1 or args.force_ddp
    own = ddp and not args.torch_ddp and not args.bf16_buckets and not (args.segment_graphs or 0)
    one_graph = (not ddp or (own and os.environ.get("OCTIC_DDP_GRAPH", "1") != "0")) and not args.no_graph
    # segment graphs are opt-in: on one MI355X the 2 x 8 replays run the step in 75.7 ms against 67.8 ms for eager launches
    # (gradient clones out of the static buffers, cut fusion links, per-slice mask pools) - they pay once the eager step
    # is host-bound (tools/host_profile.py: ~45 ms of Python / ctypes per step), which at 67 ms of kernels it is not
    nseg = args.segment_graphs or 0
    trainer = Trainer(model, distributed=ddp, local_rank=local_rank, accum_steps=args.accum,
                      bf16_buckets=args.bf16_buckets, segment_graphs=nseg, own_reducer=own if ddp else None, **tkw)
    samples, targets = synthetic_batch(args.batch * args.accum, 1000, dev, 4242 + rank)
    # the timed steps rotate over three resident batches (the graph replays copy each into their input buffers)
    batches = [(samples, targets)] + [synthetic_batch(args.batch * args.accum, 1000, dev, 5000 + 17 * j + rank) for j in (1, 2)]
    n_ranks_seen = dist.get_world_size() if (world > 1 and dist.is_initialized()) else 1

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if nseg:
        trainer.capture_segments(samples[:args.batch])
        log(f"forward / backward captured as {len(trainer.segmented._segments)} x 2 hipGraphs")
    for i in range(args.warmup):
        trainer.step(samples, targets)
        if i == 0:
            torch.cuda.synchronize()
            log("first warm-up step done")
    def timed_region(step_fn):
        """`steps` full iterations between barrier + synchronize, a new batch every step as a loader would deliver it (a graph's
        static input buffers are refilled by a device-to-device copy - 38.5 MB of images + targets - that is part of the step);
        no host read of the loss inside (train._LossWatch checks two steps late on a pinned copy).  -> (seconds: max over the
        ranks, host seconds spent ENQUEUEING - the loop's wall time minus the loss watch's waits -, last loss)."""
        sync()
        w0 = trainer._watch.wait_s
        t0 = time.perf_counter()
        loss_ = None
        for i in range(args.steps):
            bx, by = batches[i % len(batches)]
            loss_ = step_fn(bx, by).detach()       # (never hold an iteration's autograd graph: see train.Trainer.step)
        issued_ = time.perf_counter() - t0 - (trainer._watch.wait_s - w0)
        sync()
        el = time.perf_counter() - t0
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), issued_, loss_

    def core_line(elapsed_, issued_, loss_, launch):
        ms_ = elapsed_ / args.steps * 1e3
        ips_ = world * args.batch * args.accum * args.steps / elapsed_
        n_oct_ = model.octic_equi_break_layer
        n_std_ = len(model.blocks) - n_oct_
        return {
            "metric": METRIC, "value": round(ips_, 2),
            "unit": "images/s", "n_gpus": world, "n_ranks_seen": n_ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic (randn images, multi-hot targets; random-init weights)",
            "config": {"workload": f"{args.model} ({n_oct_} octic + {n_std_} standard blocks"
                                   f"{', invariant hand-off' if getattr(model, 'invariant', False) else ''}, drop_path 0.5) "
                                   "DeiT-III train step: "
                                   f"bf16-autocast fwd + bwd + LAMB + EMA, 224x224, batch {args.batch}/GPU "
                                   + (f"x {args.accum} accumulated micro-batches (BASELINE configs[2] shape), "
                                      if args.accum > 1 else "(BASELINE configs[1]), ")
                                   + f"data parallel over {world} GPU(s)",
                       "global_batch": world * args.batch * args.accum, "per_gpu_batch": args.batch,
                       "accum_steps": args.accum, "parallelism": f"dp{world}",
                       "library_gemm_table": bool(trainer.tuned_gemms),
                       "gradient_reduction": (None if not ddp else
                                              f"train.GradReducer: {len(trainer._reducer.buckets) + 1} flat f32 buckets, one RCCL "
                                              f"all-reduce (AVG) each, {trainer._reducer.early} issued during the backward pass"
                                              if trainer._reducer is not None else "torch DistributedDataParallel bucket hooks"),
                       "launch": launch},
            "loss": float(loss_.item()), "host_issue_ms_per_step": round(issued_ / args.steps * 1e3, 2),
            # (FLOP_PER_IMG_STEP is the ViT-H/14 count of SURVEY 8d: other --model choices are development figures)
            "step_mfma_frac": round(ips_ * FLOP_PER_IMG_STEP / (world * MFMA_PEAK_BF16), 4) if "huge_patch14" in args.model else None,
        }

    # Data parallel with the captured step (the default at N > 1): the collectives inside a hipGraph have never run on more than
    # one GPU of this pool (no multi-GPU node in six rounds), so the SAFE measurement comes first - `steps` eager steps of the same
    # reducer - and the captured one runs under a watchdog: if capture + replays do not finish within OCTIC_BENCH_WATCHDOG_S
    # (default 240 s), rank 0 prints the eager line and every rank leaves.  A line is produced either way.
    eager_dp = None
    watchdog = None
    if ddp and own and one_graph and not nseg and (world > 1 or args.force_ddp):
        eager_dp = timed_region(trainer.step)
        log(f"eager data-parallel steps: {eager_dp[0] / args.steps * 1e3:.2f} ms per step; now the captured step")
        import threading
        fallback = core_line(eager_dp[0], eager_dp[1], eager_dp[2],
                             "eager (the captured data-parallel step did not finish in time: watchdog)") if rank == 0 else None

        def _bail():
            log("WATCHDOG: the captured data-parallel step did not finish; reporting the eager steps measured before it")
            if rank == 0:
                print(json.dumps(fallback), file=real_stdout, flush=True)
            os._exit(0)
        watchdog = threading.Timer(float(os.environ.get("OCTIC_BENCH_WATCHDOG_S", "240")), _bail)
        watchdog.daemon = True
        watchdog.start()
    graphed = None
    if one_graph and not nseg:
        try:
            graphed = trainer.capture(samples, targets, warmup=1)
            for _ in range(2):
                graphed.replay()
            if os.environ.get("OCTIC_BENCH_FAKE_HANG") and watchdog is not None:     # developer: exercise the watchdog
                time.sleep(1e6)
            # torch.cuda.graph() empties the caching allocator before it captures: one more eager step here lets the
            # eager kernel-timing steps of the timed region reuse cached blocks instead of calling hipMalloc
            if not args.no_kernel_timing:
                trainer.step(samples, targets)
            torch.cuda.synchronize()
        except Exception as e:      # a failed capture can leave the stream unusable: measure eagerly in a fresh child
            if ddp:
                # (every rank runs the same capture: it fails on all of them or on none; the ranks cannot be re-started from
                # here, so the eager data-parallel step carries on in this process)
                log(f"graph capture failed ({type(e).__name__}: {e}); continuing with eager data-parallel steps")
                graphed = None
                torch.cuda.synchronize()
            else:
                log(f"graph capture failed ({type(e).__name__}: {e}); re-running eagerly in a child process")
                import subprocess
                real_stdout.flush()     # fd 1 is stderr in this process (see above): hand the child the real stdout for its line
                raise SystemExit(subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--no-graph"],
                                                stdout=real_stdout).returncode)
    log("warm-up done, timing")
    # The timed region is `steps` full iterations and nothing else: graph replays (eager steps where nothing was captured).  The
    # per-kernel HIP events (~1800 records per step, which stretch a step by ~20 %) are taken AFTER it, in the same process on
    # the same weights: see `step_breakdown.note`.
    if graphed is None and eager_dp is not None:
        elapsed, issued, loss = eager_dp            # (already measured; the capture failed)
    else:
        elapsed, issued, loss = timed_region(graphed.replay if graphed is not None else trainer.step)
    if watchdog is not None:
        watchdog.cancel()
    sampled = []
    if not args.no_kernel_timing:
        ops.KERNEL_TIMER.enable()
        sampled = [0, 1]
        for _ in sampled:
            trainer.step(samples, targets)
        sync()
        ops.KERNEL_TIMER.disable()
    log(f"timed region: {elapsed:.3f} s for {args.steps} steps")
    ms = elapsed / args.steps * 1e3
    ips = world * args.batch * args.accum * args.steps / elapsed

    # ---- forward-only throughput in the reference's own protocol (experiments/complexity.py:13-15,40-56,60-95): eval,
    # no_grad, batch 64, 10 warm-up + 100 timed forwards each followed by a synchronize, mean time -> images/s.  The
    # reference runs torch.amp.autocast (fp16) + torch.compile; here bf16 autocast (the engine's compute dtype) and the
    # HIP kernels.  Reported under "extra", outside the timed train region.
    fwd_only = None
    if rank == 0 and not args.no_forward_only:
        model.eval()
        times = []
        with torch.no_grad():
            img = torch.randn(64, 3, 224, 224, device=dev)
            for i in range(10 + 100):
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    t1 = time.time()
                    model(img)
                    torch.cuda.synchronize()
                    if i >= 10:
                        times.append(time.time() - t1)
        mean = sum(times) / len(times)
        fwd_only = {"images_per_s": round(64 / mean, 1), "ms_per_forward": round(mean * 1e3, 3),
                    "protocol": "experiments/complexity.py: eval, no_grad, batch 64, 224x224, 10 warm-up + 100 forwards, "
                                "synchronize after each, mean; bf16 autocast (reference: fp16 autocast + torch.compile)"}
        try:                                          # the same protocol with the forward as one hipGraph replay (serve.py)
            from octic_vits_amd.serve import GraphedForward
            gf = GraphedForward(model, img)
            gt = []
            for i in range(10 + 100):
                t1 = time.time()
                gf(img, copy_out=False)
                torch.cuda.synchronize()
                if i >= 10:
                    gt.append(time.time() - t1)
            gm = sum(gt) / len(gt)
            fwd_only["graph_images_per_s"] = round(64 / gm, 1)
            fwd_only["graph_ms_per_forward"] = round(gm * 1e3, 3)
            del gf
        except Exception as e:
            fwd_only["graph_error"] = f"{type(e).__name__}: {e}"[:200]
        model.train()
        log(f"forward-only (reference protocol): {fwd_only['images_per_s']} img/s")
    if world > 1:
        dist.barrier()

    if rank == 0:
        line = core_line(elapsed, issued, loss,
                         ("hipGraph replay (whole step" + (", RCCL all-reduces inside)" if ddp else ")")) if graphed is not None else
                         (f"{len(trainer.segmented._segments)} x 2 segment hipGraphs + eager loss / all-reduce hooks / optimizer"
                          if nseg else "eager"))
        if eager_dp is not None and graphed is not None:
            line["eager_reducer_ms_per_step"] = round(eager_dp[0] / args.steps * 1e3, 3)
            line["eager_reducer_host_issue_ms_per_step"] = round(eager_dp[1] / args.steps * 1e3, 2)
        kern = ops.KERNEL_TIMER.summary()
        if kern:
            traffic_db = {}
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                with open(tpath) as f:
                    traffic_db = json.load(f)
            sampled_steps = len(sampled)
            agg = ops.KERNEL_TIMER._agg()

            def family_of(name):
                if name.startswith("library_gemm"):
                    return "library GEMM (hipBLASLt, standard half)"
                if name.startswith(("dense_nt_kernel", "dense_tn_kernel")):
                    return "hand-written dense GEMM (standard half)"
                if name.startswith(("linear_d8", "wgrad_", "mlp_d8")):
                    return "octic irrep GEMM (LinearD8 fwd/dgrad/wgrad)"
                if name.startswith("attn_"):
                    return "attention"
                if name.startswith(("lamb_step", "adamw_step", "dense_prep_batch", "linear_prep_batch")):
                    return "optimizer (fused LAMB + EMA, bf16 / prepared weight copies)"
                return "row kernels (LayerNorm, GELU, residual tails, casts)"

            def merged(names, label):
                """One record for a set of timer entries (e.g. all library weight-gradient shapes = one rocprof family)."""
                rs = [agg[n] for n in names]
                tot = sum(r["total_us"] for r in rs)
                launches = sum(r["launches"] for r in rs)
                return {"name": label, "launches": launches, "total_us": tot, "avg_us": tot / launches,
                        "alg_bytes_per_launch": sum(r["bytes"] for r in rs) / launches,
                        "flops_per_launch": sum(r["flops"] for r in rs) / launches}

            def roof(k, bound=None):
                sec = k["avg_us"] * 1e-6
                tr = traffic_db.get(k["name"], {}).get("hbm_bytes_per_launch")
                common = {"kernel": k["name"], "launches": k["launches"], "avg_us": round(k["avg_us"], 2),
                          "share_of_step": round(k["total_us"] / (ms * 1e3 * sampled_steps), 4), "traffic": tr}
                if tr is not None:
                    common["traffic_source"] = "profiles/traffic.json (PMC FETCH_SIZE x2 + WRITE_SIZE, builder's rocprofv3 run)"
                if (bound or ops.KERNEL_TIMER.bound_of(k["name"])) == "mfma":
                    ach = k["flops_per_launch"] / sec
                    return {"bound": "mfma", "achieved": round(ach / 1e12, 1), "peak": MFMA_PEAK_BF16 / 1e12,
                            "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_BF16, 4),
                            "alg_flops_per_launch": int(k["flops_per_launch"]), **common}
                ach = k["alg_bytes_per_launch"] / sec
                return {"bound": "hbm", "achieved": round(ach / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK, 4), "alg_bytes_per_launch": int(k["alg_bytes_per_launch"]),
                        "mfma_tflops": round(k["flops_per_launch"] / sec / 1e12, 1), **common}

            # `roofline` = the kernel with the largest total time in the timed steps.  Library GEMMs are ranked the way a
            # profiler sees them - per operation kind (all weight-gradient shapes run the same hipBLASLt kernel family),
            # not split by shape - so a 10 % library family is not hidden behind a 6 % hand-written kernel.
            # one candidate per kernel SYMBOL: dense_nt_kernel<0, 5> (256 x 320 tile: proj / fc2 forward and the input
            # gradients of qkv / proj / fc1 at N = 1280, timer names ending in "@320") and dense_nt_kernel<0, 4> (plain
            # epilogue on the 256 x 256 tile: qkv forward)
            plain = ("dense_nt_kernel<plain>", "dense_nt_kernel<dgrad>", "dense_nt_kernel<proj>", "dense_nt_kernel<fc2>", "dense_nt_kernel<0>")
            nt5 = [n for n in agg if n.endswith("@320")]
            nt0 = [n for n in agg if n in plain]
            cands = {n: a for n, a in agg.items() if not n.startswith(("library_gemm", "dense_tn_kernel<")) and n not in nt0 and n not in nt5}
            tn = [n for n in agg if n.startswith("dense_tn_kernel<")]
            if tn:    # one kernel symbol (csrc/dense_wgrad.hip) over the four weight-gradient shapes
                cands["dense_tn_kernel<wgrad, all shapes>"] = merged(tn, "dense_tn_kernel<wgrad, all shapes>")
            if nt0:
                cands["dense_nt_kernel<0, 4>"] = merged(nt0, "dense_nt_kernel<0, 4>")
            if nt5:
                cands["dense_nt_kernel<0, 5>"] = merged(nt5, "dense_nt_kernel<0, 5>")
            for kind in ("fwd", "dgrad", "wgrad"):
                names = [n for n in agg if n.startswith(f"library_gemm<{kind} ")]
                if names:
                    cands[f"library_gemm<{kind}, all shapes>"] = merged(names, f"library_gemm<{kind}, all shapes>")
            top = max(cands.values(), key=lambda a: a["total_us"])
            line["roofline"] = roof(top)
            hand = [a for n, a in cands.items() if not n.startswith("library_gemm")]
            if hand:
                kh = max(hand, key=lambda a: a["total_us"])
                if kh["name"] != top["name"]:
                    line["roofline_top_handwritten"] = roof(kh)
                hb = [a for a in hand if ops.KERNEL_TIMER.bound_of(a["name"]) == "hbm"]
                if hb:
                    kb = max(hb, key=lambda a: a["total_us"])
                    if kb["name"] not in (top["name"], kh["name"]):
                        line["roofline_hbm_kernel"] = roof(kb)
            line["kernels"] = kern
            # families: where the step goes, each against its own roofline
            fams = {}
            for n, a in agg.items():
                f = fams.setdefault(family_of(n), {"ms_per_step": 0.0, "bytes": 0.0, "flops": 0.0, "us": 0.0,
                                                   "mfma": ops.KERNEL_TIMER.bound_of(n) == "mfma"})
                f["ms_per_step"] += a["total_us"] / sampled_steps / 1e3
                f["us"] += a["total_us"]
                f["bytes"] += a["bytes"]
                f["flops"] += a["flops"]
            timed_us = sum(v["total_us"] for v in kern.values()) / sampled_steps
            out_f = {}
            for name, f in sorted(fams.items(), key=lambda kv: -kv[1]["ms_per_step"]):
                sec = f["us"] * 1e-6
                rec = {"ms_per_step": round(f["ms_per_step"], 2), "share_of_step": round(f["ms_per_step"] / ms, 4)}
                if f["flops"]:
                    rec["TFLOPs"] = round(f["flops"] / sec / 1e12, 1)
                    rec["frac_mfma_peak"] = round(f["flops"] / sec / MFMA_PEAK_BF16, 4)
                rec["GBps"] = round(f["bytes"] / sec / 1e9, 1)
                rec["frac_hbm_peak"] = round(f["bytes"] / sec / HBM_PEAK, 4)
                out_f[name] = rec
            out_f["ATen glue (loss, head, copies, RNG; untimed)"] = {"ms_per_step": round(ms - timed_us / 1e3, 2),
                                                                     "share_of_step": round(1 - timed_us / 1e3 / ms, 4)}
            line["families"] = out_f
            line["kernel_time_source"] = ("HIP event pairs around every engine launch (kernels, the fused optimizer call, the "
                                          "weight-copy launches) on the launch stream, 2 eager steps after the timed region; "
                                          "rocprofv3 --kernel-trace of the same command: profiles/rocprof_r6_bench.txt")
            # how much of a step the per-kernel table explains: engine kernels + the BLAS-library GEMMs are timed
            # individually; the rest is ATen glue (casts, reductions, RNG, copies) and the fused optimizer
            line["step_breakdown"] = {"timed_kernels_ms": round(timed_us / 1e3, 2),
                                      "other_ms": round(ms - timed_us / 1e3, 2),
                                      "note": "per-launch HIP events are recorded in 2 eager steps run right AFTER the timed region "
                                              "(same process, same weights; ~1800 event pairs stretch such a step by ~20 % - gaps "
                                              "between launches, not kernel time; what an EMPTY event pair reads, "
                                              f"{ops.KERNEL_TIMER.overhead_us:.2f} us here, is subtracted from every bracket); every "
                                              "engine launch incl. the fused optimizer is timed, other = ATen glue (loss, head "
                                              "GEMMs, copies); "
                                              "the timed region itself is `steps` full iterations with no event records, no host read"}
        if fwd_only is not None:
            line["extra"] = {"forward_only": fwd_only}
        if world == 1 and not ddp and graphed is not None and not args.no_step_variants:
            try:
                line.update(step_variants(trainer, model, samples, targets, args))
            except Exception as e:                    # side figures only: the line and its `value` are complete without them
                line["step_variants_error"] = f"{type(e).__name__}: {e}"[:300]
        if world == 1 and not ddp and not args.no_step_variants:
            line.update(ddp_side_figure(args))
        if world == 1 and not args.no_ssl_side and not args.no_step_variants:
            line.update(ssl_side_figure())
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), file=real_stdout, flush=True)
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
