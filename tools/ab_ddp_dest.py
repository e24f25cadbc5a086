"""Developer A/B (GPU box): eager DDP step on a one-rank RCCL group with / without the weight gradients written into DDP's
bucket views (train.DDP_GRADS_IN_BUCKETS) - child processes, alternating.  usage: ab_ddp_dest.py"""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
code = ("import sys; sys.argv=['p']; import octic_vits_amd.train as TR; TR.DDP_GRADS_IN_BUCKETS = {flag}; "
        "import runpy; runpy.run_path('" + os.path.join(here, "ddp_host_probe.py") + "', run_name='__main__')")
for rnd in range(2):
    for flag in (True, False):
        r = subprocess.run([sys.executable, "-c", code.format(flag=flag)], capture_output=True, text=True, cwd=os.path.dirname(here))
        line = [l for l in r.stdout.splitlines() if l.startswith("DDP_FLAT")]
        print(f"round {rnd} grads-in-buckets {flag}: " + (line[-1] if line else (r.stderr.strip().splitlines() or ['?'])[-1][:300]), flush=True)
