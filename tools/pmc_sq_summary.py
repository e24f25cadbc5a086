"""Per-kernel SQ counter summary from a rocprofv3 --pmc pass (SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE), written as a small text table for profiles/.

  wait      = SQ_WAIT_ANY / SQ_WAVE_CYCLES        waves parked on s_waitcnt / s_barrier
  stall     = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   issue stalls (MFMA read-after-write, busy pipes)
  issue     = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)   matrix-pipe occupancy while the CU is busy
(the three wave fractions are disjoint and sum to ~1, MI355X_MICROARCH.md counter table).
Usage: python tools/pmc_sq_summary.py <dir> <out.txt> [min launches]"""
import collections
import csv
import glob
import sys

d, out = sys.argv[1], sys.argv[2]
minl = int(sys.argv[3]) if len(sys.argv) > 3 else 8
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            a = acc[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
rows = []
for k, cs in acc.items():
    g = lambda n: cs[n][0] / cs[n][1] if n in cs and cs[n][1] else 0.0
    n = max(v[1] for v in cs.values())
    wc = g("SQ_WAVE_CYCLES")
    if n < minl or wc <= 0:
        continue
    busy = g("SQ_BUSY_CU_CYCLES")
    rows.append((g("GRBM_GUI_ACTIVE") * n, k, n, g("GRBM_GUI_ACTIVE"), g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc,
                 g("SQ_ACTIVE_INST_ANY") / wc, g("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * busy) if busy else 0.0))
rows.sort(reverse=True)
with open(out, "w") as fh:
    fh.write("# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES "
             "SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py (per-launch averages; see tools/pmc_sq_summary.py)\n")
    fh.write(f"{'launches':>8} {'gpu_cycles':>11} {'wait':>6} {'stall':>6} {'issue':>6} {'mfma_busy':>9}  kernel\n")
    for _, k, n, cyc, w, s, a, m in rows[:40]:
        fh.write(f"{n:8d} {cyc:11.0f} {w:6.3f} {s:6.3f} {a:6.3f} {m:9.3f}  {k[:150]}\n")
print(open(out).read())
