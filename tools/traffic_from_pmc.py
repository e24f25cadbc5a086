"""Build profiles/traffic.json from two rocprofv3 --pmc passes over bench.py (FETCH_SIZE, then WRITE_SIZE).

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (both reported in KiB): on gfx950 FETCH_SIZE counts the 128-byte
requests of wide streaming reads at 64 bytes (MI355X_MICROARCH.md, section HBM), WRITE_SIZE is exact for 16-byte-per-
lane stores.  Kernels whose loads are narrower are marked "uncalibrated".
Usage: python tools/traffic_from_pmc.py <fetch_dir> <write_dir> <out.json>"""
import collections
import csv
import glob
import json
import re
import sys

# rocprof kernel symbol (substring, mangled or demangled) -> ops.KernelTimer name
NAMES = [
    (r"linear_d8_wreg_kernel(IDF16bLi0E|<__bf16, 0>)", "linear_d8_wreg_kernel<bf16,0>"),
    (r"linear_d8_wreg_kernel(IfLi1E|<float, 1>)", "linear_d8_wreg_kernel<f32,1>"),
    (r"linear_d8_ring_kernel(IDF16bDF16bLi0E|<__bf16, __bf16, 0)", "linear_d8_ring_kernel<bf16,bf16,0>"),
    (r"linear_d8_ring_kernel(IDF16bfLi1E|<__bf16, float, 1|<bool _Accum, 1)", "linear_d8_ring_kernel<bf16,f32,1>"),
    (r"wgrad_ring_kernel", "wgrad_ring_kernel<bf16>"),
    (r"attn_fwd(_persist)?_kernel|a80.{0,4}fwd(_os)?_kernel|fwd_os_kernel", "attn_fwd_kernel"),
    (r"dense_tn_kernel", "dense_tn_kernel<wgrad, all shapes>"),
    (r"dense_nt_kernel(ILi0ELi5E|<0, 5>)", "dense_nt_kernel<0, 5>"),
    (r"dense_nt_kernel(ILi0ELi4E|<0, 4>)", "dense_nt_kernel<0, 4>"),
    (r"dense_nt_kernel(ILi1ELi4E|<1, 4>)", "dense_nt_kernel<1>"),
    (r"dense_nt_kernel(ILi3ELi4E|<3, 4>)", "dense_nt_kernel<3>"),
    (r"a80.{0,6}bwd_kernel|bwd_kernel\(octic::AttnBwdArgs", "attn_bwd_kernel"),
    (r"heads_permute_kernel", "heads_permute_kernel<bf16>"),
    (r"lamb_stage1_kernel", "lamb_stage1_kernel"),
    (r"lamb_stage2_kernel", "lamb_stage2_kernel"),
    (r"attn_bwd_dq_kernel", "attn_bwd_dq_kernel"),
    (r"attn_bwd_dkv_kernel", "attn_bwd_dkv_kernel"),
    (r"ln_fwd_g8_kernel", "ln_fwd_kernel<bf16>"),
    (r"ln_bwd_g8_kernel", "ln_bwd_kernel<bf16>"),
    (r"dense_ln_fwd_kernel", "dense_ln_fwd_kernel<bf16>"),
    (r"dense_ln_bwd_tail_kernel", "dense_ln_bwd_tail_kernel<bf16>"),
    (r"dense_ln_bwd(_wide)?_kernel", "dense_ln_bwd_kernel<bf16>"),
    (r"dense_resid_ln_fwd_kernel", "dense_resid_ln_fwd_kernel<bf16>"),
    (r"dense_colsum_kernel", "dense_colsum_kernel"),
    (r"scale_residual_fwd_kernel", "scale_residual_fwd_kernel<bf16>"),
    (r"scale_residual_bwd_kernel", "scale_residual_bwd_kernel<bf16>"),
    (r"dense_gelu_bwd_kernel", "dense_gelu_bwd_kernel"),
    (r"gelu_fwd4?_kernel", "gelu_fwd_kernel<bf16>"),
    (r"gelu_bwd4?_kernel", "gelu_bwd_kernel<bf16>"),
    (r"Cijk_.*MT256x256x32", "library_gemm<wgrad, all shapes>"),
    (r"cast_rowscale_kernel", "cast_rowscale_kernel<bf16>"),
]


def collect(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                for pat, name in NAMES:
                    if re.search(pat, r["Kernel_Name"]):
                        a = acc[name]
                        a[0] += float(r["Counter_Value"])
                        a[1] += 1
                        break
    return {k: (s / n, n) for k, (s, n) in acc.items()}


fetch = collect(sys.argv[1], "FETCH_SIZE")
write = collect(sys.argv[2], "WRITE_SIZE")
out = {}
for name in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(name, (0.0, 0))
    w, nw = write.get(name, (0.0, 0))
    out[name] = {"fetch_size_kib": round(f, 1), "write_size_kib": round(w, 1), "launches_sampled": [nf, nw],
                 "hbm_bytes_per_launch": int((2 * f + w) * 1024),
                 "note": "2 x FETCH_SIZE + WRITE_SIZE (gfx950 wide-read correction); per launch, averaged over the "
                         "launches of this kernel in bench.py"}
    print(f"{name:42s} fetch {f / 1024:8.1f} MiB (x2)  write {w / 1024:8.1f} MiB  hbm {out[name]['hbm_bytes_per_launch'] / 1e6:8.1f} MB")
with open(sys.argv[3], "w") as fh:
    json.dump(out, fh, indent=1)
