#!/usr/bin/env python3
"""Builds profiles/pmc_traffic.json from the rocprofv3 --pmc passes of scripts/pmc_pass.sh (one counter group per pass).
usage: python3 scripts/pmc_to_json.py gpurun_out/<dir> [out.json]
The QKV projection is the one NT / EPI_NONE launch of the 128x128 fp32 tile per scoring step."""
import csv, glob, json, os, sys, collections

src = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
# round 3: the QKV projection runs the LEAN (buffer-load) instance of the 128x128 NT tile
KERNEL = os.environ.get("PMC_KERNEL", "gemm_f32_kernel<128, 128, 32, true, true, 0, 0, false, true>")
vals = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
assert vals, f"no {KERNEL} rows under {src}"
mean = {k: sum(v) / len(v) for k, v in sorted(vals.items())}
M, N, K = 12003, 3072, 1024
fetch_kb, write_kb = mean["FETCH_SIZE"], mean["WRITE_SIZE"]
doc = {
    "source": "rocprofv3 --kernel-trace --pmc <group> (one group per pass, scripts/pmc_pass.sh + scripts/pmc_to_json.py) on "
              "`python bench.py --steps 3 --warmup 2`, MI355X, round 3 (LEAN buffer-load instance, peeled last k-tile)",
    "kernel": f"sumk::{KERNEL} (QKV projection, M={M} N={N} K={K})",
    "launches_averaged": len(vals["FETCH_SIZE"]),
    "counters_mean_per_launch": mean,
    "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
    "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide (16 B/lane) loads -> doubled "
                  "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
    "gemm_qkv_hbm_bytes_per_launch": int(round((2 * fetch_kb + write_kb) * 1024)),
    "algorithmic_bytes_per_launch": 4 * (M * K + N * K + M * N),
    "note": "reads: X (49.2 MB) is fetched by 4 of the 8 XCD L2s and one weight quarter per XCD under the 2x4 XCD tile map; "
            "Infinity-Cache hits are counted by these counters",
}
json.dump(doc, open(out, "w"), indent=1)
busy = mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0 / max(1.0, mean.get("GRBM_GUI_ACTIVE", 1) / 8.0)
print(f"traffic {doc['gemm_qkv_hbm_bytes_per_launch']/1e6:.1f} MB vs algorithmic {doc['algorithmic_bytes_per_launch']/1e6:.1f} MB; MFMA busy {busy:.3f}; "
      f"LDS conflict cycles {mean.get('SQ_LDS_BANK_CONFLICT', 0):.0f}")
