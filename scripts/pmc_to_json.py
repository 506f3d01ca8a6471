#!/usr/bin/env python3
"""Builds a PMC summary (profiles/pmc_traffic.json by default) from the rocprofv3 --pmc passes of scripts/pmc_pass.sh (one counter group
per pass) for ONE kernel of the scoring step: the QKV projection.
usage: python3 scripts/pmc_to_json.py gpurun_out/<dir> [out.json]
env: PMC_KERNEL = substring of the kernel name (default: the exact-fp32 LEAN instance HEAD runs), PMC_BYTES_IN = operand bytes per element
(4: fp32 operands; 6 / 4: three / two bf16 planes), PMC_NOTE = free text."""
import csv, glob, json, os, sys, collections

src = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
KERNEL = os.environ.get("PMC_KERNEL", "gemm_f32_kernel<128, 128, 32, true, true, 0, 0, true>")
BYTES_IN = float(os.environ.get("PMC_BYTES_IN", "4"))
vals = collections.defaultdict(list)
names = set()
for f in glob.glob(os.path.join(src, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
            names.add(r["Kernel_Name"])
assert vals, f"no {KERNEL} rows under {src}"
assert len(names) == 1, f"{KERNEL} matches several kernels: {sorted(names)}"
mean = {k: sum(v) / len(v) for k, v in sorted(vals.items())}
M, N, K = 12003, 3072, 1024
fetch_kb, write_kb = mean["FETCH_SIZE"], mean["WRITE_SIZE"]
doc = {
    "source": "rocprofv3 --kernel-trace --pmc <group> (one group per pass, scripts/pmc_pass.sh + scripts/pmc_to_json.py) on "
              "`python bench.py --steps 3 --warmup 2 --headline-only`, MI355X, round 5",
    "kernel": f"{sorted(names)[0]} (QKV projection, M={M} N={N} K={K})",
    "launches_averaged": len(vals["FETCH_SIZE"]),
    "counters_mean_per_launch": mean,
    "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
    "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide (16 B/lane) loads -> doubled "
                  "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
    "gemm_qkv_hbm_bytes_per_launch": int(round((2 * fetch_kb + write_kb) * 1024)),
    "algorithmic_bytes_per_launch": int(BYTES_IN * (M * K + N * K) + 4 * M * N),
    "note": os.environ.get("PMC_NOTE", "reads: the A operand is fetched by 4 of the 8 XCD L2s and one weight quarter per XCD under the 2x4 XCD tile map; "
                                       "Infinity-Cache hits are counted by these counters"),
}
if mean.get("SQ_VALU_MFMA_BUSY_CYCLES") and mean.get("GRBM_GUI_ACTIVE"):
    doc["mfma_busy_frac"] = round(mean["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (mean["GRBM_GUI_ACTIVE"] / 8.0), 4)
json.dump(doc, open(out, "w"), indent=1)
print(f"{sorted(names)[0][:90]}: traffic {doc['gemm_qkv_hbm_bytes_per_launch']/1e6:.1f} MB vs algorithmic {doc['algorithmic_bytes_per_launch']/1e6:.1f} MB; "
      f"MFMA busy {doc.get('mfma_busy_frac')}; LDS conflict cycles {mean.get('SQ_LDS_BANK_CONFLICT', 0):.0f}")
