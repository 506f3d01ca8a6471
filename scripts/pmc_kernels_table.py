#!/usr/bin/env python3
"""Per-kernel table from the rocprofv3 --pmc passes of scripts/pmc_pass.sh (one counter group per pass): for every kernel of the profiled
command with at least MIN_CALLS launches, the mean per launch of each counter and the derived figures MI355X_MICROARCH.md defines --
matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 / (GRBM_GUI_ACTIVE / 8), L2 hit = TCC_HIT / (TCC_HIT + TCC_MISS), fetched bytes =
2 x FETCH_SIZE KB (gfx950: 64 B counted per 128-B request of wide loads), written bytes = WRITE_SIZE KB.
usage: python3 scripts/pmc_kernels_table.py gpurun_out/<dir> out.json [note]"""
import csv, glob, json, os, sys, collections

src, out = sys.argv[1], sys.argv[2]
MIN_CALLS = int(os.environ.get("PMC_MIN_CALLS", "5"))
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        vals[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = {}
for k, c in vals.items():
    n = max(len(v) for v in c.values())
    if n < MIN_CALLS or k.startswith("void at::") or "rocclr" in k:
        continue
    m = {name: sum(v) / len(v) for name, v in sorted(c.items())}
    d = {"launches_averaged": n, "counters_mean_per_launch": m}
    if m.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None and m.get("GRBM_GUI_ACTIVE"):
        d["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (m["GRBM_GUI_ACTIVE"] / 8.0), 4)
    if m.get("TCC_HIT_sum") is not None and (m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0)) > 0:
        d["l2_hit"] = round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 4)
    if m.get("FETCH_SIZE") is not None:
        d["fetched_mb"] = round(2 * m["FETCH_SIZE"] / 1024.0, 1)
    if m.get("WRITE_SIZE") is not None:
        d["written_mb"] = round(m["WRITE_SIZE"] / 1024.0, 1)
    if m.get("SQ_LDS_BANK_CONFLICT") is not None and m.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_frac"] = round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], 4)
    rows[k] = d
doc = {"source": "rocprofv3 --kernel-trace --pmc <group>, one group per pass (scripts/pmc_pass.sh), summarised by scripts/pmc_kernels_table.py",
       "note": sys.argv[3] if len(sys.argv) > 3 else "", "kernels": rows}
json.dump(doc, open(out, "w"), indent=1)
for k, d in sorted(rows.items(), key=lambda kv: -kv[1]["counters_mean_per_launch"].get("GRBM_GUI_ACTIVE", 0)):
    print(f"{k[:90]:90s} n={d['launches_averaged']:3d} mfma {d.get('mfma_busy_frac')} l2hit {d.get('l2_hit')} fetch {d.get('fetched_mb')} MB write {d.get('written_mb')} MB lds-conflict {d.get('lds_conflict_frac')}")
