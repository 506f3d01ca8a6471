#!/usr/bin/env python3
"""profiles/r04_pmc_attn_strips.json from the rocprofv3 --pmc passes of `scripts/pmc_pass.sh <dir> --headline-only --mode train --precision bf16`
(one counter group per pass): the two fused attention strip launches of the bf16 training step (csrc/attn_b16.hip) and, beside them, the
merged dV + dK launch.  usage: python3 scripts/pmc_attn_strips.py gpurun_out/<dir> [out.json]"""
import csv, glob, json, os, sys, collections
src = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_pmc_attn_strips.json")
KERNELS = {"attn_strip_kernel<false>": "forward strip (logits, softmax + dropout, alpha.V)", "attn_strip_kernel<true>": "backward strip (dC.V^T, softmax backward, dS.K)",
           "gemm_b16_kernel<false, false, 0>": "dV + dK, one launch of 2 n_seq problems"}
vals = {k: collections.defaultdict(list) for k in KERNELS}
for f in glob.glob(os.path.join(src, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        for k in KERNELS:
            if k in r["Kernel_Name"]:
                vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
doc = {"source": "scripts/pmc_pass.sh <dir> --headline-only --mode train --precision bf16 (separate rocprofv3 --kernel-trace --pmc passes: FETCH_SIZE | WRITE_SIZE | "
                 "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES | SQ_VALU_MFMA_BUSY_CYCLES ... | SQ_LDS_* | TCC_HIT_sum TCC_MISS_sum), S-TVSum batch (50 videos, 12 003 frames, D = 1024), "
                 "per launch, averaged over the launches of each kernel; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md",
       "kernels": {}}
for k, what in KERNELS.items():
    v = vals[k]
    if not v:
        continue
    m = {c: sum(x) / len(x) for c, x in sorted(v.items())}
    e = {"what": what, "launches_averaged": len(v.get("FETCH_SIZE", [])), "counters_mean_per_launch": m}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["hbm_side_bytes_per_launch"] = int((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        e["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / max(1.0, m["GRBM_GUI_ACTIVE"] / 8.0), 4)     # 1024 SIMDs; GRBM counts per XCD
    if "TCC_HIT_sum" in m:
        e["l2_hit_frac"] = round(m["TCC_HIT_sum"] / max(1.0, m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 4)
    if "SQ_LDS_BANK_CONFLICT" in m:
        e["lds_bank_conflict_frac_of_lds_cycles"] = round(m["SQ_LDS_BANK_CONFLICT"] / max(1.0, m["SQ_LDS_IDX_ACTIVE"]), 4)
    doc["kernels"][k] = e
json.dump(doc, open(out, "w"), indent=1)
for k, e in doc["kernels"].items():
    print(k, {x: e.get(x) for x in ("mfma_busy_frac", "l2_hit_frac", "lds_bank_conflict_frac_of_lds_cycles", "hbm_side_bytes_per_launch")})
