"""One step's kernel timeline out of a rocprofv3 kernel-trace CSV.  usage: trace_timeline.py p_kernel_trace.csv <anchor kernel substring> <occurrence> [<occurrence> ...]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n)
    return n.replace("sumk::", "")[:70]
names = [short(r["Kernel_Name"]) for r in rows]
idx = [i for i, n in enumerate(names) if sys.argv[2] in n]
print(len(idx), "occurrences of", sys.argv[2])
for occ in sys.argv[3:]:
    a, b = idx[int(occ)], idx[int(occ) + 1]
    t0 = int(rows[a]["Start_Timestamp"])
    for i in range(a, b):
        s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
        print(f"{(s-t0)/1e3:8.1f} +{(e-s)/1e3:7.1f}  {names[i]}  grid={int(rows[i]['Grid_Size_X'])//int(rows[i]['Workgroup_Size_X'])}x{rows[i]['Workgroup_Size_X']}")
    print("period", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, "\n")
