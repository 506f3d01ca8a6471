"""Kernel timeline / per-kernel statistics out of a rocprofv3 results .db (rocpd sqlite schema) -- used when the CSV writer of
rocprofv3 did not run (the profiled python process crashed at exit).  usage: rocpd_timeline.py p_results.db [anchor_kernel [occurrence]]"""
import re
import sqlite3
import sys
import collections


def load(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    return c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    return n[:100]


if __name__ == "__main__":
    rows = load(sys.argv[1])
    names = [short(r[0]) for r in rows]
    if len(sys.argv) > 2:
        anchor, occ = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 100
        idx = [i for i, n in enumerate(names) if anchor in n]
        a, b = idx[occ], idx[occ + 1]
        t0 = rows[a][1]
        for i in range(a, b):
            print(f"{(rows[i][1]-t0)/1e3:9.1f} us +{(rows[i][2]-rows[i][1])/1e3:8.1f}  {names[i]}")
        print("period us", (rows[b][1] - rows[a][1]) / 1e3, " kernel sum us", sum(rows[i][2] - rows[i][1] for i in range(a, b)) / 1e3)
    else:
        st = collections.defaultdict(list)
        for n, r in zip(names, rows):
            st[n].append((r[2] - r[1]) / 1e3)
        tot = sum(sum(v) for v in st.values())
        print("name,calls,total_us,avg_us,min_us,max_us,pct")
        for n, v in sorted(st.items(), key=lambda kv: -sum(kv[1])):
            print(f"\"{n}\",{len(v)},{sum(v):.1f},{sum(v)/len(v):.2f},{min(v):.2f},{max(v):.2f},{100*sum(v)/tot:.2f}")
