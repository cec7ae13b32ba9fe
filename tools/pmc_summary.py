"""Per-kernel averages of every counter found under <dir>/set*/ (rocprofv3 --pmc, csv): the last third of the dispatches
(the settled part of the run) of the five most expensive kernels."""
import collections
import csv
import glob
import sys

base = sys.argv[1]
table = collections.defaultdict(dict)
for f in sorted(glob.glob(base + "/set*/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    rows = rows[len(rows) * 2 // 3:]
    acc = collections.defaultdict(list)
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (name, c), v in acc.items():
        table[name][c] = sum(v) / len(v)
        table[name]["_n"] = len(v)
for name, d in sorted(table.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0) * kv[1].get("_n", 0)):
    print(name)
    for c, v in sorted(d.items()):
        print(f"    {c:28s} {v:16.1f}")
