"""Per-kernel statistics of the LAST K steps of a rocprofv3 --kernel-trace run of bench.py (= its profile pass,
the steps bench.py times with HIP events for the `roofline` object), so the two clocks are compared on the same
launches. A step ends with k_step_position.

    python tools/trace_tail.py gpurun_out/prof_c2/<host>/<pid>_kernel_trace.csv 100 > profiles/r1_c2_profile_pass_stats.csv
"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
k = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [int(r["End_Timestamp"]) for r in rows if "k_step_position" in r["Kernel_Name"]]
t0 = ends[-k - 1] if len(ends) > k else 0
d = collections.defaultdict(list)
for r in rows:
    if int(r["Start_Timestamp"]) >= t0:
        d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "CallsPerStep", "MinNs", "MaxNs"])
for name, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([name, len(v), sum(v), round(sum(v) / len(v), 3), round(len(v) / k, 3), min(v), max(v)])
