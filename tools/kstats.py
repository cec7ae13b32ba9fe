"""Print a rocprofv3 kernel_stats.csv as per-step microseconds (steps = calls of k_step_position)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
calls = [int(r["Calls"]) for r in rows if "k_step_position" in r["Name"]][0]
tot = 0.0
for r in rows:
    n = r["Name"].split("(")[0][-40:]
    print(f"{n:42s} {r['Calls']:>7s} avg {float(r['AverageNs']) / 1e3:7.2f} us  per step {float(r['TotalDurationNs']) / calls / 1e3:8.2f} us")
    tot += float(r["TotalDurationNs"])
print(f"kernel time per step {tot / calls / 1e3:.1f} us over {calls} steps")
