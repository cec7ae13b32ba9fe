#!/usr/bin/env python3
"""The results table of DESIGN.md section 6 from a bench.py JSON line:  python tools/results_table.py profiles/r3_bench.json"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
rows = [("**C5** 256k-box tower (headline)", d)] + [(v.get("workload", k), v) for k, v in d.get("other_workloads", {}).items()]
print("| workload | bodies | manifolds / colours | steps/s (ms per step), median | 64 consecutive updates: steps/s, slowest, worst vs its neighbours | roofline kernel (live HIP events) | algorithmic rate = frac of 8 TB/s | PMC traffic vs algorithmic bytes per launch | CPU oracle steps/s: 1 thread / OpenMP |")
print("|---|---|---|---|---|---|---|---|---|")
for name, v in rows:
    st = v["scene_stats"]
    r = v.get("roofline", {})
    fp = v.get("full_period")
    cb = v.get("cpu_baseline", {})
    per = f"{fp['steps_per_sec']:.0f}, {fp['slowest_step_ms']:.2f} ms, {fp['worst_step_over_median_of_its_8_neighbours']:.2f}×" if fp else "-"
    tr = r.get("traffic")
    trs = f"{tr / 1e9:.3f} GB vs {r['algorithmic_bytes_per_launch'] / 1e9:.3f} GB" if tr else f"- vs {r.get('algorithmic_bytes_per_launch', 0) / 1e9:.3f} GB"
    omp = cb.get("openmp", {})
    cpu = f"{cb.get('steps_per_sec', 0):.4g} / {omp.get('steps_per_sec', 0):.4g} ({omp.get('cores', '-')} cores)" if cb else "-"
    print(f"| {name} | {st['n_bodies']:,} | {st['n_manifolds']:,} / {st['n_colors']} | **{v['steps_per_sec']:.0f}** ({v['ms_per_step']:.3f} ms) | {per} | "
          f"`{r.get('kernel', '-')}` {r.get('avg_launch_us', 0):.1f} µs × {r.get('launches_per_step', 0):g} | "
          f"{r.get('achieved', 0):.0f} GB/s = {r.get('frac', 0):.3f} | {trs} | {cpu} |")
print()
for name, v in rows:
    if "stages" in v:
        print(f"{name} stages (ms per step, HIP events): " + ", ".join(f"{k} {s['ms_per_step']:.3f}" for k, s in v["stages"].items()) + ".")
if "other_workloads" in d and "cg" in d["other_workloads"].get("ref_cg", {}):
    print("ref_cg:", {k: v for k, v in d["other_workloads"]["ref_cg"]["cg"].items() if k != "note"})
