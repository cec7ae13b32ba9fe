#!/usr/bin/env python3
"""Cut the timed WINDOW out of rocprofv3 runs of `bench.py --workload <wl> --profile-window` and write
profiles/r3_<wl>_window.json: per kernel the launches, average / min / max duration and (from the two PMC passes)
the HBM traffic per launch, together with the window and the scene state they were taken at - bench.py hands these
numbers out only for a run of the same window at the same state.

    python tools/profile_window.py <wl> <dir with trace_<wl>/, pmc_<wl>_FETCH_SIZE/, pmc_<wl>_WRITE_SIZE/, window_<wl>.json>

A step ends with k_step_position (k_step_full without the collision stages), so the window = everything after the (K+1)-th k_step_position from the end,
in dispatch order - the same rule for the kernel trace and for the per-dispatch counter tables.
Units / corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
half of the bytes of wide coalesced reads, so traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def window_rows(rows, k, name_key, order_key):
    rows.sort(key=lambda r: int(r[order_key]))
    ends = [i for i, r in enumerate(rows) if "k_step_position" in r[name_key] or "k_step_full" in r[name_key]]
    first = ends[-k - 1] + 1 if len(ends) > k else 0
    return rows[first:ends[-1] + 1]


def find(base, sub, pattern):
    hits = glob.glob(os.path.join(base, sub, "**", pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None  # the newest pass (a directory may hold earlier ones)


def main():
    wl, base = sys.argv[1], sys.argv[2]
    info = json.loads(open(os.path.join(base, f"window_{wl}.json")).read().strip().splitlines()[-1])
    win = info["profile_window"]
    k = win["steps"]
    out = {"window": win, "scene_stats": info["scene_stats"],
           "command": f"bench.py --workload {wl} --warmup {win['warmup']} --steps {k} --profile-window",
           "method": "rocprofv3 --kernel-trace (durations) and separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes; rows "
                     "of the last K steps only; traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch",
           "kernels": {}}
    trace = find(base, f"trace_{wl}", "*kernel_trace.csv")
    rows = window_rows(list(csv.DictReader(open(trace))), k, "Kernel_Name", "Start_Timestamp")
    dur = collections.defaultdict(list)
    for r in rows:
        dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
    out["window_device_span_ms_per_step"] = round(span / k / 1e6, 4)
    traffic = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        f = find(base, f"pmc_{wl}_{counter}", "*counter_collection.csv")
        if not f:
            continue
        prow = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
        prow = window_rows(prow, k, "Kernel_Name", "Dispatch_Id")
        d = collections.defaultdict(list)
        for r in prow:
            d[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        traffic[counter] = {kn: sum(v) / len(v) for kn, v in d.items()}
    for name, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        rec = {"launches": len(v), "launches_per_step": round(len(v) / k, 3), "avg_us": round(sum(v) / len(v) / 1e3, 3),
               "min_us": round(min(v) / 1e3, 3), "max_us": round(max(v) / 1e3, 3), "total_us": round(sum(v) / 1e3, 1)}
        fe, wr = traffic.get("FETCH_SIZE", {}).get(name), traffic.get("WRITE_SIZE", {}).get(name)
        if fe is not None and wr is not None:
            rec["fetch_kib"], rec["write_kib"] = round(fe, 1), round(wr, 1)
            rec["traffic_bytes"] = int((2 * fe + wr) * 1024)
        out["kernels"][name] = rec
    path = os.path.join(ROOT, "profiles", f"r3_{wl}_window.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)
    for name, r in list(out["kernels"].items())[:12]:
        print(f"  {name[-48:]:48s} x{r['launches_per_step']:7.2f}/step avg {r['avg_us']:9.2f} us  traffic {r.get('traffic_bytes')}")


if __name__ == "__main__":
    main()
