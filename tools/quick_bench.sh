#!/bin/bash
# quick_bench.sh <tag> <workloads...>: bench.py per workload (no CPU baseline / extras), one summary line each
tag=$1; shift
for wl in "$@"; do
  steps=200; [ "$wl" = "t1m" ] && steps=60; [ "$wl" = "c5" ] && steps=60
  timeout -k 10 400 python bench.py --workload $wl --steps $steps --no-cpu-baseline --no-extra > gpurun_out/qb_${tag}_${wl}.log 2>&1 || { echo "$wl FAILED"; tail -5 gpurun_out/qb_${tag}_${wl}.log; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/qb_${tag}_${wl}.log").read().strip().splitlines()[-1])
print("$wl", "steps/s", d["steps_per_sec"], "ms", d["ms_per_step"], {k:(v["ms_per_step"],v["launches_per_step"]) for k,v in d["stages"].items() if "solve" in k or k in ("color","rows","narrow","pairs")}, "roof", d["roofline"]["kernel"], d["roofline"]["frac"])
PY
done
