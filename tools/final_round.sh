#!/bin/bash
# One call for the round's evidence: GPU test suite, profiles (tools/profile_all.sh), the default bench line and
# the other scenes.
set -e
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
./tools/profile_all.sh
timeout -k 10 900 python bench.py > gpurun_out/bench_final.log 2>&1
echo "bench done"
for wl in c1 c3 c4 c5; do
  timeout -k 10 400 python bench.py --workload $wl --no-cpu-baseline --no-extra --steps 200 > gpurun_out/fin_$wl.log 2>&1
  python - <<PY
import json
d=json.loads(open("gpurun_out/fin_$wl.log").read().strip().splitlines()[-1])
print("$wl", d["steps_per_sec"], d["ms_per_step"], d["pairs_per_sec"], d["scene_stats"], d.get("roofline",{}).get("kernel"), d.get("roofline",{}).get("frac"))
PY
done
