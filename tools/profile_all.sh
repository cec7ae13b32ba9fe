#!/bin/bash
# Round profile collection: rocprofv3 kernel-trace stats and the two PMC passes (separate runs) for the bench
# workload (c2) and the 1M target scene. Outputs under gpurun_out/; tools/kstats.py and profiles/collect_pmc.py
# turn them into the files kept under profiles/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for wl in c2 t1m; do
  # the same commands bench.py runs by default: C2 with its default 1000 timed steps, the 1M scene with 60
  args="--workload $wl --no-cpu-baseline --no-extra"; [ "$wl" = "t1m" ] && args="$args --steps 60"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$wl -- python3 bench.py $args > gpurun_out/prof_$wl.log 2>&1
  echo "trace $wl done"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 600 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${wl}_$c -- python3 bench.py $args > gpurun_out/pmc_${wl}_$c.log 2>&1
    echo "pmc $wl $c done"
  done
done
