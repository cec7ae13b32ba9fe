#!/bin/bash
# Round profile collection on the GPU box: for every workload the SAME command three times under rocprofv3 -
# kernel trace, then the two PMC passes (separate runs: the guide's rule, and gpurun refuses --pmc next to tracing
# domains other than the kernel trace). tools/profile_window.py then cuts the timed window out of each and writes
# profiles/r3_<wl>_window.json (run it here, on the merged gpurun_out/).
#   tools/profile_all.sh c5 t1m c3 c2 c4 t1m_settled ref_1m ref_cg
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_r3
mkdir -p $out
for wl in "$@"; do
  args="--workload $wl --warmup 5 --steps 20 --profile-window"
  python3 bench.py $args > $out/window_$wl.json 2> $out/window_$wl.err
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$wl -- python3 bench.py $args > $out/trace_$wl.log 2>&1
  echo "trace $wl done"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $out/pmc_${wl}_$c -- python3 bench.py $args > $out/pmc_${wl}_$c.log 2>&1
    echo "pmc $wl $c done"
  done
  # cut the timed window out on the box (the raw per-dispatch tables of eight workloads exceed what gpurun copies
  # back): profiles/r3_<wl>_window.json + the rocprofv3 --stats kernel summary of the same command
  python3 tools/profile_window.py $wl $out > $out/window_cut_$wl.log 2>&1 || cat $out/window_cut_$wl.log
  mkdir -p gpurun_out/profiles_r3
  cp profiles/r3_${wl}_window.json gpurun_out/profiles_r3/ 2>/dev/null || true
  stats=$(find $out/trace_$wl -name "*kernel_stats.csv" | head -1)
  [ -n "$stats" ] && cp "$stats" gpurun_out/profiles_r3/r3_${wl}_kernel_stats.csv
  rm -rf $out/trace_$wl $out/pmc_${wl}_FETCH_SIZE $out/pmc_${wl}_WRITE_SIZE
done
du -sh $out gpurun_out/profiles_r3
