#!/bin/bash
# build everything first (the GPU box runs the prebuilt in-tree .so files), then hand the command to gpurun.
# Exit code 3 of gpurun = no slot / box free right now, nothing ran and nothing was charged: wait and ask again
# (only that case is retried - a command that ran and failed is never started again from here).
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > /tmp/build.log 2>&1 || { tail -30 /tmp/build.log; exit 1; }
mkdir -p gpurun_out
for attempt in $(seq 1 ${GPU_ATTEMPTS:-20}); do
  gpurun --timeout "${GPU_TIMEOUT:-1100}" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 120
done
exit 3
