#!/bin/bash
# build everything first (the GPU box runs the prebuilt in-tree .so files), then hand the command to gpurun
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > /tmp/build.log 2>&1 || { tail -30 /tmp/build.log; exit 1; }
mkdir -p gpurun_out
exec gpurun --timeout "${GPU_TIMEOUT:-1100}" -- "$@"
