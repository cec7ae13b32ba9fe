#!/bin/bash
# statistics of the lockstep probe (tools/race_probe.py): six runs per solver mode
run() { echo "== $*"; for i in 1 2 3 4 5 6; do timeout -k 10 100 python tools/race_probe.py 300 "$@" 2>&1 | grep -v amdgpu.ids | grep "^step\|no difference\|manifold\|missing\|extra" | cut -c1-${WIDTH:-90}; done; }
run percolour
run flow
