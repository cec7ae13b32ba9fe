#!/bin/bash
# One call for the round's evidence: GPU test suite, profiles (tools/profile_all.sh), the default bench line and
# the other scenes.
set -e
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
./tools/profile_all.sh c5 t1m c3 c2
timeout -k 10 900 python bench.py > gpurun_out/bench_final.log 2>&1
echo "bench done"
# N = 2 rehearsal of the sharded flow on this one GPU (gloo over pinned host buffers), and the one-rank RCCL path
PHYS_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 100 > gpurun_out/bench_n2_rehearsal.log 2>&1 && tail -c 400 gpurun_out/bench_n2_rehearsal.log | head -c 200; echo
PHYS_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 timeout -k 10 300 python bench.py --steps 100 > gpurun_out/bench_dist1.log 2>&1 && echo "one-rank RCCL ok"
for wl in c1 c3 c4 c5; do
  timeout -k 10 400 python bench.py --workload $wl --no-cpu-baseline --no-extra > gpurun_out/fin_$wl.log 2>&1
  python - <<PY
import json
d=json.loads(open("gpurun_out/fin_$wl.log").read().strip().splitlines()[-1])
print("$wl", d["steps_per_sec"], d["ms_per_step"], d["pairs_per_sec"], d["scene_stats"], d.get("roofline",{}).get("kernel"), d.get("roofline",{}).get("frac"))
PY
done
