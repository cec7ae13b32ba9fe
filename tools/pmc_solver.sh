#!/bin/bash
# Hardware counters of the C5 solver kernels (diagnosis, not a committed profile): one rocprofv3 --pmc pass per counter
# set over `bench.py --workload c5 --profile-window`, per-kernel averages printed by tools/pmc_summary.py.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_diag
mkdir -p $out
wl=${1:-c5}
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $out/set$i -- python3 bench.py --workload $wl --warmup 5 --steps 10 --profile-window > $out/set$i.log 2>&1 || { echo "set $i ($set) failed"; tail -3 $out/set$i.log; }
  echo "set $i done"
done
python3 tools/pmc_summary.py $out > $out/summary.txt 2>&1 || true
tail -40 $out/summary.txt
find $out -name "*.csv" -size +20M -delete
