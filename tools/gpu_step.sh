#!/bin/bash
# gpu_step.sh <tag> <timeout seconds> <command...>: one GPU step under its own timeout, output to gpurun_out/<tag>.log,
# a one-line verdict on stdout. Exit code = the step's (so steps chained with && stop at the first failure / time-out).
tag=$1; limit=$2; shift 2
mkdir -p gpurun_out
timeout -k 10 "$limit" "$@" > "gpurun_out/$tag.log" 2>&1
rc=$?
echo "[$tag] rc=$rc $(tail -1 gpurun_out/$tag.log | cut -c1-200)"
exit $rc
