#!/bin/bash
# the whole GPU suite in ONE invocation, N times back to back (consecutive single-invocation runs)
N=${1:-1}; TAG=${2:-suite}; O=${OUT_ROOT:-gpurun_out}/${SUITE_DIR:-r04d}; mkdir -p $O
for i in $(seq 1 $N); do
  t0=$SECONDS
  timeout 1500 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=25 > $O/${TAG}_$i.log 2>&1
  echo "rc=$? wall_s=$((SECONDS - t0))" >> $O/${TAG}_$i.log
  grep -E "passed|failed" $O/${TAG}_$i.log | tail -1; tail -1 $O/${TAG}_$i.log
done
