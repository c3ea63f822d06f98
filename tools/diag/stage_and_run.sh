#!/bin/bash
# usage: tools/diag/stage_and_run.sh <timeout_s> <logfile> <command run inside the staged copy>
# gpurun snapshots /root/repo when it gets a GPU slot - possibly minutes after it was started, in the middle of later edits
# (it happened: r04b ran a half-edited unet.py).  This helper FREEZES a copy of the tree under gpurun_stage/ first and runs the
# command from there; outputs go to the snapshot root's gpurun_out/ (OUT_ROOT), which gpurun merges back.
T=$1; L=$2; shift 2
cd /root/repo
rm -rf gpurun_stage && mkdir gpurun_stage
tar --exclude=./.git --exclude=./gpurun_out --exclude=./gpurun_stage --exclude=.pytest_cache --exclude=__pycache__ -cf - . | tar -xf - -C gpurun_stage
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "export OUT_ROOT=\$PWD/gpurun_out; cd gpurun_stage && $*" > $L 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
