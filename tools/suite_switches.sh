#!/bin/bash
# The GPU suite under the fallback switches (re-record the eager run after the env fix) -> $1/gpu_suite_switches.txt
out=${1:-gpurun_out/r06}; mkdir -p $out; log=$out/gpu_suite_switches.txt; : > $log
for sw in "PANDORA_HIPGRAPH=0" "PANDORA_CFG_STREAMS=0" "PANDORA_CFG_BATCH=1" "PANDORA_STATS_I64=0" "PANDORA_LN_PAIR_STREAM=0"; do
  t0=$(date +%s)
  env $sw timeout 1500 python -m pytest tests/ -q -m "gpu and not slow" -p no:cacheprovider 2>&1 | tail -1 > $out/.last
  echo "$sw : $(cat $out/.last)  [$(( $(date +%s) - t0 )) s wall]" >> $log
done
rm -f $out/.last; cat $log
