#!/bin/bash
# r04 first GPU call: (1) reproduce the r03 in-process abort with stderr visible, (2) the child-process form, (3) the whole
# GPU suite in one invocation with per-test durations.
O=gpurun_out/r04a; mkdir -p $O
export TORCH_SHOW_CPP_STACKTRACES=1
for i in 1 2 3; do
  timeout 400 python3 -X faulthandler -m pytest tools/diag/segmented_inprocess_r03.py -x -q -s -p no:cacheprovider > $O/inproc_$i.log 2>&1
  echo "rc=$?" >> $O/inproc_$i.log
done
timeout 1200 python3 -X faulthandler -m pytest tests/test_peer_gpu.py tools/diag/segmented_inprocess_r03.py -x -q -s -m gpu -p no:cacheprovider > $O/inproc_after_peer.log 2>&1
echo "rc=$?" >> $O/inproc_after_peer.log
for i in 1 2; do
  timeout 600 python3 -m pytest tests/test_segmented_gpu.py -x -q -p no:cacheprovider > $O/child_$i.log 2>&1
  echo "rc=$?" >> $O/child_$i.log
done
timeout 1700 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=70 > $O/full_suite.log 2>&1
echo "rc=$?" >> $O/full_suite.log
tail -5 $O/*.log
