#!/bin/bash
# Round-3 A/B of the 256x128 ring kernel (PANDORA_GEMM_RINGW = 0 off | 1 by prefer_ringw | 2 wherever the ring kernel runs unsplit).
out=gpurun_out/r03
mkdir -p $out
V=${1:-2}
PANDORA_GEMM_RINGW=2 PANDORA_GEMM_RING=2 timeout 900 python -m pytest tests/test_ops_gpu.py -x -q > $out/ops_ringw_forced.log 2>&1
echo "ops forced rc=$?" >> $out/ops_ringw_forced.log
tail -3 $out/ops_ringw_forced.log
for res in 576x1024 320x512; do
  for v in 0 $V; do
    PANDORA_GEMM_RINGW=$v timeout 600 python tools/shape_profile.py --res $res > $out/shape_${res}_ringw_$v.txt 2>&1
    grep "^# $res" $out/shape_${res}_ringw_$v.txt
  done
  for op in gemm ln_gemm conv3x3 conv_t3; do python tools/shape_ab.py $out/shape_${res}_ringw_0.txt $out/shape_${res}_ringw_$V.txt $op | tail -${2:-14}; done
done
