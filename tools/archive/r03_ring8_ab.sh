#!/bin/bash
# Round-3 A/B of the 8-loader ring kernel (PANDORA_GEMM_RING8=1: wherever the ring kernel is chosen and the variant is legal).
out=gpurun_out/r03
mkdir -p $out
PANDORA_GEMM_RING8=1 PANDORA_GEMM_RING=2 timeout 900 python -m pytest tests/test_ops_gpu.py -x -q > $out/ops_ring8_forced.log 2>&1
echo "ops forced rc=$?" >> $out/ops_ring8_forced.log
tail -3 $out/ops_ring8_forced.log
for v in 0 1; do PANDORA_GEMM256=0 PANDORA_GEMM_RING=2 PANDORA_GEMM_RING8=$v timeout 300 python tools/gemm256_probe.py 2>&1 | grep "^M=" > $out/probe_ring8_$v.txt; done
paste -d'\n' $out/probe_ring8_0.txt $out/probe_ring8_1.txt
for res in 576x1024 320x512; do
  for v in 0 1; do
    PANDORA_GEMM_RING8=$v timeout 600 python tools/shape_profile.py --res $res > $out/shape_${res}_ring8_$v.txt 2>&1
    grep "^# $res" $out/shape_${res}_ring8_$v.txt
  done
  for op in gemm ln_gemm conv3x3 conv_t3; do python tools/shape_ab.py $out/shape_${res}_ring8_0.txt $out/shape_${res}_ring8_1.txt $op | tail -30; done
done
