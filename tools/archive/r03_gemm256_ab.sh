#!/bin/bash
# Round-3 A/B of the 256x256 8-phase GEMM kernel (PANDORA_GEMM256 = 0 off | 1 by gemm256_wanted | 2 wherever legal):
# ops parity with the kernel forced everywhere it is legal, isolated shapes, then per-shape tables at both resolutions.
out=gpurun_out/r03
mkdir -p $out
PANDORA_GEMM256=2 timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "gemm or geglu or ln_gemm" > $out/ops_gemm256_forced.log 2>&1
echo "ops forced rc=$?" >> $out/ops_gemm256_forced.log
tail -3 $out/ops_gemm256_forced.log
for v in 0 2; do PANDORA_GEMM256=$v timeout 300 python tools/gemm256_probe.py > $out/probe_gemm256_$v.txt 2>&1; cat $out/probe_gemm256_$v.txt; done
for res in 576x1024 320x512; do
  for v in 0 2; do
    PANDORA_GEMM256=$v timeout 600 python tools/shape_profile.py --res $res > $out/shape_${res}_gemm256_$v.txt 2>&1
    grep "^# $res" $out/shape_${res}_gemm256_$v.txt
  done
  for op in gemm ln_gemm; do python tools/shape_ab.py $out/shape_${res}_gemm256_0.txt $out/shape_${res}_gemm256_2.txt $op | head -24; done
done
