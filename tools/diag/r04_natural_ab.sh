#!/bin/bash
# r04: natural column order of the all-f32 GEMM / conv epilogues (gemm_common.hpp cperm) against the interleaved form:
# ops parity first, then per-shape tables and the bench step with the diagnostics library, PANDORA_GEMM_NATURAL = 0 | 1
O=${OUT_ROOT:-gpurun_out}/r04j; mkdir -p $O
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -4 > $O/ops_tests.txt
export PANDORA_DIAG_LIB=1
for res in 576x1024 320x512; do
  for v in 0 1 0 1; do
    PANDORA_GEMM_NATURAL=$v timeout 300 python3 tools/shape_profile.py --res $res --reps 3 > $O/shape_${res}_nat${v}_$RANDOM.txt 2>&1
  done
done
for v in 0 1 0 1; do
  PANDORA_GEMM_NATURAL=$v timeout 300 python3 bench.py --steps 20 --warmup 5 --cpu-baseline off --emulate-shard off > $O/bench_nat${v}_$RANDOM.json 2>/dev/null
done
