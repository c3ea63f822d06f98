#!/bin/bash
# Round-6 measurement artefacts on the GPU box -> $OUT_ROOT/r06 (copy what is to be judged into profiles/r06 afterwards).
# usage: tools/collect_profiles_r06.sh [a|b|c|all]   a = bench + kernel stats + per-forward tables; b = SQ counters of the
# attention shapes; c = HBM-side traffic per launch (FETCH_SIZE / WRITE_SIZE passes of their own) -> pmc_traffic.json
set -u
PART=${1:-all}; O=${OUT_ROOT:-gpurun_out}/r06; mkdir -p $O
export TMPDIR=/tmp
PMC_SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
INC="--kernel-include-regex pm"
if [ $PART = a ] || [ $PART = all ]; then
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline off --emulate-shard off --parity-mode off > $O/bench_under_rocprof.json 2>/dev/null
cp $O/bench_trace/*/*kernel_stats.csv $O/bench_kernel_stats.csv; rm -rf $O/bench_trace
for res in 320x512 576x1024; do
  for n in 2 6; do
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fwd_${res}_$n -- python3 tools/fwd_only.py $n $res > /dev/null 2>&1
  done
  python3 tools/diff_stats.py $O/fwd_${res}_2/*/*kernel_stats.csv 2 $O/fwd_${res}_6/*/*kernel_stats.csv 6 > $O/forward_kernel_breakdown_$res.txt
  rm -rf $O/fwd_${res}_2 $O/fwd_${res}_6
done
timeout 600 python3 bench.py --steps 10 --warmup 3 --dtype f16 --cpu-baseline off --emulate-shard off --parity-mode off > $O/bench_f16.json 2> $O/bench_f16.err
timeout 900 python3 bench.py --steps 10 --warmup 3 --cpu-baseline off --emulate-shard off --parity-mode off --fp8-attention --multiround 5 > $O/bench_fp8_multiround.json 2>/dev/null
fi
if [ $PART = b ] || [ $PART = all ]; then
timeout 600 rocprofv3 --kernel-trace --pmc $PMC_SQ $INC --output-format csv -d $O/pmc_attn -- python3 tools/attn_pmc.py > /dev/null 2>&1
python3 tools/pmc_table.py $O/pmc_attn > $O/pmc_attention_n9216.txt; rm -rf $O/pmc_attn
fi
if [ $PART = c ] || [ $PART = all ]; then
args=""
for res in 320x512 576x1024; do
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE $INC --output-format csv -d $O/tf_$res -- python3 tools/fwd_only.py 1 $res > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE $INC --output-format csv -d $O/tw_$res -- python3 tools/fwd_only.py 1 $res > /dev/null 2>&1
  args="$args $res=$O/tf_$res,$O/tw_$res"
done
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE $INC --output-format csv -d $O/tf_attn -- python3 tools/attn_pmc.py 9216 prod > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE $INC --output-format csv -d $O/tw_attn -- python3 tools/attn_pmc.py 9216 prod > /dev/null 2>&1
python3 tools/pmc_traffic.py $O/pmc_traffic.json $args attention_n9216=$O/tf_attn,$O/tw_attn > /dev/null
rm -rf $O/tf_* $O/tw_*
fi
ls -la $O
