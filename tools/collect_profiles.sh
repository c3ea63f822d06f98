#!/bin/bash
# Collect the round's measurement artefacts on the GPU box into gpurun_out/$1 (copy what is to be judged into
# profiles/$1 afterwards).  usage: tools/collect_profiles.sh r03 [part]   part = a (bench, kernel stats, tables, A/Bs)
# | b (PMC passes of the probe launches) | c (PMC traffic passes over one forward) | all.  Every command runs under its own timeout and scratch directories are removed as soon as they
# are summarised, so a slow pass costs its own result only.
set -u
R=${1:-r03}; PART=${2:-all}; O=gpurun_out/$R; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PMC_SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
if [ $PART = a ] || [ $PART = all ]; then
# 1. the driver's command, plain and under rocprofv3 --kernel-trace --stats
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline off --emulate-shard off > $O/bench_under_rocprof.json 2>/dev/null
cp $O/bench_trace/*/*kernel_stats.csv $O/bench_kernel_stats.csv; rm -rf $O/bench_trace
# 2. per-forward kernel breakdown (difference of two graph-free runs) at both resolutions
for res in 320x512 576x1024; do
  for n in 2 6; do
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fwd_${res}_$n -- python3 tools/fwd_only.py $n $res > /dev/null 2>&1
  done
  python3 tools/diff_stats.py $O/fwd_${res}_2/*/*kernel_stats.csv 2 $O/fwd_${res}_6/*/*kernel_stats.csv 6 > $O/forward_kernel_breakdown_$res.txt
  rm -rf $O/fwd_${res}_2 $O/fwd_${res}_6
  timeout 600 python3 tools/shape_profile.py --res $res --reps 3 > $O/per_shape_table_$res.txt 2>/dev/null
done
# 3. attention A/B (bf16 variants, ceiling probes 11/12/13, fp8), panel kernel A/B
timeout 600 python3 tools/attn_bench.py --rounds 7 --variants 9,1,3,5,11,12,13,101 > $O/attention_ab.txt 2>/dev/null
timeout 600 python3 tools/lngemm_bench.py --reps 30 > $O/lngemm_ab.txt 2>/dev/null
# 4. configs[4] (fp8 attention, 5 rounds)
timeout 900 python3 bench.py --steps 10 --warmup 3 --cpu-baseline off --emulate-shard off --fp8-attention --multiround 5 > $O/bench_fp8_multiround.json 2>/dev/null
fi
if [ $PART = b ] || [ $PART = all ]; then
INC="--kernel-include-regex pm"   # counters on this library's kernels only (the one-off weight synthesis runs unprofiled)
# 5. counters of the dominant kernels: SQ set, then FETCH_SIZE and WRITE_SIZE in passes of their own
timeout 600 rocprofv3 --kernel-trace --pmc $PMC_SQ $INC --output-format csv -d $O/pmc_sq -- python3 tools/gemm_probe.py > /dev/null 2>&1
python3 tools/pmc_table.py $O/pmc_sq > $O/pmc_sq.txt; rm -rf $O/pmc_sq
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE $INC --output-format csv -d $O/pmc_fetch -- python3 tools/gemm_probe.py > /dev/null 2>&1
python3 tools/pmc_table.py $O/pmc_fetch > $O/pmc_fetch.txt; rm -rf $O/pmc_fetch
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE $INC --output-format csv -d $O/pmc_write -- python3 tools/gemm_probe.py > /dev/null 2>&1
python3 tools/pmc_table.py $O/pmc_write > $O/pmc_write.txt; rm -rf $O/pmc_write
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE $INC --output-format csv -d $O/pmc_lds -- python3 tools/gemm_probe.py > /dev/null 2>&1
python3 tools/pmc_table.py $O/pmc_lds > $O/pmc_lds.txt; rm -rf $O/pmc_lds
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE $INC --output-format csv -d $O/pmc_calib -- python3 tools/fetch_calib.py > $O/fetch_calib.txt 2>/dev/null
python3 tools/pmc_table.py $O/pmc_calib >> $O/fetch_calib.txt; rm -rf $O/pmc_calib
timeout 600 rocprofv3 --kernel-trace --pmc $PMC_SQ $INC --output-format csv -d $O/pmc_attn -- python3 tools/attn_pmc.py > /dev/null 2>&1
python3 tools/pmc_table.py $O/pmc_attn > $O/pmc_attention_n9216.txt; rm -rf $O/pmc_attn
fi
if [ $PART = c ] || [ $PART = all ]; then
INC="--kernel-include-regex pm"
# 6. HBM-side bytes per launch of the families of one eager forward (FETCH_SIZE and WRITE_SIZE passes of their own) and of
#    the N = 9216 attention -> pmc_traffic.json (bench.py quotes it as roofline.traffic while the sources match)
args=""
for res in 320x512 576x1024; do
  date
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE $INC --output-format csv -d $O/tf_$res -- python3 tools/fwd_only.py 1 $res > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE $INC --output-format csv -d $O/tw_$res -- python3 tools/fwd_only.py 1 $res > /dev/null 2>&1
  args="$args $res=$O/tf_$res,$O/tw_$res"
done
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE $INC --output-format csv -d $O/tf_attn -- python3 tools/attn_pmc.py 9216 prod > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE $INC --output-format csv -d $O/tw_attn -- python3 tools/attn_pmc.py 9216 prod > /dev/null 2>&1
python3 tools/pmc_traffic.py $O/pmc_traffic.json $args attention_n9216=$O/tf_attn,$O/tw_attn > /dev/null
rm -rf $O/tf_* $O/tw_*
fi
du -sh $O; ls -la $O
