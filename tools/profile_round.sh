#!/bin/bash
# development helper: the rocprofv3 evidence of one round (kernel-trace stats of the default bench, HBM byte counters,
# SQ / cache counters), written under gpurun_out/prof_round/ ; tools/profile_collect.py assembles profiles/ from them
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_round
cd /tmp && export TMPDIR=/tmp
cd $R
if [ "$1" = "stats" ]; then     # "stats": only the kernel-trace run of the default bench (after profile_collect.py has written the round's
  mkdir -p $O                   # traffic file, so that the bench line of the committed run carries the committed traffic figure)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py > $O/bench_under_rocprof.log 2>&1
  cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
  grep '^{' $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
  rm -rf $O/stats
  exit 0
fi
if [ "$1" != "dragon" ]; then   # "dragon": only the big-mesh counter passes (merged over the earlier files of the round)
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py > $O/bench_under_rocprof.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
grep '^{' $O/bench_under_rocprof.log | tail -1 > $O/bench_under_rocprof.json
rm -rf $O/stats
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc_$c.log 2>&1
  python3 tools/pmc_sum.py $O/pmc_$c > $O/pmc_$c.json
  rm -rf $O/pmc_$c
done
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQC_ICACHE_REQ SQC_ICACHE_MISSES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_set$i -- python3 bench.py --no-cpu-baseline --spp 128 --steps 1 --warmup 0 > $O/pmc_set$i.log 2>&1
  python3 tools/pmc_sum.py $O/pmc_set$i > $O/pmc_set$i.json
  rm -rf $O/pmc_set$i
done
echo done; ls $O
fi
mkdir -p $O
# the 871 k-triangle stand-in at BASELINE config 5's resolution (3840x2160, one GPU, few spp): where a big tree spends its time
j=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  j=$((j+1))
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_dragon$j -- python3 bench.py --no-cpu-baseline --scene cornell_dragon.json --width 3840 --height 2160 --spp 8 --steps 1 --warmup 0 > $O/pmc_dragon$j.log 2>&1
  python3 tools/pmc_sum.py $O/pmc_dragon$j > $O/pmc_dragon$j.json
  rm -rf $O/pmc_dragon$j
done
echo done; ls $O
