#!/bin/bash
# development helper: rocprofv3 evidence for the BASELINE configs beside the headline one (1, 3a, 3b, 4) at their stated sizes:
#   kernel-trace stats of one bench step (the whole job) + three --pmc passes at 128 spp, per config, under gpurun_out/prof_configs/
# tools/profile_collect_configs.py <round> turns them into profiles/r<NN>_kernel_stats_<cfg>.csv and r<NN>_pmc_<cfg>.json
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_configs
cd /tmp && export TMPDIR=/tmp
cd $R
rm -rf $O; mkdir -p $O
cfg() {
  case $1 in
    c1) echo "--scene cornell_coat.json --width 512 --height 512 --spp 64";;
    c3a) echo "--scene cornell_roughcond.json --env sky --spp 4096";;
    c3b) echo "--scene cornell_roughdiel.json --env sky --spp 4096";;
    c4) echo "--scene cornell_media.json --env sky --phase hg --spp 4096";;
  esac
}
for c in ${CONFIGS:-c1 c3a c3b c4}; do
  args=$(cfg $c)
  steps=1; warm=0; [ $c = c1 ] && { steps=5; warm=1; }
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$c -- python3 bench.py --no-cpu-baseline --steps $steps --warmup $warm $args > $O/bench_$c.log 2>&1
  cp $(find $O/stats_$c -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$c.csv
  grep '^{' $O/bench_$c.log | tail -1 > $O/bench_$c.json
  rm -rf $O/stats_$c
  pargs=$(echo $args | sed 's/--spp 4096/--spp 128/')
  i=0
  for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM_WR" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $O/pmc_${c}_$i -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 $pargs > $O/pmc_${c}_$i.log 2>&1
    python3 tools/pmc_sum.py $O/pmc_${c}_$i > $O/pmc_${c}_$i.json
    rm -rf $O/pmc_${c}_$i
  done
  echo "$c done"; cat $O/bench_$c.json | cut -c1-200
done
ls $O
