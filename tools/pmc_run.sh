#!/bin/bash
# development helper: one rocprofv3 --pmc pass per argument (a quoted, space-separated counter list), summed per counter
#   tools/pmc_run.sh "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQC_ICACHE_REQ SQC_ICACHE_MISSES"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "$@"; do
  i=$((i+1))
  d=$R/gpurun_out/pmc_$i
  rm -rf $d
  (cd $R && rocprofv3 --pmc $set --output-format csv -d $d -- python3 bench.py --no-cpu-baseline --spp ${SPP:-128} --steps 1 --warmup 0 > $d.log 2>&1) || { tail -5 $d.log; exit 1; }
  python3 $R/tools/pmc_sum.py $d > $R/gpurun_out/pmc_$i.json
  echo "== $set"; cat $R/gpurun_out/pmc_$i.json
  rm -rf $d
done
