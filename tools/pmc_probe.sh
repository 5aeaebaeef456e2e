#!/bin/bash
# development helper: extra SQ counters of the headline kernel (scalar unit, branches, instruction fetch, VALU mix)
# usage (GPU box): bash tools/pmc_probe.sh [bench args...]   -> gpurun_out/pmc_probe/*.json
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_probe
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
i=0
for set in "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SMEM SQ_INSTS_SENDMSG SQ_WAVE_CYCLES SQ_CYCLES" \
           "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/set$i -- python3 bench.py --no-cpu-baseline --spp 128 --steps 1 --warmup 0 "$@" > $O/set$i.log 2>&1
  python3 tools/pmc_sum.py $O/set$i > $O/set$i.json
  rm -rf $O/set$i
done
grep -h '^{' $O/set1.log | tail -1 > $O/bench.json
ls $O
