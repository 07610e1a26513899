#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: kernel stats, then FETCH_SIZE / WRITE_SIZE / MFMA counters
# each in a pass of its own (MI355X_MICROARCH.md, HBM section).  Run on the GPU box from the repo root:
#   bash tools/profile_round.sh r01 40000 [battgp|matern32]       -> gpurun_out/<tag>/...
# then here:  python tools/pmc_summary.py gpurun_out/<tag> <tag> 40000
TAG=${1:-r01}; N=${2:-40000}; KERNEL=${3:-battgp}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
W="python $REPO/tools/profile_workload.py $N $KERNEL"
timeout 300 rocprofv3 --list-avail > $OUT/avail.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o st -- $W > $OUT/stats.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pf -- $W > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pw -- $W > $OUT/pmc_write.log 2>&1
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o pm -- $W > $OUT/pmc_mfma.log 2>&1
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_mops -o po -- $W > $OUT/pmc_mops.log 2>&1
# large CSVs stay on the box: keep per-kernel aggregates only
python $REPO/tools/pmc_aggregate.py $OUT
ls -la $OUT $OUT/*
