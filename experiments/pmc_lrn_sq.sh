#!/bin/bash
# Where the cluster writer's waves spend their cycles: one SQ pass over lrn_rows_ab.py.  usage: bash experiments/pmc_lrn_sq.sh (GPU box)
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf $R/gpurun_out/pmc_lrn_sq
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc_lrn_sq/a -- python3 $R/experiments/lrn_rows_ab.py > $R/gpurun_out/pmc_lrn_sq.log 2>&1
cd $R && python3 tools/pmc_by_grid.py gpurun_out/pmc_lrn_sq lrn_emit > gpurun_out/pmc_lrn_sq.txt
find gpurun_out/pmc_lrn_sq -name "*.csv" -size +1M -delete
cat gpurun_out/pmc_lrn_sq.txt
