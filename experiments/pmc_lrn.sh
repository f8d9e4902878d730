#!/bin/bash
# HBM traffic of the two one-launch .lrn writers: FETCH_SIZE and WRITE_SIZE, one pass each, over lrn_rows_ab.py (cluster
# writer: 100 clusters of 2 000 rows and 200 000 rows in one call, from the [D][N] table and from the resident copy) and
# lrn_emit_time.py (build writer, 200 000 x 189).   usage: bash experiments/pmc_lrn.sh (GPU box)
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf $R/gpurun_out/pmc_lrn
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_lrn/rows_$c -- python3 $R/experiments/lrn_rows_ab.py > $R/gpurun_out/pmc_lrn_rows_$c.log 2>&1
    timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_lrn/emit_$c -- python3 $R/experiments/lrn_emit_time.py > $R/gpurun_out/pmc_lrn_emit_$c.log 2>&1
    echo "pass $c done"
done
cd $R && python3 tools/pmc_by_grid.py gpurun_out/pmc_lrn lrn_emit > gpurun_out/pmc_lrn.txt
find gpurun_out/pmc_lrn -name "*.csv" -size +1M -delete
cat gpurun_out/pmc_lrn.txt
