#!/bin/bash
# HBM traffic of the k-means step in both arithmetics at configs[1]'s shape: FETCH_SIZE and WRITE_SIZE, one pass each.
# usage: bash experiments/pmc_km.sh (GPU box)
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf $R/gpurun_out/pmc_km
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_km/$c -- python3 $R/experiments/km_arith.py > $R/gpurun_out/pmc_km_$c.log 2>&1
    echo "pass $c done"
done
cd $R && python3 tools/pmc_summary.py gpurun_out/pmc_km km_step > gpurun_out/pmc_km.txt
find gpurun_out/pmc_km -name "*.csv" -size +1M -delete
cat gpurun_out/pmc_km.txt
