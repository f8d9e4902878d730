#!/bin/bash
# HBM traffic of the separator's root node at configs[1] (abw_sep_node without per-dimension output, as bench.py's separator
# leg and the drop-in adapter call it): FETCH_SIZE and WRITE_SIZE, one pass each.  usage: bash experiments/pmc_sep_root.sh (GPU box)
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
export NO_PER_DIM_ONLY=1
rm -rf $R/gpurun_out/pmc_sep_root
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_sep_root/$c -- python3 $R/experiments/sep_root.py > $R/gpurun_out/pmc_sep_root_$c.log 2>&1
    echo "pass $c done"
done
cd $R && python3 tools/pmc_summary.py gpurun_out/pmc_sep_root > gpurun_out/pmc_sep_root.txt
find gpurun_out/pmc_sep_root -name "*.csv" -size +1M -delete
