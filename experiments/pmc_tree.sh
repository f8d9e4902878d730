#!/bin/bash
# PMC passes over abw_sep_tree at 1 M scaffolds (each pass a run of its own).  usage: bash experiments/pmc_tree.sh [pass ...]  (on the GPU box)
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
if [ $# -gt 0 ]; then PASSES=("$@"); else PASSES=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"); fi
rm -rf $R/gpurun_out/pmc_tree
for pass in "${PASSES[@]}"; do
    tag=$(echo $pass | cut -d' ' -f1)
    timeout -s KILL 240 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc_tree/$tag -- python3 $R/experiments/tree_time.py 1000000 20 200 > $R/gpurun_out/pmc_tree_$tag.log 2>&1
    echo "pass $tag done"
done
cd $R && python3 tools/pmc_summary.py gpurun_out/pmc_tree tree_ > gpurun_out/pmc_tree.txt
find gpurun_out/pmc_tree -name "*.csv" -size +1M -delete
