#!/bin/bash
# usage: run_abl.sh VARIANT...  -- rocprof kernel stats of experiments/sep_time.py for each experiments/lib_abl_<VARIANT>.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  export ABW_LIB_PATH=experiments/lib_abl_$v.so
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_$v -- python3 experiments/sep_time.py > gpurun_out/abl_$v.log 2>&1 || exit 1
  f=$(ls -t gpurun_out/abl_$v/*/*_kernel_stats.csv | head -1)
  echo "== $v"; python3 tools/trim_stats.py $f /dev/null | grep -E "sep_wide|sep_accum"
done
