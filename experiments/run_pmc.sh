#!/bin/bash
# HBM traffic counters, one pass per counter (rocprofv3 does not split them): FETCH_SIZE, WRITE_SIZE for the
# separator-only and signature-only scripts.  usage: bash experiments/run_pmc.sh   (on the GPU box, from the repo root)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for s in sep sig; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 180 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${s}_$c -- python3 experiments/${s}_time.py > gpurun_out/pmc_${s}_$c.log 2>&1 || exit 1
    echo "== $s $c"; python3 tools/pmc_summary.py gpurun_out/pmc_${s}_$c ${s}_ | head -12
    [ $s = sig ] && python3 tools/pmc_summary.py gpurun_out/pmc_${s}_$c seg_ | head -8
  done
done
