#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for hw in 128 32; do
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/gdnprof_$hw -o p -- python3 $GRAFT_REPO_ROOT/tools/experiments/r6_gdn_prof.py $hw > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
for hw in 128 32; do
f=$(find gpurun_out/gdnprof_$hw -name "*kernel_stats.csv" | head -1)
echo "== $hw"; cut -d, -f1-4 "$f" | cut -c1-150 | head -14
done
