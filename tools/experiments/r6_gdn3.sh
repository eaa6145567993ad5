#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/sweep_conv.py --wgrad --top 8 --shapes gdn192,gdn192@32 > gpurun_out/r6_gdn_wgrad_sweep.txt 2>&1
cat gpurun_out/r6_gdn_wgrad_sweep.txt | cut -c1-600
cd /tmp && export TMPDIR=/tmp
export CRDR_GDN_UNFUSED_BWD=1
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/gdnprof_u128 -o p -- python3 $GRAFT_REPO_ROOT/tools/experiments/r6_gdn_prof.py 128 > /dev/null 2>&1
