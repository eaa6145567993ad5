#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gdn.py -x -q > gpurun_out/r6_gdn_tests.log 2>&1
tail -n 5 gpurun_out/r6_gdn_tests.log
timeout 300 python tools/gdn_bw.py > gpurun_out/r6_gdn_bw_fused.json 2> gpurun_out/r6_gdn_bw_fused.err
grep -h "bwd_us\|bwd_GB" gpurun_out/r6_gdn_bw_fused.json
cd /tmp && export TMPDIR=/tmp
for hw in 128 32; do
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/gdnprof_$hw -o p -- python3 $GRAFT_REPO_ROOT/tools/experiments/r6_gdn_prof.py $hw > /dev/null 2>&1
done
