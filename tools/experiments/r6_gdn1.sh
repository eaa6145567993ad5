#!/bin/bash
# round 6: fused GDN backward -- tests, bandwidth, and the chain-handle tidy's step tests
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gdn.py -x -q > gpurun_out/r6_gdn_tests.log 2>&1
tail -n 15 gpurun_out/r6_gdn_tests.log
timeout 300 python tools/gdn_bw.py > gpurun_out/r6_gdn_bw_fused.json 2> gpurun_out/r6_gdn_bw_fused.err
CRDR_GDN_UNFUSED_BWD=1 timeout 300 python tools/gdn_bw.py > gpurun_out/r6_gdn_bw_unfused.json 2>> gpurun_out/r6_gdn_bw_fused.err
grep -h "bwd_us\|bwd_GB" gpurun_out/r6_gdn_bw_fused.json gpurun_out/r6_gdn_bw_unfused.json
timeout 1200 python -m pytest tests/test_gpu_step.py tests/test_gpu_graph.py -x -q -k "test_stage3_step or rate_index or two_iterations or graph" > gpurun_out/r6_chain_tidy.log 2>&1
tail -n 5 gpurun_out/r6_chain_tidy.log
