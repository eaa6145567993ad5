cd /root/repo
export TMPDIR=/tmp
export CRDR_PARITY_DUMP=gpurun_out/r6_margins_partial.json
timeout 1500 python -m pytest tests/test_gpu_step.py -x -q -k "imposed or 256" 2>&1 | tail -30 > gpurun_out/r6_step_tests2.log
tail -n 5 gpurun_out/r6_step_tests2.log
