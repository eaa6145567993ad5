cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_step.py tests/test_gpu_graph.py -x -q -k "test_stage3_step or graph_replay or two_iterations or skip_gate" --durations=8 2>&1 | tail -25 > gpurun_out/r6_reuse_tests.log
tail -n 12 gpurun_out/r6_reuse_tests.log
timeout 600 python -m pytest tests/test_gpu_wino.py -x -q -k "f4x4" 2>&1 | tail -4 > gpurun_out/r6_wino_tests.log
tail -n 3 gpurun_out/r6_wino_tests.log
timeout 900 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r6_b_reuse.log 2> gpurun_out/r6_b_reuse.err
cut -c1-200 gpurun_out/r6_b_reuse.log; tail -n 2 gpurun_out/r6_b_reuse.err
bash tools/pmc_1x1.sh t256k5s2 256 64 256 5 2 1 > /dev/null 2>&1
bash tools/pmc_1x1.sh c192k5s2 192 128 192 5 2 0 > /dev/null 2>&1
{ python3 tools/pmc_summary.py t256k5s2 214.75 342.1; python3 tools/pmc_summary.py c192k5s2 120.80 255.3; } > gpurun_out/r6_pmc_order.txt 2>&1
cat gpurun_out/r6_pmc_order.txt
