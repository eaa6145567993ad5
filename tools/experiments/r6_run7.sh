cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_graph.py -x -q -k "filter_caches" 2>&1 | tail -15 > gpurun_out/r6_graph_test2.log
tail -n 3 gpurun_out/r6_graph_test2.log
export CRDR_TUNE_ROUNDS=2 CRDR_TUNE_COLD=1
timeout 3300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --bf16x6 --save-tune-db gpurun_out/tune_r6_bf6.json --tune-log gpurun_out/tune_r6_bf6.log > gpurun_out/bench_tune_bf6.log 2> gpurun_out/bench_tune_bf6.err
cut -c1-600 gpurun_out/bench_tune_bf6.log; tail -n 3 gpurun_out/bench_tune_bf6.err
