set -x
export CRDR_TUNE_ROUNDS=3 CRDR_TUNE_COLD=1
timeout 900 python -m pytest tests/test_gpu_charm.py tests/test_gpu_codec_parity.py -x -q -m gpu > gpurun_out/x_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/x_tests.log
timeout 1500 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --tune-db tools/data/tune_r3_e.json --save-tune-db gpurun_out/tune_r3_f.json --shape-table gpurun_out/r3_x_shapes.txt > gpurun_out/bench_x.log 2> gpurun_out/bench_x.err
timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-secondary --tune-db gpurun_out/tune_r3_f.json > gpurun_out/bench_x2.log 2>> gpurun_out/bench_x.err
tail -3 gpurun_out/x_tests.log; cat gpurun_out/bench_x.log gpurun_out/bench_x2.log | cut -c1-400
