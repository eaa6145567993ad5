set -x
export CRDR_TUNE_ROUNDS=3 CRDR_TUNE_COLD=1
timeout 600 python -m pytest tests/test_gpu_wino.py -x -q -s -m gpu > gpurun_out/z_tests.log 2>&1; echo "rc=$?" >> gpurun_out/z_tests.log
tail -12 gpurun_out/z_tests.log
timeout 1500 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --tune-db none --retune-k3 tools/data/tune_r3_f.json --save-tune-db gpurun_out/tune_r3_g.json --tune-log gpurun_out/tune_r3_g.log --shape-table gpurun_out/r3_z_shapes.txt > gpurun_out/bench_z.log 2> gpurun_out/bench_z.err
timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-secondary --tune-db gpurun_out/tune_r3_g.json > gpurun_out/bench_z2.log 2>> gpurun_out/bench_z.err
cat gpurun_out/bench_z.log gpurun_out/bench_z2.log | cut -c1-300; tail -5 gpurun_out/bench_z.err
