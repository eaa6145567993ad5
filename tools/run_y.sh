set -x
timeout 600 python -m pytest tests/test_gpu_wino.py -x -q -s -m gpu > gpurun_out/y_tests.log 2>&1; echo "rc=$?" >> gpurun_out/y_tests.log
tail -30 gpurun_out/y_tests.log
timeout 600 python tools/bench_wino.py > gpurun_out/y_bench.log 2>&1; cat gpurun_out/y_bench.log
