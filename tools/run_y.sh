set -x
timeout 600 python -m pytest tests/test_gpu_wino.py -x -q -s -m gpu > gpurun_out/y_tests.log 2>&1; echo "rc=$?" >> gpurun_out/y_tests.log
grep -v "^$" gpurun_out/y_tests.log | tail -25
timeout 900 python tools/bench_wino.py --wgrad 2>&1 | grep -v amdgpu.ids
