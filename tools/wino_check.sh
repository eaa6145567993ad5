#!/bin/bash
# Winograd kernels: their GPU tests + the single-shape comparison against the implicit-GEMM kernel (run from the repo root).
set -x
timeout 600 python -m pytest tests/test_gpu_wino.py -x -q -m gpu > gpurun_out/y_tests.log 2>&1; echo "rc=$?" >> gpurun_out/y_tests.log
tail -4 gpurun_out/y_tests.log
timeout 900 python tools/bench_wino.py 2>&1 | grep -v amdgpu.ids
timeout 900 python tools/bench_wino.py --wgrad 2>&1 | grep -v amdgpu.ids
