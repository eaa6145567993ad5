#!/bin/bash
# Rebuild the shipped perf database on the GPU box (run from the repo root): the 3x3 / 5x5 launches are timed again (cold caches,
# best of 3) on top of the shipped entries, then one confirmation run; copy gpurun_out/tune_*.json to crdr_amd/hip/tune_gfx950.json.
set -x
export TMPDIR=/tmp
export CRDR_TUNE_ROUNDS=3 CRDR_TUNE_COLD=1
timeout 2400 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --bf16x3 --tune-db none --retune-k3 crdr_amd/hip/tune_gfx950.json --retune-k1 --save-tune-db gpurun_out/tune_r3_o.json --tune-log gpurun_out/tune_r3_o.log --shape-table gpurun_out/r3_a_shapes.txt > gpurun_out/bench_a.log 2> gpurun_out/bench_a.err
cut -c1-300 gpurun_out/bench_a.log; tail -2 gpurun_out/bench_a.err
timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-secondary --tune-db gpurun_out/tune_r3_o.json > gpurun_out/bench_a2.log 2>> gpurun_out/bench_a.err
cut -c1-300 gpurun_out/bench_a2.log
