#!/bin/bash
# The perf-database entries of `precision: bf16x6` (run from the repo root on the GPU box): the stage-3 and stage-1 steps in that mode with the
# tuner live (cold caches, best of 2; every candidate must agree with the mode's built-in plan, ops._autotune), on top of the shipped database
# -- whose exact-fp32 entries are keyed differently (the mode travels in the conv keys' flags and in a suffix of the weight-gradient keys) and are
# not touched.  The saved database = the shipped entries + the new ones: review, then copy it to crdr_amd/hip/tune_gfx950.json.
set -x
export TMPDIR=/tmp
export CRDR_TUNE_ROUNDS=2 CRDR_TUNE_COLD=1
timeout 3300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --bf16x6 --save-tune-db gpurun_out/tune_r6_bf6.json --tune-log gpurun_out/tune_r6_bf6.log > gpurun_out/bench_tune_bf6.log 2> gpurun_out/bench_tune_bf6.err
cut -c1-600 gpurun_out/bench_tune_bf6.log; tail -n 3 gpurun_out/bench_tune_bf6.err
timeout 1500 python bench.py --stage 1 --bs 8 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --tune-db gpurun_out/tune_r6_bf6.json --precision bf16x6 --save-tune-db gpurun_out/tune_r6_bf6.json > gpurun_out/bench_tune_bf6_s1.log 2>> gpurun_out/bench_tune_bf6.err
cut -c1-300 gpurun_out/bench_tune_bf6_s1.log
