#!/bin/bash
# Rebuild the shipped perf database on the GPU box (run from the repo root): the 3x3 / 5x5 launches are timed again (cold caches,
# best of 3) on top of the shipped entries -- every candidate also has to agree with the built-in plan's result (ops._autotune) --
# then one confirmation run and the acceptance run: the full-size oracle steps and the plan replay on the candidate database.
# Since round 6 the acceptance run is DETERMINISTIC: the full-size steps hand the product's ReLU masks to the oracle
# (tests/test_gpu_step.py, UPSTREAM_IMPOSED_TOL), so the upstream gradients no longer depend on which masks a plan set happens to flip --
# rounds 4-5 had to pick the database whose draw passed an 8e-3 cap (two re-tunes measured 9.4e-3 / 9.8e-3 un-imposed; the second one is kept as
# tools/data/tune_r5_rejected_c.json and passes the deterministic gate at 4e-6).  A candidate that fails now has a kernel bug, not bad luck.
# Only a candidate with gpurun_out/tune_accept.log ending in `rc=0` is copied to crdr_amd/hip/tune_gfx950.json.
# A second database without the F(4x4, 3x3) / F(3x3, 4x4) kernels (CRDR_WINO4=0: F(2x2) + direct) for bench.py's `stage3_no_f4x4` line:
# tools/data/tune_r5_no_f4x4.json.  The bf16x6 entries of the database (precision: bf16x6) come from tools/tune_bf16x6.sh.
set -x
export TMPDIR=/tmp
export CRDR_TUNE_ROUNDS=3 CRDR_TUNE_COLD=1
timeout 2400 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --bf16x3 --tune-db none --retune-k3 crdr_amd/hip/tune_gfx950.json --save-tune-db gpurun_out/tune_r5.json --tune-log gpurun_out/tune_r5.log --shape-table gpurun_out/r5_tune_shapes.txt > gpurun_out/bench_tune.log 2> gpurun_out/bench_tune.err
cut -c1-300 gpurun_out/bench_tune.log; tail -2 gpurun_out/bench_tune.err
timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-secondary --tune-db gpurun_out/tune_r5.json > gpurun_out/bench_tuned.log 2>> gpurun_out/bench_tune.err
cut -c1-400 gpurun_out/bench_tuned.log
cp crdr_amd/hip/tune_gfx950.json /tmp/tune_shipped.json && cp gpurun_out/tune_r5.json crdr_amd/hip/tune_gfx950.json
timeout 1500 python -m pytest -x -q -m gpu tests/test_gpu_step.py -k "256_tuned or every_tuned" tests/test_gpu_tuned_plans.py > gpurun_out/tune_accept.log 2>&1; echo "rc=$?" >> gpurun_out/tune_accept.log
cp /tmp/tune_shipped.json crdr_amd/hip/tune_gfx950.json; tail -4 gpurun_out/tune_accept.log
[ -n "$SKIP_NO_F4X4" ] || CRDR_WINO4=0 timeout 1800 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --tune-db none --retune-k3 crdr_amd/hip/tune_gfx950.json --save-tune-db gpurun_out/tune_r5_no_f4x4.json > gpurun_out/bench_tune_no_f4x4.log 2>> gpurun_out/bench_tune.err
[ -n "$SKIP_NO_F4X4" ] || cut -c1-300 gpurun_out/bench_tune_no_f4x4.log
