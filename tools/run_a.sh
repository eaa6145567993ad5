set -x
export TMPDIR=/tmp
bash tools/parity_margins.sh > gpurun_out/a_margins.log 2>&1; tail -4 gpurun_out/a_margins.log
export CRDR_TUNE_ROUNDS=3 CRDR_TUNE_COLD=1
timeout 2400 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --bf16x3 --tune-db none --retune-k3 crdr_amd/hip/tune_gfx950.json --save-tune-db gpurun_out/tune_r3_j.json --tune-log gpurun_out/tune_r3_j.log --shape-table gpurun_out/r3_a_shapes.txt > gpurun_out/bench_a.log 2> gpurun_out/bench_a.err
cut -c1-300 gpurun_out/bench_a.log; tail -2 gpurun_out/bench_a.err
