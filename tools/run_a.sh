set -x
export TMPDIR=/tmp
bash tools/parity_margins.sh > gpurun_out/a_margins.log 2>&1; tail -5 gpurun_out/a_margins.log
export CRDR_TUNE_ROUNDS=3 CRDR_TUNE_COLD=1
timeout 1500 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --bf16x3 --tune-db none --retune-k3 tools/data/tune_r3_f.json --save-tune-db gpurun_out/tune_r3_h.json --tune-log gpurun_out/tune_r3_h.log > gpurun_out/bench_a.log 2> gpurun_out/bench_a.err
cut -c1-400 gpurun_out/bench_a.log
