cd /root/repo
export TMPDIR=/tmp
export CRDR_TUNE_ROUNDS=2 CRDR_TUNE_COLD=1
A=gpurun_out/tune_r6_a.json
timeout 2400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --bf16x6 --bf16x3 --tune-db tools/data/tune_r6_bf6_candidate.json --save-tune-db $A --tune-log gpurun_out/tune_r6_a.log > gpurun_out/bench_tune_a.log 2> gpurun_out/bench_tune_a.err
cut -c1-160 gpurun_out/bench_tune_a.log
timeout 1500 python bench.py --stage 1 --bs 8 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --tune-db $A --precision bf16x6 --save-tune-db gpurun_out/tune_r6_b.json --tune-log gpurun_out/tune_r6_b.log > gpurun_out/bench_tune_b.log 2>> gpurun_out/bench_tune_a.err
cut -c1-160 gpurun_out/bench_tune_b.log
timeout 900 python bench.py --stage 1 --bs 8 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --tune-db gpurun_out/tune_r6_b.json --save-tune-db gpurun_out/tune_r6_c.json --tune-log gpurun_out/tune_r6_c.log > gpurun_out/bench_tune_c.log 2>> gpurun_out/bench_tune_a.err
cut -c1-160 gpurun_out/bench_tune_c.log
wc -l gpurun_out/tune_r6_a.log gpurun_out/tune_r6_b.log gpurun_out/tune_r6_c.log
