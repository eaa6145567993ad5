cd /root/repo
export TMPDIR=/tmp
DB=crdr_amd/hip/tune_gfx950.json
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --tune-db $DB --shape-table gpurun_out/r6_shapes_fp32.txt > gpurun_out/r6_b_fp32.log 2> gpurun_out/r6_b.err
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --tune-db $DB --precision bf16x6 --shape-table gpurun_out/r6_shapes_bf6.txt > gpurun_out/r6_b_bf6.log 2>> gpurun_out/r6_b.err
cut -c1-200 gpurun_out/r6_b_fp32.log gpurun_out/r6_b_bf6.log
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r6_prof_bf6 -- python3 /root/repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --tune-db /root/repo/$DB --precision bf16x6 --profile-steps 0 > /root/repo/gpurun_out/r6_prof_bf6.log 2>&1
cd /root/repo; ls gpurun_out/r6_prof_bf6/*/ | head
