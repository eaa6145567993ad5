cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_wino.py -x -q -k "persistent or f4x4_fwd" 2>&1 | tail -3
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r6_prof_fb -- python3 /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > /root/repo/gpurun_out/r6_prof_fb.log 2>/dev/null
cd /root/repo; cut -c1-150 gpurun_out/r6_prof_fb.log; grep "filter_batched\|pack_weights_batched\|wgrad_reduce_batched\|adam_dyn" $(find gpurun_out/r6_prof_fb -name '*kernel_stats.csv' | head -1) | cut -c1-200
