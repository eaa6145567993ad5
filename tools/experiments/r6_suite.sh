cd /root/repo
export TMPDIR=/tmp
rm -f gpurun_out/r6_parity_margins.json gpurun_out/r6_plan_replay.json
CRDR_PARITY_DUMP=gpurun_out/r6_parity_margins.json CRDR_PLAN_REPLAY_DUMP=gpurun_out/r6_plan_replay.json timeout 3000 python -m pytest tests -m gpu -q --durations=45 -p no:cacheprovider > gpurun_out/r6_suite.log 2>&1
tail -n 60 gpurun_out/r6_suite.log
timeout 600 python tools/experiments/filter_cache_inventory.py > gpurun_out/r6_filter_cache_inventory.log 2>&1
tail -n 5 gpurun_out/r6_filter_cache_inventory.log
