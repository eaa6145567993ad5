#!/bin/bash
# full GPU suite with the oracle's thread count capped (tests/conftest.py) + the default bench line (cpu_baseline scans 16 / 32 / 64 threads)
cd /root/repo
export TMPDIR=/tmp
rm -f gpurun_out/r6_parity_margins.json gpurun_out/r6_plan_replay.json
CRDR_PARITY_DUMP=gpurun_out/r6_parity_margins.json CRDR_PLAN_REPLAY_DUMP=gpurun_out/r6_plan_replay.json timeout 3000 python -m pytest tests -m gpu -x -q --durations=45 -p no:cacheprovider > gpurun_out/r6_suite_d.log 2>&1
tail -n 4 gpurun_out/r6_suite_d.log
timeout 900 python bench.py > gpurun_out/r6_bench_d.json 2> gpurun_out/r6_bench_d.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_bench_d.json"))
print(d["value"], d["ms_per_step"], d["cpu_baseline"])
PY
