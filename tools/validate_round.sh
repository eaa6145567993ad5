#!/bin/bash
# Round-end validation on the GPU box (run from the repo root): the whole GPU test suite, the smoke entry, then
# tools/evidence_round.sh (bench lines, rocprofv3 stats, per-shape table, PMC passes, codec sweep) -> gpurun_out/evidence/.
set -x
export TMPDIR=/tmp
rm -f gpurun_out/parity_margins.json gpurun_out/plan_replay.json
CRDR_PARITY_REMEASURE=${CRDR_PARITY_REMEASURE:-0} CRDR_PARITY_DUMP=gpurun_out/parity_margins.json CRDR_PLAN_REPLAY_DUMP=gpurun_out/plan_replay.json timeout 2400 python -m pytest tests -x -q -m gpu --durations=40 > gpurun_out/b_tests.log 2>&1; echo "rc=$?" >> gpurun_out/b_tests.log; tail -5 gpurun_out/b_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/b_smoke.log 2>&1; tail -2 gpurun_out/b_smoke.log
bash tools/evidence_round.sh > gpurun_out/b_evidence.log 2>&1; tail -3 gpurun_out/b_evidence.log
cut -c1-300 gpurun_out/evidence/bench_default.json
