cd /root/repo
export TMPDIR=/tmp
bash tools/evidence_round.sh > gpurun_out/r6_evidence_b.log 2>&1
rm -f gpurun_out/r6_parity_margins_b.json gpurun_out/r6_plan_replay_b.json
CRDR_PARITY_DUMP=gpurun_out/r6_parity_margins_b.json CRDR_PLAN_REPLAY_DUMP=gpurun_out/r6_plan_replay_b.json timeout 3000 python -m pytest tests -m gpu -q --durations=12 -p no:cacheprovider > gpurun_out/r6_suite_b.log 2>&1
tail -n 22 gpurun_out/r6_suite_b.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
