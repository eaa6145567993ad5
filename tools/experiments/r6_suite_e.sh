#!/bin/bash
# re-measure the parity margins with the oracle's thread count fixed at 16 (tests/conftest.py), then the full GPU suite against them
cd /root/repo
export TMPDIR=/tmp
rm -f gpurun_out/r6_parity_margins.json
CRDR_PARITY_REMEASURE=1 CRDR_PARITY_DUMP=gpurun_out/r6_parity_margins.json timeout 3000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r6_suite_e_measure.log 2>&1
tail -n 3 gpurun_out/r6_suite_e_measure.log
cp gpurun_out/r6_parity_margins.json profiles/r6_parity_margins.json
timeout 3000 python -m pytest tests -m gpu -x -q --durations=45 -p no:cacheprovider > gpurun_out/r6_suite_e.log 2>&1
tail -n 4 gpurun_out/r6_suite_e.log
