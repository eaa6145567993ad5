#!/bin/bash
# One round of counter evidence (run from the repo root on the GPU box): HBM traffic per kernel family of one eager stage-3
# step (two --pmc passes) and MFMA / wait / traffic counters of three conv shapes with the tuned algorithm.
export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_step_f -- python3 tools/pmc_step.py > gpurun_out/pmc_step_f.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_step_w -- python3 tools/pmc_step.py > gpurun_out/pmc_step_w.log 2>&1
python3 tools/pmc_families.py gpurun_out/pmc_step_f gpurun_out/pmc_step_w gpurun_out/hbm_families.json > /dev/null
bash tools/pmc_1x1.sh e192k1 192 128 96 1 1 0
bash tools/pmc_1x1.sh d256k1 256 128 128 1 1 0
bash tools/pmc_1x1.sh e96k3 96 128 96 3 1 0
{
python3 tools/pmc_summary.py e192k1 9.664 301.99
python3 tools/pmc_summary.py d256k1 17.18 402.65
python3 tools/pmc_summary.py e96k3 43.49 201.33
} > gpurun_out/pmc_shapes.txt 2>&1
