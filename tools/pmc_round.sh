#!/bin/bash
# One round of counter evidence (run from the repo root on the GPU box): HBM traffic per kernel family of one eager stage-3
# step (two --pmc passes) and MFMA / wait / traffic counters of three conv shapes with the tuned algorithm.
export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_step_f -- python3 tools/pmc_step.py > gpurun_out/pmc_step_f.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_step_w -- python3 tools/pmc_step.py > gpurun_out/pmc_step_w.log 2>&1
python3 tools/pmc_families.py gpurun_out/pmc_step_f gpurun_out/pmc_step_w gpurun_out/hbm_families.json > /dev/null
# round 4: the three largest DIRECT implicit-GEMM shapes of the step (hoisted Charm conv, the decoder's 5x5 stride-2 transposed conv, the
# encoder's 5x5 stride-2 conv) and the decoder's 3x3 bottleneck layer (tuned: the F(4x4, 3x3) Winograd kernel)
bash tools/pmc_1x1.sh h320k5 320 16 4256 5 1 0
bash tools/pmc_1x1.sh t256k5s2 256 64 256 5 2 1
bash tools/pmc_1x1.sh c192k5s2 192 128 192 5 2 0
bash tools/pmc_1x1.sh d128k3 128 128 128 3 1 0
{
python3 tools/pmc_summary.py h320k5 278.92 211.1
python3 tools/pmc_summary.py t256k5s2 214.75 342.1
python3 tools/pmc_summary.py c192k5s2 120.80 255.3
python3 tools/pmc_summary.py d128k3 77.31 269.0
} > gpurun_out/pmc_shapes.txt 2>&1
