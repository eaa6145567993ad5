#!/bin/bash
# PMC passes for one conv shape with the autotuned algorithm: tools/pmc_1x1.sh <tag> CIN H COUT K STRIDE TRANSPOSED [BS]
tag=$1; shift
export TMPDIR=/tmp
A="$@"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_a -- python3 tools/pmc_one.py $A 16 -1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_f -- python3 tools/pmc_one.py $A 16 -1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_w -- python3 tools/pmc_one.py $A 16 -1 > /dev/null 2>&1
