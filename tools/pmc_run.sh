#!/bin/bash
# usage: tools/pmc_run.sh <tag> <pmc_one args...>   (runs two counter passes with rocprofv3, csv under gpurun_out/pmc_<tag>_{a,b})
tag=$1; shift
export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_a -- python3 tools/pmc_one.py "$@" > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_b -- python3 tools/pmc_one.py "$@" > /dev/null 2>&1
ls gpurun_out/pmc_${tag}_a/*/ gpurun_out/pmc_${tag}_b/*/ 2>/dev/null | head
