#!/bin/bash
# Re-measure the parity margins of the GPU tests against the oracle (run from the repo root on the GPU box):
# writes gpurun_out/parity_margins.json; review and copy to profiles/r5_parity_margins.json.
rm -f gpurun_out/parity_margins.json
CRDR_PARITY_REMEASURE=1 CRDR_PARITY_DUMP=gpurun_out/parity_margins.json python3 -m pytest tests/test_gpu_model.py tests/test_gpu_step.py tests/test_gpu_charm.py \
    tests/test_gpu_codec_parity.py tests/test_gpu_gdn.py tests/test_gpu_bf16x3.py -q -m gpu -x
