#!/bin/bash
# forward GDN kernel: scalar squares + spread LDS-DMA (in-tree build) against the same with deferred stores (_exp/lib_d.so)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gdn.py -x -q 2>&1 | tail -n 3
for rep in 1 2; do
timeout 300 python tools/gdn_bw.py 2>/dev/null | grep -h "fwd_us\|bwd_us" | tr '\n' ' '; echo " <- in-tree"
CRDR_HIP_LIB=$PWD/_exp/lib_d.so timeout 300 python tools/gdn_bw.py 2>/dev/null | grep -h "fwd_us\|bwd_us" | tr '\n' ' '; echo " <- _exp variant"
done
CRDR_HIP_LIB=$PWD/_exp/lib_d.so timeout 900 python -m pytest tests/test_gpu_gdn.py -x -q 2>&1 | tail -n 3
