cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16x6.py -x -q 2>&1 | tail -15 > gpurun_out/r6_bf6_test3.log
S=dec128k3,enc96k3,dec64k3,dec256k1,enc192k1@128,up3T,enc5s2,hoist4256,charm224,charm480,nlam160,D256s2
timeout 600 python tools/sweep_conv.py --shapes $S --top 4 --bf16x6 > gpurun_out/r6_sweep_bf6_v4.log 2>&1
tail -5 gpurun_out/r6_bf6_test3.log
