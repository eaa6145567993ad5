cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16x6.py -x -q 2>&1 | tail -5 > gpurun_out/r6_bf6_test5.log
S=dec128k3,enc96k3,dec64k3,dec256k1,up3T,enc5s2,hoist4256,charm224,D256s2
timeout 600 python tools/sweep_conv.py --shapes $S --top 3 --bf16x6 > gpurun_out/r6_sweep_bf6_v5.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_step.py -x -q -k "over_the_rate_index or two_iterations or 256 or test_stage3_step or test_stage1_step" 2>&1 | tail -40 > gpurun_out/r6_step_tests.log
timeout 600 python -m pytest tests/test_gpu_graph.py tests/test_gpu_wino.py -x -q -k "graph or persistent" 2>&1 | tail -8 > gpurun_out/r6_graph_tests.log
tail -3 gpurun_out/r6_bf6_test5.log gpurun_out/r6_step_tests.log gpurun_out/r6_graph_tests.log
