cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_graph.py -x -q -k "rate_index or graph_replay or two_iterations or test_stage3_step" 2>&1 | tail -4
for i in 1 2 3; do
  timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | cut -c1-130
  CRDR_HIP_LIB=$PWD/_exp/ds/libcrdr_hip.so timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | cut -c1-130
done
