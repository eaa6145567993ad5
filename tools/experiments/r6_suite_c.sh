#!/bin/bash
# full GPU suite + smoke + default bench after the GDN / squared-operand weight-gradient changes (library still v600)
cd /root/repo
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q --durations=45 -p no:cacheprovider > gpurun_out/r6_suite_c.log 2>&1
tail -n 8 gpurun_out/r6_suite_c.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r6_smoke_c.log 2>&1; tail -n 2 gpurun_out/r6_smoke_c.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r6_bench_c.json 2> gpurun_out/r6_bench_c.err; cut -c1-400 gpurun_out/r6_bench_c.json
