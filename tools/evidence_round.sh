#!/bin/bash
# Round-end evidence on the GPU box (run from the repo root): default bench line, rocprofv3 kernel stats of the same step,
# per-shape table, counter passes (tools/pmc_round.sh), full-resolution codec split.  Outputs under gpurun_out/evidence/.
export TMPDIR=/tmp
O=gpurun_out/evidence
mkdir -p $O
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2> $O/prof.err
cp $(find $O/prof -name '*kernel_stats.csv' | head -1) $O/bench_kernel_stats.csv
rm -rf $O/prof
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --shape-table $O/conv_shapes_in_step.txt > $O/bench_shapes.json 2> /dev/null
bash tools/pmc_round.sh
cp gpurun_out/hbm_families.json gpurun_out/pmc_shapes.txt $O/
timeout 600 python3 tools/fullres_codec.py > $O/fullres_codec.json 2> $O/fullres_codec.err
timeout 900 python3 tools/fullres_sweep.py > $O/fullres_sweep.json 2> $O/fullres_sweep.err
timeout 600 python3 bench.py --no-cpu-baseline --steps 100 --warmup 20 --no-secondary > $O/bench_100steps.json 2> /dev/null
timeout 300 python3 tools/gdn_bw.py > $O/gdn_bandwidth.json 2> /dev/null
timeout 300 python3 tools/bench_gauss_cond.py > $O/gauss_cond_bandwidth.txt 2> /dev/null
timeout 600 python3 tools/bench_wino.py > $O/wino_shapes.txt 2> /dev/null
timeout 600 python3 tools/bench_wino.py --k5s2 >> $O/wino_shapes.txt 2> /dev/null
timeout 600 python3 tools/bench_wino.py --k5 >> $O/wino_shapes.txt 2> /dev/null
ls -la $O
