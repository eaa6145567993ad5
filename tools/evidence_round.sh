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
# round 6: counter passes of the bf16x6 forms (precision: bf16x6) of the decoder's 256 -> 128 1x1 layer @128^2 (tuner's choice among the mode's
# tiled / streaming forms) and of the encoder's 96 -> 96 3x3 layer @128^2 on the mode's 256 x 128 tile (forced: the tuner would keep the exact-fp32
# F(4x4) kernel there), and of the decoder's 128 -> 128 3x3 layer likewise
export CRDR_PRECISION=bf16x6
bash tools/pmc_1x1.sh b6_d256k1 256 128 128 1 1 0
bash tools/pmc_1x1.sh b6_e96k3 96 128 96 3 1 0 16 22
bash tools/pmc_1x1.sh b6_d128k3 128 128 128 3 1 0 16 22
unset CRDR_PRECISION
{
python3 tools/pmc_summary.py b6_d256k1 17.18 403.0
python3 tools/pmc_summary.py b6_e96k3 43.49 202.0
python3 tools/pmc_summary.py b6_d128k3 77.31 269.0
} > $O/pmc_bf16x6_shapes.txt 2>&1
timeout 600 python3 tools/aten_sources.py 2 > $O/aten_sources.txt 2>&1
ls -la $O
