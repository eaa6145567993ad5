for m in 0 auto 1; do
  if [ "$m" = auto ]; then unset CRDR_K_CMAJOR; else export CRDR_K_CMAJOR=$m; fi
  timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 40 --warmup 10 --shape-table gpurun_out/r3_h_shapes_$m.txt > gpurun_out/r3_h_bench_$m.json 2>/dev/null
  python -c "
import json;d=json.loads(open('gpurun_out/r3_h_bench_$m.json').read().strip().splitlines()[-1]);print('$m',d['value'],d['ms_per_step'],d['roofline']['achieved'])"
done
