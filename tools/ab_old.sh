#!/bin/bash
# same-box A/B: the tree before the Charm support-scatter change (_ab_old) against the current tree, alternating
for rep in 1 2 3; do
  (cd _ab_old && timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 60 --warmup 10 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('old',d['value'],d['ms_per_step'],d['roofline']['achieved'])")
  timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 60 --warmup 10 --tune-db tools/data/tune_r3_f.json 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('new',d['value'],d['ms_per_step'],d['roofline']['achieved'])"
done
