#!/bin/bash
# same-box A/B: round-2 tree (_ab_r2, library v200) against the current tree, alternating, 60 timed steps each
for rep in 1 2; do
  (cd _ab_r2 && timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 60 --warmup 10 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('r2 ',d['value'],d['ms_per_step'],d['roofline']['achieved'])")
  timeout 600 python bench.py --no-cpu-baseline --no-secondary --steps 60 --warmup 10 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('r3 ',d['value'],d['ms_per_step'],d['roofline']['achieved'])"
done
