#!/bin/bash
# timing experiments on k_project_score (CR_GEO_EXP bitmask: 1 no chamfer, 2 no plane stores, 4 rcp division, 8 f32 chamfer, 16 no exp)
for e in 0 1 2 3 4 8 12 16 28 31; do
  CR_GEO_EXP=$e python bench.py --workload geometry --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('exp $e kernel_ms %.4f frac %.3f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"
done
