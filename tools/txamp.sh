#!/bin/bash
# Transform kernel times of the headline workload with ONE residual amplitude everywhere (which arithmetic path runs where):
#   bash tools/txamp.sh  -> prints launch_ms of the four transform launches for mixed | 60 | 250 | 1023
for a in mixed 60 250 1023; do
  SVTAV1_BENCH_TXAMP=$a python3 bench.py --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --no-lf 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$a', [round(e['launch_ms'],4) for e in d['roofline_all'] if 'txfm' in e['kernel']], d['ms_per_step'])"
done
