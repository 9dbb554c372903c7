#!/bin/bash
# Steady-state skin sweep of the default bench workload on one box: tools/skin_sweep.sh out.txt 0.16 0.18 0.2 ...
out=$1; shift
: > $out
for s in "$@"; do
  python bench.py --no-cpu-baseline --no-live-pmc --skin $s --steps 60 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
m = d['device_ms_per_step']
nl = d['config'].get('neighbour_lists', {})
print('skin %-5s %.4g upd/s  %.3f ms/step  force %.3f  build %.2f ms x %.4f/step  rebin %.3f' % ('$s', d['value'], d['ms_per_step'], m['force'], m.get('list_build_ms_per_build', 0), m.get('list_builds_per_step', 0), m['rebin']))
" >> $out || echo "skin $s FAILED" >> $out
  tail -1 $out
done
