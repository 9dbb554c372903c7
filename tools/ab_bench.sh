#!/bin/bash
# Same-box A/B of library variants (tools/ab_variant.sh): tools/ab_bench.sh <out.txt> <tag|base> [<tag> ...] [-- bench flags]
# Every variant runs the default bench workload (no CPU baseline, no PMC child passes), the whole list twice (A B A B) so that
# drift of the box shows.  One line per run: tag, particle-updates/s, ms/step, force-pass ms, rebin ms/step.
out=$1; shift
tags=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do tags+=("$1"); shift; done
[ "$1" == "--" ] && shift
flags="${@:---steps 60 --warmup 10}"
: > $out
for rep in 1 2; do
  for t in "${tags[@]}"; do
    lib=""; [ "$t" != "base" ] && lib="ls1-mardyn_amd/lib/variants/libls1hip_$t.so"
    LS1HIP_LIB=$lib python bench.py --no-cpu-baseline --no-live-pmc $flags 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
m = d['device_ms_per_step']
print('%-12s %.4g upd/s  %.3f ms/step  force %.3f  rebin %.3f  integrate %.3f  builds %s' % ('$t', d['value'], d['ms_per_step'], m['force'], m['rebin'], m['integrate'], d['config'].get('neighbour_lists', {}).get('list_builds')))
" >> $out || echo "$t FAILED" >> $out
    tail -1 $out
  done
done
