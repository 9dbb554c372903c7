#!/bin/bash
# multi-site A/B: tools/ab_ms.sh <workload> <tag|base> [<tag> ...]   (one line per run, the list twice)
wl=$1; shift
for rep in 1 2; do for t in "$@"; do
  lib=""; [ "$t" != "base" ] && lib="ls1-mardyn_amd/lib/variants/libls1hip_$t.so"
  LS1HIP_LIB=$lib python bench.py --workload $wl --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['device_ms_per_step']
print('%-10s %-8s %.4g upd/s  %.3f ms/step  force %.3f  build/ea %.2f  integrate %.3f' % ('$t', '$wl', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], m['list_build_ms_per_build'] or 0, m['integrate']))" || echo "$t FAILED"
done; done
