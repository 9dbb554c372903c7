#!/bin/bash
# one bench run of a library variant, one summary line: tools/ab_one.sh <tag|base> [bench flags]
t=$1; shift
lib=""; [ "$t" != "base" ] && lib="ls1-mardyn_amd/lib/variants/libls1hip_$t.so"
LS1HIP_LIB=$lib python bench.py --no-cpu-baseline --no-live-pmc "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
m = d['device_ms_per_step']
print('%-16s %s  %.4g upd/s  %.3f ms/step  force %.3f  build/ea %s  builds/step %.4f  rebin %.3f' % ('$t', ' '.join('$*'.split()[:6]), d['value'], d['ms_per_step'], m['force'], m['list_build_ms_per_build'], m['list_builds_per_step'], m['rebin']))
" || echo "$t FAILED"
