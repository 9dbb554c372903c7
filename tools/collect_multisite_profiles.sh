#!/bin/bash
# rocprofv3 kernel-trace statistics of the multi-site force kernels at BASELINE configs[3] / [4] sizes
#   usage (inside gpurun):  bash tools/collect_multisite_profiles.sh TAG
# Writes gpurun_out/prof_TAG_ms/{ethane,mixed}_{brick,generic}.{txt,csv}
set -e -o pipefail
TAG=$1
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_${TAG}_ms
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
run() {  # name, args...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- python3 tools/bench_multisite.py "$@" > "$OUT/$name.txt" 2> "$OUT/$name.log"
  f=$(ls "$OUT/$name"/*/*kernel_stats.csv | head -1); cp "$f" "$OUT/$name.csv"; rm -rf "$OUT/$name"
  cat "$OUT/$name.txt"
}
run ethane_brick Ethan_equilibrated.inp 32.1254 10 rep 0
run ethane_generic Ethan_equilibrated.inp 32.1254 10 rep 1
run mixed_brick VectorizationMultiComponentMultiPotentials.inp 35 171 bcc 0
run mixed_generic VectorizationMultiComponentMultiPotentials.inp 35 171 bcc 1
