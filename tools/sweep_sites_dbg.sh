#!/bin/bash
# site kernel phase timing: LS1HIP_SITES_DBG = 3 (tables + region table), 4 (+ scans + staging), 2 (+ own sites, sort, stores),
# 1 (+ search), 0 (everything)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
if [ "$1" = ethane ]; then ARGS="Ethan_equilibrated.inp 32.1254 10 rep 0"; else ARGS="VectorizationMultiComponentMultiPotentials.inp 35 171 bcc 0"; fi
for d in 3 4 2 1 0; do
	LS1HIP_SITES_SHAPE=$2 LS1HIP_SITES_DBG=$d timeout -k 10 200 python3 tools/bench_multisite.py $ARGS 2>&1 | grep -E "force" | sed "s/^/[$2 dbg=$d] /" | cut -c1-120
done
