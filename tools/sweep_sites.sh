#!/bin/bash
# Sweep of the site kernel's launch geometry (LS1HIP_SITES_SHAPE = bx,by,bz,lanes per molecule,molecules per lane slot)
#   usage (inside gpurun): bash tools/sweep_sites.sh ethane|mixed "8,4,4,1,1 6,4,4,2,2 ..."
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export LS1HIP_DEBUG_SITES=1
if [ "$1" = ethane ]; then ARGS="Ethan_equilibrated.inp 32.1254 10 rep 0"; else ARGS="VectorizationMultiComponentMultiPotentials.inp 35 171 bcc 0"; fi
for shape in $2; do
	LS1HIP_SITES_SHAPE=$shape timeout -k 10 200 python3 tools/bench_multisite.py $ARGS 2>&1 | grep -E "force|site kernel" | sort -u | sed "s/^/[$shape] /"
done
