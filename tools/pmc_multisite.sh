#!/bin/bash
# PMC counters of the multi-site force kernel on one bench_multisite.py case.
#   usage (inside gpurun):  bash tools/pmc_multisite.sh TAG <bench_multisite.py args...>
# Writes gpurun_out/pmc_ms_TAG.json (per-launch means of the k_force_ms* / k_force_generic launches).
set -e -o pipefail
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_ms_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_ANY SQ_INSTS_BRANCH"; do
	name=$(echo $pass | cut -d' ' -f1)
	timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/$name" -- python3 tools/bench_multisite.py "$@" > /dev/null 2> "$OUT/$name.log" || echo "pass $name failed"
done
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, os, sys
out = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_force_ms" in n or "k_force_generic" in n or "k_force_sites" in n:
            per[n.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            per[n.split("(")[0]]["VGPR"].append(float(r.get("VGPR_Count", 0) or 0))
            per[n.split("(")[0]]["LDS"].append(float(r.get("LDS_Block_Size", 0) or 0))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in per.items()}
json.dump(res, open(out + ".json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf "$OUT"
