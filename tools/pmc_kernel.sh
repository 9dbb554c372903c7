#!/bin/bash
# Counter passes for chosen kernels of the bench loop (comma-separated name substrings), summed per counter and divided by the number of launches.
#   usage (inside gpurun):  bash tools/pmc_kernel.sh KERNEL_SUBSTRING OUT.json [bench.py args...]
set -e -o pipefail
K=$1; OUT=$2; shift; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$REPO"
ARGS="--steps 12 --warmup 3 --no-cpu-baseline --no-live-pmc $*"
D=$REPO/gpurun_out/pmck_tmp
rm -rf "$D"; mkdir -p "$D"
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN"; do
	name=$(echo $pass | cut -d' ' -f1)
	timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$D/$name" -- python3 bench.py $ARGS > /dev/null 2> "$D/$name.log" || echo "pass $name failed"
done
python3 - "$K" "$OUT" "$D" <<'PY'
import csv, glob, json, sys
ks, out, d = sys.argv[1:4]
res = {}
rows = [row for f in glob.glob(d + "/*/*/*counter_collection.csv") for row in csv.DictReader(open(f))]
for k in ks.split(","):  # several kernels: comma-separated substrings
    acc, n = {}, {}
    for row in rows:
        if k not in row["Kernel_Name"]:
            continue
        c = row["Counter_Name"]
        acc[c] = acc.get(c, 0.0) + float(row["Counter_Value"])
        n[c] = n.get(c, 0) + 1
    res[k] = {c: acc[c] / n[c] for c in acc}
    res[k]["_launches_seen"] = max(n.values()) if n else 0
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(res, sort_keys=True))
PY
rm -rf "$D"
