#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline object on the GPU box.
#   usage (inside gpurun):  bash tools/collect_profiles.sh TAG [bench.py args...]
# Writes gpurun_out/prof_TAG/{stats,pmc_*}/ and gpurun_out/prof_TAG/{kernel_stats.csv,pmc_summary.json,bench.json}.
# PMC passes are separate runs with --kernel-trace only (no sys/hip/hsa trace domains), as the pool requires.
set -e -o pipefail
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$REPO"
ARGS="--steps ${PROF_STEPS:-20} --warmup 3 --no-cpu-baseline --no-live-pmc $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py $ARGS > "$OUT/bench_under_profiler.json" 2> "$OUT/stats.log"
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
	name=$(echo $pass | cut -d' ' -f1)
	timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d "$OUT/pmc_$name" -- python3 bench.py $ARGS > /dev/null 2> "$OUT/pmc_$name.log"
	echo "pass $name done"
done
python3 tools/pmc_summarize.py "$OUT" $*
# keep the summaries, drop the raw per-dispatch outputs (gpurun copies back at most 64 MiB)
find "$OUT" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
