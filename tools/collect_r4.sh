#!/bin/bash
# Round-4 evidence, one gpurun call: rocprofv3 statistics + PMC passes of the default bench command, kernel statistics of the
# multi-site workloads, the bench lines of every configuration, the driver seam's speed line.  Outputs under gpurun_out/r4/final/.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
O=gpurun_out/r4/final
mkdir -p $O
export TMPDIR=/tmp
bash tools/collect_profiles.sh r4 > $O/collect_profiles.log 2>&1
cp gpurun_out/prof_r4/kernel_stats.csv $O/kernel_stats.csv
cp gpurun_out/prof_r4/pmc_summary.json $O/pmc_summary.json
cp gpurun_out/prof_r4/bench_under_profiler.json $O/bench_under_profiler.json
cp gpurun_out/prof_r4/kernel_stats_timed_window.txt $O/kernel_stats_timed_window.txt
echo "profiles done"
for wl in ethane mixed; do
  # (PMC passes of the multi-site workloads: VALU / LDS busy, FP64 share, traffic of the pair-stream force pass)
  bash tools/collect_profiles.sh r4_$wl --workload $wl > $O/collect_profiles_$wl.log 2>&1 && cp gpurun_out/prof_r4_$wl/pmc_summary.json $O/pmc_$wl.json
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$wl -- python3 bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline --no-live-pmc > $O/bench_${wl}_under_profiler.json 2> $O/stats_$wl.log
  f=$(ls $O/stats_$wl/*/*kernel_stats.csv | head -1); cp "$f" $O/kernel_stats_$wl.csv; rm -rf $O/stats_$wl
  echo "stats $wl done"
done
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench default done"
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>/dev/null; echo "bench driver cmd done"
python bench.py --n-per-dim 171 > $O/bench_1e7.json 2>/dev/null; echo "bench 1e7 done"
python bench.py --nvt > $O/bench_1e8_nvt.json 2>/dev/null; echo "bench nvt done"
python bench.py --workload ethane > $O/bench_ethane.json 2>/dev/null; echo "bench ethane done"
python bench.py --workload mixed > $O/bench_mixed.json 2>/dev/null; echo "bench mixed done"
(SEAM_B_TIMERS= python tools/seam_b_speed.py 171 100 lists-only; python tools/seam_b_speed.py 171 400 lists-only) > $O/seam_b_speed.txt 2>&1; echo "seam b done"
python - <<PY
import json
for f in ("bench","bench_driver_cmd","bench_1e7","bench_1e8_nvt","bench_ethane","bench_mixed"):
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
        c=d["roofline"].get("compute") or {}
        print(f, "%.4g"%d["value"], "%.3f ms"%d["ms_per_step"], "steady", d.get("steady_state_value"), "force %.3f"%d["roofline"]["avg_launch_ms"], "frac %.4f"%d["roofline"]["frac"], "traffic %.4g"%(d["roofline"]["traffic"] or 0), "valu %.3f lds %.3f fp64 %.3f"%(c.get("valu_busy_frac_per_simd",0),c.get("lds_busy_frac_per_cu",0),c.get("fp64_frac",0)), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e: print(f, "ERR", e)
PY
cat $O/seam_b_speed.txt | cut -c1-200
