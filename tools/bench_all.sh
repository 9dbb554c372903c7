mkdir -p gpurun_out/r4
python bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench_drv_b.json 2> gpurun_out/r4/bench_drv_b.err
python bench.py > gpurun_out/r4/bench_default_b.json 2>/dev/null
python bench.py --n-per-dim 171 > gpurun_out/r4/bench_1e7_b.json 2>/dev/null
python bench.py --nvt --no-cpu-baseline > gpurun_out/r4/bench_nvt_b.json 2>/dev/null
python bench.py --workload ethane > gpurun_out/r4/bench_ethane_b.json 2>/dev/null
python - <<PY
import json
for f in ("drv_b","default_b","1e7_b","nvt_b","ethane_b"):
    try:
        d=json.loads(open("gpurun_out/r4/bench_%s.json"%f).read().strip().splitlines()[-1])
        print(f, "%.4g"%d["value"], "%.3f ms"%d["ms_per_step"], "steady", d.get("steady_state_value"), "force", "%.3f"%d["roofline"]["avg_launch_ms"], "frac %.4f"%d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], d["device_ms_per_step"]["list_build_ms_per_build"])
    except Exception as e: print(f, "ERR", e)
PY
