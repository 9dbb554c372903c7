"""Condenses the rocprofv3 outputs of tools/collect_profiles.sh into kernel_stats.csv + pmc_summary.json.

Corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are
reported in KB; on gfx950 FETCH_SIZE counts half of the bytes for 8 B/lane coalesced reads (calibrated here on
k_kick_then_kick_drift, whose traffic is known exactly: 72 B read, 48 B written per molecule)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

out = sys.argv[1]
bench_args = sys.argv[2:]
stats = glob.glob(os.path.join(out, "stats", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(out, "kernel_stats.csv"))
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        per[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
kern = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in per.items()}
bench = json.loads(open(os.path.join(out, "bench_under_profiler.json")).read().strip().splitlines()[-1])
n = bench["config"]["molecules_per_gpu"]
summary = {"source": "tools/collect_profiles.sh (rocprofv3 --pmc, one pass per counter group, --kernel-trace only), "
                     "bench.py " + " ".join(bench_args) + ", per-launch means",
           "units": {"FETCH_SIZE": "KB as reported", "WRITE_SIZE": "KB as reported", "SQ_*_CYCLES": "quad-cycles"},
           "molecules": n, "neighbour_lists": bench["config"].get("neighbour_lists") is not None, "kernels": kern}
# calibration kernel with exactly known traffic (72 B read, 48 B written per molecule): the fused kick pair, or — when
# the force pass does the integration itself — the plain kick+drift pass of the first step
kd = [k for k in kern if "k_kick_then_kick_drift" in k] or [k for k in kern if "k_kick_drift" in k]
if kd and "FETCH_SIZE" in kern[kd[0]]:
    k = kern[kd[0]]
    summary["calibration"] = {"kernel": kd[0], "known_read_bytes": 72.0 * n, "known_write_bytes": 48.0 * n,
                              "FETCH_SIZE_bytes_raw": k["FETCH_SIZE"] * 1024, "WRITE_SIZE_bytes_raw": k["WRITE_SIZE"] * 1024,
                              "fetch_correction": 72.0 * n / (k["FETCH_SIZE"] * 1024)}
fk = [k for k in kern if "k_force_" in k and "reduce" not in k]
# the dominant force kernel = the one with the largest total time in the kernel trace (list build / per-step kernels of
# the first step also match the name pattern)
tot = {}
for f in glob.glob(os.path.join(out, "kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        tot[r["Name"].split("(")[0].replace("void ", "")] = float(r["TotalDurationNs"])
fk.sort(key=lambda name: -tot.get(name, 0.0))
summary["force_kernels_by_time_ms"] = {name: tot.get(name, 0.0) / 1e6 for name in fk}
if fk:
    k = kern[fk[0]]
    alg = bench["roofline"]["algorithmic_bytes_per_launch"]
    fs = {"name": fk[0], "algorithmic_bytes_per_launch": alg}
    if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
        fs["fetch_bytes_corrected"] = 2.0 * k["FETCH_SIZE"] * 1024
        fs["write_bytes"] = k["WRITE_SIZE"] * 1024
        fs["traffic_bytes_per_launch"] = fs["fetch_bytes_corrected"] + fs["write_bytes"]
    if "TCC_HIT_sum" in k:
        fs["l2_hit_rate"] = k["TCC_HIT_sum"] / max(k["TCC_HIT_sum"] + k["TCC_MISS_sum"], 1.0)
    if "GRBM_GUI_ACTIVE" in k:
        cyc = k["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
        fs["kernel_cycles"] = cyc
        if "SQ_ACTIVE_INST_VALU" in k:
            fs["valu_busy_frac_per_simd"] = k["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cyc)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in k:
            fs["mfma_busy_frac"] = k["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)
        if "SQ_LDS_IDX_ACTIVE" in k:
            fs["lds_busy_frac_per_cu"] = k["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc)
            fs["lds_bank_conflict_frac"] = k["SQ_LDS_BANK_CONFLICT"] / max(k["SQ_LDS_IDX_ACTIVE"], 1.0)
    if "SQ_INSTS_VALU" in k:
        fs["valu_wave_insts_per_launch"] = k["SQ_INSTS_VALU"]
        fs["valu_lane_insts_per_molecule"] = k["SQ_INSTS_VALU"] * 64.0 / n
        f64 = sum(k.get(c, 0.0) for c in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64"))
        fs["fp64_arith_share_of_valu"] = f64 / k["SQ_INSTS_VALU"]
        # FP64 vector work actually issued (lane operations, FMA = 2): the compute-side ceiling of this kernel is the
        # 78.6 TFLOP/s FP64 vector rate, not HBM
        fs["fp64_flop_per_launch"] = 64.0 * (2.0 * k.get("SQ_INSTS_VALU_FMA_F64", 0.0) + k.get("SQ_INSTS_VALU_ADD_F64", 0.0) +
                                             k.get("SQ_INSTS_VALU_MUL_F64", 0.0))
    summary["force_kernel"] = fs
# The --stats average covers EVERY launch of the run (melt + warm-up + timed + profiling steps); bench.py's roofline.avg_launch_ms
# covers the TIMED window only.  From the per-dispatch kernel trace: the dominant force kernel's launches of exactly that window
# (one list force launch per step; launches [melt + warmup, melt + warmup + steps) of that kernel).
trace = glob.glob(os.path.join(out, "stats", "*", "*kernel_trace.csv"))
if trace and fk:
    rows = [r for r in csv.DictReader(open(trace[0])) if r["Kernel_Name"].split("(")[0].replace("void ", "") == fk[0]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    cfgb = bench.get("config", {}).get("untimed_steps_before_the_timed_window", {})
    first = int(cfgb.get("melt", 0)) + int(cfgb.get("warmup", bench.get("warmup", 0)))
    # (round 4: the melt phase is extended by a few steps that place the timed window in the list lifetime)
    first += int((bench.get("config", {}).get("timed_window") or {}).get("extra_untimed_steps_for_alignment", 0))
    win = rows[first:first + int(bench["steps"])]
    if win:
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in win]
        alld = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
        summary["force_kernel_timed_window"] = {
            "kernel": fk[0], "launches_in_trace": len(rows), "window": [first, first + len(win)],
            "mean_ms_timed_window": sum(d) / len(d), "min_ms": min(d), "max_ms": max(d),
            "mean_ms_all_launches": sum(alld) / len(alld),
            "bench_avg_launch_ms_hip_events_same_run": bench["roofline"]["avg_launch_ms"],
            "note": "the --stats CSV averages all launches incl. the melting lattice (shorter lists); the timed window is the one bench.py reports"}
        with open(os.path.join(out, "kernel_stats_timed_window.txt"), "w") as fh:
            fh.write(json.dumps(summary["force_kernel_timed_window"], indent=1) + "\n")
json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary.get("force_kernel", {}), indent=1))
print(json.dumps(summary.get("calibration", {}), indent=1))
