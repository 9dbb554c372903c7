#!/usr/bin/env python3
"""Kernel timeline of the MULTI-RANK driver seam: the unmodified reference driver with the device container and
DomainDecompHip, one rank in RCCL loopback (LS1HIP_LOOPBACK=1: all 26 directions leave through RcclTransport and come back), under
rocprofv3 --kernel-trace.  Prints, for both LS1HIP_OVERLAP settings, the driver's speed line, and for the overlapped run the kernels
of a few reuse steps with the share of the halo phase (refresh + pack + RCCL + import) that lies INSIDE the inner force pass.
usage: python tools/seam_b_timeline.py [n_per_dim=100] [steps=40] [out_dir=gpurun_out/r4/seam_tl]"""
import csv
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_seam_a import HEAD, LJ1  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
out = os.path.abspath(sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "r4", "seam_tl"))
N = 2 * n ** 3
L = (N / 0.785302672) ** (1 / 3)
cfg = HEAD.format(dt=0.002, steps=steps, temp=0.95, L=repr(L), rc=2.5, components=LJ1,
                  phasespace='<generator name="CubicGridGenerator"><specification>density</specification>'
                             '<density>0.785302672</density><binaryMixture>false</binaryMixture></generator>')
binary = os.path.join(ROOT, "oracle", "_ref", "MarDyn_hipB")
os.makedirs(out, exist_ok=True)
for overlap in ("1", "0"):
    d = os.path.join(out, "overlap" + overlap)
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "config.xml"), "w").write(cfg)
    env = dict(os.environ, OMP_NUM_THREADS="16", LS1HIP_LOOPBACK="1", LS1HIP_TRANSPORT="rccl", LS1HIP_OVERLAP=overlap,
               LS1HIP_MIRROR_SYNC_FINAL="0", TMPDIR="/tmp")
    cmd = ["timeout", "-k", "10", "400", "rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", os.path.join(d, "prof"), "--",
           binary, "config.xml", "--steps", str(steps), "--final-checkpoint=0"]
    p = subprocess.run(cmd, cwd=d, env=env, capture_output=True, text=True)
    if p.returncode != 0:
        print(p.stdout[-2000:], p.stderr[-2000:])
        sys.exit(1)
    speed = re.search(r"Simulation speed:\s*([0-9.eE+-]+)", p.stdout)
    loop = re.search(r"Computation in main loop took:\s*([0-9.eE+-]+)", p.stdout)
    lists = re.search(r"neighbour lists (on|off): (\d+) builds for (\d+)", p.stdout)
    print(f"# N={N} steps={steps} 1 rank, RCCL loopback, LS1HIP_OVERLAP={overlap} (under rocprofv3): Simulation speed {float(speed.group(1)):.4g} "
          f"molecule-updates/s, main loop {loop.group(1) if loop else '?'} s; lists {lists.group(1)}, {lists.group(2)} builds / {lists.group(3)} evaluations")
    f = glob.glob(os.path.join(d, "prof") + "/**/*kernel_trace.csv", recursive=True)
    if not f:
        print("# no kernel trace written")
        continue
    rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
    force = [i for i, r in enumerate(rows) if "k_force_lj_verlet" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]]
    if overlap == "1":
        # inner passes = the longer launches of a reuse step; overlap of the halo-phase kernels with them
        halo_names = ("k_halo_refresh", "k_refresh_pack", "k_copy_segments", "rccl", "k_refresh_import", "k_halo_gen", "k_export", "k_import")
        inside = total = 0
        ivals = [(int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])) for i in force]
        for r in rows:
            nm = r["Kernel_Name"]
            if not any(h in nm for h in halo_names):
                continue
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            total += e - s
            for a, b in ivals:
                inside += max(0, min(e, b) - max(s, a))
        print(f"# halo-phase kernel time inside a force pass: {inside / 1e3:.0f} of {total / 1e3:.0f} us ({100.0 * inside / max(total, 1):.0f} %)")
        a = force[len(force) // 2]
        b = force[min(len(force) // 2 + 6, len(force) - 1)]
        t0 = int(rows[a]["Start_Timestamp"])
        prev_end = t0
        for r in rows[a:b]:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
            print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f}  q{r.get('Queue_Id', '?'):>3s}  {name}")
            prev_end = max(prev_end, e)
