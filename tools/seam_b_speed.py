#!/usr/bin/env python3
"""The reference driver's own `Simulation speed` line (MarDyn.cpp:253-266) with the device container + integrator (seam B,
oracle/_ref/MarDyn_hipB) at BASELINE configs[1]: 1CLJ, N = 2*171^3 = 10 000 422 from the reference's CubicGridGenerator, NVT as
every shipped config, 100 steps, no output plugins, no final checkpoint.   usage: python tools/seam_b_speed.py [n_per_dim] [steps] [lists-only]"""
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_seam_a import HEAD, LJ1  # noqa: E402  (the XML skeleton of the seam tests)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 171
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
N = 2 * n ** 3
L = (N / 0.785302672) ** (1 / 3)
cfg = HEAD.format(dt=0.002, steps=steps, temp=0.95, L=repr(L), rc=2.5, components=LJ1,
                  phasespace='<generator name="CubicGridGenerator"><specification>density</specification>'
                             '<density>0.785302672</density><binaryMixture>false</binaryMixture></generator>')
binary = os.path.join(ROOT, "oracle", "_ref", "MarDyn_hipB")
skins = ("default",) if len(sys.argv) > 3 and sys.argv[3] == "lists-only" else ("default", "0")
for skin in skins:
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "config.xml"), "w").write(cfg)
        env = dict(os.environ, OMP_NUM_THREADS="16")
        if skin != "default":
            env["LS1HIP_SKIN"] = skin
        t0 = time.time()
        out = subprocess.run([binary, "config.xml", "--steps", str(steps), "--final-checkpoint=0"], cwd=td, env=env,
                             capture_output=True, text=True, timeout=1500)
        wall = time.time() - t0
    if out.returncode != 0:
        print(out.stdout[-2000:], out.stderr[-2000:])
        sys.exit(1)
    if os.environ.get("SEAM_B_TIMERS"):  # the driver's own timer table
        for ln in out.stdout.splitlines():
            if "LS1HIP_PROFILE" in ln or ((" took" in ln or "speed" in ln) and "took: 0 sec" not in ln):
                print("   ", ln.strip())
    speed = re.search(r"Simulation speed:\s*([0-9.eE+-]+)", out.stdout)
    lists = re.search(r"neighbour lists (on|off): (\d+) builds for (\d+)", out.stdout)
    comp = re.search(r"Computation took:\s*([0-9.eE+-]+)", out.stdout)
    print(f"N={N} steps={steps} LS1HIP_SKIN={skin}: Simulation speed {float(speed.group(1)):.4g} molecule-updates/s "
          f"(unmodified Simulation::simulate, device container + integrator, NVT; lists {lists.group(1)}, {lists.group(2)} builds / "
          f"{lists.group(3)} evaluations; whole process incl. generator and upload {wall:.1f} s"
          + (f"; Computation took {comp.group(1)} s" if comp else "") + ")", flush=True)
