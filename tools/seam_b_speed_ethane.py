#!/usr/bin/env python3
"""The reference driver's own `Simulation speed` line (MarDyn.cpp:253-266) with the device container + integrator (seam B,
oracle/_ref/MarDyn_hipB) at BASELINE configs[3]: 2CLJ ethane, the reference's Ethan_equilibrated box replicated k^3 by the
reference's own io/ReplicaGenerator.cpp (homogeneous; the box is handed to it as the reference's binary checkpoint pair, written
by inp.write_checkpoint), NVT as every shipped config, no output plugins, no final checkpoint.
usage: python tools/seam_b_speed_ethane.py [k = 10] [steps = 100]"""
import gzip
import importlib
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_seam_a import HEAD, ETHANE  # noqa: E402  (the XML skeleton of the seam tests)
inp = importlib.import_module("ls1-mardyn_amd.inp")

k = int(sys.argv[1]) if len(sys.argv) > 1 else 10
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
binary = os.environ.get("LS1_SEAM_BINARY", os.path.join(ROOT, "oracle", "_ref", "MarDyn_hipB"))  # (the plain reference: config check)
with tempfile.TemporaryDirectory() as td:
    with gzip.open(os.path.join(ROOT, "tests", "golden", "inputs", "Ethan_equilibrated.inp.gz"), "rb") as fi, \
            open(os.path.join(td, "ethan.inp"), "wb") as fo:
        shutil.copyfileobj(fi, fo)
    ps = inp.read_inp(os.path.join(td, "ethan.inp"))
    ps.time = 0.0
    inp.write_checkpoint(os.path.join(td, "ethan"), ps)
    L = float(ps.length[0]) * k
    N = len(ps.ids) * k ** 3
    # thermostat target = the kinetic temperature of the fixture's own state, (sum m v^2 + sum I w^2) / (5 N) = 2.3755e-4 (bench.py's
    # last_step sums; the .inp header's 9.5e-4 would heat the box fivefold in the first step)
    cfg = HEAD.format(dt=0.5, steps=steps, temp=2.3755e-4, L=repr(L), rc=32.1254, components=ETHANE,
                      phasespace='<generator name="ReplicaGenerator"><type>homogeneous</type><files><vapor><header>ethan.header.xml</header>'
                                 f'<data>ethan.dat</data></vapor></files><numblocks><xz>{k}</xz><vapor>{k}</vapor></numblocks></generator>')
    open(os.path.join(td, "config.xml"), "w").write(cfg)
    env = dict(os.environ, OMP_NUM_THREADS="16")
    t0 = time.time()
    out = subprocess.run([binary, "config.xml", "--steps", str(steps), "--final-checkpoint=0"], cwd=td, env=env,
                         capture_output=True, text=True, timeout=1500)
    wall = time.time() - t0
if out.returncode != 0:
    print(out.stdout[-3000:], out.stderr[-2000:])
    sys.exit(1)
if os.environ.get("SEAM_B_TIMERS"):
    for ln in out.stdout.splitlines():
        if "LS1HIP_PROFILE" in ln or ((" took" in ln or "speed" in ln) and "took: 0 sec" not in ln):
            print("   ", ln.strip())
speed = re.search(r"Simulation speed:\s*([0-9.eE+-]+)", out.stdout)
lists = re.search(r"neighbour lists (on|off): (\d+) builds for (\d+)", out.stdout)
what = (f"device container + integrator, NVT; lists {lists.group(1)}, {lists.group(2)} builds / {lists.group(3)} evaluations" if lists
        else "the reference's own container (no device)")
print(f"ethane N={N} (ReplicaGenerator {k}^3) steps={steps}: Simulation speed {float(speed.group(1)):.4g} molecule-updates/s "
      f"(unmodified Simulation::simulate, {what}; whole process incl. generator and upload {wall:.1f} s)", flush=True)
