"""debug probe: multi-site list loop step by step (which step loses molecules?)"""
import importlib, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_io import input_path, manifest, sorted_phase_space
inp = importlib.import_module("ls1-mardyn_amd.inp"); engine_mod = importlib.import_module("ls1-mardyn_amd.engine")
MAN = manifest()
ps = inp.read_inp(input_path(MAN["ethan"]["input"])); st = sorted_phase_space(ps)
q = st["q"] / np.linalg.norm(st["q"], axis=1, keepdims=True)
nvt = len(sys.argv) > 1 and sys.argv[1] == "nvt"
scale = 1.0 if nvt else 3.0
e = engine_mod.DeviceEngine(0)
e.set_components(ps.components, MAN["ethan"]["rc"]); e.set_verlet(1.5); e.set_domain(ps.length)
e.upload(st["ids"], st["cid"], st["r"], st["v"] * scale, q, st["D"])
if nvt: e.set_thermostat(True, ps.temperature)
e.rebin(); e.halo(); e.forces(0)
L = ps.length
for s in range(12):
    try:
        out = e.run(0.5 if nvt else 2.0, 1)
    except Exception as ex:
        print("step", s, "FAILED:", ex); break
    stt = e.download_state()
    r = stt["r"]
    print("step", s, "builds", e.get_option("verlet_builds"), "evals", e.get_option("verlet_steps"), "upot", out["upot"],
          "rmin", r.min(0), "rmax", (r - L).max(0), "finite", np.isfinite(r).all(), "vmax", np.sqrt((stt["v"] ** 2).sum(1).max()))
