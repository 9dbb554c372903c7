"""Long run of the multi-site path on the device: ethane-like 2CLJ liquid (fixture component) on a jittered lattice,
NVE then NVT (device thermostat): ids intact, no lost molecules, quaternions normalised, energy behaviour reported."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from golden_io import input_path
inp = importlib.import_module("ls1-mardyn_amd.inp")
engine_mod = importlib.import_module("ls1-mardyn_amd.engine")
ps0 = inp.read_inp(input_path("Ethan_equilibrated.inp"))
comps, rc = ps0.components, 32.1254
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rng = np.random.default_rng(5)
N = 2 * n ** 3
rho = 20 * 9826 / 571.607759 ** 3  # 20x the fixture's vapour density: liquid-like neighbour counts
L = (N / rho) ** (1 / 3)
a = L / n
g = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 3) * a
r = (np.concatenate([g + 0.25 * a, g + 0.75 * a]) + 0.05 * a * rng.uniform(-0.5, 0.5, (N, 3))) % L
q = rng.normal(size=(N, 4)); q /= np.linalg.norm(q, axis=1)[:, None]
T = ps0.temperature
m = comps.components[0].mass if hasattr(comps.components[0], "mass") else 0.03
v = rng.normal(0, np.sqrt(T / m), (N, 3)); v -= v.mean(0)
e = engine_mod.DeviceEngine(0)
e.set_components(comps, rc)
e.set_domain([L, L, L])
e.upload(np.arange(1, N + 1, dtype=np.uint64), np.zeros(N, np.int32), r, v, q, np.zeros((N, 3)))
e.rebin(); e.halo(); u0, _ = e.forces(0)
print("N", N, "L", L, "cells/dim", int(L / rc), "kernel family", e.get_option("last_force_kernel"), "u0/N", u0 / N)
dt = 0.5
t0 = time.time()
E = []
for blk in range(steps // 50):
    o = e.run(dt, 50)
    E.append(0.5 * (o["summv2"] + o["sumIw2"]) + o["upot"])
print("NVE: %d steps in %.2fs; E first/last %.6g %.6g; rel drift %.2e" % (steps, time.time() - t0, E[0], E[-1], (E[-1] - E[0]) / abs(E[0])))
e.set_thermostat(True, T)
for blk in range(4):
    o = e.run(dt, 50)
Tnow = (o["summv2"] + o["sumIw2"]) / (3 * o["n"] + o["rot_dof"])
st = e.download_state()
print("NVT: T target %.6g, measured %.6g; ids intact %s; |q|-1 max %.1e" % (
    T, Tnow, bool(np.array_equal(np.sort(st["ids"]), np.arange(1, N + 1, dtype=np.uint64))),
    np.abs(np.linalg.norm(st["q"], axis=1) - 1).max()))
