"""Long NVE run on the device (default 3000 steps, N = 2*60^3 = 432 000): energy conservation, momentum, no lost /
overflowed molecules, ids intact.  Usage: python tools/soak.py [n_per_dim] [steps]
Environment: LS1_SOAK_SKIN (default 0.2: the neighbour-list loop; 0: per-step kernels), LS1_SOAK_SHIFTED=1 (truncated and
shifted LJ), LS1_SOAK_PRECISION=1|2 (SPDP / SPSP list pass)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
inp = importlib.import_module("ls1-mardyn_amd.inp")
engine_mod = importlib.import_module("ls1-mardyn_amd.engine")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
L, _ids, r, v = __import__("importlib").import_module("ls1-mardyn_amd.synth").bcc_box(n)
N = len(r)
e = engine_mod.DeviceEngine(0)
shifted = int(os.environ.get("LS1_SOAK_SHIFTED", "0"))  # 1: truncated AND shifted LJ (U continuous at r_c)
comps = inp.ComponentSet([inp.make_component(lj=[(0., 0., 0., 1., 1., 1., bench.RC, shifted)])], np.zeros((0, 2)), 1e10)
e.set_components(comps, bench.RC)
skin = float(os.environ.get("LS1_SOAK_SKIN", "0.2"))  # neighbour-list loop (0: per-step kernels)
if skin > 0:
    e.set_verlet(skin)
prec = int(os.environ.get("LS1_SOAK_PRECISION", "0"))  # 1 SPDP, 2 SPSP
if prec:
    e.set_option("precision", prec)
e.set_domain([L, L, L])
e.upload(np.arange(1, N + 1, dtype=np.uint64), np.zeros(N, np.int32), r, v)
e.rebin(); e.halo(); u0, _ = e.forces(0)
ek0 = 0.5 * float((v * v).sum())
t0 = time.time()
E = []
for blk in range(steps // 100):
    out = e.run(bench.DT, 100)
    E.append(0.5 * out["summv2"] + out["upot"])
dt = time.time() - t0
st = e.download_state()
E = np.array(E)
print(f"N={N} steps={steps} wall={dt:.2f}s ({N*steps/dt:.3e} updates/s)  E0={ek0+u0:.6f}  E range [{E.min():.6f}, {E.max():.6f}] "
      f"drift/|E|={(E[-1]-E[0])/abs(E[0]):.3e} fluct/|E|={(E.max()-E.min())/abs(E[0]):.3e}  T*={out['summv2']/(3*N):.4f}")
print(f"list loop: skin {skin}, {e.get_option('verlet_builds')} builds for {e.get_option('verlet_steps')} list steps, precision in use {e.get_option('precision_in_use')}")
print("ids intact:", bool(np.array_equal(np.sort(st["ids"]), np.arange(1, N + 1, dtype=np.uint64))),
      " |sum p|/sqrt(N)=", float(np.abs(st["v"].sum(0)).max() / np.sqrt(N)))
