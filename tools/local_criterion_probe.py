"""Long comparison of the neighbour-list loop with the LOCAL rebuild criterion against the per-step search kernels: hot liquid,
many list lifetimes.  A listed pair released too early would show as a trajectory difference far above the 1e-9 printed here.
Usage: python tools/local_criterion_probe.py [n_per_dim] [steps] [T*]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
inp = importlib.import_module("ls1-mardyn_amd.inp")
engine_mod = importlib.import_module("ls1-mardyn_amd.engine")
synth = importlib.import_module("ls1-mardyn_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
temp = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
L, ids, r, v = synth.bcc_box(n, temp=temp)
comps = inp.ComponentSet([inp.make_component(lj=[(0., 0., 0., 1., 1., 1., bench.RC, 0)])], np.zeros((0, 2)), 1e10)
res = {}
for mode, skin, local in (("per-step", None, 1), ("list, local criterion", 0.2, 1), ("list, global criterion", 0.2, 0)):
    e = engine_mod.DeviceEngine(0)
    e.set_components(comps, bench.RC)
    e.set_option("local_rebuild", local)
    e.set_verlet(skin, force=True)
    e.set_domain([L] * 3)
    e.upload(ids, np.zeros(len(ids), np.int32), r, v)
    e.rebin(); e.halo(); e.forces(0)
    out = e.run(bench.DT, steps)
    st = e.download_state()
    o = np.argsort(st["ids"])
    res[mode] = (st["r"][o], st["v"][o], out, e.get_option("verlet_builds") if skin else 0)
    e.close()
a = res["per-step"]
for mode in list(res)[1:]:
    b = res[mode]
    dr = a[0] - b[0]
    dr -= L * np.round(dr / L)
    print(f"{mode}: N={len(ids)} steps={steps} T*={temp} builds={b[3]}  max|dr|={np.abs(dr).max():.3e}  max|dv|={np.abs(a[1]-b[1]).max():.3e}  "
          f"dU/U={(b[2]['upot']-a[2]['upot'])/abs(a[2]['upot']):.3e}")
