"""Debug: compare the MFMA kernel with the generic kernel on the bench workload and report where they differ."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
inp = importlib.import_module("ls1-mardyn_amd.inp")
capi = importlib.import_module("ls1-mardyn_amd.capi")
engine_mod = importlib.import_module("ls1-mardyn_amd.engine")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 171
L, _ids, r, v = __import__("importlib").import_module("ls1-mardyn_amd.synth").bcc_box(n)
eng = engine_mod.DeviceEngine(0)
eng.set_components(bench.lj_components(inp), bench.RC)
eng.set_domain([L, L, L])
N = len(r)
eng.upload(np.arange(1, N + 1, dtype=np.uint64), np.zeros(N, np.int32), r, v)
eng.rebin(); eng.halo()
eng.set_option("force_kernel", capi.FK_GENERIC); eng.forces(0); Fg = eng.download_forces()["F"]
eng.set_option("force_kernel", capi.FK_AUTO); eng.forces(0); Fm = eng.download_forces()["F"]
st = eng.download_state()
d = np.abs(Fg - Fm).max(1)
bad = np.nonzero(d > 1e-9)[0]
print("N", N, "bad", len(bad), "max diff", d.max())
g = eng.get_grid() if hasattr(eng, "get_grid") else None
print("grid", g)
if len(bad):
    rb = st["r"][bad]
    clen = L / int(L / bench.RC)
    cells = np.floor(rb / clen).astype(int)
    for ax in range(3):
        u, c = np.unique(cells[:, ax], return_counts=True)
        print("axis", ax, "cells with bad:", dict(zip(u.tolist()[:12], c.tolist()[:12])), "... n distinct", len(u))
    print("sample", bad[:10], d[bad[:10]], Fg[bad[:3]], Fm[bad[:3]])
