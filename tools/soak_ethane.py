"""Long NVE run of the 2CLJ ethane box (the reference's Ethan_equilibrated.inp replicated k^3) through the pair-stream list loop with
the rigid-body integration FUSED into the list pass, and once more with the separate integrator passes: total energy (translation +
rotation + potential) over the run, rebuild count, and the two final states compared bit for bit.
usage: python tools/soak_ethane.py [k = 4] [steps = 2000]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
inp = importlib.import_module("ls1-mardyn_amd.inp")
engine_mod = importlib.import_module("ls1-mardyn_amd.engine")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
ps0, _ = bench.ethane_fixture(inp)
big = bench.replicate_phase_space(inp, ps0, k)
N = len(big.ids)
final = {}
for fuse in (1, 0):
    e = engine_mod.DeviceEngine(0)
    e.set_components(big.components, bench.ETHANE_RC)
    e.set_verlet(6.0)
    e.set_domain(big.length)
    e.set_option("fuse_integration", fuse)
    e.upload(big.ids, big.cid, big.r, big.v, big.q, big.D)
    e.rebin(); e.halo(); e.forces(0)
    t0 = time.time()
    E = []
    for blk in range(steps // 100):
        out = e.run(bench.ETHANE_DT, 100)
        E.append(0.5 * (out["summv2"] + out["sumIw2"]) + out["upot"])
    wall = time.time() - t0
    E = np.array(E)
    st = e.download_state()
    order = np.argsort(st["ids"])
    final[fuse] = {key: st[key][order] for key in ("r", "v", "q", "D")}
    print(f"fuse_integration={fuse}: N={N} steps={steps} wall={wall:.2f}s ({N*steps/wall:.3e} updates/s)  E range [{E.min():.6f}, {E.max():.6f}] "
          f"drift/|E|={(E[-1]-E[0])/abs(E[0]):.3e} fluct/|E|={(E.max()-E.min())/abs(E[0]):.3e}  "
          f"{e.get_option('verlet_builds')} list builds, ids intact: {bool(np.array_equal(st['ids'][order], big.ids))}")
    e.close()
print("fused and separate integration, final r v q D bitwise equal:", all(np.array_equal(final[1][key], final[0][key]) for key in final[1]))
